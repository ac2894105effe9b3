#!/usr/bin/env python3
"""Checker for flash_viterbi_amd/src/run_hip.py (kept OUT of the product package: it drives binaries built
from the reference's own sources through oracle/build_ref.py).

Runs the reference program of every (program, parameter set) run_hip.py would run, on the same input files,
and writes {"<program>|K|M|T|prob|N|B": "<md5 of the decoded path>"} for `run_hip.py --ref-md5 FILE`.

  python3 tools/ref_paths.py OUT.json            # needs /root/reference (or prebuilt oracle/_ref binaries)
"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import build_ref  # noqa: E402


def load_run_hip():
    spec = importlib.util.spec_from_file_location("run_hip", os.path.join(ROOT, "flash_viterbi_amd", "src", "run_hip.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def reference_md5s(run_hip, programs=None, parameters=None, data_dir=None):
    out = {}
    for filename in programs or run_hip.file_names:
        kind = "flashbs" if "BS" in filename else "flash"
        for p in parameters or run_hip.parameters:
            try:
                exe = build_ref.build(kind, p["K_STATE"], p["obserRouteLEN"], p["prob"], p["MAX_THREADS"],
                                      p["BeamSearchWidth"] if kind == "flashbs" else None, M=p["T_STATE"])
            except FileNotFoundError:
                continue
            path = build_ref.run(exe, data_dir or run_hip.data_path)["path"]
            out[run_hip.ref_key(filename, p)] = run_hip.path_md5(path)
    return out


if __name__ == "__main__":
    rh = load_run_hip()
    res = reference_md5s(rh)
    with open(sys.argv[1], "w") as fh:
        json.dump(res, fh, indent=1)
    print(f"{len(res)} reference paths hashed into {sys.argv[1]}")
