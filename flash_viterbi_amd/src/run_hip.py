#!/usr/bin/env python3
"""run_hip.py — bench driver for the MI355X host programs; counterpart of the reference's
src/run.py (same `parameters` / `file_names` lists, same config patching, same
`time:` / `memory:` scraping, same CSV columns first) with the additions SURVEY 8(f-2)
asks for: the decoded path is kept and hashed, the GPU count, cells/s, the roofline
fraction of the dominant kernel and further stderr statistics are recorded, and a
path-equality flag against the reference program can be filled in from a file of
expected path hashes (plain JSON; tools/ref_paths.py writes one by running binaries
compiled from the reference's own sources — that checker lives outside this package).

  python3 run_hip.py                      # every parameter set x every program
  python3 run_hip.py --gen                # generate missing input files first (generate_data counterpart)
  python3 run_hip.py --ref-md5 FILE.json  # fill ref_path_equal from {"<program>|K|M|T|prob|N|B": "<md5 of the path>"}
"""
import csv
import hashlib
import json
import os
import re
import subprocess
import sys
from datetime import datetime

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
sys.path.insert(0, ROOT)

from flash_viterbi_amd import build as fvbuild  # noqa: E402

base_path = HERE + os.sep
data_path = os.path.join(HERE, "data") + os.sep
result_path = os.path.join(HERE, "result") + os.sep
file_names = ["FLASH_Viterbi_hip", "FLASH_BS_Viterbi_hip"]
parameters = [
    {"K_STATE": 3965, "T_STATE": 50, "obserRouteLEN": 256, "prob": 0.112, "MAX_THREADS": 8, "BeamSearchWidth": 32},
    {"K_STATE": 3965, "T_STATE": 50, "obserRouteLEN": 256, "prob": 0.169, "MAX_THREADS": 8, "BeamSearchWidth": 32},
]
SEED = 12          # generate_data -s
CSV_HEADER = ["timestamp", "K_STATE", "T_STATE", "obserRouteLEN", "prob", "MAX_THREADS", "BeamSearchWidth",
              "time", "memory", "n_gpus", "cells_per_s", "roofline_frac", "path_md5", "ref_path_equal",
              "gpu_ms", "model_upload_s", "device_bytes"]


def patch_config(text, filename, p):
    """The substitutions of the reference's run.py:29-47, applied to our source."""
    text = re.sub(r"#define K_STATE \d+", f"#define K_STATE {p['K_STATE']}", text)
    text = re.sub(r"#define T_STATE \d+", f"#define T_STATE {p['T_STATE']}", text)
    text = re.sub(r"#define obserRouteLEN \d+", f"#define obserRouteLEN {p['obserRouteLEN']}", text)
    text = re.sub(r"const float prob = \d+\.\d+;", f"const float prob = {p['prob']};", text)
    text = re.sub(r'const char data_path\[\] = "[^"]*";', f'const char data_path[] = "{data_path}";', text)
    text = re.sub(r"#define MAX_THREADS \d+", f"#define MAX_THREADS {p['MAX_THREADS']}", text)
    if "BS" in filename:
        text = re.sub(r"const int BeamSearchWidth = \d+;", f"const int BeamSearchWidth = {p['BeamSearchWidth']};", text)
    s = str(p["prob"])
    digits = len(s.split(".")[1]) if "." in s else 0
    return re.sub(r"prob%\.\d+f", f"prob%.{digits}f", text)


def ensure_inputs(p):
    from flash_viterbi_amd.generate_data import data_script
    K, T, prob = p["K_STATE"], p["obserRouteLEN"], p["prob"]
    if all(os.path.isfile(os.path.join(data_path, f"{k}_K{K}_T{T}_prob{prob}.txt")) for k in ("A", "B", "Pi", "ob")):
        return
    print(f"generating inputs K={K} T={T} prob={prob} into {data_path}")
    A, B, Pi = data_script.make_model64(K, p["T_STATE"], SEED, prob)
    data_script.write_files(data_path, K, T, prob, A, B, Pi, data_script.make_observations(T, p["T_STATE"], SEED))


def run_c(filename, p):
    with open(base_path + filename + ".c") as f:
        content = patch_config(f.read(), filename, p)
    modified = base_path + filename + "_modified"
    with open(modified + ".c", "w") as f:
        f.write(content)
    res = subprocess.run(fvbuild.program_cc(modified + ".c", modified), capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"compile ERROR: {res.stderr}")
    # raw float32 caches next to the text files: safe by default because a cache is bound to the size and mtime
    # of the text it was parsed from (include/flashvit_host.h, fvh_read_bin_src) and re-made when they differ
    env = dict(os.environ, FV_BIN_CACHE=os.environ.get("FV_BIN_CACHE", "1"))
    if "--gpus" in sys.argv:     # devices of the one host process (FV_NGPUS, fv_create_multi); the n_gpus column reports it
        env["FV_NGPUS"] = sys.argv[sys.argv.index("--gpus") + 1]
    res = subprocess.run([modified], capture_output=True, text=True, env=env)
    if res.returncode != 0:
        raise RuntimeError(f"run ERROR ({res.returncode}): {res.stderr}")
    out, err = res.stdout, res.stderr
    info = {"time": re.search(r"time: ([\d.]+)", out).group(1),            # reference run.py:75
            "memory": re.search(r"memory: (\d+)", out).group(1),            # reference run.py:76
            "path": re.search(r"path: \[([^\]]*)\]", out).group(1).split()}
    for key in ("cells_per_s", "gpu_ms", "model_upload_s", "device_bytes", "n_gpus", "roofline_frac"):
        m = re.search(rf"{key}: (\S+)", err)
        info[key] = m.group(1) if m else ""
    print(f"{filename} Time: {info['time']}, Memory: {info['memory']}")
    return info


def ref_key(filename, p):
    """Key of one (program, parameter set) in a --ref-md5 file."""
    beam = p["BeamSearchWidth"] if "BS" in filename else 0
    return "|".join(str(x) for x in (filename, p["K_STATE"], p["T_STATE"], p["obserRouteLEN"], p["prob"], p["MAX_THREADS"], beam))


def path_md5(path_tokens):
    return hashlib.md5((" ".join(str(x) for x in path_tokens)).encode()).hexdigest()


def main():
    os.makedirs(result_path, exist_ok=True)
    os.makedirs(data_path, exist_ok=True)
    fvbuild.build_host()
    fvbuild.build_hip()
    ref_md5 = {}
    if "--ref-md5" in sys.argv:
        with open(sys.argv[sys.argv.index("--ref-md5") + 1]) as fh:
            ref_md5 = json.load(fh)
    for filename in file_names:
        csv_name = result_path + filename + "_result.csv"
        new = not os.path.exists(csv_name)
        if not new:
            # a results file written under another column layout is set aside, never appended to (its rows would sit
            # under the wrong header)
            with open(csv_name, encoding="utf-8", newline="") as fh:
                first = fh.readline().rstrip("\r\n")
            if first != ",".join(CSV_HEADER):
                n = 1
                while os.path.exists(f"{csv_name[:-4]}.v{n}.csv"):
                    n += 1
                os.rename(csv_name, f"{csv_name[:-4]}.v{n}.csv")
                print(f"{csv_name}: header differs from this version's; kept as {csv_name[:-4]}.v{n}.csv", file=sys.stderr)
                new = True
        with open(csv_name, "a", encoding="utf-8", newline="") as fh:
            w = csv.writer(fh)
            if new:
                w.writerow(CSV_HEADER)
            for p in parameters:
                if "--gen" in sys.argv:
                    ensure_inputs(p)
                info = run_c(filename, p)
                md5 = path_md5(info["path"])
                want = ref_md5.get(ref_key(filename, p))
                same = "" if want is None else str(want == md5)
                w.writerow([datetime.now().strftime("%Y-%m-%d %H:%M:%S"), p["K_STATE"], p["T_STATE"], p["obserRouteLEN"],
                            p["prob"], p.get("MAX_THREADS", "N/A"), p.get("BeamSearchWidth", "N/A"), info["time"],
                            info["memory"], info["n_gpus"], info["cells_per_s"], info["roofline_frac"], md5, same,
                            info["gpu_ms"], info["model_upload_s"], info["device_bytes"]])
                fh.flush()


if __name__ == "__main__":
    main()
