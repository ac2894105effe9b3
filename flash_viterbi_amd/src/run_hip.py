#!/usr/bin/env python3
"""run_hip.py — bench driver for the MI355X host programs; counterpart of the reference's
src/run.py (same `parameters` / `file_names` lists, same config patching, same
`time:` / `memory:` scraping, same CSV columns first) with three additions the
reference lacks: the decoded path is kept and hashed, extra stderr statistics are
recorded, and --check compares the path with a binary built from the reference's own
source when /root/reference is present.

  python3 run_hip.py                 # every parameter set x every program
  python3 run_hip.py --gen           # generate missing input files first (generate_data counterpart)
  python3 run_hip.py --check         # also run the reference program and compare paths
"""
import csv
import hashlib
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
              "time", "memory", "path_md5", "cells_per_s", "gpu_ms", "model_upload_s", "device_bytes", "ref_path_equal"]


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
    env = dict(os.environ, FV_BIN_CACHE=os.environ.get("FV_BIN_CACHE", "1"))
    res = subprocess.run([modified], capture_output=True, text=True, env=env)
    if res.returncode != 0:
        raise RuntimeError(f"run ERROR ({res.returncode}): {res.stderr}")
    out, err = res.stdout, res.stderr
    info = {"time": re.search(r"time: ([\d.]+)", out).group(1),            # reference run.py:75
            "memory": re.search(r"memory: (\d+)", out).group(1),            # reference run.py:76
            "path": re.search(r"path: \[([^\]]*)\]", out).group(1).split()}
    for key in ("cells_per_s", "gpu_ms", "model_upload_s", "device_bytes"):
        m = re.search(rf"{key}: (\S+)", err)
        info[key] = m.group(1) if m else ""
    print(f"{filename} Time: {info['time']}, Memory: {info['memory']}")
    return info


def reference_path(filename, p):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_ref
    kind = "flashbs" if "BS" in filename else "flash"
    try:
        exe = build_ref.build(kind, p["K_STATE"], p["obserRouteLEN"], p["prob"], p["MAX_THREADS"],
                              p["BeamSearchWidth"] if kind == "flashbs" else None, M=p["T_STATE"])
    except FileNotFoundError:
        return None
    return [str(x) for x in build_ref.run(exe, data_path)["path"]]


def main():
    os.makedirs(result_path, exist_ok=True)
    os.makedirs(data_path, exist_ok=True)
    fvbuild.build_host()
    fvbuild.build_hip()
    check = "--check" in sys.argv
    for filename in file_names:
        csv_name = result_path + filename + "_result.csv"
        new = not os.path.exists(csv_name)
        with open(csv_name, "a", encoding="utf-8", newline="") as fh:
            w = csv.writer(fh)
            if new:
                w.writerow(CSV_HEADER)
            for p in parameters:
                if "--gen" in sys.argv:
                    ensure_inputs(p)
                info = run_c(filename, p)
                same = ""
                if check:
                    ref = reference_path(filename, p)
                    same = "" if ref is None else str(ref == info["path"])
                w.writerow([datetime.now().strftime("%Y-%m-%d %H:%M:%S"), p["K_STATE"], p["T_STATE"], p["obserRouteLEN"],
                            p["prob"], p.get("MAX_THREADS", "N/A"), p.get("BeamSearchWidth", "N/A"), info["time"],
                            info["memory"], hashlib.md5((" ".join(info["path"])).encode()).hexdigest(),
                            info["cells_per_s"], info["gpu_ms"], info["model_upload_s"], info["device_bytes"], same])
                fh.flush()


if __name__ == "__main__":
    main()
