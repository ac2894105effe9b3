"""Summary of tools/gather_calib.sh: per access pattern the bytes actually read against FETCH_SIZE (KB) and TCC_MISS_sum,
and the factors that turn the counters into bytes; written to profiles/r03_gather_calibration.json."""
import csv, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
known = {"stream16": B * K * 8, "rows8": B * K * 8, "rows4": B * K * 2}
rows = {}
for d in ("gcal1", "gcal2", "gcal3"):
    for path in glob.glob(os.path.join(ROOT, "gpurun_out", d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                rows.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
out = {"K": K, "B": B, "patterns": {}}
for pat, nbytes in known.items():
    m = {c: statistics.median(v) for (k, c), v in rows.items() if k == pat}
    if not m:
        continue
    ent = {"bytes_read": nbytes, **{c: m[c] for c in sorted(m)}}
    if "FETCH_SIZE" in m:
        ent["bytes_per_FETCH_SIZE_KB"] = nbytes / m["FETCH_SIZE"]
        ent["fetch_factor"] = nbytes / (m["FETCH_SIZE"] * 1024)          # multiply FETCH_SIZE (KB) * 1024 by this
    if "TCC_MISS_sum" in m:
        ent["bytes_per_TCC_MISS"] = nbytes / m["TCC_MISS_sum"]
    out["patterns"][pat] = ent
    print(pat, json.dumps(ent))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_gather_calibration.json"), "w"), indent=1)
