"""PMC summary of the FLASH-BS step kernels from tools/prof_beam_pmc.sh passes (gpurun_out/bp_<workload>_{1,2,3}):
per kernel and launch shape (passes sharing the launch = Grid_Size / single-pass grid) FETCH_SIZE, WRITE_SIZE,
TCC_HIT_sum, TCC_MISS_sum, converted to bytes with the factors calibrated on this very access pattern
(tools/micro/gather_calib.hip -> profiles/r03_gather_calibration.json: FETCH_SIZE x 2, 128 B per TCC miss).
Appends the single-pass figures to profiles/traffic.json and writes profiles/r03_pmc_beam_<workload>.csv.
   python tools/pmc_beam_summary.py cfg4 [cfg5]"""
import csv, datetime, glob, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {"cfg4": dict(K=16384, B=256), "cfg5": dict(K=65536, B=1024)}
cal = json.load(open(os.path.join(ROOT, "profiles", "r03_gather_calibration.json")))["patterns"]
tj_path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tj_path))
for wl in sys.argv[1:]:
    K, B = SHAPES[wl]["K"], SHAPES[wl]["B"]
    rows = {}
    for d in (1, 2, 3):
        for path in glob.glob(os.path.join(ROOT, "gpurun_out", f"bp_{wl}_{d}", "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for r in csv.DictReader(f):
                    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    if not name.startswith("fvb::beam_step"):
                        continue
                    # grid = column panels x passes x 1024 threads; panels = K/64 (beam_step) or K/128 (beam_step_q16)
                    panels = (K + 63) // 64 if name == "fvb::beam_step" else (K + 127) // 128
                    npass = int(r["Grid_Size"]) // (panels * 1024)
                    rows.setdefault((name, npass, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    out_csv = os.path.join(ROOT, "profiles", f"r03_pmc_beam_{wl}.csv")
    with open(out_csv, "w") as f:
        f.write("kernel,passes_in_launch,counter,dispatches,mean,median,min,max\n")
        for (k, n, c), v in sorted(rows.items()):
            f.write(f"\"{k}\",{n},{c},{len(v)},{statistics.mean(v):.3f},{statistics.median(v):.3f},{min(v):.3f},{max(v):.3f}\n")
    for name in ("fvb::beam_step", "fvb::beam_step_q16"):
        m = {c: statistics.median(v) for (k, n, c), v in rows.items() if k == name and n == 1}
        if not {"FETCH_SIZE", "WRITE_SIZE", "TCC_MISS_sum"} <= set(m):
            continue
        pat = "rows8" if name == "fvb::beam_step" else "rows4"
        ff = cal[pat]["fetch_factor"]
        table = B * K * (8 if name == "fvb::beam_step" else 2)
        ent = {"workload": wl, "launch": "single pass (the whole-sequence pass's launches)", "FETCH_SIZE_KB_raw": m["FETCH_SIZE"],
               "WRITE_SIZE_KB": m["WRITE_SIZE"], "TCC_MISS_sum": m["TCC_MISS_sum"], "TCC_HIT_sum": m.get("TCC_HIT_sum"),
               "fetch_factor_calibrated": ff, "hbm_bytes_per_launch": int(m["FETCH_SIZE"] * 1024 * ff + m["WRITE_SIZE"] * 1024),
               "tcc_miss_bytes_per_launch": int(m["TCC_MISS_sum"] * cal[pat]["bytes_per_TCC_MISS"]),
               "table_bytes_the_sweep_reads": table, "algorithmic_bytes": 4 * B * K}
        tj["by_kernel"][f"{name}@{wl}"] = ent
        print(name, wl, json.dumps(ent))
tj["beam_note"] = ("FLASH-BS entries (<kernel>@<workload>): median over the single-pass launches of one decode, three separate --pmc passes "
                   "(tools/prof_beam_pmc.sh), converted with factors calibrated on the kernels' own gather patterns "
                   "(tools/micro/gather_calib.hip, profiles/r03_gather_calibration.json: both patterns read FETCH_SIZE x 2.0 and 128 B per TCC miss, "
                   "within 1 %); " + datetime.date.today().isoformat())
json.dump(tj, open(tj_path, "w"), indent=1)
