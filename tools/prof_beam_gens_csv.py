"""Per-generation break-down of the last complete decode in a rocprofv3 kernel trace (CSV) of a FLASH-BS bench run:
   python tools/prof_beam_gens_csv.py [gpurun_out/prof_cfg4/beam_kernel_trace.csv]   — select and step durations per lock-step"""
import csv, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_cfg4/beam_kernel_trace.csv"
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
starts = [i for i, n in enumerate(names) if "clear_outputs" in n]
a, b = starts[-2], starts[-1]
inits = [i for i in range(a, b) if "init_rows" in names[i]] + [b]
for g in range(len(inits) - 1):
    sel = [round(dur[i], 1) for i in range(inits[g], inits[g + 1]) if "topb_select" in names[i]]
    stp = [(round(dur[i], 1), int(rows[i]["Grid_Size_Y"])) for i in range(inits[g], inits[g + 1]) if "beam_step" in names[i]]
    if g == 0:
        print(f"gen 0: {len(sel)} selects, median {sorted(sel)[len(sel) // 2]} us, sum {sum(sel):.0f}; steps sum {sum(d for d, _ in stp):.0f}")
        continue
    print(f"gen {g}: selects (us) {sel}  sum {sum(sel):.0f}")
    print(f"        steps (us, passes) {stp}  sum {sum(d for d, _ in stp):.0f}")
