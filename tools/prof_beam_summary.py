"""Summary of gpurun_out/prof_beam/beam_results.db (written by tools/prof_beam.sh)."""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_beam/beam_results.db")
cur = db.cursor()
for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print(f"{r[0][:60]:60s} calls {r[1]:6d} total_us {r[2]:12.1f} avg_us {r[3]:9.2f} {r[4]:5.1f}%")
rows = list(cur.execute("select name, start, end from kernels order by start"))
sel = np.array([(e - s) / 1e3 for n, s, e in rows if "topb_select" in n])
if len(sel):
    cut = 3 * np.median(sel)
    print("topb_select percentiles 0/25/50/75/100:", np.percentile(sel, [0, 25, 50, 75, 100]))
    print(f"  plain {int((sel < cut).sum())} x {sel[sel < cut].mean():.2f} us; with replay {int((sel >= cut).sum())} x "
          f"{(sel[sel >= cut].mean() if (sel >= cut).any() else 0):.2f} us")
