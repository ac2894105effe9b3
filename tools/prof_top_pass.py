"""Break-down of the whole-sequence pass of the first decode in a rocprofv3 kernel trace (tools/prof_beam.sh):
   python tools/prof_top_pass.py [gpurun_out/prof_beam/beam_results.db]"""
import collections, sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_beam/beam_results.db")
rows = list(db.cursor().execute("select name, start, end from kernels order by start"))
names = [r[0] for r in rows]; st = np.array([r[1] for r in rows]); en = np.array([r[2] for r in rows])
inits = [i for i, n in enumerate(names) if "init_rows" in n]
i0, i1 = inits[0], inits[1]
print(f"whole-sequence pass: {i1 - i0} kernels, wall {(en[i1 - 1] - st[i0]) / 1e6:.3f} ms; busy {(en[i0:i1] - st[i0:i1]).sum() / 1e6:.3f} ms")
tot = collections.defaultdict(float); cnt = collections.Counter()
for i in range(i0, i1):
    k = names[i][:48]; tot[k] += (en[i] - st[i]) / 1e3; cnt[k] += 1
for k, v in sorted(tot.items(), key=lambda x: -x[1]):
    print(f"  {k:48s} {cnt[k]:5d} x {v / cnt[k]:8.1f} us = {v / 1e3:8.3f} ms")
sel = np.array([(en[i] - st[i]) / 1e3 for i in range(i0, i1) if "topb_select" in names[i]])
print("  select durations (us), histogram:", np.histogram(sel, bins=[0, 9, 12, 20, 30, 50, 80, 120, 200, 400, 1000, 100000]))
print("  first 120:", np.round(sel[:120]).astype(int).tolist())
stp = np.array([(en[i] - st[i]) / 1e3 for i in range(i0, i1) if "beam_step" in names[i]])
print("  step percentiles 0/25/50/75/95/100:", np.round(np.percentile(stp, [0, 25, 50, 75, 95, 100]), 1))
