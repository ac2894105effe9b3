"""Per-generation break-down of the first FLASH-BS decode in a tools/prof_beam.sh trace.  python tools/prof_beam_gens.py [db]"""
import collections, sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_beam/beam_results.db")
rows = list(db.cursor().execute("select name, start, end from kernels order by start"))
names = [r[0] for r in rows]; st = np.array([r[1] for r in rows], dtype=np.int64); en = np.array([r[2] for r in rows], dtype=np.int64)
inits = [i for i, n in enumerate(names) if "init_rows" in n]
# generations of the first decode: consecutive init_rows until the gap pattern repeats (6 generations at T=256,N=8; 10 at T=1024)
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for g in range(ng):
    a = inits[g]; b = inits[g + 1] if g + 1 < len(inits) else len(names)
    wall = (en[a:b].max() - st[a]) / 1e3
    tot = collections.defaultdict(float); cnt = collections.Counter()
    for i in range(a, b):
        k = names[i].split("(")[0].replace("void ", "")[:40]; tot[k] += (en[i] - st[i]) / 1e3; cnt[k] += 1
    print(f"gen {g}: {b - a} kernels, wall {wall:.0f} us")
    for k, v in sorted(tot.items(), key=lambda x: -x[1])[:6]: print(f"     {k:40s} {cnt[k]:4d} x {v / cnt[k]:7.1f} us = {v:8.0f}")
