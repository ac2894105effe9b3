"""Timeline of one full-state decode out of a rocprofv3 kernel trace of tools/time_kernels.py: per generation, per lock-step
wall time, busy time and launches.   python tools/prof_full_timeline.py results.db [decode index from the end]"""
import collections, sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = list(db.cursor().execute("select name, start, end from kernels order by start"))
names = [r[0] for r in rows]; st = np.array([r[1] for r in rows], dtype=np.int64); en = np.array([r[2] for r in rows], dtype=np.int64)
# a decode = from an init_rows whose next init_rows is > 200 kernels away (generation 0) up to the last backtrack before the next such
inits = [i for i, n in enumerate(names) if "init_rows" in n]
starts = [i for k, i in enumerate(inits) if k + 1 < len(inits) and inits[k + 1] - i > 200]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
i0 = starts[which]; i_end = starts[which + 1] if which + 1 < len(starts) and which != -1 else len(names)
gens = [i for i in inits if i0 <= i < i_end]
print(f"decode: kernels {i0}..{i_end}, wall {(en[i_end - 1] - st[i0]) / 1e6:.3f} ms, {len(gens)} generations")
for g, a in enumerate(gens):
    b = gens[g + 1] if g + 1 < len(gens) else i_end
    wall = (en[a:b].max() - st[a]) / 1e3
    busy = (en[a:b] - st[a:b]).sum() / 1e3
    cnt = collections.Counter(n.split("(")[0].replace("void ", "")[:44] for n in names[a:b])
    tot = collections.defaultdict(float)
    for i in range(a, b): tot[names[i].split("(")[0].replace("void ", "")[:44]] += (en[i] - st[i]) / 1e3
    print(f" gen {g}: {b - a:4d} kernels wall {wall:8.1f} us, summed kernel time {busy:8.1f} us")
    for k, v in sorted(tot.items(), key=lambda x: -x[1])[:5]: print(f"      {k:44s} {cnt[k]:4d} x {v / cnt[k]:6.1f} us")
