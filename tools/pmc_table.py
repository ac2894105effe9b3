"""Per-kernel means of every counter found under the given rocprofv3 --pmc output directories.
  python tools/pmc_table.py gpurun_out/pmcA gpurun_out/pmcB [substring of kernel name ...]"""
import csv, glob, os, statistics, sys
dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
subs = [a for a in sys.argv[1:] if not os.path.isdir(a)]
rows = {}
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
kernels = sorted({k for k, _ in rows})
for k in kernels:
    if k.startswith("__amd") or (subs and not any(s in k for s in subs)):
        continue
    print(k[:110])
    for (kk, c), v in sorted(rows.items()):
        if kk == k:
            print(f"   {c:28s} n={len(v):5d} mean={statistics.mean(v):14.1f}")
