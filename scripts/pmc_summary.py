"""Per-kernel means of every counter in a rocprofv3 --pmc CSV tree: python scripts/pmc_summary.py DIR [name-filter]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for kname, cs in sorted(acc.items()):
    if flt and flt not in kname:
        continue
    print(kname)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
