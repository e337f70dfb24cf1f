"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/pmc_traffic.json.
usage: python scripts/pmc_to_json.py <fetch_dir> <write_dir> <out.json> <csv_out>"""
import collections, csv, glob, json, re, sys

def short(name):
    m = re.search(r"(spmm_csr_kernel|sage_fused_kernel)<(\d+), (\d+)", name)
    if not m:
        return None
    return f"{m.group(1)}<LPR={m.group(2)},U={m.group(3)}" + (",SAGE_GCN>" if m.group(1) == "spmm_csr_kernel" else ">")

agg = collections.defaultdict(dict)
rows_out = ["kernel,counter,dispatches,mean_reported_KiB,corrected_GB"]
for d, counter, corr in ((sys.argv[1], "FETCH_SIZE", 2.0), (sys.argv[2], "WRITE_SIZE", 1.0)):
    f = sorted(glob.glob(d + "/*/*counter_collection.csv"))[-1]
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter:
            vals[k].append(float(r["Counter_Value"]))
    for k, v in sorted(vals.items()):
        m = sum(v) / len(v)
        agg[k][counter] = m * 1024 * corr
        rows_out.append(f'"{k}",{counter},{len(v)},{m:.1f},{m * 1024 * corr / 1e9:.2f}')
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 2 --warmup 1 "
                 "--no-cpu-baseline`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); KiB units",
       "per_launch_bytes": {k: {"fetch": a.get("FETCH_SIZE"), "write": a.get("WRITE_SIZE"),
                                "total": (a.get("FETCH_SIZE") or 0) + (a.get("WRITE_SIZE") or 0)} for k, a in agg.items()}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
open(sys.argv[4], "w").write("\n".join(rows_out) + "\n")
print(json.dumps(out["per_launch_bytes"], indent=1))
