"""Fold the three passes of scripts/pmc_l2.sh into one JSON: per kernel the mean counter values per launch, exact fabric read
bytes (32 / 64 / 128-byte requests), L2 hit rate, and the calibration on the 4 GiB copy.  usage: pmc_l2_summary.py <dir> <out.json>"""
import collections, csv, glob, json, re, sys

root, out_path = sys.argv[1], sys.argv[2]


def short(name):
    m = re.search(r"(spmm_csr_kernel|sage_fused_kernel|spmm_gpr_kernel)<(\d+), (\d+)", name)
    if m:
        return f"{m.group(1)}<LPR={m.group(2)},U={m.group(3)}" + (">" if m.group(1) == "sage_fused_kernel" else ",SAGE_GCN>")
    if "move_rows_kernel" in name:
        return "copy"
    return None


vals = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("rd", "hit", "wr"):
    files = sorted(glob.glob(f"{root}/{sub}/*/*counter_collection.csv"))
    if not files:
        continue
    for r in csv.DictReader(open(files[-1])):
        k = short(r["Kernel_Name"])
        if k:
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in vals.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"launches": max(len(v) for v in cs.values()), "counters_mean_per_launch": m}
    if "TCC_EA0_RDREQ_sum" in m:
        n32, n64, n128 = m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0)
        e["fabric_read_bytes_by_size"] = 32 * n32 + 64 * n64 + 128 * n128
        e["fabric_read_requests_other_size"] = m["TCC_EA0_RDREQ_sum"] - n32 - n64 - n128
    if "TCC_HIT_sum" in m:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        e["l2_miss_bytes_at_128B_lines"] = 128 * m["TCC_MISS_sum"]
        e["read_requests_routed_to_local_memory"] = m.get("TCC_EA0_RDREQ_DRAM_sum")
    if "TCC_EA0_WRREQ_sum" in m:
        e["fabric_write_bytes"] = 64 * m.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (m["TCC_EA0_WRREQ_sum"] - m.get("TCC_EA0_WRREQ_64B_sum", 0))
    res[k] = e
if "copy" in res and "fabric_read_bytes_by_size" in res["copy"]:
    true = 4.0 * (1 << 30) + 8.0 * (1 << 22)          # the matrix + the int64 row ids
    res["copy"]["known_bytes_read"] = true
    res["copy"]["calibration_read_ratio"] = res["copy"]["fabric_read_bytes_by_size"] / true
    if "fabric_write_bytes" in res["copy"]:
        res["copy"]["calibration_write_ratio"] = res["copy"]["fabric_write_bytes"] / true
out = {"source": "scripts/pmc_l2.sh: rocprofv3 --kernel-trace --pmc passes of scripts/pmc_target.py (4 GiB calibration copy + two products-shaped "
                 "teacher forwards); bytes from the per-size request counters; no MALL hit counter exists among the TCC counters",
       "kernels": res}
json.dump(out, open(out_path, "w"), indent=1)
for k, e in res.items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items() if a != "counters_mean_per_launch"})
