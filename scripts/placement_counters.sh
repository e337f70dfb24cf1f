#!/bin/bash
# usage (GPU box, repo root): scripts/placement_counters.sh -> gpurun_out/r06_placement_counters.txt
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r06_placement_counters.txt
: > "$OUT"
for grp in "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES TCC_BUBBLE_sum"; do
  rm -rf /tmp/plc; timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/plc -- python scripts/placement_counters.py > /tmp/plc.log 2>&1
  grep "ms per buffer" /tmp/plc.log >> "$OUT"
  f=$(find /tmp/plc -name "*counter_collection.csv" | head -1)
  python - "$f" >> "$OUT" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sage_fused" in r["Kernel_Name"]]
byd = collections.OrderedDict()
for r in rows:
    byd.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ds = list(byd.values())[-6:]
for name in ds[0]:
    fast = [d[name] for d in ds[:3]]
    slow = [d[name] for d in ds[3:]]
    mf, msl = sum(fast) / 3, sum(slow) / 3
    print(f"  {name:40s} fast {mf:16.0f}   slow {msl:16.0f}   slow / fast {msl / mf if mf else float('nan'):.3f}")
PY
done
cat "$OUT"
