#!/bin/bash
# usage (GPU box, repo root): scripts/cls_pmc.sh -> SQ / TCP counters of cls_fwd_kernel (one --pmc pass per group; no tracing options besides --kernel-trace)
export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/clspmc; CLS_PROBE_ONE="2 1" rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/clspmc -- python scripts/cls_probe.py > /dev/null 2>&1
  f=$(find /tmp/clspmc -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "cls_fwd_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = v[len(v) // 2:]
    print(f"{k}: mean {sum(v) / len(v):.0f} over {len(v)} launches")
PY
done
