#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_l2.sh TAG
# What rocprofv3 exposes about the L2 <-> fabric traffic of the aggregation launches on gfx950, CALIBRATED on a copy of known
# size in the same passes (MI355X_MICROARCH.md: FETCH_SIZE tallies 128-byte requests as 64 bytes; the per-size request counters
# below give exact bytes).  Separate --pmc passes (TCC has 4 slots), --kernel-trace only.  There is NO Infinity-Cache (MALL)
# hit counter among the TCC counters (rocprofv3 -L): TCC_EA0_RDREQ_DRAM counts requests routed to local memory, hits included.
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_l2_$TAG
mkdir -p "$OUT"
CMD="python scripts/pmc_target.py"
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$OUT/rd" -- $CMD > "$OUT/rd.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum --output-format csv -d "$OUT/hit" -- $CMD > "$OUT/hit.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum --output-format csv -d "$OUT/wr" -- $CMD > "$OUT/wr.log" 2>&1
python scripts/pmc_l2_summary.py "$OUT" "$OUT/pmc_l2.json" | tee "$OUT/summary.txt"
for f in "$OUT"/*.log; do tail -n 3 "$f"; done
rm -rf "$OUT"/rd/*/*kernel_trace.csv "$OUT"/hit/*/*kernel_trace.csv "$OUT"/wr/*/*kernel_trace.csv 2>/dev/null
find "$OUT" -name "*.db" -delete 2>/dev/null
