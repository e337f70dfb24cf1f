#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_round.sh TAG
# HBM traffic of the aggregation launches: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench command,
# folded into gpurun_out/pmc_TAG/pmc_traffic.json + a per-kernel CSV by scripts/pmc_to_json.py (FETCH_SIZE doubled per
# MI355X_MICROARCH.md).  Copy both into profiles/ to make them the constants bench.py reports as roofline.traffic.
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --reorder none --no-clustered-leg --no-verify --no-small-students --no-xl-leg --no-arxiv-leg --no-chunked-leg"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
python scripts/pmc_to_json.py "$OUT/fetch" "$OUT/write" "$OUT/pmc_traffic.json" "$OUT/${TAG}_pmc_hbm_traffic.csv"
rm -rf "$OUT/fetch" "$OUT/write"
