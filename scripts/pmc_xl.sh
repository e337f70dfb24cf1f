#!/bin/bash
# HBM traffic of the config-4 (XL shard) aggregation launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_xl
mkdir -p "$OUT"
CMD="python bench.py --workload xl --steps 3 --warmup 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
python scripts/pmc_to_json.py "$OUT/fetch" "$OUT/write" "$OUT/pmc_traffic_xl.json" "$OUT/r02_pmc_hbm_traffic_xl.csv"
rm -rf "$OUT/fetch" "$OUT/write"
