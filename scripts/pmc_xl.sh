#!/bin/bash
# usage (GPU box, repo root): scripts/pmc_xl.sh TAG -> gpurun_out/pmc_xl_TAG/
# BASELINE configs[4] (bench.py --workload xl: one rank's 3-layer forward, peers emulated): rocprofv3 kernel stats + HBM traffic of its
# launches (FETCH_SIZE x2 per MI355X_MICROARCH.md + WRITE_SIZE, separate --pmc passes, --kernel-trace only).  Every layer runs as
# 4 chunk launches per forward; the summary multiplies the per-launch means by 4 and sets them against the algorithmic bytes.
set -u
TAG=${1:-r04}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_xl_$TAG
mkdir -p "$OUT"
CMD="python bench.py --workload xl --steps 2 --warmup 1 --no-verify"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
f=$(ls "$OUT"/stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_xl_kernel_stats.csv"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
python scripts/pmc_to_json.py "$OUT/fetch" "$OUT/write" "$OUT/pmc_traffic_xl.json" "$OUT/${TAG}_pmc_hbm_traffic_xl.csv"
python - "$OUT/pmc_traffic_xl.json" <<'PY' | tee "$OUT/summary.txt"
import json, sys
d = json.load(open(sys.argv[1]))["per_launch_bytes"]
rows, nnz = 12_500_000, 250_000_000
alg = {"spmm_csr_kernel<LPR=32,U=8,SAGE_GCN>": ("layer 1 aggregation D=128", nnz * (4 * 128 + 4) + rows * (8 * 128 + 8)),
       "sage_fused_kernel<LPR=64,U=8>": ("layer 2 fused D=256 -> 256 -> 47", nnz * (4 * 256 + 4) + rows * (4 * 256 + 4 * 47 + 8)),
       "spmm_csr_kernel<LPR=16,U=8,SAGE_GCN>": ("layer 3 aggregation D=47", nnz * (4 * 47 + 4) + rows * (8 * 47 + 8))}
for k, (what, b) in alg.items():
    if k in d:
        t = 4 * d[k]["total"]
        print(f"{what:36s} counter bytes per forward {t / 1e9:8.2f} GB  algorithmic {b / 1e9:8.2f} GB  ratio {t / b:.3f}")
PY
head -8 "$OUT/${TAG}_xl_kernel_stats.csv" | cut -c1-150
rm -rf "$OUT/fetch" "$OUT/write" "$OUT/stats"
