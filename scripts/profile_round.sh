#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh TAG
# rocprofv3 --kernel-trace --stats of the default bench command (shortened) and of the sampled-block teacher training;
# the kernel_stats CSVs land in gpurun_out/prof_TAG/ -- copy the ones to be judged into profiles/.
set -u
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-leg --reorder none --no-clustered-leg --no-verify --no-small-students --no-xl-leg --no-arxiv-leg --no-chunked-leg > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train" -- python scripts/bench_train_sage.py ${2:-ogbn-arxiv} > "$OUT/train.log" 2>&1
for d in bench train; do
  f=$(ls "$OUT"/$d/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_${d}_kernel_stats.csv"
done
rm -rf "$OUT"/bench/*/*kernel_trace.csv "$OUT"/train/*/*kernel_trace.csv "$OUT"/bench/*/*.db "$OUT"/train/*/*.db 2>/dev/null
ls -la "$OUT"
head -25 "$OUT/${TAG}_train_kernel_stats.csv"
tail -3 "$OUT/train.log"
