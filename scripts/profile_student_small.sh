#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_student_small.sh TAG
# rocprofv3 --kernel-trace --stats of 200 steps of the B = 512 arxiv students (MLP 128-256-256-40, MLP3w4 128-1024-1024-40): per-kernel
# statistics + the launch-by-launch timeline of steady-state steps.  Results in gpurun_out/prof_TAG/ -- copy what is to be judged to profiles/.
set -u
TAG=${1:-r03}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
for w in mlp w4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$w -- python $ROOT/scripts/trace_student_small.py $w > /dev/null 2>&1
  f=$(ls /tmp/st_$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_student_${w}_kernel_stats.csv"
  { echo "# rocprofv3 --kernel-trace, scripts/trace_student_small.py $w: two steady-state steps, every kernel (start us, gap to the previous kernel, duration us)";
    python $ROOT/scripts/trace_timeline3.py /tmp/st_$w 1500 30; } > "$OUT/${TAG}_student_${w}_timeline.txt"
done
ls -la "$OUT"
cat "$OUT/${TAG}_student_mlp_timeline.txt"
