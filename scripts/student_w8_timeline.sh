#!/bin/bash
# usage (GPU box, repo root): scripts/student_w8_timeline.sh TAG -> gpurun_out/student_w8_TAG.txt: ms/step + two steady-state steps of the MLP3w8 products student, launch by launch
set -u
TAG=${1:-r04}
P=${2:-0.2}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/student_w8_$TAG.txt
python scripts/trace_student_any.py 100-2048-2048-47 4096 batch $P kl 2>&1 | grep "ms per step" > "$OUT"
rm -rf /tmp/st_w8; rocprofv3 --kernel-trace --output-format csv -d /tmp/st_w8 -- python scripts/trace_student_any.py 100-2048-2048-47 4096 batch $P kl > /dev/null 2>&1
python scripts/trace_timeline3.py /tmp/st_w8 2000 40 >> "$OUT"
cat "$OUT"
