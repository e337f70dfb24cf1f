#!/bin/bash
# Interleaved A/B of several builds of libglnn_hip.so on the default bench's teacher forward (per-layer ms): scripts/ab_multi.sh lib1.so lib2.so ... ("default" = the in-tree build)
for i in 1 2 3; do
  for v in default "$@"; do
    if [ "$v" = default ]; then unset GLNN_LIB_PATH; else export GLNN_LIB_PATH=$PWD/$v; fi
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-leg --reorder none --no-clustered-leg --no-verify --no-small-students --no-xl-leg --no-arxiv-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],2), [(x['d'], round(x['ms'],3)) for x in d['roofline']['launches']])"
  done
done
unset GLNN_LIB_PATH
