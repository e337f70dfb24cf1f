#!/bin/bash
# Interleaved A/B of two builds of libglnn_hip.so on the default bench (teacher forward per-layer ms) and, with "xl", on the XL shard:
#   scripts/ab_lib.sh variants/libglnn_prev.so [xl]
PREV=$1
for i in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then export GLNN_LIB_PATH=$PREV; else unset GLNN_LIB_PATH; fi
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-leg --reorder none --no-clustered-leg --no-verify --no-small-students --no-xl-leg --no-arxiv-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],2), [(x['d'], round(x['ms'],3)) for x in d['roofline']['launches']])"
  done
done
if [ "$2" = xl ]; then
  for i in 1 2; do
    for v in prev new; do
      if [ $v = prev ]; then export GLNN_LIB_PATH=$PREV; else unset GLNN_LIB_PATH; fi
      python bench.py --workload xl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v xl', round(d['ms_per_step'],2))"
    done
  done
fi
unset GLNN_LIB_PATH
