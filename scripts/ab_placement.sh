#!/bin/bash
# Interleaved A/B of the placement tuner (ops.placed_for_gather) on the default bench's teacher forward: GLNN_PLACEMENT_CANDIDATES = 1 (off) vs 6
for i in 1 2 3 4; do
  for c in 1 12; do
    GLNN_PLACEMENT_CANDIDATES=$c python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-leg --reorder none --no-clustered-leg --no-small-students --no-xl-leg --no-arxiv-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('candidates $c', round(d['ms_per_step'],2), d['verified'], [(x['d'], round(x['ms'],3)) for x in d['roofline']['launches']], d.get('placement'))"
  done
done
