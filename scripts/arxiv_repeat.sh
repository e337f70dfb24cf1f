#!/bin/bash
# The arxiv-shaped forward (BASELINE configs[1]) five times, each in its own process: ms per forward and per aggregation launch.
#   scripts/arxiv_repeat.sh [variants/libglnn_x.so]      (an argument = that library build instead of the in-tree one)
[ -n "$1" ] && export GLNN_LIB_PATH=$1
for i in 1 2 3 4 5; do
  python bench.py --workload arxiv --no-cpu-baseline 2>/dev/null | grep '^DETAIL ' | tail -1 | cut -c8- | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(round(d['ms_per_step'],3), [(x['d'], round(x['avg_ms'],3)) for x in d['roofline']['all_aggregation_launches']])"
done
