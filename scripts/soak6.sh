#!/bin/bash
# Round 6 soak (GPU box, repo root): scripts/soak6.sh FIRST LAST -> gpurun_out/soak6.log.  The round-5 fuzzers (scripts/soak.sh) on the round-6 library
# + the sharded one-launch fuzzer, with seeds the suite does not use.
A=${1:-50}; B=${2:-55}
OUT=gpurun_out/soak6.log; : > $OUT
for s in $(seq $A $B); do
  for f in fuzz_sharded fuzz_gemm fuzz_vs_torch fuzz_small_step; do
    echo "== $f seed $s: $(timeout 900 python scripts/$f.py $s 2>&1 | grep -E 'BAD|worst|Error|error|Traceback' | tr '\n' ' ' | cut -c1-700)" >> $OUT
  done
done
GLNN_FUZZ_CASES=40 timeout 1200 python -m pytest tests/test_teacher_gpu.py -m gpu -q -x -k "random_shapes_vs_oracle" 2>&1 | tail -3 >> $OUT
cat $OUT
