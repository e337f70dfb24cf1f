#!/bin/bash
# Fuzzers with seeds the test suite does not use (GPU box, repo root): scripts/soak.sh FIRST LAST -> gpurun_out/soak.log (failures + worst errors)
A=${1:-10}; B=${2:-20}
OUT=gpurun_out/soak.log; : > $OUT
for s in $(seq $A $B); do
  for f in fuzz_gemm fuzz_vs_torch fuzz_small_step; do
    echo "== $f seed $s: $(timeout 600 python scripts/$f.py $s 2>&1 | grep -E 'BAD|worst|Error|error|Traceback' | tr '\n' ' ' | cut -c1-600)" >> $OUT
  done
done
for s in $(seq $A $B); do
  echo "== kernel sweeps seed $s: $(GLNN_SHAPE_SEED=$s GLNN_SWEEP_SEED=$s timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "randomized_shape_sweep or randomized_gemm_sweep" 2>&1 | tail -4 | tr '\n' ' ' | cut -c1-700)" >> $OUT
done
GLNN_FUZZ_CASES=60 timeout 1500 python -m pytest tests/test_teacher_gpu.py -m gpu -q -x -k "random_shapes_vs_oracle" 2>&1 | tail -3 >> $OUT
cat $OUT
