#!/bin/bash
# Interleaved A/B of the teacher-training step's layer-0 BatchNorm-backward forms on the products configuration:
#   11 = dy in the transposed aggregation's epilogue + apply in the dW_0 GEMM,  10 = separate dy pass + apply in the GEMM,  00 = plain three-pass form
for i in 1 2 3; do
  for v in 11 10 00; do
    echo "apply=${v:0:1} dy=${v:1:1}: $(GLNN_SAGE_FUSE_BN_APPLY=${v:0:1} GLNN_SAGE_FUSE_BN_DY=${v:1:1} GLNN_BENCH_EPOCHS=1 python scripts/bench_train_sage.py ogbn-products 2>&1 | grep 'per step\|epoch 0' | tr '\n' ' ')"
  done
done
