#!/bin/bash
# Interleaved A/B of GLNN_SAGE_FUSE_BN_APPLY (layer 0's BatchNorm-backward apply inside the dW_0 GEMM) on the products teacher-training step
for i in 1 2 3; do
  for v in 1 0; do
    echo "fuse=$v: $(GLNN_SAGE_FUSE_BN_APPLY=$v GLNN_BENCH_EPOCHS=1 python scripts/bench_train_sage.py ogbn-products 2>&1 | grep 'per step\|epoch 0')"
  done
done
