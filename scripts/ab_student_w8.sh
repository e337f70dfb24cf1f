#!/bin/bash
# Interleaved A/B of library builds on the MLP3w8 products student step (ms per step of scripts/trace_student_any.py, 200 steps each):
#   scripts/ab_student_w8.sh variants/libglnn_a.so [variants/libglnn_b.so ...]      ("new" = the in-tree build, always last in a round)
for i in 1 2 3; do
  for v in "$@" new; do
    if [ "$v" = new ]; then unset GLNN_LIB_PATH; else export GLNN_LIB_PATH=$v; fi
    echo "$v $(python scripts/trace_student_any.py 100-2048-2048-47 4096 batch 0.2 kl 2>&1 | grep 'ms per step' | awk '{print $(NF-3)}')"
  done
done
unset GLNN_LIB_PATH
