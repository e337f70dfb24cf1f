#!/bin/bash
# usage (GPU box, repo root): scripts/cls_prof.sh [lib.so ...]  -> kernel-only durations (rocprofv3 --kernel-trace --stats) of cls_fwd_kernel on the
# MLP3w8 classifier shape, with / without the criterion, for the in-tree library and every variant given
export TMPDIR=/tmp
for v in in-tree "$@"; do
  if [ "$v" = in-tree ]; then unset GLNN_LIB_PATH; else export GLNN_LIB_PATH=$v; fi
  for loss in 1 0; do
    rm -rf /tmp/clsp; CLS_PROBE_ONE="2 $loss" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/clsp -- python scripts/cls_probe.py > /dev/null 2>&1
    f=$(find /tmp/clsp -name "*kernel_stats.csv" | head -1)
    echo "$v loss=$loss $(grep cls_fwd_kernel "$f" | awk -F, '{printf "calls %s avg %.1f us min %.1f us", $2, $4/1000, $6/1000}')"
  done
done
