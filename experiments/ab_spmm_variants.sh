#!/bin/bash
# usage: scripts/ab_spmm_variants.sh variants/libglnn_A.so variants/libglnn_B.so ...   (each run prints wave/row vs group/row of that build)
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset GLNN_LIB_PATH; else export GLNN_LIB_PATH=$PWD/$lib; fi
  echo "== $lib"
  python scripts/ab_spmm.py 1.0 12.5 2>&1 | grep -E "D=128|D=256|D=100 " 
done
