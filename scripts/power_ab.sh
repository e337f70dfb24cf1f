#!/bin/bash
# usage (GPU box, repo root): scripts/power_ab.sh -> gpurun_out/power/power.txt: socket power + engine clock per kernel under sustained launches
set -u
OUT=$PWD/gpurun_out/power; mkdir -p "$OUT"
rocm-smi --showpower --showclocks -d 0 > "$OUT/smi_sample.txt" 2>&1
rocm-smi --showmaxpower -d 0 >> "$OUT/smi_sample.txt" 2>&1
python scripts/power_probe.py 3 2>&1 | grep -v amdgpu.ids | tee "$OUT/power.txt"
for v in variants/libglnn_rwp_*.so; do
  [ -e "$v" ] || continue
  GLNN_POWER_LABEL=$(basename $v .so) GLNN_POWER_KERNELS=walk GLNN_LIB_PATH=$PWD/$v python scripts/power_probe.py 3 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/power.txt"
done
for v in variants/libglnn_rwp_*.so; do
  [ -e "$v" ] || continue
  GLNN_LIB_PATH=$PWD/$v python scripts/rowwalk_probe.py "$(basename $v .so)" 1 2>&1 | grep -v amdgpu.ids | head -4 | tee -a "$OUT/power.txt"
done
