#!/bin/bash
# usage: scripts/build_variant.sh NAME "<extra hipcc -D flags>" [file.hip ...]   -> variants/libglnn_NAME.so
# Rebuilds the named csrc files with the extra flags and links them with the default objects of the others.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C="$R/graphless-neural-networks_amd/csrc"
NAME=$1; FLAGS=$2; shift 2
mkdir -p "$R/variants/obj_$NAME"
OBJS=""
for s in "$C"/*.hip; do
  b=$(basename "$s" .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I"$R/include" -I"$C" $FLAGS -c "$s" -o "$R/variants/obj_$NAME/$b.o"
    OBJS="$OBJS $R/variants/obj_$NAME/$b.o"
  else
    OBJS="$OBJS $C/build/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/variants/libglnn_$NAME.so" $OBJS
rm -rf "$R/variants/obj_$NAME"
echo "built variants/libglnn_$NAME.so"
