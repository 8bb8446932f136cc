#!/bin/bash
# Builds libdge_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libdge_hip.so
SRCS="capi.hip conv_igemm.hip s2_kernels.hip $(ls *_kernels.hip | grep -v s2_kernels.hip || true)"
objs=""
for f in $SRCS; do
  o="build/${f%.hip}.o"
  mkdir -p build
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ conv_params.h -nt "$o" ] || [ ../../include/dge_hip.h -nt "$o" ]; then
    echo "hipcc $f"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o "$o" &
  fi
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
echo "built $OUT"
