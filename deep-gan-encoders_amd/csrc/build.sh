#!/bin/bash
# Builds libdge_hip.so for gfx950 in-tree (cross-compiles without a GPU).
# Every object is compiled to a temporary name and renamed on success, and every compile job is waited for by pid:
# a failed compile stops the build instead of linking a stale object.
set -e
cd "$(dirname "$0")"
OUT=../libdge_hip.so
SRCS="capi.hip conv_pp.hip up_pp.hip conv_igemm.hip conv_igemm_f32.hip conv_stream.hip upconv_stream.hip conv_pw.hip conv_small.hip wgrad_dma.hip s2_kernels.hip $(ls *_kernels.hip | grep -v s2_kernels.hip || true)"
mkdir -p build
objs=""
pids=""
for f in $SRCS; do
  o="build/${f%.hip}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ conv_params.h -nt "$o" ] || [ conv_epilogue.h -nt "$o" ] || [ conv_igemm_impl.h -nt "$o" ] || [ ../../include/dge_hip.h -nt "$o" ]; then
    echo "hipcc $f"
    rm -f "$o"
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o "$o.tmp.$$" && mv "$o.tmp.$$" "$o" ) &
    pids="$pids $!"
  fi
  objs="$objs $o"
done
for p in $pids; do
  wait "$p" || { echo "build.sh: a compile job failed" >&2; rm -f build/*.tmp.$$; exit 1; }
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$OUT.tmp.$$"
mv "$OUT.tmp.$$" $OUT
echo "built $OUT"
