#!/bin/bash
# Tuning build: libdge_hip.so with tools/probes/conv_stream_experiments.hip (the copy of conv_stream.hip that carries the DGE_SC_*
# timing / tuning switches) compiled under extra -D flags in place of conv_stream.o (other objects reused from csrc/build).
#   tools/build_variant.sh NAME "-DDGE_SC_NR=8 -DDGE_SC_D=5 -DDGE_SC_ONLY=32,32"   ->  deep-gan-encoders_amd/variants/libdge_NAME.so
# Select at run time with DGE_LIB_PATH.
set -e
cd "$(dirname "$0")/../deep-gan-encoders_amd/csrc"
mkdir -p ../variants build
# third argument "main": compile the product's conv_stream.hip (it carries DGE_SC_TIMING job stamps and DGE_SC_ONLY) instead of the experiment copy
SRC=../../tools/probes/conv_stream_experiments.hip
[ "$3" = "main" ] && SRC=conv_stream.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c $SRC -o build/conv_stream_$1.o
objs=$(ls build/*.o | grep -v "build/conv_stream")
hipcc --offload-arch=gfx950 -shared -fPIC $objs build/conv_stream_$1.o -o ../variants/libdge_$1.so
rm -f build/conv_stream_$1.o
echo "built variants/libdge_$1.so"
