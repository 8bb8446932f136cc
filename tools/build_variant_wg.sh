#!/bin/bash
# Tuning build of wgrad_dma.hip (other objects reused from csrc/build):  tools/build_variant_wg.sh NAME "-DDGE_WG_TIMING ..."
set -e
cd "$(dirname "$0")/../deep-gan-encoders_amd/csrc"
mkdir -p ../variants build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c wgrad_dma.hip -o build/wgv_$1.o
objs=$(ls build/*.o | grep -v "build/wgrad_dma.o" | grep -v "build/wgv_")
hipcc --offload-arch=gfx950 -shared -fPIC $objs build/wgv_$1.o -o ../variants/libdge_$1.so
rm -f build/wgv_$1.o
echo "built variants/libdge_$1.so"
