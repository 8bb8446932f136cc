#!/bin/bash
# Tuning build of conv_igemm.hip under extra -D flags: tools/build_variant_igemm.sh NAME "-D..." -> variants/libdge_NAME.so
set -e
cd "$(dirname "$0")/../deep-gan-encoders_amd/csrc"
mkdir -p ../variants build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c conv_igemm.hip -o build/conv_igemm_$1.o.v
objs=$(ls build/*.o | grep -v "conv_igemm")
hipcc --offload-arch=gfx950 -shared -fPIC $objs build/conv_igemm_$1.o.v -o ../variants/libdge_$1.so
rm -f build/conv_igemm_$1.o.v
echo "built variants/libdge_$1.so"
