#!/bin/bash
# Tuning build of ONE source of csrc/ under extra -D flags (other objects reused from csrc/build):
#   tools/build_variant_src.sh NAME up_pp.hip "-DDGE_UP_TIMING"   ->  deep-gan-encoders_amd/variants/libdge_NAME.so  (select with DGE_LIB_PATH)
set -e
cd "$(dirname "$0")/../deep-gan-encoders_amd/csrc"
mkdir -p ../variants build
base=${2%.hip}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $3 -c $2 -o build/var_$1.o
objs=$(ls build/*.o | grep -v "build/$base.o" | grep -v "build/var_")
hipcc --offload-arch=gfx950 -shared -fPIC $objs build/var_$1.o -o ../variants/libdge_$1.so
rm -f build/var_$1.o
echo "built variants/libdge_$1.so"
