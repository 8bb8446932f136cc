for v in "" ep2; do echo "== ${v:-base}"; if [ -n "$v" ]; then export DGE_LIB_PATH=deep-gan-encoders_amd/variants/libdge_$v.so; else unset DGE_LIB_PATH; fi
for cfg in "8 128 128 256 3" "8 256 256 128 3" "8 512 512 64 3" "8 512 512 32 3"; do python tools/perf_conv.py $cfg 2>&1 | grep DBG; done; done
