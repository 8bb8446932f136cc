for cfg in "8 128 128 256 3" "8 256 256 128 3" "8 512 512 64 3" "8 512 512 32 3" "8 512 512 16 3" "8 512 512 8 3"; do python tools/perf_conv.py $cfg 2>&1 | grep DBG; done
python tools/perf_conv.py 8 128 64 256 3 up | grep DBG
timeout 1400 python -m pytest tests -m gpu -q -x 2>&1 | tail -12
