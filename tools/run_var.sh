python tools/perf_s2.py 2>&1 | grep -E "synthesis|up=1"
timeout 600 python -m pytest tests/test_upconv_gpu.py tests/test_s2_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "up or synthesis" 2>&1 | tail -3
