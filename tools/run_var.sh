timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "stride1 or encoder_conv or dgrad or data_grad or lpips_first or ragged" 2>&1 | tail -15
for v in p1 p2 p3; do echo "== $v"; DGE_LIB_PATH=deep-gan-encoders_amd/variants/libdge_$v.so python tools/perf_stream_one.py g 8 1024 32 32 2>&1 | grep stream; DGE_LIB_PATH=deep-gan-encoders_amd/variants/libdge_$v.so python tools/perf_stream_one.py dot 8 1024 32 32 2>&1 | grep stream; done
python tools/perf_stream.py 8
