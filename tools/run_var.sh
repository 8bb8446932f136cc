timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -3
python tools/bench_embed.py 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-roofline --no-synthesis 2>&1 | tail -1 | cut -c1-900
