#!/bin/bash
# same-box A/B of two builds of libdge_hip.so on the headline step:  tools/ab_bench.sh <libA.so> <libB.so> [rounds] [steps]
# (interleaved rounds; prints ms_per_step of each run and the median step time)
A=$1; B=$2; R=${3:-2}; S=${4:-20}
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    DGE_LIB_PATH=$L python bench.py --steps $S --warmup 3 --no-cpu-baseline --no-extras --no-synthesis 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3), 'img/s', round(d['value'],1))"
  done
done
