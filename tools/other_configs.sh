#!/bin/bash
# the other BASELINE configs on the HIP path (bench.py --mtype 1|3|4, tools/bench_embed.py): tools/other_configs.sh <out-name>
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/$1.txt
: > $out
X="--steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --no-synthesis"
python bench.py --mtype 1 --img-size 256 --start-features 64 $X 2>/dev/null | tail -1 >> $out
python bench.py --mtype 1 --img-size 256 --start-features 64 --batch 32 $X 2>/dev/null | tail -1 >> $out
python bench.py --mtype 1 $X 2>/dev/null | tail -1 >> $out
python bench.py --mtype 3 --img-size 256 --start-features 64 $X 2>/dev/null | tail -1 >> $out
python bench.py --mtype 4 --img-size 256 --start-features 64 $X 2>/dev/null | tail -1 >> $out
python tools/bench_embed.py --iters 60 2>/dev/null | tail -1 >> $out            # eager: host-bound at batch 1 (~570 launches per iteration)
python tools/bench_embed.py --iters 60 --graph 2>/dev/null | tail -1 >> $out    # hipGraph replay of the captured iteration: GPU-bound
cut -c1-200 $out
