#!/bin/bash
# kernel trace of a bench.py command summarised per kernel AND per (kernel, grid):
#   tools/prof_shapes.sh <out-name> [bench.py args...]  -> gpurun_out/<out-name>.txt, gpurun_out/<out-name>_shapes.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o r -- python $R/bench.py "$@" > /tmp/prof_$name.log 2>&1
tail -2 /tmp/prof_$name.log | cut -c1-400
db=$(find /tmp/prof_$name -name "*.db" | head -1)
mkdir -p $R/gpurun_out
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py $*"; python $R/tools/rocpd_stats.py $db 80; } > $R/gpurun_out/$name.txt
python $R/tools/trace_by_shape.py $db 160 3 > $R/gpurun_out/${name}_shapes.txt
python $R/tools/gpu_gaps.py $db > $R/gpurun_out/${name}_gaps.txt
head -12 $R/gpurun_out/$name.txt
