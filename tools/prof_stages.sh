#!/bin/bash
# per-stage kernel tables of the headline step:  tools/prof_stages.sh <out-name>  -> gpurun_out/<out-name>.txt
cd /tmp && export TMPDIR=/tmp
# one stream: counter collection serialises kernels (a step with its side streams did not finish under --pmc), and the stage
# tables want every launch at its isolated duration
export DGE_SIDE_STREAMS=0
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace -d /tmp/prof_$name -o r -- python $R/tools/stage_trace.py run > /tmp/prof_$name.log 2>&1
tail -3 /tmp/prof_$name.log | cut -c1-300
db=$(find /tmp/prof_$name -name "*.db" | head -1)
mkdir -p $R/gpurun_out
cp /tmp/prof_$name.log $R/gpurun_out/$name.log; python $R/tools/stage_trace.py report $db 45 > $R/gpurun_out/$name.txt 2>&1
head -5 $R/gpurun_out/$name.txt
