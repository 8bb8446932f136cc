#!/bin/bash
# rocprofv3 kernel trace of any python command of the repo on the GPU box, summarised per kernel (tools/rocpd_stats.py):
#   tools/prof_cmd.sh <out-name> <script.py> [args...]   -> gpurun_out/<out-name>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
rm -rf /tmp/prof_$name
rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o r -- python $R/"$@" > /tmp/prof_$name.log 2>&1
tail -2 /tmp/prof_$name.log | cut -c1-400
db=$(find /tmp/prof_$name -name "*.db" | head -1)
mkdir -p $R/gpurun_out
{ echo "# rocprofv3 --kernel-trace --stats -- python $*"; python $R/tools/rocpd_stats.py $db 60; } > $R/gpurun_out/$name.txt
head -12 $R/gpurun_out/$name.txt
