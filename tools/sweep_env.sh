#!/bin/bash
# same-box sweep of run-time switches on the headline step: tools/sweep_env.sh "VAR=val" "VAR2=val VAR3=val" ...   ("" = default)
S=${STEPS:-20}
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg python bench.py --steps $S --warmup 3 --no-cpu-baseline --no-extras --no-synthesis --no-roofline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$cfg]', round(d['ms_per_step'],3), 'median', round(d['step_ms']['median'],3))"
done
done
