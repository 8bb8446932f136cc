#!/bin/bash
# every 3x3 weight-gradient shape of the E.BE backward at config 3 (batch 8):  tools/perf_wgrad_all.sh [ENV=value ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
envs=("$@")
for sh in "16 16 1024" "16 32 1024" "32 32 512" "32 64 512" "64 64 256" "64 128 256" "128 128 128" "128 256 128" "256 256 64" "256 512 64" "512 512 32" "512 512 16" "512 512 8"; do
  set -- $sh
  env "${envs[@]}" python tools/perf_wgrad.py 8 $1 $2 $3 3 2>/dev/null | tail -1
done
