#!/bin/bash
# whole-step A/B (hipGraph replay, three branches) of environment switches.  usage: r03_pad.sh "<env A>" "<env B>" ...   (each 2 processes)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for q in "$@"; do
  for rep in 1 2; do
    env $q NVFI_BENCH_CHILD=1 timeout 300 python bench.py --graph on --no-cpu-baseline --profile-steps 0 --no-extras 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('cfg3 graph [$q]', round(d['value']), round(d['ms_per_step'],3))
except Exception as e: print('ERR [$q]', e)"
  done
done
