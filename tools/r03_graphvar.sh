#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for q in "NVFI_GRAPH_DEPTH=2" "NVFI_GRAPH_DEPTH=1" "NVFI_GRAPH_DEPTH=3" "NVFI_GRAPH_DEPTH=0"; do
  for rep in 1 2 3 4 5; do
    env $q NVFI_GRAPH_TRIES=1 NVFI_BENCH_CHILD=1 python bench.py --graph on --no-cpu-baseline --profile-steps 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('graph [$q]', round(d['value']), round(d['ms_per_step'],3), d['config']['graph_capture_attempts_ms'])"
  done
done
