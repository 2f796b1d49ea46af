#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for q in "NVFI_GRAPH_STREAM=default" "NVFI_GRAPH_STREAM=pool"; do
  for rep in 1 2 3 4 5; do
    env $q NVFI_BENCH_CHILD=1 python bench.py --graph on --no-cpu-baseline --profile-steps 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cfg3 graph [$q]', round(d['value']), round(d['ms_per_step'],3))"
  done
  env $q NVFI_BENCH_CHILD=1 python bench.py --workload cfg2 --graph on --no-cpu-baseline --profile-steps 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cfg2 graph [$q]', round(d['value']), round(d['ms_per_step'],3))"
done
