#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for q in "NVFI_WGRAD_WGS=256" "NVFI_WGRAD_WGS=248" "NVFI_WGRAD_WGS=240" "NVFI_WGRAD_WGS=224" "NVFI_WGRAD_WGS=192"; do
  for rep in 1 2; do
    env $q python bench.py --graph off --no-cpu-baseline --profile-steps 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('eager [$q]', round(d['value']), round(d['ms_per_step'],3))"
  done
done
