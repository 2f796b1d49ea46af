#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for cfg in "base" "r2hi NVFI_PRIO_R2=-1" "r2hi_r1hi NVFI_PRIO_R2=-1 NVFI_PRIO_R1=-1" "r2norm NVFI_PRIO_R2=0" "pdelo_r2hi NVFI_PRIO_R2=-1 NVFI_PRIO_R1=-1 NVFI_PRIO_PDE=0" "q8 GPU_MAX_HW_QUEUES=8 NVFI_PRIO_R2=-1"; do
  label=${cfg%% *}; envs=${cfg#* }; [ "$envs" = "$cfg" ] && envs=""
  for rep in 1 2; do
  env $envs timeout 300 python bench.py --graph off --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$cfg', round(d['value']), round(d['ms_per_step'],3))"
  done
done
