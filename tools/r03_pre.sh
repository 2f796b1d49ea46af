#!/bin/bash
# prefilter class time over 20 profiled one-stream steps + the hipGraph step, per environment setting (3 processes each for the step time)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for q in "$@"; do
  env $q NVFI_BENCH_CHILD=1 timeout 300 python bench.py --graph off --no-cpu-baseline --no-extras --profile-steps 20 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); pc=d['roofline']['per_class']
    print('[$q] prefilter ms', round(pc['pde_prefilter']['ms_per_step'],4), 'frac', round(pc['pde_prefilter']['frac'],4), 'rk2_fwd ms', round(pc['rk2_fwd']['ms_per_step'],4), 'frac', round(pc['rk2_fwd']['frac'],4), 'rk2_bwd ms', round(pc['rk2_bwd']['ms_per_step'],4), 'serial', round(d['work_per_step']['ms_per_step_profiled_serial'],3), 'eager', round(d['value']))
except Exception as e: print('ERR [$q]', e)"
  for rep in 1 2 3; do
  env $q NVFI_BENCH_CHILD=1 timeout 300 python bench.py --graph on --no-cpu-baseline --profile-steps 0 --no-extras 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('[$q] graph', round(d['value']), round(d['ms_per_step'],3))
except Exception as e: print('ERR [$q]', e)"
  done
done
