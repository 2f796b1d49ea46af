#!/bin/bash
# prefilter class time (profiled one-stream pass) and step time (eager three streams, then hipGraph replay) per environment setting.
# usage: r03_modes.sh "<env A>" "<env B>" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for q in "$@"; do
  env $q NVFI_BENCH_CHILD=1 timeout 300 python bench.py --graph off --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); pc=d['roofline']['per_class']
    print('[$q] eager', round(d['value']), round(d['ms_per_step'],3), 'serial', round(d['work_per_step']['ms_per_step_profiled_serial'],3), 'prefilter ms', round(pc['pde_prefilter']['ms_per_step'],3), 'kept', d['work_per_step']['P_kept'])
except Exception as e: print('ERR [$q]', e)"
  env $q NVFI_BENCH_CHILD=1 timeout 300 python bench.py --graph on --no-cpu-baseline --profile-steps 0 --no-extras 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print('[$q] graph', round(d['value']), round(d['ms_per_step'],3))
except Exception as e: print('ERR [$q]', e)"
done
