#!/bin/bash
# A/B of an environment switch on the profiled one-stream pass: per-kernel lines matching a pattern.  usage: r03_ab.sh <pattern> <env A> <env B>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
pat=$1; shift
for e in "$@"; do
  l=$(echo $e | tr -c 'A-Za-z0-9\n' '_')
  bash $REPO/tools/r03_kstats.sh ab_$l $e > /dev/null 2>&1
  echo "== $e"; grep -E "$pat" $REPO/gpurun_out/kstats_ab_$l.csv | cut -c1-110
  tail -1 $REPO/gpurun_out/kstats_ab_$l.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('rays/s', round(d['value']), 'ms', round(d['ms_per_step'],3), 'serial', round(d['work_per_step']['ms_per_step_profiled_serial'],3))
except Exception as e: print('ERR', e)"
done
