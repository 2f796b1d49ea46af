#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for mode in on off; do
  rm -rf /tmp/prof_$mode
  NVFI_BENCH_CHILD=1 rocprofv3 --kernel-trace -d /tmp/prof_$mode -- python $REPO/bench.py --graph $mode --no-cpu-baseline --profile-steps 0 --no-extras > $OUT/r03d_trace_$mode.log 2>&1
  DB=$(find /tmp/prof_$mode -name "*.db" | head -1)
  python $REPO/tools/step_trace.py $DB 4 > $OUT/r03d_step_trace_$mode.txt 2>&1
  grep -a "^{" $OUT/r03d_trace_$mode.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$mode', round(d['value']), d['ms_per_step'])"
done
