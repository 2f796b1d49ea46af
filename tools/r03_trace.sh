#!/bin/bash
# GPU tests + kernel traces of one step in graph and eager mode (timed region only: no profiled pass)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1800 python -m pytest tests -m gpu -q > $OUT/r03b_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r03b_pytest.log
tail -8 $OUT/r03b_pytest.log
cd /tmp && export TMPDIR=/tmp
for mode in on off; do
  rm -rf /tmp/prof_$mode
  NVFI_BENCH_CHILD=1 rocprofv3 --kernel-trace -d /tmp/prof_$mode -- python $REPO/bench.py --graph $mode --no-cpu-baseline --profile-steps 0 > $OUT/r03b_trace_$mode.log 2>&1
  DB=$(find /tmp/prof_$mode -name "*.db" | head -1)
  python $REPO/tools/step_trace.py $DB 4 > $OUT/r03b_step_trace_$mode.txt 2>&1
  python $REPO/tools/timeline.py $DB $OUT/r03b_timeline_$mode.csv > $OUT/r03b_timeline_$mode.txt 2>&1
  grep -a "^{" $OUT/r03b_trace_$mode.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$mode', round(d['value']), d['ms_per_step'])"
done
head -3 $OUT/r03b_step_trace_on.txt
