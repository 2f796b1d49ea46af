#!/bin/bash
# timeline (start offset, duration, queue, kernel) of one steady-state iteration of the reference's loop body (--mode dropin)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ds
env "$@" rocprofv3 --kernel-trace -d /tmp/prof_ds -- python $REPO/bench.py --mode dropin --no-cpu-baseline --profile-steps 0 > $OUT/dropin_steptrace.log 2>&1
DB=$(find /tmp/prof_ds -name "*.db" | head -1)
python -c "
import sys; sys.path.insert(0, '$REPO/tools')
import step_trace; step_trace.main('$DB', 4, 'void k_rk2_split<')" > $OUT/dropin_steptrace.txt 2>&1
grep -a '^{' $OUT/dropin_steptrace.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), d['ms_per_step'])"
