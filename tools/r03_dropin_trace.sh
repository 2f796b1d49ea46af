#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_d
rocprofv3 --kernel-trace -d /tmp/prof_d -- python $REPO/bench.py --mode dropin --no-cpu-baseline --profile-steps 0 > $OUT/dropin_trace.log 2>&1
DB=$(find /tmp/prof_d -name "*.db" | head -1)
python $REPO/tools/timeline.py $DB $OUT/dropin_timeline.csv k_plane_regs > $OUT/dropin_timeline.txt 2>&1
python $REPO/tools/step_gaps.py $DB > $OUT/dropin_gaps.txt 2>&1; cat $OUT/dropin_gaps.txt

grep -a "^{" $OUT/dropin_trace.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), d['ms_per_step'])"
