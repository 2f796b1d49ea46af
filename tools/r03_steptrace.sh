#!/bin/bash
# timeline of one steady-state three-stream step (start offset, duration, queue, kernel).  usage: r03_steptrace.sh <label> [env...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out; mkdir -p $OUT
label=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$label
env "$@" NVFI_BENCH_CHILD=1 rocprofv3 --kernel-trace -d /tmp/prof_$label -- python $REPO/bench.py --graph off --no-cpu-baseline --profile-steps 0 --no-extras > $OUT/steptrace_$label.log 2>&1
DB=$(find /tmp/prof_$label -name "*.db" | head -1)
python $REPO/tools/step_trace.py $DB 4 > $OUT/steptrace_$label.txt 2>&1
tail -1 $OUT/steptrace_$label.log | cut -c1-200
