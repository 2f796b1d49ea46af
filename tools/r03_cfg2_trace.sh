#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c2
NVFI_BENCH_CHILD=1 rocprofv3 --kernel-trace -d /tmp/prof_c2 -- python $REPO/bench.py --workload cfg2 --graph off --no-cpu-baseline --profile-steps 0 --no-extras > $OUT/cfg2_trace.log 2>&1
DB=$(find /tmp/prof_c2 -name "*.db" | head -1)
python $REPO/tools/step_trace.py $DB 4 > $OUT/cfg2_step_trace.txt 2>&1
cat $OUT/cfg2_step_trace.txt | cut -c1-110
