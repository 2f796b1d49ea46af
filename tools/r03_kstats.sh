#!/bin/bash
# per-kernel statistics of the profiled (one-stream) pass.  usage: r03_kstats.sh <label> [env assignments...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
label=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$label
env "$@" NVFI_BENCH_CHILD=1 rocprofv3 --kernel-trace -d /tmp/prof_$label -- python $REPO/bench.py --graph off --no-cpu-baseline > $OUT/kstats_$label.log 2>&1
DB=$(find /tmp/prof_$label -name "*.db" | head -1)
python $REPO/tools/rocpd_stats.py $DB $OUT/kstats_$label.csv --after-marker > /dev/null
head -30 $OUT/kstats_$label.csv | cut -c1-150
