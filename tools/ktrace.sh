#!/bin/bash
# usage (on the GPU box, via gpurun): tools/ktrace.sh <out.csv> <command...>   - rocprofv3 kernel trace, per-kernel summary
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -- "$@" > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python $REPO/tools/rocpd_stats.py $DB $REPO/$OUT > /dev/null
head -${KT_HEAD:-30} $REPO/$OUT | cut -c1-${KT_COLS:-150}
