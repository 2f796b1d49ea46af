#!/bin/bash
# per-kernel statistics of the radiance-only (cfg2) step for the two tile scatters.  usage: r03_cfg2_kstats.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in mfma lds; do
  rm -rf /tmp/prof_c2$m
  NVFI_SCATTER=$m NVFI_BENCH_CHILD=1 rocprofv3 --kernel-trace -d /tmp/prof_c2$m -- python $REPO/bench.py --workload cfg2 --graph off --no-cpu-baseline --no-extras > $OUT/kstats_c2$m.log 2>&1
  DB=$(find /tmp/prof_c2$m -name "*.db" | head -1)
  python $REPO/tools/rocpd_stats.py $DB $OUT/kstats_c2$m.csv --after-marker > /dev/null
  echo "== $m"; tail -1 $OUT/kstats_c2$m.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  head -16 $OUT/kstats_c2$m.csv | cut -c1-120
done
