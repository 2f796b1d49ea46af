#!/bin/bash
# kernel trace of the drop-in loop (bench.py --mode dropin): per-kernel statistics + device idle gaps of one steady-state iteration
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
NVFI_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -- python $REPO/bench.py --mode dropin --steps 10 --warmup 3 --no-extras --no-cpu-baseline --profile-steps 0 "$@" > /tmp/kt.log 2>&1
grep -a "^{" /tmp/kt.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dropin under the profiler:', round(d['value']), round(d['ms_per_step'],3))"
DB=$(find /tmp/kt -name "*.db" | head -1)
mkdir -p $REPO/gpurun_out
python $REPO/tools/rocpd_stats.py $DB $REPO/gpurun_out/dropin_kernel_stats.csv > /dev/null
python $REPO/tools/step_gaps.py $DB | head -40
python - $DB <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
rows = c.execute("select name, start, end from kernels order by start").fetchall()
marks = [r[1] for r in rows if r[0].startswith("void k_rk2_split<")]
lo, hi = marks[-4], marks[-3]
step = [r for r in rows if lo <= r[1] < hi]
ours = [r for r in step if "k_" in r[0].split("(")[0] and ("at::" not in r[0])]
torchk = [r for r in step if r not in ours]
print(f"one iteration: {len(step)} kernels, {len(torchk)} of them torch's; device time ours {sum(e-s for _,s,e in ours)/1e3:.0f} us, torch {sum(e-s for _,s,e in torchk)/1e3:.0f} us")
agg = {}
for n, s, e in torchk:
    k = n[:70]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"   {cnt:3d} x {t/1e3:7.1f} us  {k}")
PY
