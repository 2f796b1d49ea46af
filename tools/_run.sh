REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
NVFI_BENCH_CHILD=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -- python $REPO/bench.py --mode dropin --steps 10 --warmup 3 --no-extras --no-cpu-baseline --profile-steps 0 > /tmp/kt.log 2>&1
DB=$(find /tmp/kt -name "*.db" | head -1)
python - $DB > $REPO/gpurun_out/dropin_seq.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]).cursor()
cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select name, start, end, {qcol} from kernels order by start").fetchall()
marks = [r[1] for r in rows if r[0].startswith("void k_rk2_split<")]
lo, hi = marks[-4], marks[-3]
step = [r for r in rows if lo <= r[1] < hi]
qs = {}
prev_end = step[0][1]
for n, s, e, q in step:
    qs.setdefault(q, len(qs))
    print(f"{(s-lo)/1e3:9.1f} {(e-s)/1e3:8.1f} q{qs[q]} gap {(s-prev_end)/1e3:7.1f}  {n[:60]}")
    prev_end = max(prev_end, e)
PY
