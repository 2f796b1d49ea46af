#!/bin/bash
# first GPU pass of round 3: tests + the bench lines in their launch variants
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r03a_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r03a_pytest.log
tail -5 $OUT/r03a_pytest.log
timeout 600 python bench.py > $OUT/r03a_bench_line.json 2> $OUT/r03a_bench_line.err; echo "bench rc $?"
timeout 600 python bench.py --graph off --no-cpu-baseline > $OUT/r03a_bench_line_eager.json 2>> $OUT/r03a_bench_line.err
timeout 600 python bench.py --mode dropin --no-cpu-baseline > $OUT/r03a_bench_line_dropin.json 2>> $OUT/r03a_bench_line.err
timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $OUT/r03a_bench_line_cfg2.json 2>> $OUT/r03a_bench_line.err
timeout 600 python bench.py --workload cfg2 --graph off --no-cpu-baseline > $OUT/r03a_bench_line_cfg2_eager.json 2>> $OUT/r03a_bench_line.err
for f in $OUT/r03a_bench_line*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(round(d["value"]), round(d["ms_per_step"],3), d["config"]["launch"][:20], (d.get("roofline") or {}).get("kernel"), round((d.get("roofline") or {}).get("frac",0),3))
except Exception as e: print("ERR", e)
PY
done
tail -5 $OUT/r03a_bench_line.err
