#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_maskfield.py tests/test_gpu_training_loop.py tests/test_cfg1.py -m gpu -x -q > $OUT/r03c_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r03c_pytest.log
tail -5 $OUT/r03c_pytest.log
for cfg in "engine 2 512" "ring 2 512" "ring 3 256" "ring 3 512" "ring 2 768" "ring 4 256"; do
  set -- $cfg
  NVFI_WGRAD=$1 NVFI_WGRAD_SLOTS=$2 NVFI_WGRAD_WGS=$3 timeout 300 python bench.py --graph off --no-cpu-baseline > $OUT/r03c_bench_$1_$2_$3.json 2>> $OUT/r03c.err
  python - $OUT/r03c_bench_$1_$2_$3.json "$cfg" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); pc=d["roofline"]["per_class"]["wgrad"]
    print(sys.argv[2], round(d["value"]), round(d["ms_per_step"],3), "wgrad ms", round(pc["ms_per_step"],3), "frac", round(pc["frac"],3), "serial", round(d["work_per_step"]["ms_per_step_profiled_serial"],3))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
done
grep -v amdgpu.ids $OUT/r03c.err | tail -5
