set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_prefilter16.py -x -q -s 2>&1 | tail -14
for m in fp32 fp16band split32; do
  NVFI_PDE_PREFILTER=$m python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'], d['roofline']['per_class'].get('pde_prefilter'))"
  NVFI_OVERLAP=0 NVFI_PDE_PREFILTER=$m python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m one-stream', d['value'], d['ms_per_step'])"
done
export TMPDIR=/tmp
NVFI_PDE_PREFILTER=fp16band rocprofv3 --kernel-trace --stats -d gpurun_out/pre16_prof -o p -- python tests/pre16_check.py /tmp/o.npz 262144 --bench > /dev/null 2>&1
python - <<'PY'
import csv,glob
import subprocess
print(subprocess.run("python tools/rocpd_stats.py gpurun_out/pre16_prof/p_results.db | head -8 | cut -c1-120", shell=True, capture_output=True, text=True).stdout)
PY
