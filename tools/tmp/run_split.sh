cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_prefilter16.py -x -q -k split32 2>&1 | tail -2
for nt in ${NTS:-4}; do
for m in ${MODES:-split32}; do
  NVFI_SPLIT_NT=$nt NVFI_PDE_PREFILTER=$m python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m nt=$nt', round(d['value']), round(d['ms_per_step'],3), d['roofline']['per_class'].get('pde_prefilter'))"
done; done
