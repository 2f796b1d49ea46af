#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
python -m pytest tests/test_gpu_parity.py tests/test_cfg1.py tests/test_cfg2.py tests/test_gpu_training_loop.py tests/test_gpu_graph.py tests/test_gpu_edges.py tests/test_sh_mode.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do python bench.py --workload cfg2 --no-cpu-baseline --profile-steps 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cfg2', round(d['value']), round(d['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in d['launch_modes'].items()})"; done
NVFI_BWD_FORK=0 python bench.py --workload cfg2 --no-cpu-baseline --profile-steps 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cfg2 nofork', round(d['value']), round(d['ms_per_step'],3))"
for f in 1 0 1 0; do NVFI_BWD_FORK=$f python bench.py --mode dropin --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('dropin fork=$f', round(d['value']), round(d['ms_per_step'],3))"; done
