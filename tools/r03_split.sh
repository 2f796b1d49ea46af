#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
python -m pytest tests/test_gpu_parity.py tests/test_gpu_training_loop.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_chessboard.py tests/test_gpu_convergence.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
for f in 1 0 1 0; do NVFI_PDE_SPLIT=$f python bench.py --mode dropin --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('dropin split=$f', round(d['value']), round(d['ms_per_step'],3))"; done
