#!/bin/bash
# quick gradient tests on the GPU, then r03_exp.sh with the given configs
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_maskfield.py tests/test_gpu_training_loop.py tests/test_cfg1.py tests/test_sh_mode.py -m gpu -x -q > $OUT/quick_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/quick_pytest.log
tail -4 $OUT/quick_pytest.log
bash tools/r03_exp.sh "$@"
