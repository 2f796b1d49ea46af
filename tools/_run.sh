cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "pde or three_training or fullsize" 2>&1 | tail -4
bash tools/gpu_ab.sh "NVFI_PDE_FUSE=0" "NVFI_PDE_FUSE=1"
