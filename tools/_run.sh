cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu -k "pde or three_training" 2>&1 | tail -3
bash tools/pde_fuse_timing.sh
NVFI_OVERLAP=0 NVFI_LIB=$PWD/.exp/pftiming2/libnvfi_hip.so timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --graph off --profile-steps 0 2>&1 >/dev/null | grep "pde fuse timing" | tail -1
bash tools/gpu_ab.sh "NVFI_PDE_FUSE=0" "NVFI_PDE_FUSE=1"
