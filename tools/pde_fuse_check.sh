#!/bin/bash
# first GPU contact of a pde_fuse.hip build: PDE parity tests with the fused kernel, then the same-box A/B against NVFI_PDE_FUSE=0
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "pde or three_training or fullsize" > gpurun_out/pf_tests.txt 2>&1; echo "rc=$?" >> gpurun_out/pf_tests.txt
tail -15 gpurun_out/pf_tests.txt
timeout 900 bash tools/gpu_ab.sh "NVFI_PDE_FUSE=0" "NVFI_PDE_FUSE=1" 2>&1 | tee gpurun_out/pf_ab.txt
