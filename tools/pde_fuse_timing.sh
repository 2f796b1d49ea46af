#!/bin/bash
# Per-phase shader-clock intervals of the fused PDE adjoint kernel (pde_fuse.hip, -DPF_TIMING).
#   here:        tools/pde_fuse_timing.sh build      builds .exp/pftiming/libnvfi_hip.so
#   GPU box:     gpurun -- 'bash tools/pde_fuse_timing.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
if [ "$1" = "build" ]; then exec bash tools/build_exp.sh pftiming pde_fuse.hip -DPF_TIMING; fi
mkdir -p gpurun_out
NVFI_OVERLAP=0 NVFI_LIB=$PWD/.exp/pftiming/libnvfi_hip.so timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --graph off --profile-steps 0 \
    > gpurun_out/pf_timing.json 2> gpurun_out/pf_timing.err
grep "pde fuse timing" gpurun_out/pf_timing.err | tail -2
