#!/bin/bash
# Per-phase shader-clock intervals of the fused RK2 adjoint kernel (vel_fuse.hip, -DFUSE_TIMING).
#   here:        tools/fuse_timing.sh build      builds .exp/timing/libnvfi_hip.so
#   GPU box:     gpurun -- 'bash tools/fuse_timing.sh'      prints the "[fuse timing]" lines of a short bench run
cd "${GRAFT_REPO_ROOT:-/root/repo}"
if [ "$1" = "build" ]; then exec bash tools/build_exp.sh timing vel_fuse.hip -DFUSE_TIMING; fi
mkdir -p gpurun_out
NVFI_LIB=$PWD/.exp/timing/libnvfi_hip.so timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --graph off --profile-steps 0 \
    > gpurun_out/fuse_timing.json 2> gpurun_out/fuse_timing.err
grep "fuse timing" gpurun_out/fuse_timing.err | tail -2
