#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
NVFI_LIB=$PWD/.exp/timing/libnvfi_hip.so timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --graph off --profile-steps 0 > gpurun_out/fuse_timing.json 2> gpurun_out/fuse_timing.err
grep "fuse timing" gpurun_out/fuse_timing.err | tail -4
tail -2 gpurun_out/fuse_timing.err
