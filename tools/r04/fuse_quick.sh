#!/bin/bash
# timing build (per-phase shader-clock intervals of workgroup 0) + the bench line of the default build, per-class times
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
if [ -f .exp/timing/libnvfi_hip.so ]; then
NVFI_LIB=$PWD/.exp/timing/libnvfi_hip.so timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --graph off --profile-steps 0 > gpurun_out/fuse_timing.json 2> gpurun_out/fuse_timing.err
grep "fuse timing" gpurun_out/fuse_timing.err | tail -2
fi
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train_grads" 2>&1 | tail -3
for f in ${FUSE_LIST:-1 0}; do
NVFI_RK2_FUSE=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --graph off > gpurun_out/fq_$f.json 2> gpurun_out/fq_$f.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/fq_$f.json").read().strip().splitlines()[-1])
    pc = d["roofline"]["per_class"]
    print("FUSE=$f", round(d["value"]), round(d["ms_per_step"], 3), "serial", round(d["work_per_step"]["ms_per_step_profiled_serial"], 3), " ".join("%s=%.3f" % (k, v["ms_per_step"]) for k, v in pc.items() if k in ("rk2_bwd", "wgrad", "pde_bwd", "pde_fwd", "rk2_fwd", "pde_prefilter")))
except Exception as e:
    print("FUSE=$f failed", e); print(open("gpurun_out/fq_$f.err").read()[-1500:])
PY
done
