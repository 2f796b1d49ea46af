#!/bin/bash
# A/B of the fused RK2 adjoint + weight-gradient kernel (vel_fuse.hip): gradient goldens with the default, then the bench line both ways
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_training_loop.py -x -q -m gpu -k "train_grads or vs_oracle or training" > gpurun_out/fuse_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/fuse_tests.log
tail -5 gpurun_out/fuse_tests.log
for f in 1 0 1 0; do
  NVFI_RK2_FUSE=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --graph off > gpurun_out/fuse_bench_$f.json 2> gpurun_out/fuse_bench_$f.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/fuse_bench_$f.json").read().strip().splitlines()[-1])
    pc = d.get("per_class_ms", d.get("profile", {}))
    print("FUSE=$f", d["value"], d["ms_per_step"], json.dumps({k: v for k, v in d.items() if "class" in k or k == "profile"})[:1500])
except Exception as e:
    print("FUSE=$f failed", e); print(open("gpurun_out/fuse_bench_$f.err").read()[-1500:])
PY
done
