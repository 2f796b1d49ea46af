#!/bin/bash
# same-box A/B of environment switches on the default fused step: tools/gpu_ab.sh "A=1" "B=2 C=3" ...   (an empty string = the default)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
i=0
for rep in 1 2; do
for e in "$@"; do
  i=$((i+1))
  env $e timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --graph off > gpurun_out/sw_$i.json 2> gpurun_out/sw_$i.err
  python - "$e" gpurun_out/sw_$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    pc = d["roofline"]["per_class"]
    print("[%s]" % sys.argv[1], round(d["value"]), round(d["ms_per_step"], 3), "serial", round(d["work_per_step"]["ms_per_step_profiled_serial"], 3), " ".join("%s=%.3f" % (k, v["ms_per_step"]) for k, v in pc.items() if v["ms_per_step"] > 0.2))
except Exception as ex:
    print("[%s] failed" % sys.argv[1], ex)
PY
done
done
