#!/bin/bash
# the whole GPU suite + the default bench line (what the driver runs at round end)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gputests.txt 2>&1; echo "rc=$?" >> gpurun_out/gputests.txt
tail -8 gpurun_out/gputests.txt
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], json.dumps(d.get("extras"))[:1500])
PY
