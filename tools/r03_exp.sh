#!/bin/bash
# wgrad experiments: per-class time of the wgrad launches for library variants / env settings.  usage: r03_exp.sh "<label> <env assignments...>" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
i=0
for cfg in "$@"; do
  i=$((i+1))
  label=${cfg%% *}; envs=${cfg#* }
  [ "$envs" = "$cfg" ] && envs=""
  env $envs timeout 300 python bench.py --graph off --no-cpu-baseline > $OUT/exp_$label.json 2>> $OUT/exp.err
  python - $OUT/exp_$label.json "$cfg" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); pc=d["roofline"]["per_class"]
    print(sys.argv[2], "| rays/s", round(d["value"]), "step", round(d["ms_per_step"],3), "serial", round(d["work_per_step"]["ms_per_step_profiled_serial"],3),
          "|", " ".join(f"{k}={v['ms_per_step']:.3f}" for k,v in pc.items() if v['ms_per_step']>0.02))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
done
grep -v amdgpu.ids $OUT/exp.err | tail -5
