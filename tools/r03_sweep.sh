#!/bin/bash
# environment-knob sweep on the default bench step (hipGraph replay + eager, no extras), two interleaved repetitions.
# usage: r03_sweep.sh "<label> [ENV=val ...]" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for rep in 1 2; do
for cfg in "$@"; do
  label=${cfg%% *}; envs=${cfg#* }
  [ "$envs" = "$cfg" ] && envs=""
  env $envs timeout 300 python bench.py --no-extras --no-cpu-baseline > $OUT/sw_${label}_$rep.json 2>> $OUT/sw.err
  python - $OUT/sw_${label}_$rep.json "$label#$rep" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); pc=d["roofline"]["per_class"]; lm=d.get("launch_modes",{})
    print(sys.argv[2], "| rays/s", round(d["value"]), "step", round(d["ms_per_step"],3), {k:round(v["ms_per_step"],3) for k,v in lm.items()}, "serial", round(d["work_per_step"]["ms_per_step_profiled_serial"],3),
          "|", " ".join(f"{k}={v['ms_per_step']:.3f}" for k,v in pc.items() if v['ms_per_step']>0.2))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
done
done
grep -v amdgpu.ids $OUT/sw.err | tail -5
