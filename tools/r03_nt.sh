#!/bin/bash
# non-temporal stash stores / loads (engine.h STASH_ST / STASH_LD): library variants under .exp/, two runs each.  usage: r03_nt.sh <variant dirs...>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd $REPO
for rep in 1 2; do
for v in base "$@"; do
  lib=$REPO/.exp/$v/libnvfi_hip.so; [ $v = base ] && lib=$REPO/nvfi_amd/csrc/libnvfi_hip.so
  NVFI_LIB=$lib timeout 300 python bench.py --no-extras --graph off --no-cpu-baseline > $OUT/nt_${v}_$rep.json 2>> $OUT/nt.err
  python - $OUT/nt_${v}_$rep.json "$v#$rep" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); pc=d["roofline"]["per_class"]
    print(sys.argv[2], "| rays/s", round(d["value"]), "step", round(d["ms_per_step"],3), "serial", round(d["work_per_step"]["ms_per_step_profiled_serial"],3),
          "|", " ".join(f"{k}={v['ms_per_step']:.3f}" for k,v in pc.items() if v['ms_per_step']>0.02))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
done
done
grep -v amdgpu.ids $OUT/nt.err | tail -5
