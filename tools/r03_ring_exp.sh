#!/bin/bash
# weight-gradient ring variants (libraries under .exp/): per-kernel time of the profiled pass.  usage: r03_ring_exp.sh <variant> ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for v in base "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="NVFI_LIB=$REPO/.exp/$v/libnvfi_hip.so"; fi
  bash $REPO/tools/r03_kstats.sh rg_$v $lib > /dev/null 2>&1
  echo "== $v"; grep -E "wgrad" $REPO/gpurun_out/kstats_rg_$v.csv | cut -c1-120
done
