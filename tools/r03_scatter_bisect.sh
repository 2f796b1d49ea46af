#!/bin/bash
# scatter bisection: per-kernel time of the tile-scatter kernels for library variants.  usage: r03_scatter_bisect.sh <variant dir under .exp> ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for v in base "$@"; do
  if [ "$v" = base ]; then lib=""; else lib="NVFI_LIB=$REPO/.exp/$v/libnvfi_hip.so"; fi
  bash $REPO/tools/r03_kstats.sh sc_$v $lib > /dev/null 2>&1
  echo "== $v"; grep -E "tile_scatter|k_og|tile_hist|tile_fill" $REPO/gpurun_out/kstats_sc_$v.csv | cut -c1-120
done
