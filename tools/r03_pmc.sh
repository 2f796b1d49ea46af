#!/bin/bash
# SQ counters of selected kernels.  usage: r03_pmc.sh <label> <kernel-substring> [env assignments...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
label=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --graph off --steps 3 --warmup 1 --prime 1 --profile-steps 0 --no-cpu-baseline"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_INSTS_SALU"; do
  rm -rf /tmp/pmc_$label
  env "$@" NVFI_BENCH_CHILD=1 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$label -- $B > $OUT/pmc_$label.log 2>&1
  python $REPO/tools/pmc_summary.py $(find /tmp/pmc_$label -name "*.db" | head -1) | grep -i "$pat"
done
