#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 passes of bench.py, summarised into gpurun_out/<tag>_*.
# usage: tools/collect_profiles.sh <tag>      (e.g. r02)
set -u
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k /tmp/prof_k16 /tmp/prof_f /tmp/prof_w /tmp/prof_s
# 0. plain bench lines (no profiler attached): default (with the CPU baseline), other drivers / workloads
python $REPO/bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
python $REPO/bench.py --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_eager.json 2>/dev/null
python $REPO/bench.py --no-extras --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_line_3steps.json 2>/dev/null
python $REPO/bench.py --no-extras --steps 100 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_line_100steps.json 2>/dev/null
NVFI_OVERLAP=0 python $REPO/bench.py --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_one_stream.json 2>/dev/null
python $REPO/bench.py --no-extras --live --no-cpu-baseline > $OUT/${TAG}_bench_line_live.json 2>/dev/null
python $REPO/bench.py --no-extras --mode dropin --no-cpu-baseline > $OUT/${TAG}_bench_line_dropin.json 2>/dev/null
NVFI_DROPIN_FUSED_ADAM=1 python $REPO/bench.py --no-extras --mode dropin --no-cpu-baseline > $OUT/${TAG}_bench_line_dropin_fused_adam.json 2>/dev/null
NVFI_SPLIT_VOUT=0 NVFI_SPLIT_NT=2 python $REPO/bench.py --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_prefilter_out_on_mfma.json 2>/dev/null
python $REPO/bench.py --no-extras --workload cfg2 --no-cpu-baseline > $OUT/${TAG}_bench_line_cfg2.json 2>/dev/null
NVFI_WGRAD=engine python $REPO/bench.py --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_wgrad_engine.json 2>/dev/null
# round 4: the round-3 pair of kernels (k_rk2_split_bwd + k_wgrad_ring8 over the full adjoint stash) instead of vel_fuse.hip, same box; the other BASELINE configs
NVFI_RK2_FUSE=0 python $REPO/bench.py --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_rk2_unfused.json 2>/dev/null
NVFI_PDE_FUSE=0 python $REPO/bench.py --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_pde_unfused.json 2>/dev/null
NVFI_VEL_FP16_TRAIN=1 python $REPO/bench.py --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_fp16train.json 2>/dev/null
python $REPO/bench.py --workload chessboard > $OUT/${TAG}_bench_line_chessboard.json 2>/dev/null
python $REPO/bench.py --workload segm > $OUT/${TAG}_bench_line_segm.json 2>/dev/null
NVFI_MASK_FP16=1 NVFI_VEL_FP16=1 python $REPO/bench.py --workload segm --no-cpu-baseline > $OUT/${TAG}_bench_line_segm_fp16_mfma.json 2>/dev/null
python $REPO/bench.py --no-extras --graph off --no-cpu-baseline --rays 256 --pts 32768 > $OUT/${TAG}_bench_line_strong_shard_1of8.json 2>/dev/null
# other kernel selections: opt-in fp16-input pre-pass with the fp32 band; the engine kernels of vel.hip instead of vel_split.hip
NVFI_PDE_PREFILTER=fp16band python $REPO/bench.py --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_line_fp16band.json 2>/dev/null
NVFI_PDE_PREFILTER=split16band python $REPO/bench.py --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_line_split16band.json 2>/dev/null
NVFI_PDE_PREFILTER=engine32 NVFI_RK2_SPLIT=0 NVFI_RK2_SPLIT_BWD=0 python $REPO/bench.py --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_line_engine32.json 2>/dev/null
NVFI_BENCH_CHILD=1 rocprofv3 --kernel-trace -d /tmp/prof_k16 -- env NVFI_PDE_PREFILTER=fp16band python $REPO/bench.py --no-extras --graph off --no-cpu-baseline > /dev/null 2>&1
python $REPO/tools/rocpd_stats.py $(find /tmp/prof_k16 -name "*.db" | head -1) $OUT/${TAG}_kernel_stats_fp16band.csv --after-marker > /dev/null
# 1. kernel trace of the default bench command: per-kernel statistics of the whole run and of the profiled pass (after the marker)
NVFI_BENCH_CHILD=1 rocprofv3 --kernel-trace -d /tmp/prof_k -- python $REPO/bench.py --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
grep -a "^{" $OUT/${TAG}_prof.log > $OUT/${TAG}_bench_line_rocprof.json
DB=$(find /tmp/prof_k -name "*.db" | head -1)
python $REPO/tools/rocpd_stats.py $DB $OUT/${TAG}_kernel_stats_whole_run.csv > /dev/null
python $REPO/tools/rocpd_stats.py $DB $OUT/${TAG}_kernel_stats.csv --after-marker > /dev/null
python $REPO/tools/timeline.py $DB $OUT/${TAG}_timeline.csv > /dev/null
# 2./3. HBM traffic: separate counter passes (kernel trace only); counter collection serialises the dispatches
export NVFI_BENCH_CHILD=1
B="python $REPO/bench.py --no-extras --graph off --steps 3 --warmup 1 --prime 1 --profile-steps 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -- $B > /dev/null 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_f -name "*.db" | head -1) $OUT/${TAG}_pmc_fetch_size.csv > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -- $B > /dev/null 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_w -name "*.db" | head -1) $OUT/${TAG}_pmc_write_size.csv > /dev/null
# 4. SQ counters (MFMA utilisation, waits)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/prof_s -- $B > /dev/null 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_s -name "*.db" | head -1) $OUT/${TAG}_pmc_sq.csv > /dev/null
python $REPO/tools/make_traffic.py $OUT/${TAG}_pmc_fetch_size.csv $OUT/${TAG}_pmc_write_size.csv $OUT/${TAG}_traffic.json "bench.py --no-extras --steps 3 --warmup 1" > /dev/null
ls -la $OUT | grep ${TAG}_
