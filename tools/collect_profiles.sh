#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 passes of bench.py, summarised into gpurun_out/<tag>_*.
# usage: tools/collect_profiles.sh <tag> [quick]      (e.g. r06; quick: the headline set only)
set -u
TAG=${1:-r06}
QUICK=${2:-}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_*
B0="python $REPO/bench.py"
# 0. plain bench lines (no profiler attached): default (with the CPU baseline and the extras), other drivers / workloads
$B0 > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_line.err
$B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_eager.json 2>/dev/null
if [ -z "$QUICK" ]; then
$B0 --no-extras --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_bench_line_3steps.json 2>/dev/null
$B0 --no-extras --steps 100 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_line_100steps.json 2>/dev/null
NVFI_OVERLAP=0 $B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_one_stream.json 2>/dev/null
$B0 --no-extras --live --no-cpu-baseline > $OUT/${TAG}_bench_line_live.json 2>/dev/null
$B0 --no-extras --mode dropin --no-cpu-baseline > $OUT/${TAG}_bench_line_dropin.json 2>/dev/null
NVFI_DROPIN_FUSED_ADAM=1 $B0 --no-extras --mode dropin --no-cpu-baseline > $OUT/${TAG}_bench_line_dropin_fused_adam.json 2>/dev/null
$B0 --no-extras --workload cfg2 --graph off --steps 200 --warmup 50 --no-cpu-baseline > $OUT/${TAG}_bench_line_cfg2.json 2>/dev/null
$B0 --no-extras --workload cfg2 --graph on --steps 200 --warmup 50 --no-cpu-baseline > $OUT/${TAG}_bench_line_cfg2_hipgraph.json 2>/dev/null
# round 5: the launch chain of rounds 1-4 (count + k_fill launches, per-call fragment packing) on the same box
NVFI_FUSED_LAUNCH=0 NVFI_FRAG_CACHE=0 $B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_r4_launch_chain.json 2>/dev/null
NVFI_FUSED_LAUNCH=0 NVFI_FRAG_CACHE=0 $B0 --no-extras --graph off --no-cpu-baseline --rays 256 --pts 32768 > $OUT/${TAG}_bench_line_strong_shard_r4_launch_chain.json 2>/dev/null
NVFI_RK2_FUSE=0 $B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_rk2_unfused.json 2>/dev/null
NVFI_PDE_FUSE=0 $B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_pde_unfused.json 2>/dev/null
NVFI_VEL_FP16_TRAIN=1 $B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_fp16train.json 2>/dev/null
$B0 --workload chessboard > $OUT/${TAG}_bench_line_chessboard.json 2>/dev/null
$B0 --workload segm > $OUT/${TAG}_bench_line_segm.json 2>/dev/null
NVFI_MASK_FP16=1 NVFI_VEL_FP16=1 $B0 --workload segm --no-cpu-baseline > $OUT/${TAG}_bench_line_segm_fp16_mfma.json 2>/dev/null
$B0 --no-extras --graph off --no-cpu-baseline --rays 256 --pts 32768 > $OUT/${TAG}_bench_line_strong_shard_1of8.json 2>/dev/null
NVFI_PDE_PREFILTER=fp16band $B0 --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_line_fp16band.json 2>/dev/null
NVFI_PDE_PREFILTER=split16band $B0 --no-extras --no-cpu-baseline > $OUT/${TAG}_bench_line_split16band.json 2>/dev/null
python $REPO/tools/bench_extras.py eval_frame > $OUT/${TAG}_eval_frame.json 2>/dev/null
# the reference's chunk loop for the same frame (2048-ray calls on the current stream), and the frame with the fp32 warp
NVFI_EVAL_CHUNK=2048 NVFI_EVAL_STREAMS=1 python $REPO/tools/bench_extras.py eval_frame > $OUT/${TAG}_eval_frame_chunk2048.json 2>/dev/null
NVFI_RK2_X6=0 python $REPO/tools/bench_extras.py eval_frame > $OUT/${TAG}_eval_frame_fp32_warp.json 2>/dev/null
# every x6 kernel off (the fp32 MFMA kernels of round 4 for the render warp, its adjoint, the PDE prefilter and the Jacobian forward), and at one workgroup per CU
NVFI_RK2_X6=0 NVFI_PDE_PREFILTER=fp32 NVFI_FUSE_X6=0 NVFI_PDE_JET_X6=0 NVFI_INTEGRATE_X6=0 $B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_fp32_mfma_only.json 2>/dev/null
# round 6 A/B: the fused RK2 adjoint / the Jacobian forward back on their fp32 MFMA kernels, one at a time
NVFI_FUSE_X6=0 $B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_fuse_fp32.json 2>/dev/null
NVFI_PDE_JET_X6=0 $B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_bench_line_jet_fp32.json 2>/dev/null
$B0 --no-extras --graph on --no-cpu-baseline --rays 256 --pts 32768 > $OUT/${TAG}_bench_line_strong_shard_1of8_hipgraph.json 2>/dev/null
fi
# 1. kernel trace of the default bench command: per-kernel statistics of the whole run and of the profiled pass (after the marker)
export NVFI_BENCH_CHILD=1
rocprofv3 --kernel-trace -d /tmp/prof_k -- $B0 --no-extras --graph off --no-cpu-baseline > $OUT/${TAG}_prof.log 2>&1
grep -a "^{" $OUT/${TAG}_prof.log > $OUT/${TAG}_bench_line_rocprof.json
DB=$(find /tmp/prof_k -name "*.db" | head -1)
python $REPO/tools/rocpd_stats.py $DB $OUT/${TAG}_kernel_stats_whole_run.csv > /dev/null
python $REPO/tools/rocpd_stats.py $DB $OUT/${TAG}_kernel_stats.csv --after-marker > /dev/null
python $REPO/tools/timeline.py $DB $OUT/${TAG}_timeline.csv > /dev/null
# 2./3. HBM traffic: separate counter passes (kernel trace only); counter collection serialises the dispatches
B="$B0 --no-extras --graph off --steps 3 --warmup 1 --prime 1 --profile-steps 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f -- $B > /dev/null 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_f -name "*.db" | head -1) $OUT/${TAG}_pmc_fetch_size.csv > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_w -- $B > /dev/null 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_w -name "*.db" | head -1) $OUT/${TAG}_pmc_write_size.csv > /dev/null
# 4. SQ counters (MFMA utilisation, waits)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d /tmp/prof_s -- $B > /dev/null 2>&1
python $REPO/tools/pmc_summary.py $(find /tmp/prof_s -name "*.db" | head -1) $OUT/${TAG}_pmc_sq.csv > /dev/null
python $REPO/tools/make_traffic.py $OUT/${TAG}_pmc_fetch_size.csv $OUT/${TAG}_pmc_write_size.csv $OUT/${TAG}_traffic.json "bench.py --no-extras --steps 3 --warmup 1" > /dev/null
# 5. round 5 (verdict item 4a): the same three passes for the other BASELINE workloads - kernel statistics of the profiled pass and the
#    counter traffic of the WHOLE step (cfg2's roofline.traffic is this figure)
if [ -z "$QUICK" ]; then
for W in cfg2 chessboard segm; do
  WB="$B0 --workload $W --no-extras --graph off --no-cpu-baseline"
  rm -rf /tmp/prof_wk /tmp/prof_wf /tmp/prof_ww
  rocprofv3 --kernel-trace -d /tmp/prof_wk -- $WB > /dev/null 2>&1
  DBW=$(find /tmp/prof_wk -name "*.db" | head -1)
  if [ "$W" = "segm" ]; then python $REPO/tools/rocpd_stats.py $DBW $OUT/${TAG}_kernel_stats_$W.csv > /dev/null
  else python $REPO/tools/rocpd_stats.py $DBW $OUT/${TAG}_kernel_stats_$W.csv --after-marker > /dev/null; fi
  WC="$WB --steps 3 --warmup 1 --prime 1 --profile-steps 0"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_wf -- $WC > /dev/null 2>&1
  python $REPO/tools/pmc_summary.py $(find /tmp/prof_wf -name "*.db" | head -1) $OUT/${TAG}_pmc_fetch_size_$W.csv > /dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_ww -- $WC > /dev/null 2>&1
  python $REPO/tools/pmc_summary.py $(find /tmp/prof_ww -name "*.db" | head -1) $OUT/${TAG}_pmc_write_size_$W.csv > /dev/null
  # steps of the counter pass: prime 1 + warm-up 1 + 3 timed (+ 2 more priming steps of the segm driver); counted from a once-per-step kernel
  MARK="k_adam"; [ "$W" = "segm" ] && MARK="k_maskfield_fwd"
  python $REPO/tools/make_traffic.py --whole-step $OUT/${TAG}_pmc_fetch_size_$W.csv $OUT/${TAG}_pmc_write_size_$W.csv $OUT/${TAG}_traffic_$W.json "bench.py --workload $W --steps 3 --warmup 1 --prime 1" 0 "$MARK" > /dev/null
done
fi
ls -la $OUT | grep ${TAG}_ | wc -l
