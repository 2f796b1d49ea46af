#!/usr/bin/env python
"""The `extras` of bench.py's default line: other lines of the same build, each measured in its own child process and attached to the
headline so that the driver records them with it.  None of them is `value`.

    collect(args, out, model)           called by bench.py's default invocation (rank 0, one GPU, fused driver, bat scene)
    python tools/bench_extras.py eval_frame [--steps N]     one JSON line: a full 800x800 test-mode frame

Round 5: moved out of bench.py's main(); the opt-in variants that only differ by an environment switch share fewer, shorter children."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BENCH = os.path.join(ROOT, "bench.py")


def run_child(args, extra_args, env=None, steps_factor=1, scene_args=True, profile=False, timeout=300, full=False, script=BENCH):
    """one bench.py (or bench_extras.py) child; returns the entry of `extras` built from its JSON line"""
    # (the radiance-only step is 0.8 ms: K = 20 steps are a 16 ms timed region, shorter than the clock ramp of an idle GPU - 10 K there)
    cmd = [sys.executable, script]
    if script == BENCH:
        cmd += ["--steps", str(args.steps * steps_factor), "--warmup", str(args.warmup * steps_factor),
                "--profile-steps", "5" if (profile or full) else "0", "--no-extras"] + ([] if full else ["--no-cpu-baseline"])
        if scene_args:
            cmd += ["--rays", str(args.rays), "--pts", str(args.pts), "--grid", str(args.grid), "--samples", str(args.samples)]
    cmd += extra_args
    try:
        r = subprocess.run(cmd, env=dict(os.environ, NVFI_BENCH_CHILD="1", **(env or {})), stdout=subprocess.PIPE, text=True, timeout=timeout)
        ln = [x for x in r.stdout.splitlines() if x.startswith("{")]
        d = json.loads(ln[-1])
        if script != BENCH:
            return d
        e = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "launch": d["config"].get("launch", "eager launches").split(" (")[0]}
        if profile:
            e["_line"] = d
        if full:        # a workload of its own: its roofline and CPU baseline travel with it
            r_ = d.get("roofline") or {}
            e["roofline"] = {k: r_.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic") if k in r_}
            e["cpu_baseline"] = d.get("cpu_baseline")
            e["workload"] = d["config"]["workload"]
        return e
    except Exception as e:
        return {"error": repr(e)}


def strong_shard(args, out, model):
    """the step one of 8 ranks runs under --scaling strong (256 rays per render, 32768 collocation points), on this one GPU: its time, its
    per-class times and launch count, and what they project for the 8-GPU step"""
    e = run_child(args, ["--rays", str(max(1, args.rays // 8)), "--pts", str(max(128, args.pts // 8)), "--grid", str(args.grid), "--samples", str(args.samples),
                         "--graph", "off"], scene_args=False, profile=True)
    d = e.pop("_line", None)
    if d is None:
        return e
    pc = (d.get("roofline") or {}).get("per_class", {})
    e["per_class_ms"] = {k: round(v["ms_per_step"], 4) for k, v in pc.items() if v.get("launches_per_step")}
    e["kernel_class_launches_per_step"] = sum(v.get("launches_per_step", 0) for v in pc.values())
    e["serial_ms_per_step"] = (d.get("work_per_step") or {}).get("ms_per_step_profiled_serial")
    # projection for 8 GPUs, strong scaling: the shard step + the gradient exchange of the flat buffer.  Ring all-reduce of S bytes over N
    # ranks moves 2 (N - 1) / N x S per link direction; ONE ring at the 153 GB/s of one xGMI link is the conservative figure (RCCL can
    # stripe rings over the 7 links of the fully connected node).  The head of the buffer (planes + render MLP, 38 MB) is exchanged under
    # the PDE chain, so the exposed part lies between 0 and the whole transfer.
    nbytes = 4.0 * sum(p.numel() for p in model.parameters())
    t_ar = 2.0 * 7.0 / 8.0 * nbytes / 153e9 * 1e3
    t1, t8 = out["ms_per_step"], d["ms_per_step"]
    # the same shard replayed as ONE captured hipGraph (the shard is launch-bound: ~22 class launches for ~1 ms of device time)
    g = run_child(args, ["--rays", str(max(1, args.rays // 8)), "--pts", str(max(128, args.pts // 8)), "--grid", str(args.grid), "--samples", str(args.samples),
                         "--graph", "on"], scene_args=False)
    if "ms_per_step" in g:
        e["hipgraph_replay_ms_per_step"] = g["ms_per_step"]
        t8 = min(t8, g["ms_per_step"])
    e["projection_uses_ms_per_step"] = t8
    e["projection_8gpu_strong"] = {
        "gradient_bytes": nbytes, "ring_allreduce_ms_one_link": t_ar,
        "step_ms_if_exchange_hidden": t8, "step_ms_if_exchange_exposed": t8 + t_ar,
        "speedup_over_1gpu": [t1 / (t8 + t_ar), t1 / t8], "efficiency": [t1 / (t8 + t_ar) / 8.0, t1 / t8 / 8.0],
        "note": "a PROJECTION from one-GPU measurements (this pool has one GPU per box); north_star's target is >= 6x at 8 GPUs"}
    return e


def collect(args, out, model):
    me = os.path.abspath(__file__)
    X = lambda *a, **k: run_child(args, *a, **k)
    return {
        "dropin": dict(X(["--mode", "dropin"]), what="the loop body of the reference's train_nvfi.py:139-249 verbatim on the `models` alias (plain autograd, "
                       "torch.optim.Adam, reference-signature regularisers, the per-iteration .item() waits): what tools/run_reference_driver.py gets"),
        "dropin_fused_adam": dict(X(["--mode", "dropin"], {"NVFI_DROPIN_FUSED_ADAM": "1"}),
                                  what="the same loop with `fused=True` in the caller's torch.optim.Adam (tools/run_reference_driver.py --fused-adam; opt-in: "
                                       "the reference's call as written is the line above)"),
        "cfg2_radiance_only": dict(X(["--workload", "cfg2", "--graph", "off"], steps_factor=10), what="BASELINE configs[1]: bat.yaml radiance-only, 2048-ray batches, HBM-bound (gathers / scatters); 10 K steps, eager launches"),
        "cfg2_radiance_only_hipgraph": dict(X(["--workload", "cfg2", "--graph", "on"], steps_factor=10), what="the same step replayed as one captured hipGraph"),
        "eval_frame": dict(X([ "eval_frame"], script=me, timeout=600),
                           what="SURVEY 8d config 2's other half: one full 800x800 TEST-mode frame (640 000 rays, 128 samples per ray; train_nvfi.py:395-459 -> "
                                "Renderer.render(mode='test')) at a non-keyframe time with the velocity field, and radiance-only; ms per frame, rays/s, roofline"),
        "fp32_mfma_prefilter": dict(X(["--graph", "off"], {"NVFI_PDE_PREFILTER": "fp32"}),
                                    what="the default of rounds 2-4: the PDE occupancy prefilter on the fp32 MFMA kernel (vel_split.hip) instead of x6 (vel_x6.hip: the same fp32 "
                                         "products formed exactly from three bfloat16 terms per operand on the 16-bit matrix pipe)"),
        "fp32_mfma_only": dict(X(["--graph", "off"], {"NVFI_PDE_PREFILTER": "fp32", "NVFI_RK2_X6": "0", "NVFI_FUSE_X6": "0", "NVFI_INTEGRATE_X6": "0"}),
                               what="the conservative figure beside the headline (VERDICT r5): EVERY x6 kernel off - prefilter, render warp and the adjoint's dgrad all on "
                                    "the fp32 MFMA kernels of round 4 (v_mfma_f32_32x32x2_f32 only)"),
        "optin_split16band_prefilter": dict(X(["--graph", "off"], {"NVFI_PDE_PREFILTER": "split16band"}),
                                            what="opt-in (NOT the headline): the PDE occupancy prefilter with fp32 products emulated on the fp16 matrix pipe (two binary16 "
                                                 "terms per operand, three MFMAs, fp32 accumulation: ~2^-21 relative per product) + an fp32 re-evaluation band of 0.1 %; "
                                                 "identical kept set on every test field"),
        "optin_fp16_forward_warp": dict(X(["--graph", "off"], {"NVFI_VEL_FP16_TRAIN": "1"}),
                                        what="opt-in (NOT the headline; the reference's counterpart is --disable_fp32, train_nvfi.py:96,144): the velocity warp of training renders "
                                             "evaluates VelBasis FORWARD with fp16-input MFMAs (fp32 accumulation, fp32 stashes); adjoint and weight gradients stay fp32 MFMA on "
                                             "those stashes; digit-level parity with the oracle in the same arithmetic"),
        "live": dict(X(["--live", "--graph", "off"]), what="the optimiser moves the field it renders (train_nvfi.py:243), as in real training: on random targets the blob "
                     "thickens and the PDE kept set grows 34 k -> 114 k points within these steps, so the step is slower than the stationary headline (same kernels)"),
        "strong_shard_1of8": dict(strong_shard(args, out, model), what="the fused step at 1/8 of the global batch (--rays 256 --pts 32768: what each of 8 ranks runs under --scaling strong), one GPU"),
        "chessboard": dict(X(["--workload", "chessboard"], scene_args=False, timeout=600, full=True),
                           what="BASELINE configs[3], one GPU's share: the same loop on the InDoorSeg chessboard box at its final 199x199x200 grid (K = 4, surround-box gate "
                                "with step rejection, no white background, 688 samples per ray, P = 131072)"),
        "segm": dict(X(["--workload", "segm"], scene_args=False, full=True),
                     what="BASELINE configs[4], one GPU's share: train_segm.py's MaskField step on the fan scene (= the bat box, K = 16, bat.yaml:142-154: 64^3 lattice -> "
                          "occupied points -> integrate_pos over 20-30 RK2 steps -> MaskField fwd + bwd + Adam); points/s"),
        "segm_fp16_mfma": dict(X(["--workload", "segm"], {"NVFI_MASK_FP16": "1", "NVFI_VEL_FP16": "1"}, scene_args=False),
                               what="the same step with configs[4]'s 'fp16 MFMA MLP': MaskField forward / adjoint on v_mfma_f32_32x32x16_f16 (fp32 accumulation, fp32 stashes and "
                                    "weight gradients) and integrate_pos on the fp16-input inference kernel; opt-in, stated in dtype"),
    }


# ---------------------------------------------------------------- eval frame
def eval_frame_main(argv):
    import numpy as np
    import torch
    import bench
    from nvfi_amd.models import Ray, Renderer
    reps = 3
    if "--steps" in argv:
        reps = int(argv[argv.index("--steps") + 1])
    dev = torch.device("cuda", 0)
    m = bench.build_scene(dev, 199, 128, True)
    f = m.nvfi
    f.eval()
    o, d = bench.camera_bundle(dev)
    ren = Renderer(m, 0, 0, 2048)
    rays = Ray(o, d, 1.0, 8.0)
    n_rays = o.shape[0]

    def frame(t):
        return ren.render(t, rays, white_background=True, mode="test")

    def timed(t):
        frame(t); torch.cuda.synchronize()
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            frame(t)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    def work(t):
        """V / N / M of the whole frame from the chunks' device counters (one extra pass, not timed)"""
        tot = np.zeros(8)
        ro, rd = o, d
        with torch.no_grad():
            for c in range(0, n_rays, 65536):
                f(t, ro[c:c + 65536], rd[c:c + 65536], True)
                tot += f.last_counters.cpu().numpy()
        return tot

    res = {"what": "one 800x800 test-mode frame of the bat scene (199^3, K = 16, 128 samples per ray, no alpha mask), best of %d" % reps, "rays": n_rays}
    t_non = 19.0 / 60.0
    c = work(t_non)
    dt = timed(t_non)
    flop = c[3] * bench.VEL_FLOP + c[2] * bench.APP_FLOP
    res["velocity_nonkey_t19"] = {
        "ms_per_frame": dt * 1e3, "rays_per_s": n_rays / dt, "dtype": "f32",
        "work": {"valid_samples": c[0], "warped_samples": c[1], "masked_samples": c[2], "velocity_net_evaluations": c[3], "gflop": flop / 1e9},
        "roofline": {"bound": "mfma", "kernel": "whole frame: k_rk2_split_uni (RK2 warp, %d evaluations) + k_app_fwd" % int(c[3]), "achieved": flop / dt / 1e12,
                     "peak": bench.PEAK_FP32_MFMA, "unit": "TFLOP/s", "frac": flop / dt / 1e12 / bench.PEAK_FP32_MFMA, "traffic": None,
                     "floor_ms_at_peak": flop / (bench.PEAK_FP32_MFMA * 1e12) * 1e3}}
    f.use_vel = False
    c = work(t_non)
    dt = timed(t_non)
    nbytes = c[0] * 2304.0 + c[2] * 4608.0 + n_rays * (24.0 + (5.0 + 128) * 4.0)
    res["radiance_only"] = {
        "ms_per_frame": dt * 1e3, "rays_per_s": n_rays / dt, "dtype": "f32",
        "work": {"valid_samples": c[0], "masked_samples": c[2], "gflop_app": c[2] * bench.APP_FLOP / 1e9, "algorithmic_bytes": nbytes},
        "roofline": {"bound": "hbm", "kernel": "whole frame: plane gathers (k_density_q, k_app_fwd gather) + per-ray I/O", "achieved": nbytes / dt / 1e9,
                     "peak": bench.PEAK_HBM_GBS, "unit": "GB/s", "frac": nbytes / dt / 1e9 / bench.PEAK_HBM_GBS, "traffic": None}}
    f.use_vel = True
    res["value"] = res["velocity_nonkey_t19"]["rays_per_s"]; res["unit"] = "rays/s"; res["ms_per_frame"] = res["velocity_nonkey_t19"]["ms_per_frame"]
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "eval_frame":
        eval_frame_main(sys.argv[2:])
    else:
        raise SystemExit(__doc__)
