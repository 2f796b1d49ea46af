#!/usr/bin/env python
"""bench.py - training rays/s of the NVFi hot path on MI355X (BASELINE.json metric).

One "step" = one optimiser iteration of the reference's --static_dynamic loop (train_nvfi.py:139-249)
on the synthetic 'bat' scene of BASELINE.md: a 2048-ray render at a random non-keyframe time, a
2048-ray render at a keyframe time, the plane regularisers, the velocity PDE loss on 262144
collocation points, backward of all of it and the Adam step.  value = rays rendered per second
(2 x 2048 per step per GPU), whole job.

  python bench.py [--gpus N --steps K --warmup W --workload cfg3|cfg2]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VEL_FLOP = 139776          # 2*(28*128 + 4*128^2 + 128*6)   one VelBasis net evaluation
APP_FLOP = 64768           # 2*(48*32 + 110*128 + 128^2 + 128*3)
PEAK_FP32_MFMA = 157.3     # TFLOP/s, MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
PEAK_HBM_GBS = 8000.0      # GB/s, MI355X_MICROARCH.md (HBM3E)
CLASSES = ["rk2_fwd", "rk2_bwd", "app_fwd", "app_bwd", "wgrad", "pde_fwd", "pde_bwd", "density_fwd", "density_bwd", "pde_prefilter", "density_scatter", "app_scatter", "other"]


def bat_cfg(S=128, use_vel=True):
    from nvfi_amd.utils import CfgNode
    import yaml
    # values of config/InDoorObj/bat.yaml that reach the hot path, restated
    n = dict(model_name="TensorVMKeyframeTimeKplane", density_n_comp=[24, 24, 24], appearance_n_comp=[48, 48, 48], app_dim=32,
             densityMode="Density", shadingMode="MLP_PE", alphaMask_thres=1e-4, rayMarch_weight_thres=1e-4, density_shift=-10,
             distance_scale=25, pos_pe=6, view_pe=6, fea_pe=6, featureC=128, step_ratio=0.5, fea2denseAct="softplus",
             max_n_samples=S, num_keyframes=16, num_keyframes_end=16, tmax=0.75, use_vel=use_vel)
    return CfgNode(dict(nvfi=n))


def build_scene(device, G=199, S=128, use_vel=True, seed=233):
    """Synthetic 'bat' field: reference-style init + the Gaussian blob recipe of BASELINE.md section 2."""
    from nvfi_amd.models import NVFi
    torch.manual_seed(seed)
    cfg = bat_cfg(S, use_vel)
    aabb = torch.tensor([[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]])
    near, far = 1.0, 8.0
    # S samples must span near->far: stepSize = min((far-near)/S, 0.999*diag/(S-1))
    units = 4.0 / (G - 1)
    diag = float(np.sqrt(3 * 16.0))
    step = min((far - near) / S, 0.999 * diag / (S - 1))
    cfg.nvfi.step_ratio = step / units
    m = NVFi(cfg, "cpu", aabb, [G, G, G], [near, far])
    f = m.nvfi
    with torch.no_grad():
        for i in range(3):
            p = f.density_plane_space[i]
            H, W = p.shape[-2:]
            yy = torch.linspace(-1, 1, H)[:, None]
            xx = torch.linspace(-1, 1, W)[None, :]
            blob = 3.2 * torch.sqrt(torch.exp(-xx ** 2 / (2 * 0.35 ** 2)) * torch.exp(-yy ** 2 / (2 * 0.35 ** 2)))
            p.mul_(blob[None, None])
    m = m.to(device)
    assert f.nSamples == S, (f.nSamples, S)
    return m


def pose_spherical(theta, phi, radius):
    th, ph = np.deg2rad(theta), np.deg2rad(phi)
    tr = np.eye(4); tr[2, 3] = radius
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1.0]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    return torch.tensor(flip @ rt @ rp @ tr, dtype=torch.float32)


def camera_bundle(device, H=800, W=800, angle_x=0.6911112):
    from nvfi_amd.models import Camera
    focal = 0.5 * W / np.tan(0.5 * angle_x)
    cam = Camera(pose_spherical(30.0, -30.0, 4.0).to(device), H, W, focal, torch.zeros(1, 1, 3, device=device), 1.0, 8.0)
    return cam.rays.ray_origins.reshape(-1, 3).contiguous(), cam.rays.ray_directions.reshape(-1, 3).contiguous()


class Step:
    """One training iteration of the hot path (mirrors train_nvfi.py:139-249 with --static_dynamic)."""

    def __init__(self, model, device, n_rays, n_pts, world, rank, workload):
        from nvfi_amd.models import Renderer
        from nvfi_amd.utils import TVLoss
        from nvfi_amd.dist import GradBucket
        self.m, self.dev, self.n_rays, self.n_pts, self.world, self.rank = model, device, n_rays, n_pts, world, rank
        self.workload = workload
        self.ren = Renderer(model, 0, 0, n_rays)
        self.tv = TVLoss()
        groups = model.get_optparam_groups(0.02, 1e-3)
        if os.environ.get("NVFI_TORCH_ADAM"):    # A/B: PyTorch's own fused Adam
            self.opt = torch.optim.Adam(groups, betas=(0.9, 0.99), fused=True)
        else:                                    # same update rule in one HIP launch (nvfi_adam_step); also clears the gradients
            from nvfi_amd.optim import Adam
            self.opt = Adam(groups, betas=(0.9, 0.99))
        self.fused_zero = not os.environ.get("NVFI_TORCH_ADAM")
        self.bucket = GradBucket([p for g in self.opt.param_groups for p in g["params"]])
        self.o, self.d = camera_bundle(device)
        self.gen = torch.Generator(device=device); self.gen.manual_seed(233 + rank)
        self.rng = np.random.default_rng(233 + rank)
        self.lr_factor = 0.1 ** (1 / 30000)
        self.L1w, self.tvd, self.tva, self.vw = 8e-4, 1.0, 1.0, 1.0
        self.counters = []
        self.pde_counters = []
        self.stepped = False
        self.fused_regs = True
        self.inplace = True
        from nvfi_amd.dist import PdeGradStage
        self.pde_stage = PdeGradStage(model.nvfi._pde_params()) if world > 1 else None
        self.tail_off = self.bucket.tail_offset(list(model.nvfi.vel.parameters())) if (world > 1 and model.nvfi.use_vel) else None
        model.nvfi.accumulate_grads_inplace = True   # .grad tensors are views of the GradBucket's flat buffer

    def rays(self):
        from nvfi_amd.models import Ray
        idx = torch.randint(0, self.o.shape[0], (self.n_rays,), device=self.dev, generator=self.gen)
        target = torch.rand(self.n_rays, 3, device=self.dev, generator=self.gen)
        return Ray(self.o[idx], self.d[idx], 1.0, 8.0), target

    def __call__(self):
        from nvfi_amd.dist import pde_rank_weight
        m, f = self.m, self.m.nvfi
        if not m.training:
            m.train()
        if not (self.fused_zero and self.stepped):   # after the first step the Adam launch has already cleared the gradients
            self.bucket.zero()
        loss = 0

        def pde_term():
            self.vw *= self.lr_factor
            # same term as `loss += vw * get_vel_loss()`, fused: d(vw * loss_vel) is accumulated by the PDE kernels - straight into
            # .grad on one GPU, through a small staging buffer re-weighted by W*n_r/sum(n_r) (device-side) on several
            m.vel_loss_weight = self.vw
            if self.world > 1:
                self.pde_stage.zero()
                m.vel_grad_targets = self.pde_stage.views
            self.last_lv = m.get_vel_loss(self.n_pts)
            self.pde_counters.append(f.last_pde_counters)
            if self.world > 1:
                self.pde_stage.commit(f.last_pde_n_kept)

        # One GPU: the PDE term goes first - its one host sync (kept count) then waits on nothing that is already queued, and the
        # renders + backward + Adam that follow are launched without any host wait.  Several GPUs: it goes LAST, so that the
        # all-reduce of the plane / render-MLP gradients (38 MB, final after the renders) runs underneath it.
        overlap = self.world > 1 and self.tail_off is not None
        if self.workload == "cfg3" and not overlap:
            pde_term()
        if self.workload == "cfg3":
            i = int(self.rng.integers(0, 46))
            while i % 3 == 0:                       # frame times i/60; keyframes every 0.05 = 3/60
                i = int(self.rng.integers(0, 46))
            rays, target = self.rays()
            out = self.ren.render(i / 60.0, rays, white_background=True, mode="train")
            loss = loss + torch.nn.functional.mse_loss(out[0], target)
            self.counters.append(f.last_counters)
            t_key = 3 * int(self.rng.integers(0, 16)) / 60.0
        else:
            t_key = float(self.rng.integers(0, 46)) / 60.0   # radiance-only: continuous time rows
        rays, target = self.rays()
        out = self.ren.render(t_key, rays, white_background=True, mode="train")
        loss = loss + torch.nn.functional.mse_loss(out[0], target)
        self.counters.append(f.last_counters)
        self.L1w *= self.lr_factor; self.tvd *= self.lr_factor; self.tva *= self.lr_factor
        if not self.fused_regs:
            loss = loss + self.L1w * f.density_L1() + self.tvd * f.TV_loss_density(self.tv) + self.tva * f.TV_loss_app(self.tv)
        loss.backward()
        if self.fused_regs:   # same regularisers + their gradients, fused into one pass per plane (nvfi_plane_regs)
            self.last_regs = f.regularizers_backward_(self.L1w, self.tvd, self.tva)
        if overlap:
            h = self.bucket.all_reduce_head_start(self.tail_off)
            if self.workload == "cfg3":
                pde_term()
            self.bucket.all_reduce_finish(h, self.tail_off)
        else:
            self.bucket.all_reduce_mean()
        if self.fused_zero:
            self.opt.step(zero_grad=True); self.stepped = True
        else:
            self.opt.step()
        for g in self.opt.param_groups:
            g["lr"] = g["lr"] * self.lr_factor
        return loss


def cpu_baseline(model, workload, seconds_hint=20):
    """Oracle (C restatement, OpenMP) on a bounded sample of the same step: 256+256 rays, P=32768."""
    from oracle import oracle as orc
    f = model.nvfi
    sd = {k[len("nvfi."):]: v.detach().cpu().contiguous().numpy() for k, v in model.state_dict().items() if not k.startswith("nvfi.vel.vel_net.")}
    meta = dict(aabb=f.aabb.cpu().numpy(), gridSize=np.array(f.gridSize.tolist()), num_keyframes=f.num_keyframes, tmax=f.tmax,
                near=f.near_far[0], far=f.near_far[1], step_ratio=f.step_ratio, max_n_samples=f.max_n_samples,
                density_shift=f.density_shift, distance_scale=f.distance_scale, alphaMask_thres=f.alphaMask_thres,
                rayMarch_weight_thres=f.rayMarch_weight_thres, stepSize=f._step_host, nSamples=f.nSamples, use_sur=0, eps=0.03)
    fs = orc.FieldSpec(sd, meta)
    # 64 threads is the oracle's sweet spot on the GPU box's 256-thread host (measured: 644 rays/s at 64, 300 at 256 on this sample)
    threads = min(64, os.cpu_count() or 1)
    orc.set_threads(threads)
    o, d = camera_bundle("cpu")
    rng = np.random.default_rng(5)
    R, P = 256, 32768
    def one_render(t):
        idx = rng.integers(0, o.shape[0], R)
        oo, dd = o[idx].numpy(), d[idx].numpy()
        u = rng.uniform(0, 1, (R, 1)).astype(np.float32)
        r = orc.render(fs, oo, dd, t, u=u, train=True, white_bg=True, keep_ctx=True, use_vel=(workload == "cfg3"))
        tg = rng.uniform(0, 1, (R, 3)).astype(np.float32)
        r.backward(fs, g_rgb=2 * (r.rgb - tg) / (R * 3))
        r.free()
    def step():
        if workload == "cfg3":
            one_render(19 / 60.0)
        one_render(0.30)
        if workload == "cfg3":
            mn, mx = fs.aabb
            pts = rng.uniform(0, 1, (P, 3)).astype(np.float32) * (mx - mn) + mn
            tt = rng.uniform(0, 1, (P, 1)).astype(np.float32)
            orc.pde_loss(fs, pts, tt)
    step()  # warm-up (page-in)
    t0 = time.perf_counter()
    n = 0
    while True:
        step(); n += 1
        if time.perf_counter() - t0 > seconds_hint * 0.5 or n >= 3:
            break
    dt = (time.perf_counter() - t0) / n
    rays = R * (2 if workload == "cfg3" else 1)
    return dict(value=rays / dt, unit="rays/s", cores=threads, kind="port",
                sample=f"oracle/nvfi_oracle.c (OpenMP x{threads}) on 1/8 of the step: {R} rays per render"
                       + (f" x2 renders + PDE with P={P}" if workload == "cfg3" else "") + f", fwd+bwd, {n} reps")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2"])
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--pts", type=int, default=262144)
    ap.add_argument("--grid", type=int, default=199)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the NVFi hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev          # (ranks share a device only in the single-GPU logic test, see NVFI_BENCH_BACKEND)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("NVFI_BENCH_BACKEND", "nccl")   # "nccl" = RCCL over xGMI; "gloo" only to exercise the path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    from nvfi_amd import _lib
    L = _lib.lib()

    model = build_scene(device, args.grid, args.samples, use_vel=True)
    if args.workload == "cfg2":
        model.nvfi.use_vel = False
    step = Step(model, device, args.rays, args.pts, world, rank, args.workload)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    step.counters.clear(); step.pde_counters.clear()
    barrier()
    L.nvfi_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ncls = L.nvfi_prof_nclasses()
    tot = (C.c_double * ncls)(); cnt = (C.c_int64 * ncls)()
    _lib.check(L.nvfi_prof_collect(tot, cnt))
    L.nvfi_prof_enable(0)
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    renders = 2 if args.workload == "cfg3" else 1
    rays_per_step = args.rays * renders * world
    value = rays_per_step * args.steps / dt

    # ---- roofline of the dominant kernel class (algorithmic MLP FLOPs / HIP-event time of that class)
    c = torch.stack(step.counters).sum(0).cpu().numpy() if step.counters else np.zeros(8)
    pc = torch.stack(step.pde_counters).sum(0).cpu().numpy() if step.pde_counters else np.zeros(8)
    V_evals, M, kept, pre_evals = float(c[3]), float(c[2]), float(pc[4]), float(pc[3])
    flops = {"rk2_fwd": V_evals * VEL_FLOP, "rk2_bwd": V_evals * VEL_FLOP, "app_fwd": M * APP_FLOP, "app_bwd": M * APP_FLOP,
             "wgrad": V_evals * VEL_FLOP + M * APP_FLOP + kept * 6 * VEL_FLOP, "pde_fwd": kept * 6 * VEL_FLOP,
             "pde_bwd": kept * 6 * VEL_FLOP, "pde_prefilter": pre_evals * VEL_FLOP}
    times = {CLASSES[i]: (tot[i], cnt[i]) for i in range(min(ncls, len(CLASSES)))}
    dom = max((k for k in flops if times.get(k, (0, 0))[1] > 0), key=lambda k: times[k][0], default=None)
    roof = None
    if dom:
        ms, n = times[dom]
        ach = flops[dom] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic = None
        try:   # HBM bytes per launch of that kernel class from the committed PMC passes (profiles/README.md)
            traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))["bytes_per_launch"].get(dom)
        except Exception:
            pass
        roof = dict(bound="mfma", kernel=dom, achieved=ach, peak=PEAK_FP32_MFMA, unit="TFLOP/s", frac=ach / PEAK_FP32_MFMA,
                    traffic=traffic, launches=int(n), avg_launch_ms=ms / max(n, 1), flop_per_launch=flops[dom] / max(n, 1),
                    per_class_ms_per_step={k: times[k][0] / args.steps for k in times})

    if args.workload == "cfg2":
        # radiance-only: gather / scatter bound: algorithmic bytes per step = 6912 B per valid sample + 13824 B per
        # appearance-masked sample (forward gathers + read-modify-write of the plane gradients, no reuse credit) + the per-ray
        # inputs/outputs, over the WHOLE step time (the MLP contractions ride along); peak = 8 TB/s HBM3E
        V = float(c[0])
        nbytes = V * 6912.0 + M * 13824.0 + args.steps * renders * args.rays * (24.0 + (5.0 + args.samples) * 4.0)
        gbs = nbytes / dt / 1e9
        roof = dict(bound="hbm", kernel="whole step: plane gathers + plane-gradient scatters (k_density_q, k_og, k_tile_scatter, k_app_fwd gather)",
                    achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS, traffic=None,
                    bytes_per_step=nbytes / args.steps, valid_samples_per_step=V / args.steps, masked_samples_per_step=M / args.steps,
                    mfma=roof, per_class_ms_per_step={k: times[k][0] / args.steps for k in times})
        if roof["mfma"]:
            roof["mfma"].pop("per_class_ms_per_step", None)

    out = {
        "metric": "training rays/sec (fwd+bwd incl. PDE loss), 'bat' scene", "value": value, "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("bat.yaml + velocity field + PDE divergence loss (configs[2]): 199^3 grid, K=16, 128 samples/ray, "
                                "2 renders x 2048 rays + PDE on 262144 collocation points + plane regularisers + Adam, per GPU"
                                if args.workload == "cfg3" else
                                "bat.yaml radiance-only (configs[1]): 199^3 grid, 128 samples/ray, 2048-ray batches of the 800x800 frame"),
                   "rays_per_step_per_gpu": args.rays * renders, "pde_points_per_gpu": args.pts if args.workload == "cfg3" else 0,
                   "grid": args.grid, "samples_per_ray": args.samples, "parallelism": f"ray-sharded x{world}"},
        "roofline": roof,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(model, args.workload)
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
