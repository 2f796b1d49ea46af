#!/usr/bin/env python
"""bench.py - training rays/s of the NVFi hot path on MI355X (BASELINE.json metric).

One "step" = one optimiser iteration of the reference's --static_dynamic loop (train_nvfi.py:139-249)
on the synthetic 'bat' scene of BASELINE.md: a 2048-ray render at a random non-keyframe time, a
2048-ray render at a keyframe time, the plane regularisers, the velocity PDE loss on 262144
collocation points, backward of all of it and the Adam step.  value = rays rendered per second
(2 x 2048 per step per GPU), whole job.

  python bench.py [--gpus N --steps K --warmup W --workload cfg3|cfg2 --mode fused|dropin --live --scaling weak|strong]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

The workload is STATIONARY by default: the optimiser step is executed in full every iteration, but on a shadow copy of the
parameters (same tensors, same kernels, same bytes), so the field - and with it the number of valid / appearance-masked
samples and of PDE points that survive the occupancy prefilter - is the BASELINE.md blob for every step, and a 3-step run
measures the same thing as a 200-step run.  `--live` applies the updates to the field itself (training on random targets thickens
the blob: the PDE kept set grows 34k -> 114k points within 20 steps, and the step time with it).

`--mode fused` (default) is this repository's counterpart driver of the loop: gradients accumulated in place into ONE flat
buffer, the regulariser value+gradient kernels, the one-launch Adam, the PDE term fused with its backward.
`--mode dropin` is the reference's loop body verbatim on the `models` alias: plain autograd, torch.optim.Adam, torch-op
regularisers, `if loss_vel > 0`, the `.item()` calls - what train_nvfi.py gets without touching it.

The timed region carries no instrumentation.  The roofline figures come from a second, profiled pass over the same number of
steps (HIP events around every launch of a kernel class, on the launch stream).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# ROCm 7's hipGraph "packet capture" fast path (AQL packets of a single-branch graph pre-built at instantiation) faults on the second
# replay of the captured radiance-only iteration at full size (Memory access fault, reproducible with tools/graph_debug.py cfg2); the
# ordinary graph launch path is unaffected and just as fast for these graphs.  Must be set before the HIP runtime starts.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def mse_loss(x, target):
    """the photometric loss of the fused driver: F.mse_loss with value and gradient from one launch (nvfi_mse); the drop-in loop keeps torch's"""
    from nvfi_amd.utils import mse_loss as f
    return f(x, target)


VEL_FLOP = 139776          # 2*(28*128 + 4*128^2 + 128*6)   one VelBasis net evaluation
APP_FLOP = 64768           # 2*(48*32 + 110*128 + 128^2 + 128*3)
PEAK_FP32_MFMA = 157.3     # TFLOP/s, MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA = 2500.0    # TFLOP/s dense, MI355X_MICROARCH.md (v_mfma_f32_32x32x16_bf16)
PEAK_HBM_GBS = 8000.0      # GB/s, MI355X_MICROARCH.md (HBM3E)
CLASSES = ["rk2_fwd", "rk2_bwd", "app_fwd", "app_bwd", "wgrad", "pde_fwd", "pde_bwd", "density_fwd", "density_bwd", "pde_prefilter", "density_scatter", "app_scatter", "other"]
PROFILE_TAG = "r06"


def bat_cfg(S=128, use_vel=True):
    from nvfi_amd.utils import CfgNode
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


def chessboard_cfg():
    from nvfi_amd.utils import CfgNode
    # values of config/InDoorSeg/chessboard.yaml that reach the hot path (reference lines 53, 64-68, 85-87, 101-103, 119-121, 137), restated
    n = dict(model_name="TensorVMKeyframeTimeKplane", density_n_comp=[24, 24, 24], appearance_n_comp=[48, 48, 48], app_dim=32,
             densityMode="Density", shadingMode="MLP_PE", alphaMask_thres=1e-4, rayMarch_weight_thres=1e-4, density_shift=-5,
             distance_scale=10, pos_pe=6, view_pe=6, fea_pe=6, featureC=128, step_ratio=0.5, fea2denseAct="softplus",
             max_n_samples=1024, num_keyframes=4, num_keyframes_end=4, tmax=0.75, use_vel=True,
             sur_x=[-2.5, 2.5], sur_y=[-2.5, 2.5], sur_z=[0.02, 5.95])
    return CfgNode(dict(nvfi=n))


def build_scene_chessboard(device, final=True, seed=233):
    """BASELINE configs[3] ('chessboard dynamic indoor scene'): the InDoorSeg box (x, y in +-3.03, z in [-0.03, 6.03]), surround-box velocity gate
    with step rejection, K = 4 keyframes, no white background; a Gaussian blob in the density planes (the initial field is transparent) and a
    velocity head scaled so that the gate is exercised - the recipe of tests/test_gpu_fullsize_chessboard.py.  final: the 64^3 field upsampled to
    the LAST entry of the shipped voxel schedule, 199 x 199 x 200 (688 samples per ray: 1.4 M samples per 2048-ray render); else 64^3 / 219."""
    from nvfi_amd.models import NVFi
    from nvfi_amd.utils import N_to_reso
    torch.manual_seed(seed)
    aabb = torch.tensor([[-3.03, -3.03, -0.03], [3.03, 3.03, 6.03]])
    m = NVFi(chessboard_cfg(), "cpu", aabb, [64, 64, 64], [0.8, 8.1])
    f = m.nvfi
    with torch.no_grad():
        for i in range(3):
            p = f.density_plane_space[i]
            H, W = p.shape[-2:]
            yy = torch.linspace(-1, 1, H)[:, None]; xx = torch.linspace(-1, 1, W)[None, :]
            p.mul_((2.6 * torch.sqrt(torch.exp(-xx ** 2 / (2 * 0.35 ** 2)) * torch.exp(-yy ** 2 / (2 * 0.35 ** 2))))[None, None])
        f.vel_net.weight_net[7][0].weight.mul_(4.0)
        f.vel_net.weight_net[7][0].bias.add_(0.3)
    m = m.to(device)
    if final:
        f.upsample_volume_grid(N_to_reso(7999998, f.aabb.cpu()), 4)
        assert f.nSamples == 688, f.nSamples
    else:
        assert f.nSamples == 219, f.nSamples
    return m


def camera_bundle_chessboard(device, H=800, W=800, angle_x=0.6911112):
    """800 x 800 pinhole camera on a sphere of radius 4 around the centre of the chessboard box (0, 0, 3) (assumed pose)"""
    from nvfi_amd.models import Camera
    focal = 0.5 * W / np.tan(0.5 * angle_x)
    pose = pose_spherical(30.0, -30.0, 4.0)
    pose[2, 3] += 3.0
    cam = Camera(pose.to(device), H, W, focal, torch.zeros(1, 1, 3, device=device), 0.8, 8.1)
    return cam.rays.ray_origins.reshape(-1, 3).contiguous(), cam.rays.ray_directions.reshape(-1, 3).contiguous()


SCENES = {
    # keyframes every tmax / (K - 1): bat 0.05 = 3 / 60, chessboard 0.25 = 15 / 60 (frame times are i / 60, i = 0..45)
    "bat": dict(white_bg=True, kf_stride=3, K=16, near=1.0, far=8.0, pts=262144),
    "chessboard": dict(white_bg=False, kf_stride=15, K=4, near=0.8, far=8.1, pts=131072),
}


def pose_spherical(theta, phi, radius):
    th, ph = np.deg2rad(theta), np.deg2rad(phi)
    tr = np.eye(4); tr[2, 3] = radius
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1.0]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    return torch.tensor(flip @ rt @ rp @ tr, dtype=torch.float32)


H_IMG = W_IMG = 800
ANGLE_X = 0.6911112


def camera_bundle(device, H=H_IMG, W=W_IMG, angle_x=ANGLE_X):
    from nvfi_amd.models import Camera
    focal = 0.5 * W / np.tan(0.5 * angle_x)
    cam = Camera(pose_spherical(30.0, -30.0, 4.0).to(device), H, W, focal, torch.zeros(1, 1, 3, device=device), 1.0, 8.0)
    return cam.rays.ray_origins.reshape(-1, 3).contiguous(), cam.rays.ray_directions.reshape(-1, 3).contiguous()


def _shadow_groups(groups):
    """Optimiser groups over clones of the parameters whose .grad IS the real parameter's .grad tensor: the optimiser step does all
    of its work (reads p, g, m, v; writes p, m, v) without moving the field that the next step renders."""
    out, pairs = [], []
    for g in groups:
        ps = []
        for p in g["params"]:
            s = torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format))
            pairs.append((p, s))
            ps.append(s)
        out.append({"params": ps, "lr": g["lr"]})
    return out, pairs


class Step:
    """One training iteration of the hot path (mirrors train_nvfi.py:139-249 with --static_dynamic), optimised counterpart driver."""

    def __init__(self, model, device, n_rays, n_pts, world, rank, workload, live=False, scene="bat"):
        from nvfi_amd.models import Renderer
        from nvfi_amd.utils import TVLoss
        from nvfi_amd.dist import GradBucket
        self.m, self.dev, self.n_rays, self.n_pts, self.world, self.rank = model, device, n_rays, n_pts, world, rank
        self.workload = workload
        self.scene = SCENES[scene]
        self.white_bg, self.kf = self.scene["white_bg"], self.scene["kf_stride"]
        self.ren = Renderer(model, 0, 0, n_rays)
        self.tv = TVLoss()
        groups = model.get_optparam_groups(0.02, 1e-3)
        groups = [dict(params=list(g["params"]), lr=g["lr"]) for g in groups]
        self.bucket = GradBucket([p for g in groups for p in g["params"]])
        opt_groups = groups
        if not live:
            opt_groups, pairs = _shadow_groups(groups)
            for p, s in pairs:
                s.grad = p.grad          # (p.grad may be None for parameters outside the bucket: frozen ones)
        if os.environ.get("NVFI_TORCH_ADAM"):    # A/B: PyTorch's own fused Adam
            self.opt = torch.optim.Adam(opt_groups, betas=(0.9, 0.99), fused=True)
        else:                                    # same update rule in one HIP launch (nvfi_adam_step); also clears the gradients
            from nvfi_amd.optim import Adam
            self.opt = Adam(opt_groups, betas=(0.9, 0.99))
        self.fused_zero = not os.environ.get("NVFI_TORCH_ADAM")
        self.o, self.d = camera_bundle(device) if scene == "bat" else camera_bundle_chessboard(device)
        self.gen = torch.Generator(device=device); self.gen.manual_seed(233 + rank)
        self.rng = np.random.default_rng(233 + rank)
        self.lr_factor = 0.1 ** (1 / 30000)
        self.L1w, self.tvd, self.tva, self.vw = 8e-4, 1.0, 1.0, 1.0
        self.counters = []
        self.pde_counters = []
        self.stepped = False
        self.fused_regs = True
        from nvfi_amd.dist import PdeGradStage
        self.pde_stage = PdeGradStage(model.nvfi._pde_params()) if world > 1 else None
        self.tail_off = self.bucket.tail_offset(list(model.nvfi.vel.parameters())) if (world > 1 and model.nvfi.use_vel) else None
        model.nvfi.accumulate_grads_inplace = True   # .grad tensors are views of the GradBucket's flat buffer
        # NVFI_ALLREDUCE=abi: the gradient exchange through nvfi_allreduce_grads (RCCL behind the C ABI) instead of torch.distributed
        self.comm = None
        if world > 1 and os.environ.get("NVFI_ALLREDUCE") == "abi":
            from nvfi_amd.dist import RcclComm
            self.comm = RcclComm()
        # The PDE term and the two renders of an iteration are independent until their gradients meet in the flat buffer (every
        # accumulation into it is atomic), so they are issued on three HIP streams: each chain's workgroups fill the CUs the others
        # leave idle in their tails (most engine kernels run one 128-sample workgroup per CU: a grid of 1055 workgroups is 4.1 rounds).
        self.streams = None
        # (several ranks: on by default with RCCL; gloo - the one-GPU logic tests - moves CUDA tensors through the host and serialises badly
        #  against side streams, so there it needs NVFI_OVERLAP=1)
        ov = os.environ.get("NVFI_OVERLAP")
        gloo = world > 1 and os.environ.get("NVFI_BENCH_BACKEND", "nccl") != "nccl"
        self.s_r2 = None
        if workload == "cfg3" and (ov == "1" or (ov is None and not gloo)):
            model.nvfi.fork_backward = False      # the three chains already fill the device: no library-side fork inside a render's backward
            # (experiment switches: NVFI_PRIO_PDE / NVFI_PRIO_R1 / NVFI_PRIO_R2 = HIP stream priorities, -1 = high; with NVFI_PRIO_R2 set the
            #  keyframe render runs on a third side stream of that priority instead of the current stream)
            pr = [int(os.environ.get(k, "0")) for k in ("NVFI_PRIO_PDE", "NVFI_PRIO_R1")]
            self.streams = [torch.cuda.Stream(device=device, priority=p) for p in pr]
            if os.environ.get("NVFI_PRIO_R2") is not None:
                self.s_r2 = torch.cuda.Stream(device=device, priority=int(os.environ["NVFI_PRIO_R2"]))

    # ---- the iteration's random inputs: ONE launch (nvfi_draw_batch) + ONE upload (the reference's CPU-generator jitter of both renders)
    def _draw_setup(self):
        from nvfi_amd import _lib
        R, P, dev = self.n_rays, self.n_pts, self.dev
        nb = self.renders = 2 if self.workload == "cfg3" else 1
        self.d_ro = [torch.empty(R, 3, device=dev) for _ in range(nb)]
        self.d_rd = [torch.empty(R, 3, device=dev) for _ in range(nb)]
        self.d_tg = [torch.empty(R, 3, device=dev) for _ in range(nb)]
        self.d_pts = torch.empty(P, 3, device=dev) if self.workload == "cfg3" else None
        self.d_t = torch.empty(P, device=dev) if self.workload == "cfg3" else None
        d = _lib.DrawDesc()
        d.seed = 233 + self.rank
        d.n_batches, d.R, d.n_pixels = nb, R, self.o.shape[0]
        d.bundle_o, d.bundle_d, d.target_img = _lib.ptr(self.o), _lib.ptr(self.d), None      # synthetic targets: uniform in [0, 1)
        for b in range(nb):
            d.rays_o[b], d.rays_d[b], d.target[b] = _lib.ptr(self.d_ro[b]), _lib.ptr(self.d_rd[b]), _lib.ptr(self.d_tg[b])
        if self.d_pts is not None:
            d.P = P
            d.aabb[:] = [float(v) for v in self.m.nvfi.aabb.reshape(-1).tolist()]
            d.points, d.t = _lib.ptr(self.d_pts), _lib.ptr(self.d_t)
        self.draw_desc = d
        self.draw_it = 0
        self.jit_dev = torch.empty(nb * R, device=dev)
        self.jit_ring = [[torch.empty(nb * R, 1).pin_memory(), None] for _ in range(4)]

    def draw(self, it_dev=None):
        """pixel batches + targets of the iteration's renders and the PDE collocation points / times: one launch on the current stream
        (it_dev: the iteration counter in device memory - hipGraph replay)"""
        from nvfi_amd import _lib
        d = self.draw_desc
        d.seed = self.seed_override if getattr(self, "seed_override", None) is not None else 233 + self.rank
        if it_dev is None:
            self.draw_it += 1
            d.iteration, d.iteration_dev = self.draw_it, None
        else:
            d.iteration_dev = _lib.ptr(it_dev)
        _lib.check(_lib.lib().nvfi_draw_batch(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def jitter(self):
        """the per-ray jitter of the iteration's renders - the reference's CPU-generator draw (tensorf_base.py:302-306), non-keyframe render
        first - through a ring of pinned buffers: one asynchronous upload per iteration"""
        slot = self.jit_ring[self.draw_it % len(self.jit_ring)]
        if slot[1] is not None:
            slot[1].synchronize()
        R = self.n_rays
        for k in range(self.renders):
            torch.rand(R, 1, out=slot[0][k * R:(k + 1) * R])
        self.jit_dev.copy_(slot[0].view(-1), non_blocking=True)
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record()
        return [self.jit_dev[k * R:(k + 1) * R] for k in range(self.renders)]

    def frame_times(self):
        """the host's draws of an iteration: a non-keyframe frame i/60 and a keyframe time (cfg3), or any frame time (radiance-only)"""
        if self.workload == "cfg3":
            i = int(self.rng.integers(0, 46))
            while i % self.kf == 0:                 # frame times i/60; keyframes every 0.05 = 3/60 (bat), 0.25 = 15/60 (chessboard)
                i = int(self.rng.integers(0, 46))
            return i / 60.0, self.kf * int(self.rng.integers(0, self.scene["K"])) / 60.0
        return None, float(self.rng.integers(0, 46)) / 60.0      # radiance-only: continuous time rows

    def _pde_term(self):
        m, f = self.m, self.m.nvfi
        self.vw *= self.lr_factor
        # same term as `loss += vw * get_vel_loss()`, fused: d(vw * loss_vel) is accumulated by the PDE kernels - straight into
        # .grad on one GPU, through a small staging buffer re-weighted by W*n_r/sum(n_r) (device-side) on several
        m.vel_loss_weight = self.vw
        if self.world > 1:
            self.pde_stage.zero()
            m.vel_grad_targets = self.pde_stage.views
        self.last_lv = m.get_vel_loss(self.n_pts, points=self.d_pts, t=self.d_t)
        self.pde_counters.append(f.last_pde_counters)
        return f.last_pde_out

    def _render_term(self, t, k, jit, wait=None):
        """one training render + its mse + the backward of both, gradients accumulated in place (field.render_mse_backward_)"""
        f = self.m.nvfi
        loss, _ = f.render_mse_backward_(t, self.d_ro[k], self.d_rd[k], self.d_tg[k], white_bg=self.white_bg, jitter=jit, wait_before_backward=wait)
        self.counters.append(f.last_counters)
        return loss

    def __call__(self):
        """Round 5: no torch launch and no autograd graph in the iteration.  Per step: one draw launch, one jitter upload, one fragment repack,
        the PDE term, the two renders (forward + mse + backward each), the regulariser pass, the Adam launch - on three streams when the
        driver overlaps the chains, the gradient exchange of several ranks in between."""
        m, f = self.m, self.m.nvfi
        if not m.training:
            m.train()
        if not (self.fused_zero and self.stepped):   # after the first step the Adam launch has already cleared the gradients
            self.bucket.zero()
        if getattr(self, "draw_desc", None) is None:
            self._draw_setup()
        main = torch.cuda.current_stream()
        multi = self.world > 1
        cfg3 = self.workload == "cfg3"
        t1, t_key = self.frame_times()
        self.draw()
        jit = self.jitter()
        f.repack_frags()           # the weights changed in the last optimiser step: every fragment set in one launch, before the chains fork
        self.L1w *= self.lr_factor; self.tvd *= self.lr_factor; self.tva *= self.lr_factor
        streams = self.streams if cfg3 else None
        pde_out, early_regs = None, False
        if streams is not None:
            # PDE term | non-keyframe render | keyframe render on three streams; the regularisers and the optimiser step follow on the main stream
            # once the three have joined (several ranks: the all-reduce of the renders' share starts underneath the PDE chain)
            s_pde, s_r1 = streams
            start = torch.cuda.Event(); start.record(main)
            with torch.cuda.stream(s_pde):
                s_pde.wait_event(start)
                pde_out = self._pde_term()
            with torch.cuda.stream(s_r1):
                s_r1.wait_event(start)
                self._render_term(t1, 0, jit[0])
            if self.s_r2 is not None:
                with torch.cuda.stream(self.s_r2):
                    self.s_r2.wait_event(start)
                    loss = self._render_term(t_key, 1, jit[1])
                main.wait_stream(self.s_r2)
            else:
                loss = self._render_term(t_key, 1, jit[1])
            main.wait_stream(s_r1)
            if not multi:
                main.wait_stream(s_pde)
        else:
            # one stream.  Several GPUs: the PDE term goes LAST, so that the all-reduce of the plane / render-MLP gradients (38 MB, final after
            # the renders) runs underneath it; one GPU: first (order is immaterial: no call on this path waits for the device)
            overlap1 = multi and self.tail_off is not None and self.comm is None
            if cfg3 and not overlap1:
                pde_out = self._pde_term()
            if cfg3:
                self._render_term(t1, 0, jit[0])
            # single chain: the regulariser pass (a plain read-modify-write of the plane gradients; they are zero here - the Adam launch cleared them)
            # runs on a side stream beside the forward half of the render; the backward half, whose scatters add atomically on top, waits for it
            early_regs = not cfg3 and not multi and self.fused_zero and self.stepped and os.environ.get("NVFI_EARLY_REGS", "1") != "0"
            if early_regs:
                if getattr(self, "_s_reg", None) is None:
                    self._s_reg = torch.cuda.Stream(device=self.dev)
                self._s_reg.wait_stream(main)
                with torch.cuda.stream(self._s_reg):
                    self.last_regs = f.regularizers_backward_(self.L1w, self.tvd, self.tva)
            loss = self._render_term(t_key, self.renders - 1, jit[-1], wait=self._s_reg if early_regs else None)
        if not (streams is None and early_regs):
            self.last_regs = f.regularizers_backward_(self.L1w, self.tvd, self.tva)
        if multi:
            split = self.tail_off is not None and self.comm is None
            h = self.bucket.all_reduce_head_start(self.tail_off) if split else None
            if streams is not None:
                main.wait_stream(streams[0])
            elif cfg3 and pde_out is None:
                pde_out = self._pde_term()
            if cfg3:
                self.pde_stage.commit_device(pde_out)
            if split:
                self.bucket.all_reduce_finish(h, self.tail_off)
            else:
                self.bucket.all_reduce_mean(self.comm)
        if self.fused_zero:
            self.opt.step(zero_grad=True); self.stepped = True
        else:
            self.opt.step()
        for g in self.opt.param_groups:
            g["lr"] = g["lr"] * self.lr_factor
        return loss


class GraphedStep:
    """The same iteration as Step (three streams), captured ONCE as a hipGraph and replayed: per iteration the host draws what the
    reference draws on the host (frame index, keyframe index, the CPU-generator jitter of both renders), advances the decaying loss
    weights / learning rates / Adam bias corrections, writes all of it into one pinned record, uploads it with one asynchronous copy and
    launches the graph.  Everything the kernels need per iteration is read from that record in device memory:
      rec[0] t of the non-keyframe render, rec[1] t of the keyframe render   (DeviceTime -> nvfi_render_fwd_t / _bwd_t: keyframe row,
                                                                              RK2 step sizes and times are derived on the device)
      rec[2:5] L1 / TV-density / TV-app weights (nvfi_plane_regs_dev), rec[5] PDE weight (nvfi_pde_loss_dev)
      rec[16:16+1+n] Adam: 1/sqrt(1-b2^t), lr_k/(1-b1^t) per tensor (nvfi_adam_step_dev)
      then the two per-ray jitter vectors.
    Ray indices / targets and the collocation points are drawn on the device generators INSIDE the graph (registered with it), as in Step.
    The host is out of the step: ~0.1 ms of Python per iteration instead of ~4 ms of launches."""
    HEAD = 16

    def __init__(self, step):
        from nvfi_amd.models import DeviceTime
        self.DeviceTime = DeviceTime
        self.s = s = step
        self.n_t = sum(1 for g in s.opt.param_groups for p in g["params"] if p.grad is not None)
        self.n_hyper = 1 + self.n_t
        self.renders = 2 if s.workload == "cfg3" else 1
        self.off_jit = (self.HEAD + self.n_hyper + 63) // 64 * 64
        self.total = self.off_jit + self.renders * s.n_rays
        self.rec = torch.zeros(self.total, device=s.dev)
        self.ring = [[torch.zeros(self.total).pin_memory(), None] for _ in range(4)]
        self.k = 0
        self.graph = None
        self.flags = []          # counters tensors of the captured renders (entry 7: device-time plan mismatch)

    def host_record(self):
        """one iteration's host-side draws and decays, in Step._step_streams' order -> pinned record -> device (asynchronous)"""
        s = self.s
        slot = self.ring[self.k % len(self.ring)]
        self.k += 1
        if slot[1] is not None:
            slot[1].synchronize()       # the upload that last read this pinned buffer has executed
        buf = slot[0]
        if s.workload == "cfg3":
            i = int(s.rng.integers(0, 46))
            while i % 3 == 0:
                i = int(s.rng.integers(0, 46))
            buf[0] = float(np.float32(i / 60.0))
            buf[1] = float(np.float32(3 * int(s.rng.integers(0, 16)) / 60.0))
            s.vw *= s.lr_factor
        else:
            buf[1] = float(np.float32(float(s.rng.integers(0, 46)) / 60.0))
        s.L1w *= s.lr_factor; s.tvd *= s.lr_factor; s.tva *= s.lr_factor
        buf[2], buf[3], buf[4], buf[5] = s.L1w, s.tvd, s.tva, s.vw
        h = s.opt.next_hyper()
        buf[self.HEAD:self.HEAD + self.n_hyper] = torch.tensor(h, dtype=torch.float32)
        for g in s.opt.param_groups:
            g["lr"] = g["lr"] * s.lr_factor
        # the per-ray jitter of the renders: the reference's CPU-generator draw (tensorf_base.py:302-306), non-keyframe render first
        for k in range(self.renders):
            torch.rand(s.n_rays, 1, out=buf[self.off_jit + k * s.n_rays: self.off_jit + (k + 1) * s.n_rays].view(s.n_rays, 1))
        s.draw_it += 1
        buf[8:10].view(torch.int64)[0] = s.draw_it         # the iteration counter of nvfi_draw_batch (rec[8:10] as one int64)
        self.rec.copy_(buf, non_blocking=True)
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record()

    def body(self):
        """the device side of one iteration (Step.__call__'s launches); every per-iteration scalar is a view of self.rec"""
        s, rec, DT = self.s, self.rec, self.DeviceTime
        m, f = s.m, s.m.nvfi
        R = s.n_rays
        jit = [rec[self.off_jit + k * R: self.off_jit + (k + 1) * R] for k in range(self.renders)]
        main = torch.cuda.current_stream()
        if getattr(s, "draw_desc", None) is None:
            s._draw_setup()
        s.draw(it_dev=rec[8:10].view(torch.int64))
        f.repack_frags()
        cfg3 = s.workload == "cfg3"
        if cfg3:
            s_pde, s_r1 = s.streams if s.streams is not None else (main, main)
            start = torch.cuda.Event(); start.record(main)
            with torch.cuda.stream(s_pde):
                s_pde.wait_event(start)
                m.vel_loss_weight = rec[5:6]
                s.last_lv = m.get_vel_loss(s.n_pts, points=s.d_pts, t=s.d_t)
            with torch.cuda.stream(s_r1):
                s_r1.wait_event(start)
                f.render_mse_backward_(DT(19.0 / 60.0, rec[0:1]), s.d_ro[0], s.d_rd[0], s.d_tg[0], white_bg=True, jitter=jit[0])      # plan: a non-keyframe time (1 RK2 step)
                self.flags.append(f.last_counters)
        early_regs = not cfg3 and os.environ.get("NVFI_EARLY_REGS", "1") != "0"      # single chain: the regulariser pass beside the forward (Step.__call__)
        if early_regs:
            if getattr(s, "_s_reg", None) is None:
                s._s_reg = torch.cuda.Stream(device=s.dev)
            s._s_reg.wait_stream(main)
            with torch.cuda.stream(s._s_reg):
                s.last_regs = f.regularizers_backward_(rec[2:5])
        # plan: a keyframe time / any time without a velocity field
        loss, _ = f.render_mse_backward_(DT(0.05 if cfg3 else 0.3, rec[1:2]), s.d_ro[-1], s.d_rd[-1], s.d_tg[-1], white_bg=True, jitter=jit[-1],
                                         wait_before_backward=s._s_reg if early_regs else None)
        self.flags.append(f.last_counters)
        if cfg3:
            main.wait_stream(s_pde); main.wait_stream(s_r1)
        if not early_regs:
            s.last_regs = f.regularizers_backward_(rec[2:5])
        s.opt.step(zero_grad=True, hyper_dev=rec[self.HEAD:self.HEAD + self.n_hyper])
        self.loss = loss

    # ---- several ranks: three graphs around the eager exchange (round 5, verdict item 6) ------------------------------------------------
    # G_H: draw + fragment repack | G_P: the PDE term into the staging buffer (replayed on the PDE stream) | G_R: both renders + regularisers
    # (replayed on the main stream, render 1 on its own branch).  Between the replays the host issues what cannot be captured with every
    # backend - the head all-reduce under the PDE chain, the kept-count exchange + commit of the staged PDE gradients, the tail all-reduce -
    # and the one-launch Adam.  Per iteration the host makes 3 graph launches + the exchange instead of ~55 kernel launches.
    def body_head(self):
        s, rec = self.s, self.rec
        if getattr(s, "draw_desc", None) is None:
            s._draw_setup()
        s.draw(it_dev=rec[8:10].view(torch.int64))
        s.m.nvfi.repack_frags()

    def body_pde(self):
        s, rec = self.s, self.rec
        m = s.m
        s.pde_stage.zero()
        m.vel_grad_targets = s.pde_stage.views
        m.vel_loss_weight = rec[5:6]
        s.last_lv = m.get_vel_loss(s.n_pts, points=s.d_pts, t=s.d_t)
        self.pde_out = m.nvfi.last_pde_out

    def body_renders(self):
        s, rec, DT = self.s, self.rec, self.DeviceTime
        f = s.m.nvfi
        R = s.n_rays
        jit = [rec[self.off_jit + k * R: self.off_jit + (k + 1) * R] for k in range(self.renders)]
        main = torch.cuda.current_stream()
        s_r1 = s.streams[1] if s.streams is not None else main
        if s_r1 is not main:
            start = torch.cuda.Event(); start.record(main)
            with torch.cuda.stream(s_r1):
                s_r1.wait_event(start)
                f.render_mse_backward_(DT(19.0 / 60.0, rec[0:1]), s.d_ro[0], s.d_rd[0], s.d_tg[0], white_bg=True, jitter=jit[0])
                self.flags.append(f.last_counters)
        else:
            f.render_mse_backward_(DT(19.0 / 60.0, rec[0:1]), s.d_ro[0], s.d_rd[0], s.d_tg[0], white_bg=True, jitter=jit[0])
            self.flags.append(f.last_counters)
        loss, _ = f.render_mse_backward_(DT(0.05, rec[1:2]), s.d_ro[-1], s.d_rd[-1], s.d_tg[-1], white_bg=True, jitter=jit[-1])
        self.flags.append(f.last_counters)
        if s_r1 is not main:
            main.wait_stream(s_r1)
        s.last_regs = f.regularizers_backward_(rec[2:5])
        self.loss = loss

    def _multi_iteration(self, run_h, run_p, run_r):
        """one iteration of several ranks: run_* either replay the graphs or call the bodies (the eager pass before the capture)"""
        s, rec = self.s, self.rec
        main = torch.cuda.current_stream()
        s_pde = s.streams[0] if s.streams is not None else None
        run_h()
        if s_pde is not None:
            ev = torch.cuda.Event(); ev.record(main)
            with torch.cuda.stream(s_pde):
                s_pde.wait_event(ev)
                run_p()
            run_r()
        else:
            run_r()
        h = s.bucket.all_reduce_head_start(s.tail_off)      # 38 MB of plane / render-MLP gradients: underneath the PDE chain
        if s_pde is not None:
            main.wait_stream(s_pde)
        else:
            run_p()
        s.pde_stage.commit_device(self.pde_out)
        s.bucket.all_reduce_finish(h, s.tail_off)
        s.opt.step(zero_grad=True, hyper_dev=rec[self.HEAD:self.HEAD + self.n_hyper])
        s.stepped = True

    def _capture_multi(self):
        s = self.s
        if s.workload != "cfg3" or s.tail_off is None or s.comm is not None or not s.fused_zero:
            raise SystemExit("--graph on with several ranks: the bat loop with the split torch.distributed exchange and the one-launch Adam only")
        if not s.stepped:
            s.bucket.zero()
        self.host_record()
        torch.cuda.synchronize()
        self._multi_iteration(self.body_head, self.body_pde, self.body_renders)      # eager once: allocator pools, plans
        torch.cuda.synchronize()
        self.flags.clear()
        self.host_record()
        torch.cuda.synchronize()
        self.gH, self.gP, self.gR = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        for g, body in ((self.gH, self.body_head), (self.gP, self.body_pde), (self.gR, self.body_renders)):
            cap = torch.cuda.Stream(device=s.dev)
            cap.wait_stream(torch.cuda.current_stream())
            # G_P and G_R read the fragment cache that G_H's pack node fills: every iteration replays G_H first (_multi_iteration)
            s.m.nvfi.frags_packed_by_earlier_graph(g is not self.gH)
            try:
                with torch.cuda.graph(g, stream=cap):      # (each graph its own memory pool: G_P and G_R replay side by side)
                    body()
            finally:
                s.m.nvfi.frags_packed_by_earlier_graph(False)
            torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        self.graph = self.gR
        self.graphs = [self.gR]
        self.multi = True
        self.turn = 0
        # the captured iteration itself (its record was uploaded above; the captures executed nothing)
        self._multi_iteration(self.gH.replay, self.gP.replay, self.gR.replay)
        torch.cuda.synchronize()

    def _capture_once(self):
        s = self.s
        if s.world > 1:
            return self._capture_multi()
        self.host_record()
        torch.cuda.synchronize()
        cap = torch.cuda.Stream(device=s.dev)
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            self.body()                 # eager once on the capture stream: first-touch of this stream's allocator pools
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        self.flags.clear()
        self.host_record()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            self.body()
        self.graph = g
        self.graphs = [g]
        self.turn = 0
        g.replay()                      # the captured iteration itself (its record was uploaded above)
        torch.cuda.synchronize()

    def capture(self):
        """Capture the iteration and time a few replays (`graph_capture_attempts_ms` in the line).  NVFI_GRAPH_TRIES > 1 repeats the capture on
        fresh stream objects and keeps the fastest - an experiment that ruled the capture OUT as the cause of the bimodal replay times seen
        in round 3 (every attempt replays at 5.35-5.5 ms); the cause was the depth of the host's run-ahead, see __call__.  Setup, before the W
        warm-up steps; the timed region replays one fixed graph."""
        s = self.s
        tries = 1 if s.world > 1 else max(1, int(os.environ.get("NVFI_GRAPH_TRIES", "1")))
        best = None
        self.capture_ms = []
        for k in range(tries):
            if k > 0:       # fresh stream objects for the branches and for the launch
                if s.streams is not None:
                    s.streams = [torch.cuda.Stream(device=s.dev) for _ in s.streams]
                self._gstream = None
                self.flags = []
            self._capture_once()
            for _ in range(2):
                self()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                self()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 4 * 1e3
            self.capture_ms.append(ms)
            if best is None or ms < best[0]:
                best = (ms, self.graph, self.graphs, s.streams, self._gstream, self.flags)
        _, self.graph, self.graphs, s.streams, self._gstream, self.flags = best
        self.turn = 0

    def __call__(self):
        # the replay is launched on the current (default) stream; NVFI_GRAPH_STREAM=pool|high: a dedicated pool / high-priority stream
        # (experiments: with the paced run-ahead below the launch stream makes no difference for the three-branch graph - 5.37-5.44 ms - and the
        # per-iteration cross-stream events of a dedicated stream cost the short radiance-only step 8 %)
        gs = getattr(self, "_gstream", None)
        if gs is None:
            kind = os.environ.get("NVFI_GRAPH_STREAM", "default")
            gs = self._gstream = (torch.cuda.Stream(device=self.s.dev, priority=-1 if kind == "high" else 0) if kind != "default" else False)
        graphs = getattr(self, "graphs", None) or [self.graph]
        g = graphs[getattr(self, "turn", 0) % len(graphs)]
        # Bounded run-ahead (NVFI_GRAPH_DEPTH, default 2 iterations in flight): the host needs ~0.1 ms per iteration, so without pacing it
        # queues the whole run - thousands of packets over the graph's branch queues - at once, and the same graph then replays at 5.5 ms or,
        # in one process out of three, at 6.2-6.5 ms (the one-stream time: the branches stop overlapping).  Measured, 5 processes each:
        # depth 2: 5.34-5.37 ms; depth 3: 5.38; depth 1: 5.48-5.53 (the device waits for the next launch); unbounded: 5.48-6.5.
        depth = self.__dict__.setdefault("_depth", int(os.environ.get("NVFI_GRAPH_DEPTH", "2")))
        if depth > 0:
            q = self.__dict__.setdefault("_inflight", [])
            if len(q) >= depth:
                q.pop(0).synchronize()
        if getattr(self, "multi", False):
            self.host_record()
            self._multi_iteration(self.gH.replay, self.gP.replay, self.gR.replay)
        elif gs:
            gs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(gs):
                self.host_record()
                g.replay()
            torch.cuda.current_stream().wait_stream(gs)
        else:
            self.host_record()
            g.replay()
        self.turn = getattr(self, "turn", 0) + 1
        if depth > 0:
            e = torch.cuda.Event()
            e.record()
            self._inflight.append(e)
        return self.loss

    def check(self):
        """after the run: did any replayed render see a time outside its captured plan class?"""
        bad = sum(int(c[7].item()) for c in self.flags)
        if bad:
            raise SystemExit("GraphedStep: a device-side frame time did not match the captured RK2 plan (counters[7])")


class DropinStep:
    """The reference's loop body (train_nvfi.py:139-249, --static_dynamic) on the `models` alias package, unchanged in structure:
    plain autograd through `renderer.render`, `nvfi.get_vel_loss` + `if loss_vel > 0`, the torch-op regularisers, torch.optim.Adam with
    the reference's groups and betas, `optimizer.zero_grad(set_to_none=True)`, the per-iteration `.item()` calls and lr decay.
    What is NOT mirrored is the dataset: the posed image is the synthetic camera, and the pixel batch is drawn on the device
    (`Camera.sample_rays_device`, f-2) unless `host_rays` asks for the reference's per-iteration `Camera(...).sample_rays` on the host."""

    def __init__(self, model, device, n_rays, n_pts, world, rank, workload, live=False, host_rays=False):
        import models                      # the drop-in alias of the reference's package
        self.models = models
        self.m, self.dev, self.n_rays, self.n_pts, self.world, self.workload = model, device, n_rays, n_pts, world, workload
        self.ren = models.Renderer(model, 0, 0, n_rays)
        from nvfi_amd.utils import TVLoss
        self.tvreg = TVLoss()
        self.groups = model.get_optparam_groups(0.02, 1e-3, 1e-3)
        self.live = live
        # NVFI_DROPIN_FUSED_ADAM=1 (`tools/run_reference_driver.py --fused-adam`): the caller's optimiser built with fused=True - one kwarg away from
        # train_nvfi.py:95; the default stays the reference's call as written (torch's multi-tensor implementation)
        akw = dict(fused=True) if os.environ.get("NVFI_DROPIN_FUSED_ADAM") == "1" else {}
        self.fused_adam = bool(akw)
        if live:
            self.opt = torch.optim.Adam(self.groups, betas=(0.9, 0.99), **akw)
        else:
            sg, self.pairs = _shadow_groups([dict(params=list(g["params"]), lr=g["lr"]) for g in self.groups])
            self.opt = torch.optim.Adam(sg, betas=(0.9, 0.99), **akw)
        self.host_rays = host_rays
        self.pose = pose_spherical(30.0, -30.0, 4.0).to("cpu" if host_rays else device)
        self.focal = 0.5 * W_IMG / np.tan(0.5 * ANGLE_X)
        self.target_img = torch.rand(H_IMG, W_IMG, 3, device="cpu" if host_rays else device)
        self.gen = torch.Generator(device=device); self.gen.manual_seed(233 + rank)
        self.rng = np.random.default_rng(233 + rank)
        np.random.seed(233 + rank)
        self.lr_factor = 0.1 ** (1 / 30000)
        self.L1w, self.tvd, self.tva, self.vw = 8e-4, 1.0, 1.0, 1.0
        self.counters, self.pde_counters = [], []
        # the opt-in a plain autograd driver can switch on without touching its loop: gradients accumulated in place into the field's own flat buffer
        # (NVFI_INPLACE_GRADS=0: the library default, pure autograd; NVFI_AUTO_OVERLAP=1: train-mode calls on the field's side streams - measured slower)
        model.nvfi.accumulate_grads_inplace = "arena" if os.environ.get("NVFI_INPLACE_GRADS", "1") != "0" else False
        model.nvfi.auto_overlap = os.environ.get("NVFI_AUTO_OVERLAP", "0") == "1"
        model.vel_loss_weight = None

    def batch(self):
        cam = self.models.Camera(self.pose, H_IMG, W_IMG, self.focal, self.target_img, 1.0, 8.0)
        if self.host_rays:
            rays, target = cam.sample_rays(self.n_rays)
            return rays.to(self.dev), target.to(self.dev)
        rays, ids = cam.sample_rays_device(self.n_rays, generator=self.gen)
        return rays, self.target_img.reshape(-1, 3)[ids]

    def __call__(self):
        nvfi, f = self.m, self.m.nvfi
        # NVFI_DROPIN_TIMING=1: host timestamps of the iteration's stations (tools/hostprof.py dropin prints their means)
        tm = self.__dict__.setdefault("_tm", [] if os.environ.get("NVFI_DROPIN_TIMING") == "1" else None)
        mark = (lambda: row.append(time.perf_counter())) if tm is not None else (lambda: None)
        row = []
        if tm is not None:
            tm.append(row)
        mark()
        nvfi.train(); self.ren.train()
        loss = 0
        if self.workload == "cfg3":
            i = int(self.rng.integers(0, 46))
            while i % 3 == 0:
                i = int(self.rng.integers(0, 46))
            rays, target = self.batch()
            rgb_map = self.ren.render(i / 60.0, rays, white_background=True, mode="train")[0]
            rgb_loss = torch.nn.functional.mse_loss(rgb_map[..., :3], target[..., :3])
            loss = rgb_loss
            mark()
            rgb_loss_t = rgb_loss.item()                                 # train_nvfi.py:161
            mark()
            self.counters.append(f.last_counters)
            t_key = 3 * int(self.rng.integers(0, 16)) / 60.0
        else:
            t_key = float(self.rng.integers(0, 46)) / 60.0
        rays, target = self.batch()
        rgb_map = self.ren.render(t_key, rays, white_background=True, mode="train")[0]
        rgb_loss0 = torch.nn.functional.mse_loss(rgb_map[..., :3], target[..., :3])
        loss = loss + 1.0 * rgb_loss0
        self.counters.append(f.last_counters)
        self.L1w *= self.lr_factor
        loss = loss + self.L1w * f.density_L1()
        self.tvd *= self.lr_factor
        loss = loss + f.TV_loss_density(self.tvreg) * self.tvd
        self.tva *= self.lr_factor
        loss = loss + f.TV_loss_app(self.tvreg) * self.tva
        mark()
        if self.workload == "cfg3":
            self.vw *= self.lr_factor
            loss_vel = nvfi.get_vel_loss(self.n_pts)
            mark()
            self.pde_counters.append(f.last_pde_counters)
            if loss_vel > 0:                                             # train_nvfi.py:233 (a device sync on a tensor)
                loss = loss + self.vw * loss_vel
        mark()
        self.opt.zero_grad(set_to_none=True)
        if not self.live:
            for p, s in self.pairs:
                p.grad = None
        loss.backward()
        mark()
        if not self.live:
            for p, s in self.pairs:
                s.grad = p.grad
        self.opt.step()
        if not self.live:
            f.invalidate_frags()     # stationary mode: the optimiser moved a shadow copy - the next iteration repacks the fragment cache as a real step's would
        for g in self.opt.param_groups:
            g["lr"] = g["lr"] * self.lr_factor
        mark()
        self.last_psnr = rgb_loss0.item()                                # train_nvfi.py:252
        mark()
        return loss


class SegmStep:
    """One iteration of the reference's segmentation training (train_segm.py:126-198) on the hot path: jittered 64^3 lattice -> density at
    t = 0 -> occupied points -> integrate_pos to a random time in [min_t, tmax] (up to 30 RK2 steps per point) -> MaskField(3 -> 128 x 4 -> K,
    softmax) forward + backward -> Adam.  The reference's rigid-fit / kNN-smoothness / entropy losses are PyTorch code that SURVEY section 8 leaves
    out of scope; a stand-in of the same arity (a quadratic on the mask + the entropy term, test_maskfield.py's convergence test) drives the
    backward, so the timed work is exactly rows a-6 / a-9 / a-19: point pipeline + MaskField training kernels."""

    def __init__(self, model, device, n_sample_res=64, n_object=8, fp16=False):
        from nvfi_amd.models import MaskField
        self.f = model.nvfi
        self.f.eval()
        self.res = n_sample_res
        torch.manual_seed(233)
        self.mf = MaskField(n_layer=4, n_dim=128, input_dim=3, skips=[], mask_dim=n_object).to(device)
        self.mf.mfma_fp16 = bool(fp16)
        self.opt = torch.optim.Adam(self.mf.parameters(), lr=0.005, betas=(0.9, 0.999), fused=True)
        self.points, self.steps_rk = 0, 0
        self.time_advect, self.advect_events, self.advect_evals = False, [], 0

    def __call__(self):
        from nvfi_amd.utils.segm_utils import segm_points
        f = self.f
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if self.time_advect else None
        xyz, flow, t = segm_points(f, self.res, min_t=0.5, alpha_scale=10.0, advect_events=ev)
        if ev is not None:
            self.advect_events.append(ev)
            self.advect_evals += int(xyz.shape[0]) * int(np.ceil(t / (0.5 * f.tmax / (f.num_keyframes - 1)) - 1e-6))
        mask = self.mf(xyz)
        target = (flow / (flow.norm(dim=1, keepdim=True) + 1e-9))[:, :1]
        loss = ((mask[:, :1] - target) ** 2).mean() - 1e-3 * (mask * torch.log(mask + 1e-8)).sum(1).mean()
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.opt.step()
        self.points += int(xyz.shape[0])
        dt_max = 0.5 * f.tmax / (f.num_keyframes - 1)
        self.steps_rk += int(xyz.shape[0]) * int(np.ceil(t / dt_max - 1e-6))
        return loss


def segm_main(args, device):
    """`--workload segm`: BASELINE configs[4]'s single-GPU share - train_segm.py's MaskField step on the fan scene (fan.yaml =
    the bat box, K = 16, with bat.yaml:142-154's segmentation block: n_object 8, n_sample_res 64, min_t 0.5, alpha_scale 10); prints its own line.
    NVFI_SEGM_SCENE=chessboard: the round-4/5 stand-in scene (K = 4).  NVFI_MASK_FP16=1 / NVFI_VEL_FP16=1|2 select the fp16-input MFMA modes
    (stated in `dtype`)."""
    scene = os.environ.get("NVFI_SEGM_SCENE", "fan")
    # train_segm.py works on the TRAINED field's t = 0 volume: the final 199^3 grid of the schedule (bat.yaml:95)
    model = build_scene_chessboard(device, final=False) if scene == "chessboard" else build_scene(device, 199, 128, True)
    fp16 = os.environ.get("NVFI_MASK_FP16", "0") == "1"
    step = SegmStep(model, device, 64, 8, fp16)
    for _ in range(max(args.prime, 3)):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    step.points = step.steps_rk = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pts, evals = step.points / args.steps, 2.0 * step.steps_rk / args.steps
    # the dominant kernel - integrate_pos' RK2 advection (k_rk2_x6w since round 6) - bracketed by HIP events on its own stream in a second pass
    step.time_advect = True
    step.advect_events, step.advect_evals = [], 0
    for _ in range(min(args.steps, 5)):
        step()
    torch.cuda.synchronize()
    adv_ms = sum(a.elapsed_time(b) for a, b in step.advect_events)
    adv_tf = 2.0 * step.advect_evals * VEL_FLOP / (adv_ms * 1e-3) / 1e12 if adv_ms > 0 else 0.0
    step.time_advect = False
    mf_flop = pts * 3 * 2 * (3 * 128 + 3 * 128 * 128 + 128 * 8)          # forward + dgrad + wgrad GEMMs
    flop = evals * VEL_FLOP + mf_flop
    vel16 = model.nvfi.vel_fp16
    f = model.nvfi
    dt_max = 0.5 * f.tmax / (f.num_keyframes - 1)
    x6 = not vel16 and os.environ.get("NVFI_INTEGRATE_X6", "1") != "0"
    peak = PEAK_BF16_MFMA / 6.0 if x6 else PEAK_FP32_MFMA
    out = {"metric": "MaskField training points/sec (train_segm.py step: density lattice + RK2 advection + MaskField fwd/bwd + Adam)",
           "value": pts * args.steps / dt, "unit": "points/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" + (" (MaskField: fp16-input MFMA, fp32 accumulation)" if fp16 else "") + (f" (integrate_pos: vel_fp16={vel16})" if vel16 else ""),
           "data": "synthetic",
           "config": {"workload": (f"train_segm.py:126-198 on the {'fan (= bat box, bat.yaml:142-154 segmentation block)' if scene != 'chessboard' else 'chessboard-like'} field "
                                   f"(configs[4], one GPU's share): 64^3 jittered lattice at t = 0, occupied points advected to t in [0.5, {f.tmax}] "
                                   f"(K = {f.num_keyframes}: {int(np.ceil(0.5 / dt_max))}-{int(np.ceil(f.tmax / dt_max))} RK2 steps), "
                                   "MaskField 3 -> 128 x 4 -> 8 softmax, stand-in loss, torch Adam(fused)"),
                      "scene": scene, "grid": [int(g) for g in f.gridSize.tolist()], "K": int(f.num_keyframes),
                      "points_per_step": pts, "velocity_net_evaluations_per_step": evals},
           "roofline": {"bound": "mfma", "kernel": ("k_rk2_x6w (integrate_pos, per-point schedule)" if x6 else "integrate_pos (fp32 MFMA / fp16-input kernel)"),
                        "achieved": adv_tf, "peak": peak, "unit": "TFLOP/s", "frac": adv_tf / peak, "traffic": None,
                        "peak_note": ("2500 / 6: an x6 kernel issues six bf16 MFMAs per fp32 product" if x6 else "dense fp32 MFMA"),
                        "ms_per_launch": adv_ms / max(1, len(step.advect_events)),
                        "whole_step": {"achieved": flop / (dt / args.steps) / 1e12, "unit": "TFLOP/s (algorithmic fp32 FLOPs of integrate_pos + MaskField GEMMs / step time)"},
                        "gflop_per_step": {"integrate_pos": evals * VEL_FLOP / 1e9, "maskfield": mf_flop / 1e9}}}
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = segm_cpu_baseline(model, step, "chessboard" if scene == "chessboard" else "bat")
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out))


def _oracle_field(model, scene):
    """the oracle's view of the field (FieldSpec) from the product module's state_dict"""
    from oracle import oracle as orc
    f = model.nvfi
    sd = {k[len("nvfi."):]: v.detach().cpu().contiguous().numpy() for k, v in model.state_dict().items() if not k.startswith("nvfi.vel.vel_net.")}
    meta = dict(aabb=f.aabb.cpu().numpy(), gridSize=np.array(f.gridSize.tolist()), num_keyframes=f.num_keyframes, tmax=f.tmax,
                near=f.near_far[0], far=f.near_far[1], step_ratio=f.step_ratio, max_n_samples=f.max_n_samples,
                density_shift=f.density_shift, distance_scale=f.distance_scale, alphaMask_thres=f.alphaMask_thres,
                rayMarch_weight_thres=f.rayMarch_weight_thres, stepSize=f._step_host, nSamples=f.nSamples, use_sur=0, eps=0.03)
    if scene == "chessboard":
        meta.update(use_sur=1, sur_bounds=f.vel.bounds.detach().float().cpu().numpy().reshape(2, 3))
        del meta["eps"]
    return orc.FieldSpec(sd, meta)


def segm_cpu_baseline(model, step, scene="bat"):
    """The WHOLE step of train_segm.py:127-198 on the oracle (plain C, OpenMP), like for like with the GPU line: density of the 64^3 jittered lattice at
    t = 0 -> occupied points -> integrate_pos to a time in [0.5, tmax] (the RK2 advection is most of the step's FLOPs) -> MaskField forward +
    parameter gradients.  One warm-up step, then whole steps for ~10 s."""
    from oracle import oracle as orc
    from nvfi_amd.utils.segm_utils import sample_volume_points
    threads = min(64, os.cpu_count() or 1)
    orc.set_threads(threads)
    f = model.nvfi
    fs = _oracle_field(model, scene)
    ab = f._aabb_host
    bounds = [[ab[0], ab[3]], [ab[1], ab[4]], [ab[2], ab[5]]]
    lo, hi = np.array(ab[:3], np.float32), np.array(ab[3:], np.float32)
    ps = [p.detach().cpu().numpy() for p in step.mf.parameters()]
    rng = np.random.default_rng(3)
    res = step.res

    def one():
        xyz_w = sample_volume_points(bounds, res, perturb=True).reshape(-1, 3).numpy()
        xyz = ((xyz_w - lo) * (2.0 / (hi - lo)) - 1.0).astype(np.float32)
        feat = orc.density_feature(fs, np.concatenate([xyz, np.full((xyz.shape[0], 1), -1.0, np.float32)], 1))      # normalised time of t = 0
        sigma = orc.feature2density(fs, feat[:, 0])
        alpha = 1.0 - np.exp(-sigma * 0.01)
        occ = xyz[alpha > f.alphaMask_thres * 10.0]
        t = float(0.5 + (f.tmax - 0.5) * rng.uniform())
        n = occ.shape[0]
        orc.integrate_pos(fs, occ, np.zeros(n, np.float32), np.full(n, t, np.float32))
        g = rng.standard_normal((n, 8)).astype(np.float32)
        orc.maskfield(ps, occ, g)
        return n

    one()
    t0 = time.perf_counter(); reps = 0; pts = 0
    while time.perf_counter() - t0 < 15 and reps < 8:
        pts += one(); reps += 1
    dt = (time.perf_counter() - t0) / reps
    return dict(value=pts / reps / dt, unit="points/s", cores=threads, kind="port", s_per_step=dt, points_per_step=pts / reps,
                sample=f"oracle/nvfi_oracle.c (OpenMP x{threads}), the whole step: orc_density_feature on the {res}^3 lattice + orc_integrate_pos of the occupied points "
                       f"(same time range as the GPU line) + orc_maskfield forward and parameter gradients; {reps} whole steps")


def cpu_baseline(model, workload, seconds_hint=20, scene="bat"):
    """Oracle (C restatement, OpenMP) on a bounded sample of the same step: 256+256 rays, P=32768 (chessboard at 688 samples per ray: 64+64 rays, P=16384)."""
    from oracle import oracle as orc
    sc = SCENES[scene]
    fs = _oracle_field(model, scene)
    # 64 threads is the oracle's sweet spot on the GPU box's 256-thread host (measured: 644 rays/s at 64, 300 at 256 on this sample)
    threads = min(64, os.cpu_count() or 1)
    orc.set_threads(threads)
    o, d = camera_bundle("cpu") if scene == "bat" else camera_bundle_chessboard("cpu")
    rng = np.random.default_rng(5)
    R, P = (256, 32768) if scene == "bat" else (64, 16384)
    t_non, t_key = (19 / 60.0, 0.30) if scene == "bat" else (22 / 60.0, 0.25)
    def one_render(t):
        idx = rng.integers(0, o.shape[0], R)
        oo, dd = o[idx].numpy(), d[idx].numpy()
        u = rng.uniform(0, 1, (R, 1)).astype(np.float32)
        r = orc.render(fs, oo, dd, t, u=u, train=True, white_bg=sc["white_bg"], keep_ctx=True, use_vel=(workload == "cfg3"))
        tg = rng.uniform(0, 1, (R, 3)).astype(np.float32)
        r.backward(fs, g_rgb=2 * (r.rgb - tg) / (R * 3))
        r.free()
    def step():
        if workload == "cfg3":
            one_render(t_non)
        one_render(t_key)
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
                sample=f"oracle/nvfi_oracle.c (OpenMP x{threads}) on 1/{2048 // R} of the step: {R} rays per render"
                       + (f" x2 renders + PDE with P={P}" if workload == "cfg3" else "") + f", fwd+bwd, {n} reps")


def _load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def _spawn_ranks(n, ndev, backend):
    """Re-run this command as n ranks under torch.distributed.run (one process per GPU).  Returns the launcher's exit code."""
    import socket
    import subprocess
    if backend == "nccl" and ndev < n:
        print(f"bench.py: --gpus {n} needs {n} GPUs on this node, found {ndev} (one RCCL rank per GPU)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg2", "chessboard", "segm"],
                    help="cfg3: BASELINE configs[2] (headline); cfg2: configs[1] radiance-only; chessboard: configs[3]'s single-GPU share (the cfg3 loop on the "
                         "InDoorSeg box at its final 199x199x200 grid: K = 4, surround-box gate, no white background, 688 samples/ray, P = 131072); "
                         "segm: configs[4]'s single-GPU share (train_segm.py's MaskField step)")
    ap.add_argument("--chessboard-init-res", action="store_true", help="chessboard at the initial 64^3 grid (219 samples/ray) instead of the final one")
    ap.add_argument("--mode", default="fused", choices=["fused", "dropin"])
    ap.add_argument("--live", action="store_true", help="apply the optimiser updates to the field itself (non-stationary workload)")
    ap.add_argument("--host-rays", action="store_true", help="dropin mode: rebuild the 800x800 Camera and np.random.choice the batch on the host, as the reference does")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="strong: the GLOBAL batch (rays, collocation points) is fixed and split over the ranks")
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--pts", type=int, default=None, help="PDE collocation points per step (default: the scene's vel_reg_n_pts: 262144 bat, 131072 chessboard)")
    ap.add_argument("--grid", type=int, default=199)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--profile-steps", type=int, default=-1, help="steps of the separate profiled pass (default: min(steps, 5); 0: none)")
    ap.add_argument("--prime", type=int, default=8, help="setup iterations before the W warm-up steps (first-touch of the workspaces, allocator pools of the three streams, clocks): the first ~5 iterations of a process run 5-10 %% slower than steady state")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short extra runs whose results are attached to the default line under `extras` "
                    "(the reference's loop body verbatim, the radiance-only workload, the opt-in prefilter)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the whole iteration as one captured hipGraph with a device-side time schedule (auto: on for the single-GPU fused driver)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the NVFi hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("NVFI_BENCH_BACKEND", "nccl")       # "nccl" = RCCL over xGMI; "gloo" only to exercise the path on one GPU
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL rendezvous on 127.0.0.1) - the same
        # command line the driver uses with torch.distributed.run in front
        sys.exit(_spawn_ranks(args.gpus, ndev, backend))
    graph_line = None
    if args.workload in ("chessboard", "segm"):
        args.graph = "off"           # (the captured iteration is the bat loop's)
    if (args.graph == "auto" and args.mode == "fused" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not os.environ.get("NVFI_TORCH_ADAM")
            and not os.environ.get("NVFI_BENCH_CHILD")):
        # auto: measure BOTH launch modes of the same step and report the faster one (both values go into the line).  The hipGraph replay
        # takes the host out of the iteration - its time does not depend on the host's launch rate - while eager launches on three streams
        # are ~3 % faster when the host keeps up (rocprof: the graph's branches start a little later).  The captured-graph run happens in
        # a child process: a fault inside a graph replay (a runtime bug, see above) kills the process and cannot be caught.
        import subprocess
        child_args = [a for a in sys.argv[1:] if a != "--no-cpu-baseline"] + ["--graph", "on", "--no-cpu-baseline"]
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + child_args, env=dict(os.environ, NVFI_BENCH_CHILD="1"),
                               stdout=subprocess.PIPE, text=True, timeout=900)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                graph_line = json.loads(lines[-1])
            else:
                print(f"bench.py: the hipGraph run failed (exit code {r.returncode}); measuring the eager step only", file=sys.stderr)
        except subprocess.TimeoutExpired:
            print("bench.py: the hipGraph run did not finish in 900 s; measuring the eager step only", file=sys.stderr)
        args.graph = "off"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # round 6: several ranks under STRONG scaling run a launch-bound shard (~22 class launches for ~1 ms of device time): `auto` takes the captured
    # three-graph iteration there (GraphedStep._multi_iteration; weak scaling - the SCALE run's default - stays on eager launches)
    if (args.graph == "auto" and world > 1 and args.scaling == "strong" and args.mode == "fused" and args.workload == "cfg3"
            and os.environ.get("NVFI_ALLREDUCE", "torch") != "abi" and not os.environ.get("NVFI_TORCH_ADAM")):
        args.graph = "on"
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} RCCL ranks need {world} GPUs, this node shows {ndev} (one process per GPU; two ranks cannot share "
                         "a device in one RCCL communicator)")
    dev_index = local_rank % ndev          # (ranks share a device only in the single-GPU logic test: NVFI_BENCH_BACKEND=gloo)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm_info = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        # what the process group really is: ranks, backend and the distinct devices behind them (reported in the line)
        alld = [None] * dist.get_world_size()
        dist.all_gather_object(alld, dev_index)
        comm_info = dict(backend=dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""), ranks=dist.get_world_size(),
                         devices=sorted({int(x) for x in alld}),
                         exchange="nvfi_allreduce_grads (C ABI)" if os.environ.get("NVFI_ALLREDUCE") == "abi" else "torch.distributed.all_reduce")
        world = dist.get_world_size()
    from nvfi_amd import _lib
    L = _lib.lib()
    scene = "bat"
    if args.workload == "segm":
        if world > 1 or args.mode != "fused":
            raise SystemExit("--workload segm is a single-process step")
        return segm_main(args, device)
    if args.workload == "chessboard":
        scene, args.workload = "chessboard", "cfg3"          # the same loop on another field
        if args.mode != "fused":
            raise SystemExit("--workload chessboard runs the fused driver")
    if args.pts is None:
        args.pts = SCENES[scene]["pts"]

    n_rays, n_pts = args.rays, args.pts
    if args.scaling == "strong":           # 256 rays and P/8 points per GPU at 8 GPUs
        n_rays, n_pts = max(1, args.rays // world), max(128, args.pts // world)
    model = build_scene(device, args.grid, args.samples, use_vel=True) if scene == "bat" else build_scene_chessboard(device, final=not args.chessboard_init_res)
    if args.workload == "cfg2":
        model.nvfi.use_vel = False
    if args.mode == "dropin":
        if world > 1:
            raise SystemExit("--mode dropin is the single-process reference loop")
        step = DropinStep(model, device, n_rays, n_pts, world, rank, args.workload, live=args.live, host_rays=args.host_rays)
    else:
        step = Step(model, device, n_rays, n_pts, world, rank, args.workload, live=args.live, scene=scene)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.prime):          # setup
        step()
    use_graph = args.mode == "fused" and args.graph == "on"
    run = step
    if use_graph:
        run = GraphedStep(step)      # several ranks: three graphs around the eager exchange (GraphedStep._multi_iteration)
        run.capture()
    for _ in range(args.warmup):         # the contract's W untimed warm-up steps
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if use_graph:
        run.check()
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())

    renders = 2 if args.workload == "cfg3" else 1
    rays_per_step = n_rays * renders * world
    value = rays_per_step * args.steps / dt

    # ---- profiled pass (outside the timed region): HIP events around every launch of a kernel class on the launch stream, and the
    #      device-side work counters of those same steps
    psteps = min(args.steps, 5) if args.profile_steps < 0 else args.profile_steps
    roof, work = None, None
    if psteps > 0:
        step.counters.clear(); step.pde_counters.clear()
        streams, step.streams = getattr(step, "streams", None), None     # serial: an event interval then brackets one kernel class only
        # ... which also means no LIBRARY-side fork (the backward's side streams of the one-stream / drop-in modes: an interval on one stream would
        # include the contention of the kernels on the other - round 5's `wgrad 0.31 ms x 4` in those lines)
        _fld = getattr(getattr(step, "m", None), "nvfi", None)
        _forks = (_fld.fork_backward, _fld.auto_overlap) if _fld is not None else None
        if _fld is not None:
            _fld.fork_backward, _fld.auto_overlap = False, False
        torch.cuda.synchronize()
        L.nvfi_prof_enable(1)
        step()                      # one unrecorded profiled step: the event pool is created here (~0.1 s of hipEventCreate in a fresh process)
        torch.cuda.synchronize()
        step.counters.clear(); step.pde_counters.clear()
        L.nvfi_prof_enable(1)       # recycles the events; (also drops a k_nvfi_prof_marker launch into a rocprofv3 trace: the per-kernel statistics under profiles/ are taken after the LAST marker)
        tp0 = time.perf_counter()
        for _ in range(psteps):
            step()
        torch.cuda.synchronize()
        tprof = (time.perf_counter() - tp0) / psteps
        ncls = L.nvfi_prof_nclasses()
        tot = (C.c_double * ncls)(); cnt = (C.c_int64 * ncls)()
        _lib.check(L.nvfi_prof_collect(tot, cnt))
        L.nvfi_prof_enable(0)
        step.streams = streams
        if _fld is not None:
            _fld.fork_backward, _fld.auto_overlap = _forks
        c = torch.stack(step.counters).sum(0).cpu().numpy() if step.counters else np.zeros(8)
        pc = torch.stack(step.pde_counters).sum(0).cpu().numpy() if step.pde_counters else np.zeros(8)
        V, Nw, M, E = float(c[0]), float(c[1]), float(c[2]), float(c[3])
        kept, pre_evals = float(pc[4]), float(pc[3])
        pre_mode = os.environ.get("NVFI_PDE_PREFILTER", "x6")
        flops = {"rk2_fwd": E * VEL_FLOP, "rk2_bwd": E * VEL_FLOP, "app_fwd": M * APP_FLOP, "app_bwd": M * APP_FLOP,
                 "wgrad": E * VEL_FLOP + M * APP_FLOP + kept * 6 * VEL_FLOP, "pde_fwd": kept * 6 * VEL_FLOP,
                 "pde_bwd": kept * 6 * VEL_FLOP, "pde_prefilter": pre_evals * VEL_FLOP}
        if os.environ.get("NVFI_RK2_FUSE", "1") != "0":
            # vel_fuse.hip: the RK2 adjoint also forms the four 128 x 128 weight gradients of the render warp (2 * 4 * 128^2 FLOP per evaluation);
            # the wgrad class keeps the two edge layers of that launch
            flops["rk2_bwd"] += E * 131072.0
            flops["wgrad"] -= E * 131072.0
        if os.environ.get("NVFI_PDE_FUSE", "1") != "0":
            # pde_fuse.hip: the Jacobian adjoint also forms weight_net's four 128 x 128 weight gradients for the five columns of every kept point
            flops["pde_bwd"] += kept * 5 * 131072.0
            flops["wgrad"] -= kept * 5 * 131072.0
        if pre_mode in ("fp16band", "split16band"):      # opt-in: the class is an fp16-MFMA pass + a short fp32 list; no fp32-MFMA figure applies to it
            del flops["pde_prefilter"]
        times = {CLASSES[i]: (tot[i], cnt[i]) for i in range(min(ncls, len(CLASSES)))}
        # per step: what the kernels counted (V valid samples, N samples warped by RK2, M appearance-masked samples, E velocity-net
        # evaluations of the render warp, P' collocation points kept by the occupancy prefilter, prefilter net evaluations)
        work = dict(steps=psteps, V=V / psteps, N=Nw / psteps, M=M / psteps, E=E / psteps, P_kept=kept / psteps, prefilter_evals=pre_evals / psteps,
                    gflop_per_step={k: v / psteps / 1e9 for k, v in flops.items()}, gflop_per_step_total=sum(flops.values()) / psteps / 1e9,
                    ms_per_step_profiled_serial=tprof * 1e3, prefilter=pre_mode, prefilter_fp32_reevaluated_points=float(pc[5]) / psteps,
                    streams=("3 HIP streams in the timed region (PDE term | non-keyframe render | keyframe render); the profiled pass is issued on one stream"
                             if streams is not None else "one stream"))
        traffic_all = (_load_json(f"{PROFILE_TAG}_traffic.json") or {}).get("bytes_per_launch", {})
        per_class = {}
        for k, (ms, n) in times.items():
            e = dict(ms_per_step=ms / psteps, launches_per_step=n / psteps)
            if k in flops and ms > 0:
                e["tflops"] = flops[k] / (ms * 1e-3) / 1e12
                e["frac"] = e["tflops"] / PEAK_FP32_MFMA
            x6_warp = (k == "rk2_fwd" and os.environ.get("NVFI_RK2_X6", "1") != "0"
                       and os.environ.get("NVFI_VEL_FP16_TRAIN", "0") != "1")
            x6_bwd = (k == "rk2_bwd" and os.environ.get("NVFI_FUSE_X6", "1") != "0" and os.environ.get("NVFI_RK2_FUSE", "1") != "0")                 # round 6: vel_fuse.hip, both roles on x6
            x6_jet = (k == "pde_fwd" and os.environ.get("NVFI_PDE_JET_X6", "1") != "0")      # round 6: pde_jet6.hip
            if ((k == "pde_prefilter" and pre_mode == "x6") or x6_warp or x6_bwd or x6_jet) and "tflops" in e:
                # the x6 kernels run on the 16-bit matrix pipe, six MFMAs per fp32 product: their speed of light in ALGORITHMIC fp32 FLOPs is the dense
                # bfloat16 peak / 6 (416.7 TFLOP/s) - NOT the fp32 MFMA peak, which they may (and do) exceed
                e["peak"] = PEAK_BF16_MFMA / 6.0
                e["frac_of_fp32_mfma_peak"] = e["frac"]
                e["frac"] = e["tflops"] / e["peak"]
                e["matrix_pipe"] = "bf16 (x6: fp32 products from three bfloat16 terms per operand, six v_mfma_f32_32x32x16_bf16 per K step, fp32 accumulation)"
            if k == "pde_prefilter" and pre_mode == "x6" and "tflops" in e:
                # vel_x6.hip: every fp32 product of the four 128 x 128 hidden layers and of the input layer is formed from six bfloat16 term products
                # on the 16-bit matrix pipe (2.5 PFLOP/s dense): `tflops` / `frac` stay the ALGORITHMIC fp32 FLOPs against the fp32 MFMA peak (the figure
                # that compares with the fp32 kernel it replaces); the executed matrix work is priced against the pipe it runs on here
                hid = pre_evals * 2.0 * (32 * 128 + 4 * 128 * 128)      # layer 0 occupies two 16-wide K steps
                e["matrix_pipe"] = "bf16 (x6: fp32 products from three bfloat16 terms per operand, six v_mfma_f32_32x32x16_bf16 per K step, fp32 accumulation)"
                e["executed_tflops_bf16"] = 6.0 * hid / (ms * 1e-3) / 1e12
                e["executed_frac_of_bf16_peak"] = e["executed_tflops_bf16"] / PEAK_BF16_MFMA
            per_class[k] = e
        dom = max((k for k in flops if times.get(k, (0, 0))[1] > 0), key=lambda k: times[k][0], default=None)
        if dom:
            ms, n = times[dom]
            ach = flops[dom] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            dpeak = per_class[dom].get("peak", PEAK_FP32_MFMA)
            roof = dict(bound="mfma", kernel=dom, achieved=ach, peak=dpeak, unit="TFLOP/s", frac=ach / dpeak,
                        peak_note=("dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMAs per fp32 product (x6 kernel, vel_x6.hip); achieved = algorithmic fp32 FLOPs"
                                   if dpeak != PEAK_FP32_MFMA else "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                        traffic=traffic_all.get(dom), traffic_source=f"static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profiles/{PROFILE_TAG}_traffic.json (PMC counters cannot be read in-process)",
                        launches=int(n), avg_launch_ms=ms / max(n, 1), flop_per_launch=flops[dom] / max(n, 1),
                        **({k2: per_class[dom][k2] for k2 in ("matrix_pipe", "executed_tflops_bf16", "executed_frac_of_bf16_peak", "frac_of_fp32_mfma_peak") if k2 in per_class[dom]}),
                        whole_step=dict(tflops=sum(flops.values()) / psteps / (dt / args.steps) / 1e12,
                                        frac=sum(flops.values()) / psteps / (dt / args.steps) / 1e12 / PEAK_FP32_MFMA),
                        per_class=per_class)
        if args.workload == "cfg2":
            # radiance-only: gather / scatter bound: algorithmic bytes per step = 6912 B per valid sample + 13824 B per
            # appearance-masked sample (forward gathers + read-modify-write of the plane gradients, no reuse credit) + the per-ray
            # inputs/outputs, over the WHOLE step time (the MLP contractions ride along); peak = 8 TB/s HBM3E
            nbytes = (V * 6912.0 + M * 13824.0) / psteps + renders * n_rays * (24.0 + (5.0 + args.samples) * 4.0)
            gbs = nbytes / (dt / args.steps) / 1e9
            tr2 = _load_json(f"{PROFILE_TAG}_traffic_cfg2.json") or {}
            roof = dict(bound="hbm", kernel="whole step: plane gathers + plane-gradient scatters (k_density_q, k_og, k_tile_scatter, k_app_fwd gather)",
                        achieved=gbs, peak=PEAK_HBM_GBS, unit="GB/s", frac=gbs / PEAK_HBM_GBS, traffic=tr2.get("bytes_per_step"),
                        traffic_source=f"static: HBM bytes of the WHOLE step from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --workload cfg2`, profiles/{PROFILE_TAG}_traffic_cfg2.json",
                        bytes_per_step=nbytes, mfma=roof)

    out = {
        "metric": "training rays/sec (fwd+bwd incl. PDE loss), 'bat' scene", "value": value, "unit": "rays/s",
        "n_gpus": world, "process_group": comm_info, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": {"fp16band": "f32 (opt-in: fp16-input pre-pass of the PDE occupancy prefilter, fp32 re-evaluation band)",
                  "split16band": "f32 (opt-in: PDE occupancy prefilter with fp32 products emulated by two binary16 terms per operand on the fp16 MFMA, fp32 re-evaluation band)"
                  }.get(os.environ.get("NVFI_PDE_PREFILTER", "x6"), "f32")
                 + (" (opt-in: FORWARD of the training renders' velocity warp on the fp16-input MFMA, fp32 accumulation and stashes; adjoint, weight gradients, "
                    "PDE term and render MLP fp32)" if os.environ.get("NVFI_VEL_FP16_TRAIN", "0") == "1" else ""),
        "data": "synthetic",
        "config": {"workload": (f"chessboard.yaml (configs[3], one GPU's share): InDoorSeg box, {'x'.join(str(int(g)) for g in model.nvfi.gridSize.tolist())} grid, K=4, surround-box velocity gate, "
                                f"no white background, {model.nvfi.nSamples} samples/ray, 2 renders x {n_rays} rays + PDE on {n_pts} collocation points + plane regularisers + Adam, per GPU"
                                if scene == "chessboard" else
                                "bat.yaml + velocity field + PDE divergence loss (configs[2]): 199^3 grid, K=16, 128 samples/ray, "
                                f"2 renders x {n_rays} rays + PDE on {n_pts} collocation points + plane regularisers + Adam, per GPU"
                                if args.workload == "cfg3" else
                                "bat.yaml radiance-only (configs[1]): 199^3 grid, 128 samples/ray, 2048-ray batches of the 800x800 frame"),
                   "rays_per_step_per_gpu": n_rays * renders, "pde_points_per_gpu": n_pts if args.workload == "cfg3" else 0,
                   "grid": [int(g) for g in model.nvfi.gridSize.tolist()], "samples_per_ray": int(model.nvfi.nSamples), "parallelism": f"ray-sharded x{world}",
                   "driver": args.mode,
                   "pde_kept_fraction": (work["P_kept"] / n_pts if (work and args.workload == "cfg3" and n_pts) else None),      # P' / P of the profiled steps
                   "launch": (("three hipGraph replays per iteration (draw + fragment repack | PDE term on its stream | both renders + regularisers) around the eager gradient "
                               "exchange and the one-launch Adam; frame times, loss weights, learning rates and jitter read from a device record uploaded per iteration")
                              if (use_graph and world > 1) else
                              "one hipGraph replay per iteration (three captured streams; frame times, loss weights, learning rates and jitter read from a device record uploaded per iteration)"
                              if use_graph else "eager launches"),
                   "field": "live (updated by the optimiser)" if args.live else "stationary (optimiser steps a shadow copy)"},
        "work_per_step": work,
        "roofline": roof,
    }
    if use_graph:
        out["config"]["graph_capture_attempts_ms"] = [round(x, 3) for x in run.capture_ms]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(model, args.workload, scene=scene)
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out["cpu_baseline"] = {"error": repr(e)}
        ref = _load_json(f"{PROFILE_TAG}_cpu_bridge.json")
        if ref and "reference" in ref and scene == "bat":      # the reference itself (PyTorch CPU) on that same sample, measured in the build container
            out["reference_cpu"] = dict(ref["reference"], host=ref.get("host"), oracle_on_that_host=ref.get("oracle", {}).get("value"),
                                        source=f"profiles/{PROFILE_TAG}_cpu_bridge.json (tools/cpu_bridge.py; /root/reference cannot run on the GPU box)")
    # Other lines of the same build, measured here so that they are recorded with the headline (each in its own process, same K / W, no
    # roofline pass); none of them is `value`.
    default_invocation = (rank == 0 and world == 1 and args.mode == "fused" and args.workload == "cfg3" and scene == "bat" and not args.live and not args.no_extras
                          and not os.environ.get("NVFI_BENCH_CHILD") and os.environ.get("NVFI_PDE_PREFILTER", "x6") == "x6")
    if default_invocation:
        from tools.bench_extras import collect      # (round 5: the extras live in tools/bench_extras.py)
        out["extras"] = collect(args, out, model)
    if graph_line is not None:
        modes = {"hipgraph_replay": {"value": graph_line["value"], "ms_per_step": graph_line["ms_per_step"]},
                 "eager_three_streams": {"value": out["value"], "ms_per_step": out["ms_per_step"]}}
        if graph_line["value"] > out["value"]:      # the child's line carries its own roofline pass; the CPU baseline was measured here
            for k in ("cpu_baseline", "reference_cpu", "extras"):
                if k in out:
                    graph_line[k] = out[k]
            out = graph_line
        out["launch_modes"] = modes
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
