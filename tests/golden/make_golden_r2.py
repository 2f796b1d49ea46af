#!/usr/bin/env python
"""Round-2 golden vectors (build container only; imports the reference from /root/reference on PyTorch CPU):

    python tests/golden/make_golden_r2.py        # rewrites tests/golden/r2.npz

* field "C": BASELINE.json configs[1] in small - the bat-like field WITHOUT a velocity net (`use_vel: False`): the time coordinate
  of every lookup is the continuous normalised frame time (models/tensorf_keyframe.py:703-704), i.e. a true bilinear
  interpolation in (space, time) whose gradient lands on TWO time rows.  Eval renders at a fractional row, at an integer row,
  at t = 0 and beyond tmax (rows outside the plane: zero padding), and a train-mode render with every parameter gradient.
* grid maintenance on the small velocity fields (models/tensorf_keyframe.py:327-499): getDenseAlpha over the 60 frame times
  (plain and transfer), updateAlphaMask (mask volume + returned box), shrink (cropped planes, corrected box, step size, sample
  count) followed by an eval render of the shrunk field with its alpha mask, and upsample_volume_grid (all 12 planes, K 4 -> 6).
The fixtures are arrays only (inputs, parameters, outputs)."""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, make_cfg, camera_rays, npf  # noqa: E402


def build_field_c(R, seed=91):
    """bat-like anisotropic field, no velocity net; blob + perturbed time planes like field A."""
    import torch
    cfg = make_cfg(R, "A")
    cfg.nvfi.use_vel = False
    torch.manual_seed(seed)
    aabb = torch.stack([torch.tensor(cfg.nvfi[b]) for b in ["bbox_x", "bbox_y", "bbox_z"]], dim=-1)
    with contextlib.redirect_stdout(io.StringIO()):
        nv = R["NVFi"](cfg, "cpu", aabb, [20, 18, 16], [cfg.dataset.near, cfg.dataset.far])
    f = nv.nvfi
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for i in range(3):
            p = f.density_plane_space[i]
            H, W = p.shape[-2:]
            yy = torch.linspace(-1, 1, H)[:, None]
            xx = torch.linspace(-1, 1, W)[None, :]
            p.mul_((3.2 * torch.sqrt(torch.exp(-xx ** 2 / (2 * 0.45 ** 2)) * torch.exp(-yy ** 2 / (2 * 0.45 ** 2))))[None, None])
            f.density_plane_time[i].add_(0.2 * torch.randn(f.density_plane_time[i].shape, generator=g))
            f.app_plane_time[i].add_(0.3 * torch.randn(f.app_plane_time[i].shape, generator=g))
            f.app_plane_space[i].mul_(4.0)
        f.renderModule.mlp[-1].bias.add_(0.2 * torch.randn(3, generator=g))
    return cfg, nv


def build_field_s(R, kind, seed):
    """small velocity field with a TIGHT blob (so that the alpha mask / shrink actually crop something)."""
    import torch
    from make_golden import build_field
    cfg, nv = build_field(R, kind, seed=seed)
    f = nv.nvfi
    with torch.no_grad():
        for i in range(3):
            p = f.density_plane_space[i]
            H, W = p.shape[-2:]
            yy = torch.linspace(-1, 1, H)[:, None] - 0.1
            xx = torch.linspace(-1, 1, W)[None, :] + 0.15
            blob = 1.6 * torch.exp(-(xx ** 2 + yy ** 2) / (2 * 0.22 ** 2))
            if kind == "B":
                # density_shift = -5 keeps alpha above the mask threshold wherever the feature is >= -2.5: the chessboard-like field
                # gets O(1) positive (x,z) / (y,z) planes and an (x,y) plane that goes clearly negative away from the blob, so that
                # it has empty space to cull as well
                if i == 0:
                    p.mul_((6.0 * blob - 1.5)[None, None])
                else:
                    p.copy_(0.5 + 0.5 * torch.rand(p.shape, generator=torch.Generator().manual_seed(seed + 10 + i)))
            else:
                p.mul_(blob[None, None])
    return cfg, nv


def dump_sd(out, prefix, nv):
    for k, v in nv.state_dict().items():
        if not k.startswith("nvfi.vel.vel_net."):
            out[prefix + "sd:" + k] = npf(v)


def main():
    R = import_reference()
    import torch
    torch.set_num_threads(4)
    out = {}

    # ------------------------------------------------------------------ field C: radiance-only (configs[1])
    cfg, nv = build_field_c(R)
    f = nv.nvfi
    dump_sd(out, "C:", nv)
    meta = dict(aabb=npf(f.aabb), gridSize=np.array(f.gridSize.tolist(), np.int64), num_keyframes=f.num_keyframes, tmax=f.tmax,
                near=f.near_far[0], far=f.near_far[1], step_ratio=f.step_ratio, max_n_samples=f.max_n_samples,
                density_shift=f.density_shift, distance_scale=f.distance_scale, alphaMask_thres=f.alphaMask_thres,
                rayMarch_weight_thres=f.rayMarch_weight_thres, stepSize=float(f.stepSize), nSamples=f.nSamples,
                white_background=True, use_sur=0, eps=0.03)
    for k, v in meta.items():
        out["C:meta:" + k] = np.asarray(v)
    ren = R["Renderer"](nv, 0, 0, 2048)
    o, d = camera_rays(R, "A")
    out["C:rays_o"], out["C:rays_d"] = npf(o), npf(d)
    ts = f.tmax / (f.num_keyframes - 1)
    gen = torch.Generator().manual_seed(17)
    for name, tt in (("frac", 19.0 / 60.0), ("row", ts * 2), ("zero", 0.0), ("beyond", 0.93), ("last", 0.74)):
        r = ren.render(tt, R["Ray"](o, d, 0, 1), white_background=True, mode="test")
        out[f"C:render_{name}:t"] = np.float64(tt)
        for nm, v in zip(("rgb", "depth", "acc", "weight"), r[:4]):
            out[f"C:render_{name}:{nm}"] = npf(v)
    # continuous-time feature lookups
    xyzt = torch.rand(300, 4, generator=gen) * 2.3 - 1.15
    with torch.no_grad():
        out["C:feat:xyzt"] = npf(xyzt)
        df = f.compute_densityfeature(xyzt)
        out["C:feat:density"] = npf(df)
        out["C:feat:app"] = npf(f.compute_appfeature(xyzt))
    for name, tt in (("train_frac", 19.0 / 60.0), ("train_late", 0.71)):
        nv.zero_grad(set_to_none=True)
        target = torch.rand(o.shape[0], 3, generator=gen)
        torch.manual_seed(21)
        u = torch.rand(o.shape[0], 1)
        torch.manual_seed(21)
        r = ren.render(tt, R["Ray"](o, d, 0, 1), white_background=True, mode="train")
        gw = torch.rand(r[3].shape, generator=gen) * 0.01
        loss = torch.nn.functional.mse_loss(r[0], target) + 0.01 * r[1].mean() + 0.02 * (r[2] ** 2).mean() + (r[3] * gw).sum()
        loss.backward()
        out[f"C:{name}:t"] = np.float64(tt)
        out[f"C:{name}:u"], out[f"C:{name}:target"], out[f"C:{name}:gw"] = npf(u), npf(target), npf(gw)
        out[f"C:{name}:loss"] = npf(loss)
        for nm, v in zip(("rgb", "depth", "acc", "weight"), r[:4]):
            out[f"C:{name}:{nm}"] = npf(v)
        for k, p in nv.named_parameters():
            if name == "train_late" and "plane_time" not in k:
                continue
            out[f"C:{name}:grad:{k}"] = npf(p.grad) if p.grad is not None else np.zeros(0, np.float32)

    # ------------------------------------------------------------------ grid maintenance (f-3)
    for kind, seed in (("A", 233), ("B", 77)):
        cfg, nv = build_field_s(R, kind, seed)
        f = nv.nvfi
        f.eval()
        pre = f"{kind}s:"
        dump_sd(out, pre, nv)
        gs = f.gridSize.tolist()
        out[pre + "gridSize"] = np.array(gs, np.int64)
        with contextlib.redirect_stdout(io.StringIO()):
            alpha, dense = f.getDenseAlpha(gs)
            out[pre + "dense_alpha"] = npf(alpha)
            if kind == "A":
                alpha_t, _ = f.getDenseAlpha(gs, transfer=True)
                out[pre + "dense_alpha_transfer"] = npf(alpha_t)
            new_aabb = f.updateAlphaMask(gs)
        out[pre + "mask_volume"] = npf(f.alphaMask.alpha_volume)
        out[pre + "new_aabb"] = npf(new_aabb)
        with contextlib.redirect_stdout(io.StringIO()):
            f.shrink(new_aabb)
        out[pre + "shrink:aabb"] = npf(f.aabb)
        out[pre + "shrink:gridSize"] = np.array(f.gridSize.tolist(), np.int64)
        out[pre + "shrink:stepSize"] = np.float64(float(f.stepSize))
        out[pre + "shrink:nSamples"] = np.int64(f.nSamples)
        for nm in ("density_plane_space", "density_plane_time", "app_plane_space", "app_plane_time"):
            for i in range(3):
                out[pre + f"shrink:{nm}.{i}"] = npf(getattr(f, nm)[i])
        ren = R["Renderer"](nv, 0, 0, 2048)
        o, d = camera_rays(R, kind)
        r = ren.render(19.0 / 60.0, R["Ray"](o, d, 0, 1), white_background=cfg.dataset.white_background, mode="test")
        for nm, v in zip(("rgb", "depth", "acc"), r[:3]):
            out[pre + f"shrink:render:{nm}"] = npf(v)
        # upsample the shrunk field (the shipped schedule upsamples after shrinking, train_nvfi.py:337-347)
        tgt = [int(g * 1.3) + 1 for g in f.gridSize.tolist()]
        newK = 6 if kind == "A" else 16
        with contextlib.redirect_stdout(io.StringIO()):
            f.upsample_volume_grid(tgt, newK)
        out[pre + "up:target"] = np.array(tgt, np.int64)
        out[pre + "up:K"] = np.int64(newK)
        out[pre + "up:stepSize"] = np.float64(float(f.stepSize))
        out[pre + "up:nSamples"] = np.int64(f.nSamples)
        for nm in ("density_plane_space", "density_plane_time", "app_plane_space", "app_plane_time"):
            for i in range(3):
                out[pre + f"up:{nm}.{i}"] = npf(getattr(f, nm)[i])
        f.alphaMask = None
        r = ren.render(19.0 / 60.0, R["Ray"](o, d, 0, 1), white_background=cfg.dataset.white_background, mode="test")
        for nm, v in zip(("rgb", "depth", "acc"), r[:3]):
            out[pre + f"up:render:{nm}"] = npf(v)

    # ------------------------------------------------------------------ a-19 in TRAIN mode: the mask branch is differentiable
    # (tensorf_keyframe.py:673-676, 749-753): gradients of a loss on mask_map w.r.t. the MaskField parameters and - through the
    # weights - the density planes.  (vel_net also receives a gradient through the MaskField's input points in the reference; the
    # HIP path does not differentiate MaskField inputs, so that family is recorded but not compared.)
    from make_golden import build_field
    cfgA, nvA = build_field(R, "A")
    f = nvA.nvfi
    torch.manual_seed(41)
    f.mask_field = R["MaskField"](n_layer=4, n_dim=128, skips=[], mask_dim=8, input_dim=3, mask_act="softmax")
    for k, v in f.mask_field.state_dict().items():
        out["A:mask_train:sd:" + k] = npf(v)
    o, d = camera_rays(R, "A")
    f.train()
    nvA.zero_grad(set_to_none=True)
    g = torch.Generator().manual_seed(43)
    gm = torch.rand(o.shape[0], 8, generator=g)
    torch.manual_seed(21)
    u = torch.rand(o.shape[0], 1)
    torch.manual_seed(21)
    r = f(19.0 / 60.0, o, d, True, False)
    loss = (r[4] * gm).sum() + 0.5 * r[0].mean()
    loss.backward()
    out["A:mask_train:u"], out["A:mask_train:gm"] = npf(u), npf(gm)
    out["A:mask_train:map"], out["A:mask_train:rgb"], out["A:mask_train:loss"] = npf(r[4]), npf(r[0]), npf(loss)
    for k, p in f.mask_field.named_parameters():
        out["A:mask_train:grad:mask_field." + k] = npf(p.grad)
    for k in ("density_plane_space.0", "density_plane_time.1", "app_plane_space.2", "basis_mat.weight"):
        pp = dict(f.named_parameters())[k]
        out["A:mask_train:grad:" + k] = npf(pp.grad)
    f.mask_field = None

    # ------------------------------------------------------------------ shadingMode "SH" (a-17 inside the render call): app_dim = 27, no render MLP
    # (tensorf_base.py:196-197, tensorf_model_utils.py:292-296).  Field "D" = the bat-like field A with that shading: eval renders and a
    # train-mode render with every gradient.
    cfgD = make_cfg(R, "A")
    cfgD.nvfi.shadingMode = "SH"
    cfgD.nvfi.app_dim = 27
    torch.manual_seed(233)
    aabbD = torch.stack([torch.tensor(cfgD.nvfi[b]) for b in ["bbox_x", "bbox_y", "bbox_z"]], dim=-1)
    with contextlib.redirect_stdout(io.StringIO()):
        nvD = R["NVFi"](cfgD, "cpu", aabbD, [20, 18, 16], [cfgD.dataset.near, cfgD.dataset.far])
    f = nvD.nvfi
    gD = torch.Generator().manual_seed(71)
    with torch.no_grad():
        for i in range(3):
            pl = f.density_plane_space[i]
            H, W = pl.shape[-2:]
            yy = torch.linspace(-1, 1, H)[:, None]; xx = torch.linspace(-1, 1, W)[None, :]
            pl.mul_((3.2 * torch.sqrt(torch.exp(-xx ** 2 / (2 * 0.45 ** 2)) * torch.exp(-yy ** 2 / (2 * 0.45 ** 2))))[None, None])
            f.density_plane_time[i].add_(0.15 * torch.randn(f.density_plane_time[i].shape, generator=gD))
            f.app_plane_time[i].add_(0.25 * torch.randn(f.app_plane_time[i].shape, generator=gD))
            f.app_plane_space[i].mul_(6.0)
        f.basis_mat.weight.mul_(25.0)           # SH coefficients of O(1): colours leave the relu(. + 0.5) plateau
        last = f.vel_net.weight_net[-1][0]
        last.weight.mul_(4.0); last.bias.add_(0.3 * torch.randn(last.bias.shape, generator=gD))
    dump_sd(out, "D:", nvD)
    o, d = camera_rays(R, "A")
    renD = R["Renderer"](nvD, 0, 0, 2048)
    for name, tt in (("nonkey", 19.0 / 60.0), ("key", 0.25)):
        r = renD.render(tt, R["Ray"](o, d, 0, 1), white_background=True, mode="test")
        for nm, v in zip(("rgb", "depth", "acc"), r[:3]):
            out[f"D:render_{name}:{nm}"] = npf(v)
    nvD.zero_grad(set_to_none=True)
    tgtD = torch.rand(o.shape[0], 3, generator=gD)
    torch.manual_seed(21)
    out["D:train:u"] = npf(torch.rand(o.shape[0], 1))
    torch.manual_seed(21)
    r = renD.render(19.0 / 60.0, R["Ray"](o, d, 0, 1), white_background=True, mode="train")
    lossD = torch.nn.functional.mse_loss(r[0], tgtD) + 0.01 * r[1].mean()
    lossD.backward()
    out["D:train:target"], out["D:train:loss"], out["D:train:rgb"] = npf(tgtD), npf(lossD), npf(r[0])
    for k, pp in nvD.named_parameters():
        if k.startswith("nvfi.vel.vel_net.") or pp.grad is None:
            continue
        out["D:train:grad:" + k] = npf(pp.grad)

    # ------------------------------------------------------------------ the training loop itself, three iterations
    # The body of train_nvfi.py:139-249 (--static_dynamic) on field A with explicit inputs instead of the dataset: per iteration a train-mode
    # render at a non-keyframe time and one at a keyframe time (MSE against fixed targets), the L1 / TV regularisers with their decaying
    # weights, get_vel_loss, backward, torch.optim.Adam(get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99)).step(), lr decay.  Recorded:
    # the CPU-generator draws (jitter, collocation points) so that the HIP path can replay them, the loss of every iteration and the
    # parameters after the third step.
    from utils.tensorf_utils import TVLoss
    cfgL, nvL = build_field(R, "A")
    f = nvL.nvfi
    o, d = camera_rays(R, "A")
    gL = torch.Generator().manual_seed(61)
    tgt1, tgt2 = torch.rand(o.shape[0], 3, generator=gL), torch.rand(o.shape[0], 3, generator=gL)
    out["A:loop:target1"], out["A:loop:target2"] = npf(tgt1), npf(tgt2)
    renL = R["Renderer"](nvL, 0, 0, 2048)
    f.requires_grad_(True)
    opt = torch.optim.Adam(nvL.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    lr_factor = 0.1 ** (1 / 30000)
    L1w, tvd, tva, vw = 8e-4, 1.0, 1.0, 1.0
    tvreg = TVLoss()
    ts = f.tmax / (f.num_keyframes - 1)
    P = 4096
    for it in range(3):
        nvL.train(); renL.train()
        torch.manual_seed(100 + it); out[f"A:loop:{it}:u1"] = npf(torch.rand(o.shape[0], 1))
        torch.manual_seed(100 + it)
        r1 = renL.render(19.0 / 60.0, R["Ray"](o, d, 0, 1), white_background=True, mode="train")
        loss = torch.nn.functional.mse_loss(r1[0], tgt1)
        torch.manual_seed(200 + it); out[f"A:loop:{it}:u2"] = npf(torch.rand(o.shape[0], 1))
        torch.manual_seed(200 + it)
        r2 = renL.render(ts * 1, R["Ray"](o, d, 0, 1), white_background=True, mode="train")
        loss = loss + 1.0 * torch.nn.functional.mse_loss(r2[0], tgt2)
        L1w *= lr_factor; loss = loss + L1w * f.density_L1()
        tvd *= lr_factor; loss = loss + f.TV_loss_density(tvreg) * tvd
        tva *= lr_factor; loss = loss + f.TV_loss_app(tvreg) * tva
        vw *= lr_factor
        torch.manual_seed(300 + it)
        mn, mx = f.aabb
        pts = torch.rand(P, 3) * (mx - mn) + mn
        tt = torch.rand(P, 1)
        out[f"A:loop:{it}:points"], out[f"A:loop:{it}:t"] = npf(pts), npf(tt)
        torch.manual_seed(300 + it)
        loss_vel = nvL.get_vel_loss(P)
        if loss_vel > 0:
            loss = loss + vw * loss_vel
        out[f"A:loop:{it}:loss"] = npf(loss)
        out[f"A:loop:{it}:loss_vel"] = npf(loss_vel if torch.is_tensor(loss_vel) else torch.tensor(loss_vel))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        for gq in opt.param_groups:
            gq["lr"] = gq["lr"] * lr_factor
    for k, v in nvL.state_dict().items():
        if not k.startswith("nvfi.vel.vel_net."):
            out["A:loop:final:" + k] = npf(v)

    np.savez_compressed(os.path.join(HERE, "r2.npz"),
                        **{k: (v.astype(np.float32) if v.dtype == np.float64 and v.ndim > 0 else v) for k, v in out.items()})
    print("wrote", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "r2.npz")), "bytes")


if __name__ == "__main__":
    main()
