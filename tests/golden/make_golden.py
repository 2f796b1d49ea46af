#!/usr/bin/env python
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the reference implementation from /root/reference (PyTorch CPU), builds two small
synthetic fields and records inputs + expected outputs of every function on the hot path
 as float32 .npz fixtures next to this script.  The fixtures are data
(inputs, parameters, outputs); no reference source travels.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The reference pins torch==1.12.1; this container has torch 2.10 (CPU kernels) - the fixtures
are therefore "reference code on torch 2.10 CPU".
"""
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    """Stub the modules the reference imports at module top but never touches on the hot path."""
    sys.dont_write_bytecode = True
    for name in ["cv2", "torchvision", "torchvision.transforms", "lpips", "imageio", "wandb"]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            sys.modules[name] = m
    sys.modules["cv2"].COLORMAP_JET = 2
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    import torch  # noqa
    from models.nvfi import NVFi
    from models.renderer import Renderer
    from models.camera import Camera, Ray
    from models.tensorf_model_utils import AlphaGridMask, raw2alpha, SHRender
    from models.mask_field import MaskField
    from utils.cfgnode import CfgNode
    from utils.tensorf_utils import TVLoss
    return dict(NVFi=NVFi, Renderer=Renderer, Camera=Camera, Ray=Ray, AlphaGridMask=AlphaGridMask,
                raw2alpha=raw2alpha, SHRender=SHRender, MaskField=MaskField, CfgNode=CfgNode,
                TVLoss=TVLoss)


def pose_spherical(theta, phi, radius, center=(0.0, 0.0, 0.0)):
    """D-NeRF style camera-to-world (restated from the formula in datasets/load_blender.py:62-67)."""
    import torch
    th, ph = theta / 180.0 * np.pi, phi / 180.0 * np.pi
    tr = np.eye(4); tr[2, 3] = radius
    rp = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1.0]])
    rt = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1.0]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    c2w = flip @ rt @ rp @ tr
    c2w[:3, 3] += np.asarray(center)
    return torch.tensor(c2w, dtype=torch.float32)


def make_cfg(R, kind):
    import yaml
    with open(os.path.join(REF, "config/InDoorObj/bat.yaml")) as f:
        d = yaml.safe_load(f)
    cfg = R["CfgNode"](d)
    n = cfg.nvfi
    if kind == "A":  # bat-like, anisotropic box/grid so that axis swaps are caught
        n.bbox_x, n.bbox_y, n.bbox_z = [-2.0, 2.0], [-1.8, 2.1], [-2.2, 1.9]
        n.num_keyframes = 4
        n.num_keyframes_end = 4
        cfg.dataset.near, cfg.dataset.far = 1.0, 8.0
        cfg.dataset.white_background = True
    else:  # chessboard-like: surround-box gate, no white bg, K=16 time rows
        n.bbox_x, n.bbox_y, n.bbox_z = [-3.03, 3.03], [-3.03, 3.03], [-0.03, 6.03]
        n.sur_x, n.sur_y, n.sur_z = [-2.5, 2.5], [-2.5, 2.5], [0.02, 5.95]
        n.density_shift = -5
        n.distance_scale = 10
        n.num_keyframes = 16
        n.num_keyframes_end = 16
        cfg.dataset.near, cfg.dataset.far = 0.8, 8.1
        cfg.dataset.white_background = False
    return cfg


def build_field(R, kind, seed=233, shared_nets=None):
    import torch
    cfg = make_cfg(R, kind)
    torch.manual_seed(seed)
    bbox = [torch.tensor(cfg.nvfi[b]) for b in ["bbox_x", "bbox_y", "bbox_z"]]
    aabb = torch.stack(bbox, dim=-1)
    res = [20, 18, 16] if kind == "A" else [16, 17, 18]
    near_far = [cfg.dataset.near, cfg.dataset.far]
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        nvfi = R["NVFi"](cfg, "cpu", aabb, res, near_far)
    f = nvfi.nvfi
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        # semi-transparent blob + perturbed time planes (SURVEY 8c recipe, randomised)
        for i in range(3):
            p = f.density_plane_space[i]
            H, W = p.shape[-2:]
            yy = torch.linspace(-1, 1, H)[:, None]
            xx = torch.linspace(-1, 1, W)[None, :]
            blob = (3.2 if kind == "A" else 2.3) * torch.sqrt(torch.exp(-xx ** 2 / (2 * 0.45 ** 2)) * torch.exp(-yy ** 2 / (2 * 0.45 ** 2)))
            p.mul_(blob[None, None])
            f.density_plane_time[i].add_(0.15 * torch.randn(f.density_plane_time[i].shape, generator=g))
            f.app_plane_time[i].add_(0.25 * torch.randn(f.app_plane_time[i].shape, generator=g))
            f.app_plane_space[i].mul_(4.0)
        # livelier velocity / acceleration heads
        for net in (f.vel_net.weight_net, f.vel_net.a_weight_net):
            last = net[-1][0]
            last.weight.mul_(4.0)
            last.bias.add_(0.3 * torch.randn(last.bias.shape, generator=g))
        f.renderModule.mlp[-1].bias.add_(0.2 * torch.randn(3, generator=g))
        if shared_nets is not None:
            f.vel_net.load_state_dict(shared_nets["vel_net"])
            f.renderModule.load_state_dict(shared_nets["render"])
            f.basis_mat.load_state_dict(shared_nets["basis"])
    return cfg, nvfi


def npf(x):
    import torch
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x)


def dump_field(path, cfg, nvfi, kind):
    f = nvfi.nvfi
    out = {}
    for k, v in nvfi.state_dict().items():
        if k.startswith("nvfi.vel.vel_net."):
            continue  # duplicate registration of nvfi.vel_net.* (tensorf_keyframe.py:94,106)
        out["sd:" + k] = npf(v)
    meta = dict(
        aabb=npf(f.aabb), gridSize=np.array(f.gridSize.tolist(), np.int64), num_keyframes=f.num_keyframes,
        tmax=f.tmax, near=f.near_far[0], far=f.near_far[1], step_ratio=f.step_ratio,
        max_n_samples=f.max_n_samples, density_shift=f.density_shift, distance_scale=f.distance_scale,
        alphaMask_thres=f.alphaMask_thres, rayMarch_weight_thres=f.rayMarch_weight_thres,
        stepSize=float(f.stepSize), nSamples=f.nSamples, white_background=bool(cfg.dataset.white_background),
        use_sur=int(kind == "B"),
    )
    if kind == "B":
        meta["sur_bounds"] = npf(f.vel.bounds)
        meta["surround"] = npf(f.vel.surround)
    else:
        meta["eps"] = f.vel.eps
    for k, v in meta.items():
        out["meta:" + k] = np.asarray(v)
    np.savez(path, **out)


def camera_rays(R, kind, H=16, W=16, full=800, angle_x=0.6911112):
    """Centre HxW crop of an 800x800 pinhole camera (SURVEY 8d config 1 recipe)."""
    import torch
    center = (0, 0, 0) if kind == "A" else (0.0, 0.0, 3.0)
    pose = pose_spherical(30.0, -30.0, 4.0, center)
    focal = 0.5 * full / np.tan(0.5 * angle_x)
    cam = R["Camera"](pose, full, full, focal, torch.zeros(full, full, 3), 1.0, 8.0)
    o = cam.rays.ray_origins
    d = cam.rays.ray_directions
    # a coarse stride so the crop covers the object, not 16 adjacent pixels
    stride = 24
    i0 = full // 2 - (H // 2) * stride
    sl_i = slice(i0, i0 + H * stride, stride)
    j0 = full // 2 - (W // 2) * stride
    sl_j = slice(j0, j0 + W * stride, stride)
    return o[sl_i, sl_j].reshape(-1, 3).contiguous(), d[sl_i, sl_j].reshape(-1, 3).contiguous()


def main():
    R = import_reference()
    import torch
    torch.set_num_threads(4)
    cfgA, nvA = build_field(R, "A")
    shared = dict(vel_net=nvA.nvfi.vel_net.state_dict(), render=nvA.nvfi.renderModule.state_dict(),
                  basis=nvA.nvfi.basis_mat.state_dict())
    cfgB, nvB = build_field(R, "B", seed=77, shared_nets=shared)
    dump_field(os.path.join(HERE, "field_A.npz"), cfgA, nvA, "A")
    # field B shares the MLPs with A: drop them from its dump to keep the fixture small
    dump_field(os.path.join(HERE, "field_B.npz"), cfgB, nvB, "B")
    zb = dict(np.load(os.path.join(HERE, "field_B.npz")))
    zb = {k: v for k, v in zb.items() if not (k.startswith("sd:nvfi.vel_net") or k.startswith("sd:nvfi.renderModule")
                                              or k.startswith("sd:nvfi.basis_mat."))}
    np.savez(os.path.join(HERE, "field_B.npz"), **zb)

    fx = {}
    for kind, cfg, nv in (("A", cfgA, nvA), ("B", cfgB, nvB)):
        f = nv.nvfi
        ren = R["Renderer"](nv, 0, 0, 2048)
        o, d = camera_rays(R, kind)
        fx[f"{kind}:rays_o"], fx[f"{kind}:rays_d"] = npf(o), npf(d)
        gen = torch.Generator().manual_seed(5 if kind == "A" else 6)

        # ---- a-3 sample_ray
        f.eval()
        pts, z, valid = f.sample_ray(o, d)
        fx[f"{kind}:sample_eval:pts"], fx[f"{kind}:sample_eval:z"], fx[f"{kind}:sample_eval:valid"] = npf(pts[:32]), npf(z), npf(valid)
        f.train()
        torch.manual_seed(11)
        u = torch.rand(o.shape[0], 1)
        torch.manual_seed(11)
        pts, z, valid = f.sample_ray(o, d)
        fx[f"{kind}:sample_train:u"] = npf(u)
        fx[f"{kind}:sample_train:pts"], fx[f"{kind}:sample_train:z"], fx[f"{kind}:sample_train:valid"] = npf(pts[:32]), npf(z), npf(valid)
        # slab-entry branch of sample_ray (all origin coordinates outside the box range)
        o_far = torch.tensor([[9.0, 9.5, 10.0]]).repeat(8, 1)
        d_far = -o_far / o_far.norm(dim=-1, keepdim=True) + 0.05 * torch.randn(8, 3, generator=gen)
        d_far[0, 1] = 0.0
        f.eval()
        pts, z, valid = f.sample_ray(o_far, d_far)
        fx[f"{kind}:sample_far:o"], fx[f"{kind}:sample_far:d"] = npf(o_far), npf(d_far)
        fx[f"{kind}:sample_far:pts"], fx[f"{kind}:sample_far:z"], fx[f"{kind}:sample_far:valid"] = npf(pts), npf(z), npf(valid)

        # ---- a-7/a-8 velocity basis
        N = 257
        xt = torch.cat([torch.rand(N, 3, generator=gen) * 2.1 - 1.05, torch.rand(N, 1, generator=gen)], -1)
        with torch.no_grad():
            fx[f"{kind}:vel:xt"] = npf(xt)
            fx[f"{kind}:vel:u"] = npf(f.vel_net(xt))
            fx[f"{kind}:vel:get_vel"] = npf(f.vel_net.get_vel(xt))
            fx[f"{kind}:vel:gated"] = npf(f.vel(xt))

        # ---- a-6 integrate_pos (eval mode: inputs are cloned by the reference)
        f.eval()
        ts = f.tmax / (f.num_keyframes - 1)
        x0 = torch.rand(N, 3, generator=gen) * 1.9 - 0.95
        t0 = torch.rand(N, 1, generator=gen)  # up to 1.0 > tmax: multi-step extrapolation
        base = torch.round((t0 / ts).clamp(0.0, f.num_keyframes - 1)) * ts
        with torch.no_grad():
            xk = f.integrate_pos(x0.clone(), t0.clone(), base.clone())
        fx[f"{kind}:integrate:x0"], fx[f"{kind}:integrate:t"], fx[f"{kind}:integrate:base"] = npf(x0), npf(t0), npf(base)
        fx[f"{kind}:integrate:xk"] = npf(xk)
        # forward in time (train_segm.py:166 argument order, negative dt)
        t1 = torch.rand(N, 1, generator=gen) * 0.25 + 0.5
        with torch.no_grad():
            xf = f.integrate_pos(x0.clone(), torch.zeros_like(t1), t1.clone())
        fx[f"{kind}:integrate_fwd:t_target"] = npf(t1)
        fx[f"{kind}:integrate_fwd:xk"] = npf(xf)

        # ---- a-9/a-12 feature lookups incl. out-of-range points and fractional time rows
        xyzt = torch.rand(N, 4, generator=gen) * 2.3 - 1.15
        xyzt[:8, 3] = torch.tensor([-1.0, 1.0, 0.0, 1.0 / 3.0, -1.0 / 3.0, 0.99999, -0.99999, 0.5])
        with torch.no_grad():
            fx[f"{kind}:feat:xyzt"] = npf(xyzt)
            df = f.compute_densityfeature(xyzt)
            fx[f"{kind}:feat:density"] = npf(df)
            fx[f"{kind}:feat:app"] = npf(f.compute_appfeature(xyzt))
            sig = f.feature2density(df, {})
            fx[f"{kind}:feat:sigma"] = npf(sig)

        # ---- a-10/a-11 sigma -> alpha -> weights
        sg = torch.rand(37, 41, generator=gen) ** 4 * 30.0
        sg[0] = 0.0
        sg[1, 5] = 900.0
        ds = torch.full((37, 41), float(f.stepSize)) * f.distance_scale
        ds[:, -1] = 0
        a, w, bgw = R["raw2alpha"](sg, ds)
        fx[f"{kind}:alpha:sigma"], fx[f"{kind}:alpha:dist"] = npf(sg), npf(ds)
        fx[f"{kind}:alpha:alpha"], fx[f"{kind}:alpha:weight"] = npf(a), npf(w)
        xs = torch.linspace(-30, 30, 121)
        fx[f"{kind}:alpha:softplus_in"] = npf(xs)
        fx[f"{kind}:alpha:softplus_out"] = npf(torch.nn.functional.softplus(xs + f.density_shift))

        # ---- a-13 render MLP
        pts_n = torch.rand(N, 3, generator=gen) * 2 - 1
        vd = torch.randn(N, 3, generator=gen)
        feat = torch.randn(N, 32, generator=gen) * 0.3
        with torch.no_grad():
            fx[f"{kind}:mlp:pts"], fx[f"{kind}:mlp:view"], fx[f"{kind}:mlp:feat"] = npf(pts_n), npf(vd), npf(feat)
            fx[f"{kind}:mlp:rgb"] = npf(f.renderModule(pts_n, vd, feat, {}))

        # ---- a-1,2,5,14 end-to-end eval renders
        tk = ts * 2  # a keyframe time
        cases = dict(key=(tk, False), nonkey=(19.0 / 60.0, False), extrap=(0.93, False), transfer=(0.2, True))
        for name, (tt, transfer) in cases.items():
            out = ren.render(tt, R["Ray"](o, d, 0, 1), white_background=cfg.dataset.white_background,
                             mode="test", transfer_vel=transfer)
            fx[f"{kind}:render_{name}:t"] = np.float64(tt)
            for nm, v in zip(("rgb", "depth", "acc", "weight"), out[:4]):
                fx[f"{kind}:render_{name}:{nm}"] = npf(v)
        out = ren.render(19.0 / 60.0, R["Ray"](o, d, 0, 1), white_background=(not cfg.dataset.white_background), mode="test")
        for nm, v in zip(("rgb", "depth", "acc", "weight"), out[:4]):
            fx[f"{kind}:render_flipbg:{nm}"] = npf(v)
        # a-15 alpha-mask culling (eval only)
        gs = f.gridSize.tolist()
        vol = (torch.rand(gs[2], gs[1], gs[0], generator=gen) > 0.35).float()
        f.alphaMask = R["AlphaGridMask"]("cpu", f.aabb, vol)
        out = ren.render(19.0 / 60.0, R["Ray"](o, d, 0, 1), white_background=cfg.dataset.white_background, mode="test")
        fx[f"{kind}:render_amask:volume"] = npf(vol)
        for nm, v in zip(("rgb", "depth", "acc", "weight"), out[:4]):
            fx[f"{kind}:render_amask:{nm}"] = npf(v)
        pq = torch.rand(300, 3, generator=gen) * 2.2 - 1.1
        fx[f"{kind}:amask:pts"] = npf(pq)
        fx[f"{kind}:amask:alpha"] = npf(f.alphaMask.sample_alpha(pq))
        f.alphaMask = None

        # ---- a-18 train-mode render + mse backward: every parameter gradient
        for name, tt in (("train_nonkey", 19.0 / 60.0), ("train_key", tk), ("train_extrap", 0.93)):
            nv.zero_grad(set_to_none=True)
            target = torch.rand(o.shape[0], 3, generator=gen)
            torch.manual_seed(21)
            u = torch.rand(o.shape[0], 1)
            coin = bool(torch.rand((1,)) < 0.5)
            torch.manual_seed(21)
            out = ren.render(tt, R["Ray"](o, d, 0, 1), white_background=cfg.dataset.white_background, mode="train")
            # a loss that exercises every output: rgb mse + depth + acc + weights
            gw = torch.rand(out[3].shape, generator=gen) * 0.01
            loss = torch.nn.functional.mse_loss(out[0], target) + 0.01 * out[1].mean() + 0.02 * (out[2] ** 2).mean() + (out[3] * gw).sum()
            loss.backward()
            fx[f"{kind}:{name}:t"] = np.float64(tt)
            fx[f"{kind}:{name}:u"], fx[f"{kind}:{name}:coin"] = npf(u), np.asarray(coin)
            fx[f"{kind}:{name}:target"], fx[f"{kind}:{name}:gw"] = npf(target), npf(gw)
            fx[f"{kind}:{name}:loss"] = npf(loss)
            for nm, v in zip(("rgb", "depth", "acc", "weight"), out[:4]):
                fx[f"{kind}:{name}:{nm}"] = npf(v)
            if name == "train_extrap" and kind == "B":
                keep = ("vel_net.weight_net.0", "vel_net.weight_net.5", "density_plane_time")
            elif name == "train_key":
                keep = ("density_plane", "basis_mat", "renderModule.mlp.4")
            elif name == "train_extrap":
                keep = ("vel_net.weight_net.1", "vel_net.weight_net.5", "app_plane_time")
            elif kind == "B":
                keep = ("plane", "basis_mat", "renderModule.mlp.0", "renderModule.mlp.4", "weight_net.0", "weight_net.2", "weight_net.5")
            else:
                keep = None
            for k, p in nv.named_parameters():
                if k.startswith("nvfi.vel.vel_net."):
                    continue
                if keep is not None and not any(s in k for s in keep):
                    continue
                fx[f"{kind}:{name}:grad:{k}"] = npf(p.grad) if p.grad is not None else np.zeros(0, np.float32)

        # ---- a-16 PDE loss with explicit collocation points
        P = 4096
        mn, mx = f.aabb
        torch.manual_seed(31)
        points = torch.rand(P, 3) * (mx - mn) + mn
        t = torch.rand(P, 1)
        torch.manual_seed(31)
        nv.zero_grad(set_to_none=True)
        f.train()
        loss = nv.get_vel_loss(P)
        loss.backward()
        fx[f"{kind}:pde:points"], fx[f"{kind}:pde:t"] = npf(points), npf(t)
        fx[f"{kind}:pde:loss"] = npf(loss)
        # kept mask recomputed the way nvfi.py:50-64 does
        with torch.no_grad():
            pn = f.normalize_coord(points)
            base = torch.round((t / ts).clamp(0.0, f.num_keyframes - 1)) * ts
            pp = f.integrate_pos(pn.clone(), t.clone(), base)
            sf = f.compute_densityfeature(torch.cat([pp, f.normalize_time_coord(base)], -1))
            sgm = f.feature2density(sf, {})
            alpha = 1 - torch.exp(-sgm * 0.01 * 25)
            kept = alpha >= f.alphaMask_thres
        fx[f"{kind}:pde:kept"] = npf(kept)
        fx[f"{kind}:pde:alpha"] = npf(alpha)
        from functorch import vmap, jacrev
        xk = torch.cat([pn, t], -1)[kept][:64]
        jac = vmap(jacrev(lambda q: f.vel_net(q)))(xk)
        fx[f"{kind}:pde:jac64"] = npf(jac)
        for k, p in f.vel_net.named_parameters():
            if kind == "B" and not any(s in k for s in ("net.0", "net.3", "net.5")):
                continue
            fx[f"{kind}:pde:grad:{k}"] = npf(p.grad)

        # ---- next-row f-1: plane regularisers
        tv = R["TVLoss"]()
        with torch.no_grad():
            fx[f"{kind}:regs:L1"] = npf(f.density_L1())
            fx[f"{kind}:regs:TVd"] = npf(f.TV_loss_density(tv))
            fx[f"{kind}:regs:TVa"] = npf(f.TV_loss_app(tv))

    # ---- a-19 mask branch (field A), bypassing Renderer's 3-channel reshape
    f = nvA.nvfi
    torch.manual_seed(41)
    f.mask_field = R["MaskField"](n_layer=4, n_dim=128, skips=[], mask_dim=8, input_dim=3, mask_act="softmax") \
        if "input_dim" in R["MaskField"].__init__.__code__.co_varnames else R["MaskField"]()
    f.eval()
    o, d = camera_rays(R, "A")
    with torch.no_grad():
        out = f(19.0 / 60.0, o, d, True, False)
    fx["A:mask:map"] = npf(out[4])
    for k, v in f.mask_field.state_dict().items():
        fx["A:mask:sd:" + k] = npf(v)
    f.mask_field = None

    # ---- a-17 SH shading
    g = torch.Generator().manual_seed(51)
    vd = torch.randn(200, 3, generator=g)
    vd = vd / vd.norm(dim=-1, keepdim=True)
    ft = torch.randn(200, 27, generator=g)
    fx["sh:view"], fx["sh:feat"] = npf(vd), npf(ft)
    fx["sh:rgb"] = npf(R["SHRender"](None, vd, ft, {}))

    np.savez_compressed(os.path.join(HERE, "hotpath.npz"), **{k: (v.astype(np.float32) if v.dtype == np.float64 and v.ndim > 0 else v) for k, v in fx.items()})
    tot = sum(os.path.getsize(os.path.join(HERE, n)) for n in os.listdir(HERE) if n.endswith(".npz"))
    print("fixtures written, total bytes", tot)


if __name__ == "__main__":
    main()
