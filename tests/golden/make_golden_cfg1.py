#!/usr/bin/env python
"""Golden vectors for BASELINE.json configs[0] - "config/InDoorObj/bat.yaml, 64x64 crop, 32 samples/ray, PyTorch CPU reference" -
generated from the REFERENCE implementation in the build container:  python tests/golden/make_golden_cfg1.py

Scene: bat.yaml field at 64^3, K = 16, reference init under torch.manual_seed(233), density space planes
multiplied by the blob 3.2*sqrt(g(x) g(y)), g(u) = exp(-u^2 / (2*0.35^2)); camera pose_spherical(30, -30, 4), 800x800,
camera_angle_x 0.6911112, centre 64x64 crop -> 4096 rays (two chunks of 2048); 32 samples per ray via max_n_samples = 32 and
step_ratio = (far - near) / (32 * mean(units)); t = 19/60 (non-keyframe) and t = 0.30 (keyframe); targets uniform random, seed 233.

Writes tests/golden/cfg1.npz: the field (state_dict + meta), rays, eval renders at both times, and a train-mode step at t = 19/60
(jitter from the CPU generator, MSE against the targets): rgb/depth/acc, loss, every MLP gradient in full and, for the 12 plane
tensors, the gradient's sum / abs-sum / a strided sample (the planes' full gradients would double the fixture)."""
import io
import contextlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, pose_spherical, npf, REF  # noqa: E402


def main():
    R = import_reference()
    import torch
    import yaml
    torch.set_num_threads(8)
    with open(os.path.join(REF, "config/InDoorObj/bat.yaml")) as f:
        cfg = R["CfgNode"](yaml.safe_load(f))
    n = cfg.nvfi
    near, far = 1.0, 8.0
    aabb = torch.stack([torch.tensor(n[b]) for b in ["bbox_x", "bbox_y", "bbox_z"]], dim=-1)
    G, S = 64, 32
    units = ((aabb[1] - aabb[0]) / (G - 1)).mean().item()
    n.max_n_samples = S
    n.step_ratio = (far - near) / (S * units)
    torch.manual_seed(233)
    with contextlib.redirect_stdout(io.StringIO()):
        nv = R["NVFi"](cfg, "cpu", aabb, [G, G, G], [near, far])
    f = nv.nvfi
    assert f.nSamples == S, f.nSamples
    with torch.no_grad():
        for i in range(3):
            p = f.density_plane_space[i]
            H, W = p.shape[-2:]
            yy = torch.linspace(-1, 1, H)[:, None]
            xx = torch.linspace(-1, 1, W)[None, :]
            p.mul_((3.2 * torch.sqrt(torch.exp(-xx ** 2 / (2 * 0.35 ** 2)) * torch.exp(-yy ** 2 / (2 * 0.35 ** 2))))[None, None])
        # the initial velocity head is ~0: give the warp something to do (same tweak as the small fixtures)
        last = f.vel_net.weight_net[-1][0]
        last.weight.mul_(4.0); last.bias.add_(0.2)
    out = {}
    for k, v in nv.state_dict().items():
        if not k.startswith("nvfi.vel.vel_net."):
            out["sd:" + k] = npf(v)
    meta = dict(aabb=npf(f.aabb), gridSize=np.array(f.gridSize.tolist(), np.int64), num_keyframes=f.num_keyframes, tmax=f.tmax,
                near=near, far=far, step_ratio=f.step_ratio, max_n_samples=f.max_n_samples, density_shift=f.density_shift,
                distance_scale=f.distance_scale, alphaMask_thres=f.alphaMask_thres, rayMarch_weight_thres=f.rayMarch_weight_thres,
                stepSize=float(f.stepSize), nSamples=f.nSamples, white_background=True, use_sur=0, eps=f.vel.eps)
    for k, v in meta.items():
        out["meta:" + k] = np.asarray(v)

    full = 800
    focal = 0.5 * full / np.tan(0.5 * 0.6911112)
    cam = R["Camera"](pose_spherical(30.0, -30.0, 4.0), full, full, focal, torch.zeros(full, full, 3), near, far)
    c0 = full // 2 - 32
    o = cam.rays.ray_origins[c0:c0 + 64, c0:c0 + 64].reshape(-1, 3).contiguous()
    d = cam.rays.ray_directions[c0:c0 + 64, c0:c0 + 64].reshape(-1, 3).contiguous()
    out["rays_o"], out["rays_d"] = npf(o), npf(d)
    ren = R["Renderer"](nv, 0, 0, 2048)
    for name, t in (("nonkey", 19.0 / 60.0), ("key", 0.30)):
        r = ren.render(t, R["Ray"](o, d, 0, 1), white_background=True, mode="test")
        for nm, v in zip(("rgb", "depth", "acc"), r[:3]):
            out[f"eval_{name}:{nm}"] = npf(v)
    # train step at the non-keyframe time
    torch.manual_seed(233)
    target = torch.rand(4096, 3)
    out["train:target"] = npf(target)
    torch.manual_seed(21)
    u = torch.cat([torch.rand(2048, 1), torch.rand(2048, 1)])      # one jitter draw per 2048-ray chunk, in order
    out["train:u"] = npf(u)
    torch.manual_seed(21)
    nv.zero_grad()
    r = ren.render(19.0 / 60.0, R["Ray"](o, d, 0, 1), white_background=True, mode="train")
    loss = torch.nn.functional.mse_loss(r[0], target)
    loss.backward()
    out["train:loss"] = np.float64(loss.item())
    for nm, v in zip(("rgb", "depth", "acc"), r[:3]):
        out[f"train:{nm}"] = npf(v)
    for k, p in nv.named_parameters():
        if k.startswith("nvfi.vel.vel_net.") or p.grad is None:
            continue
        g = p.grad
        if "plane" in k:
            flat = g.reshape(-1)
            out[f"train:gsum:{k}"] = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
            out[f"train:gsample:{k}"] = npf(flat[::97])
        else:
            out[f"train:grad:{k}"] = npf(g)
    np.savez_compressed(os.path.join(HERE, "cfg1.npz"), **out)
    print("wrote", len(out), "arrays; acc mean", float(out["eval_nonkey:acc"].mean()), "loss", loss.item())


if __name__ == "__main__":
    main()
