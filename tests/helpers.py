"""Shared helpers: build the product NVFi module (nvfi_amd.models) from a golden field fixture."""
import os

import numpy as np
import torch

from conftest import GOLD


def field_cfg(meta):
    from nvfi_amd.utils import CfgNode
    n = dict(model_name="TensorVMKeyframeTimeKplane", density_n_comp=[24, 24, 24], appearance_n_comp=[48, 48, 48], app_dim=32,
             densityMode="Density", shadingMode="MLP_PE", alphaMask_thres=float(meta["alphaMask_thres"]),
             rayMarch_weight_thres=float(meta["rayMarch_weight_thres"]), density_shift=float(meta["density_shift"]),
             distance_scale=float(meta["distance_scale"]), pos_pe=6, view_pe=6, fea_pe=6, featureC=128,
             step_ratio=float(meta["step_ratio"]), fea2denseAct="softplus", max_n_samples=int(meta["max_n_samples"]),
             num_keyframes=int(meta["num_keyframes"]), num_keyframes_end=int(meta["num_keyframes"]), tmax=float(meta["tmax"]),
             use_vel=True)
    if int(meta.get("use_sur", 0)):
        sur = np.asarray(meta["surround"]).reshape(2, 3)
        n["sur_x"], n["sur_y"], n["sur_z"] = [[float(sur[0, i]), float(sur[1, i])] for i in range(3)]
    else:
        n["eps"] = float(meta["eps"])
    return CfgNode(dict(nvfi=n))


def load_meta(kind):
    z = np.load(os.path.join(GOLD, f"field_{kind}.npz"))
    meta = {k[5:]: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files if k.startswith("meta:")}
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd:")}
    return meta, sd


def make_model(kind, device="cuda", use_vel=True):
    from nvfi_amd.models import NVFi
    meta, sd = load_meta(kind)
    if kind == "B":  # field B shares the MLPs of field A
        _, sda = load_meta("A")
        for k, v in sda.items():
            sd.setdefault(k, v)
    cfg = field_cfg(meta)
    cfg.nvfi.use_vel = use_vel
    aabb = torch.tensor(np.asarray(meta["aabb"]).reshape(2, 3), dtype=torch.float32)
    res = [int(g) for g in meta["gridSize"]]
    m = NVFi(cfg, "cpu", aabb, res, [float(meta["near"]), float(meta["far"])])
    own = m.state_dict()
    for k, v in sd.items():
        if k in own:
            own[k].copy_(torch.from_numpy(np.ascontiguousarray(v)))
    m = m.to(device)
    assert m.nvfi.nSamples == int(meta["nSamples"]), (m.nvfi.nSamples, meta["nSamples"])
    return m, meta


def named_grads(model):
    """{reference parameter name: grad (numpy, logical NCHW)} skipping the duplicated vel.vel_net keys."""
    out = {}
    for k, p in model.named_parameters():
        if k.startswith("nvfi.vel.vel_net."):
            continue
        out[k[len("nvfi."):]] = None if p.grad is None else p.grad.detach().cpu().contiguous().numpy()
    return out
