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


def model_from_npz(z, prefix, device="cuda", use_vel=True):
    """Product NVFi module from a fixture that stores `<prefix>sd:*` / `<prefix>meta:*` arrays (tests/golden/r2.npz)."""
    from nvfi_amd.models import NVFi
    pm, ps = prefix + "meta:", prefix + "sd:"
    meta = {k[len(pm):]: (z[k].item() if z[k].ndim == 0 else z[k]) for k in z.files if k.startswith(pm)}
    cfg = field_cfg(meta)
    cfg.nvfi.use_vel = use_vel
    aabb = torch.tensor(np.asarray(meta["aabb"]).reshape(2, 3), dtype=torch.float32)
    m = NVFi(cfg, "cpu", aabb, [int(g) for g in meta["gridSize"]], [float(meta["near"]), float(meta["far"])])
    own = m.state_dict()
    n = 0
    for k in z.files:
        if k.startswith(ps) and k[len(ps):] in own:
            own[k[len(ps):]].copy_(torch.from_numpy(np.ascontiguousarray(z[k])))
            n += 1
    assert n >= 19, n
    m = m.to(device)
    assert m.nvfi.nSamples == int(meta["nSamples"]), (m.nvfi.nSamples, meta["nSamples"])
    return m, meta


# ---- the 1e-4 relative contract (BASELINE.json north_star) -------------------------------------------------------------------
# rgb / depth / acc are compared with a PURE relative tolerance of 1e-4 plus the fp32 rounding floor of the composite itself
# (rgb and acc are sums of weights <= 1 and `1 - acc`: a few ulp of 1.0; depth adds `(1 - acc) * far`: a few ulp of far).
# A sample whose weight lies within rounding of the 1e-4 appearance threshold (or a point within rounding of a box face) may take the
# other branch than the reference; that moves its ray by at most ~1e-4 ABSOLUTE.  Such rays are COUNTED: at most `max_frac` of the
# rays may need that band, and none may leave it.
FP32_FLOOR = {"rgb": 2e-6, "acc": 2e-6, "depth": 2e-5, "weight": 2e-6}


def assert_contract(got, ref, what, rtol=1e-4, max_frac=0.005, band=None, label=""):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (label, got.shape, ref.shape)
    err = np.abs(got - ref)
    ok = err <= rtol * np.abs(ref) + FP32_FLOOR[what]
    rays_bad = (~ok).reshape(ok.shape[0], -1).any(axis=1)
    n_bad, n = int(rays_bad.sum()), ok.shape[0]
    big = np.abs(ref) > 1e-3
    max_rel = float((err[big] / np.abs(ref[big])).max()) if big.any() else 0.0
    band = 1e-4 * max(1.0, float(np.abs(ref).max())) if band is None else band
    print(f"[contract] {label}:{what}: max rel err {max_rel:.3e}, rays outside pure rtol={rtol:g}: {n_bad}/{n}, max abs err {err.max():.3e}")
    assert n_bad <= max(1, int(max_frac * n)), f"{label}:{what}: {n_bad}/{n} rays outside rtol={rtol:g} (max rel {max_rel:.3e})"
    assert err.max() <= band + rtol * np.abs(ref).max(), f"{label}:{what}: max abs err {err.max():.3e} leaves the threshold-flip band {band:.1e}"


def maskfield_fp16_oracle(params, pts, g=None, round16=True):
    """numpy restatement of MaskField (reference models/mask_field.py:68-83 as train_segm.py:97-102 builds it: Linear + ReLU x n_layer,
    Linear -> softmax) in the arithmetic of the fp16-input MFMA mode (mask.hip: k_maskfield_fwd16 / _bwd16): every MFMA operand -
    weights, layer inputs, and in the backward the adjoints - is rounded to fp16 (round-to-nearest-even), products accumulate in fp32,
    bias / ReLU / softmax and every stash stay fp32, weight gradients are fp32 products of the UNROUNDED fp32 stashes.
    round16=False gives the plain fp32 network (sanity check against the reference goldens).
    params: [W0, b0, W1, b1, ...] (out, in) row-major.  Returns mask (N, K) and, with g (N, K) = d loss / d mask, the parameter grads."""
    r = (lambda a: a.astype(np.float16).astype(np.float32)) if round16 else (lambda a: a.astype(np.float32))
    Ws, bs = params[0::2], params[1::2]
    h = [np.asarray(pts, np.float32)]
    for l, (W, b) in enumerate(zip(Ws, bs)):
        z = r(h[-1]) @ r(np.asarray(W, np.float32)).T + np.asarray(b, np.float32)
        if l + 1 < len(Ws):
            h.append(np.maximum(z, 0.0).astype(np.float32))
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    mask = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    if g is None:
        return mask
    g = np.asarray(g, np.float32)
    gz = (mask * (g - (mask * g).sum(axis=1, keepdims=True))).astype(np.float32)      # softmax backward of sum(mask * g)
    grads = [None] * len(params)
    for l in range(len(Ws) - 1, -1, -1):
        grads[2 * l] = (gz.T @ h[l]).astype(np.float32)          # fp32 weight gradient from the fp32 stashes
        grads[2 * l + 1] = gz.sum(axis=0).astype(np.float32)
        if l > 0:
            gh = r(gz) @ r(np.asarray(Ws[l], np.float32))        # dgrad on the fp16-input MFMA
            gz = (gh * (h[l] > 0)).astype(np.float32)
    return mask, grads
