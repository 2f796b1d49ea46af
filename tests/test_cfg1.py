"""BASELINE.json configs[0]: bat.yaml field at 64^3, centre 64x64 crop of the 800x800 camera (4096 rays = two 2048-ray chunks),
32 samples per ray - the reference's own CPU-runnable case.  Golden vectors come from the reference implementation
(tests/golden/make_golden_cfg1.py).  CPU: the oracle against them; GPU: the HIP path (Renderer -> ctypes -> C ABI) against them."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, assert_grad, relerr
from helpers import assert_contract

PATH = os.path.join(GOLD, "cfg1.npz")
T_NONKEY, T_KEY = 19.0 / 60.0, 0.30


@pytest.fixture(scope="module")
def g1():
    return np.load(PATH)


def _check_planes(g1, grads, tol):
    n = 0
    for k in g1.files:
        if not k.startswith("train:gsum:nvfi."):
            continue
        name = k[len("train:gsum:nvfi."):]
        g = np.asarray(grads[name], np.float64).reshape(-1)
        ref_sum, ref_abs = g1[k]
        assert abs(np.abs(g).sum() - ref_abs) <= tol * ref_abs, name
        assert abs(g.sum() - ref_sum) <= tol * ref_abs, name
        ref_s = g1["train:gsample:nvfi." + name]
        if np.abs(ref_s).max() > 0:
            assert relerr(g[::97].astype(np.float32), ref_s) < 5 * tol, name
        n += 1
    assert n == 12


def test_oracle_cfg1_matches_reference(g1):
    from oracle import oracle as orc
    fs = orc.FieldSpec.from_npz(PATH)
    o, d = g1["rays_o"], g1["rays_d"]
    for name, t in (("nonkey", T_NONKEY), ("key", T_KEY)):
        r = orc.render(fs, o, d, t, train=False, white_bg=True)
        for nm, got in (("rgb", r.rgb), ("depth", r.depth), ("acc", r.acc)):
            assert_contract(got, g1[f"eval_{name}:{nm}"], nm, label=f"oracle cfg1 eval_{name}")
    r = orc.render(fs, o, d, T_NONKEY, u=g1["train:u"], train=True, white_bg=True, keep_ctx=True)
    assert_contract(r.rgb, g1["train:rgb"], "rgb", label="oracle cfg1 train")
    assert_contract(r.depth, g1["train:depth"], "depth", label="oracle cfg1 train")
    tgt = g1["train:target"]
    loss = float(np.mean((r.rgb.astype(np.float64) - tgt) ** 2))
    assert abs(loss - float(g1["train:loss"])) < 1e-5 * float(g1["train:loss"]) + 1e-8
    grads = r.backward(fs, g_rgb=(2.0 * (r.rgb - tgt) / tgt.size).astype(np.float32))
    r.free()
    n = 0
    for k in g1.files:
        if k.startswith("train:grad:nvfi."):
            name = k[len("train:grad:nvfi."):]
            if name == "basis_mat_density.weight":
                continue
            assert_grad(grads[name], g1[k], 5e-4, name)   # fp32 sums over 131 072 samples in a different order
            n += 1
    assert n >= 19
    _check_planes(g1, grads, 5e-4)


@pytest.fixture(scope="module")
def model1(g1):
    from helpers import field_cfg
    from nvfi_amd.models import NVFi
    meta = {k[5:]: (g1[k].item() if g1[k].ndim == 0 else g1[k]) for k in g1.files if k.startswith("meta:")}
    cfg = field_cfg(meta)
    aabb = torch.tensor(np.asarray(meta["aabb"]).reshape(2, 3), dtype=torch.float32)
    m = NVFi(cfg, "cpu", aabb, [int(x) for x in meta["gridSize"]], [float(meta["near"]), float(meta["far"])])
    own = m.state_dict()
    for k in g1.files:
        if k.startswith("sd:") and k[3:] in own:
            own[k[3:]].copy_(torch.from_numpy(np.ascontiguousarray(g1[k])))
    m = m.to("cuda")
    assert m.nvfi.nSamples == 32
    return m


@pytest.mark.gpu
def test_gpu_cfg1_matches_reference(g1, model1):
    from helpers import named_grads
    from nvfi_amd.models import Renderer, Ray
    ren = Renderer(model1, 0, 0, 2048)          # ray_chunk 2048: the 4096 rays are rendered as two chunks, as in the reference
    o, d = torch.from_numpy(g1["rays_o"]).cuda(), torch.from_numpy(g1["rays_d"]).cuda()
    for name, t in (("nonkey", T_NONKEY), ("key", T_KEY)):
        out = ren.render(t, Ray(o, d, 0, 1), white_background=True, mode="test")
        for i, nm in enumerate(("rgb", "depth", "acc")):     # the 1e-4 relative contract; rays in the threshold-flip band are counted
            assert_contract(out[i].cpu().numpy(), g1[f"eval_{name}:{nm}"], nm, label=f"hip cfg1 eval_{name}")
    model1.zero_grad(set_to_none=True)
    torch.manual_seed(21)                        # same CPU-generator stream as the reference: one jitter draw per chunk
    out = ren.render(T_NONKEY, Ray(o, d, 0, 1), white_background=True, mode="train")
    for i, nm in enumerate(("rgb", "depth", "acc")):
        assert_contract(out[i].detach().cpu().numpy(), g1[f"train:{nm}"], nm, label="hip cfg1 train")
    loss = torch.nn.functional.mse_loss(out[0], torch.from_numpy(g1["train:target"]).cuda())
    np.testing.assert_allclose(loss.item(), float(g1["train:loss"]), rtol=1e-4)
    loss.backward()
    grads = named_grads(model1)
    n = 0
    for k in g1.files:
        if k.startswith("train:grad:nvfi."):
            name = k[len("train:grad:nvfi."):]
            if name == "basis_mat_density.weight":
                continue
            assert_grad(grads[name], g1[k], 5e-4, name)
            n += 1
    assert n >= 19
    _check_planes(g1, grads, 5e-4)
    model1.zero_grad(set_to_none=True)
