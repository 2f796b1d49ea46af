"""BASELINE.json configs[1] - radiance-only (`use_vel: False`): every plane lookup takes the CONTINUOUS normalised frame time
(reference models/tensorf_keyframe.py:703-704), a true bilinear interpolation in (space, time) whose gradient lands on two time rows.
Golden vectors come from the reference (tests/golden/make_golden_r2.py, field "C").  CPU: the oracle against them; GPU: the HIP path
(Renderer -> ctypes -> C ABI) against them, and against the oracle on a 256-ray slice at the full configs[1] size (199^3, 128
samples per ray)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, assert_grad, relerr
from helpers import assert_contract

PATH = os.path.join(GOLD, "r2.npz")
EVAL = ["frac", "row", "zero", "beyond", "last"]


@pytest.fixture(scope="module")
def g2():
    return np.load(PATH)


@pytest.fixture(scope="module")
def fsC(g2):
    from oracle import oracle as orc
    return orc.FieldSpec.from_npz(g2, prefix="C:")


def _loss_grads_np(rgb, depth, acc, weight, target, gw):
    R = rgb.shape[0]
    return (2.0 * (rgb - target) / target.size).astype(np.float32), np.full(R, 0.01 / R, np.float32), (0.04 * acc / R).astype(np.float32), gw.astype(np.float32)


@pytest.mark.parametrize("case", EVAL)
def test_oracle_cfg2_eval(g2, fsC, case):
    from oracle import oracle as orc
    r = orc.render(fsC, g2["C:rays_o"], g2["C:rays_d"], float(g2[f"C:render_{case}:t"]), train=False, white_bg=True, use_vel=False)
    for nm, got in (("rgb", r.rgb), ("depth", r.depth), ("acc", r.acc), ("weight", r.weight)):
        assert_contract(got, g2[f"C:render_{case}:{nm}"], nm, label=f"oracle C:{case}")


def test_oracle_cfg2_features(g2, fsC):
    from oracle import oracle as orc
    xyzt = g2["C:feat:xyzt"]
    np.testing.assert_allclose(orc.density_feature(fsC, xyzt), g2["C:feat:density"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(orc.app_feature(fsC, xyzt), g2["C:feat:app"], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["train_frac", "train_late"])
def test_oracle_cfg2_train_grads(g2, fsC, name):
    from oracle import oracle as orc
    t = float(g2[f"C:{name}:t"])
    r = orc.render(fsC, g2["C:rays_o"], g2["C:rays_d"], t, u=g2[f"C:{name}:u"], train=True, white_bg=True, keep_ctx=True, use_vel=False)
    for nm, got in (("rgb", r.rgb), ("depth", r.depth), ("acc", r.acc), ("weight", r.weight)):
        assert_contract(got, g2[f"C:{name}:{nm}"], nm, label=f"oracle C:{name}")
    grads = r.backward(fsC, *_loss_grads_np(r.rgb, r.depth, r.acc, r.weight, g2[f"C:{name}:target"], g2[f"C:{name}:gw"]))
    r.free()
    _check_grads(g2, name, grads, 5e-4)   # max-norm relative; fp32 sums in a different order


def _check_grads(g2, name, grads, tol):
    pre = f"C:{name}:grad:nvfi."
    n = 0
    for k in g2.files:
        if not k.startswith(pre):
            continue
        pn, ref = k[len(pre):], g2[k]
        if pn == "basis_mat_density.weight":
            continue
        assert_grad(grads[pn], ref, tol, pn)
        if "plane_time" in pn:
            # continuous time: the gradient must land on exactly the two rows that bracket the frame time (both non-zero)
            rows = np.flatnonzero(np.abs(np.asarray(grads[pn])).reshape(ref.shape)[0].sum(axis=(0, 2)) > 0)
            rows_ref = np.flatnonzero(np.abs(ref[0]).sum(axis=(0, 2)) > 0)
            assert list(rows) == list(rows_ref) and len(rows_ref) == 2, (pn, rows, rows_ref)
        n += 1
    assert n >= (6 if name == "train_late" else 19), n


# ------------------------------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def modelC(g2):
    from helpers import model_from_npz
    return model_from_npz(g2, "C:", use_vel=False)


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("case", EVAL)
def test_gpu_cfg2_eval(g2, modelC, case):
    from nvfi_amd.models import Renderer, Ray
    model, _ = modelC
    ren = Renderer(model, 0, 0, 2048)
    out = ren.render(float(g2[f"C:render_{case}:t"]), Ray(_cuda(g2["C:rays_o"]), _cuda(g2["C:rays_d"]), 0, 1), white_background=True, mode="test")
    for i, nm in enumerate(("rgb", "depth", "acc", "weight")):
        assert_contract(out[i].cpu().numpy(), g2[f"C:render_{case}:{nm}"], nm, label=f"hip C:{case}")


@pytest.mark.gpu
def test_gpu_cfg2_features(g2, modelC):
    f = modelC[0].nvfi
    d = f.compute_densityfeature(_cuda(g2["C:feat:xyzt"])).cpu().numpy()
    np.testing.assert_allclose(d, g2["C:feat:density"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["train_frac", "train_late"])
def test_gpu_cfg2_train_grads(g2, modelC, name):
    from helpers import named_grads
    from nvfi_amd.models import Renderer, Ray
    model, _ = modelC
    model.zero_grad(set_to_none=True)
    ren = Renderer(model, 0, 0, 2048)
    torch.manual_seed(21)       # the reference's CPU-generator stream: one jitter draw
    out = ren.render(float(g2[f"C:{name}:t"]), Ray(_cuda(g2["C:rays_o"]), _cuda(g2["C:rays_d"]), 0, 1), white_background=True, mode="train")
    for i, nm in enumerate(("rgb", "depth", "acc", "weight")):
        assert_contract(out[i].detach().cpu().numpy(), g2[f"C:{name}:{nm}"], nm, label=f"hip C:{name}")
    target, gw = _cuda(g2[f"C:{name}:target"]), _cuda(g2[f"C:{name}:gw"])
    loss = torch.nn.functional.mse_loss(out[0], target) + 0.01 * out[1].mean() + 0.02 * (out[2] ** 2).mean() + (out[3] * gw).sum()
    loss.backward()
    np.testing.assert_allclose(loss.item(), g2[f"C:{name}:loss"][0], rtol=1e-4)
    _check_grads(g2, name, named_grads(model), 5e-4)
    model.zero_grad(set_to_none=True)


@pytest.mark.gpu
def test_gpu_cfg2_fullsize_slice_matches_oracle():
    """configs[1] at its full size - bat 199^3, K = 16, 128 samples per ray, no velocity net - train mode forward + backward at a
    fractional frame time: a 256-ray slice against the oracle (rgb / depth / acc contract, plane and MLP gradients)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_scene, camera_bundle
    from helpers import named_grads
    from oracle import oracle as orc
    torch.manual_seed(0)
    model = build_scene("cuda", 199, 128, use_vel=False)
    f = model.nvfi
    assert not f.use_vel and f.nSamples == 128
    with torch.no_grad():       # lively time planes: the interpolation between rows must matter
        g = torch.Generator().manual_seed(3)
        for i in range(3):
            f.density_plane_time[i].add_((0.2 * torch.randn(f.density_plane_time[i].shape, generator=g)).cuda())
            f.app_plane_time[i].add_((0.3 * torch.randn(f.app_plane_time[i].shape, generator=g)).cuda())
    o, d = camera_bundle("cuda")
    rng = np.random.default_rng(11)
    idx = torch.from_numpy(rng.integers(0, o.shape[0], 256)).cuda()
    oo, dd = o[idx].contiguous(), d[idx].contiguous()
    u = rng.uniform(0, 1, (256, 1)).astype(np.float32)
    t = 22.0 / 60.0
    f.train()
    model.zero_grad(set_to_none=True)
    f.jitter_override = torch.from_numpy(u.copy())
    try:
        out = f(t, oo, dd, True)
    finally:
        f.jitter_override = None
    tg = rng.uniform(0, 1, (256, 3)).astype(np.float32)
    loss = torch.nn.functional.mse_loss(out[0], _cuda(tg)) + 0.01 * out[1].mean()
    loss.backward()
    sd = {k[len("nvfi."):]: v.detach().cpu().contiguous().numpy() for k, v in model.state_dict().items()}
    meta = dict(aabb=f.aabb.cpu().numpy(), gridSize=np.array(f.gridSize.tolist()), num_keyframes=f.num_keyframes, tmax=f.tmax,
                near=f.near_far[0], far=f.near_far[1], step_ratio=f.step_ratio, max_n_samples=f.max_n_samples,
                density_shift=f.density_shift, distance_scale=f.distance_scale, alphaMask_thres=f.alphaMask_thres,
                rayMarch_weight_thres=f.rayMarch_weight_thres, stepSize=f._step_host, nSamples=f.nSamples, use_sur=0, eps=0.03)
    fs = orc.FieldSpec(sd, meta)
    ref = orc.render(fs, oo.cpu().numpy(), dd.cpu().numpy(), t, u=u, train=True, white_bg=True, keep_ctx=True, use_vel=False)
    for i, (nm, r) in enumerate((("rgb", ref.rgb), ("depth", ref.depth), ("acc", ref.acc))):
        assert_contract(out[i].detach().cpu().numpy(), r, nm, label="hip cfg2 199^3/128")
    gref = ref.backward(fs, g_rgb=2 * (ref.rgb - tg) / tg.size, g_depth=np.full(256, 0.01 / 256, np.float32))
    ref.free()
    g = named_grads(model)
    for pn in ("density_plane_space.0", "density_plane_time.0", "density_plane_time.2", "app_plane_space.1", "app_plane_time.1",
               "basis_mat.weight", "renderModule.mlp.0.weight", "renderModule.mlp.2.weight", "renderModule.mlp.4.bias"):
        e = relerr(g[pn], gref[pn])
        assert e < 5e-4, (pn, e)
    rows = np.flatnonzero(np.abs(g["density_plane_time.0"])[0].sum(axis=(0, 2)) > 0)
    assert len(rows) == 2, rows
