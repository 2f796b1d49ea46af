"""shadingMode "SH" inside the render call (a-17 selectable; reference models/tensorf_base.py:196-197, tensorf_model_utils.py:292-296):
app_dim = 27, `basis_mat` 48 -> 27, rgb = relu(sum_k SH_k(viewdir) feat[9c + k] + 0.5), no render MLP.  Golden vectors from the reference
(tests/golden/make_golden_r2.py, field "D" = the bat-like field A with that shading): eval renders at a non-key and a key time and a
train-mode render with every parameter gradient.  CPU: the oracle against them; GPU: the HIP path."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, assert_grad, relerr
from helpers import assert_contract, load_meta

T_NONKEY, T_KEY = 19.0 / 60.0, 0.25


@pytest.fixture(scope="module")
def g2():
    return np.load(os.path.join(GOLD, "r2.npz"))


def _meta():
    meta, _ = load_meta("A")
    meta = dict(meta)
    meta["shadingMode"] = "SH"
    return meta


def _check_grads(g2, grads, tol=5e-4):
    n = 0
    for k in g2.files:
        if not k.startswith("D:train:grad:nvfi."):
            continue
        name = k[len("D:train:grad:nvfi."):]
        assert_grad(grads[name], g2[k], tol, name)
        n += 1
    assert n == 25, n       # 12 planes + basis_mat + 12 velocity tensors


def test_oracle_sh_mode_matches_reference(g2):
    from oracle import oracle as orc
    params = {k[len("D:sd:nvfi."):]: g2[k] for k in g2.files if k.startswith("D:sd:nvfi.")}
    assert params["basis_mat.weight"].shape == (27, 48) and "renderModule.mlp.0.weight" not in params
    fs = orc.FieldSpec(params, _meta())
    gold = np.load(os.path.join(GOLD, "hotpath.npz"))
    o, d = gold["A:rays_o"], gold["A:rays_d"]
    for name, t in (("nonkey", T_NONKEY), ("key", T_KEY)):
        r = orc.render(fs, o, d, t, train=False, white_bg=True)
        for nm, got in (("rgb", r.rgb), ("depth", r.depth), ("acc", r.acc)):
            assert_contract(got, g2[f"D:render_{name}:{nm}"], nm, label=f"oracle D:{name}")
    r = orc.render(fs, o, d, T_NONKEY, u=g2["D:train:u"], train=True, white_bg=True, keep_ctx=True)
    assert_contract(r.rgb, g2["D:train:rgb"], "rgb", label="oracle D:train")
    tgt = g2["D:train:target"]
    R = o.shape[0]
    grads = r.backward(fs, g_rgb=(2.0 * (r.rgb - tgt) / tgt.size).astype(np.float32), g_depth=np.full(R, 0.01 / R, np.float32))
    r.free()
    _check_grads(g2, grads)


@pytest.mark.gpu
def test_gpu_sh_mode_matches_reference(g2):
    from helpers import field_cfg, named_grads
    from nvfi_amd.models import NVFi, Renderer, Ray
    meta = _meta()
    cfg = field_cfg(meta)
    cfg.nvfi.shadingMode = "SH"
    cfg.nvfi.app_dim = 27
    aabb = torch.tensor(np.asarray(meta["aabb"]).reshape(2, 3), dtype=torch.float32)
    m = NVFi(cfg, "cpu", aabb, [int(g) for g in meta["gridSize"]], [float(meta["near"]), float(meta["far"])])
    own = m.state_dict()
    assert not any(k.startswith("nvfi.renderModule") for k in own)          # SHRender is a function: no parameters, as in the reference
    n = 0
    for k in g2.files:
        if k.startswith("D:sd:") and k[5:] in own:
            own[k[5:]].copy_(torch.from_numpy(np.ascontiguousarray(g2[k]))); n += 1
    assert n >= 38, n
    m = m.to("cuda")
    assert len(m.get_optparam_groups(0.02, 1e-3)) == 7                       # 4 plane groups, basis_mat, basis_mat_density, vel (no renderModule group)
    gold = np.load(os.path.join(GOLD, "hotpath.npz"))
    o, d = torch.from_numpy(gold["A:rays_o"]).cuda(), torch.from_numpy(gold["A:rays_d"]).cuda()
    ren = Renderer(m, 0, 0, 2048)
    for name, t in (("nonkey", T_NONKEY), ("key", T_KEY)):
        out = ren.render(t, Ray(o, d, 0, 1), white_background=True, mode="test")
        for i, nm in enumerate(("rgb", "depth", "acc")):
            assert_contract(out[i].cpu().numpy(), g2[f"D:render_{name}:{nm}"], nm, label=f"hip D:{name}")
    m.zero_grad(set_to_none=True)
    torch.manual_seed(21)
    out = ren.render(T_NONKEY, Ray(o, d, 0, 1), white_background=True, mode="train")
    assert_contract(out[0].detach().cpu().numpy(), g2["D:train:rgb"], "rgb", label="hip D:train")
    loss = torch.nn.functional.mse_loss(out[0], torch.from_numpy(g2["D:train:target"]).cuda()) + 0.01 * out[1].mean()
    np.testing.assert_allclose(loss.item(), float(g2["D:train:loss"].reshape(-1)[0]), rtol=1e-4)
    loss.backward()
    _check_grads(g2, named_grads(m))
