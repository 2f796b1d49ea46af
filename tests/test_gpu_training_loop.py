"""The reference's training loop, three iterations, on the HIP path - against the reference itself.

tests/golden/make_golden_r2.py ran the body of train_nvfi.py:139-249 (--static_dynamic) with the REFERENCE on field A: per iteration a
train-mode render at a non-keyframe time and one at a keyframe time (MSE against fixed targets), `density_L1()` / `TV_loss_density(reg)` /
`TV_loss_app(reg)` with their decaying weights, `get_vel_loss`, `loss.backward()`, `torch.optim.Adam(get_optparam_groups(0.02, 1e-3),
betas=(0.9, 0.99)).step()`, lr decay - and recorded its CPU-generator draws, the loss of every iteration and every parameter after the
third step.  Here the same lines run on the `models` alias (plain autograd, torch.optim.Adam): iteration 2 and 3 render a field that
this path's own gradients and optimiser steps produced, so the loss trajectory and the final parameters check the whole drop-in loop."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from helpers import make_model

pytestmark = pytest.mark.gpu


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("driver", ["dropin", "dropin-side-streams", "dropin-pure-autograd", "fused"])
def test_three_training_iterations_match_the_reference(driver):
    import models                                  # the drop-in alias of the reference's package
    from nvfi_amd.utils import TVLoss
    g2 = np.load(os.path.join(GOLD, "r2.npz"))
    gold = np.load(os.path.join(GOLD, "hotpath.npz"))
    nvfi, meta = make_model("A")
    f = nvfi.nvfi
    o, d = _cu(gold["A:rays_o"]), _cu(gold["A:rays_d"])
    tgt1, tgt2 = _cu(g2["A:loop:target1"]), _cu(g2["A:loop:target2"])
    renderer = models.Renderer(nvfi, 0, 0, 2048)
    f.requires_grad_(True)
    groups = nvfi.get_optparam_groups(0.02, 1e-3)
    if driver == "fused":      # bench.py:Step's machinery: gradients in place in one flat buffer, fused regularisers + PDE backward, one-launch Adam
        from nvfi_amd.dist import GradBucket
        from nvfi_amd.optim import Adam
        groups = [dict(params=list(g["params"]), lr=g["lr"]) for g in groups]
        bucket = GradBucket([p for g in groups for p in g["params"]])
        optimizer = Adam(groups, betas=(0.9, 0.99))
        f.accumulate_grads_inplace = True
    else:
        # "dropin": a plain autograd driver with the opt-in gradient arena (what tools/run_reference_driver.py and bench.py --mode dropin switch on);
        # "dropin-pure-autograd": the library default - gradients handed back to the autograd engine, everything on the caller's stream
        # "dropin-side-streams": + the opt-in auto_overlap (train-mode calls on the field's own streams, joined at the end of backward)
        if driver == "dropin-pure-autograd":
            assert f.accumulate_grads_inplace is False and not f.auto_overlap      # the library default
        else:
            f.accumulate_grads_inplace = "arena"                                     # what tools/run_reference_driver.py / bench.py --mode dropin switch on
            f.auto_overlap = driver == "dropin-side-streams"
        driver = "dropin"
        optimizer = torch.optim.Adam(groups, betas=(0.9, 0.99))
    lr_factor = 0.1 ** (1 / 30000)
    L1w, tvd, tva, vw = 8e-4, 1.0, 1.0, 1.0
    tvreg = TVLoss()
    ts = f.tmax / (f.num_keyframes - 1)
    losses = []
    for it in range(3):
        nvfi.train(); renderer.train()
        if driver == "fused":
            bucket.zero()
        f.jitter_override = torch.from_numpy(g2[f"A:loop:{it}:u1"].copy())
        rgb_map = renderer.render(19.0 / 60.0, models.Ray(o, d, 0, 1), white_background=True, mode="train")[0]
        loss = torch.nn.functional.mse_loss(rgb_map[..., :3], tgt1)
        f.jitter_override = torch.from_numpy(g2[f"A:loop:{it}:u2"].copy())
        rgb_map = renderer.render(ts * 1, models.Ray(o, d, 0, 1), white_background=True, mode="train")[0]
        f.jitter_override = None
        loss = loss + 1.0 * torch.nn.functional.mse_loss(rgb_map[..., :3], tgt2)
        L1w *= lr_factor; tvd *= lr_factor; tva *= lr_factor; vw *= lr_factor
        pts, tt = _cu(g2[f"A:loop:{it}:points"]), _cu(g2[f"A:loop:{it}:t"])
        if driver == "dropin":
            loss = loss + L1w * f.density_L1()
            loss = loss + f.TV_loss_density(tvreg) * tvd
            loss = loss + f.TV_loss_app(tvreg) * tva
            loss_vel = nvfi.get_vel_loss(points=pts, t=tt)
            if loss_vel > 0:
                loss = loss + vw * loss_vel
            optimizer.zero_grad(set_to_none=True)
            loss.backward()
            optimizer.step()
            total = loss.item()
        else:
            nvfi.vel_loss_weight = vw
            lv = nvfi.get_vel_loss(points=pts, t=tt)             # value + gradient (weighted) in one pass, no host wait
            loss.backward()
            regs = f.regularizers_backward_(L1w, tvd, tva)
            optimizer.step(zero_grad=True)
            total = float(loss.detach() + L1w * regs[0] + tvd * regs[1] + tva * regs[2] + vw * lv)
        losses.append(total)
        for g in optimizer.param_groups:
            g["lr"] = g["lr"] * lr_factor
        np.testing.assert_allclose(total, float(g2[f"A:loop:{it}:loss"].reshape(-1)[0]), rtol=2e-4, err_msg=f"iteration {it}")
    nvfi.vel_loss_weight = None
    assert losses[2] < losses[1] < losses[0]
    # parameters after three Adam steps.  Adam normalises every element's step by its own gradient history, so an element whose gradient
    # is a near-cancelling sum can move differently (its relative error is not bounded by the max-norm gradient parity): the bulk must agree
    # to 1e-3 of the distance moved, and nothing may differ by more than the distance three steps can move an element.
    start = {k[3:]: v for k, v in np.load(os.path.join(GOLD, "field_A.npz")).items() if k.startswith("sd:")}
    sd = nvfi.state_dict()
    n = 0
    for k in g2.files:
        if not k.startswith("A:loop:final:"):
            continue
        name = k[len("A:loop:final:"):]
        ref, got = g2[k].astype(np.float64), sd[name].detach().cpu().contiguous().numpy().astype(np.float64)
        if not np.issubdtype(g2[k].dtype, np.floating) or ref.size < 2 or "frequency_bands" in name or name.endswith("aabb"):
            continue
        moved = np.abs(ref - start[name].astype(np.float64))
        lr = 0.02 if "plane" in name else 1e-3
        err = np.abs(got - ref)
        assert err.max() <= 3.1 * lr, (name, err.max())
        if moved.max() > 0:
            frac_bad = np.mean(err > 1e-3 * moved.max() + 1e-7)
            assert frac_bad < 0.02, (name, frac_bad, err.max(), moved.max())
            n += 1
    assert n >= 40, n
