"""Helper of test_deterministic_mode (run in a subprocess with NVFI_DETERMINISTIC=0|1, read once per process): the same train-mode
render + backward three times; prints a JSON line with, per parameter family, whether the three gradients are bit-identical."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_model, named_grads      # noqa: E402
from conftest import GOLD, relerr                # noqa: E402


def main():
    gold = np.load(os.path.join(GOLD, "hotpath.npz"))
    model, meta = make_model("A")
    f = model.nvfi
    f.train()
    rng = np.random.default_rng(3)
    # many rays through the same texels: the contention that makes float atomics order-dependent
    R = 2048
    o = np.tile(np.array([[2.6, -2.2, 2.4]], np.float32), (R, 1))
    tgt = rng.uniform(-0.5, 0.5, (R, 3)).astype(np.float32)
    d = (tgt - o) / np.linalg.norm(tgt - o, axis=1, keepdims=True)
    u = rng.uniform(0, 1, (R, 1)).astype(np.float32)
    target = torch.from_numpy(rng.uniform(0, 1, (R, 3)).astype(np.float32)).cuda()
    runs = []
    for k in range(3):
        model.zero_grad(set_to_none=True)
        f.jitter_override = torch.from_numpy(u.copy())
        out = f(19.0 / 60.0, torch.from_numpy(o).cuda(), torch.from_numpy(d.astype(np.float32)).cuda(), True)
        f.jitter_override = None
        (torch.nn.functional.mse_loss(out[0], target) + 0.01 * out[1].mean()).backward()
        runs.append({k2: v.copy() for k2, v in named_grads(model).items() if v is not None})
    same = {k: bool(all(np.array_equal(runs[0][k], r[k]) for r in runs[1:])) for k in runs[0]}
    # the PDE term: its fused adjoint kernel hands the acceleration net's tiles out through a queue unless the mode asks for the static share
    pts = torch.from_numpy(rng.uniform(-1.2, 1.2, (16384, 3)).astype(np.float32)).cuda()
    tt = torch.from_numpy(rng.uniform(0, 1, (16384, 1)).astype(np.float32)).cuda()
    pruns = []
    for k in range(3):
        model.zero_grad(set_to_none=True)
        model.get_vel_loss(points=pts, t=tt).backward()
        pruns.append({"pde:" + k2: v.copy() for k2, v in named_grads(model).items() if v is not None and "vel_net" in k2})
    same.update({k: bool(all(np.array_equal(pruns[0][k], r[k]) for r in pruns[1:])) for k in pruns[0]})
    # and against the golden gradients of the standard fixture (the mode must not change the numbers beyond tolerance)
    model.zero_grad(set_to_none=True)
    torch.manual_seed(21)
    from nvfi_amd.models import Renderer, Ray
    ren = Renderer(model, 0, 0, 2048)
    out = ren.render(float(gold["A:train_nonkey:t"]), Ray(torch.from_numpy(gold["A:rays_o"]).cuda(), torch.from_numpy(gold["A:rays_d"]).cuda(), 0, 1),
                     white_background=True, mode="train")
    tg, gw = torch.from_numpy(gold["A:train_nonkey:target"]).cuda(), torch.from_numpy(gold["A:train_nonkey:gw"]).cuda()
    (torch.nn.functional.mse_loss(out[0], tg) + 0.01 * out[1].mean() + 0.02 * (out[2] ** 2).mean() + (out[3] * gw).sum()).backward()
    g = named_grads(model)
    worst = 0.0
    for k in gold.files:
        pre = "A:train_nonkey:grad:nvfi."
        if k.startswith(pre) and gold[k].size and k[len(pre):] != "basis_mat_density.weight":
            worst = max(worst, relerr(g[k[len(pre):]], gold[k]))
    print(json.dumps({"bit_identical": same, "worst_relerr_vs_reference": worst}))


if __name__ == "__main__":
    main()
