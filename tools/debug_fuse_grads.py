"""debug helper: velocity-net gradients of the golden training renders with the fused RK2 adjoint in its fp32 and x6 forms, tensor by tensor"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLD, maxrel, rel_l2
from helpers import make_model, named_grads
from nvfi_amd.models import Renderer, Ray
gold = np.load(os.path.join(GOLD, "hotpath.npz"))
for kind in ("A", "B"):
    for name in ("train_nonkey", "train_extrap"):
        model, meta = make_model(kind)
        t = float(gold[f"{kind}:{name}:t"])
        torch.manual_seed(21)
        ren = Renderer(model, 0, 0, 2048)
        o, d = torch.from_numpy(gold[f"{kind}:rays_o"]).cuda(), torch.from_numpy(gold[f"{kind}:rays_d"]).cuda()
        out = ren.render(t, Ray(o, d, 0, 1), white_background=bool(meta["white_background"]), mode="train")
        target, gw = torch.from_numpy(gold[f"{kind}:{name}:target"]).cuda(), torch.from_numpy(gold[f"{kind}:{name}:gw"]).cuda()
        loss = torch.nn.functional.mse_loss(out[0], target) + 0.01 * out[1].mean() + 0.02 * (out[2] ** 2).mean() + (out[3] * gw).sum()
        loss.backward()
        g = named_grads(model)
        for k in sorted(gold.files):
            pre = f"{kind}:{name}:grad:nvfi."
            if k.startswith(pre) and "vel_net.weight_net" in k:
                pn = k[len(pre):]
                print(kind, name, pn, "maxrel %.2e l2 %.2e" % (maxrel(g[pn], gold[k]), rel_l2(g[pn], gold[k])), "| peak", float(np.abs(gold[k]).max()))
                if pn.endswith("5.0.bias") and kind == "A" and name == "train_nonkey":
                    np.set_printoptions(precision=3, linewidth=220)
                    print("got/ref:", (g[pn] / (gold[k] + 1e-30))[:64])
