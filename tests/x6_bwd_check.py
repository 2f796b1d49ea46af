"""helper of tests/test_gpu_x6_bwd.py (subprocess: NVFI_FUSE_X6 / NVFI_PDE_JET_X6 are read once per process): velocity-net gradients of a
4096-ray training render on the 199^3 bench field and of the PDE term on 65 536 collocation points; saves them + the loss values"""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import bench
from helpers import named_grads
dev = torch.device("cuda", 0)
m = bench.build_scene(dev, 199, 128, True)
f = m.nvfi
f.train()
with torch.no_grad():                       # a velocity field that moves (the init is ~0: gradients would be rounding noise)
    last = f.vel_net.weight_net[-1][0]
    last.weight.mul_(6.0); last.bias.copy_(torch.tensor([0.5, -0.3, 0.2, 0.1, -0.2, 0.4], device=dev))
ob, db = bench.camera_bundle(dev)
sel = torch.randperm(ob.shape[0], generator=torch.Generator().manual_seed(3))[:4096].to(dev)
o, d = ob[sel].contiguous(), db[sel].contiguous()
gg = torch.Generator().manual_seed(5)
jit = torch.rand(4096, 1, generator=gg)
target = torch.rand(4096, 3, generator=gg).to(dev)
out = {}
runs = []
for rep in range(3):
    m.zero_grad(set_to_none=True)
    f.jitter_override = jit.clone()
    r = f(19.0 / 60.0, o, d, True)
    f.jitter_override = None
    loss = torch.nn.functional.mse_loss(r[0], target) + 0.01 * r[1].mean()
    loss.backward()
    runs.append({k: v.copy() for k, v in named_grads(m).items() if v is not None and k.startswith("vel_net.weight_net")})
out["render:loss"] = np.float64(loss.detach().cpu())
for k, v in runs[0].items():
    out["render:" + k] = v
out["render:repeat_identical"] = np.array(all(np.array_equal(runs[0][k], r[k]) for r in runs[1:] for k in runs[0]))
out["render:evals"] = f.last_counters.cpu().numpy()
rng = np.random.default_rng(9)
ab = f.aabb.detach().cpu().numpy()
pts = torch.from_numpy((rng.uniform(0, 1, (65536, 3)) * (ab[1] - ab[0]) + ab[0]).astype(np.float32)).to(dev)
tt = torch.from_numpy(rng.uniform(0, 1, (65536, 1)).astype(np.float32)).to(dev)
m.zero_grad(set_to_none=True)
lv = m.get_vel_loss(points=pts, t=tt)
lv.backward()
out["pde:loss"] = np.float64(lv.detach().cpu())
out["pde:kept"] = np.int64(f.last_pde_n_kept)
for k, v in named_grads(m).items():
    if v is not None and k.startswith("vel_net."):
        out["pde:" + k] = v.copy()
np.savez(sys.argv[1], **out)
print("render loss", float(out["render:loss"]), "repeat identical:", bool(out["render:repeat_identical"]), "| pde loss", float(out["pde:loss"]), "kept", int(out["pde:kept"]))
