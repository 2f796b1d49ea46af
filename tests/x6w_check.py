"""helper of tests/test_gpu_x6.py: integrate_pos of 262 221 points (0..9.5 half-keyframe steps) through the x6 kernel the environment selects; saves the positions"""
import os, sys, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench
dev = torch.device("cuda", 0)
m = bench.build_scene(dev, 199, 128, True); f = m.nvfi; f.eval()
g = torch.Generator(device="cuda").manual_seed(1)
N = (1 << 18) + 77
ts = f.tmax / (f.num_keyframes - 1)
x = (torch.rand(N, 3, device=dev, generator=g) * 1.6 - 0.8)
tt = (torch.rand(N, 1, device=dev, generator=g) * ts * 0.5 * 9.5)
base = torch.zeros(N, 1, device=dev)
f.vel_fp16 = 3
with torch.no_grad():
    outs = [f.integrate_pos(x.clone(), tt.clone(), base).clone() for _ in range(3)]
torch.cuda.synchronize()
print("repeat identical:", all(torch.equal(o, outs[0]) for o in outs[1:]))
np.save(sys.argv[1], outs[0].cpu().numpy())
# the render warp (uniform schedule): an eval render of 4097 rays on field A, and a training render's stash consumers (gradients) on the same rays
sys.path.insert(0, os.path.join(root, "tests"))
from helpers import make_model
model, meta = make_model("A"); fa = model.nvfi; fa.eval()
gg = torch.Generator().manual_seed(1)
o = torch.tensor([[2.6, -2.2, 2.4]]).expand(4097, 3).contiguous()
d = (torch.rand(4097, 3, generator=gg) - 0.5) * 3.0 - o
d = (d / d.norm(dim=-1, keepdim=True)).contiguous().cuda(); o = o.cuda()
with torch.no_grad():
    r = fa(0.41, o, d, True)
fa.train(); model.zero_grad(set_to_none=True)
fa.jitter_override = torch.rand(4097, 1, generator=gg)
out = fa(0.41, o, d, True)
fa.jitter_override = None
out[0].square().mean().backward()
gv = fa.vel_net.weight_net[4][0].weight.grad
np.savez(sys.argv[1] + ".render.npz", rgb=r[0].cpu().numpy(), depth=r[1].cpu().numpy(), train_rgb=out[0].detach().cpu().numpy(), g=gv.cpu().numpy())
