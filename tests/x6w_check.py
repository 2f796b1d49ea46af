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
