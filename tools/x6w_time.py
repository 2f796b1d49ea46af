#!/usr/bin/env python
"""Timing of the per-point x6 kernels on a uniform workload (N points x `steps` RK2 steps = 2 * steps evaluations each): ms per call and SIMD cycles per
tile evaluation at 2.4 GHz.   python tools/x6w_time.py [N] [steps]      (used with the X6W_PROBE_* build flags of vel_x6w.hip: results are then garbage)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m = bench.build_scene(torch.device("cuda"), 199, 128, True)
f = m.nvfi
f.eval()
g = torch.Generator(device="cuda").manual_seed(1)
ts = f.tmax / (f.num_keyframes - 1)
x = torch.rand(N, 3, device="cuda", generator=g) * 1.6 - 0.8
tt = torch.full((N, 1), float(np.float32(ts * 0.5 * steps * 0.999)), device="cuda")
base = torch.zeros(N, 1, device="cuda")
f.vel_fp16 = 3
with torch.no_grad():
    for _ in range(3):
        f.integrate_pos(x.clone(), tt.clone(), base)
    xs = [x.clone() for _ in range(10)]
    tts = [tt.clone() for _ in range(10)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(10):
        f.integrate_pos(xs[k], tts[k], base)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
evals = N / 32 * 2 * steps
print(f"N {N} steps {steps}: {ms:.3f} ms per call; {ms * 1e-3 * 2.4e9 * 1024 / evals:.0f} SIMD cycles per tile evaluation (2.4 GHz, 1024 SIMDs); "
      f"{evals * 32 * 139776 / (ms * 1e-3) / 1e12:.1f} algorithmic TFLOP/s")
