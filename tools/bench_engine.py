#!/usr/bin/env python
"""Engine micro-benchmarks on one GPU: vel_eval (one VelBasis evaluation per point, both nets) and integrate_pos with a
fixed number of RK2 steps.  Prints TFLOP/s of MLP GEMM work (139 776 FLOP per net evaluation) against the fp32 MFMA peak."""
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_scene   # noqa: E402


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    m = build_scene(torch.device("cuda:0"), G=64)
    f = m.nvfi
    N = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(1)
    xt = torch.rand(N, 4, device="cuda", generator=g) * 1.8 - 0.9
    xt[:, 3] = torch.rand(N, device="cuda", generator=g) * 0.7
    ms = timeit(lambda: f._vel_eval(xt, False))
    print(f"vel_eval ungated (2 nets)   N={N}: {ms:.3f} ms  {2 * N * 139776 / ms / 1e9:.1f} TFLOP/s")
    ms = timeit(lambda: f._vel_eval(xt, True))
    print(f"vel_eval gated (1 net)      N={N}: {ms:.3f} ms  {N * 139776 / ms / 1e9:.1f} TFLOP/s")
    ts = f.tmax / (f.num_keyframes - 1)
    for nsteps in (1, 2, 4):
        base = torch.zeros(N, device="cuda")
        t = torch.full((N,), float(ts * 0.5 * nsteps * 0.999), device="cuda")
        x = xt[:, :3].contiguous()
        ms = timeit(lambda: f.integrate_pos(x, t, base))
        print(f"integrate_pos {nsteps} step(s)    N={N}: {ms:.3f} ms  {2 * nsteps * N * 139776 / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
