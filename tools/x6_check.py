#!/usr/bin/env python
"""x6 (vel_x6.hip) against the fp32 MFMA kernels and a float64 restatement of integrate_pos (tensorf_keyframe.py:575-611 + velocity_field.py):
accuracy of both device paths against float64, and their speed.   python tools/x6_check.py [N]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def f64_net(f):
    from nvfi_amd.models.velocity_field import VelBasis
    lins = VelBasis.linears(f.vel_net.weight_net)
    return [(l.weight.detach().double().cpu().numpy(), l.bias.detach().double().cpu().numpy()) for l in lins]


def vel64(net, x, t, gate_lo, gate_hi):
    """gated VelBasis.get_vel in float64: x (N,3) normalised, t (N,)"""
    q = np.concatenate([x, t[:, None]], 1)
    enc = [q]
    for k in range(3):
        enc += [np.sin(q * 2.0 ** k), np.cos(q * 2.0 ** k)]
    h = np.concatenate(enc, 1)
    for i, (W, b) in enumerate(net):
        h = h @ W.T + b
        if i < 5:
            h = h / (1.0 + np.exp(-h))
    w = h
    v = np.stack([w[:, 0] - w[:, 4] * x[:, 2] + w[:, 5] * x[:, 1],
                  w[:, 1] + w[:, 3] * x[:, 2] - w[:, 5] * x[:, 0],
                  w[:, 2] - w[:, 3] * x[:, 1] + w[:, 4] * x[:, 0]], 1)
    out = ((x < gate_lo) | (x > gate_hi)).any(1)
    v[out] = 0.0
    return v


def integrate64(f, x0, t, base):
    net = f64_net(f)
    gsur, lo, hi = f._gate()
    lo, hi = np.array(lo, np.float64), np.array(hi, np.float64)
    dtm = 0.5 * f.tmax / (f.num_keyframes - 1)
    x = x0.astype(np.float64).copy(); tc = t.astype(np.float64).copy(); off = tc - base.astype(np.float64)
    while np.any(np.abs(off) > 0):
        live = np.abs(off) > 0
        m = np.minimum(np.abs(off), dtm)
        dt = np.where(off > 0, m, -m) * live
        v1 = vel64(net, x, tc, lo, hi)
        pm = x - 0.5 * dt[:, None] * v1
        v2 = vel64(net, pm, tc - 0.5 * dt, lo, hi)
        xn = x - dt[:, None] * v2
        if gsur:        # VelocityAABBSur: a step that would leave the surround box is rejected (tensorf_keyframe.py:603-605)
            live = live & ~((xn < lo) | (xn > hi)).any(1)
        x = np.where(live[:, None], xn, x)
        off = off - dt; tc = tc - dt
        off[np.abs(off) < 1e-12] = 0.0
    return x


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
    dev = torch.device("cuda", 0)
    m = bench.build_scene(dev, 199, 128, True)
    f = m.nvfi
    f.eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    res = {}
    ts = f.tmax / (f.num_keyframes - 1)
    for nsteps in (1, 4, 10):
        x = (torch.rand(N, 3, device=dev, generator=g) * 1.6 - 0.8)
        tt = torch.full((N, 1), float(np.float32(ts * 0.5 * nsteps * 0.999)), device=dev)
        base = torch.zeros(N, 1, device=dev)
        ref = integrate64(f, x.cpu().numpy(), tt.cpu().numpy()[:, 0], base.cpu().numpy()[:, 0])
        row = {}
        for name, mode in (("fp32", "fp32"), ("x6", 3), ("split16", 2), ("fp16", True)):
            f.vel_fp16 = mode
            with torch.no_grad():
                out = f.integrate_pos(x.clone(), tt.clone(), base)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    f.integrate_pos(x, tt, base)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 5 * 1e3
            e = np.abs(out.cpu().numpy().astype(np.float64) - ref)
            if name == "x6" and os.environ.get("X6_DEBUG"):
                em = e.max(1)
                idx = np.argsort(-em)[:8]
                print("x6 percentiles", {p: float(np.percentile(em, p)) for p in (50, 90, 99, 99.9, 99.99)}, "frac > 1e-6", float((em > 1e-6).mean()))
                for i in idx:
                    print("  worst", i, i % 32, (i // 32) % 2, em[i], x.cpu().numpy()[i], ref[i] - x.cpu().numpy()[i])
            row[name] = dict(max_abs_err=float(e.max()), rms_err=float(np.sqrt((e ** 2).mean())), ms=ms, Mevals_per_s=2 * nsteps * N / ms / 1e3)
        f.vel_fp16 = False
        res[f"{nsteps}_rk2_steps"] = row
        print(nsteps, {k: (f"{v['max_abs_err']:.2e}", f"{v['rms_err']:.2e}", f"{v['ms']:.3f} ms") for k, v in row.items()}, flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
