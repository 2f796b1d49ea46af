#!/usr/bin/env python
"""Micro-benchmark: density / appearance gathers (nvfi_density_at, nvfi_app_at) on random vs spatially sorted points."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
m = bench.build_scene(dev)
f = m.nvfi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
g = torch.Generator(device="cpu").manual_seed(1)
x = (torch.rand(N, 3, generator=g) * 2 - 1)
tn = torch.full((N, 1), 0.2)
xyzt = torch.cat([x, tn], 1).to(dev)

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def cellkey(x, G=199, T=16, morton=False):
    c = ((x + 1) * 0.5 * (G - 1)).floor().clamp(0, G - 1).long() // T
    if not morton:
        return (c[:, 2] * 64 + c[:, 1]) * 64 + c[:, 0]
    k = torch.zeros_like(c[:, 0])
    for b in range(4):
        for a in range(3):
            k |= ((c[:, a] >> b) & 1) << (3 * b + a)
    return k

view = torch.randn(N, 3, device=dev)
for name, order in [("random", None), ("cell16-lex", cellkey(x).argsort()), ("cell16-morton", cellkey(x, morton=True).argsort()), ("cell8-morton", cellkey(x, T=8, morton=True).argsort()), ("full-sort-x", x[:, 0].argsort())]:
    p = xyzt if order is None else xyzt[order.to(dev)].contiguous()
    v = view if order is None else view[order.to(dev)].contiguous()
    td = timeit(lambda: f.compute_densityfeature(p))
    ta = timeit(lambda: f.app_at(p, v))
    print(f"{name:16s} N={N} density_at {td*1e3:8.1f} us   app_at {ta*1e3:8.1f} us", flush=True)
