#!/usr/bin/env python
"""Experiment: does a whole bench step (three streams, autograd, ctypes launches) survive hipGraph capture, and what does a replay cost
against the eager step?  Fixed frame times (the graph freezes host scalars) - a timing probe, not a training loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
model = bench.build_scene(dev, 199, 128, True)
if workload == "cfg2":
    model.nvfi.use_vel = False
step = bench.Step(model, dev, 2048, 262144, 1, 0, workload)
model.nvfi.jitter_override = torch.rand(2048, device=dev)


class FixedRng:
    def __init__(self): self.k = 0
    def integers(self, lo, hi):
        self.k += 1
        return 19 if hi == 46 else 5


def timeit(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(12):
    step()
print("eager (random t)   ms/step", timeit(step))
step.rng = FixedRng()
for _ in range(3):
    step()
print("eager (fixed t)    ms/step", timeit(step))
streams = step.streams
step.streams = None
for _ in range(3):
    step()
print("eager one stream   ms/step", timeit(step))
step.streams = streams

g = torch.cuda.CUDAGraph()
g.register_generator_state(step.gen)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, stream=side):
        step()
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    print("graph replay (3 streams captured) ms/step", timeit(g.replay))
except Exception as e:
    print("capture with streams failed:", repr(e)[:400])

step.streams = None
g2 = torch.cuda.CUDAGraph()
g2.register_generator_state(step.gen)
try:
    with torch.cuda.graph(g2, stream=side):
        step()
    torch.cuda.synchronize()
    for _ in range(5):
        g2.replay()
    print("graph replay (one stream captured) ms/step", timeit(g2.replay))
except Exception as e:
    print("capture one stream failed:", repr(e)[:400])
