#!/usr/bin/env python
"""The opt-in fp16-input inference mode of the velocity field (field.vel_fp16 / NVFI_VEL_FP16=1) against the default fp32 path on one
GPU: integrate_pos as train_segm.py uses it (64^3-like point sets, many RK2 steps), one full 800x800 eval frame of the bat scene at a
non-keyframe time (313 chunks of 2048 rays), getDenseAlpha at 128^3.  One JSON line (dtype stated per entry; never the headline)."""
import json
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402


MODES = (False, True, 2)      # fp32 MFMA (default) | one binary16 term per operand | two terms (fp32 products emulated)
NAME = {False: "fp32", True: "fp16", 2: "split16"}


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    from nvfi_amd.models import Ray, Renderer
    dev = torch.device("cuda:0")
    m = bench.build_scene(dev, 199, 128, True)
    f = m.nvfi
    f.eval()
    out = {"what": "opt-in inference modes of VelBasis (nvfi_field_desc.vel_fp16 = 1: fp16-input MFMA; = 2: fp32 products emulated by two binary16 terms per operand) vs the default fp32 MFMA path", "entries": []}
    g = torch.Generator(device="cuda").manual_seed(1)
    N = 1 << 18
    x = torch.rand(N, 3, device=dev, generator=g) * 1.6 - 0.8
    ts = f.tmax / (f.num_keyframes - 1)
    for nsteps in (2, 10, 30):
        t = torch.full((N, 1), float(ts * 0.5 * nsteps * 0.999), device=dev)
        base = torch.zeros(N, 1, device=dev)
        row = {"op": f"integrate_pos, {N} points x {nsteps} RK2 steps", "evals": 2 * nsteps * N}
        for sw in MODES:
            f.vel_fp16 = sw
            with torch.no_grad():
                ms = timeit(lambda: f.integrate_pos(x.clone(), t.clone(), base))
            row[NAME[sw] + "_ms"] = ms
            row[NAME[sw] + "_Mevals_per_s"] = row["evals"] / ms / 1e3
        out["entries"].append(row)
    o, d = bench.camera_bundle(dev)
    ren = Renderer(m, 0, 0, 2048)
    rays = Ray(o, d, 1.0, 8.0)
    row = {"op": "eval render of one 800x800 frame (640000 rays, 128 samples per ray) at t = 19/60 (1 RK2 step)"}
    for sw in MODES:
        f.vel_fp16 = sw
        ms = timeit(lambda: ren.render(19.0 / 60.0, rays, white_background=True, mode="test"), reps=2)
        row[NAME[sw] + "_ms"] = ms
        row[NAME[sw] + "_rays_per_s"] = 640000 / ms * 1e3
    out["entries"].append(row)
    row = {"op": "eval render of one 800x800 frame at t = 55/60 (extrapolation: 7 RK2 steps)"}
    for sw in MODES:
        f.vel_fp16 = sw
        ms = timeit(lambda: ren.render(55.0 / 60.0, rays, white_background=True, mode="test"), reps=1)
        row[NAME[sw] + "_ms"] = ms
        row[NAME[sw] + "_rays_per_s"] = 640000 / ms * 1e3
    out["entries"].append(row)
    row = {"op": "getDenseAlpha 128^3 (60 frame times)"}
    for sw in MODES:
        f.vel_fp16 = sw
        with torch.no_grad():
            ms = timeit(lambda: f.getDenseAlpha((128, 128, 128)), reps=1)
        row[NAME[sw] + "_ms"] = ms
    out["entries"].append(row)
    f.vel_fp16 = False
    print(json.dumps(out))


if __name__ == "__main__":
    main()
