#!/usr/bin/env python
"""CPU bridge measurement, run in the BUILD CONTAINER only (it imports /root/reference):
the reference's own PyTorch-CPU implementation and this repo's C/OpenMP restatement (oracle) are timed on the SAME bounded
sample of the bench step - the synthetic bat scene of bench.py (199^3, K=16, 128 samples/ray), 256 rays x 2 renders (t=19/60
and t=0.30, train mode, fwd+bwd) + get_vel_loss on P=32768 collocation points (fwd+bwd) - so that the `cpu_baseline` the
bench reports from the GPU box's host (oracle) can be related to the reference itself.

    python tools/cpu_bridge.py profiles/r02_cpu_bridge.json      # prints one JSON object and writes it (bench.py quotes it as "reference_cpu")
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    model = bench.build_scene("cpu")                       # our module is only a parameter container here
    cpu = "unknown"
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    out = {"host_threads": threads, "host": f"build container: {threads} vCPU, {cpu}, torch {torch.__version__} (CPU kernels)"}
    # --- oracle (same function bench.py uses on the GPU box)
    out["oracle"] = bench.cpu_baseline(model, "cfg3", seconds_hint=40)
    # --- the reference itself on the same field
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as mg
    R = mg.import_reference()
    import yaml
    with open("/root/reference/config/InDoorObj/bat.yaml") as f:
        cfg = R["CfgNode"](yaml.safe_load(f))
    cfg.nvfi.max_n_samples = 128
    cfg.nvfi.step_ratio = float(model.nvfi.step_ratio)
    aabb = torch.tensor([[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]])
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):      # the reference prints its configuration
        ref = R["NVFi"](cfg, "cpu", aabb, [199, 199, 199], [1.0, 8.0])
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    missing = ref.load_state_dict(sd, strict=False)
    assert ref.nvfi.nSamples == 128, ref.nvfi.nSamples
    ren = R["Renderer"](ref, 0, 0, 2048)
    o, d = bench.camera_bundle("cpu")
    rng = np.random.default_rng(5)
    Rr, P = 256, 32768

    def step():
        ref.zero_grad(set_to_none=True)
        loss = 0
        for t in (19 / 60.0, 0.30):
            idx = torch.from_numpy(rng.integers(0, o.shape[0], Rr))
            rays = R["Ray"](o[idx], d[idx], 1.0, 8.0)
            rgb = ren.render(t, rays, white_background=True, mode="train")[0]
            loss = loss + torch.nn.functional.mse_loss(rgb, torch.rand(Rr, 3))
        lv = ref.get_vel_loss(P)
        loss = loss + lv
        loss.backward()

    step()
    t0 = time.perf_counter(); n = 0
    while n < 3:
        step(); n += 1
    dt = (time.perf_counter() - t0) / n
    out["reference"] = dict(value=2 * Rr / dt, unit="rays/s", cores=threads, kind="reference",
                            sample=f"vLAR-group/NVFi on torch {torch.__version__} CPU x{threads}: {Rr} rays per render x2 renders + get_vel_loss(P={P}), fwd+bwd, {n} reps",
                            state_dict_missing=len(missing.missing_keys), state_dict_unexpected=len(missing.unexpected_keys))
    out["oracle_over_reference"] = out["oracle"]["value"] / out["reference"]["value"]
    txt = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
