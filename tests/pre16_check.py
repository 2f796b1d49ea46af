"""Helper of test_gpu_prefilter16 (run in a subprocess: NVFI_PDE_PREFILTER is read once per process): get_vel_loss + backward on
N random collocation points per field; saves the kept mask, the loss, the velocity-net gradients and the device counters.

    python tests/pre16_check.py OUT.npz [N] [--bench]      (--bench adds the 199^3 synthetic bat field of bench.py)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_model, model_from_npz, named_grads      # noqa: E402
from conftest import GOLD                                        # noqa: E402


def run(name, model, N, out, seed):
    f = model.nvfi
    rng = np.random.default_rng(seed)
    aabb = f.aabb.detach().cpu().numpy()
    pts = (rng.uniform(0, 1, (N, 3)) * (aabb[1] - aabb[0]) + aabb[0]).astype(np.float32)
    t = rng.uniform(0, 1, (N, 1)).astype(np.float32)
    model.zero_grad(set_to_none=True)
    f.pde_debug = 1
    try:
        loss = model.get_vel_loss(points=torch.from_numpy(pts).cuda(), t=torch.from_numpy(t).cuda())
    finally:
        f.pde_debug = 0
    loss.backward()
    torch.cuda.synchronize()
    out[f"{name}:kept"] = f.last_pde_kept.cpu().numpy()
    out[f"{name}:loss"] = np.float64(loss.detach().cpu())
    out[f"{name}:counters"] = f.last_pde_counters.cpu().numpy()
    for k, v in named_grads(model).items():
        if v is not None and k.startswith("vel_net."):
            out[f"{name}:grad:{k}"] = v.copy()
    # timing of the whole call (forward only would skip nothing: loss and gradients come from one pass)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    pc, tc = torch.from_numpy(pts).cuda(), torch.from_numpy(t).cuda()
    for _ in range(2):
        model.get_vel_loss(points=pc, t=tc)
    ev[0].record()
    for _ in range(5):
        model.get_vel_loss(points=pc, t=tc)
    ev[1].record(); torch.cuda.synchronize()
    out[f"{name}:ms"] = np.float64(ev[0].elapsed_time(ev[1]) / 5)


def flips_run(path, total):
    """kept masks of `total` random collocation points of the bench field in batches of 2^20 (forward only), packed to bits"""
    import bench
    model = bench.build_scene(torch.device("cuda"))
    f = model.nvfi
    rng = np.random.default_rng(77)
    aabb = f.aabb.detach().cpu().numpy()
    B = 1 << 20
    masks, band = [], 0
    with torch.no_grad():
        for _ in range((total + B - 1) // B):
            pts = (rng.uniform(0, 1, (B, 3)) * (aabb[1] - aabb[0]) + aabb[0]).astype(np.float32)
            t = rng.uniform(0, 1, (B, 1)).astype(np.float32)
            f.pde_debug = 1
            try:
                model.get_vel_loss(points=torch.from_numpy(pts).cuda(), t=torch.from_numpy(t).cuda())
            finally:
                f.pde_debug = 0
            masks.append(np.packbits(f.last_pde_kept.cpu().numpy().astype(bool)))
            band += int(f.last_pde_counters.cpu().numpy()[5])
    np.savez(path, kept=np.concatenate(masks), band=np.int64(band), n=np.int64(len(masks) * B))


def repeat_run(total):
    """the prefilter pre-pass eight times on the same `total` points of the bench field: kept mask and counters must not change"""
    import bench
    model = bench.build_scene(torch.device("cuda"))
    f = model.nvfi
    rng = np.random.default_rng(5)
    aabb = f.aabb.detach().cpu().numpy()
    pts = torch.from_numpy((rng.uniform(0, 1, (total, 3)) * (aabb[1] - aabb[0]) + aabb[0]).astype(np.float32)).cuda()
    t = torch.from_numpy(rng.uniform(0, 1, (total, 1)).astype(np.float32)).cuda()
    outs = []
    with torch.no_grad():
        for _ in range(8):
            f.pde_debug = 1
            try:
                model.get_vel_loss(points=pts, t=t)
            finally:
                f.pde_debug = 0
            outs.append((f.last_pde_kept.clone(), f.last_pde_counters.clone()))
    same = all(torch.equal(a[0], outs[0][0]) and torch.equal(a[1][:6], outs[0][1][:6]) for a in outs[1:])
    print("kept", int(outs[0][0].sum()), "counters", outs[0][1].tolist(), "repeat identical:", same)


def main():
    path = sys.argv[1]
    if "--repeat" in sys.argv:
        return repeat_run(int(sys.argv[sys.argv.index("--repeat") + 1]))
    if "--flips" in sys.argv:
        return flips_run(path, int(sys.argv[sys.argv.index("--flips") + 1]))
    N = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else 262144
    out = {}
    for kind in ("A", "B"):
        model, _ = make_model(kind)
        run(kind, model, N, out, 11)
    z = np.load(os.path.join(GOLD, "cfg1.npz"))
    run("cfg1", model_from_npz(z, "")[0], N, out, 12)
    if "--bench" in sys.argv:
        import bench
        run("bench", bench.build_scene(torch.device("cuda")), N, out, 13)
    np.savez(path, **out)
    print({k: (v.tolist() if v.size < 10 else v.shape) for k, v in out.items() if k.endswith((":loss", ":counters", ":ms"))})


if __name__ == "__main__":
    main()
