"""Worker of test_gpu_edges.py::test_scatter_kernels_agree: the plane gradients of one training backward (a keyframe time and a
non-keyframe time, grid sizes that are not multiples of the tile edge) for the scatter kernel selected by the environment
(NVFI_SCATTER / NVFI_SCATTER_TILES are read once per process).  usage: scatter_modes_worker.py <out.npz> [Gx Gy Gz]"""
import sys

import numpy as np
import torch

from helpers import make_model


def main(out, grid=(37, 50, 41)):
    model, meta = make_model("A")
    f = model.nvfi
    f.upsample_volume_grid([int(g) for g in grid], int(meta["num_keyframes"]))
    f.train()
    g = torch.Generator().manual_seed(11)
    n = 3000
    o = torch.tensor([[2.6, -2.2, 2.4]]).expand(n, 3).contiguous().cuda()
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 4.4          # some rays graze or leave the box
    d = tgt - o.cpu()
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous().cuda()
    model.zero_grad(set_to_none=True)
    torch.manual_seed(3)                                        # per-ray jitter and the random-white coin are CPU draws
    for t in (0.0, 0.41):
        rgb, depth = f(t, o, d, True)[:2]
        (rgb.square().mean() + 0.01 * depth.mean()).backward()
    planes = {}
    for name, p in f.named_parameters():
        if "plane" in name and p.grad is not None:
            planes[name] = p.grad.detach().float().cpu().numpy()
    assert len(planes) == 12, sorted(planes)
    np.savez(out, **planes)


if __name__ == "__main__":
    main(sys.argv[1], tuple(sys.argv[2:5]) if len(sys.argv) >= 5 else (37, 50, 41))
