"""Per-iteration point pipeline of the reference's segmentation training (train_segm.py:126-170), on the HIP kernels:
jittered volume points -> density at t = 0 -> keep occupied points -> advect them to a random time with the velocity field ->
(points, flow) for the MaskField losses.  The losses themselves (dynamic rigid-fit / kNN smoothness / entropy,
train_segm.py:172-186) are PyTorch code in the reference and stay PyTorch here (out of scope)."""
import numpy as np
import torch


def sample_volume_points(volume_bounds, n_sample_point=64, perturb=False, device=None):
    """utils/point_util.py:4-21: one jittered (or centred) point per cell of an n^3 lattice over volume_bounds (3,2) -> (n,n,n,3).
    The random draw is made on the CPU generator, as in the reference (n x 3 numbers).  device: build the n^3 lattice THERE from the n x 3
    per-axis coordinates (same values; the reference builds it on the host and copies 3 MB per iteration)."""
    t_vals = torch.linspace(0.0, 1.0, steps=n_sample_point + 1).unsqueeze(1)
    vb = volume_bounds.detach().cpu().float() if not isinstance(volume_bounds, (list, tuple)) else torch.tensor(volume_bounds, dtype=torch.float32)
    xyz_vals = vb[:, 0] * (1 - t_vals) + vb[:, 1] * t_vals
    lower, upper = xyz_vals[:-1], xyz_vals[1:]
    if perturb:
        points = lower + (upper - lower) * torch.rand(n_sample_point, 3)
    else:
        points = 0.5 * (lower + upper)
    if device is not None:
        points = points.to(device)
    x, y, z = torch.meshgrid(points[:, 0], points[:, 1], points[:, 2], indexing="ij")
    return torch.stack([x, y, z], 3)


def balanced_sample(xyz, object_bounds):
    """train_segm.py:35-52: keep every foreground point (inside object_bounds (3,2)) and at most as many background points."""
    ob = object_bounds.to(xyz.device)
    fg = ((xyz > ob[:, 0]) & (xyz < ob[:, 1])).all(dim=1)
    xyz_fg, xyz_bg = xyz[fg], xyz[~fg]
    if xyz_bg.shape[0] > xyz_fg.shape[0]:
        idx = np.random.choice(xyz_bg.shape[0], xyz_fg.shape[0], replace=False)
        xyz_bg = xyz_bg[torch.from_numpy(idx).to(xyz.device)]
    return torch.cat([xyz_fg, xyz_bg], 0)


@torch.no_grad()
def segm_points(kplane, n_sample_res, min_t, alpha_scale=10.0, object_bounds=None, t=None, dists=0.01, advect_events=None):
    """train_segm.py:127-170.  Returns (xyz (N,3) normalised keyframe-0 points, flow (N,3), t) for one iteration."""
    device = kplane.aabb.device
    ab = kplane._aabb_host                       # host copy of the box (no device read per iteration)
    volume_bounds = [[ab[0], ab[3]], [ab[1], ab[4]], [ab[2], ab[5]]]
    xyz = sample_volume_points(volume_bounds, n_sample_res, perturb=True, device=device).reshape(-1, 3)
    xyz = kplane.normalize_coord(xyz)
    t0 = torch.zeros(xyz.shape[0], 1, device=device)
    sigma_feature = kplane.compute_densityfeature(torch.cat([xyz, kplane.normalize_time_coord(t0)], dim=1))
    sigma = kplane.feature2density(sigma_feature, {})
    alpha = 1.0 - torch.exp(-sigma * dists)
    xyz = xyz[alpha > (kplane.alphaMask_thres * alpha_scale)]
    xyz_org = (xyz + 1) / kplane.invaabbSize + kplane.aabb[0]
    if object_bounds is not None:
        xyz_org = balanced_sample(xyz_org, object_bounds)
    xyz = kplane.normalize_coord(xyz_org)
    n = xyz.shape[0]
    t0 = torch.zeros(n, 1, device=device)
    if t is None:
        t = float(min_t + (kplane.tmax - min_t) * torch.rand(1))
    tt = t0 + t
    if advect_events is not None:       # (bench.py: HIP events around the advection on the stream it is launched on)
        advect_events[0].record()
    xyz2 = kplane.integrate_pos(xyz.clone(), t0, tt)
    if advect_events is not None:
        advect_events[1].record()
    return xyz, xyz2 - xyz, t
