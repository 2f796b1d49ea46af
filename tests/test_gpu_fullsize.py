"""BASELINE.json full-size checks (bat 199^3 grid, K=16, 128 samples/ray, 2048 rays, P=262144): the oracle cannot run the
whole step in seconds, so parity is established through (a) the oracle on a 256-ray slice of the same full-size field,
(b) size-independent properties: composite identities, directional finite differences of the loss against the analytic
gradient, PDE-loss linearity in the sample set, ray-sharding invariance."""
import numpy as np
import pytest
import torch

from helpers import assert_contract

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    import bench
    model = bench.build_scene(torch.device("cuda", 0), 199, 128, True)
    o, d = bench.camera_bundle(torch.device("cuda", 0))
    return model, o, d


def _rays(o, d, n, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    idx = torch.randint(0, o.shape[0], (n,), device="cuda", generator=g)
    return o[idx].contiguous(), d[idx].contiguous()


def test_fullsize_slice_matches_oracle(scene):
    from oracle import oracle as orc
    import bench
    model, o, d = scene
    f = model.nvfi
    ro, rd = _rays(o, d, 256, 1)
    u = torch.rand(256, 1)
    f.train()
    f.jitter_override = u
    try:
        out = f(19.0 / 60.0, ro, rd, True)
    finally:
        f.jitter_override = None
    sd = {k[len("nvfi."):]: v.detach().cpu().contiguous().numpy() for k, v in model.state_dict().items() if not k.startswith("nvfi.vel.vel_net.")}
    meta = dict(aabb=f.aabb.cpu().numpy(), gridSize=np.array(f.gridSize.tolist()), num_keyframes=f.num_keyframes, tmax=f.tmax,
                near=f.near_far[0], far=f.near_far[1], step_ratio=f.step_ratio, max_n_samples=f.max_n_samples,
                density_shift=f.density_shift, distance_scale=f.distance_scale, alphaMask_thres=f.alphaMask_thres,
                rayMarch_weight_thres=f.rayMarch_weight_thres, stepSize=f._step_host, nSamples=f.nSamples, use_sur=0, eps=0.03)
    fs = orc.FieldSpec(sd, meta)
    orc.set_threads(8)
    ref = orc.render(fs, ro.cpu().numpy(), rd.cpu().numpy(), 19.0 / 60.0, u=u.numpy(), train=True, white_bg=True)
    for i, (nm, r) in enumerate((("rgb", ref.rgb), ("depth", ref.depth), ("acc", ref.acc))):     # the 1e-4 contract, flip-band rays counted
        assert_contract(out[i].detach().cpu().numpy(), r, nm, label="hip bat 199^3/128")


def test_fullsize_composite_identities_and_sharding(scene):
    model, o, d = scene
    f = model.nvfi
    ro, rd = _rays(o, d, 2048, 2)
    f.eval()
    with torch.no_grad():
        rgb, depth, acc, w, _ = f(0.32, ro, rd, True)
        assert torch.allclose(w.sum(-1), acc, rtol=1e-5, atol=1e-6)            # acc = sum of weights
        assert (rgb >= 0).all() and (rgb <= 1).all() and (w >= 0).all() and (acc <= 1 + 1e-5).all()
        assert float(acc.mean()) > 0.02                                         # the blob is visible
        assert int(f.last_counters[0]) > 0.3 * 2048 * 128 and int(f.last_counters[2]) > 1000
        # rays are independent: rendering two shards separately reproduces the joint render bit-for-bit
        a = f(0.32, ro[:1024], rd[:1024], True)
        b = f(0.32, ro[1024:], rd[1024:], True)
        assert torch.equal(torch.cat([a[0], b[0]]), rgb) and torch.equal(torch.cat([a[1], b[1]]), depth)
        # keyframe time == the same time with the velocity switched off (no warp at a keyframe)
        k1 = f(0.30, ro[:512], rd[:512], True)
        f.use_vel = False
        try:
            k2 = f(0.30, ro[:512], rd[:512], True)
        finally:
            f.use_vel = True
        assert torch.allclose(k1[0], k2[0], atol=2e-6)


def test_fullsize_directional_derivative(scene):
    """loss(theta + eps*dir) - loss(theta - eps*dir) ~ 2 eps <grad, dir> for the full step loss (render + PDE)"""
    model, o, d = scene
    f = model.nvfi
    ro, rd = _rays(o, d, 2048, 3)
    u = torch.rand(2048, 1)
    g = torch.Generator(device="cuda"); g.manual_seed(9)
    target = torch.rand(2048, 3, device="cuda", generator=g)
    mn, mx = f.aabb
    pts = torch.rand(262144, 3, device="cuda", generator=g) * (mx - mn) + mn
    tt = torch.rand(262144, 1, device="cuda", generator=g)
    params = {"dps": f.density_plane_space[0], "apt": f.app_plane_time[1], "rW": f.renderModule.mlp[2].weight,
              "vW": f.vel_net.weight_net[4][0].weight, "aW": f.vel_net.a_weight_net[3][0].weight}

    def loss_fn():
        f.train()
        f.jitter_override = u
        try:
            out = f(19.0 / 60.0, ro, rd, True)
        finally:
            f.jitter_override = None
        lv = model.get_vel_loss(points=pts, t=tt)
        return torch.nn.functional.mse_loss(out[0], target) + 0.1 * out[1].mean() * 0.01 + 1.0 * lv

    model.zero_grad(set_to_none=True)
    loss = loss_fn()
    loss.backward()
    assert int(f.last_pde_out[1]) > 1000
    for name, p in params.items():
        gg = torch.Generator(device="cuda"); gg.manual_seed(11)
        direction = torch.randn(p.shape, device="cuda", generator=gg)
        if p.dim() == 4:
            direction = direction.contiguous(memory_format=torch.channels_last)
        analytic = float((p.grad.double() * direction.double()).sum())
        eps = {"dps": 2e-3, "apt": 2e-3, "rW": 1e-3, "vW": 2e-3, "aW": 5e-3}[name]
        with torch.no_grad():
            p.add_(eps * direction); lp = float(loss_fn().double())
            p.add_(-2 * eps * direction); lm = float(loss_fn().double())
            p.add_(eps * direction)
        numeric = (lp - lm) / (2 * eps)
        assert abs(numeric - analytic) <= 0.05 * max(abs(analytic), abs(numeric)) + 2e-5, (name, numeric, analytic)


def test_fullsize_pde_linearity(scene):
    """sum-of-squares bookkeeping: the PDE statistics of two disjoint point sets add up to those of their union"""
    model, o, d = scene
    f = model.nvfi
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    mn, mx = f.aabb
    pts = torch.rand(262144, 3, device="cuda", generator=g) * (mx - mn) + mn
    tt = torch.rand(262144, 1, device="cuda", generator=g)
    with torch.no_grad():
        model.get_vel_loss(points=pts, t=tt); full = f.last_pde_out.double().cpu().numpy()
        model.get_vel_loss(points=pts[:131072], t=tt[:131072]); a = f.last_pde_out.double().cpu().numpy()
        model.get_vel_loss(points=pts[131072:], t=tt[131072:]); b = f.last_pde_out.double().cpu().numpy()
    assert full[1] == a[1] + b[1] and full[1] > 1000
    np.testing.assert_allclose(full[2], a[2] + b[2], rtol=1e-5)
    np.testing.assert_allclose(full[3], a[3] + b[3], rtol=1e-5)
    np.testing.assert_allclose(full[0], 5 * full[2] / full[1] + 0.1 * full[3] / (3 * full[1]), rtol=1e-5)
