"""BASELINE.json configs[3] ('chessboard dynamic indoor scene'): the values of config/InDoorSeg/chessboard.yaml that reach the hot
path at the shipped initial resolution - anisotropic aabb (x,y in +-3.03, z in [-0.03, 6.03]), surround box
(VelocityAABBSur with step rejection), K=4 keyframes, density_shift -5, distance_scale 10, near 0.8 / far 8.1, no white
background, step_ratio 0.5 (S = 219 samples per ray at 64^3), P = 131072 collocation points.  Parity: the oracle on a 192-ray
slice of the same field (train and eval mode), plus size-independent properties on the full 2048-ray / 131072-point step."""
import numpy as np
import pytest
import torch

from helpers import assert_contract

pytestmark = pytest.mark.gpu


def chessboard_cfg():
    from nvfi_amd.utils import CfgNode
    n = dict(model_name="TensorVMKeyframeTimeKplane", density_n_comp=[24, 24, 24], appearance_n_comp=[48, 48, 48], app_dim=32,
             densityMode="Density", shadingMode="MLP_PE", alphaMask_thres=1e-4, rayMarch_weight_thres=1e-4, density_shift=-5,
             distance_scale=10, pos_pe=6, view_pe=6, fea_pe=6, featureC=128, step_ratio=0.5, fea2denseAct="softplus",
             max_n_samples=1024, num_keyframes=4, num_keyframes_end=4, tmax=0.75, use_vel=True,
             sur_x=[-2.5, 2.5], sur_y=[-2.5, 2.5], sur_z=[0.02, 5.95])
    return CfgNode(dict(nvfi=n))


@pytest.fixture(scope="module")
def scene():
    from nvfi_amd.models import NVFi
    torch.manual_seed(233)
    aabb = torch.tensor([[-3.03, -3.03, -0.03], [3.03, 3.03, 6.03]])
    m = NVFi(chessboard_cfg(), "cpu", aabb, [64, 64, 64], [0.8, 8.1])
    f = m.nvfi
    with torch.no_grad():      # the blob recipe of BASELINE.md section 2 (the init field is fully transparent)
        for i in range(3):
            p = f.density_plane_space[i]
            H, W = p.shape[-2:]
            yy = torch.linspace(-1, 1, H)[:, None]; xx = torch.linspace(-1, 1, W)[None, :]
            p.mul_((2.6 * torch.sqrt(torch.exp(-xx ** 2 / (2 * 0.35 ** 2)) * torch.exp(-yy ** 2 / (2 * 0.35 ** 2))))[None, None])
        f.vel_net.weight_net[7][0].weight.mul_(4.0)       # visible motion, so that the surround-box rejection is exercised
        f.vel_net.weight_net[7][0].bias.add_(0.3)
    m = m.to("cuda")
    assert f.nSamples == 219, f.nSamples
    # camera on the +x side looking at the aabb centre (0,0,3), r = 4 (assumed pose)
    c = torch.tensor([0.0, 0.0, 3.0])
    eye = c + torch.tensor([4.0 * 0.8, 4.0 * 0.45, 4.0 * 0.4])
    g = torch.Generator().manual_seed(1)
    tgt = c + (torch.rand(4096, 3, generator=g) - 0.5) * torch.tensor([5.0, 5.0, 5.0])
    d = tgt - eye
    d = d / d.norm(dim=-1, keepdim=True)
    return m, eye.expand(4096, 3).contiguous().cuda(), d.contiguous().cuda()


def _field_spec(model):
    from oracle import oracle as orc
    f = model.nvfi
    sd = {k[len("nvfi."):]: v.detach().cpu().contiguous().numpy() for k, v in model.state_dict().items() if not k.startswith("nvfi.vel.vel_net.")}
    sur = f.vel.bounds.detach().float().cpu().numpy().reshape(2, 3)      # normalised surround box (velocity_field.py:44)
    meta = dict(aabb=f.aabb.cpu().numpy(), gridSize=np.array(f.gridSize.tolist()), num_keyframes=f.num_keyframes, tmax=f.tmax,
                near=f.near_far[0], far=f.near_far[1], step_ratio=f.step_ratio, max_n_samples=f.max_n_samples,
                density_shift=f.density_shift, distance_scale=f.distance_scale, alphaMask_thres=f.alphaMask_thres,
                rayMarch_weight_thres=f.rayMarch_weight_thres, stepSize=f._step_host, nSamples=f.nSamples, use_sur=1, sur_bounds=sur)
    return orc.FieldSpec(sd, meta)


@pytest.mark.parametrize("train", [False, True])
def test_chessboard_slice_matches_oracle(scene, train):
    from oracle import oracle as orc
    model, o, d = scene
    f = model.nvfi
    fs = _field_spec(model)
    orc.set_threads(8)
    ro, rd = o[:192].contiguous(), d[:192].contiguous()
    t = 22.0 / 60.0          # keyframes at 0, 0.25, 0.5, 0.75: |t - 0.25| = 0.117 -> one RK2 step of dt_max = 0.125
    if train:
        f.train()
        u = torch.rand(192, 1)
        f.jitter_override = u
        try:
            out = f(t, ro, rd, True)           # white_bg=True: the random-white coin is not drawn
        finally:
            f.jitter_override = None
        ref = orc.render(fs, ro.cpu().numpy(), rd.cpu().numpy(), t, u=u.numpy(), train=True, white_bg=True)
    else:
        f.eval()
        with torch.no_grad():
            out = f(t, ro, rd, False)          # chessboard.yaml: white_background False
        ref = orc.render(fs, ro.cpu().numpy(), rd.cpu().numpy(), t, train=False, white_bg=False)
    for i, (nm, r) in enumerate((("rgb", ref.rgb), ("depth", ref.depth), ("acc", ref.acc))):      # the 1e-4 contract, flip-band rays counted
        assert_contract(out[i].detach().cpu().numpy(), r, nm, label=f"hip chessboard train={train}")
    assert float(ref.acc.mean()) > 0.01


def test_chessboard_full_step_properties(scene):
    model, o, d = scene
    f = model.nvfi
    ro, rd = o[:2048].contiguous(), d[:2048].contiguous()
    f.eval()
    with torch.no_grad():
        rgb, depth, acc, w, _ = f(40.0 / 60.0, ro, rd, False)         # two RK2 steps towards the keyframe at 0.75
        assert torch.allclose(w.sum(-1), acc, rtol=1e-5, atol=1e-6)
        assert (rgb >= 0).all() and (rgb <= 1).all() and torch.isfinite(depth).all()
        a = f(40.0 / 60.0, ro[:1024], rd[:1024], False)
        b = f(40.0 / 60.0, ro[1024:], rd[1024:], False)
        assert torch.equal(torch.cat([a[0], b[0]]), rgb)               # rays are independent (what the 8-GPU sharding relies on)
        # PDE statistics of disjoint point sets add up (P = 131072, chessboard.yaml vel_reg_n_pts)
        g = torch.Generator(device="cuda"); g.manual_seed(5)
        mn, mx = f.aabb
        pts = torch.rand(131072, 3, device="cuda", generator=g) * (mx - mn) + mn
        tt = torch.rand(131072, 1, device="cuda", generator=g)
        model.get_vel_loss(points=pts, t=tt); full = f.last_pde_out.double().cpu().numpy()
        model.get_vel_loss(points=pts[:65536], t=tt[:65536]); pa = f.last_pde_out.double().cpu().numpy()
        model.get_vel_loss(points=pts[65536:], t=tt[65536:]); pb = f.last_pde_out.double().cpu().numpy()
    assert full[1] == pa[1] + pb[1] and full[1] > 500
    np.testing.assert_allclose(full[2], pa[2] + pb[2], rtol=1e-5)
    np.testing.assert_allclose(full[3], pa[3] + pb[3], rtol=1e-5)


def test_chessboard_pde_slice_matches_oracle(scene):
    from oracle import oracle as orc
    model, o, d = scene
    f = model.nvfi
    fs = _field_spec(model)
    orc.set_threads(8)
    g = torch.Generator(device="cuda"); g.manual_seed(6)
    mn, mx = f.aabb
    pts = torch.rand(4096, 3, device="cuda", generator=g) * (mx - mn) + mn
    tt = torch.rand(4096, 1, device="cuda", generator=g)
    model.zero_grad(set_to_none=True)
    lv = model.get_vel_loss(points=pts, t=tt)
    ref = orc.pde_loss(fs, pts.cpu().numpy(), tt.cpu().numpy())
    assert abs(int(f.last_pde_out[1]) - int(ref["n_kept"])) <= 2 and ref["n_kept"] > 50
    np.testing.assert_allclose(float(lv.detach()), ref["loss"], rtol=2e-3)
    lv.backward()
    gW = f.vel_net.weight_net[4][0].weight.grad.cpu().numpy()
    rW = ref["grads"]["vel_net.weight_net.4.0.weight"]
    assert np.abs(gW - rW).max() <= 2e-3 * np.abs(rW).max() + 1e-9


def test_chessboard_final_resolution_slice(scene):
    """configs[3] at the END of the shipped schedule: the 64^3 field upsampled to 199 x 199 x 200 (N_to_reso of 8e6 voxels in the
    chessboard box, train_nvfi.py:99-112,343-347) -> 688 samples per ray.  A 64-ray train-mode slice (forward + backward through the
    surround-box gate) against the oracle, and a 1024-ray eval chunk for the composite identities at 1.4 M samples per chunk."""
    import copy
    from oracle import oracle as orc
    from nvfi_amd.utils import N_to_reso
    model0, o, d = scene
    model = copy.deepcopy(model0)
    f = model.nvfi
    f.vel_net._owner = __import__("weakref").ref(f)
    reso = N_to_reso(7999998, f.aabb.cpu())      # last entry of the shipped voxel schedule, evaluated on the host like the reference
    assert list(reso) == [199, 199, 200], reso
    f.upsample_volume_grid(reso, 4)
    assert f.nSamples == 688, f.nSamples
    fs = _field_spec(model)
    orc.set_threads(8)
    ro, rd = o[:64].contiguous(), d[:64].contiguous()
    t = 22.0 / 60.0
    f.train()
    model.zero_grad(set_to_none=True)
    u = torch.rand(64, 1)
    f.jitter_override = u
    try:
        out = f(t, ro, rd, True)
    finally:
        f.jitter_override = None
    tg = torch.rand(64, 3, device="cuda")
    (torch.nn.functional.mse_loss(out[0], tg) + 0.01 * out[1].mean()).backward()
    ref = orc.render(fs, ro.cpu().numpy(), rd.cpu().numpy(), t, u=u.numpy(), train=True, white_bg=True, keep_ctx=True)
    for i, (nm, r) in enumerate((("rgb", ref.rgb), ("depth", ref.depth), ("acc", ref.acc))):
        assert_contract(out[i].detach().cpu().numpy(), r, nm, label="hip chessboard 199x199x200 / 688")
    gref = ref.backward(fs, g_rgb=2 * (ref.rgb - tg.cpu().numpy()) / (64 * 3), g_depth=np.full(64, 0.01 / 64, np.float32))
    ref.free()
    for pn, g in (("density_plane_space.0", f.density_plane_space[0].grad), ("app_plane_time.2", f.app_plane_time[2].grad),
                  ("renderModule.mlp.2.weight", f.renderModule.mlp[2].weight.grad), ("vel_net.weight_net.3.0.weight", f.vel_net.weight_net[3][0].weight.grad)):
        gr = gref[pn]
        e = np.abs(g.cpu().numpy() - gr).max() / (np.abs(gr).max() + 1e-30)
        assert e < 5e-4, (pn, e)
    f.eval()
    with torch.no_grad():
        rgb, depth, acc, w, _ = f(40.0 / 60.0, o[:1024].contiguous(), d[:1024].contiguous(), False)
    assert w.shape == (1024, 688) and torch.allclose(w.sum(-1), acc, rtol=1e-5, atol=1e-6) and torch.isfinite(depth).all()
