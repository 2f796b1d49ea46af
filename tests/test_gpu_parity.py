"""GPU parity: the HIP path (through nvfi_amd.models -> ctypes -> libnvfi_hip.so) against the golden
vectors captured from the reference and against the CPU oracle on the same seeded inputs.

Tolerances: BASELINE.json's contract is rgb/depth within 1e-4 relative of the PyTorch-CPU reference.
Discrete decisions (weight > 1e-4 appearance mask, in-box tests) can flip for samples within
rounding of a threshold; a flip moves a ray's colour by at most ~1e-4 absolute, so composited
outputs are compared with a PURE rtol=1e-4 (+ the fp32 rounding floor of the composite); the rays that
need the absolute threshold-flip band are counted and must stay below 0.5 % (helpers.assert_contract).
"""
import numpy as np
import pytest
import torch

from conftest import assert_grad, relerr
from helpers import make_model, named_grads, assert_contract

pytestmark = pytest.mark.gpu
KINDS = ["A", "B"]


def _cuda(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


@pytest.fixture(scope="module")
def models():
    return {k: make_model(k) for k in KINDS}


def test_native_library_loaded():
    from nvfi_amd import _lib
    L = _lib.lib()
    assert L.nvfi_abi_version() == 5
    import ctypes as C
    err = C.c_float(-1)
    _lib.check(L.nvfi_selftest(C.byref(err), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert 0 <= err.value < 2e-4, err.value   # fp32 MFMA fragment layout (A: weights, B: samples, D rows) is right


@pytest.mark.parametrize("kind", KINDS)
def test_velocity_basis(gold, models, kind):
    f = models[kind][0].nvfi
    xt = _cuda(gold[f"{kind}:vel:xt"])
    u = f.vel_net(xt).cpu().numpy()
    np.testing.assert_allclose(u, gold[f"{kind}:vel:u"], rtol=1e-4, atol=1e-5)
    g = f.vel(xt).cpu().numpy()
    ref = gold[f"{kind}:vel:gated"]
    assert np.array_equal(g == 0, ref == 0)
    np.testing.assert_allclose(g, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("kind", KINDS)
def test_integrate_pos(gold, models, kind):
    f = models[kind][0].nvfi
    x0, t, base = (_cuda(gold[f"{kind}:integrate:{n}"]) for n in ("x0", "t", "base"))
    xk = f.integrate_pos(x0, t, base).cpu().numpy()
    np.testing.assert_allclose(xk, gold[f"{kind}:integrate:xk"], rtol=1e-4, atol=1e-5)
    xf = f.integrate_pos(x0, torch.zeros_like(t), _cuda(gold[f"{kind}:integrate_fwd:t_target"])).cpu().numpy()
    np.testing.assert_allclose(xf, gold[f"{kind}:integrate_fwd:xk"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("kind", KINDS)
def test_density_and_appearance(gold, fields, models, kind):
    from oracle import oracle as orc
    f = models[kind][0].nvfi
    xyzt = gold[f"{kind}:feat:xyzt"]
    d = f.compute_densityfeature(_cuda(xyzt)).cpu().numpy()
    np.testing.assert_allclose(d, gold[f"{kind}:feat:density"], rtol=1e-4, atol=1e-5)
    view = gold[f"{kind}:mlp:view"][: xyzt.shape[0]]
    rgb = f.app_at(_cuda(xyzt), _cuda(view)).cpu().numpy()
    feat = orc.app_feature(fields[kind], xyzt)
    ref = orc.render_mlp(fields[kind], xyzt[:, :3], view, feat)
    np.testing.assert_allclose(feat, gold[f"{kind}:feat:app"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(rgb, ref, rtol=1e-4, atol=1e-5)


def _render(model, meta, gold, kind, t, mode, white=None, transfer=False):
    from nvfi_amd.models import Renderer, Ray
    ren = Renderer(model, 0, 0, 2048)
    o, d = _cuda(gold[f"{kind}:rays_o"]), _cuda(gold[f"{kind}:rays_d"])
    wb = bool(meta["white_background"]) if white is None else white
    return ren.render(t, Ray(o, d, 0, 1), white_background=wb, mode=mode, transfer_vel=transfer)


def _check_maps(out, gold, key, counts=True):
    rgb, depth, acc, w = (x.detach().cpu().numpy() for x in out[:4])
    for name, got in (("rgb", rgb), ("depth", depth), ("acc", acc)):
        assert_contract(got, gold[f"{key}:{name}"], name, label=f"hip {key}")
    assert_contract(w, gold[f"{key}:weight"], "weight", rtol=2e-4, label=f"hip {key}")


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("case", ["key", "nonkey", "extrap", "transfer", "flipbg", "amask"])
def test_render_eval(gold, models, kind, case):
    from nvfi_amd.models import AlphaGridMask
    model, meta = models[kind]
    t, white, transfer = 19.0 / 60.0, None, False
    if case in ("key", "nonkey", "extrap", "transfer"):
        t = float(gold[f"{kind}:render_{case}:t"])
    if case == "flipbg":
        white = not bool(meta["white_background"])
    transfer = case == "transfer"
    f = model.nvfi
    if case == "amask":
        f.alphaMask = AlphaGridMask("cuda", f.aabb, _cuda(gold[f"{kind}:render_amask:volume"]))
    try:
        out = _render(model, meta, gold, kind, t, "test", white, transfer)
    finally:
        f.alphaMask = None
    _check_maps(out, gold, f"{kind}:render_{case}")


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("name", ["train_nonkey", "train_key", "train_extrap"])
def test_render_train_grads(gold, models, kind, name):
    model, meta = models[kind]
    model.zero_grad(set_to_none=True)
    t = float(gold[f"{kind}:{name}:t"])
    torch.manual_seed(21)   # same CPU-generator stream as the reference: jitter, then the white coin
    out = _render(model, meta, gold, kind, t, "train")
    _check_maps(out, gold, f"{kind}:{name}")
    target, gw = _cuda(gold[f"{kind}:{name}:target"]), _cuda(gold[f"{kind}:{name}:gw"])
    loss = torch.nn.functional.mse_loss(out[0], target) + 0.01 * out[1].mean() + 0.02 * (out[2] ** 2).mean() + (out[3] * gw).sum()
    loss.backward()
    np.testing.assert_allclose(loss.item(), gold[f"{kind}:{name}:loss"][0], rtol=1e-4)
    g = named_grads(model)
    checked = 0
    for k in gold.files:
        pre = f"{kind}:{name}:grad:nvfi."
        if not k.startswith(pre):
            continue
        pn, ref = k[len(pre):], gold[k]
        if pn == "basis_mat_density.weight":
            continue
        if ref.size == 0:
            assert g[pn] is None or not np.any(g[pn]), pn
            continue
        assert_grad(g[pn], ref, 5e-4, pn)   # max-norm AND L2 relative, element-wise outliers bounded; fp32 atomics + different summation order
        checked += 1
    assert checked >= 3


@pytest.mark.parametrize("kind", KINDS)
def test_render_vs_oracle_bigger(fields, models, kind):
    """HIP vs oracle on seeded rays at a size the oracle finishes in seconds (R=768, train mode, fwd+bwd)."""
    from oracle import oracle as orc
    from nvfi_amd.models import Renderer, Ray
    model, meta = models[kind]
    fs = fields[kind]
    rng = np.random.default_rng(7)
    R = 768
    o = np.tile(np.array([[2.6, -2.2, 2.4]], np.float32) + (0 if kind == "A" else np.array([[0, 0, 3.0]], np.float32)), (R, 1))
    tgt = rng.uniform(-0.9, 0.9, (R, 3)).astype(np.float32) * (fs.aabb[1] - fs.aabb[0]) * 0.5 + (fs.aabb[1] + fs.aabb[0]) * 0.5
    d = (tgt - o) / np.linalg.norm(tgt - o, axis=1, keepdims=True) * rng.uniform(0.9, 1.1, (R, 1)).astype(np.float32)
    d = d.astype(np.float32)
    u = rng.uniform(0, 1, (R, 1)).astype(np.float32)
    t = 0.41
    wb = bool(meta["white_background"])
    f = model.nvfi
    f.train()
    model.zero_grad(set_to_none=True)
    f.jitter_override = torch.from_numpy(u.copy())     # explicit per-ray jitter instead of the CPU-RNG draw
    try:
        out = f(t, _cuda(o), _cuda(d), wb or True)
    finally:
        f.jitter_override = None
    tg = _cuda(rng.uniform(0, 1, (R, 3)).astype(np.float32))
    loss = torch.nn.functional.mse_loss(out[0], tg) + 0.01 * out[1].mean()
    loss.backward()
    ref = orc.render(fs, o, d, t, u=u, train=True, white_bg=True, keep_ctx=True)
    rgb = out[0].detach().cpu().numpy()
    assert_contract(rgb, ref.rgb, "rgb", label=f"hip vs oracle {kind} R=768")
    assert_contract(out[1].detach().cpu().numpy(), ref.depth, "depth", label=f"hip vs oracle {kind} R=768")
    cnt = f.last_counters.cpu().numpy()
    assert abs(int(cnt[0]) - int(ref.counters[0])) <= 2 and abs(int(cnt[2]) - int(ref.counters[2])) <= 4
    gref = ref.backward(fs, g_rgb=2 * (ref.rgb - tg.cpu().numpy()) / (R * 3), g_depth=np.full(R, 0.01 / R, np.float32))
    ref.free()
    g = named_grads(model)
    for pn in ("density_plane_space.0", "density_plane_time.2", "app_plane_space.1", "app_plane_time.0", "basis_mat.weight",
               "renderModule.mlp.0.weight", "renderModule.mlp.2.bias", "renderModule.mlp.4.weight",
               "vel_net.weight_net.1.weight", "vel_net.weight_net.4.0.weight", "vel_net.weight_net.7.0.bias"):
        e = relerr(g[pn], gref[pn])
        assert e < 5e-4, (pn, e)


@pytest.mark.parametrize("kind", KINDS)
def test_render_module_is_callable_like_the_reference(gold, models, kind):
    """field.renderModule(pts, viewdirs, features) (tensorf_base.py:88-98) against the reference's own outputs for given features"""
    model, meta = models[kind]
    f = model.nvfi
    with torch.no_grad():
        rgb = f.renderModule(_cuda(gold[f"{kind}:mlp:pts"]), _cuda(gold[f"{kind}:mlp:view"]), _cuda(gold[f"{kind}:mlp:feat"]), {})
    np.testing.assert_allclose(rgb.cpu().numpy(), gold[f"{kind}:mlp:rgb"], rtol=1e-4, atol=2e-6)
    with pytest.raises(NotImplementedError):       # forward only: a stand-alone call that wants gradients is refused, not detached
        f.renderModule(_cuda(gold[f"{kind}:mlp:pts"]), _cuda(gold[f"{kind}:mlp:view"]), _cuda(gold[f"{kind}:mlp:feat"]).requires_grad_(), {})


@pytest.mark.parametrize("kind", KINDS)
def test_pde_loss(gold, models, kind):
    model, meta = models[kind]
    f = model.nvfi
    model.zero_grad(set_to_none=True)
    f.pde_debug = 64
    try:
        loss = model.get_vel_loss(points=_cuda(gold[f"{kind}:pde:points"]), t=_cuda(gold[f"{kind}:pde:t"]))
    finally:
        f.pde_debug = 0
    kept = f.last_pde_kept.cpu().numpy().astype(bool)
    ref_kept = gold[f"{kind}:pde:kept"]
    assert np.mean(kept != ref_kept) < 2e-3      # alpha within rounding of the 1e-4 threshold may flip
    # functorch Jacobians of the first 64 kept points of each side, compared on the points BOTH kept (a borderline point may flip
    # the kept set; until round 3 a single flip skipped this check altogether)
    ours_ids, ref_ids = np.nonzero(kept)[0][:64], np.nonzero(ref_kept)[0][:64]
    common, io, ir = np.intersect1d(ours_ids, ref_ids, return_indices=True)
    assert len(common) >= 56, len(common)
    # rows 0-2 = d v / d (x, y, z, t): all the loss reads (nvfi.py:77-79).  Rows 3-5 of the golden are the Jacobian of the ACCELERATION head, which the
    # reference's jacrev forms and never uses; neither the oracle nor the kernels form them
    np.testing.assert_allclose(f.last_pde_jac.cpu().numpy()[io, :3], gold[f"{kind}:pde:jac64"][ir, :3], rtol=2e-4, atol=5e-5)
    assert float(f.last_pde_jac[:, 3:].abs().max()) == 0.0
    np.testing.assert_allclose(float(loss.detach()), gold[f"{kind}:pde:loss"][0], rtol=5e-4)
    (loss * 1.0).backward()
    g = named_grads(model)
    n = 0
    for k in gold.files:
        pre = f"{kind}:pde:grad:"
        if k.startswith(pre):
            assert_grad(g["vel_net." + k[len(pre):]], gold[k], 1e-3, k)
            n += 1
    assert n >= 8


@pytest.mark.parametrize("kind", KINDS)
def test_plane_regularisers(gold, models, kind):
    """fused nvfi_plane_regs vs the golden values and vs torch autograd of the mirrored reference formulas"""
    from nvfi_amd.utils import TVLoss
    model, meta = models[kind]
    f = model.nvfi
    model.zero_grad(set_to_none=True)
    tv = TVLoss()
    w = (8e-4, 0.7, 1.3)
    f.regs_torch_ops = True        # the mirrored reference formulas as torch ops (independent of the kernel)
    try:
        vals_t = [f.density_L1(), f.TV_loss_density(tv), f.TV_loss_app(tv)]
        (w[0] * vals_t[0] + w[1] * vals_t[1] + w[2] * vals_t[2]).backward()
    finally:
        f.regs_torch_ops = False
    ref = {k: v.copy() for k, v in named_grads(model).items() if v is not None}
    model.zero_grad(set_to_none=True)
    # the reference-signature methods run the same kernel behind autograd (value forward, gradient pass with a device-side weight)
    vals = [f.density_L1(), f.TV_loss_density(tv), f.TV_loss_app(tv)]
    assert all(v.requires_grad for v in vals)
    (w[0] * vals[0] + w[1] * vals[1] + w[2] * vals[2]).backward()
    for a, b in zip(vals, vals_t):
        np.testing.assert_allclose(a.item(), b.item(), rtol=1e-5)
    g = named_grads(model)
    for k, r in ref.items():
        if "plane" in k:
            assert relerr(g[k], r) < 1e-5, k
    model.zero_grad(set_to_none=True)
    out = f.regularizers_backward_(*w).cpu().numpy()
    np.testing.assert_allclose(out[0], gold[f"{kind}:regs:L1"][0], rtol=1e-5)
    np.testing.assert_allclose(out[1], gold[f"{kind}:regs:TVd"][0], rtol=1e-4)
    np.testing.assert_allclose(out[2], gold[f"{kind}:regs:TVa"][0], rtol=1e-4)
    g = named_grads(model)
    n = 0
    for k, r in ref.items():
        if "plane" in k:
            assert relerr(g[k], r) < 1e-5, k
            n += 1
    assert n == 9


def test_default_autograd_contract_and_arena_hook_fallback(gold, models):
    """by default the field behaves like any nn.Module under autograd - torch.autograd.grad returns the gradients and leaves .grad alone,
    parameter hooks fire; with the opt-in arena a field that carries a hook on any parameter falls back to pure autograd by itself"""
    model, meta = models["A"]
    f = model.nvfi
    assert f.accumulate_grads_inplace is False
    ps = [f.density_plane_space[0], f.vel_net.weight_net[1].weight, f.renderModule.mlp[0].weight]

    def loss_of():
        torch.manual_seed(21)
        out = _render(model, meta, gold, "A", 19.0 / 60.0, "train")
        return out[0].mean() + 0.01 * out[1].mean()

    model.zero_grad(set_to_none=True)
    gs = torch.autograd.grad(loss_of(), ps)
    assert all(g is not None and torch.isfinite(g).all() and g.abs().max() > 0 for g in gs)
    assert all(p.grad is None for p in ps)
    fired = []
    h = ps[1].register_hook(lambda g: fired.append(float(g.abs().max())))
    try:
        for mode in (False, "arena"):
            f.accumulate_grads_inplace = mode
            model.zero_grad(set_to_none=True)
            fired.clear()
            loss_of().backward()
            assert len(fired) == 1 and fired[0] > 0, (mode, fired)
            for p, g in zip(ps, gs):
                assert relerr(p.grad.cpu().numpy(), g.cpu().numpy()) < 2e-4
    finally:
        h.remove()
        f.accumulate_grads_inplace = False
        model.zero_grad(set_to_none=True)


def test_inplace_gradient_accumulation_matches_autograd(gold, models):
    """accumulate_grads_inplace (kernels add into p.grad / GradBucket views) == ordinary autograd gradients"""
    from nvfi_amd.dist import GradBucket
    model, meta = models["A"]
    f = model.nvfi

    def run(inplace):
        model.zero_grad(set_to_none=True)
        f.accumulate_grads_inplace = inplace
        bucket = GradBucket([p for g in model.get_optparam_groups() for p in g["params"]]) if inplace else None
        torch.manual_seed(21)
        out = _render(model, meta, gold, "A", 19.0 / 60.0, "train")
        loss = out[0].mean() + 0.01 * out[1].mean()
        model.vel_loss_weight = 0.5 if inplace else None
        lv = model.get_vel_loss(points=_cuda(gold["A:pde:points"]), t=_cuda(gold["A:pde:t"]))
        if not inplace:
            loss = loss + 0.5 * lv
        loss.backward()
        g = named_grads(model)
        f.accumulate_grads_inplace = False
        model.vel_loss_weight = None
        return g, float(lv)

    g0, l0 = run(False)
    g1, l1 = run(True)
    np.testing.assert_allclose(l0, l1, rtol=1e-5)
    n = 0
    for k, v in g0.items():
        if v is None or not np.any(v):
            continue
        assert relerr(g1[k], v) < 2e-4, k
        n += 1
    assert n > 30
    model.zero_grad(set_to_none=True)


def test_mask_branch(gold, models):
    """a-19: composited MaskField map (golden captured by calling the reference field directly, bypassing Renderer's reshape bug)"""
    from nvfi_amd.models import MaskField
    model, meta = models["A"]
    f = model.nvfi
    mf = MaskField(n_layer=4, n_dim=128, skips=[], mask_dim=8).cuda()
    sd = {k[len("A:mask:sd:"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("A:mask:sd:")}
    mf.load_state_dict(sd)
    f.mask_field = mf
    f.eval()
    try:
        with torch.no_grad():
            out = f(19.0 / 60.0, _cuda(gold["A:rays_o"]), _cuda(gold["A:rays_d"]), True, False)
    finally:
        f.mask_field = None
    ref = gold["A:mask:map"]
    assert out[4].shape == ref.shape
    np.testing.assert_allclose(out[4].cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * ref.max())
    assert_contract(out[0].cpu().numpy(), gold["A:render_nonkey:rgb"], "rgb", label="hip A:mask render")


def test_sh_render(gold):
    """a-17"""
    from nvfi_amd.models import TensorVMKeyframeTimeKplane
    rgb = TensorVMKeyframeTimeKplane.sh_render(_cuda(gold["sh:view"]), _cuda(gold["sh:feat"])).cpu().numpy()
    np.testing.assert_allclose(rgb, gold["sh:rgb"], rtol=1e-5, atol=1e-6)


def test_dense_alpha_and_alpha_mask(fields, models):
    """f-3: getDenseAlpha / updateAlphaMask / shrink against the oracle composed from its building blocks"""
    from oracle import oracle as orc
    model, meta = make_model("A")        # private copy: shrink() replaces the planes
    f = model.nvfi
    fs = fields["A"]
    gs = [int(g) for g in meta["gridSize"]]
    alpha, dense = f.getDenseAlpha(gs)
    pts = dense.reshape(-1, 3).cpu().numpy()
    aabb = fs.aabb
    pn = ((pts - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1).astype(np.float32)
    ts = float(meta["tmax"]) / (int(meta["num_keyframes"]) - 1)
    ref = np.zeros(pts.shape[0], np.float32)
    for t in (np.linspace(0, 59, 60) / 60)[::7]:        # a subset of the 60 frame times keeps the oracle fast
        tt = np.full((pts.shape[0], 1), np.float32(t), np.float32)
        base = (np.round(np.clip(tt / np.float32(ts), 0, int(meta["num_keyframes"]) - 1)) * np.float32(ts)).astype(np.float32)
        prev = orc.integrate_pos(fs, pn, tt, base)
        tn = (base * 2 / np.float32(meta["tmax"]) - 1).astype(np.float32)
        sig = orc.feature2density(fs, orc.density_feature(fs, np.concatenate([prev, tn], 1)))
        ref = np.maximum(ref, 1 - np.exp(-sig * np.float32(meta["stepSize"])))
        a1 = f.compute_alpha(torch.cat([dense.reshape(-1, 3), torch.full((pts.shape[0], 1), float(t), device="cuda")], -1), length=float(meta["stepSize"]))
        np.testing.assert_allclose(a1.cpu().numpy(), 1 - np.exp(-sig * np.float32(meta["stepSize"])), rtol=2e-4, atol=2e-6)
    got = alpha.reshape(-1).cpu().numpy()
    assert (got >= ref - 1e-5).all()        # the full 60-time maximum dominates the 9-time subset
    new_aabb = f.updateAlphaMask(gs)
    assert f.alphaMask is not None and tuple(f.alphaMask.alpha_volume.shape[-3:]) == (gs[2], gs[1], gs[0])
    assert (new_aabb[0] >= f.aabb[0] - 1e-5).all() and (new_aabb[1] <= f.aabb[1] + 1e-5).all()
    before = f.gridSize.tolist()
    f.shrink(new_aabb)
    after = f.gridSize.tolist()
    assert all(a <= b for a, b in zip(after, before))
    for p in list(f.density_plane_space) + list(f.app_plane_time):
        assert p.is_contiguous(memory_format=torch.channels_last)
    f.eval()
    from nvfi_amd.models import Renderer, Ray
    o, d = _cuda(np.load(__import__("os").path.join(__import__("conftest").GOLD, "hotpath.npz"))["A:rays_o"]), None
    out = f(0.3, o, _cuda(np.load(__import__("os").path.join(__import__("conftest").GOLD, "hotpath.npz"))["A:rays_d"]), True)
    assert torch.isfinite(out[0]).all()


def test_device_side_rays(gold):
    """f-2: nvfi_gen_rays == Camera.get_ray_bundle at the selected pixels"""
    import bench
    from nvfi_amd.models import Camera
    pose = bench.pose_spherical(30.0, -30.0, 4.0).cuda()
    cam = Camera(pose, 120, 160, 210.0, torch.zeros(1, 1, 3, device="cuda"), 1.0, 8.0)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    rays, ids = cam.sample_rays_device(777, generator=g)
    yy, xx = ids // 160, ids % 160
    assert torch.allclose(rays.ray_directions, cam.rays.ray_directions[yy, xx], rtol=1e-6, atol=1e-6)
    assert torch.allclose(rays.ray_origins, cam.rays.ray_origins[yy, xx])


def test_eval_driver_psnr_and_png(fields, models, tmp_path):
    """f-4: render_test_evaluation renders pose/time frames in test mode, writes r_%03d.png and reports PSNR.  Targets are the
    oracle's renders of the same frames, so the PSNR measures the HIP-vs-reference image error: > 70 dB (rgb within ~1e-4),
    and the PSNR against a noisy target agrees with the oracle's own PSNR within the 0.05 dB contract of BASELINE.json."""
    from oracle import oracle as orc
    from nvfi_amd.models import Renderer, Camera
    from nvfi_amd.utils.evaluation_utils import render_test_evaluation
    from nvfi_amd.utils import mse2psnr
    model, meta = models["A"]
    fs = fields["A"]
    model.nvfi.alphaMask = None
    H = W = 24
    focal = 30.0
    poses, times, targets, noisy = [], [], [], []
    rng = np.random.default_rng(3)
    for k, (th, t) in enumerate([(20.0, 0.30), (75.0, 19 / 60.0)]):
        c, s = np.cos(np.deg2rad(th)), np.sin(np.deg2rad(th))
        R_ = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = R_
        pose[:3, 3] = R_ @ np.array([0, 0, 4.0], np.float32) + (fs.aabb[0] + fs.aabb[1]) * 0.5
        cam = Camera(torch.from_numpy(pose), H, W, focal, None, float(meta["near"]), float(meta["far"]))
        o = cam.rays.ray_origins.reshape(-1, 3).numpy().astype(np.float32)
        d = cam.rays.ray_directions.reshape(-1, 3).numpy().astype(np.float32)
        ref = orc.render(fs, o, d, t, train=False, white_bg=bool(meta["white_background"]))
        poses.append(pose); times.append(t)
        targets.append(ref.rgb.reshape(H, W, 3))
        noisy.append(np.clip(ref.rgb.reshape(H, W, 3) + rng.normal(0, 0.05, (H, W, 3)).astype(np.float32), 0, 1))
    ren = Renderer(model, 0, 0, 2048)
    near, far = float(meta["near"]), float(meta["far"])
    res = render_test_evaluation(model, ren, poses, times, targets, H, W, focal, near, far, white_background=bool(meta["white_background"]),
                                 savedir=str(tmp_path), update_alpha_mask=False)
    assert min(res["psnr"]) > 70.0, res["psnr"]
    res2 = render_test_evaluation(model, ren, poses, times, noisy, H, W, focal, near, far, white_background=bool(meta["white_background"]),
                                  savedir=None, update_alpha_mask=False)
    for k in range(2):
        ref_psnr = mse2psnr(float(np.mean((targets[k] - noisy[k]) ** 2)))
        assert abs(res2["psnr"][k] - ref_psnr) < 0.05, (res2["psnr"][k], ref_psnr)      # BASELINE.json: PSNR within 0.05 dB
    from PIL import Image
    img = np.asarray(Image.open(str(tmp_path / "r_001.png")))
    assert img.shape == (H, W, 3) and np.array_equal(img, res["images"][1])


def test_fused_adam_matches_torch():
    """nvfi_adam_step (one launch, per-group lr, optional gradient clearing) against torch.optim.Adam on mixed layouts / sizes"""
    from nvfi_amd.optim import Adam
    torch.manual_seed(0)
    shapes = [(1, 24, 37, 41), (1, 48, 16, 9), (128, 128), (128,), (6, 128), (3,), (1, 1, 5, 7)]
    ps_a, ps_b = [], []
    for s in shapes:
        t = torch.randn(*s, device="cuda")
        if len(s) == 4 and s[1] > 1:
            t = t.contiguous(memory_format=torch.channels_last)
        ps_a.append(torch.nn.Parameter(t.clone(memory_format=torch.preserve_format)))
        ps_b.append(torch.nn.Parameter(t.clone(memory_format=torch.preserve_format)))
    ga = [dict(params=ps_a[:2], lr=0.02), dict(params=ps_a[2:], lr=1e-3)]
    gb = [dict(params=ps_b[:2], lr=0.02), dict(params=ps_b[2:], lr=1e-3)]
    oa, ob = Adam(ga, betas=(0.9, 0.99)), torch.optim.Adam(gb, betas=(0.9, 0.99))
    for it in range(7):
        for pa, pb in zip(ps_a, ps_b):
            g = torch.randn_like(pa) * (10.0 ** (it % 3 - 1))
            pa.grad = g.clone(memory_format=torch.preserve_format); pb.grad = g.clone(memory_format=torch.preserve_format)
        oa.step(zero_grad=(it % 2 == 0)); ob.step()
        for grp_a, grp_b in zip(oa.param_groups, ob.param_groups):
            grp_a["lr"] *= 0.9; grp_b["lr"] *= 0.9
        for pa, pb in zip(ps_a, ps_b):
            assert pa.stride() == pb.stride()
            np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
            if it % 2 == 0:
                assert not pa.grad.any()
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 7
        np.testing.assert_allclose(sa["state"][k]["exp_avg_sq"].cpu().numpy(), sb["state"][k]["exp_avg_sq"].cpu().numpy(), rtol=2e-6, atol=1e-12)
    ob.load_state_dict(sa)      # our state (python-int step) loads into torch.optim.Adam and vice versa
    oa.load_state_dict(ob.state_dict())
    for pa in ps_a:
        pa.grad = torch.ones_like(pa, memory_format=torch.preserve_format)
    oa.step()
    assert oa.state[ps_a[0]]["step"] == 8


def test_mask_branch_train_gradients(gold, models):
    """a-19 in train mode (tensorf_keyframe.py:673-676, 749-753): mask_map is differentiable.  Against reference goldens
    (tests/golden/make_golden_r2.py): mask_map, and the gradients of  sum(mask_map * gm) + 0.5 mean(rgb)  w.r.t. every MaskField
    parameter and - through the weights - the density / appearance planes."""
    import os
    from conftest import GOLD
    from nvfi_amd.models import MaskField
    g2 = np.load(os.path.join(GOLD, "r2.npz"))
    model, meta = models["A"]
    f = model.nvfi
    mf = MaskField(n_layer=4, n_dim=128, skips=[], mask_dim=8).cuda()
    mf.load_state_dict({k[len("A:mask_train:sd:"):]: torch.from_numpy(g2[k]) for k in g2.files if k.startswith("A:mask_train:sd:")})
    f.mask_field = mf
    f.train()
    model.zero_grad(set_to_none=True)
    try:
        torch.manual_seed(21)
        out = f(19.0 / 60.0, _cuda(gold["A:rays_o"]), _cuda(gold["A:rays_d"]), True, False)
        loss = (out[4] * _cuda(g2["A:mask_train:gm"])).sum() + 0.5 * out[0].mean()
        loss.backward()
    finally:
        f.mask_field = None
    ref = g2["A:mask_train:map"]
    np.testing.assert_allclose(out[4].detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * ref.max())
    assert_contract(out[0].detach().cpu().numpy(), g2["A:mask_train:rgb"], "rgb", label="hip A:mask_train")
    np.testing.assert_allclose(loss.item(), g2["A:mask_train:loss"].reshape(-1)[0], rtol=1e-4)
    n = 0
    for k, p in mf.named_parameters():
        e = relerr(p.grad.cpu().numpy(), g2["A:mask_train:grad:mask_field." + k])
        assert e < 5e-4, (k, e)
        n += 1
    assert n == 10
    g = named_grads(model)
    for k in ("density_plane_space.0", "density_plane_time.1", "app_plane_space.2", "basis_mat.weight"):
        e = relerr(g[k], g2["A:mask_train:grad:" + k])
        assert e < 5e-4, (k, e)
    model.zero_grad(set_to_none=True)


def test_fused_adam_reloaded_state_with_persistent_grads():
    """the cached launch table also holds the exp_avg / exp_avg_sq pointers.  With persistent gradient buffers (GradBucket:
    same p / g pointers every step) a load_state_dict() replaces the moment tensors - the next step must use the LOADED moments."""
    from nvfi_amd.optim import Adam
    torch.manual_seed(1)
    shapes = [(1, 24, 13, 11), (128, 128), (6,)]
    ps_a = [torch.nn.Parameter((torch.randn(*s, device="cuda")).contiguous(memory_format=torch.channels_last) if len(s) == 4 else torch.randn(*s, device="cuda")) for s in shapes]
    ps_b = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in ps_a]
    oa, ob = Adam([dict(params=ps_a, lr=1e-2)], betas=(0.9, 0.99)), torch.optim.Adam([dict(params=ps_b, lr=1e-2)], betas=(0.9, 0.99))
    for p in ps_a + ps_b:
        p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)       # persistent buffers: pointers never change

    def step(k):
        for pa, pb in zip(ps_a, ps_b):
            g = torch.randn_like(pa) * (k + 1)
            pa.grad.copy_(g); pb.grad.copy_(g)
        oa.step(); ob.step()

    for k in range(3):
        step(k)
    # a foreign state (3 more steps of history on other moments) is loaded into both optimisers
    sd = ob.state_dict()
    for st in sd["state"].values():
        st["exp_avg"] = st["exp_avg"] * 0.5 + 0.1
        st["exp_avg_sq"] = st["exp_avg_sq"] * 2.0 + 0.01
    import copy
    oa.load_state_dict(copy.deepcopy(sd)); ob.load_state_dict(copy.deepcopy(sd))
    for k in range(3, 6):
        step(k)
    for pa, pb in zip(ps_a, ps_b):
        np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
    sa, sb = oa.state_dict(), ob.state_dict()
    for k in sa["state"]:
        np.testing.assert_allclose(sa["state"][k]["exp_avg"].cpu().numpy(), sb["state"][k]["exp_avg"].cpu().numpy(), rtol=2e-6, atol=1e-6)      # (torch forms exp_avg with lerp: rounding differs in the last bit)
