"""Opt-in fp16-input MFMA inference mode of the velocity field (nvfi_field_desc.vel_fp16; the reference's counterpart is its autocast
switch --disable_fp32, train_nvfi.py:96,144).  Every no-grad back-advection - integrate_pos (tensorf_keyframe.py:575-611, as
train_segm.py:150-166 uses it), the warp of eval-mode renders, getDenseAlpha - evaluates VelBasis with weights and layer inputs rounded to
binary16 and fp32 accumulation.  Checked against the oracle switched to the SAME arithmetic (orc.set_vel_fp16), so the comparison is
digit-level; training renders and gradients must not change at all."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from helpers import assert_contract, make_model, named_grads


def test_oracle_f16_round_is_ieee_binary16():
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(500).astype(np.float32) * s for s in (1e-8, 1e-6, 1e-4, 1e-2, 1, 100, 70000)]
                       + [np.array([0, 65504, 65519.9, 65520, 6.1e-5, 5.96e-8, 2.98e-8, 8.9e-8, -1.5, 1 + 2 ** -11, 1 + 3 * 2 ** -11], np.float32)])
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(orc.f16_round(x), ref)


def test_oracle_fp16_mode_is_a_small_perturbation_of_fp32(fields):
    """the oracle's fp16 arithmetic switch changes integrate_pos by fp16-sized amounts and is off again afterwards"""
    from oracle import oracle as orc
    fs = fields["A"]
    rng = np.random.default_rng(1)
    x = (rng.random((512, 3), dtype=np.float32) * 1.6 - 0.8)
    t = rng.random((512, 1), dtype=np.float32) * 0.7
    base = np.zeros_like(t)
    ref = orc.integrate_pos(fs, x, t, base)
    orc.set_vel_fp16(True)
    try:
        h = orc.integrate_pos(fs, x, t, base)
    finally:
        orc.set_vel_fp16(False)
    d = np.abs(h - ref).max()
    assert 1e-7 < d < 2e-2, d
    assert np.array_equal(orc.integrate_pos(fs, x, t, base), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["A", "B"])
def test_gpu_integrate_pos_fp16_matches_the_fp16_oracle(fields, kind):
    from oracle import oracle as orc
    model, meta = make_model(kind)
    f = model.nvfi
    f.eval()
    rng = np.random.default_rng(7)
    N = 3000
    x = (rng.random((N, 3), dtype=np.float32) * 2.1 - 1.05)            # some points outside the velocity gate / the surround box
    t = rng.random((N, 1), dtype=np.float32) * float(f.tmax) * 1.3     # up to ~20 RK2 steps, extrapolation included
    base = np.where(rng.random((N, 1)) < 0.5, 0.0, np.round(t / (float(f.tmax) / (f.num_keyframes - 1))) * (float(f.tmax) / (f.num_keyframes - 1))).astype(np.float32)
    xg, tg, bg = (torch.from_numpy(a).cuda() for a in (x, t, base))
    with torch.no_grad():
        ref32 = f.integrate_pos(xg.clone(), tg.clone(), bg).cpu().numpy()
        f.vel_fp16 = True
        got = f.integrate_pos(xg.clone(), tg.clone(), bg).cpu().numpy()
        f.vel_fp16 = False
    orc.set_vel_fp16(True)
    try:
        ref16 = orc.integrate_pos(fields[kind], x, t, base)
    finally:
        orc.set_vel_fp16(False)
    # same arithmetic: what is left is the fp32 summation order (and a gate decision within rounding of a face: counted, bounded)
    err = np.abs(got - ref16).max(axis=1)
    assert np.mean(err > 2e-5) < 2e-3, (np.mean(err > 2e-5), err.max())
    assert np.median(err) < 2e-6
    # and it IS a different arithmetic from the fp32 path, by fp16-sized amounts
    d = np.abs(got - ref32).max(axis=1)
    assert d.max() > 1e-6 and np.median(d) < 1e-3
    print(f"[{kind}] fp16 integrate_pos vs fp16 oracle: median {np.median(err):.2e}, max {err.max():.2e}; vs the fp32 path: median {np.median(d):.2e}, max {d.max():.2e}")


@pytest.mark.gpu
def test_gpu_eval_render_fp16_matches_the_fp16_oracle(gold, fields):
    """eval-mode renders at a non-keyframe and an extrapolated time with the fp16 warp against the oracle in the same arithmetic; a
    training render with the switch on is bit-identical to the switch off (training stays fp32)"""
    from oracle import oracle as orc
    from nvfi_amd.models import Ray, Renderer
    model, meta = make_model("A")
    f = model.nvfi
    ren = Renderer(model, 0, 0, 2048)
    o, d = gold["A:rays_o"], gold["A:rays_d"]
    rays = Ray(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), 0, 1)
    for t in (19.0 / 60.0, 50.0 / 60.0):
        f.vel_fp16 = True
        out = ren.render(t, rays, white_background=True, mode="test")
        f.vel_fp16 = False
        out32 = ren.render(t, rays, white_background=True, mode="test")
        orc.set_vel_fp16(True)
        try:
            ref = orc.render(fields["A"], o, d, t, train=False, white_bg=True)
        finally:
            orc.set_vel_fp16(False)
        assert_contract(out[0].cpu().numpy(), ref.rgb, "rgb", label=f"fp16 eval t={t:.3f}")
        assert_contract(out[1].cpu().numpy(), ref.depth, "depth", label=f"fp16 eval t={t:.3f}")
        assert not torch.equal(out[0], out32[0])          # the switch does something ...
        assert (out[0] - out32[0]).abs().max() < 2e-2     # ... of fp16 size
    # training: unaffected
    torch.manual_seed(3)
    u = torch.rand(o.shape[0], 1)
    res = []
    for sw in (False, True):
        f.vel_fp16 = sw
        f.jitter_override = u
        model.zero_grad(set_to_none=True)
        out = ren.render(19.0 / 60.0, rays, white_background=True, mode="train")
        out[0].square().mean().backward()
        res.append((out[0].detach().clone(), f.vel_net.weight_net[1].weight.grad.detach().clone()))
    f.vel_fp16 = False
    f.jitter_override = None
    assert torch.equal(res[0][0], res[1][0])
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-4, atol=1e-9)      # (atomics: order, not arithmetic)


@pytest.mark.gpu
def test_gpu_dense_alpha_fp16_is_close_to_fp32():
    """getDenseAlpha (60 frame times x G^3 points through the RK2 warp) with the fp16 switch: the occupancy volume moves by fp16-sized amounts"""
    model, meta = make_model("A")
    f = model.nvfi
    f.eval()
    with torch.no_grad():
        a32, _ = f.getDenseAlpha((20, 18, 16))
        f.vel_fp16 = True
        a16, _ = f.getDenseAlpha((20, 18, 16))
        f.vel_fp16 = False
    d = (a32 - a16).abs()
    assert float(d.max()) < 5e-2 and float(d.mean()) < 1e-3
    assert float(((a32 > f.alphaMask_thres) != (a16 > f.alphaMask_thres)).float().mean()) < 5e-3


@pytest.mark.gpu
def test_gpu_split16_inference_meets_the_fp32_contract(gold, fields):
    """vel_fp16 = 2: fp32 products emulated by two binary16 terms per operand (three fp16 MFMAs, ~2^-21 relative per product).  Unlike the
    one-term mode this one is held to the FP32 bar: integrate_pos against the plain fp32 oracle, eval renders against the REFERENCE's own
    goldens under the 1e-4 contract."""
    from oracle import oracle as orc
    model, meta = make_model("A")
    f = model.nvfi
    f.eval()
    rng = np.random.default_rng(9)
    N = 3000
    x = (rng.random((N, 3), dtype=np.float32) * 2.1 - 1.05)
    t = rng.random((N, 1), dtype=np.float32) * float(f.tmax) * 1.3
    base = np.zeros_like(t)
    xg, tg, bg = (torch.from_numpy(a).cuda() for a in (x, t, base))
    with torch.no_grad():
        ref32 = f.integrate_pos(xg.clone(), tg.clone(), bg).cpu().numpy()
        f.vel_fp16 = 2
        got = f.integrate_pos(xg.clone(), tg.clone(), bg).cpu().numpy()
    ref = orc.integrate_pos(fields["A"], x, t, base)                     # the plain fp32 oracle
    err = np.abs(got - ref).max(axis=1)
    e32 = np.abs(ref32 - ref).max(axis=1)
    print(f"split16 integrate_pos vs fp32 oracle: median {np.median(err):.2e}, max {err.max():.2e}   (fp32 MFMA path: median {np.median(e32):.2e}, max {e32.max():.2e})")
    assert np.mean(err > 2e-5) < 2e-3 and np.median(err) < 5e-6
    f.vel_fp16 = False
    # every eval-render and integrate_pos golden of the REFERENCE (both fields: keyframe, non-keyframe, extrapolated, transfer, flipped
    # background, alpha mask) under the unchanged 1e-4 contract with the switch on for the whole process
    import subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, NVFI_VEL_FP16="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_cfg1.py"),
                        "-k", "render_eval or integrate_pos or cfg1_matches"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    print(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["A", "B"])
def test_gpu_training_render_with_fp16_forward_warp_matches_the_fp16_oracle(gold, fields, kind):
    """round 4, opt-in `field.vel_fp16_train` (nvfi_field_desc.vel_fp16 bit 2): the velocity warp of a TRAINING render evaluates VelBasis
    forward with fp16-input MFMAs (fp32 accumulation, fp32 stashes); the adjoint and the weight gradients stay fp32 on those stashes - the
    arithmetic of the reference's autocast forward (--disable_fp32, train_nvfi.py:96,144) with an fp32 backward, which is exactly what the
    oracle does under set_vel_fp16 (fp16 forward, reverse mode in fp32 on the stashed pre-activations).  Render under the 1e-4 contract,
    every gradient - planes through the warped positions, render MLP, velocity net through the RK2 adjoint - within 1e-3."""
    from oracle import oracle as orc
    from conftest import assert_grad
    model, meta = make_model(kind)
    f = model.nvfi
    fs = fields[kind]
    rng = np.random.default_rng(17)
    R = 384
    o, d = gold[f"{kind}:rays_o"][:R], gold[f"{kind}:rays_d"][:R]
    R = o.shape[0]
    u = rng.uniform(0, 1, (R, 1)).astype(np.float32)
    tg = rng.uniform(0, 1, (R, 3)).astype(np.float32)
    t = 19.0 / 60.0

    def run(fp16):
        f.train()
        f.vel_fp16_train = fp16
        model.zero_grad(set_to_none=True)
        f.jitter_override = torch.from_numpy(u.copy())
        try:
            out = f(t, torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), True)
        finally:
            f.jitter_override = None
            f.vel_fp16_train = False       # (the backward plans its workspace with the forward's setting, whatever the field says by then)
        (torch.nn.functional.mse_loss(out[0], torch.from_numpy(tg).cuda()) + 0.01 * out[1].mean()).backward()
        return [x.detach().cpu().numpy() for x in out[:3]], named_grads(model)

    out32, g32 = run(False)
    out16, g16 = run(True)
    orc.set_vel_fp16(True)
    try:
        ref = orc.render(fs, o, d, t, u=u, train=True, white_bg=True, keep_ctx=True)
        gref = ref.backward(fs, g_rgb=2 * (ref.rgb - tg) / (R * 3), g_depth=np.full(R, 0.01 / R, np.float32))
        ref.free()
    finally:
        orc.set_vel_fp16(False)
    assert_contract(out16[0], ref.rgb, "rgb", label=f"fp16-forward training warp {kind}")
    assert_contract(out16[1], ref.depth, "depth", label=f"fp16-forward training warp {kind}")
    n = 0
    for pn in ("density_plane_space.0", "app_plane_space.1", "app_plane_time.0", "basis_mat.weight", "renderModule.mlp.0.weight",
               "vel_net.weight_net.1.weight", "vel_net.weight_net.4.0.weight", "vel_net.weight_net.6.0.weight", "vel_net.weight_net.7.0.bias"):
        assert_grad(g16[pn], gref[pn], 1e-3, f"fp16train:{kind}:{pn}")
        n += 1
    # and it IS another arithmetic than the fp32 path (fp16-sized differences in the velocity-net gradients), which must be untouched by the switch
    dv = np.abs(g16["vel_net.weight_net.4.0.weight"] - g32["vel_net.weight_net.4.0.weight"]).max() / np.abs(g32["vel_net.weight_net.4.0.weight"]).max()
    assert 1e-6 < dv < 5e-2, dv
    print(f"[{kind}] fp16-forward training warp: {n} gradients within 1e-3 of the fp16 oracle; velocity-net gradient differs from the fp32 path by {dv:.2e} (max-norm)")


@pytest.mark.gpu
def test_fp16_input_kernels_repeat_bit_for_bit():
    """round 6: pre16.hip carries 16-bit MFMAs for eight waves per workgroup - two per SIMD, the occupancy at which the
    x6 kernels once glitched with packed-fp32 VALU code beside them.  The unit is compiled with the same fence now
    (nvfi_amd/build.py: -fno-slp-vectorize + check_no_packed_f32) and gets the same run-to-run identity test as x6: 8 x 524 288 points
    through k_rk2_inf16 per-point (vel_fp16 = 1 and the two-term mode 2), six eval renders (uniform schedule), six training forwards
    with the fp16 warp (stash variant), and the fp16band / split16band PDE prefilters (k_rk2_pre16) in a subprocess."""
    import os
    import subprocess
    import sys
    import bench
    from conftest import ROOT
    from helpers import make_model
    m = bench.build_scene(torch.device("cuda"), 199, 128, True)
    f = m.nvfi
    f.eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    N = 1 << 19
    ts = f.tmax / (f.num_keyframes - 1)
    x = torch.rand(N, 3, device="cuda", generator=g) * 1.6 - 0.8
    tt = torch.full((N, 1), float(np.float32(ts * 0.5 * 4 * 0.999)), device="cuda")
    base = torch.zeros(N, 1, device="cuda")
    for mode in (1, 2):
        f.vel_fp16 = mode
        try:
            with torch.no_grad():
                outs = [f.integrate_pos(x.clone(), tt.clone(), base).clone() for _ in range(8)]
        finally:
            f.vel_fp16 = False
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), (mode, int((o != outs[0]).any(-1).sum()))
    model, meta = make_model("A")
    fa = model.nvfi
    gg = torch.Generator().manual_seed(1)
    o = torch.tensor([[2.6, -2.2, 2.4]]).expand(4096, 3).contiguous()
    d = (torch.rand(4096, 3, generator=gg) - 0.5) * 3.0 - o
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous().cuda()
    o = o.cuda()
    fa.eval()
    fa.vel_fp16 = 1
    try:
        with torch.no_grad():
            r = [[t.clone() for t in fa(0.41, o, d, True)[:4]] for _ in range(6)]
    finally:
        fa.vel_fp16 = False
    assert int(fa.last_counters[3]) > 100000
    for k in range(1, 6):
        for a, b in zip(r[0], r[k]):
            assert torch.equal(a, b)
    fa.train()
    fa.vel_fp16_train = True
    fa.jitter_override = torch.rand(4096, 1, generator=gg)
    try:
        with torch.no_grad():
            rt = [[t.clone() for t in fa(0.41, o, d, True)[:4]] for _ in range(6)]
    finally:
        fa.jitter_override = None
        fa.vel_fp16_train = False
        fa.eval()
    for k in range(1, 6):
        for a, b in zip(rt[0], rt[k]):
            assert torch.equal(a, b)
    for mode in ("fp16band", "split16band"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pre16_check.py"), "-", "--repeat", str(N)], env=dict(os.environ, NVFI_PDE_PREFILTER=mode),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "repeat identical: True" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
