"""The x6 kernels (nvfi_amd/csrc/vel_x6.hip): the hidden layers of the velocity net with every fp32 product formed EXACTLY from three
bfloat16 terms per operand on the 16-bit matrix pipe (six MFMAs per K step, three accumulators by magnitude class, fp32 accumulation).

* accuracy: integrate_pos (tensorf_keyframe.py:575-611) through x6 and through the fp32 MFMA kernels against a float64 restatement on the
  199^3 bench field and on fields A / B: the x6 error is not larger than the fp32 kernels' (both sit on the rounding floor of the fp32
  RK2 recurrence), no outliers, bit-reproducible from run to run - 8 x 524 288 points, the size at which the build WITH packed-fp32 VALU
  code showed 2000-3000 differing points per run at two workgroups per CU (vel_x6.hip, bottom; nvfi_amd/build.py FILE_FLAGS);
* the PDE occupancy prefilter on x6 (NVFI_PDE_PREFILTER=x6): kept set, loss and gradients against the fp32 prefilter on 10^6 points of
  three fields and the bench field, ragged counts, the reference's PDE goldens, and every keep / drop decision on 10^7 points."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, relerr
from helpers import make_model

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _fields():
    import bench
    yield "bench", bench.build_scene(torch.device("cuda"), 199, 128, True)
    for kind in ("A", "B"):
        yield kind, make_model(kind)[0]


@pytest.mark.parametrize("nsteps", [1, 4, 10])
def test_x6_integrate_pos_is_as_accurate_as_the_fp32_kernels(nsteps):
    from x6_check import integrate64
    N = (1 << 16) if nsteps == 1 else ((1 << 14) if nsteps == 4 else (1 << 13))      # (the float64 reference runs in numpy on the host: 2 * nsteps evaluations per point)
    g = torch.Generator(device="cuda").manual_seed(3)
    for name, model in _fields():
        f = model.nvfi
        f.eval()
        ts = f.tmax / (f.num_keyframes - 1)
        x = torch.rand(N, 3, device="cuda", generator=g) * 1.6 - 0.8
        tt = torch.full((N, 1), float(np.float32(ts * 0.5 * nsteps * 0.999)), device="cuda")
        base = torch.zeros(N, 1, device="cuda")
        ref = integrate64(f, x.cpu().numpy(), tt.cpu().numpy()[:, 0], base.cpu().numpy()[:, 0])
        err = {}
        for mode_name, mode in (("fp32", "fp32"), ("x6", 3), ("default", False)):
            f.vel_fp16 = mode
            with torch.no_grad():
                outs = [f.integrate_pos(x.clone(), tt.clone(), base).cpu().numpy() for _ in range(3)]
            assert all(np.array_equal(outs[0], o) for o in outs[1:]), (name, mode_name, "not reproducible")
            e = np.abs(outs[0].astype(np.float64) - ref).max(1)
            err[mode_name] = (float(e.max()), float(np.sqrt((e ** 2).mean())), float((e > 20 * np.median(e) + 1e-7).mean()))
        f.vel_fp16 = False
        print(f"{name}, {nsteps} RK2 steps: max / rms error against float64: fp32 kernels {err['fp32'][0]:.2e} / {err['fp32'][1]:.2e}, "
              f"x6 {err['x6'][0]:.2e} / {err['x6'][1]:.2e}")
        assert err["x6"][0] <= 1.25 * err["fp32"][0] + 1e-9 and err["x6"][1] <= 1.05 * err["fp32"][1] + 1e-10, (name, err)
        assert err["x6"][2] <= err["fp32"][2], (name, "outliers", err)      # (a point that takes another gate branch than float64 does is an outlier of BOTH kernels)
        assert err["default"] == err["x6"], (name, err)                       # round 6: x6 is what integrate_pos runs by default


def test_x6_repeats_bit_for_bit_at_two_workgroups_per_cu():
    """the glitch hunt of round 5 (vel_x6.hip, bottom): eight runs of 524 288 points through the default launch (two workgroups per CU) must
    be identical; so must the render warp of an eval render (uniform schedule) and of a training render with a fixed jitter"""
    import bench
    m = bench.build_scene(torch.device("cuda"), 199, 128, True)
    f = m.nvfi
    f.eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    N = 1 << 19
    ts = f.tmax / (f.num_keyframes - 1)
    x = torch.rand(N, 3, device="cuda", generator=g) * 1.6 - 0.8
    tt = torch.full((N, 1), float(np.float32(ts * 0.5 * 4 * 0.999)), device="cuda")
    base = torch.zeros(N, 1, device="cuda")
    f.vel_fp16 = 3
    try:
        with torch.no_grad():
            outs = [f.integrate_pos(x.clone(), tt.clone(), base).clone() for _ in range(8)]
    finally:
        f.vel_fp16 = False
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), int((o != outs[0]).any(-1).sum())
    # the render warp (k_rk2_x6_uni): 4096 rays, ~3 x 10^5 warped samples, six renders
    model, meta = make_model("A")
    fa = model.nvfi
    fa.eval()
    gg = torch.Generator().manual_seed(1)
    o = torch.tensor([[2.6, -2.2, 2.4]]).expand(4096, 3).contiguous()
    d = (torch.rand(4096, 3, generator=gg) - 0.5) * 3.0 - o
    d = (d / d.norm(dim=-1, keepdim=True)).contiguous().cuda()
    o = o.cuda()
    with torch.no_grad():
        r = [[t.clone() for t in fa(0.41, o, d, True)[:4]] for _ in range(6)]
    assert int(fa.last_counters[3]) > 100000
    for k in range(1, 6):
        for a, b in zip(r[0], r[k]):
            assert torch.equal(a, b)
    # training renders keep the four-waves-per-tile kernel at TWO workgroups per CU (k_rk2_x6_uni<1, stash>) - the configuration the glitch lived
    # in: six forwards with one fixed jitter, 4096 rays (~3 x 10^5 warped samples each), identical bits
    fa.train()
    fa.jitter_override = torch.rand(4096, 1, generator=gg)
    try:
        with torch.no_grad():
            rt = [[t.clone() for t in fa(0.41, o, d, True)[:4]] for _ in range(6)]
    finally:
        fa.jitter_override = None
        fa.eval()
    for k in range(1, 6):
        for a, b in zip(rt[0], rt[k]):
            assert torch.equal(a, b)
    f.train()
    # ... and on the bench field: 2 x 2048 rays at a non-keyframe time, the shape of the headline step
    ob, db = bench.camera_bundle(torch.device("cuda"))
    sel = torch.randperm(ob.shape[0], generator=torch.Generator().manual_seed(3))[:4096].cuda()
    ob, db = ob[sel].contiguous(), db[sel].contiguous()
    f.jitter_override = torch.rand(4096, 1, generator=gg)
    try:
        with torch.no_grad():
            rb = [[t.clone() for t in f(19.0 / 60.0, ob, db, True)[:4]] for _ in range(6)]
    finally:
        f.jitter_override = None
        f.eval()
    assert int(f.last_counters[3]) > 100000
    for k in range(1, 6):
        for a, b in zip(rb[0], rb[k]):
            assert torch.equal(a, b)


def test_one_wave_per_tile_kernel_is_bit_identical_to_the_four_wave_kernel(tmp_path):
    """vel_x6w.hip (the default for the prefilter / integrate_pos since round 5 late: a whole 128-wide layer per wave, the epilogue in the MFMAs'
    VALU slots, no LDS exchange) forms the same products in the same order as k_rk2_x6: identical positions, ragged count, 0..19 evaluations
    per point"""
    outs = []
    for w in ("0", "1"):
        out = str(tmp_path / f"x6w{w}.npy")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "x6w_check.py"), out], env=dict(os.environ, NVFI_X6W=w, NVFI_X6W_MIN_TILES="0", NVFI_X6W_UNI="2" if w == "1" else "0"),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "repeat identical: True" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
        outs.append((np.load(out), np.load(out + ".render.npz")))
    assert np.array_equal(outs[0][0], outs[1][0]), int((outs[0][0] != outs[1][0]).any(1).sum())
    # the render warp: eval render and training render (same stash -> same forward; the gradient through the fp32 adjoint differs by its atomics only)
    for k in ("rgb", "depth", "train_rgb"):
        assert np.array_equal(outs[0][1][k], outs[1][1][k]), k
    assert relerr(outs[1][1]["g"], outs[0][1]["g"]) < 2e-5


def _run(tmp_path, mode, extra=(), n=262144, **switches):
    out = str(tmp_path / f"x6_{mode}_{n}_{'_'.join(f'{k}{v}' for k, v in switches.items())}.npz")
    env = dict(os.environ, NVFI_PDE_PREFILTER=mode, **{k: str(v) for k, v in switches.items()})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pre16_check.py"), out, str(n), *extra], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("n,extra,nt", [(1000000, (), 0), (1000000, (), 1), (262144, ("--bench",), 0), (262144, ("--bench",), 1), (37, (), 0), (1000, (), 1)])
def test_x6_prefilter_keeps_the_fp32_set(tmp_path, n, extra, nt):
    """both prefilters decide `alpha >= alphaMask_thres` on positions that differ by rounding (another summation order of the same fp32
    products): a point would have to sit within ~1e-7 of the threshold to flip - at most a handful among 10^6"""
    # nt = 0: the one-wave-per-tile kernel (vel_x6w.hip) at every size; 1: the four-waves-per-tile kernel (round 6: the default up to 4096 tiles)
    sw = dict(NVFI_X6W=1, NVFI_X6W_MIN_TILES=0) if nt == 0 else dict(NVFI_X6W=0)
    a, b = _run(tmp_path, "fp32", extra, n), _run(tmp_path, "x6", extra, n, **sw)
    for name in ("A", "B", "cfg1") + (("bench",) if extra else ()):
        ka, kb = a[f"{name}:kept"], b[f"{name}:kept"]
        flips = int((ka != kb).sum())
        print(f"{name}: kept {int(ka.sum())} / {int(kb.sum())} of {ka.size}, flips {flips}; get_vel_loss {float(a[f'{name}:ms']):.3f} ms fp32 -> {float(b[f'{name}:ms']):.3f} ms x6")
        assert flips <= 2, (name, flips)
        np.testing.assert_allclose(float(a[f"{name}:loss"]), float(b[f"{name}:loss"]), rtol=1e-4 if flips else 1e-5)
        if not flips:
            for k in a.files:
                if k.startswith(f"{name}:grad:"):
                    assert relerr(b[k], a[k]) < 2e-5, k


def test_x6_prefilter_flip_count_on_ten_million_points(tmp_path):
    n = 10 * (1 << 20)
    out = {}
    for mode in ("fp32", "x6"):
        path = str(tmp_path / f"flips_{mode}.npz")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pre16_check.py"), path, "--flips", str(n)], env=dict(os.environ, NVFI_PDE_PREFILTER=mode),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out[mode] = np.load(path)
    a, b = np.unpackbits(out["fp32"]["kept"]), np.unpackbits(out["x6"]["kept"])
    flips = int((a != b).sum())
    print(f"bench field: {int(out['fp32']['n'])} points, kept {int(a.sum())}; decisions that differ between the fp32 and the x6 prefilter: {flips}")
    assert int(out["fp32"]["n"]) >= 10 ** 7 and int(a.sum()) > 10 ** 6
    assert flips <= 8, flips          # (two fp32 summation orders of one quantity: a point within ~1e-7 of the threshold may land on either side)


def test_x6_prefilter_under_the_reference_goldens():
    env = dict(os.environ, NVFI_PDE_PREFILTER="x6")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        os.path.join(ROOT, "tests", "test_gpu_fullsize_chessboard.py"), os.path.join(ROOT, "tests", "test_gpu_fullsize.py"), "-k", "pde"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
