#!/usr/bin/env python
"""Study for VERDICT r1 item 5(ii) (CPU only; uses the oracle as the fp32 reference, hence lives under tests/):

Would an fp16-input MFMA pre-pass of the PDE occupancy prefilter (models/nvfi.py:50-64: RK2 back-advection of every collocation point,
density at the warped point, alpha >= 1e-4) with a GUARD BAND and an fp32 re-evaluation of the band reproduce the exact kept mask?

Emulation of `v_mfma_f32_32x32x16_f16`: weights and layer inputs rounded to fp16, products and sums in fp32 (numpy), everything else
(encoder, SiLU, basis combination, RK2 arithmetic, density) in fp32.  For 10^6 random (x, t) per field it reports the displacement
error of the warped point, the relative deviation of alpha, the kept-mask mismatches without a band, the smallest band that removes
them all, and the fraction of points that would need the fp32 re-evaluation.

    python tests/studies/prefilter_fp16_study.py [N]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
KEYS = orc.VEL_KEYS


def h16(a):
    return a.astype(np.float16).astype(np.float32)


def vel16(fs, xt):
    """gated get_vel with fp16-input layer products"""
    q = xt.astype(np.float32)
    enc = [q]
    for k in range(3):
        enc += [np.sin(q * np.float32(2 ** k)), np.cos(q * np.float32(2 ** k))]
    h = np.concatenate(enc, 1)
    for i, k in enumerate(KEYS):
        W, b = fs.p[f"vel_net.weight_net.{k}.weight"], fs.p[f"vel_net.weight_net.{k}.bias"]
        z = h16(h) @ h16(W).T + b
        h = z / (1 + np.exp(-z)) if i < 5 else z
    w = h
    x, y, zc = q[:, 0], q[:, 1], q[:, 2]
    v = np.stack([w[:, 0] - w[:, 4] * zc + w[:, 5] * y, w[:, 1] + w[:, 3] * zc - w[:, 5] * x, w[:, 2] - w[:, 3] * y + w[:, 4] * x], 1).astype(np.float32)
    sur, lo, hi = fs.gate()
    out = (q[:, :3] < lo).any(1) | (q[:, :3] > hi).any(1)
    v[out] = 0
    return v, sur, lo, hi


def integrate16(fs, x, t, base):
    x = x.astype(np.float32).copy(); t = t.astype(np.float32).copy().reshape(-1)
    off = (t - base.reshape(-1)).astype(np.float32)
    K = int(fs.meta["num_keyframes"]); dtm = np.float32(0.5 * float(fs.meta["tmax"]) / (K - 1))
    while True:
        live = np.abs(off) > 0
        if not live.any():
            break
        m = np.minimum(np.abs(off), dtm)
        dt = (np.sign(off) * m).astype(np.float32)
        idx = np.flatnonzero(live)
        xi, ti, dti = x[idx], t[idx], dt[idx]
        v1, sur, lo, hi = vel16(fs, np.concatenate([xi, ti[:, None]], 1))
        pm = xi - 0.5 * dti[:, None] * v1
        v2, *_ = vel16(fs, np.concatenate([pm, (ti - 0.5 * dti)[:, None]], 1))
        xn = xi - dti[:, None] * v2
        if sur:
            rej = (xn < lo).any(1) | (xn > hi).any(1)
            xn[rej] = xi[rej]
        x[idx] = xn
        off[idx] = off[idx] - dti; t[idx] = ti - dti
    return x


def study(name, fs, N, rng):
    mn, mx = fs.aabb
    pts = (rng.uniform(0, 1, (N, 3)).astype(np.float32) * (mx - mn) + mn).astype(np.float32)
    t = rng.uniform(0, 1, (N, 1)).astype(np.float32)
    pn = ((pts - mn) * (2.0 / (mx - mn)) - 1).astype(np.float32)
    K = int(fs.meta["num_keyframes"]); tmax = np.float32(fs.meta["tmax"]); ts = np.float32(tmax / (K - 1))
    base = (np.round(np.clip(t / ts, 0, K - 1)) * ts).astype(np.float32)
    tn = (base * 2 / tmax - 1).astype(np.float32)
    thr = np.float32(fs.meta["alphaMask_thres"])

    def alpha_at(x):
        sig = orc.feature2density(fs, orc.density_feature(fs, np.concatenate([x, tn], 1)))
        return 1 - np.exp(-sig * np.float32(0.25))

    x32 = orc.integrate_pos(fs, pn, t, base)
    x16 = integrate16(fs, pn, t, base)
    a32, a16 = alpha_at(x32), alpha_at(x16)
    k32, k16 = a32 >= thr, a16 >= thr
    mism = np.flatnonzero(k32 != k16)
    disp = np.abs(x16 - x32).max(1)
    rel = np.abs(a16 - a32) / np.maximum(a32, 1e-30)
    # smallest relative band around the threshold (on the fp16 alpha) that contains every mismatching point
    band = float(np.abs(a16[mism] / thr - 1).max()) if mism.size else 0.0
    for delta in (band * 2, 0.05, 0.1, 0.25):
        inb = np.abs(a16 / thr - 1) <= max(delta, 1e-12)
        print(f"    band +-{max(delta, 1e-12):.3g}: {inb.mean() * 100:.3f} % of the points re-evaluated in fp32, mismatches left outside the band: {int((~inb[mism]).sum())}")
    moved = np.abs(x32 - pn).max(1)
    print(f"  {name}: N={N}, kept {k32.mean() * 100:.1f} %, max displacement {moved.max():.3e}; fp16 pre-pass: max |x16-x32| {disp.max():.2e} "
          f"(median {np.median(disp):.1e}), max rel alpha deviation where alpha > thr/10: {rel[a32 > thr / 10].max():.2e}, "
          f"kept-mask mismatches without a band: {mism.size}, smallest band containing them: +-{band:.2e}")


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rng = np.random.default_rng(0)
    orc.set_threads(8)
    fa = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_A.npz"))
    fb = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_B.npz"), shared=fa)
    fc = orc.FieldSpec.from_npz(os.path.join(GOLD, "cfg1.npz"))
    for name, fs in (("field A (bat-like, K=4)", fa), ("field B (chessboard-like, surround gate, K=16)", fb), ("cfg1 (bat 64^3, K=16, blob)", fc)):
        study(name, fs, N, rng)


if __name__ == "__main__":
    main()
