#!/usr/bin/env python
"""Study for VERDICT r3 item 6 (CPU only, numpy): can the opt-in `split16band` prefilter (fp32 products emulated by two binary16 terms per
MFMA operand, ~2^-21 relative per product; pre16.hip) be PROVEN to keep exactly the fp32 prefilter's set, with a band that is still useful?

A proof needs, per collocation point, a rigorous bound on (a) the position error d of the split evaluation against the fp32 evaluation after
the RK2 back-advection (reference models/tensorf_keyframe.py:575-611 around models/velocity_field.py:54-98) and (b) the change of alpha that
d can cause (reference models/nvfi.py:50-64); the band |alpha / thres - 1| must be at least (b).  Rigorous means: no cancellation between
rounding errors may be assumed, so an error e_l at the input of layer l+1 becomes at most |W_{l+1}| e_l (entry-wise absolute values) behind
it, times max|SiLU'| = 1.0998.  This script evaluates that recurrence A POSTERIORI - with the actual activations of actual points, the
sharpest form available (the extra-MFMA-column idea of the verdict: one |W| mat-vec per layer) - and compares it with the measured error.

For each field it prints: the measured position error of the emulated split evaluation (median / max), the rigorous a-posteriori bound
(median / max), the Lipschitz bound of log(alpha) along the warped position from the density planes, the band width the two imply, and the
fraction of candidate points that such a band would send to the fp32 re-evaluation.

    python tests/studies/split16_bound_study.py [N]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
KEYS = orc.VEL_KEYS
U_SPLIT = 2.0 ** -21        # relative error of one emulated product (three binary16 MFMAs: two representation errors + the dropped lo*lo term)
U_ACC = 2.0 ** -24          # one fp32 rounding of the accumulation
SILU_D_MAX = 1.0998         # max |SiLU'|


def split16(a):
    """a ~ hi + lo / 2048 with hi, lo binary16 (pre16.hip)"""
    hi = a.astype(np.float16).astype(np.float32)
    lo = ((a - hi) * np.float32(2048)).astype(np.float16).astype(np.float32)
    return hi, lo


def encode(q):
    enc = [q]
    for k in range(3):
        enc += [np.sin(q * np.float32(2 ** k)), np.cos(q * np.float32(2 ** k))]
    return np.concatenate(enc, 1).astype(np.float32)


def net(fs, q, mode):
    """weight_net at q (N, 4).  mode 'f32': float64 reference of the fp32 path; 'split': the emulated split products;
    'bound': also returns the rigorous a-posteriori error bound of the 6 outputs (split against exact arithmetic)"""
    h = encode(q)
    e = np.zeros_like(h, dtype=np.float64)                       # |error| of the layer input (encoder arithmetic is shared fp32: 0)
    for i, k in enumerate(KEYS):
        W, b = fs.p[f"vel_net.weight_net.{k}.weight"], fs.p[f"vel_net.weight_net.{k}.bias"]
        if mode == "f32":
            z = h.astype(np.float64) @ W.T.astype(np.float64) + b
        else:
            xh, xl = split16(h.astype(np.float32))
            wh, wl = split16(W)
            z = (xh @ wh.T + (xh @ wl.T + xl @ wh.T) / np.float32(2048) + b).astype(np.float32)
        if mode == "bound":
            aW = np.abs(W).astype(np.float64)
            K = W.shape[1]
            # propagated input error + per-product error of the emulation + accumulation roundings (K terms, sequential bound)
            e = e @ aW.T + (U_SPLIT + K * U_ACC) * (np.abs(h).astype(np.float64) @ aW.T + np.abs(b))
            if i < 5:
                e = e * SILU_D_MAX
        h = (z / (1 + np.exp(-z))) if i < 5 else z
    return (h, e) if mode == "bound" else h


def vel(fs, q, mode):
    r = net(fs, q, mode)
    w, e = r if mode == "bound" else (r, None)
    x, y, z = q[:, 0], q[:, 1], q[:, 2]
    v = np.stack([w[:, 0] - w[:, 4] * z + w[:, 5] * y, w[:, 1] + w[:, 3] * z - w[:, 5] * x, w[:, 2] - w[:, 3] * y + w[:, 4] * x], 1)
    if mode != "bound":
        return v
    ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
    ev = np.stack([e[:, 0] + e[:, 4] * az + e[:, 5] * ay, e[:, 1] + e[:, 3] * az + e[:, 5] * ax, e[:, 2] + e[:, 3] * ay + e[:, 4] * ax], 1)
    return v, ev


def lipschitz_of_velocity(fs):
    """rigorous Lipschitz bound of q -> v(q) in the infinity norm: products of the layers' absolute row sums (encoder: frequencies up to 4)"""
    L = 1.0 + 2 * (1 + 2 + 4)            # d(enc)/dq: raw + sin/cos of three octaves
    for i, k in enumerate(KEYS):
        L *= np.abs(fs.p[f"vel_net.weight_net.{k}.weight"]).sum(1).max() * (SILU_D_MAX if i < 5 else 1.0)
    return L * 3.0                        # basis combination with |x|, |y|, |z| <= 1


def study(name, fs, N, rng):
    K = int(fs.meta["num_keyframes"]); tmax = float(fs.meta["tmax"]); ts = tmax / (K - 1); dtm = 0.5 * ts
    x = rng.uniform(-0.9, 0.9, (N, 3)).astype(np.float32)
    t = rng.uniform(0, tmax, (N,)).astype(np.float32)
    base = (np.rint(np.clip(t / ts, 0, K - 1)) * ts).astype(np.float32)
    steps = np.ceil(np.abs(t - base) / dtm - 1e-6).astype(int)
    # one RK2 step of the longest kind for everybody (the bound of an n-step trajectory is at least n times this one's growth)
    dt = np.float32(dtm)
    q = np.concatenate([x, t[:, None]], 1).astype(np.float32)
    v1_ref, v1_spl = vel(fs, q, "f32"), vel(fs, q, "split")
    _, ev1 = vel(fs, q, "bound")
    mid = (x - 0.5 * dt * v1_spl).astype(np.float32)
    qm = np.concatenate([mid, (t - 0.5 * dt)[:, None]], 1).astype(np.float32)
    v2_spl = vel(fs, qm, "split")
    mid_ref = x - 0.5 * dt * v1_ref
    v2_ref = vel(fs, np.concatenate([mid_ref, (t - 0.5 * dt)[:, None]], 1).astype(np.float32), "f32")
    _, ev2 = vel(fs, qm, "bound")
    Lv = lipschitz_of_velocity(fs)
    meas = np.abs((x - dt * v2_spl) - (x - dt * v2_ref)).max(1)
    # rigorous: error of v2 = own evaluation error + Lipschitz(v) * error of the midpoint (0.5 dt * ev1)
    bound = dt * (ev2.max(1) + Lv * 0.5 * dt * ev1.max(1))
    bound_local = dt * ev2.max(1)          # even ignoring the midpoint's propagation
    # sensitivity of log(alpha) to the warped position: alpha ~ exp(feat - shift) * dist near the threshold, feat = sum over 3 plane pairs
    # and 24 channels of space(x, y) * time(z, t): |d feat / d pos| <= sum_pairs sum_c (|d space| |time| + |space| |d time|), with the
    # bilinear gradient bounded by the largest texel difference times (G - 1) / 2 per unit of normalised coordinate
    Lf = 0.0
    for i in range(3):
        S = fs.p[f"density_plane_space.{i}"][0].astype(np.float64); T = fs.p[f"density_plane_time.{i}"][0].astype(np.float64)
        gs = max(np.abs(np.diff(S, axis=1)).max() * (S.shape[1] - 1) / 2, np.abs(np.diff(S, axis=2)).max() * (S.shape[2] - 1) / 2)
        gt = np.abs(np.diff(T, axis=2)).max() * (T.shape[2] - 1) / 2
        Lf += (gs * np.abs(T).max(axis=(1, 2)) + np.abs(S).max(axis=(1, 2)) * gt).sum() if np.ndim(gs) else (gs * np.abs(T).max() + np.abs(S).max() * gt) * S.shape[0]
    band = Lf * np.median(bound) * steps.mean()
    print(f"{name}: K={K}, mean RK2 steps {steps.mean():.1f}; ||v||_Lip <= {Lv:.3g}; |d log(alpha)/d pos| <= {Lf:.3g}")
    print(f"   position error of ONE split RK2 step vs fp32: measured median {np.median(meas):.2e} max {meas.max():.2e}")
    print(f"   rigorous a-posteriori bound:                  median {np.median(bound):.2e} max {bound.max():.2e}   (without midpoint propagation: median {np.median(bound_local):.2e})")
    print(f"   bound / measured (medians): {np.median(bound) / max(np.median(meas), 1e-30):.1e} (local part alone: {np.median(bound_local) / max(np.median(meas), 1e-30):.1e})")
    print(f"   band on alpha/thres that the bound implies (x mean steps): {band:.3g}  vs the 1e-3 that split16band ships (empirical: 0 flips in 1e6 points per field)")
    return dict(meas=float(np.median(meas)), bound=float(np.median(bound)), bound_local=float(np.median(bound_local)), band=float(band), Lv=float(Lv), Lf=float(Lf))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(7)
    fa = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_A.npz"))
    fb = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_B.npz"), shared=fa)
    out = {}
    for name, fs in (("A", fa), ("B", fb)):
        out[name] = study(name, fs, N, rng)
    return out


if __name__ == "__main__":
    main()
