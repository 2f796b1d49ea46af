"""ctypes front-end of the CPU oracle (oracle/nvfi_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package nvfi_amd never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

fp = C.POINTER(C.c_float)


class OrcField(C.Structure):
    _fields_ = [
        ("G", C.c_int32 * 3), ("K", C.c_int32), ("Cd", C.c_int32), ("Ca", C.c_int32), ("app_dim", C.c_int32),
        ("n_samples", C.c_int32), ("use_vel", C.c_int32), ("gate_sur", C.c_int32), ("has_amask", C.c_int32), ("shading", C.c_int32),
        ("am_dims", C.c_int32 * 3),
        ("aabb", C.c_float * 6), ("near_", C.c_float), ("far_", C.c_float), ("step_size", C.c_float),
        ("density_shift", C.c_float), ("distance_scale", C.c_float), ("weight_thres", C.c_float),
        ("alpha_thres", C.c_float), ("tmax", C.c_float),
        ("gate_lo", C.c_float * 3), ("gate_hi", C.c_float * 3),
        ("dps", fp * 3), ("dpt", fp * 3), ("aps", fp * 3), ("apt", fp * 3),
        ("basis", fp), ("rW", fp * 3), ("rb", fp * 3),
        ("vW", fp * 6), ("vb", fp * 6), ("aW", fp * 6), ("ab", fp * 6),
        ("amask", fp),
    ]


class OrcGrads(C.Structure):
    _fields_ = [
        ("dps", fp * 3), ("dpt", fp * 3), ("aps", fp * 3), ("apt", fp * 3),
        ("basis", fp), ("rW", fp * 3), ("rb", fp * 3),
        ("vW", fp * 6), ("vb", fp * 6), ("aW", fp * 6), ("ab", fp * 6),
    ]


def build(force=False):
    so = os.path.join(HERE, "liborc.so")
    src = os.path.join(HERE, "nvfi_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_render_fwd.restype = C.c_void_p
        L.orc_pde_loss.restype = C.c_float
        for n in ("orc_density_L1", "orc_tv_density", "orc_tv_app"):
            getattr(L, n).restype = C.c_float
        L.orc_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(fp) if a is not None else fp()


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


VEL_KEYS = ["1", "3.0", "4.0", "5.0", "6.0", "7.0"]  # Sequential indices of the 6 Linear layers
PARAM_NAMES = (
    [f"density_plane_space.{i}" for i in range(3)] + [f"density_plane_time.{i}" for i in range(3)]
    + [f"app_plane_space.{i}" for i in range(3)] + [f"app_plane_time.{i}" for i in range(3)]
    + ["basis_mat.weight"]
    + [f"renderModule.mlp.{i}.{wb}" for i in (0, 2, 4) for wb in ("weight", "bias")]
    + [f"vel_net.weight_net.{k}.{wb}" for k in VEL_KEYS for wb in ("weight", "bias")]
    + [f"vel_net.a_weight_net.{k}.{wb}" for k in VEL_KEYS for wb in ("weight", "bias")]
)


class FieldSpec:
    """Reference-layout parameters (numpy, NCHW / (out,in)) + the config scalars that reach the hot path."""

    def __init__(self, params, meta):
        self.p = {k: _f32(v) for k, v in params.items()}
        self.meta = dict(meta)
        self.amask = None

    @staticmethod
    def from_npz(path, shared=None, prefix=""):
        z = np.load(path) if isinstance(path, str) else path
        ps, pm = prefix + "sd:nvfi.", prefix + "meta:"
        params = {k[len(ps):]: z[k] for k in z.files if k.startswith(ps)}
        if shared is not None:
            for k, v in shared.p.items():
                params.setdefault(k, v)
        meta = {k[len(pm):]: z[k] for k in z.files if k.startswith(pm)}
        meta = {k: (v.item() if v.ndim == 0 else v) for k, v in meta.items()}
        return FieldSpec(params, meta)

    # ---- derived scalars
    @property
    def aabb(self):
        return _f32(self.meta["aabb"]).reshape(2, 3)

    def gate(self):
        if self.meta.get("use_sur", 0):
            b = _f32(self.meta["sur_bounds"]).reshape(2, 3)
            return 1, b[0].copy(), b[1].copy()
        eps = float(self.meta.get("eps", 0.03))
        lo = np.full(3, np.float32(-1 + eps), np.float32)
        hi = np.full(3, np.float32(1 - eps), np.float32)
        return 0, lo, hi

    def c_field(self, use_vel=True, n_samples=None, step_size=None):
        f = OrcField()
        m = self.meta
        G = [int(g) for g in m["gridSize"]]
        f.G[:] = G
        f.K = int(m["num_keyframes"])
        f.Cd = self.p["density_plane_space.0"].shape[1]
        f.Ca = self.p["app_plane_space.0"].shape[1]
        f.app_dim = self.p["basis_mat.weight"].shape[0]
        f.n_samples = int(n_samples if n_samples is not None else m["nSamples"])
        f.use_vel = int(use_vel)
        gs, lo, hi = self.gate()
        f.gate_sur = gs
        f.gate_lo[:] = lo.tolist()
        f.gate_hi[:] = hi.tolist()
        f.aabb[:] = self.aabb.reshape(-1).tolist()
        f.near_, f.far_ = float(m["near"]), float(m["far"])
        f.step_size = float(step_size if step_size is not None else m["stepSize"])
        f.density_shift = float(m["density_shift"])
        f.distance_scale = float(m["distance_scale"])
        f.weight_thres = float(m["rayMarch_weight_thres"])
        f.alpha_thres = float(m["alphaMask_thres"])
        f.tmax = float(m["tmax"])
        for i in range(3):
            f.dps[i] = _p(self.p[f"density_plane_space.{i}"])
            f.dpt[i] = _p(self.p[f"density_plane_time.{i}"])
            f.aps[i] = _p(self.p[f"app_plane_space.{i}"])
            f.apt[i] = _p(self.p[f"app_plane_time.{i}"])
        f.basis = _p(self.p["basis_mat.weight"])
        f.shading = 1 if str(m.get("shadingMode", "MLP_PE")) == "SH" else 0      # SH: no render MLP (tensorf_base.py:196-197)
        for i, k in enumerate((0, 2, 4) if not f.shading else ()):
            f.rW[i] = _p(self.p[f"renderModule.mlp.{k}.weight"])
            f.rb[i] = _p(self.p[f"renderModule.mlp.{k}.bias"])
        has_vel = "vel_net.weight_net.1.weight" in self.p      # radiance-only fields (use_vel: False) carry no velocity nets
        if not has_vel:
            f.use_vel = 0
        for i, k in enumerate(VEL_KEYS if has_vel else []):
            f.vW[i] = _p(self.p[f"vel_net.weight_net.{k}.weight"])
            f.vb[i] = _p(self.p[f"vel_net.weight_net.{k}.bias"])
            f.aW[i] = _p(self.p[f"vel_net.a_weight_net.{k}.weight"])
            f.ab[i] = _p(self.p[f"vel_net.a_weight_net.{k}.bias"])
        if self.amask is not None:
            f.has_amask = 1
            D, H, W = self.amask.shape
            f.am_dims[:] = [W, H, D]
            f.amask = _p(self.amask)
        return f

    def set_alpha_mask(self, vol):
        self.amask = None if vol is None else _f32(vol)

    def zero_grads(self):
        g = {k: np.zeros_like(v) for k, v in self.p.items() if k in PARAM_NAMES}
        G = OrcGrads()
        for i in range(3):
            G.dps[i] = _p(g[f"density_plane_space.{i}"])
            G.dpt[i] = _p(g[f"density_plane_time.{i}"])
            G.aps[i] = _p(g[f"app_plane_space.{i}"])
            G.apt[i] = _p(g[f"app_plane_time.{i}"])
        G.basis = _p(g["basis_mat.weight"])
        for i, k in enumerate((0, 2, 4) if "renderModule.mlp.0.weight" in g else ()):
            G.rW[i] = _p(g[f"renderModule.mlp.{k}.weight"])
            G.rb[i] = _p(g[f"renderModule.mlp.{k}.bias"])
        for i, k in enumerate(VEL_KEYS if "vel_net.weight_net.1.weight" in g else []):
            G.vW[i] = _p(g[f"vel_net.weight_net.{k}.weight"])
            G.vb[i] = _p(g[f"vel_net.weight_net.{k}.bias"])
            G.aW[i] = _p(g[f"vel_net.a_weight_net.{k}.weight"])
            G.ab[i] = _p(g[f"vel_net.a_weight_net.{k}.bias"])
        return g, G


TRAIN, WHITE_BG, TRANSFER = 1, 2, 4


def set_threads(n):
    lib().orc_set_num_threads(int(n))


def set_vel_fp16(on):
    """velocity-net forward with weights / layer inputs rounded to binary16, fp32 accumulation (checker of the product's vel_fp16 mode)"""
    lib().orc_set_vel_fp16(int(bool(on)))


def f16_round(x):
    L = lib()
    L.orc_f16_round.restype = C.c_float
    L.orc_f16_round.argtypes = [C.c_float]
    return np.array([L.orc_f16_round(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32).reshape(np.shape(x))


def sample_ray(fs, o, d, u=None, **kw):
    f = fs.c_field(**kw)
    o, d = _f32(o), _f32(d)
    R, S = o.shape[0], f.n_samples
    pts = np.empty((R, S, 3), np.float32)
    z = np.empty((R, S), np.float32)
    valid = np.empty((R, S), np.uint8)
    uu = _f32(u).reshape(-1) if u is not None else None
    lib().orc_sample_ray(C.byref(f), C.c_int64(R), _p(o), _p(d), _p(uu), _p(pts), _p(z),
                         valid.ctypes.data_as(C.POINTER(C.c_uint8)))
    return pts, z, valid.astype(bool)


def _pointwise(name, fs, x, out_dim, **kw):
    f = fs.c_field(**kw)
    x = _f32(x)
    N = x.shape[0]
    out = np.empty((N, out_dim), np.float32)
    getattr(lib(), name)(C.byref(f), C.c_int64(N), _p(x), _p(out))
    return out


def vel_net(fs, xt):
    return _pointwise("orc_vel_net", fs, xt, 6)


def get_vel(fs, xt):
    return _pointwise("orc_get_vel", fs, xt, 3)


def vel_gated(fs, xt):
    return _pointwise("orc_vel_gated", fs, xt, 3)


def density_feature(fs, xyzt):
    return _pointwise("orc_density_feature", fs, xyzt, 1)


def app_feature(fs, xyzt):
    return _pointwise("orc_app_feature", fs, xyzt, fs.p["basis_mat.weight"].shape[0])


def sample_alpha(fs, xyz):
    return _pointwise("orc_sample_alpha", fs, xyz, 1)[:, 0]


def feature2density(fs, feat):
    f = fs.c_field()
    feat = _f32(feat).reshape(-1)
    out = np.empty_like(feat)
    lib().orc_feature2density(C.byref(f), C.c_int64(feat.size), _p(feat), _p(out))
    return out


def integrate_pos(fs, x, t, base):
    f = fs.c_field()
    x, t, base = _f32(x), _f32(t).reshape(-1), _f32(base).reshape(-1)
    out = np.empty_like(x)
    lib().orc_integrate_pos(C.byref(f), C.c_int64(x.shape[0]), _p(x), _p(t), _p(base), _p(out))
    return out


def raw2alpha(sigma, dist):
    sigma, dist = _f32(sigma), _f32(dist)
    R, S = sigma.shape
    a, w = np.empty_like(sigma), np.empty_like(sigma)
    lib().orc_raw2alpha(C.c_int64(R), C.c_int64(S), _p(sigma), _p(dist), _p(a), _p(w))
    return a, w


def render_mlp(fs, pts, view, feat):
    f = fs.c_field()
    pts, view, feat = _f32(pts), _f32(view), _f32(feat)
    out = np.empty((pts.shape[0], 3), np.float32)
    lib().orc_render_mlp(C.byref(f), C.c_int64(pts.shape[0]), _p(pts), _p(view), _p(feat), _p(out))
    return out


def sh_render(view, feat):
    view, feat = _f32(view), _f32(feat)
    out = np.empty((view.shape[0], 3), np.float32)
    lib().orc_sh_render(C.c_int64(view.shape[0]), _p(view), _p(feat), _p(out))
    return out


class RenderResult:
    def __init__(self, rgb, depth, acc, weight, counters, ctx, fs_keepalive):
        self.rgb, self.depth, self.acc, self.weight = rgb, depth, acc, weight
        self.counters = counters
        self._ctx = ctx
        self._keep = fs_keepalive

    def backward(self, fs, g_rgb=None, g_depth=None, g_acc=None, g_weight=None):
        assert self._ctx
        g, G = fs.zero_grads()
        gs = [None if a is None else _f32(a) for a in (g_rgb, g_depth, g_acc, g_weight)]
        lib().orc_render_bwd(C.c_void_p(self._ctx), _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(gs[3]), C.byref(G))
        return g

    def free(self):
        if self._ctx:
            lib().orc_ctx_free(C.c_void_p(self._ctx))
            self._ctx = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def render(fs, o, d, t, u=None, train=False, white_bg=True, transfer=False, keep_ctx=False, **kw):
    f = fs.c_field(**kw)
    o, d = _f32(o), _f32(d)
    R, S = o.shape[0], f.n_samples
    rgb = np.empty((R, 3), np.float32)
    depth = np.empty(R, np.float32)
    acc = np.empty(R, np.float32)
    weight = np.empty((R, S), np.float32)
    counters = np.zeros(4, np.int64)
    uu = _f32(u).reshape(-1) if u is not None else None
    flags = (TRAIN if train else 0) | (WHITE_BG if white_bg else 0) | (TRANSFER if transfer else 0)
    ctx = lib().orc_render_fwd(C.byref(f), C.c_int64(R), _p(o), _p(d), _p(uu), C.c_float(t), C.c_int(flags),
                               _p(rgb), _p(depth), _p(acc), _p(weight),
                               counters.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int(int(keep_ctx)))
    return RenderResult(rgb, depth, acc, weight, counters, ctx, (f, o, d))


def pde_loss(fs, points, t, n_jac=0, want_grads=True):
    f = fs.c_field()
    points, t = _f32(points), _f32(t).reshape(-1)
    P = points.shape[0]
    kept = np.zeros(P, np.uint8)
    nk = C.c_int64(0)
    ev = C.c_int64(0)
    jac = np.zeros((max(n_jac, 1), 6, 4), np.float32)
    g, G = fs.zero_grads()
    loss = lib().orc_pde_loss(C.byref(f), C.c_int64(P), _p(points), _p(t), kept.ctypes.data_as(C.POINTER(C.c_uint8)),
                              C.byref(nk), C.c_int64(n_jac), _p(jac), C.byref(G) if want_grads else None, C.byref(ev))
    return dict(loss=float(loss), kept=kept.astype(bool), n_kept=nk.value, jac=jac[:n_jac], grads=g, rk2_evals=ev.value)


def regs(fs):
    f = fs.c_field()
    L = lib()
    return float(L.orc_density_L1(C.byref(f))), float(L.orc_tv_density(C.byref(f))), float(L.orc_tv_app(C.byref(f)))


def maskfield(params, xyz, g_mask=None):
    """MaskField forward (+ parameter gradients of sum(mask * g_mask)).  params: [W0, b0, ..., W4, b4] (torch Linear layout)."""
    ps = [_f32(p) for p in params]
    xyz = _f32(xyz)
    N, K = xyz.shape[0], ps[8].shape[0]
    out = np.empty((N, K), np.float32)
    arr = C.c_void_p * 5
    W = arr(*[ps[2 * i].ctypes.data for i in range(5)])
    b = arr(*[ps[2 * i + 1].ctypes.data for i in range(5)])
    if g_mask is None:
        lib().orc_maskfield(W, b, C.c_int(K), C.c_int64(N), _p(xyz), _p(out), None, None, None)
        return out
    g = _f32(g_mask)
    grads = [np.zeros_like(p) for p in ps]
    gW = arr(*[grads[2 * i].ctypes.data for i in range(5)])
    gb = arr(*[grads[2 * i + 1].ctypes.data for i in range(5)])
    lib().orc_maskfield(W, b, C.c_int(K), C.c_int64(N), _p(xyz), _p(out), _p(g), gW, gb)
    return out, grads
