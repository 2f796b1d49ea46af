"""Round 5: fewer launches per iteration, same results.

* NVFI_FUSED_LAUNCH=0 (the count + k_fill launches, the separate prologue / bookkeeping kernels of rounds 1-4) passes the same goldens, and
  the fused launches produce BIT-IDENTICAL renders and lists-derived counters in-process comparisons cannot show (one switch per process) -
  so the comparison runs the two modes as sub-processes on the same seeded inputs and compares the saved outputs exactly.
* the fragment cache (ABI v5: nvfi_pack_frags / nvfi_field_desc.frags): renders and gradients with and without it are identical; it follows
  the weights through torch's optimiser, nvfi_amd.optim.Adam and invalidate_frags().
* field.render_mse_backward_ (nvfi_render_fwd_mse + nvfi_render_bwd_t, no autograd) against the autograd path: loss, colours, every gradient.
* nvfi_draw_batch: reproducible in (seed, iteration), host and device iteration counters agree, values in range and uniform.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLD, ROOT, relerr
from helpers import make_model, named_grads

pytestmark = pytest.mark.gpu


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, os.path.join(sys.argv[2], "tests"))
from helpers import make_model, named_grads
from conftest import GOLD
gold = np.load(os.path.join(GOLD, "hotpath.npz"))
out = {}
for kind in ("A", "B"):
    model, meta = make_model(kind)
    f = model.nvfi
    o, d = torch.from_numpy(gold[f"{kind}:rays_o"]).cuda(), torch.from_numpy(gold[f"{kind}:rays_d"]).cuda()
    ts = f.tmax / (f.num_keyframes - 1)
    for name, t in (("nonkey", 0.38 * ts), ("key", ts * 2.0), ("far", f.tmax + 1.2 * ts)):
        f.eval()
        with torch.no_grad():
            r = f(t, o, d, True)
        for i, k in enumerate(("rgb", "depth", "acc", "weights")):
            out[f"{kind}:{name}:eval:{k}"] = r[i].cpu().numpy()
        out[f"{kind}:{name}:eval:counters"] = f.last_counters.cpu().numpy()
        f.train()
        f.jitter_override = torch.from_numpy(gold[f"{kind}:train_nonkey:u"].copy())
        model.zero_grad(set_to_none=True)
        r = f(t, o, d, True)
        f.jitter_override = None
        for i, k in enumerate(("rgb", "depth", "acc", "weights")):
            out[f"{kind}:{name}:train:{k}"] = r[i].detach().cpu().numpy()
        out[f"{kind}:{name}:train:counters"] = f.last_counters.cpu().numpy()
    pts, tt = torch.from_numpy(gold[f"{kind}:pde:points"]).cuda(), torch.from_numpy(gold[f"{kind}:pde:t"]).cuda()
    f.pde_debug = 4
    model.zero_grad(set_to_none=True)
    loss = f.pde_loss(pts, tt)
    out[f"{kind}:pde:kept"] = f.last_pde_kept.cpu().numpy()
    out[f"{kind}:pde:counters"] = f.last_pde_counters.cpu().numpy()
    out[f"{kind}:pde:out"] = f.last_pde_out.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_fused_launches_are_bit_identical_to_the_round4_launch_chain(tmp_path):
    """Renders (eval + train, keyframe / non-keyframe / extrapolated times, two fields) and the PDE call's kept set / counters / value with
    the fused launches against NVFI_FUSED_LAUNCH=0: the compact lists are the same entry for entry, so everything derived from them is
    bit-identical (the PDE value sums atomics: compared to rounding)."""
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    res = {}
    for mode in ("1", "0"):
        out = tmp_path / f"m{mode}.npz"
        env = dict(os.environ, NVFI_FUSED_LAUNCH=mode, NVFI_FRAG_CACHE=mode)
        r = subprocess.run([sys.executable, str(script), str(out), ROOT], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        res[mode] = np.load(out)
    a, b = res["1"], res["0"]
    assert set(a.files) == set(b.files) and len(a.files) >= 50
    for k in a.files:
        if k.endswith("pde:out"):
            np.testing.assert_allclose(a[k], b[k], rtol=2e-6, err_msg=k)
        else:
            assert np.array_equal(a[k], b[k]), k


def test_round4_launch_chain_still_matches_goldens():
    env = dict(os.environ, NVFI_FUSED_LAUNCH="0", NVFI_FRAG_CACHE="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_training_loop.py"),
                        "-q", "-x", "-m", "gpu"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def _train_grads(model, t, o, d, u, tgt):
    f = model.nvfi
    model.zero_grad(set_to_none=True)
    f.train()
    f.jitter_override = u
    try:
        out = f(t, o, d, True)
    finally:
        f.jitter_override = None
    loss = torch.nn.functional.mse_loss(out[0], tgt)
    loss.backward()
    return float(loss), out[0].detach().clone(), named_grads(model)


@pytest.mark.parametrize("kind", ["A", "B"])
def test_render_mse_backward_matches_autograd(kind):
    gold = np.load(os.path.join(GOLD, "hotpath.npz"))
    model, meta = make_model(kind)
    f = model.nvfi
    o, d = _cu(gold[f"{kind}:rays_o"]), _cu(gold[f"{kind}:rays_d"])
    u = torch.from_numpy(gold[f"{kind}:train_nonkey:u"].copy())
    tgt = _cu(gold[f"{kind}:train_nonkey:target"])
    ts = f.tmax / (f.num_keyframes - 1)
    for t in (0.38 * ts, ts * 2.0):
        ref_loss, ref_rgb, ref_g = _train_grads(model, t, o, d, u, tgt)
        model.zero_grad(set_to_none=True)
        for p in model.parameters():
            p.grad = torch.zeros_like(p)
        f.train()
        loss, rgb = f.render_mse_backward_(t, o, d, tgt, white_bg=True, jitter=u.cuda().reshape(-1))
        assert torch.equal(rgb, ref_rgb)                                 # the same forward kernels
        assert abs(float(loss) - ref_loss) <= 2e-6 * abs(ref_loss)        # the same squares, another summation order
        g = named_grads(model)
        n = 0
        for k, r in ref_g.items():
            if r is None:
                continue
            assert relerr(g[k], r) < 2e-5, (k, t)
            n += 1
        assert n >= 19
        # loss_scale scales the gradients, not the returned value
        for p in model.parameters():
            p.grad.zero_()
        loss2, _ = f.render_mse_backward_(t, o, d, tgt, white_bg=True, loss_scale=0.25, jitter=u.cuda().reshape(-1))
        assert abs(float(loss2) - ref_loss) <= 2e-6 * abs(ref_loss)
        g2 = named_grads(model)
        k = "density_plane_space.0"
        assert relerr(g2[k], 0.25 * ref_g[k]) < 2e-5


def test_fragment_cache_follows_the_weights():
    """with the cache (default) and without it (NVFI_FRAG_CACHE=0 path, forced per call here) the render is bit-identical; after an optimiser step -
    torch.optim.Adam (version counters), nvfi_amd.optim.Adam (GENERATION) - and after a raw .data edit + invalidate_frags() the cached
    fragments are the new weights'"""
    from nvfi_amd.models.tensorf_keyframe import _rt
    from nvfi_amd.optim import Adam
    gold = np.load(os.path.join(GOLD, "hotpath.npz"))
    model, meta = make_model("A")
    f = model.nvfi
    o, d = _cu(gold["A:rays_o"]), _cu(gold["A:rays_d"])
    t = 0.38 * f.tmax / (f.num_keyframes - 1)

    def render(cache):
        _rt(f)["_frag_on"] = cache
        f.eval()
        with torch.no_grad():
            r = f(t, o, d, True)
        return r[0].clone(), r[1].clone()

    def both():
        a, b = render(True), render(False)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        return a[0]

    r0 = both()
    assert _rt(f).get("_frag_cache") is not None
    mlp = [p for n, p in model.named_parameters() if "renderModule" in n or "vel_net.weight_net" in n]
    for make in (lambda ps: torch.optim.Adam(ps, lr=5e-2), lambda ps: Adam(ps, lr=5e-2)):
        opt = make(mlp)
        for p in mlp:
            p.grad = torch.randn_like(p)
        _rt(f)["_frag_on"] = True
        opt.step()
        r1 = both()
        assert not torch.equal(r1, r0)
        r0 = r1
    with torch.no_grad():
        for p in mlp:
            p.data.mul_(1.05)
    f.invalidate_frags()
    r2 = both()
    assert not torch.equal(r2, r0)
    _rt(f)["_frag_on"] = None


def _draw(desc, stream=None):
    from nvfi_amd import _lib
    _lib.check(_lib.lib().nvfi_draw_batch(C.byref(desc), C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def test_draw_batch_is_a_counter_based_generator():
    from nvfi_amd import _lib
    dev = torch.device("cuda")
    n_pix, R, P = 640000, 2048, 262144
    bo, bd = torch.randn(n_pix, 3, device=dev), torch.randn(n_pix, 3, device=dev)
    aabb = [-1.5, -1.0, -0.5, 1.0, 2.0, 0.75]

    def run(seed, it, it_dev=None, R=R, P=P):
        ro = [torch.empty(R, 3, device=dev) for _ in range(2)]; rd = [torch.empty(R, 3, device=dev) for _ in range(2)]
        tg = [torch.empty(R, 3, device=dev) for _ in range(2)]; ids = [torch.empty(R, dtype=torch.int64, device=dev) for _ in range(2)]
        pts, tt = torch.empty(P, 3, device=dev), torch.empty(P, device=dev)
        ds = _lib.DrawDesc()
        ds.seed, ds.iteration, ds.iteration_dev = seed, it, _lib.ptr(it_dev)
        ds.n_batches, ds.R, ds.n_pixels = 2, R, n_pix
        ds.bundle_o, ds.bundle_d, ds.target_img = _lib.ptr(bo), _lib.ptr(bd), None
        for b in range(2):
            ds.rays_o[b], ds.rays_d[b], ds.target[b], ds.pixel_ids[b] = _lib.ptr(ro[b]), _lib.ptr(rd[b]), _lib.ptr(tg[b]), _lib.ptr(ids[b])
        ds.P = P
        ds.aabb[:] = aabb
        ds.points, ds.t = _lib.ptr(pts), _lib.ptr(tt)
        _draw(ds)
        torch.cuda.synchronize()
        return ro, rd, tg, ids, pts, tt

    a = run(7, 3)
    b = run(7, 3)
    for x, y in zip(a[:4], b[:4]):
        assert all(torch.equal(p, q) for p, q in zip(x, y))
    assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5])
    c = run(7, 4)
    assert not torch.equal(a[4], c[4]) and not torch.equal(a[3][0], c[3][0])
    e = run(8, 3)
    assert not torch.equal(a[4], e[4])
    # the iteration counter in device memory (hipGraph replay) gives the host counter's batch
    itd = torch.tensor([3], dtype=torch.int64, device=dev)
    g = run(7, 999, it_dev=itd)
    assert torch.equal(a[4], g[4]) and torch.equal(a[3][1], g[3][1])
    # a prefix of a smaller draw is the same stream (counter = element index): shards / smaller batches see the same leading values
    h = run(7, 3, R=256, P=4096)
    assert torch.equal(h[4], a[4][:4096]) and torch.equal(h[3][0], a[3][0][:256])
    ro, rd, tg, ids, pts, tt = a
    for bch in range(2):
        assert int(ids[bch].min()) >= 0 and int(ids[bch].max()) < n_pix
        assert torch.equal(ro[bch], bo[ids[bch]]) and torch.equal(rd[bch], bd[ids[bch]])
        assert float(tg[bch].min()) >= 0.0 and float(tg[bch].max()) < 1.0
    assert not torch.equal(ids[0], ids[1])
    lo, hi = torch.tensor(aabb[:3], device=dev), torch.tensor(aabb[3:], device=dev)
    assert bool((pts >= lo).all()) and bool((pts <= hi).all())
    assert float(tt.min()) >= 0.0 and float(tt.max()) < 1.0
    # uniformity: means and variances of 262144 draws (sigma of the mean = 5.6e-4), a 64-bin histogram of t, no correlation between columns
    un = ((pts - lo) / (hi - lo)).double()
    assert float((un.mean(0) - 0.5).abs().max()) < 4e-3 and float((un.var(0) - 1.0 / 12.0).abs().max()) < 2e-3
    hist = torch.histc(tt, bins=64, min=0.0, max=1.0)
    assert float((hist - P / 64).abs().max()) < 6.0 * np.sqrt(P / 64)
    cc = torch.corrcoef(torch.cat([un, tt.double()[:, None]], 1).T)
    assert float((cc - torch.eye(4, device=dev, dtype=cc.dtype)).abs().max()) < 1e-2
    pid = ids[0].double() / n_pix
    assert abs(float(pid.mean()) - 0.5) < 0.04
    # round 6: the R pixels of a batch are DISTINCT, like np.random.choice(replace=False) (models/camera.py:160) ...
    for bch in range(2):
        assert int(torch.unique(ids[bch]).numel()) == R
    # ... also when the batch is the whole image (a permutation), and every pixel is equally likely over many iterations
    small = 1000
    bo2, bd2 = bo[:small].contiguous(), bd[:small].contiguous()
    ds = _lib.DrawDesc()
    ro2, rd2, tg2 = torch.empty(small, 3, device=dev), torch.empty(small, 3, device=dev), torch.empty(small, 3, device=dev)
    id2 = torch.empty(small, dtype=torch.int64, device=dev)
    ds.seed, ds.iteration, ds.n_batches, ds.R, ds.n_pixels = 11, 0, 1, small, small
    ds.bundle_o, ds.bundle_d, ds.target_img = _lib.ptr(bo2), _lib.ptr(bd2), None
    ds.rays_o[0], ds.rays_d[0], ds.target[0], ds.pixel_ids[0] = _lib.ptr(ro2), _lib.ptr(rd2), _lib.ptr(tg2), _lib.ptr(id2)
    _draw(ds); torch.cuda.synchronize()
    assert torch.equal(torch.sort(id2)[0], torch.arange(small, device=dev))
    ds.R = 100
    hits = torch.zeros(small, device=dev)
    first = torch.zeros(small, device=dev)
    for it in range(400):
        ds.iteration = it
        _draw(ds)
        hits.index_add_(0, id2[:100], torch.ones(100, device=dev))
        first[id2[0]] += 1
    torch.cuda.synchronize()
    # 400 x 100 draws over 1000 pixels: 40 expected per pixel (sigma 6), 0.4 as the first pixel of a batch
    assert float(hits.sum()) == 40000 and float((hits - 40).abs().max()) < 6 * 6.0, (float(hits.min()), float(hits.max()))
    assert float(first.max()) <= 6
    ds.R, ds.n_pixels = 200, 100
    assert _lib.lib().nvfi_draw_batch(C.byref(ds), C.c_void_p(torch.cuda.current_stream().cuda_stream)) != 0
    assert b"without replacement" in _lib.lib().nvfi_last_error()


def test_captured_call_never_bakes_in_a_cache_it_did_not_pack():
    """the fragment-cache pointer of a CAPTURED call is baked into the graph.  A cache that is current at capture time (an eval
    render ran after the last optimiser step) must not be used unless its pack launch is a node of the same capture - otherwise every replay
    would render stale fragments while the optimiser keeps moving the weights.  (a) without repack_frags() the captured call packs per call
    (frags NULL) and the replay follows the weights; (b) with repack_frags() at the head of the capture the cache is used AND the replay
    follows the weights; (c) a second capture without a pack node does not inherit the first one's permission."""
    from nvfi_amd.models.tensorf_keyframe import _rt
    from nvfi_amd.optim import Adam
    gold = np.load(os.path.join(GOLD, "hotpath.npz"))
    model, meta = make_model("A")
    f = model.nvfi
    f.eval()
    o, d = _cu(gold["A:rays_o"]), _cu(gold["A:rays_d"])
    t = 0.38 * f.tmax / (f.num_keyframes - 1)
    mlp = [p for n, p in model.named_parameters() if "renderModule" in n or "vel_net.weight_net" in n]
    opt = Adam(mlp, lr=5e-2)

    def eager():
        with torch.no_grad():
            return f(t, o, d, True)[0].clone()

    def step():
        for p in mlp:
            p.grad = torch.randn_like(p)
        opt.step()

    for repack in (False, True, False):
        eager()                                    # the cache is current for today's weights
        assert _rt(f)["_frag_cache"] is not None
        seen = []
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            if repack:
                f.repack_frags()
            seen.append(f._desc().frags)
            with torch.no_grad():
                out = f(t, o, d, True)[0]
        torch.cuda.current_stream().wait_stream(cap)
        assert (seen[0] is not None) == repack, (repack, seen)
        before = eager()
        step()
        after = eager()
        assert not torch.equal(before, after)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, after), ("replay rendered stale fragments", repack, float((out - after).abs().max()))
