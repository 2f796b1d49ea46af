"""N>1 path on CPU (gloo, world_size 2 and 8): ray shards + ONE flat-gradient all-reduce (issued as head + tail) reproduce the
single-process gradient of the global batch - with a ray count that does NOT divide by the world size (unequal ray shards: shard_weight)
and the naturally unequal kept counts of the PDE shards (W * n_r / sum n_r staging).  The per-rank compute stands in for the GPU kernels
with the CPU oracle (tests may use it)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import pytest
import torch.multiprocessing as mp

from conftest import GOLD, ROOT

DROP = 3        # rays left out of the golden batch: 256 - 3 = 253 divides neither by 2 nor by 8


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from nvfi_amd.dist import GradBucket, PdeGradStage, pde_rank_weight, shard_range, shard_weight
    orc.set_threads(2 if world <= 2 else 1)
    gold = np.load(os.path.join(GOLD, "hotpath.npz"))
    fs = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_A.npz"))
    o, d = gold["A:rays_o"], gold["A:rays_d"]
    u, tgt = gold["A:train_nonkey:u"], gold["A:train_nonkey:target"]
    R = o.shape[0] - DROP
    lo, hi = shard_range(R, rank, world)
    # parameters as torch tensors, two of them channels_last like the product's planes
    names = ["density_plane_space.0", "app_plane_time.1", "renderModule.mlp.0.bias", "vel_net.weight_net.3.0.weight"]
    params = []
    for n in names:
        t = torch.from_numpy(fs.p[n].copy())
        if t.dim() == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        params.append(torch.nn.Parameter(t))
    bucket = GradBucket(params)
    assert all(p.grad.stride() == p.stride() for p in params)
    bucket.zero()
    # local shard: mean over the shard's rays, times W * n_r / R (unequal shards: the first ranks hold one ray more)
    r = orc.render(fs, o[lo:hi], d[lo:hi], 19.0 / 60.0, u=u[lo:hi], train=True, white_bg=True, keep_ctx=True)
    n = hi - lo
    sw = shard_weight(n, R)
    assert abs(sw - world * n / R) < 1e-12 and (sw != 1.0 or R % world == 0)
    g = r.backward(fs, g_rgb=sw * 2 * (r.rgb - tgt[lo:hi]) / (n * 3))
    r.free()
    # PDE shard with the kept-count re-weighting
    P = gold["A:pde:points"].shape[0]
    plo, phi = shard_range(P, rank, world)
    pde = orc.pde_loss(fs, gold["A:pde:points"][plo:phi], gold["A:pde:t"][plo:phi])
    w = pde_rank_weight(pde["n_kept"])
    # the PDE gradients enter through the staging buffer (device-side W*n_r/sum(n_r) weighting), as bench.py does
    # ... and the exchange is split as in bench.py: the head (planes, render MLP - final after the renders) starts
    # asynchronously, the PDE term lands in the tail (velocity nets) meanwhile, then the tail is reduced and everything averaged
    stage = PdeGradStage([params[3]])
    off = bucket.tail_offset([params[3]])
    assert off == sum(p.numel() for p in params[:3]) and bucket.tail_offset([params[0]]) is None
    with torch.no_grad():
        stage.views[0].copy_(torch.from_numpy(pde["grads"][names[3]]))
        for p, nm in zip(params, names):
            p.grad += torch.from_numpy(g[nm])
    h = bucket.all_reduce_head_start(off)
    stage.commit(pde["n_kept"])
    bucket.all_reduce_finish(h, off)
    kept_all = [None] * world
    dist.all_gather_object(kept_all, int(pde["n_kept"]))
    if rank == 0:
        np.savez(out, **{nm: p.grad.detach().contiguous().numpy() for nm, p in zip(names, params)}, w=w, nk=np.array(kept_all), n_rays=np.array(n))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_gradients_match_single_process(tmp_path, world):
    from oracle import oracle as orc
    out = str(tmp_path / "g.npz")
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    gold = np.load(os.path.join(GOLD, "hotpath.npz"))
    fs = orc.FieldSpec.from_npz(os.path.join(GOLD, "field_A.npz"))
    o, d = gold["A:rays_o"], gold["A:rays_d"]
    u, tgt = gold["A:train_nonkey:u"], gold["A:train_nonkey:target"]
    R = o.shape[0] - DROP
    assert R % world != 0
    o, d, u, tgt = o[:R], d[:R], u[:R], tgt[:R]
    r = orc.render(fs, o, d, 19.0 / 60.0, u=u, train=True, white_bg=True, keep_ctx=True)
    g = r.backward(fs, g_rgb=2 * (r.rgb - tgt) / (R * 3))
    r.free()
    pde = orc.pde_loss(fs, gold["A:pde:points"], gold["A:pde:t"])
    for nm in ["density_plane_space.0", "app_plane_time.1", "vel_net.weight_net.3.0.weight", "renderModule.mlp.0.bias"]:
        ref = g[nm] + pde["grads"][nm]
        err = np.abs(got[nm] - ref).max() / (np.abs(ref).max() + 1e-30)
        assert err < 2e-5, (nm, err)
    nk = got["nk"]
    assert len(nk) == world and len(set(nk.tolist())) > 1 and int(nk.sum()) == int(pde["n_kept"])      # unequal kept counts, nothing lost
    assert abs(float(got["w"]) - world * nk[0] / nk.sum()) < 1e-9


def test_shard_range_covers_everything():
    from nvfi_amd.dist import shard_range
    for n in (0, 1, 7, 2048, 262145):
        for w in (1, 2, 3, 8):
            pieces = [shard_range(n, r, w) for r in range(w)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(w - 1))
