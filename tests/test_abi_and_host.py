"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares, the host mirror has
the reference's surface (names, shapes, optimiser groups), and the product path refuses to run without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from helpers import make_model, load_meta


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "nvfi_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nvfi_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import ctypes as C
    from nvfi_amd import _lib
    from nvfi_amd.build import build
    build()
    L = C.CDLL(_lib.SO)
    names = _header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/nvfi_hip.h but not exported"
    assert set(_lib.EXPORTS) <= set(names)
    assert _lib.lib().nvfi_abi_version() == 5


def test_struct_layout_matches_header():
    """ctypes mirror of nvfi_field_desc / nvfi_grads has the size the C compiler gives the header's structs."""
    import subprocess, tempfile, ctypes as C
    from nvfi_amd import _lib
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "nvfi_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(nvfi_field_desc), sizeof(nvfi_grads), '
           'sizeof(nvfi_draw_desc), offsetof(nvfi_field_desc, frags), offsetof(nvfi_draw_desc, points));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        a, b, c, o1, o2 = map(int, subprocess.check_output([os.path.join(d, "s")]).split())
    assert C.sizeof(_lib.FieldDesc) == a and C.sizeof(_lib.Grads) == b and C.sizeof(_lib.DrawDesc) == c
    assert _lib.FieldDesc.frags.offset == o1 and _lib.DrawDesc.points.offset == o2      # ABI v5 additions


@pytest.mark.parametrize("kind", ["A", "B"])
def test_state_dict_surface(kind):
    model, meta = make_model(kind, device="cpu")
    _, sd = load_meta(kind)
    own = model.state_dict()
    for k, v in sd.items():
        assert k in own and tuple(own[k].shape) == tuple(v.shape), k
    # the reference registers vel_net twice (tensorf_keyframe.py:94,106): duplicated keys must exist
    assert "nvfi.vel.vel_net.weight_net.1.weight" in own and "nvfi.vel_net.weight_net.1.weight" in own
    f = model.nvfi
    for p in list(f.density_plane_space) + list(f.app_plane_time):
        assert p.is_contiguous(memory_format=torch.channels_last)     # physical [H][W][C]
    groups = model.get_optparam_groups(0.02, 1e-3)
    assert [g["lr"] for g in groups] == [0.02] * 4 + [1e-3] * 4
    assert type(f.vel).__name__ == ("VelocityAABBSur" if kind == "B" else "VelocityAABB")
    kw = f.get_kwargs()
    assert kw["gridSize"] == [int(g) for g in meta["gridSize"]] and kw["num_keyframes"] == int(meta["num_keyframes"])


def test_no_cpu_fallback():
    from nvfi_amd._lib import NvfiError
    model, _ = make_model("A", device="cpu")
    with pytest.raises(NvfiError):
        model.render_ray(0.3, torch.zeros(4, 3), torch.ones(4, 3))
    with pytest.raises(NvfiError):
        model.get_vel_loss(64)


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under nvfi_amd/ may reference it"""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "nvfi_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h")):
                if re.search(r"\boracle\b", open(os.path.join(dp, fn), errors="ignore").read()):
                    bad.append(fn)
    assert not bad, bad


def test_upsample_keeps_layout_and_matches_reference_formula():
    model, _ = make_model("A", device="cpu")
    f = model.nvfi
    before = f.density_plane_space[0].detach().clone()
    f.upsample_volume_grid([26, 24, 22], 4)
    p = f.density_plane_space[0]
    assert tuple(p.shape) == (1, 24, 24, 26) and p.is_contiguous(memory_format=torch.channels_last)
    ref = torch.nn.functional.interpolate(before, size=(24, 26), mode="bilinear", align_corners=True)
    assert torch.allclose(p, ref)
    assert f.gridSize.tolist() == [26, 24, 22] and f._step_host == float(f.stepSize)


def test_checkpoint_round_trip(tmp_path):
    """f-4: the reference's checkpoint dictionary layout; planes come back channels_last with identical values"""
    from nvfi_amd.utils import save_checkpoint, load_checkpoint, load_model_checkpoint
    from helpers import field_cfg
    model, meta = make_model("A", device="cpu")
    opt = torch.optim.Adam(model.get_optparam_groups(), betas=(0.9, 0.99))
    path = save_checkpoint(str(tmp_path), model, opt, epoch=12)
    assert path.endswith("model_00012.ckpt")
    ck = load_checkpoint(str(tmp_path))
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "nvfi_kwarg"}
    assert ck["model_state_dict"]["nvfi.density_plane_space.0"].is_contiguous()    # logical NCHW on disk
    m2, _ = load_model_checkpoint(field_cfg(meta), ck, "cpu")
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert m2.nvfi.density_plane_space[0].is_contiguous(memory_format=torch.channels_last)
    # with an occupancy mask: the loader rebuilds AlphaGridMask from the state_dict keys and loads STRICTLY (train_nvfi.py:380-386)
    from nvfi_amd.models import AlphaGridMask
    gs = [int(g) for g in meta["gridSize"]]
    vol = (torch.rand(gs[2], gs[1], gs[0]) > 0.4).float()
    model.nvfi.alphaMask = AlphaGridMask("cpu", model.nvfi.aabb, vol)
    save_checkpoint(str(tmp_path), model, opt, epoch=13)
    ck = load_checkpoint(str(tmp_path))
    assert "nvfi.alphaMask.alpha_volume" in ck["model_state_dict"] and "alphaMask_grid" in ck["nvfi_kwarg"]
    m3, _ = load_model_checkpoint(field_cfg(meta), ck, "cpu")
    assert m3.nvfi.alphaMask is not None and torch.equal(m3.nvfi.alphaMask.alpha_volume.reshape(vol.shape), vol)
    assert m3.nvfi.nSamples == model.nvfi.nSamples and m3.nvfi._aabb_host == model.nvfi._aabb_host
    bad = dict(ck); bad["model_state_dict"] = dict(ck["model_state_dict"]); bad["model_state_dict"]["nvfi.bogus"] = torch.zeros(1)
    with pytest.raises(RuntimeError):
        load_model_checkpoint(field_cfg(meta), bad, "cpu")


def test_camera_rays_match_reference(gold):
    """f-2: Camera.get_ray_bundle against the reference's rays (models/camera.py:112-138) - the strided centre crop of the 800x800
    camera that tests/golden/make_golden.py stored as `A:rays_o/d`, `B:rays_o/d` (and `nvfi_gen_rays` is tested against this
    Camera on the GPU, so the device-side generator inherits the pin)."""
    import bench
    from nvfi_amd.models import Camera
    full, stride, H = 800, 24, 16
    focal = 0.5 * full / np.tan(0.5 * 0.6911112)
    for kind, center in (("A", (0.0, 0.0, 0.0)), ("B", (0.0, 0.0, 3.0))):
        pose = bench.pose_spherical(30.0, -30.0, 4.0)
        pose[:3, 3] += torch.tensor(center)
        cam = Camera(pose, full, full, focal, torch.zeros(1, 1, 3), 1.0, 8.0)
        i0 = full // 2 - (H // 2) * stride
        sl = slice(i0, i0 + H * stride, stride)
        o = cam.rays.ray_origins[sl, sl].reshape(-1, 3).numpy()
        d = cam.rays.ray_directions[sl, sl].reshape(-1, 3).numpy()
        assert np.array_equal(o, gold[f"{kind}:rays_o"]) and np.array_equal(d, gold[f"{kind}:rays_d"]), kind


def test_boundary_ray_types():
    """`from models import *` of the reference exposes Ray, Camera, BatchedRays (models/__init__.py:1): containers behave alike"""
    import torch
    from nvfi_amd.models import BatchedRays, Camera, Ray
    pose = torch.eye(4)
    img = torch.rand(4, 5, 3)
    b = BatchedRays([img, img * 0.5], [pose, pose], [0.1, 0.2], 4, 5, 3.0, 1.0, 8.0)
    assert len(b) == 40 and b.all_rays.shape == (40, 6) and b.all_pixels.shape == (40, 3) and b.all_ts.shape == (40, 1)
    assert float(b.all_ts[0]) == pytest.approx(0.1) and float(b.all_ts[-1]) == pytest.approx(0.2)
    cam = Camera(pose, 4, 5, 3.0, img, 1.0, 8.0)
    assert torch.equal(b.all_rays[:20, 3:], cam.rays.ray_directions.reshape(-1, 3))
    r = Ray(torch.zeros(7, 3), torch.ones(7, 3), 1.0, 8.0)
    p = r.points_sampling(6, perturb=False)
    assert p.shape == (7, 6, 3) and float(p[0, 0, 0]) == pytest.approx(1.0) and float(p[0, -1, 0]) == pytest.approx(8.0)
    r.update_near_far(torch.full((7, 1), 2.0), torch.full((7, 1), 4.0))     # buffers: tensors, as in the reference
    assert float(r.near[0]) == 2.0 and float(r.far[0]) == 4.0


def test_bench_spawn_helper_refuses_more_ranks_than_gpus(capsys):
    """bench.py --gpus N starts N RCCL ranks itself; with fewer devices than ranks it must say so and fail (not run one process)."""
    import bench
    assert bench._spawn_ranks(8, 1, "nccl") == 2
    assert "needs 8 GPUs" in capsys.readouterr().err


def test_rccl_comm_needs_a_process_group_for_the_bootstrap():
    from nvfi_amd import _lib
    from nvfi_amd.dist import RcclComm
    with pytest.raises(_lib.NvfiError, match="init_process_group"):
        RcclComm(world=2, rank=1)


def test_traffic_classes_match_the_committed_counter_pass(tmp_path):
    """bench.py quotes roofline.traffic from profiles/<tag>_traffic.json, which tools/make_traffic.py derives from the committed FETCH_SIZE /
    WRITE_SIZE summaries by kernel name.  A renamed kernel (new template arguments) must not silently turn a class into 0 bytes: every
    class the current default path launches has to match a kernel of the committed trace, and the dominant class has to carry bytes."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("make_traffic", os.path.join(ROOT, "tools", "make_traffic.py"))
    mt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mt)
    bench_src = open(os.path.join(ROOT, "bench.py")).read()
    tag = re.search(r'^PROFILE_TAG = "(\w+)"', bench_src, re.M).group(1)
    prof = os.path.join(ROOT, "profiles")
    out = tmp_path / "traffic.json"
    mt.main(os.path.join(prof, f"{tag}_pmc_fetch_size.csv"), os.path.join(prof, f"{tag}_pmc_write_size.csv"), str(out), "test")
    fresh = json.load(open(out))["bytes_per_launch"]
    assert all(v is not None and v > 0 for v in fresh.values()), fresh
    committed = json.load(open(os.path.join(prof, f"{tag}_traffic.json")))["bytes_per_launch"]
    assert committed == pytest.approx(fresh)
    assert committed["pde_prefilter"] > 1e6


def test_train_short_circuit_sees_mode_changes_made_on_the_inner_field():
    """NVFi.train() returns early for an unchanged mode (train_nvfi.py:141-142 calls it every iteration), but a mode set on the inner field -
    model.nvfi.eval(), as tests and the golden scripts do - or on a module attached later must not survive the next model.train()"""
    m, _ = make_model("A", "cpu")
    m.train()
    assert m.nvfi.training and m.nvfi.vel_net.training
    m.train()                                   # the short-circuit path
    m.nvfi.eval()                               # behind the wrapper's back
    assert not m.nvfi.training and m.training
    m.train()
    assert m.nvfi.training and all(c.training for c in m.nvfi.modules())
    m.eval(); m.eval()
    m.nvfi.renderModule.train()                 # one direct child flipped
    m.eval()
    assert not any(c.training for c in m.modules())
    m.train()
    from nvfi_amd.models.mask_field import MaskField
    m.nvfi.mask_field = MaskField(4).eval()     # attached later, in another mode
    m.train()
    assert m.nvfi.mask_field.training


def test_gradient_arena_is_opt_in():
    """the default autograd contract of a drop-in nn.Module: gradients go back to the engine (torch.autograd.grad, hooks); the in-place
    arena is something a driver switches on (NVFI_INPLACE_GRADS=arena, bench.py --mode dropin, tools/run_reference_driver.py)"""
    m, _ = make_model("A", "cpu")
    assert m.nvfi.accumulate_grads_inplace is False
    src = open(os.path.join(ROOT, "tools", "run_reference_driver.py")).read()
    assert 'setdefault("NVFI_INPLACE_GRADS", "arena")' in src and "--pure-autograd" in src


def test_ray_lazy_buffers_follow_the_module_and_camera_attributes_are_assignable():
    from nvfi_amd.models import Ray, Camera
    o, d = torch.zeros(5, 3), torch.ones(5, 3)
    r = Ray(o, d, torch.full((5, 1), 2.0), 6.0, t=torch.full((5, 1), 0.25))
    r2 = r.to(torch.float64)                                   # pending tensors are converted with the buffers
    assert r2.near.dtype == torch.float64 and r2.t.dtype == torch.float64 and r2.far.dtype == torch.float64
    assert set(Ray(o, d, 2.0, 6.0).state_dict()) == {"ray_origins", "ray_directions", "near", "far", "t"}
    cam = Camera(torch.eye(4), 4, 6, 5.0, torch.zeros(4, 6, 3), 2.0, 6.0)
    rays = cam.rays
    cam.rays = rays; cam.coords = cam.coords                   # attributes in the reference, assignable here too
    assert cam.rays is rays
