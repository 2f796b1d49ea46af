"""End-to-end sanity of the training path beyond gradient parity: a student field trained by the reference's loop body
(train_nvfi.py:139-249: two train renders per iteration, L1 / TV regularisers, the PDE term, Adam with the reference's groups) on the
frames of a teacher field must actually approach the teacher - the loss of a FIXED validation batch (eval renders at a keyframe time and
at a non-keyframe time, i.e. through the RK2 warp) has to fall by a large factor.  Small scene (48^3, 64 samples per ray), 500 iterations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(seed, G=48, S=64):
    import bench
    from nvfi_amd.models import NVFi
    torch.manual_seed(seed)
    cfg = bench.bat_cfg(S, True)
    aabb = torch.tensor([[-2.0, -2.0, -2.0], [2.0, 2.0, 2.0]])
    units = 4.0 / (G - 1)
    step = min(7.0 / S, 0.999 * float(np.sqrt(48.0)) / (S - 1))
    cfg.nvfi.step_ratio = step / units
    m = NVFi(cfg, "cpu", aabb, [G, G, G], [1.0, 8.0])
    f = m.nvfi
    with torch.no_grad():
        for i in range(3):
            p = f.density_plane_space[i]
            H, W = p.shape[-2:]
            yy = torch.linspace(-1, 1, H)[:, None]; xx = torch.linspace(-1, 1, W)[None, :]
            p.mul_((3.2 * torch.sqrt(torch.exp(-xx ** 2 / (2 * 0.35 ** 2)) * torch.exp(-yy ** 2 / (2 * 0.35 ** 2))))[None, None])
    return m.cuda()


def test_student_approaches_teacher():
    import bench
    import models                                   # the drop-in alias of the reference's package
    from nvfi_amd.utils import TVLoss
    dev = torch.device("cuda", 0)
    teacher, student = _scene(1), _scene(1)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():                            # a teacher with contrast: stronger appearance factors and output layer
        for p in list(teacher.nvfi.app_plane_space) + list(teacher.nvfi.app_plane_time):
            p.mul_(2.5)
        last = [m for m in teacher.nvfi.renderModule.modules() if isinstance(m, torch.nn.Linear)][-1]
        last.weight.mul_(4.0); last.bias.copy_(torch.tensor([0.8, -0.6, 0.2], device=dev))
    with torch.no_grad():                            # student = teacher with its geometry perturbed and its appearance forgotten
        for p in list(student.nvfi.density_plane_space) + list(student.nvfi.density_plane_time):
            p.mul_((1.0 + 0.5 * torch.randn(p.shape, generator=g)).to(dev))
        for p in list(student.nvfi.app_plane_space) + list(student.nvfi.app_plane_time):
            p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
        for p in student.nvfi.renderModule.parameters():
            p.copy_((0.05 * torch.randn(p.shape, generator=g)).to(dev))
    ren_t, ren_s = models.Renderer(teacher, 0, 0, 1024), models.Renderer(student, 0, 0, 1024)
    poses = [bench.pose_spherical(th, ph, 4.0).to(dev) for th, ph in ((30, -30), (120, -20), (210, -40), (300, -25))]
    focal = 0.5 * 128 / np.tan(0.5 * bench.ANGLE_X)
    cams = [models.Camera(p, 128, 128, focal, torch.zeros(128, 128, 3, device=dev), 1.0, 8.0) for p in poses]
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    rng = np.random.default_rng(5)

    def batch(n):
        cam = cams[int(rng.integers(0, len(cams)))]
        rays, _ = cam.sample_rays_device(n, generator=gen)
        return rays

    def target(t, rays):
        with torch.no_grad():
            return ren_t.render(t, rays, white_background=True, mode="test")[0]

    val = [(20 / 60.0, batch(1024)), (27 / 60.0, batch(1024)), (4 / 60.0, batch(1024))]     # non-key, key (27/60 = 0.45), non-key
    val = [(t, r, target(t, r)) for t, r in val]

    def val_mse():
        with torch.no_grad():
            return float(sum(torch.mean((ren_s.render(t, r, white_background=True, mode="test")[0] - tg) ** 2) for t, r, tg in val) / len(val))

    groups = student.get_optparam_groups(0.02, 1e-3, 1e-3)
    opt = torch.optim.Adam(groups, betas=(0.9, 0.99))
    tv = TVLoss()
    f = student.nvfi
    before = val_mse()
    L1w, tvd, tva, vw, fac = 8e-4, 1.0, 1.0, 1.0, 0.1 ** (1 / 30000)
    hist = []
    for it in range(500):
        student.train()
        i = int(rng.integers(0, 46))
        while i % 3 == 0:
            i = int(rng.integers(0, 46))
        rays = batch(1024)
        loss = torch.nn.functional.mse_loss(ren_s.render(i / 60.0, rays, white_background=True, mode="train")[0], target(i / 60.0, rays))
        t_key = 3 * int(rng.integers(0, 16)) / 60.0
        rays = batch(1024)
        loss = loss + torch.nn.functional.mse_loss(ren_s.render(t_key, rays, white_background=True, mode="train")[0], target(t_key, rays))
        hist.append(float(loss.detach()))
        L1w *= fac; tvd *= fac; tva *= fac; vw *= fac
        loss = loss + L1w * f.density_L1() + tvd * f.TV_loss_density(tv) + tva * f.TV_loss_app(tv)
        lv = student.get_vel_loss(8192)
        if lv > 0:
            loss = loss + vw * lv
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    after = val_mse()
    print(f"validation MSE {before:.5f} -> {after:.5f} ({before / after:.1f}x); train loss first 10 {np.mean(hist[:10]):.5f}, last 10 {np.mean(hist[-10:]):.5f}")
    for p in student.parameters():
        assert torch.isfinite(p).all()
    assert after < before / 10.0, (before, after)


def test_velocity_field_is_learned_from_images():
    """The gradient path rgb -> composite -> density / appearance at the WARPED point -> RK2 adjoint -> velocity net, end to end: a
    student that shares the teacher's radiance field but has forgotten its motion (velocity net re-drawn, ~zero velocity) is trained on
    the teacher's frames at non-keyframe times; its velocity inside the object must approach the teacher's (|v| ~ 0.8)."""
    import bench
    import models
    dev = torch.device("cuda", 0)
    teacher, student = _scene(1), _scene(1)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for p in list(teacher.nvfi.app_plane_space) + list(teacher.nvfi.app_plane_time):
            p.mul_(2.5)
        last = [m for m in teacher.nvfi.renderModule.modules() if isinstance(m, torch.nn.Linear)][-1]
        last.weight.mul_(4.0); last.bias.copy_(torch.tensor([0.8, -0.6, 0.2], device=dev))
        lt = [m for m in teacher.nvfi.vel_net.weight_net.modules() if isinstance(m, torch.nn.Linear)][-1]
        lt.weight.mul_(8.0); lt.bias.copy_(torch.tensor([0.6, -0.4, 0.3, 0.0, 0.0, 0.5], device=dev))     # a teacher that moves
        student.load_state_dict(teacher.state_dict())
        for m in student.nvfi.vel_net.weight_net.modules():
            if isinstance(m, torch.nn.Linear):
                m.weight.copy_((torch.randn(m.weight.shape, generator=g) / np.sqrt(m.weight.shape[1]) * 0.5).to(dev)); m.bias.zero_()
    ren_t, ren_s = models.Renderer(teacher, 0, 0, 1024), models.Renderer(student, 0, 0, 1024)
    focal = 0.5 * 128 / np.tan(0.5 * bench.ANGLE_X)
    cams = [models.Camera(bench.pose_spherical(th, ph, 4.0).to(dev), 128, 128, focal, torch.zeros(128, 128, 3, device=dev), 1.0, 8.0)
            for th, ph in ((30, -30), (120, -20), (210, -40), (300, -25))]
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    rng = np.random.default_rng(5)
    pts = (torch.rand(20000, 3, device=dev, generator=gen) - 0.5) * 0.6          # normalised coordinates inside the object
    tt = torch.rand(20000, 1, device=dev, generator=gen) * 0.75

    def verr():
        with torch.no_grad():
            xt = torch.cat([pts, tt], -1)
            vt, vs = teacher.nvfi.vel_net.get_vel(xt)[..., :3], student.nvfi.vel_net.get_vel(xt)[..., :3]
            return float((vs - vt).norm(dim=-1).mean()), float(vt.norm(dim=-1).mean())

    e0, vmag = verr()
    opt = torch.optim.Adam(student.get_optparam_groups(0.02, 1e-3, 1e-3), betas=(0.9, 0.99))
    for it in range(300):
        student.train()
        loss = 0
        for key in (False, True):
            i = int(rng.integers(0, 46))
            while (i % 3 == 0) != key:
                i = int(rng.integers(0, 46))
            rays = cams[int(rng.integers(0, 4))].sample_rays_device(1024, generator=gen)[0]
            with torch.no_grad():
                tgt = ren_t.render(i / 60.0, rays, white_background=True, mode="test")[0]
            loss = loss + torch.nn.functional.mse_loss(ren_s.render(i / 60.0, rays, white_background=True, mode="train")[0], tgt)
        lv = student.get_vel_loss(8192)
        if lv > 0:
            loss = loss + lv
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    e1, _ = verr()
    print(f"mean |v_student - v_teacher| inside the object: {e0:.3f} -> {e1:.3f} (teacher mean |v| {vmag:.3f})")
    assert e0 > 0.7 * vmag and e1 < 0.35 * vmag, (e0, e1, vmag)
