"""Ray container and pinhole camera (boundary types; reference models/camera.py:34-172)."""
import numpy as np
import torch


class Ray(torch.nn.Module):
    def __init__(self, ray_o, ray_d, near, far, t=None):
        super().__init__()
        self.restore_shape = ray_o.shape[:-1]
        self.register_buffer("ray_origins", ray_o)
        self.register_buffer("ray_directions", ray_d)
        self.num_rays = ray_o.reshape(-1, 3).shape[0]
        # near / far / t are per-ray buffers in the reference (models/camera.py:38-44).  The render path never reads them (the field samples
        # between its own near / far), so they are materialised on first access instead of costing five launches per Ray.
        self.__dict__["_lazy"] = {"near": near, "far": far, "t": t}

    def __getattr__(self, name):
        lazy = self.__dict__.get("_lazy")
        if lazy is not None and name in lazy:
            v = lazy.pop(name)
            d = self._buffers["ray_directions"]
            if name == "t":
                buf = torch.zeros_like(d[..., :1]) if v is None else v
            else:
                buf = v if (isinstance(v, torch.Tensor) and v.dim() > 0) else v * torch.ones_like(d[..., :1])
            self.register_buffer(name, buf)
            return buf
        return super().__getattr__(name)

    def _apply(self, fn, *a, **k):
        """`.to()` / `.cuda()` / `.float()` also move the not-yet-materialised near / far / t tensors (the reference registers them in
        __init__, models/camera.py:38-44, so they follow the module; a pending tensor left behind would meet buffers on another device)"""
        lazy = self.__dict__.get("_lazy")
        if lazy:
            for name in list(lazy):
                if isinstance(lazy[name], torch.Tensor):
                    lazy[name] = fn(lazy[name])
        return super()._apply(fn, *a, **k)

    def state_dict(self, *a, **k):
        for name in list(self.__dict__.get("_lazy") or ()):      # the reference's keys: ray_origins, ray_directions, near, far, t
            getattr(self, name)
        return super().state_dict(*a, **k)

    def update_near_far(self, near, far):
        for k, v in (("near", near), ("far", far)):
            self.__dict__["_lazy"].pop(k, None)
            self.register_buffer(k, v) if isinstance(v, torch.Tensor) else setattr(self, k, v)

    def points_sampling(self, n_points, lindisp=False, perturb=True):
        """Stratified depths between near and far (models/camera.py:55-76); the hot path samples inside the field instead
        (tensorf_sample=True in every shipped config), this exists for callers that ask the ray container directly."""
        u = torch.linspace(0.0, 1.0, n_points, dtype=torch.float32, device=self.ray_directions.device)
        z = 1.0 / (1.0 / self.near * (1.0 - u) + 1.0 / self.far * u) if lindisp else self.near * (1.0 - u) + self.far * u
        if perturb:
            mid = 0.5 * (z[..., 1:] + z[..., :-1])
            hi = torch.cat((mid, z[..., -1:]), dim=-1)
            lo = torch.cat((z[..., :1], mid), dim=-1)
            z = lo + (hi - lo) * torch.rand(*z.shape, dtype=self.ray_origins.dtype, device=self.ray_origins.device)
        self.z_vals = z
        return self.ray_origins[..., None, :] + self.ray_directions[..., None, :] * z[..., :, None]


class BatchedRays(object):
    """All rays / pixels / frame times of a set of posed images, flattened (models/camera.py:7-31; built once by train_nvfi.py:121)."""

    def __init__(self, all_targets, all_poses, all_times, H, W, focal, near, far, ndc=False):
        self.target_images, self.poses, self.times = all_targets, all_poses, all_times
        self.H, self.W, self.focal, self.near, self.far, self.ndc = H, W, focal, near, far, ndc
        rays, pixels, ts = [], [], []
        for pose, target, time in zip(all_poses, all_targets, all_times):
            t = torch.ones_like(target[..., :1]) * time
            r = Camera(pose, H, W, focal, target, near, far, ndc, t).rays
            rays.append(torch.cat([r.ray_origins.reshape(-1, 3), r.ray_directions.reshape(-1, 3)], dim=-1))
            pixels.append(target.reshape(-1, 3))
            ts.append(t.reshape(-1, 1))
        self.all_rays, self.all_pixels, self.all_ts = torch.cat(rays, 0), torch.cat(pixels, 0), torch.cat(ts, 0)

    def __len__(self):
        return len(self.all_rays)


class Camera(object):
    """Pinhole ray bundle (models/camera.py:80-138); rays are built on the pose's device."""

    def __init__(self, pose, height, width, focal, target, near, far, ndc=False, t=None, dpt=None):
        if ndc:
            raise NotImplementedError("ndc rays are out of scope (ndc: False in every shipped config)")
        self.pose, self.height, self.width, self.focal = pose, height, width, focal
        self.target, self.near, self.far, self.t, self.dpt = target, near, far, t, dpt
        self._coords = self._rays = None

    # The reference builds the pixel grid and the whole H*W ray bundle in the constructor (camera.py:96-101) and train_nvfi.py constructs a
    # Camera every iteration; both are materialised on first use here (same values), so a driver that only draws a batch through
    # sample_rays_device pays for 2048 rays instead of 640 000.
    @property
    def coords(self):
        if self._coords is None:
            dev = self.pose.device
            ii, jj = torch.meshgrid(torch.arange(self.height, device=dev), torch.arange(self.width, device=dev), indexing="ij")
            self._coords = torch.stack([ii, jj], dim=-1).reshape(-1, 2)
        return self._coords

    @coords.setter
    def coords(self, v):          # plain attributes in the reference (camera.py:96-101): a caller may assign them
        self._coords = v

    @property
    def rays(self):
        if self._rays is None:
            ray_origins, ray_directions = self.get_ray_bundle()
            self._rays = Ray(ray_origins, ray_directions, self.near, self.far, self.t)
        return self._rays

    @rays.setter
    def rays(self, v):
        self._rays = v

    def get_ray_bundle(self):
        X, Y = torch.meshgrid(torch.arange(self.width, dtype=self.pose.dtype, device=self.pose.device),
                              torch.arange(self.height, dtype=self.pose.dtype, device=self.pose.device), indexing="xy")
        directions = torch.stack([(X - self.width * 0.5) / self.focal, -(Y - self.height * 0.5) / self.focal,
                                  -torch.ones_like(X)], dim=-1)
        ray_directions = torch.sum(directions[..., None, :] * self.pose[:3, :3], dim=-1)
        ray_origins = self.pose[:3, -1].expand(ray_directions.shape)
        return ray_origins, ray_directions

    def sample_rays_device(self, n_rays, generator=None):
        """Device-side variant of sample_rays (next-row f-2): pixel ids drawn on the GPU, rays generated by nvfi_gen_rays without
        materialising the H*W bundle.  n_rays DISTINCT pixels, like np.random.choice(replace=False) below (a device-side permutation;
        the fused driver's nvfi_draw_batch draws without replacement too, through a keyed bijection of the pixel range)."""
        import ctypes as C
        from .. import _lib
        dev = self.pose.device
        ids = torch.randperm(self.height * self.width, device=dev, generator=generator)[:n_rays].contiguous()
        ro = torch.empty(n_rays, 3, device=dev)
        rd = torch.empty(n_rays, 3, device=dev)
        pose = self.pose[:3, :4].contiguous().float()
        _lib.check(_lib.lib().nvfi_gen_rays(_lib.ptr(pose), C.c_int(self.height), C.c_int(self.width), C.c_float(float(self.focal)),
                                            C.c_int64(n_rays), _lib.ptr(ids), _lib.ptr(ro), _lib.ptr(rd),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return Ray(ro, rd, self.near, self.far), ids

    def sample_rays(self, n_rays):
        select_inds = np.random.choice(self.coords.shape[0], size=n_rays, replace=False)
        sc = self.coords[select_inds]
        sample_ray = Ray(self.rays.ray_origins[sc[:, 0], sc[:, 1], :], self.rays.ray_directions[sc[:, 0], sc[:, 1], :], self.near, self.far)
        target_pixels = self.target[sc[:, 0], sc[:, 1], :]
        if self.dpt is not None:
            return sample_ray, target_pixels, self.dpt[sc[:, 0], sc[:, 1]]
        return sample_ray, target_pixels
