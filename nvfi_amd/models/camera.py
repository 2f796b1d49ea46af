"""Ray container and pinhole camera (boundary types; reference models/camera.py:34-172)."""
import numpy as np
import torch


class Ray(torch.nn.Module):
    def __init__(self, ray_o, ray_d, near, far, t=None):
        super().__init__()
        self.restore_shape = ray_o.shape[:-1]
        self.register_buffer("ray_origins", ray_o)
        self.register_buffer("ray_directions", ray_d)
        self.register_buffer("near", near * torch.ones_like(self.ray_directions[..., :1]))
        self.register_buffer("far", far * torch.ones_like(self.ray_directions[..., :1]))
        self.num_rays = ray_o.reshape(-1, 3).shape[0]
        self.register_buffer("t", torch.zeros_like(ray_o[..., :1]) if t is None else t)


class Camera(object):
    """Pinhole ray bundle (models/camera.py:80-138); rays are built on the pose's device."""

    def __init__(self, pose, height, width, focal, target, near, far, ndc=False, t=None, dpt=None):
        if ndc:
            raise NotImplementedError("ndc rays are out of scope (ndc: False in every shipped config)")
        self.pose, self.height, self.width, self.focal = pose, height, width, focal
        self.target, self.near, self.far, self.t, self.dpt = target, near, far, t, dpt
        ii, jj = torch.meshgrid(torch.arange(height, device=pose.device), torch.arange(width, device=pose.device), indexing="ij")
        self.coords = torch.stack([ii, jj], dim=-1).reshape(-1, 2)
        ray_origins, ray_directions = self.get_ray_bundle()
        self.rays = Ray(ray_origins, ray_directions, near, far, t)

    def get_ray_bundle(self):
        X, Y = torch.meshgrid(torch.arange(self.width, dtype=self.pose.dtype, device=self.pose.device),
                              torch.arange(self.height, dtype=self.pose.dtype, device=self.pose.device), indexing="xy")
        directions = torch.stack([(X - self.width * 0.5) / self.focal, -(Y - self.height * 0.5) / self.focal,
                                  -torch.ones_like(X)], dim=-1)
        ray_directions = torch.sum(directions[..., None, :] * self.pose[:3, :3], dim=-1)
        ray_origins = self.pose[:3, -1].expand(ray_directions.shape)
        return ray_origins, ray_directions

    def sample_rays(self, n_rays):
        select_inds = np.random.choice(self.coords.shape[0], size=n_rays, replace=False)
        sc = self.coords[select_inds]
        sample_ray = Ray(self.rays.ray_origins[sc[:, 0], sc[:, 1], :], self.rays.ray_directions[sc[:, 0], sc[:, 1], :], self.near, self.far)
        target_pixels = self.target[sc[:, 0], sc[:, 1], :]
        if self.dpt is not None:
            return sample_ray, target_pixels, self.dpt[sc[:, 0], sc[:, 1]]
        return sample_ray, target_pixels
