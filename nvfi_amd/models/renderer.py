"""Renderer: ray chunk loop + reshape to rays.restore_shape (reference models/renderer.py:7-65)."""
import torch
import torch.nn as nn


class Renderer(nn.Module):
    def __init__(self, tensorf, batch_size, test_batch_size, ray_chunk, distance_scale=1, lindisp=False,
                 perturb=True, tensorf_sample=True, ndc=False):
        super().__init__()
        self.tensorf = tensorf
        self.batch_size, self.test_batch_size = batch_size, test_batch_size
        self.lindisp, self.perturb, self.distance_scale = lindisp, perturb, distance_scale
        self.tensorf_sample, self.ndc, self.ray_chunk = tensorf_sample, ndc, ray_chunk

    def forward(self, t, rays, white_background=False, transfer_vel=False):
        ray_o = rays.ray_origins.reshape(-1, 3)
        ray_d = rays.ray_directions.reshape(-1, 3)
        n_all = ray_o.shape[0]
        outs = [[], [], [], [], []]
        for c in range(n_all // self.ray_chunk + int(n_all % self.ray_chunk > 0)):
            r_o = ray_o[c * self.ray_chunk:(c + 1) * self.ray_chunk]
            r_d = ray_d[c * self.ray_chunk:(c + 1) * self.ray_chunk]
            fn = self.tensorf.render_ray_transfer if transfer_vel else self.tensorf.render_ray
            res = fn(t, r_o, r_d, white_background, self.ndc)
            for lst, v in zip(outs, res):
                lst.append(v)
        # (one chunk - every training batch - needs no concatenation: five copy launches less per render)
        rgb_map, depth_map, acc_map, weights, extra = [o[0] if len(o) == 1 else torch.cat(o, 0) for o in outs]
        shp = tuple(rays.restore_shape)
        return (rgb_map.reshape(*shp, 3), depth_map.reshape(*shp), acc_map.reshape(*shp),
                weights.reshape(*shp, -1), extra.reshape(*shp, extra.shape[-1]))

    def render(self, t, rays, white_background=False, mode="train", transfer_vel=False):
        if mode == "train":
            if not self.tensorf.training:     # (the reference calls .train()/.eval() unconditionally; the recursion is pure host cost)
                self.tensorf.train()
            return self.forward(t, rays, white_background)
        if self.tensorf.training:
            self.tensorf.eval()
        with torch.no_grad():
            return self.forward(t, rays, white_background, transfer_vel=transfer_vel)
