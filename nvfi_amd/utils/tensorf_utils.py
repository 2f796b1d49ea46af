"""Plane regularisers used every training step (next-row f-1): restated from utils/tensorf_utils.py:53-57,139-158."""
import torch
import torch.nn as nn


def N_to_reso(n_voxels, bbox):
    xyz_min, xyz_max = bbox
    dim = len(xyz_min)
    voxel_size = ((xyz_max - xyz_min).prod() / n_voxels).pow(1 / dim)
    return ((xyz_max - xyz_min) / voxel_size).long().tolist()


class TVLoss(nn.Module):
    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x, t=False):
        b, c, h, w = x.shape
        count_h = c * (h - 1) * w
        count_w = c * h * (w - 1)
        h_tv = torch.pow(x[:, :, 1:, :] - x[:, :, :h - 1, :], 2).sum()
        if t:
            h_tv = h_tv * 3
        w_tv = torch.pow(x[:, :, :, 1:] - x[:, :, :, :w - 1], 2).sum()
        return self.TVLoss_weight * 2 * (h_tv / count_h + w_tv / count_w) / b
