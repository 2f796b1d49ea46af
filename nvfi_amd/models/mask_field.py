"""MaskField parameter container (reference models/mask_field.py:34-83) as train_segm.py:97-102 builds it
(n_layer=4, n_dim=128, skips=[], softmax); evaluated by k_mask_fwd inside the mask branch of the render."""
import torch.nn as nn


class MaskField(nn.Module):
    def __init__(self, n_layer=4, n_dim=128, input_dim=3, skips=(), mask_dim=8, mask_act="softmax", point_embed=False):
        super().__init__()
        if n_layer != 4 or n_dim != 128 or input_dim != 3 or len(skips) or mask_act != "softmax" or point_embed:
            raise NotImplementedError("only the MaskField of train_segm.py:97-102 (3->128x4->K, softmax) is on the hot path")
        self.skips, self.mask_dim = list(skips), mask_dim
        self.point_embed = None
        self.point_fc = nn.ModuleList([nn.Linear(input_dim, n_dim)] + [nn.Linear(n_dim, n_dim) for _ in range(n_layer - 1)])
        self.mask_fc = nn.Linear(n_dim, mask_dim)
