"""Velocity-field parameter containers (reference models/velocity_field.py:14-98).

The modules keep the reference's structure so that state_dict keys match
(`weight_net.1.weight`, `weight_net.3.0.weight`, ...); the arithmetic runs in libnvfi_hip.so.
"""
import torch
import torch.nn as nn

from ..utils.tensorf_utils import N_to_reso  # noqa: F401  (re-exported like the reference)


class PositionEncoder(nn.Module):
    """Buffer holder only (base_network.py:20-54); the encoding is fused into the MLP kernels."""

    def __init__(self, encode_dim):
        super().__init__()
        self.encode_dim = encode_dim
        self.register_buffer("frequency_bands", 2.0 ** torch.linspace(0.0, encode_dim - 1, encode_dim, dtype=torch.float32))


class VelBasis(nn.Module):
    def __init__(self):
        super().__init__()
        in_dim, hidden = 4 + 4 * 2 * 3, 128
        self.weight_net = nn.Sequential(PositionEncoder(3), nn.Linear(in_dim, hidden), nn.SiLU())
        for _ in range(4):
            self.weight_net.append(nn.Sequential(nn.Linear(hidden, hidden), nn.SiLU()))
        self.weight_net.append(nn.Sequential(nn.Linear(hidden, 6)))
        self.a_weight_net = nn.Sequential(PositionEncoder(3), nn.Linear(in_dim, hidden), nn.ReLU())
        for _ in range(4):
            self.a_weight_net.append(nn.Sequential(nn.Linear(hidden, hidden), nn.ReLU()))
        self.a_weight_net.append(nn.Sequential(nn.Linear(hidden, 6)))
        self._owner = None  # set by the field: gives access to the C-ABI descriptor

    @staticmethod
    def linears(net):
        return [net[1]] + [net[i][0] for i in range(3, 8)]

    def forward(self, xt):
        """(v,a) for xt (N,4) - VelBasis.forward (velocity_field.py:69-75). Inference only."""
        return self._owner()._vel_eval(xt, gated=False)

    def get_vel(self, xt):
        return self.forward(xt)[..., :3]


class VelocityAABB(nn.Module):
    def __init__(self, vel_net, eps=-0.03):
        super().__init__()
        self.vel_net = vel_net
        self.eps = eps

    def forward(self, xt):
        return self.vel_net._owner()._vel_eval(xt, gated=True)


class VelocityAABBSur(nn.Module):
    def __init__(self, vel_net, aabb, surround):
        super().__init__()
        self.vel_net = vel_net
        self.aabb = aabb
        self.surround = surround
        self.bounds = (surround - aabb[0]) * 2 / (aabb[1] - aabb[0]) - 1

    def forward(self, xt):
        return self.vel_net._owner()._vel_eval(xt, gated=True)
