"""AlphaGridMask container (reference models/tensorf_model_utils.py:417-442); the lookup itself runs
inside the sampling kernel (nvfi_amd/csrc/render.hip: alpha_lookup)."""
import torch


class AlphaGridMask(torch.nn.Module):
    def __init__(self, device, aabb, alpha_volume):
        super().__init__()
        self.device = device
        self.register_buffer("alpha_aabb", aabb.to(device))
        self.register_buffer("alpha_volume", alpha_volume.view(1, 1, *alpha_volume.shape[-3:]).float().contiguous().to(device))
        self.aabbSize = self.alpha_aabb[1] - self.alpha_aabb[0]
        self.invgridSize = 1.0 / self.aabbSize * 2
        self.gridSize = torch.LongTensor([alpha_volume.shape[-1], alpha_volume.shape[-2], alpha_volume.shape[-3]]).to(device)

    def normalize_coord(self, xyz_sampled):
        return (xyz_sampled - self.alpha_aabb[0]) * self.invgridSize - 1
