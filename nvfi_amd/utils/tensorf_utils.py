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


_TV_OK = {}


def is_reference_tvloss(reg):
    """True when `reg` computes the reference's TVLoss (utils/tensorf_utils.py:139-158) - whichever module its class came from.

    train_nvfi.py builds `tvreg = TVLoss()` from the REFERENCE's own `utils` package (`from utils import *`, train_nvfi.py:17,132), so an
    exact-type test against this module's class would push every TV term of the real driver off the fused kernel.  The test is
    structural - class name, a numeric `TVLoss_weight`, `forward(x, t=False)` - plus a behavioural fingerprint: the module is evaluated
    once per class on a tiny CPU tensor (space and time variants) and must reproduce the formula.  The verdict is cached per class."""
    cls = type(reg)
    ok = _TV_OK.get(cls)
    if ok is None:
        ok = False
        try:
            import inspect
            params = list(inspect.signature(cls.forward).parameters.values())
            if cls.__name__ == "TVLoss" and len(params) == 3 and params[2].default is False:
                x = (torch.arange(24, dtype=torch.float32).reshape(1, 2, 3, 4) % 5) * 0.25 - 0.5
                probe = cls.__new__(cls)
                nn.Module.__init__(probe)
                probe.TVLoss_weight = 1.0
                mine = TVLoss(1.0)
                with torch.no_grad():
                    ok = all(abs(float(probe(x, t=tt)) - float(mine(x, t=tt))) <= 1e-6 * abs(float(mine(x, t=tt))) for tt in (False, True))
        except Exception:
            ok = False
        _TV_OK[cls] = ok
    w = getattr(reg, "TVLoss_weight", None)
    return bool(ok) and isinstance(w, (int, float)) and not isinstance(w, bool)



class _MseFn(__import__("torch").autograd.Function):
    """F.mse_loss(x, target) with value and gradient from one launch (nvfi_mse); backward is one scaling launch."""

    @staticmethod
    def forward(ctx, x, target):
        import ctypes as C
        import torch
        from .. import _lib
        xc, tc = x.contiguous(), target.contiguous()
        loss = torch.empty((), device=x.device)
        grad = torch.empty_like(xc)
        _lib.check(_lib.lib().nvfi_mse(_lib.ptr(xc), _lib.ptr(tc), C.c_int64(xc.numel()), _lib.ptr(loss), _lib.ptr(grad),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def mse_loss(x, target):
    """torch.nn.functional.mse_loss(x, target) (mean reduction) for CUDA fp32 tensors of a training batch: one launch forward, one backward.
    The kernel is ONE workgroup (a 2048-ray batch is 6144 values: launch-bound, not bandwidth-bound), so it is used for batch-sized inputs only;
    anything else - other dtypes / devices, more than 65536 values (a full image would run on one CU) - goes to torch."""
    import torch
    if x.is_cuda and x.dtype == torch.float32 and target.dtype == torch.float32 and x.shape == target.shape and 0 < x.numel() <= (1 << 16):
        return _MseFn.apply(x, target)
    return torch.nn.functional.mse_loss(x, target)
