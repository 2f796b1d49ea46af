"""MaskField (reference models/mask_field.py:34-83) as train_segm.py:97-102 builds it: 3 -> 128 x 4 (ReLU) -> K, softmax.

Same constructor, attribute names and state_dict keys as the reference.  `forward(point)` runs the HIP kernels of
nvfi_amd/csrc/mask.hip through the C ABI (nvfi_maskfield_fwd / nvfi_maskfield_bwd, include/nvfi_hip.h) behind a
torch.autograd.Function, so the reference's training step (train_segm.py:172-198: mask = model(xyz); loss(mask).backward();
Adam) runs unchanged on it.  Inside a render the same weights are evaluated by k_mask_fwd (mask branch, render.hip).
There is no CPU path: tensors must live on the GPU.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _mask_desc(mf, params):
    md = _lib.MaskDesc()
    md.n_layer, md.n_dim, md.mask_dim = len(mf.point_fc), mf.point_fc[0].out_features, mf.mask_dim
    for i in range(5):
        md.W[i] = _lib.ptr(params[2 * i]); md.b[i] = _lib.ptr(params[2 * i + 1])
    return md


class _MaskFn(torch.autograd.Function):
    """autograd boundary around nvfi_maskfield_fwd / nvfi_maskfield_bwd."""

    @staticmethod
    def forward(ctx, mf, point, *params):
        L = _lib.lib()
        N = point.shape[0]
        train = any(ctx.needs_input_grad[2:])   # False under torch.no_grad(): inference, no stash
        pts = point.detach().contiguous().float()
        params_c = [p.detach().contiguous() for p in params]
        md = _mask_desc(mf, params_c)
        nbytes = C.c_int64(0)
        _lib.check(L.nvfi_maskfield_workspace_bytes(C.byref(md), C.c_int64(N), C.c_int(1 if train else 0), C.byref(nbytes)))
        ws = torch.empty(int(nbytes.value), dtype=torch.uint8, device=pts.device)
        out = torch.empty(N, mf.mask_dim, device=pts.device, dtype=torch.float32)
        mode = (1 if train else 0) | (2 if mf.mfma_fp16 else 0)
        _lib.check(L.nvfi_maskfield_fwd(C.byref(md), C.c_int64(N), _lib.ptr(pts), _lib.ptr(out), C.c_int(mode),
                                        _lib.ptr(ws), C.c_int64(ws.numel()), _stream_ptr()))
        ctx.mf, ctx.N, ctx.ws, ctx.params_c, ctx.train, ctx.mode = mf, N, ws, params_c, train, mode
        return out

    @staticmethod
    def backward(ctx, g_out):
        if not ctx.train:
            raise _lib.NvfiError("MaskField backward without a training forward")
        L = _lib.lib()
        md = _mask_desc(ctx.mf, ctx.params_c)
        grads = [torch.zeros_like(p) if ctx.needs_input_grad[2 + k] else None for k, p in enumerate(ctx.params_c)]
        mg = _lib.MaskGrads()
        for i in range(5):
            mg.W[i] = _lib.ptr(grads[2 * i]); mg.b[i] = _lib.ptr(grads[2 * i + 1])
        g = g_out.contiguous().float()
        if ctx.N > 0:
            _lib.check(L.nvfi_maskfield_bwd(C.byref(md), C.c_int64(ctx.N), _lib.ptr(g), C.byref(mg), C.c_int(ctx.mode), _lib.ptr(ctx.ws),
                                            C.c_int64(ctx.ws.numel()), _stream_ptr()))
        ctx.ws = None
        return (None, None) + tuple(grads)


class MaskField(nn.Module):
    def __init__(self, n_layer=4, n_dim=128, input_dim=3, skips=(), mask_dim=8, mask_act="softmax", point_embed=False, mfma_fp16=False):
        super().__init__()
        # mfma_fp16 (not a reference argument): layer products on the fp16-input MFMA with fp32 accumulation (BASELINE config 5);
        # the default is the exact-fp32 MFMA path that matches the reference's fp32 train_segm.py
        self.mfma_fp16 = bool(mfma_fp16)
        if n_layer != 4 or n_dim != 128 or input_dim != 3 or len(skips) or mask_act != "softmax" or point_embed:
            raise NotImplementedError("only the MaskField of train_segm.py:97-102 (3->128x4->K, softmax) is on the hot path")
        if not 1 <= mask_dim <= 32:
            raise NotImplementedError("mask_dim must be in 1..32")
        self.skips, self.mask_dim = list(skips), mask_dim
        self.point_embed = None
        self.point_fc = nn.ModuleList([nn.Linear(input_dim, n_dim)] + [nn.Linear(n_dim, n_dim) for _ in range(n_layer - 1)])
        self.mask_fc = nn.Linear(n_dim, mask_dim)

    def _params(self):
        ps = []
        for lin in list(self.point_fc) + [self.mask_fc]:
            ps += [lin.weight, lin.bias]
        return ps

    def forward(self, point):
        """point (N,3) -> mask (N,mask_dim), rows sum to 1 (models/mask_field.py:68-83)."""
        if point.dim() != 2 or point.shape[1] != 3:
            raise ValueError("MaskField expects (N,3) points")
        if point.shape[0] == 0:
            return point.new_zeros(0, self.mask_dim)
        return _MaskFn.apply(self, point, *self._params())
