#!/usr/bin/env python
"""MaskField forward+backward throughput (points/s) on N points: exact fp32 MFMA vs the optional fp16-input MFMA mode."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvfi_amd.models import MaskField

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
torch.manual_seed(0)
mf = MaskField(n_layer=4, n_dim=128, input_dim=3, skips=[], mask_dim=8).cuda()
pts = torch.rand(N, 3, device="cuda") * 2 - 1
g = torch.randn(N, 8, device="cuda")
for fp16 in (False, True):
    mf.mfma_fp16 = fp16
    def step():
        mf.zero_grad(set_to_none=True)
        (mf(pts) * g).sum().backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    flop = N * 3 * 2 * (3 * 128 + 3 * 128 * 128 + 128 * 8)
    print(f"fp16_mfma={fp16}: N={N} fwd+bwd {ms:.3f} ms  {N / ms * 1e3:.3e} points/s  {flop / ms / 1e9:.1f} TFLOP/s (fwd+dgrad+wgrad GEMM FLOPs)")
