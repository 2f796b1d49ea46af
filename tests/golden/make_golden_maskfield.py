#!/usr/bin/env python
"""Golden vectors for the MaskField training step (BASELINE config 5), generated from the REFERENCE implementation
(/root/reference/models/mask_field.py, PyTorch CPU) in the build container:  python tests/golden/make_golden_maskfield.py
Writes tests/golden/maskfield.npz: state_dict, points, softmax mask, an upstream gradient g and every parameter gradient of
sum(mask * g) (autograd), for K = 8 (train_segm.py:97-102 with n_object = 8) and a ragged K = 3 case."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402  (stubs cv2 & co, puts /root/reference on sys.path)

MaskField = import_reference()["MaskField"]


def case(tag, K, N, seed, fx):
    torch.manual_seed(seed)
    m = MaskField(n_layer=4, n_dim=128, input_dim=3, skips=[], mask_dim=K, mask_act="softmax")
    with torch.no_grad():   # spread the logits so that the softmax is not uniform
        m.mask_fc.weight.mul_(6.0)
    pts = (torch.rand(N, 3) * 2 - 1) * 0.9
    g = torch.randn(N, K)
    mask = m(pts)
    (mask * g).sum().backward()
    for k, v in m.state_dict().items():
        fx[f"{tag}:sd:{k}"] = v.detach().numpy().astype(np.float32)
    fx[f"{tag}:pts"] = pts.numpy().astype(np.float32)
    fx[f"{tag}:g"] = g.numpy().astype(np.float32)
    fx[f"{tag}:mask"] = mask.detach().numpy().astype(np.float32)
    for k, p in m.named_parameters():
        fx[f"{tag}:grad:{k}"] = p.grad.numpy().astype(np.float32)


if __name__ == "__main__":
    fx = {}
    case("K8", 8, 333, 233, fx)
    case("K3", 3, 65, 7, fx)
    np.savez_compressed(os.path.join(HERE, "maskfield.npz"), **fx)
    print("wrote", len(fx), "arrays")
