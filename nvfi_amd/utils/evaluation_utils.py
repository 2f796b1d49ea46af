"""Checkpoint compatibility (next-row f-4): same dictionary layout and file naming as the reference
(train_nvfi.py:359-392, utils/evaluation_utils.py:20-43), so checkpoints interchange between the two implementations."""
import glob
import os

import torch


def save_checkpoint(path, nvfi, optimizer, epoch=None):
    """{logdir}/model_{epoch:05d}.ckpt with keys model_state_dict / optimizer_state_dict / nvfi_kwarg (train_nvfi.py:359-369).
    Planes are saved with their logical (1,C,H,W) shape; the channels_last physical layout is an implementation detail."""
    sd = {k: (v.contiguous() if v.dim() == 4 else v) for k, v in nvfi.state_dict().items()}
    ckpt = {"model_state_dict": sd, "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else {},
            "nvfi_kwarg": nvfi.nvfi.get_kwargs()}
    if epoch is not None:
        path = os.path.join(path, "model_%05d.ckpt" % epoch)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save(ckpt, path)
    return path


def load_checkpoint(logdir, map_location="cpu"):
    """newest *.ckpt under logdir (utils/evaluation_utils.py:20-43)"""
    files = sorted(glob.glob(os.path.join(logdir, "*.ckpt")))
    if not files:
        raise FileNotFoundError(f"no checkpoint under {logdir}")
    return torch.load(files[-1], map_location=map_location, weights_only=False)


def load_model_checkpoint(cfg, ckpt, device):
    """Rebuild NVFi at the saved aabb / gridSize / num_keyframes and load the weights (train_nvfi.py:372-392)."""
    from ..models import NVFi, Renderer
    kw = ckpt["nvfi_kwarg"]
    cfg.nvfi.num_keyframes = kw["num_keyframes"]
    nvfi = NVFi(cfg, device, kw["aabb"].to(device), kw["gridSize"], kw["near_far"])
    nvfi.load_state_dict(ckpt["model_state_dict"], strict=False)
    nvfi = nvfi.to(device)
    renderer = Renderer(nvfi, cfg.renderer.batch_size, cfg.renderer.test_batch_size, cfg.renderer.n_rays,
                        cfg.renderer.distance_scale, tensorf_sample=cfg.renderer.tensorf_sample, ndc=cfg.renderer.ndc) if "renderer" in cfg else None
    return nvfi, renderer
