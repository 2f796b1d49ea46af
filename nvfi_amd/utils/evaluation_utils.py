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
    from ..models import NVFi, Renderer, AlphaGridMask
    kw = ckpt["nvfi_kwarg"]
    cfg.nvfi.num_keyframes = kw["num_keyframes"]
    # near / far come from the dataset section like the reference (:375); the checkpoint's kwargs then override attributes (:378)
    near_far = [cfg.dataset.near, cfg.dataset.far] if "dataset" in cfg else kw["near_far"]
    nvfi = NVFi(cfg, device, kw["aabb"].to(device), kw["gridSize"], near_far).to(device)
    nvfi.update_nvfi_kwargs(kw)
    sd = ckpt["model_state_dict"]
    if "nvfi.alphaMask.alpha_aabb" in sd and "nvfi.alphaMask.alpha_volume" in sd:      # the occupancy mask is rebuilt first (:380-385)
        nvfi.nvfi.alphaMask = AlphaGridMask(device, sd["nvfi.alphaMask.alpha_aabb"].to(device), sd["nvfi.alphaMask.alpha_volume"].to(device))
    nvfi.load_state_dict(sd)            # strict: a missing or unexpected key is an error, as in the reference (:386)
    renderer = Renderer(nvfi, cfg.renderer.batch_size, cfg.renderer.test_batch_size, cfg.renderer.n_rays,
                        cfg.renderer.distance_scale, tensorf_sample=cfg.renderer.tensorf_sample, ndc=cfg.renderer.ndc) if "renderer" in cfg else None
    return nvfi, renderer


def render_test_evaluation(nvfi, renderer, poses, times, targets, H, W, focal, near, far, white_background=True,
                           savedir=None, update_alpha_mask=True, device=None):
    """Eval driver (train_nvfi.py:395-459 without the dataset / wandb plumbing): optional `updateAlphaMask` at the current grid
    (`:413`), one `Renderer.render(mode='test')` per (pose, time) frame (`:437`), 8-bit PNGs named r_%03d.png (`:449-451`,
    written with PIL - imageio is not a dependency here) and per-frame / mean PSNR against `targets` (H,W,3 in [0,1]).

    Returns {"psnr": [..], "mean_psnr": float, "images": uint8 array (N,H,W,3)}."""
    import numpy as np
    from ..models import Camera
    from .metrics import mse2psnr
    device = device or next(nvfi.parameters()).device
    nvfi.eval()
    if update_alpha_mask:
        nvfi.nvfi.updateAlphaMask(nvfi.nvfi.gridSize)
    imgs, psnrs = [], []
    with torch.no_grad():
        for idx in range(len(poses)):
            pose = torch.as_tensor(poses[idx], dtype=torch.float32, device=device)
            cam = Camera(pose, H, W, focal, None, near, far)
            rgb = renderer.render(float(times[idx]), cam.rays.to(device), white_background=white_background, mode="test")[0]
            rgb = rgb.reshape(H, W, 3)
            if targets is not None:
                tgt = torch.as_tensor(targets[idx], dtype=torch.float32, device=device).reshape(H, W, 3)
                psnrs.append(mse2psnr(float(torch.mean((rgb - tgt) ** 2))))
            imgs.append((rgb.clamp(0, 1).cpu().numpy() * 255.0).astype(np.uint8))
    if savedir is not None:
        os.makedirs(savedir, exist_ok=True)
        try:
            from PIL import Image
            for idx, img in enumerate(imgs):
                Image.fromarray(img).save(os.path.join(savedir, "r_%03d.png" % idx))
        except ImportError:      # no image writer in this environment: keep the raw frames
            np.save(os.path.join(savedir, "frames.npy"), np.stack(imgs))
    return {"psnr": psnrs, "mean_psnr": (sum(psnrs) / len(psnrs)) if psnrs else None, "images": np.stack(imgs)}
