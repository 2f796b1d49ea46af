#!/usr/bin/env python
"""Run the reference's OWN training driver on the MI355X-native hot path.

    python tools/run_reference_driver.py [--check] [--stub-missing] [--fused-adam] [--pure-autograd] /path/to/NVFi/train_nvfi.py --config config/InDoorObj/bat.yaml --static_dynamic

Why a launcher: `python train_nvfi.py` puts the SCRIPT's directory at sys.path[0], ahead of PYTHONPATH, so `from models import *`
(train_nvfi.py:16) would still resolve to the reference's own `models/` package.  This launcher runs the untouched script with
`runpy.run_path` after ordering sys.path as

    [ this repository (its top-level `models/` alias -> nvfi_amd.models),  the script's directory (`utils/`, `datasets/`, `config/`),  ... ]

so `models` is the HIP-backed mirror and everything else the driver imports (`utils.TVLoss`, `utils.CfgNode`, `datasets.load_blender_data`,
...) stays the reference's own code.  No reference file is copied or modified.

--check          resolve the three packages, print where each came from as one JSON line, and exit without running the driver
--fused-adam     opt-in: `torch.optim.Adam` built by the driver without an explicit `fused` / `foreach` choice (train_nvfi.py:95, 353-357) gets
                 `fused=True` - the same update rule in one pass per parameter group instead of torch's default ~60 multi-tensor launches, which
                 re-read the 38 MB of plane parameters, gradients and moments a dozen times (0.73 of the 8.0 ms of an iteration on an MI355X).
                 The driver's source is not touched; results differ from the default implementation by rounding only.  Off by default.
--pure-autograd  leave the library default (parameter gradients handed back to the autograd engine).  WITHOUT this flag the launcher sets
                 NVFI_INPLACE_GRADS=arena before `models` is imported: the kernels accumulate into a gradient arena that backs p.grad (no fresh
                 38 MB gradient tensors per backward node; INTEGRATION.md section 3 states what that changes for a caller: .grad tensors are
                 re-used across iterations, and a field with autograd hooks on a parameter falls back to pure autograd by itself).
--stub-missing   register empty placeholder modules for optional third-party imports of the reference that are absent on this host
                 (wandb, lpips, imageio, cv2, torchvision: logging / metrics / dataset decoding, never the render or training math).  Off by default.
"""
import importlib
import json
import os
import runpy
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPTIONAL = ["wandb", "lpips", "imageio", "cv2", "torchvision", "torchvision.transforms"]


def order_sys_path(script):
    sdir = os.path.dirname(os.path.abspath(script))
    rest = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (ROOT, sdir)]
    sys.path[:] = [ROOT, sdir] + rest
    return sdir


def stub_missing():
    made = []
    for name in OPTIONAL:
        try:
            importlib.import_module(name)
        except Exception:
            m = types.ModuleType(name)
            m.__nvfi_stub__ = True
            sys.modules[name] = m
            made.append(name)
    if "cv2" in made:
        sys.modules["cv2"].COLORMAP_JET = 2       # read at import time by utils/tensorf_utils.py
    if "torchvision.transforms" in made and "torchvision" in made:
        sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    return made


def resolve():
    out = {}
    for pkg in ("models", "utils", "datasets"):
        try:
            m = importlib.import_module(pkg)
            out[pkg] = os.path.dirname(os.path.abspath(m.__file__))
        except Exception as e:      # reported, not fatal for --check: the driver itself would stop here
            out[pkg] = f"import failed: {type(e).__name__}: {e}"
    import models
    import nvfi_amd.models
    if models.NVFi is not nvfi_amd.models.NVFi or models.Renderer is not nvfi_amd.models.Renderer:
        raise SystemExit(f"run_reference_driver: `models` resolved to {out['models']}, not to this repository's alias ({ROOT}/models)")
    return out


def fused_adam_default():
    """torch.optim.Adam(...) without a `fused` / `foreach` argument -> fused=True (CUDA parameters only; the class itself is subclassed, not edited)"""
    import torch

    base = torch.optim.Adam
    if getattr(base, "__nvfi_fused_default__", False):
        return

    class Adam(base):
        __nvfi_fused_default__ = True

        def __init__(self, params, *a, **kw):
            params = list(params)
            flat = [p for g in params for p in (g["params"] if isinstance(g, dict) else [g])]
            if "fused" not in kw and "foreach" not in kw and flat and all(p.is_cuda and torch.is_floating_point(p) for p in flat):
                kw["fused"] = True
            super().__init__(params, *a, **kw)

    Adam.__name__ = Adam.__qualname__ = "Adam"
    torch.optim.Adam = Adam


FLAGS = ("--check", "--stub-missing", "--fused-adam", "--pure-autograd")


def main(argv):
    flags = [a for a in argv if a in FLAGS]
    rest = [a for a in argv if a not in FLAGS]
    if not rest or not os.path.isfile(rest[0]):
        raise SystemExit(__doc__)
    script, args = rest[0], rest[1:]
    if "--pure-autograd" not in flags:
        os.environ.setdefault("NVFI_INPLACE_GRADS", "arena")       # read when the field is constructed
    sdir = order_sys_path(script)
    stubs = stub_missing() if "--stub-missing" in flags else []
    where = resolve()
    if "--check" in flags:
        tv = None
        try:        # the regulariser object the driver builds (train_nvfi.py:132) must stay on the fused kernel
            import utils
            from nvfi_amd.utils.tensorf_utils import is_reference_tvloss
            tv = bool(is_reference_tvloss(utils.TVLoss()))
        except Exception as e:
            tv = f"{type(e).__name__}: {e}"
        print(json.dumps(dict(script=os.path.abspath(script), script_dir=sdir, resolved=where, stubs=stubs, reference_TVLoss_on_fused_kernel=tv)))
        return 0
    if "--fused-adam" in flags:
        fused_adam_default()
    sys.argv = [script] + args
    os.chdir(sdir)                  # the driver opens config/... and data paths relative to its checkout
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
