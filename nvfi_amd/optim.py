"""Adam for the outer optimiser step in one HIP launch (nvfi_adam_step, nvfi_amd/csrc/optim.hip).

Drop-in for `torch.optim.Adam(groups, betas=(0.9, 0.99))` as train_nvfi.py:88-96 builds it (no amsgrad, no weight decay, per-group
learning rates that the training loop rescales every iteration): same update rule, same `state_dict()` layout (`step`, `exp_avg`,
`exp_avg_sq` per parameter), so optimiser checkpoints interchange with the reference's.  `step(zero_grad=True)` also clears the gradients in
the same pass (the reference's separate `optimizer.zero_grad()` becomes a no-op on already-zero buffers)."""
import ctypes as C

import torch

from . import _lib


# bumped by every step(): nvfi_adam_step writes the parameters through raw pointers, so torch's per-tensor version counters do not move;
# caches keyed on the weights (the fields' fragment caches, models/tensorf_keyframe.py:_frags) include this counter
GENERATION = 0


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0 or lr < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))

    def next_hyper(self):
        """Advance the step counters by one iteration and return the per-iteration scalars of that step as a list of floats -
        [1/sqrt(1-beta2^t)] + [lr_k/(1-beta1^t) for every tensor k in table order] - for `step(hyper_dev=...)`: the caller uploads them
        and the (possibly hipGraph-captured) launch reads them from device memory.  Same values nvfi_adam_step derives on the host."""
        import numpy as np
        out, head = [], None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    raise _lib.NvfiError("next_hyper() needs initialised state: run one ordinary step() first")
                st["step"] = int(st["step"]) + 1
                # nvfi_adam_step's arithmetic: betas and lr arrive there as C floats, the corrections are formed in double
                b1f, b2f = float(np.float32(b1)), float(np.float32(b2))
                bc1, bc2 = 1.0 - b1f ** float(st["step"]), 1.0 - b2f ** float(st["step"])
                h = float(np.float32(1.0 / np.sqrt(bc2)))
                if head is None:
                    head = (h, st["step"], float(b1), float(b2))
                elif head[1:] != (st["step"], float(b1), float(b2)):
                    raise _lib.NvfiError("step(hyper_dev=...) needs one (betas, step count) for all tensors")
                out.append(float(np.float32(float(np.float32(group["lr"])) / bc1)))
        return [head[0]] + out

    @torch.no_grad()
    def step(self, closure=None, zero_grad=False, hyper_dev=None):
        """hyper_dev (fp32 CUDA tensor holding next_hyper()'s values): the launch takes step size and bias correction from device memory and
        this call leaves the step counters alone (next_hyper() advanced them)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        # tensors that share (betas, eps, step count) go into one launch; the ctypes table is rebuilt only when a pointer changed
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr = float(group["lr"])
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    if g.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                        raise _lib.NvfiError("nvfi_amd.optim.Adam handles dense fp32 GPU parameters")
                    if not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)):
                        raise _lib.NvfiError("parameter must be dense (contiguous or channels_last)")
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif not isinstance(st["step"], int):      # state loaded from a torch.optim.Adam checkpoint
                    st["step"] = int(float(st["step"]))
                    for k in ("exp_avg", "exp_avg_sq"):
                        if st[k].stride() != p.stride() or st[k].device != p.device:
                            st[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[k])
                if hyper_dev is None:
                    st["step"] += 1
                # elementwise update: any dense memory layout works as long as p, g, m, v share it
                if g.stride() != p.stride():
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                    p.grad = g
                batches.setdefault((float(b1), float(b2), float(group["eps"]), st["step"]), []).append((p, g, st, lr))
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        global GENERATION
        GENERATION += 1
        cache = self.__dict__.setdefault("_tables", {})
        for key, items in batches.items():
            b1, b2, eps, step = key
            # every pointer the table holds: a load_state_dict() (or the stride fix above) replaces exp_avg / exp_avg_sq tensors
            sig = tuple(x for (p, g, st, lr) in items for x in (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()))
            ent = cache.get(key[:3])
            if ent is None or ent[0] != sig:
                arr = (_lib.AdamTensor * len(items))()
                for i, (p, g, st, lr) in enumerate(items):
                    arr[i].p = p.data_ptr(); arr[i].g = g.data_ptr(); arr[i].m = st["exp_avg"].data_ptr(); arr[i].v = st["exp_avg_sq"].data_ptr()
                    arr[i].n = p.numel()
                ent = (sig, arr)
                cache[key[:3]] = ent
            arr = ent[1]
            for i, it in enumerate(items):
                arr[i].lr = it[3]
            if hyper_dev is not None:
                if len(batches) != 1 or hyper_dev.numel() < 1 + len(items) or not hyper_dev.is_cuda or hyper_dev.dtype != torch.float32:
                    raise _lib.NvfiError("step(hyper_dev=...) needs one batch and an fp32 CUDA tensor of 1 + n_tensors values")
                _lib.check(L.nvfi_adam_step_dev(arr, C.c_int(len(items)), C.c_float(b1), C.c_float(b2), C.c_float(eps), _lib.ptr(hyper_dev),
                                                C.c_int(1 if zero_grad else 0), stream))
                continue
            _lib.check(L.nvfi_adam_step(arr, C.c_int(len(items)), C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_int64(step),
                                        C.c_int(1 if zero_grad else 0), stream))
        return loss
