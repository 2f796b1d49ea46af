"""TensorVMKeyframeTimeKplane - host-side mirror of the reference field module
(models/tensorf_keyframe.py:37-756 + models/tensorf_base.py:133-269) for the render hot path.

Parameters keep the reference's names and logical shapes (state_dict compatible, including the
duplicated `vel.vel_net.*` keys, tensorf_keyframe.py:94,106); the factor planes are stored
physically channel-last so one bilinear tap is one contiguous vector for the HIP kernels.
Everything between "rays in" and "rgb/depth/acc/weights (+ gradients) out" is one call into
libnvfi_hip.so through ctypes; there is no PyTorch/CPU fallback.
"""
import ctypes as C
import os
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .tensorf_model_utils import AlphaGridMask

# per-field runtime state that must not travel with copy.deepcopy / pickle of the module (HIP streams, events, pinned buffers, the
# gradient arena): kept beside the module, keyed weakly
_RUNTIME = weakref.WeakKeyDictionary()


def _rt(field):
    d = _RUNTIME.get(field)
    if d is None:
        d = _RUNTIME[field] = {}
    return d
from .velocity_field import VelBasis, VelocityAABB, VelocityAABBSur


class MLPRender_PE(nn.Module):
    """Parameter container of the appearance decoder (tensorf_base.py:67-98); evaluated by k_app_fwd."""

    def __init__(self, inChanel, viewpe=6, pospe=6, featureC=128):
        super().__init__()
        self.in_mlpC = (3 + 2 * viewpe * 3) + (3 + 2 * pospe * 3) + inChanel
        self.viewpe, self.pospe = viewpe, pospe
        self.mlp = nn.Sequential(nn.Linear(self.in_mlpC, featureC), nn.ReLU(inplace=True),
                                 nn.Linear(featureC, featureC), nn.ReLU(inplace=True), nn.Linear(featureC, 3))
        nn.init.constant_(self.mlp[-1].bias, 0)
        self._owner = None

    def forward(self, pts, viewdirs, features, kwargs=None):
        """renderModule(pts, viewdirs, features) as the reference's module can be called (tensorf_base.py:88-98): the positional encodings and
        the MLP of the fused appearance kernel on the caller's features (nvfi_render_mlp).  Forward only - inside a render the module is
        differentiated by nvfi_render_bwd; a stand-alone call that needs gradients is refused rather than silently detached."""
        field = self._owner() if callable(self._owner) else self._owner
        if field is None:
            raise _lib.NvfiError("MLPRender_PE.forward needs the field it belongs to (it is created by TensorVMKeyframeTimeKplane)")
        if torch.is_grad_enabled() and (features.requires_grad or pts.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("stand-alone MLPRender_PE.forward is forward-only: call it under torch.no_grad() "
                                      "(gradients of the render module flow through Renderer.render / NVFi.render_ray)")
        return field._render_module_call(pts, viewdirs, features)


def SHRender(xyz_sampled, viewdirs, features, kwargs=None):
    """tensorf_model_utils.py:292-296 as a stand-alone call (inside a render it is fused into k_app_fwd / k_app_bwd)."""
    return TensorVMKeyframeTimeKplane.sh_render(viewdirs, features)


class DeviceTime(float):
    """A frame time whose VALUE lives in device memory: `DeviceTime(t_plan, dev)` is a float (t_plan - it fixes the launch plan: keyframe or
    not, number of RK2 steps, workspace layout) that carries a 1-element fp32 CUDA tensor `dev` the kernels read the time from
    (nvfi_render_fwd_t).  Passing it as `t` to Renderer.render / NVFi.render_ray lets a captured hipGraph of a training iteration be
    replayed with a new time every iteration.  dev[0] must be in the same plan class as t_plan (counters[7] reports a mismatch)."""

    def __new__(cls, t_plan, dev):
        x = super().__new__(cls, float(t_plan))
        if not (isinstance(dev, torch.Tensor) and dev.is_cuda and dev.dtype == torch.float32 and dev.numel() == 1):
            raise _lib.NvfiError("DeviceTime needs a 1-element fp32 CUDA tensor")
        x.dev = dev
        return x


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _RenderFn(torch.autograd.Function):
    """autograd boundary around nvfi_render_fwd / nvfi_render_bwd (include/nvfi_hip.h)."""

    @staticmethod
    def forward(ctx, field, t, rays_o, rays_d, jitter, flags, *params):
        L = _lib.lib()
        R = rays_o.shape[0]
        dev = rays_o.device
        t_dev = getattr(t, "dev", None)        # DeviceTime: the kernels read the time from device memory; float(t) is the launch plan
        t = float(t)
        desc = field._desc()
        S = desc.n_samples
        nbytes = C.c_int64(0)
        _lib.check(L.nvfi_render_workspace_bytes_t(C.byref(desc), C.c_int64(R), C.c_int(flags), C.c_float(t), C.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        rgb = torch.empty(R, 3, device=dev)
        depth = torch.empty(R, device=dev)
        acc = torch.empty(R, device=dev)
        weights = torch.empty(R, S, device=dev)
        counters = torch.empty(_lib.NCOUNTERS, dtype=torch.int64, device=dev)      # all 8 entries are written by the call
        _lib.check(L.nvfi_render_fwd_t(C.byref(desc), C.c_int64(R), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(jitter),
                                       C.c_float(t), _lib.ptr(t_dev), C.c_int(flags), _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(acc),
                                       _lib.ptr(weights), _lib.ptr(ws), C.c_int64(ws.numel()), _lib.ptr(counters), _stream_ptr()))
        field.last_counters = counters
        field._last_ws = ws if (flags & _lib.NVFI_WANT_MASK) else None
        field._last_call = (R, t, flags)
        if flags & _lib.NVFI_TRAIN:
            ctx.field, ctx.t, ctx.flags, ctx.ws, ctx.t_on_device = field, t, flags, ws, t_dev is not None
            ctx.vel_fp16 = int(desc.vel_fp16)      # the workspace layout depends on it: the backward plans with the forward's value
            ctx.save_for_backward(rays_o, rays_d, weights, *params)
        ctx.mark_non_differentiable(counters)
        ctx.set_materialize_grads(False)     # unused outputs (depth, acc, the R x S weights) arrive as None, not as zero tensors
        return rgb, depth, acc, weights, counters

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc, g_weights, _gc):
        L = _lib.lib()
        rays_o, rays_d, weights, *params = ctx.saved_tensors
        field = ctx.field
        desc = field._desc(params)
        desc.vel_fp16 = ctx.vel_fp16
        need = ctx.needs_input_grad[6:]
        inplace = field.accumulate_grads_inplace
        if inplace == "arena":
            # library-managed in-place accumulation (the default under a plain autograd driver): see _arena_attach
            grads = field._arena_attach(field._render_params(), need)
            inplace = grads is not None
            if inplace:
                field._wait_writers()
                field._queue_join()
        elif inplace:
            # kernels accumulate (+=) straight into the parameters' .grad (e.g. views of one flat GradBucket buffer):
            # no zero-fill, no AccumulateGrad add per tensor.  Autograd then sees "no gradient" for these inputs.
            cur = field._render_params()
            grads = []
            for p, n in zip(cur, need):
                if not n:
                    grads.append(None)
                    continue
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                grads.append(p.grad)
        if not inplace:
            grads = _zero_grads(params, need)
        G = field._grads_struct(grads)
        gs = [None if g is None else g.contiguous() for g in (g_rgb, g_depth, g_acc, g_weights)]
        R = rays_o.shape[0]
        _lib.check(L.nvfi_render_bwd_t(C.byref(desc), C.c_int64(R), _lib.ptr(rays_o), _lib.ptr(rays_d), C.c_float(ctx.t), C.c_int(int(ctx.t_on_device)),
                                     C.c_int(ctx.flags), _lib.ptr(weights), _lib.ptr(gs[0]), _lib.ptr(gs[1]), _lib.ptr(gs[2]),
                                     _lib.ptr(gs[3]), C.byref(G), _lib.ptr(ctx.ws), C.c_int64(ctx.ws.numel()), _stream_ptr()))
        ctx.ws = None
        if inplace:
            return (None,) * (6 + len(params))
        return (None, None, None, None, None, None) + tuple(grads)


def _zero_grads(params, need=None):
    """Fresh zero gradients for `params` (None where `need` is false) as views of ONE zero-filled buffer: one fill launch per backward
    node instead of one per parameter (the plain-autograd path of the reference's training loop is bound by the host's launch rate)."""
    need = [True] * len(params) if need is None else list(need)
    offs, total = [], 0
    for p, n in zip(params, need):
        offs.append(total)
        if n:
            total += (p.numel() + 63) // 64 * 64      # 256-byte aligned views (the plane-gradient kernels use 16-byte accesses)
    if total == 0:
        return [None] * len(params)
    ref = next(p for p, n in zip(params, need) if n)
    flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
    # same strides as the parameter (the planes are stored channel-last under an NCHW shape; all parameters are dense)
    return [flat[o:o + p.numel()].as_strided(p.shape, p.stride()) if n else None for p, n, o in zip(params, need, offs)]


class _PdeFn(torch.autograd.Function):
    """nvfi_pde_loss computes the loss AND its parameter gradients in one pass; backward scales them."""

    @staticmethod
    def forward(ctx, field, points, t, *params):
        L = _lib.lib()
        dev = points.device
        P = points.shape[0]
        desc = field._desc()
        nbytes = C.c_int64(0)
        _lib.check(L.nvfi_pde_workspace_bytes(C.byref(desc), C.c_int64(P), C.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        out = torch.zeros(4, device=dev)
        grads = _zero_grads(params)
        G = field._grads_struct_vel(grads)
        counters = torch.zeros(_lib.NCOUNTERS, dtype=torch.int64, device=dev)
        kept = torch.zeros(P, dtype=torch.uint8, device=dev) if field.pde_debug else None
        jac = torch.zeros(field.pde_debug, 6, 4, device=dev) if field.pde_debug else None
        info = (C.c_int64 * 2)()
        # split call (field.pde_split, default on): the VALUE is complete on the current stream after the Jacobian forward, the adjoint pass and
        # the weight gradients run on the field's PDE side stream - under the reference's loop, which waits for the value right after this call
        # (train_nvfi.py:233) and then differentiates the renders, the PDE adjoint overlaps with them; backward() waits for `ctx.done`
        side = field._side_stream("p") if field.pde_split else None
        _lib.check(L.nvfi_pde_loss_split(C.byref(desc), C.c_int64(P), _lib.ptr(points), _lib.ptr(t), C.c_float(1.0), _lib.ptr(out),
                                         C.byref(G), _lib.ptr(ws), C.c_int64(ws.numel()), _lib.ptr(counters),
                                         _lib.ptr(kept), _lib.ptr(jac), C.c_int64(int(field.pde_debug)), info, _stream_ptr(),
                                         C.c_void_p(side.cuda_stream) if side is not None else None))
        ctx.done = None
        if side is not None:
            ctx.done = torch.cuda.Event()
            ctx.done.record(side)
            ws.record_stream(side)
            grads[0]._base.record_stream(side)
        field.last_pde_kept, field.last_pde_jac = kept, jac
        field.last_pde_n_kept = int(info[0])
        field.last_pde_out = out
        field.last_pde_counters = counters
        ctx.save_for_backward(grads[0]._base)            # the one flat buffer behind the views
        ctx.shapes = [(p.shape, p.stride()) for p in params]
        ctx.field = field
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        field = ctx.field
        if ctx.done is not None:
            torch.cuda.current_stream().wait_event(ctx.done)     # the adjoint half of the split call has filled the gradients
        if field.accumulate_grads_inplace == "arena":
            tail = field._arena_attach(field._pde_params(), [True] * 24, want_tail=True)
            if tail is not None and tail.numel() == ctx.saved_tensors[0].numel():
                # one launch: arena[velocity nets] += g * (gradients of the un-weighted loss); a plain read-modify-write, so the
                # backward kernels of the renders (atomic adds from other streams) are ordered behind it (_note_writer)
                field._wait_writers(True)
                tail.addcmul_(ctx.saved_tensors[0], g.to(tail.dtype).expand_as(tail))
                field._note_writer()
                field._queue_join()
                return (None,) * (3 + len(ctx.shapes))
        flat = ctx.saved_tensors[0] * g                  # one launch for all 24 tensors
        out, o = [], 0
        for shp, strides in ctx.shapes:
            n = shp.numel()
            out.append(flat[o:o + n].as_strided(shp, strides))
            o += (n + 63) // 64 * 64
        return (None, None, None) + tuple(out)


class _RegFn(torch.autograd.Function):
    """`density_L1()` / `TV_loss_density(reg)` / `TV_loss_app(reg)` of the reference (tensorf_keyframe.py:188-231) on nvfi_plane_regs (regs.hip)
    instead of ~230 torch launches each.  The kernel computes all three values in one pass over the planes, and a driver calls all three per
    iteration (train_nvfi.py:203-217), so:
      forward   the first call of an iteration runs the pass and keeps the value triple; the next two find the planes unchanged (same tensors,
                same `_version`) and return their entry of it - one pass instead of three;
      backward  with in-place gradient accumulation (arena / caller-owned .grad) the three nodes only deposit their upstream weights; the last
                one - or the engine's end-of-backward callback, if the graph held fewer - runs ONE gradient pass with all of them
                (nvfi_plane_regs_dev: weights read from device memory, no host sync).  Under pure autograd each node has to hand its own
                gradients back to the engine, so each runs its own pass as before."""

    @staticmethod
    def forward(ctx, field, which, *planes):
        L = _lib.lib()
        rt = _rt(field)
        key = tuple((p.data_ptr(), p._version) for p in planes)
        c = rt.get("_reg_fwd")
        if c is None or c[0] != key:
            # a new pass = a new iteration: whatever an aborted backward of the previous one left behind (deposited weights, a live count that
            # never reached zero because its graph was dropped) must not leak into this one's single gradient pass
            rt.pop("_reg_pending", None)
            rt["_reg_live"] = 0
            out = torch.empty(3, device=planes[0].device)
            desc = field._desc()
            _lib.check(L.nvfi_plane_regs(C.byref(desc), C.c_float(0.0), C.c_float(0.0), C.c_float(0.0), _lib.ptr(out), None, _stream_ptr()))
            c = (key, out, torch.cuda.current_stream())
            rt["_reg_fwd"] = c
        elif c[2] != torch.cuda.current_stream():
            torch.cuda.current_stream().wait_stream(c[2])
        ctx.field, ctx.which = field, which
        ctx.save_for_backward(*planes)
        if any(ctx.needs_input_grad):
            rt["_reg_live"] = rt.get("_reg_live", 0) + 1
        return c[1][which].clone()

    @staticmethod
    def _flush(field):
        """one gradient pass for every deposited weight (in-place modes)"""
        rt = _rt(field)
        pend = rt.pop("_reg_pending", None)
        rt["_reg_live"] = 0
        if not pend:
            return
        L = _lib.lib()
        planes, ws = pend["planes"], pend["w"]
        touched = set()
        for which in ws:
            touched |= set((0, 1, 2, 3, 4, 5) if which < 2 else (6, 7, 8))
        inplace = field.accumulate_grads_inplace
        cur = field._render_params()[:9]
        grads = [None] * 9
        arena = False
        if inplace == "arena":
            ag = field._arena_attach(field._render_params(), [k in touched for k in range(9)] + [False] * 22)
            arena = ag is not None
            if arena:
                grads = ag[:9]
                field._wait_writers(True)      # k_plane_regs adds with plain read-modify-writes: not next to another stream's kernels
                field._queue_join()
        if not arena:
            for k in touched:
                if cur[k].grad is None:
                    cur[k].grad = torch.zeros_like(cur[k])
                grads[k] = cur[k].grad
        G = field._grads_struct(grads + [None] * 22)
        dev = planes[0].device
        zero = None
        parts = []
        for which in range(3):
            if which in ws:
                parts.append(ws[which].reshape(1).to(torch.float32))
            else:
                zero = torch.zeros(1, device=dev) if zero is None else zero
                parts.append(zero)
        w3 = torch.cat(parts)
        out = torch.empty(3, device=dev)
        desc = field._desc(list(planes) + field._render_params()[9:])
        _lib.check(L.nvfi_plane_regs_dev(C.byref(desc), _lib.ptr(w3), _lib.ptr(out), C.byref(G), _stream_ptr()))
        if arena:
            field._note_writer()

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        field, which = ctx.field, ctx.which
        planes = ctx.saved_tensors                      # the 9 regularised planes: dps[3], dpt[3], aps[3]
        rt = _rt(field)
        rt["_reg_live"] = max(rt.get("_reg_live", 1) - 1, 0)
        rt.pop("_reg_fwd", None)       # the value triple is shared by the calls of ONE iteration only (a driver that never moves the planes - bench.py's
                                        # stationary mode - must not skip next iteration's pass)
        inplace = field.accumulate_grads_inplace
        if inplace:
            # deposit; the last live node (or the end of this backward pass) runs the single gradient pass.  A field with autograd hooks on a
            # parameter gets pure autograd from _arena_attach -> the per-node path below
            hooked = inplace == "arena" and any(p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None) for p in field._arena_params())
            if not hooked:
                pend = rt.get("_reg_pending")
                if pend is None:
                    pend = rt["_reg_pending"] = dict(planes=planes, w={})
                    ref = weakref.ref(field)
                    torch.autograd.Variable._execution_engine.queue_callback(lambda: ref() is not None and _RegFn._flush(ref()))
                pend["w"][which] = g if which not in pend["w"] else pend["w"][which] + g
                if rt["_reg_live"] == 0:
                    _RegFn._flush(field)
                return (None, None) + (None,) * 9
        touched = (0, 1, 2, 3, 4, 5) if which < 2 else (6, 7, 8)
        grads = _zero_grads(planes, [k in touched for k in range(9)])
        G = field._grads_struct(grads + [None] * 22)
        w3 = torch.zeros(3, device=g.device)
        w3[which] = g
        out = torch.empty(3, device=g.device)
        desc = field._desc(list(planes) + field._render_params()[9:])
        _lib.check(L.nvfi_plane_regs_dev(C.byref(desc), _lib.ptr(w3), _lib.ptr(out), C.byref(G), _stream_ptr()))
        return (None, None) + tuple(grads)


class TensorVMKeyframeTimeKplane(nn.Module):
    def __init__(self, aabb, gridSize, device, near_far, cfg):
        super().__init__()
        self.matModeSpace = [[0, 1], [0, 2], [1, 2]]
        self.matModeTime = [[2, 3], [1, 3], [0, 3]]
        self.cfg = cfg
        self.device = device
        self.num_keyframes = cfg.num_keyframes
        self.tmax = cfg.tmax
        self.time_scale_factor = self.tmax / (self.num_keyframes - 1) if self.num_keyframes > 1 else 1
        self.densityMode = cfg.densityMode
        if self.densityMode != "Density":
            raise NotImplementedError("only densityMode=Density is on the hot path (every shipped config)")
        self.data_dim_density = 1
        self.register_buffer("aabb", aabb.to(device).float())
        self.step_ratio = cfg.step_ratio
        self.max_n_samples = cfg.max_n_samples
        self.near_far = near_far
        self.density_n_comp = cfg.density_n_comp
        self.app_n_comp = cfg.appearance_n_comp
        self.app_dim = cfg.app_dim
        self.density_shift = cfg.density_shift
        self.distance_scale = cfg.distance_scale
        self.alphaMask = None
        self.alphaMask_thres = cfg.alphaMask_thres
        self.rayMarch_weight_thres = cfg.rayMarch_weight_thres
        self.fea2denseAct = cfg.fea2denseAct
        if self.fea2denseAct != "softplus":
            raise NotImplementedError("only fea2denseAct=softplus is on the hot path (every shipped config)")
        self.update_stepSize(gridSize)
        self.init_svd_volume(device)
        self.shadingMode = cfg.shadingMode
        self.pos_pe, self.view_pe, self.fea_pe, self.featureC = cfg.pos_pe, cfg.view_pe, cfg.fea_pe, cfg.featureC
        if self.shadingMode == "SH":
            # tensorf_base.py:196-197: renderModule is the FUNCTION SHRender (no parameters); basis_mat maps the 48 plane products to 27 SH coefficients
            if self.app_dim != 27:
                raise NotImplementedError("shadingMode=SH needs app_dim=27 (3 colours x 9 degree-2 SH coefficients)")
            self.renderModule = SHRender
        elif self.shadingMode == "MLP_PE":
            if self.pos_pe != 6 or self.view_pe != 6 or self.featureC != 128 or self.app_dim != 32:
                raise NotImplementedError("shadingMode=MLP_PE is built for pos_pe=view_pe=6, featureC=128, app_dim=32 (every shipped config)")
            self.renderModule = MLPRender_PE(self.app_dim, self.view_pe, self.pos_pe, self.featureC).to(device)
            self.renderModule._owner = weakref.ref(self)      # renderModule(pts, viewdirs, features) evaluates on this field's kernels
        else:
            raise NotImplementedError("shadingMode must be MLP_PE or SH (MLP_Fea / MLP / RGB* are not on the hot path: no shipped config uses them)")
        self.use_vel = bool(cfg.use_vel)
        if self.use_vel:
            self.vel_net = VelBasis().to(device)
            self.vel_net._owner = weakref.ref(self)
            eps = cfg.eps if "eps" in cfg else 0.03
            if all(k in cfg for k in ("sur_x", "sur_y", "sur_z")):
                sur = torch.stack([torch.tensor(cfg[k]) for k in ("sur_x", "sur_y", "sur_z")], dim=-1).to(device)
                self.vel = VelocityAABBSur(self.vel_net, self.aabb.clone(), sur)
            else:
                self.vel = VelocityAABB(self.vel_net, eps)
        self.mask_field = None
        self.contract_ray = bool("contract_ray" in cfg and cfg.contract_ray)
        if self.contract_ray:
            raise NotImplementedError("contract_ray is out of scope")
        self.last_counters = None
        # Where the backward kernels put parameter gradients:
        #   False (default)    pure autograd: gradients are returned to the engine - torch.autograd.grad(), parameter hooks and
        #                      post-accumulate-grad hooks (DDP, optimiser-in-backward, clipping hooks) behave as for any nn.Module
        #   "arena" (opt-in: NVFI_INPLACE_GRADS=arena, field.accumulate_grads_inplace = "arena"; bench.py --mode dropin and
        #                      tools/run_reference_driver.py switch it on) straight into p.grad, which the field backs with ONE persistent
        #                      flat buffer that it zeroes once per iteration (when the driver's zero_grad(set_to_none=True) left every
        #                      .grad None): no fresh 38 MB gradient tensors and no AccumulateGrad add per parameter and backward node.
        #                      p.grad tensors are then re-used across iterations (aliasing: keep a .clone() if you need last iteration's
        #                      values) and the engine is handed None for the parameters; a field with ANY autograd hook on a parameter
        #                      falls back to pure autograd for that call
        #   True               same, but the caller owns p.grad (nvfi_amd.dist.GradBucket: every .grad is a view of its flat buffer)
        self.accumulate_grads_inplace = "arena" if os.environ.get("NVFI_INPLACE_GRADS", "0") in ("1", "arena") else False
        # Opt-in (NVFI_AUTO_OVERLAP=1): train-mode render / PDE calls run on the field's own side streams (the caller's stream waits for
        # the results), so that the chains of an iteration - and their backward passes, which autograd runs on the stream of the forward -
        # can overlap on the device under a sequential driver.  MEASURED SLOWER under the reference's loop (bench.py --mode dropin:
        # 372-438 k rays/s against 481 k on one stream): that loop waits for the device three times per iteration, so it is bound by the
        # host's issue time between the waits, and the extra cross-stream waits / record_stream bookkeeping only add to it.  Off by default.
        self.auto_overlap = os.environ.get("NVFI_AUTO_OVERLAP", "0") == "1"
        self.pde_debug = 0   # >0: also return the kept mask and the first n Jacobians of get_vel_loss
        # the backward of a keyframe render may run its density half beside its appearance half on a library-owned stream (NVFI_BWD_FORK);
        # a driver that already overlaps renders / the PDE term on its own streams (bench.py's fused step) switches this off
        self.fork_backward = os.environ.get("NVFI_BWD_FORK", "1") != "0"
        self.pde_split = os.environ.get("NVFI_PDE_SPLIT", "1") != "0"     # pde_loss(): adjoint half of the call on a side stream (nvfi_pde_loss_split)
        # opt-in (not a reference attribute; the reference's counterpart is autocast via --disable_fp32, train_nvfi.py:96,144): every no-grad
        # back-advection - integrate_pos, the warp of eval-mode renders, getDenseAlpha - on the fp16-input MFMA (nvfi_field_desc.vel_fp16).
        # Training renders, the PDE term and all gradients stay fp32 whatever this says.
        # True / 1: one binary16 term per MFMA operand; 2 / "split": two terms (fp32 products emulated, ~2^-21 relative per product, 2.4x fp32 speed)
        self.vel_fp16 = {"0": False, "1": True, "2": 2, "split": 2, "3": 3, "x6": 3}.get(os.environ.get("NVFI_VEL_FP16", "0"), False)
        # opt-in (round 4; the reference's counterpart: --disable_fp32 autocasts the forward of every nn.Linear, train_nvfi.py:96,144): the velocity
        # warp of TRAINING renders evaluates VelBasis forward with fp16-input MFMAs (fp32 accumulation, fp32 stashes); its adjoint and the
        # weight gradients stay fp32 MFMA on those stashes, the PDE term and the render MLP stay fp32.  Never the default, never the headline.
        self.vel_fp16_train = os.environ.get("NVFI_VEL_FP16_TRAIN", "0") == "1"
        self.register_load_state_dict_post_hook(lambda m, k: m.update_stepSize(m.gridSize.tolist()))

    # ------------------------------------------------------------------ construction
    def update_stepSize(self, gridSize):
        """tensorf_base.py:214-227"""
        self.aabbSize = self.aabb[1] - self.aabb[0]
        self.invaabbSize = 2.0 / self.aabbSize
        self.gridSize = torch.LongTensor([int(g) for g in gridSize]).to(self.aabb.device)
        self.units = self.aabbSize / (self.gridSize - 1)
        self.stepSize = torch.mean(self.units) * self.step_ratio
        self.aabbDiag = torch.sqrt(torch.sum(torch.square(self.aabbSize)))
        self.nSamples = min(self.max_n_samples, int((self.aabbDiag / self.stepSize).item()) + 1)
        # host copies of the scalars the C ABI needs (no device sync per render call)
        self._aabb_host = [float(v) for v in self.aabb.detach().reshape(-1).cpu().tolist()]
        self._grid_host = [int(g) for g in gridSize]
        self._step_host = float(self.stepSize)

    def _init_planes(self, n_component, gridSize, numFrames, scale, device):
        """tensorf_keyframe.py:136-186 (softplus branch): uniform(0.1,0.5)*scale space planes, ones time planes."""
        space, time = [], []
        for i in range(3):
            a, b = self.matModeSpace[i]
            c = self.matModeTime[i][0]
            p = torch.empty(1, n_component[i], int(gridSize[b]), int(gridSize[a]))
            nn.init.uniform_(p, a=0.1, b=0.5)
            space.append(nn.Parameter(_cl(scale * p)))
            time.append(nn.Parameter(_cl(torch.ones(1, n_component[i], numFrames, int(gridSize[c])))))
        return nn.ParameterList(space).to(device), nn.ParameterList(time).to(device)

    def init_svd_volume(self, device):
        gs = self.gridSize.tolist()
        self.density_plane_space, self.density_plane_time = self._init_planes(self.density_n_comp, gs, self.num_keyframes, 0.8, device)
        self.app_plane_space, self.app_plane_time = self._init_planes(self.app_n_comp, gs, self.num_keyframes, 0.1, device)
        self.basis_mat = nn.Linear(self.app_n_comp[0], self.app_dim, bias=False).to(device)
        self.basis_mat_density = nn.Linear(self.density_n_comp[0], self.data_dim_density, bias=False).to(device)
        self._fix_layout()

    def _fix_layout(self):
        for pl in (self.density_plane_space, self.density_plane_time, self.app_plane_space, self.app_plane_time):
            for p in pl:
                if not p.data.is_contiguous(memory_format=torch.channels_last):
                    p.data = _cl(p.data)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        _rt(self).pop("_rp_cache", None); _rt(self).pop("_pp_cache", None)
        self._fix_layout()
        self.device = self.aabb.device
        for name in ("gridSize", "aabbSize", "invaabbSize", "units", "stepSize", "aabbDiag"):
            if hasattr(self, name) and isinstance(getattr(self, name), torch.Tensor):
                setattr(self, name, getattr(self, name).to(self.aabb.device))
        if self.use_vel and isinstance(self.vel, VelocityAABBSur):
            self.vel.bounds = self.vel.bounds.to(self.aabb.device)
        return r

    # ------------------------------------------------------------------ parameter plumbing
    def _param_key(self):
        d = self.__dict__["_modules"]
        lists = (d["density_plane_space"], d["density_plane_time"], d["app_plane_space"], d["app_plane_time"])
        return tuple(id(p) for pl in lists for p in pl._parameters.values()) + (self.use_vel, self.shadingMode, id(d.get("vel_net")), id(d.get("renderModule")))

    def _render_params(self):
        """The 31 tensors of the render path in nvfi_field_desc order.  Walking the module tree costs ~90 us of host time per call (and a
        step makes seven of them), so the list is cached and re-validated by the identities of the plane parameters / sub-modules
        (shrink / upsample_volume_grid replace them; `.to()` keeps them)."""
        key = self._param_key()
        c = _rt(self).get("_rp_cache")
        if c is not None and c[0] == key:
            return list(c[1])
        ps = self._render_params_walk()
        _rt(self)["_rp_cache"] = (key, tuple(ps))
        return ps

    def _render_params_walk(self):
        ps = list(self.density_plane_space) + list(self.density_plane_time) + list(self.app_plane_space) + list(self.app_plane_time)
        ps.append(self.basis_mat.weight)
        if self.shadingMode == "SH":
            ps += [None] * 6                      # no render MLP: the six slots stay empty so that the velocity slots keep their index
        else:
            for i in (0, 2, 4):
                ps += [self.renderModule.mlp[i].weight, self.renderModule.mlp[i].bias]
        if self.use_vel:
            for lin in VelBasis.linears(self.vel_net.weight_net):
                ps += [lin.weight, lin.bias]
        return ps

    def _pde_params(self):
        c = _rt(self).get("_pp_cache")
        vn = self.__dict__["_modules"].get("vel_net")
        if c is not None and c[0] == id(vn):
            return list(c[1])
        ps = self._pde_params_walk()
        _rt(self)["_pp_cache"] = (id(vn), tuple(ps))
        return ps

    def _pde_params_walk(self):
        ps = []
        for net in (self.vel_net.weight_net, self.vel_net.a_weight_net):
            for lin in VelBasis.linears(net):
                ps += [lin.weight, lin.bias]
        return ps

    def _gate(self):
        if not self.use_vel:
            return 0, [-1.0] * 3, [1.0] * 3
        if isinstance(self.vel, VelocityAABBSur):
            if getattr(self, "_sur_host", None) is None:
                b = self.vel.bounds.detach().float().cpu().numpy()
                self._sur_host = (b[0].tolist(), b[1].tolist())
            return 1, self._sur_host[0], self._sur_host[1]
        eps = float(self.vel.eps)
        return 0, [float(np.float32(-1 + eps))] * 3, [float(np.float32(1 - eps))] * 3

    def __deepcopy__(self, memo):
        """a copy's sub-modules (vel_net, renderModule) must evaluate on the COPY's kernels / descriptor, not on the original's:
        the weak back-references are re-pointed after the default deep copy"""
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        for child in (getattr(new, "vel_net", None), getattr(new, "renderModule", None)):
            if isinstance(child, nn.Module) and "_owner" in child.__dict__:
                child.__dict__["_owner"] = weakref.ref(new)
        return new

    def _host_state(self):
        """every host-side attribute nvfi_field_desc's scalar fields are derived from (attributes a caller may assign at any time)"""
        am = self.alphaMask
        vel = self.__dict__["_modules"].get("vel") if self.use_vel else None
        return (self.nSamples, self._step_host, tuple(self._aabb_host), tuple(self._grid_host), self.num_keyframes, self.use_vel, self.shadingMode,
                self.vel_fp16, self.__dict__.get("vel_fp16_train", False), self.density_shift, self.distance_scale, self.rayMarch_weight_thres,
                self.alphaMask_thres, self.tmax, self.near_far[0], self.near_far[1], self.app_dim,
                0 if am is None else am.alpha_volume.data_ptr(), id(vel), getattr(vel, "eps", None) if vel is not None else None)

    def _desc(self, params=None):
        """nvfi_field_desc for the current parameters (or for the tensors saved by autograd).  Building the ctypes struct costs ~25 us of
        host time and a training iteration makes eight of them, so the struct is cached and re-validated by the parameters' device
        pointers and the host attributes it is derived from; the caller gets a private copy (the backward edits vel_fp16).  `frags`
        (ABI v5) points at the field's fragment cache when that is current for these weights (_frags)."""
        ps = self._render_params() if params is None else list(params)
        pp = self._pde_params() if self.use_vel else []
        key = (tuple(0 if p is None else p.data_ptr() for p in ps), tuple(p.data_ptr() for p in pp[12:]), self._host_state())
        rt = _rt(self)
        c = rt.get("_desc_cache")
        if c is None or c[0] != key:
            c = (key, self._desc_build(ps, pp))
            rt["_desc_cache"] = c
        d = _lib.FieldDesc.from_buffer_copy(c[1])
        d.frags = self._frags(d, ps, pp)
        return d

    def _desc_build(self, ps, pp):
        d = _lib.FieldDesc()
        d.G[:] = self._grid_host
        d.K = int(self.num_keyframes)
        d.Cd, d.Ca, d.app_dim = int(self.density_n_comp[0]), int(self.app_n_comp[0]), int(self.app_dim)
        d.shading = 1 if self.shadingMode == "SH" else 0
        # 3 / "x6": every no-grad back-advection on the x6 kernels (vel_x6.hip: fp32 products formed exactly from three binary16 terms per operand)
        vf = self.vel_fp16
        # "fp32" (bit 3): keep the fp32 MFMA kernels for integrate_pos / compute_alpha, whose default since round 6 is x6 too (the A/B reference)
        d.vel_fp16 = ((8 if vf == "fp32" else 3 if (vf == 3 or vf == "x6") else (2 if vf in (2, "split", "split16") else (1 if vf else 0)))
                      | (4 if getattr(self, "vel_fp16_train", False) else 0))
        d.n_samples = int(self.nSamples)
        d.use_vel = int(self.use_vel)
        gsur, lo, hi = self._gate()
        d.gate_sur = gsur
        d.gate_lo[:] = lo
        d.gate_hi[:] = hi
        d.aabb[:] = self._aabb_host
        d.near_, d.far_ = float(self.near_far[0]), float(self.near_far[1])
        d.step_size = self._step_host
        d.density_shift, d.distance_scale = float(self.density_shift), float(self.distance_scale)
        d.weight_thres, d.alpha_thres, d.tmax = float(self.rayMarch_weight_thres), float(self.alphaMask_thres), float(self.tmax)
        for p in ps[:12]:
            if not p.is_contiguous(memory_format=torch.channels_last):
                raise _lib.NvfiError("factor planes must be channels_last (call field._fix_layout())")
        for i in range(3):
            d.dps[i] = _lib.ptr(ps[i]); d.dpt[i] = _lib.ptr(ps[3 + i]); d.aps[i] = _lib.ptr(ps[6 + i]); d.apt[i] = _lib.ptr(ps[9 + i])
        d.basis = _lib.ptr(ps[12])
        for i in range(3):
            d.rW[i] = _lib.ptr(ps[13 + 2 * i]); d.rb[i] = _lib.ptr(ps[14 + 2 * i])
        if self.use_vel:
            if len(ps) >= 31:
                for i in range(6):
                    d.vW[i] = _lib.ptr(ps[19 + 2 * i]); d.vb[i] = _lib.ptr(ps[20 + 2 * i])
            for i in range(6):
                d.aW[i] = _lib.ptr(pp[12 + 2 * i]); d.ab[i] = _lib.ptr(pp[13 + 2 * i])
        if self.alphaMask is not None:
            d.has_amask = 1
            v = self.alphaMask.alpha_volume
            d.am_dims[:] = [v.shape[-1], v.shape[-2], v.shape[-3]]
            d.amask = _lib.ptr(v)
        d.frags = None
        return d

    # ------------------------------------------------------------------ fragment cache (ABI v5: nvfi_pack_frags)
    def _frags(self, d, ps, pp):
        """Device pointer of the field's fragment cache, current for the weights `d` describes - or None (the call then repacks into its
        own workspace).  The cache is keyed on (data_ptr, _version) of every nn.Linear tensor it is built from plus the generation counter
        of nvfi_amd.optim.Adam (whose HIP launch does not pass through torch's version counter): the first call after an optimiser step
        repacks - ONE launch on the current stream instead of 7 across the iteration's calls - and later calls on other streams wait for
        that launch once.  An in-place edit through `.data` bumps neither: call invalidate_frags().  NVFI_FRAG_CACHE=0 switches it off."""
        rt = _rt(self)
        on = rt.get("_frag_on")
        if on is None:
            on = rt["_frag_on"] = os.environ.get("NVFI_FRAG_CACHE", "1") != "0"
        if not on or not ps[0].is_cuda:
            return None
        from .. import optim as _optim
        lin = [p for p in ps[12:] if p is not None] + list(pp[12:])
        key = (tuple(p.data_ptr() for p in lin), tuple(p._version for p in lin), _optim.GENERATION, rt.get("_frag_epoch", 0))
        fc = rt.get("_frag_cache")
        cur = torch.cuda.current_stream()
        # Under hipGraph capture the pointer is BAKED into the graph: a cache that is current now says nothing about the replays, which run
        # after optimiser steps the key will never see.  So while capturing, the cache is used only if its pack launch is a node
        # of THIS capture (repack_frags() at the head of the captured iteration; forked streams share the capture id) - or if the driver
        # vouches that it replays the graph holding the pack node first (frags_packed_by_earlier_graph, the three-graph multi-rank step);
        # everything else packs per call, whatever the key says.
        cap = _lib.capture_id(cur.cuda_stream) if torch.cuda.is_current_stream_capturing() else 0
        repack_now = bool(rt.get("_frag_capture_ok"))
        if cap and not repack_now and not rt.get("_frag_trust_graph"):
            if fc is None or fc["key"] != key or fc.get("capture") != cap:
                return None
        if repack_now or fc is None or fc["key"] != key or fc["buf"].device != ps[0].device:
            if cap and not repack_now:
                return None         # (a captured iteration repacks explicitly at its head: repack_frags(); anything else packs per call)
            L = _lib.lib()
            buf = fc["buf"] if fc is not None and fc["buf"].device == ps[0].device else None
            if buf is None:
                nb = C.c_int64(0)
                _lib.check(L.nvfi_frag_cache_bytes(C.byref(d), C.byref(nb)))
                buf = torch.empty(nb.value, dtype=torch.uint8, device=ps[0].device)
            d.frags = None
            _lib.check(L.nvfi_pack_frags(C.byref(d), _lib.ptr(buf), C.c_int64(buf.numel()), C.c_void_p(cur.cuda_stream)))
            ev = torch.cuda.Event()
            ev.record(cur)
            fc = rt["_frag_cache"] = dict(key=key, buf=buf, event=ev, stream=cur, waited=set(), capture=cap)
            rt["_frag_capture_ok"] = False
        elif cur != fc["stream"] and cur.cuda_stream not in fc["waited"]:
            cur.wait_event(fc["event"])
            fc["waited"].add(cur.cuda_stream)
        return fc["buf"].data_ptr()

    def render_workspace_bytes(self, R, t, train=False, transfer=False):
        """bytes nvfi_render_fwd[_t] plans for R rays at time t (include/nvfi_hip.h: nvfi_render_workspace_bytes_t)"""
        nb = C.c_int64(0)
        desc = self._desc()
        flags = (_lib.NVFI_TRAIN if train else 0) | (_lib.NVFI_TRANSFER if transfer else 0)
        _lib.check(_lib.lib().nvfi_render_workspace_bytes_t(C.byref(desc), C.c_int64(int(R)), C.c_int(flags), C.c_float(float(t)), C.byref(nb)))
        return int(nb.value)

    def invalidate_frags(self):
        """the weights were edited behind torch's version counters (p.data.copy_(), a raw kernel): the next call repacks"""
        rt = _rt(self)
        rt["_frag_epoch"] = rt.get("_frag_epoch", 0) + 1

    def repack_frags(self):
        """Repack the fragment cache NOW on the current stream (one launch) whatever its key says, and leave it current: a driver that forks
        the iteration's chains over several streams - or captures the iteration as a hipGraph - calls this at the head of the iteration.
        Under capture the pack launch becomes a node of the graph and only calls of the SAME capture read the cache (_frags)."""
        self.invalidate_frags()
        rt = _rt(self)
        rt["_frag_capture_ok"] = True
        try:
            self._desc()
        finally:
            rt["_frag_capture_ok"] = False

    def frags_packed_by_earlier_graph(self, on=True):
        """A driver that captures the iteration as SEVERAL graphs (bench.py: the multi-rank step) packs in the first and replays it before
        the others: it tells the field so for the duration of the later captures - the only case in which a captured call may read a cache
        whose pack launch is not a node of its own capture."""
        _rt(self)["_frag_trust_graph"] = bool(on)

    def _grads_struct(self, grads):
        G = _lib.Grads()
        g = list(grads) + [None] * (31 - len(grads))
        for i in range(3):
            G.dps[i] = _lib.ptr(g[i]); G.dpt[i] = _lib.ptr(g[3 + i]); G.aps[i] = _lib.ptr(g[6 + i]); G.apt[i] = _lib.ptr(g[9 + i])
        G.basis = _lib.ptr(g[12])
        for i in range(3):
            G.rW[i] = _lib.ptr(g[13 + 2 * i]); G.rb[i] = _lib.ptr(g[14 + 2 * i])
        for i in range(6):
            G.vW[i] = _lib.ptr(g[19 + 2 * i]); G.vb[i] = _lib.ptr(g[20 + 2 * i])
        return G

    def _grads_struct_vel(self, grads):
        G = _lib.Grads()
        for i in range(6):
            G.vW[i] = _lib.ptr(grads[2 * i]); G.vb[i] = _lib.ptr(grads[2 * i + 1])
            G.aW[i] = _lib.ptr(grads[12 + 2 * i]); G.ab[i] = _lib.ptr(grads[13 + 2 * i])
        return G

    # ------------------------------------------------------------------ gradient arena + side streams (plain-autograd drivers)
    def _arena_params(self):
        ps = self._render_params()
        if self.use_vel:
            ps = ps + self._pde_params()[12:]           # + the acceleration net: the PDE parameters are then the last 24 entries
        return [p for p in ps if p is not None]

    def _arena_attach(self, params, need, want_tail=False):
        """Gradient targets for `params` (None where `need` is false) as views of the field's persistent flat buffer.  A parameter whose
        .grad is None gets its (zeroed) view attached; when that is true for every parameter of the field - the state a driver's
        `zero_grad(set_to_none=True)` leaves behind - the whole buffer is cleared with ONE fill.  A .grad the caller put there is used
        as it is when its layout matches the parameter's; otherwise None is returned and the call falls back to pure autograd.
        want_tail: return the flat view that covers the 24 velocity-net tensors instead (they are contiguous at the end)."""
        ps = self._arena_params()
        # a hook on any parameter (tensor hooks, post-accumulate-grad hooks: DDP, optimiser-in-backward, clipping) only fires when the
        # engine itself accumulates the gradient: such a field gets pure autograd
        for p in ps:
            if p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
                return None
        key = tuple((id(p), tuple(p.shape), p.stride()) for p in ps)
        a = _rt(self).get("_arena")
        if a is None or a["key"] != key or a["flat"].device != ps[0].device:
            offs, total = [], 0
            for p in ps:
                offs.append(total)
                total += (p.numel() + 63) // 64 * 64          # 256-byte aligned views, same packing as _zero_grads
            flat = torch.zeros(total, dtype=torch.float32, device=ps[0].device)
            views = {id(p): flat[o:o + p.numel()].as_strided(p.shape, p.stride()) for p, o in zip(ps, offs)}
            a = dict(key=key, flat=flat, views=views, offs=offs, params=ps, event=None, tail=flat[offs[-24]:] if (self.use_vel and len(ps) >= 24) else None)
            _rt(self)["_arena"] = a
        # only the parameters THIS node differentiates are attached: a parameter no term of the loss reaches keeps .grad = None and the
        # optimiser skips it, exactly as under pure autograd
        wanted = {id(p) for p, n in zip(params, need) if n and p is not None}
        idx = [i for i, p in enumerate(a["params"]) if id(p) in wanted and p.requires_grad and p.grad is None]
        if a["event"] is not None:
            torch.cuda.current_stream().wait_event(a["event"])      # fills queued by nodes that ran on other streams
        if idx:
            lo, hi = idx[0], idx[-1]
            with torch.no_grad():
                if len(idx) == hi - lo + 1:        # a contiguous run of the buffer (the usual case): ONE fill
                    end = a["offs"][hi + 1] if hi + 1 < len(a["offs"]) else a["flat"].numel()
                    a["flat"][a["offs"][lo]:end].zero_()
                else:
                    for i in idx:
                        a["views"][id(a["params"][i])].zero_()
            for i in idx:
                p = a["params"][i]
                p.grad = a["views"][id(p)]
            ev = torch.cuda.Event()
            ev.record()
            a["event"] = ev            # other streams' backward nodes order themselves behind the fill
        out = []
        for p, n in zip(params, need):
            if not n or p is None:
                out.append(None)
                continue
            g = p.grad
            if g is None or g.stride() != p.stride() or g.dtype != torch.float32 or g.device != p.device:
                return None
            out.append(g)
        if want_tail:
            vs = self._pde_params()
            ok = a["tail"] is not None and all(p.grad is a["views"].get(id(p)) for p in vs)
            return a["tail"] if ok else None
        return out

    def _side_stream(self, kind):
        pool = _rt(self).get("_side_pool")
        if pool is None:
            dev = self.aabb.device
            pool = dict(r=[torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)], p=torch.cuda.Stream(device=dev), k=0)
            _rt(self)["_side_pool"] = pool
        if kind == "p":
            return pool["p"]
        pool["k"] += 1
        return pool["r"][pool["k"] & 1]

    def _use_side_streams(self):
        return (self.auto_overlap and self.accumulate_grads_inplace == "arena" and self.aabb.is_cuda and torch.is_grad_enabled()
                and not torch.cuda.is_current_stream_capturing())

    def _note_writer(self):
        """The current stream just queued a non-atomic read-modify-write of gradient memory: later backward nodes (any stream) wait for it."""
        ev = torch.cuda.Event()
        ev.record()
        _rt(self).setdefault("_writers", []).append(ev)

    def _wait_writers(self, also_side_streams=False):
        cur = torch.cuda.current_stream()
        for ev in _rt(self).get("_writers", ()):
            cur.wait_event(ev)
        if also_side_streams:          # a non-atomic writer must not run next to ANY other backward kernel of the field
            pool = _rt(self).get("_side_pool")
            if pool is not None:
                for st in pool["r"] + [pool["p"]]:
                    if st != cur:
                        cur.wait_stream(st)

    def _queue_join(self):
        """Once per backward pass: when the engine has run every node, the stream of the thread that called backward() waits for the
        field's side streams - the optimiser step that follows then sees complete gradients (the engine itself only orders the
        gradients it is handed, and in-place accumulation hands it none)."""
        if _rt(self).get("_join_pending"):
            return
        _rt(self)["_join_pending"] = True
        ref = weakref.ref(self)

        def join():
            f = ref()
            if f is None:
                return
            _rt(f)["_join_pending"] = False
            _rt(f)["_writers"] = []
            pool = _rt(f).get("_side_pool")
            if pool is not None:
                cur = torch.cuda.current_stream()
                for st in pool["r"] + [pool["p"]]:
                    cur.wait_stream(st)

        from torch.autograd import Variable
        Variable._execution_engine.queue_callback(join)

    # ------------------------------------------------------------------ hot path
    def forward(self, t, ray_o, ray_d, white_bg=True, ndc_ray=False, N_samples=-1, transfer_vel=False):
        """render one chunk of rays (tensorf_keyframe.py:613-755) -> rgb, depth, acc, weights, mask_map."""
        if ndc_ray:
            raise NotImplementedError("ndc rays are out of scope (ndc: False in every shipped config)")
        if N_samples > 0 and N_samples != self.nSamples:
            raise NotImplementedError("per-call N_samples override is not supported")
        if not ray_o.is_cuda or not self.aabb.is_cuda:
            raise _lib.NvfiError("NVFi HIP kernels need the field and the rays on the GPU (no CPU fallback exists)")
        ray_o = ray_o.reshape(-1, 3).contiguous().float()
        ray_d = ray_d.reshape(-1, 3).contiguous().float()
        R = ray_o.shape[0]
        training = self.training
        flags = 0
        jitter = None
        if training:
            flags |= _lib.NVFI_TRAIN
            if self.fork_backward:
                flags |= _lib.NVFI_BWD_FORK
            # the reference draws the per-ray jitter on the CPU generator (tensorf_base.py:302-306)
            jitter = self._jitter(R, ray_o.device)
        # white background or the training-time random-white coin, drawn on CPU (tensorf_keyframe.py:740)
        if white_bg or (training and bool(torch.rand((1,)) < 0.5)):
            flags |= _lib.NVFI_WHITE_BG
        if transfer_vel:
            flags |= _lib.NVFI_TRANSFER
        if self.mask_field is not None:
            flags |= _lib.NVFI_WANT_MASK
        t_dev = getattr(t, "dev", None)
        t = float(np.float32(float(t)))
        if t_dev is not None:
            if self.mask_field is not None:
                raise NotImplementedError("DeviceTime renders do not support the mask branch")
            t = DeviceTime(t, t_dev)
        params = self._render_params()
        if training and self._use_side_streams():
            # the render runs on one of the field's two render streams (ordered behind everything queued on the caller's stream so far,
            # e.g. the last optimiser step); the caller's stream waits for the outputs.  Autograd runs the backward on this stream too.
            cur = torch.cuda.current_stream()
            side = self._side_stream("r")
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                rgb, depth, acc, weights, cnt = _RenderFn.apply(self, t, ray_o, ray_d, jitter, flags, *params)
            cur.wait_stream(side)
            for x in (rgb, depth, acc, weights, cnt):
                x.record_stream(cur)
            for x in (ray_o, ray_d, jitter):
                if x is not None:
                    x.record_stream(side)
        elif training and torch.is_grad_enabled():
            rgb, depth, acc, weights, _ = _RenderFn.apply(self, t, ray_o, ray_d, jitter, flags, *params)
        else:
            with torch.no_grad():
                rgb, depth, acc, weights, _ = _RenderFn.apply(self, t, ray_o, ray_d, jitter, flags, *params)
        if self.mask_field is None:
            # (R, 3) zeros like the reference's placeholder (tensorf_keyframe.py:673-676 without a mask_field) - a stride-0 view of one cached
            # zero instead of a fill launch per render; nobody writes into it
            z = _rt(self).get("_zero1")
            if z is None or z.device != ray_o.device:
                z = _rt(self)["_zero1"] = torch.zeros(1, 1, device=ray_o.device)
            mask_map = z.expand(R, 3)
        elif training and torch.is_grad_enabled():
            mask_map = self._mask_map_train(R, weights)
        else:
            mask_map = self._mask_map(t, flags, R, weights)
        return rgb, depth, acc, weights, mask_map

    def _mask_map_train(self, R, weights):
        """Train-mode mask branch (tensorf_keyframe.py:673-676, 749-753), differentiable like the reference's: the appearance-masked
        samples of the render call are exported (warped keyframe positions), MaskField runs through its own autograd boundary
        (nvfi_maskfield_fwd / _bwd), and the composite sum_j w_j mask_j is a torch index_add - so gradients reach the MaskField
        parameters and, through `weights`, the density branch of the field.  (The points themselves carry no gradient: MaskField's
        kernels do not differentiate their input, as in train_segm.py where they are computed under no_grad.  No reference script
        differentiates this branch: test_segm_render.py:96 renders in test mode.)"""
        L = _lib.lib()
        R_, t, flags = self._last_call
        desc = self._desc()
        M = int(self.last_counters[2])          # host value: sizes the export (the reference syncs here too: `if app_mask.any()`)
        K = self.mask_field.mask_dim
        dev = weights.device
        if M == 0:
            return weights.sum(-1, keepdim=True) * torch.zeros(1, K, device=dev)
        xyz = torch.empty(M, 3, device=dev)
        idx = torch.empty(M, dtype=torch.int64, device=dev)
        ws = self._last_ws
        _lib.check(L.nvfi_render_export_masked(C.byref(desc), C.c_int64(R), C.c_float(t), C.c_int(flags), _lib.ptr(ws), C.c_int64(ws.numel()),
                                               C.c_int64(M), _lib.ptr(xyz), _lib.ptr(idx), _stream_ptr()))
        mask = self.mask_field(xyz)
        S = weights.shape[1]
        w = weights.reshape(-1)[idx]
        return torch.zeros(R, K, device=dev).index_add(0, idx // S, w[:, None] * mask)

    @torch.no_grad()
    def _mask_map(self, t, flags, R, weights):
        """mask branch of render_pts (tensorf_keyframe.py:673-676, 749-753); inference only."""
        L = _lib.lib()
        mf = self.mask_field
        md = _lib.MaskDesc()
        md.n_layer, md.n_dim, md.mask_dim = len(mf.point_fc), mf.point_fc[0].out_features, mf.mask_dim
        lins = list(mf.point_fc) + [mf.mask_fc]
        for i, lin in enumerate(lins):
            md.W[i] = _lib.ptr(lin.weight); md.b[i] = _lib.ptr(lin.bias)
        out = torch.empty(R, mf.mask_dim, device=weights.device)
        desc = self._desc()
        ws = self._last_ws
        _lib.check(L.nvfi_render_mask(C.byref(desc), C.byref(md), C.c_int64(R), C.c_float(t), C.c_int(flags), _lib.ptr(weights.detach()),
                                      _lib.ptr(out), _lib.ptr(ws), C.c_int64(ws.numel()), _stream_ptr()))
        return out

    # ------------------------------------------------------------------ building blocks (inference, no autograd)
    def normalize_coord(self, xyz_sampled):
        return (xyz_sampled - self.aabb[0]) * self.invaabbSize - 1

    def normalize_time_coord(self, time):
        if self.num_keyframes == 1 or self.tmax == 0:
            return time * 0
        return time * 2 / self.tmax - 1

    @torch.no_grad()
    def _vel_eval(self, xt, gated):
        L = _lib.lib()
        xt = xt.reshape(-1, 4).contiguous().float()
        N = xt.shape[0]
        desc = self._desc()
        nb = C.c_int64(0)
        _lib.check(L.nvfi_vel_workspace_bytes(C.byref(desc), C.c_int64(N), C.byref(nb)))
        ws = torch.empty(nb.value, dtype=torch.uint8, device=xt.device)
        u = torch.zeros(N, 6, device=xt.device)
        _lib.check(L.nvfi_vel_eval(C.byref(desc), C.c_int64(N), _lib.ptr(xt), _lib.ptr(u), C.c_int(int(gated)), _lib.ptr(ws),
                                   C.c_int64(ws.numel()), _stream_ptr()))
        return u[:, :3].contiguous() if gated else u

    @torch.no_grad()
    def integrate_pos(self, pos_init, t, base_times):
        """RK2 back-advection with per-point times (tensorf_keyframe.py:575-611). Unlike the reference this
        does not mutate its arguments."""
        L = _lib.lib()
        x = pos_init.reshape(-1, 3).contiguous().float()
        tt = t.reshape(-1).contiguous().float()
        bb = base_times.reshape(-1).contiguous().float()
        N = x.shape[0]
        desc = self._desc()
        nb = C.c_int64(0)
        _lib.check(L.nvfi_vel_workspace_bytes(C.byref(desc), C.c_int64(N), C.byref(nb)))
        ws = torch.empty(nb.value, dtype=torch.uint8, device=x.device)
        out = torch.empty_like(x)
        _lib.check(L.nvfi_integrate_pos(C.byref(desc), C.c_int64(N), _lib.ptr(x), _lib.ptr(tt), _lib.ptr(bb), _lib.ptr(out),
                                        _lib.ptr(ws), C.c_int64(ws.numel()), _stream_ptr()))
        return out

    @torch.no_grad()
    def compute_densityfeature(self, xyzt):
        """(N,4) normalised (x,y,z,t') -> (N,1) (tensorf_keyframe.py:233-272)."""
        L = _lib.lib()
        q = xyzt.reshape(-1, 4).contiguous().float()
        N = q.shape[0]
        feat = torch.empty(N, device=q.device)
        sigma = torch.empty(N, device=q.device)
        desc = self._desc()
        _lib.check(L.nvfi_density_at(C.byref(desc), C.c_int64(N), _lib.ptr(q), _lib.ptr(feat), _lib.ptr(sigma), _stream_ptr()))
        return feat.unsqueeze(-1)

    @staticmethod
    @torch.no_grad()
    def sh_render(viewdirs, features):
        """SHRender (tensorf_model_utils.py:292-296): degree-2 real SH, relu(sum + 0.5); features (N,27)."""
        L = _lib.lib()
        v = viewdirs.reshape(-1, 3).contiguous().float()
        ft = features.reshape(-1, 27).contiguous().float()
        rgb = torch.empty(v.shape[0], 3, device=v.device)
        _lib.check(L.nvfi_sh_render(C.c_int64(v.shape[0]), _lib.ptr(v), _lib.ptr(ft), _lib.ptr(rgb), _stream_ptr()))
        return rgb

    def feature2density(self, density_features, x=None):
        return F.softplus(density_features[..., 0] + self.density_shift)

    @torch.no_grad()
    def app_at(self, xyzt, viewdirs):
        """compute_appfeature + renderModule for explicit points (tensorf_keyframe.py:274-310, tensorf_base.py:88-98)."""
        L = _lib.lib()
        q = xyzt.reshape(-1, 4).contiguous().float()
        v = viewdirs.reshape(-1, 3).contiguous().float()
        N = q.shape[0]
        rgb = torch.empty(N, 3, device=q.device)
        desc = self._desc()
        nb = C.c_int64(0)
        _lib.check(L.nvfi_app_workspace_bytes(C.byref(desc), C.c_int64(N), C.byref(nb)))
        ws = torch.empty(nb.value, dtype=torch.uint8, device=q.device)
        _lib.check(L.nvfi_app_at(C.byref(desc), C.c_int64(N), _lib.ptr(q), _lib.ptr(v), _lib.ptr(rgb), _lib.ptr(ws),
                                 C.c_int64(ws.numel()), _stream_ptr()))
        return rgb

    def _render_module_call(self, pts, viewdirs, features):
        L = _lib.lib()
        if not features.is_cuda:
            raise _lib.NvfiError("NVFi HIP kernels need CUDA tensors (no CPU fallback exists)")
        x = pts.reshape(-1, 3).contiguous().float()
        v = viewdirs.reshape(-1, 3).contiguous().float()
        fe = features.reshape(-1, self.app_dim).contiguous().float()
        N = x.shape[0]
        rgb = torch.empty(N, 3, device=x.device)
        desc = self._desc()
        nb = C.c_int64(0)
        _lib.check(L.nvfi_app_workspace_bytes(C.byref(desc), C.c_int64(N), C.byref(nb)))
        ws = torch.empty(2 * nb.value, dtype=torch.uint8, device=x.device)
        _lib.check(L.nvfi_render_mlp(C.byref(desc), C.c_int64(N), _lib.ptr(x), _lib.ptr(v), _lib.ptr(fe), _lib.ptr(rgb), _lib.ptr(ws),
                                     C.c_int64(ws.numel()), _stream_ptr()))
        return rgb

    def pde_loss(self, points, t):
        """PDE regulariser on explicit collocation points (world-space (P,3), raw t (P,1)); models/nvfi.py:42-84."""
        points = points.reshape(-1, 3).contiguous().float()
        t = t.reshape(-1).contiguous().float()
        if self._use_side_streams():
            cur = torch.cuda.current_stream()
            side = self._side_stream("p")
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                loss = _PdeFn.apply(self, points, t, *self._pde_params())
            cur.wait_stream(side)
            loss.record_stream(cur)
            points.record_stream(side); t.record_stream(side)
            return loss
        return _PdeFn.apply(self, points, t, *self._pde_params())

    @torch.no_grad()
    def pde_loss_backward_(self, points, t, weight=1.0, grad_targets=None):
        """Fused value + backward of `weight * get_vel_loss`: the gradient is ACCUMULATED into the .grad of both velocity
        nets (or into `grad_targets`, 24 tensors in _pde_params() order); returns the device tensor
        [loss, n_kept, sum div^2, sum transport^2] (un-weighted loss)."""
        L = _lib.lib()
        points = points.reshape(-1, 3).contiguous().float()
        t = t.reshape(-1).contiguous().float()
        P = points.shape[0]
        desc = self._desc()
        nbytes = C.c_int64(0)
        _lib.check(L.nvfi_pde_workspace_bytes(C.byref(desc), C.c_int64(P), C.byref(nbytes)))
        ws = self._scratch("pde", nbytes.value, points.device)
        out = torch.empty(4, device=points.device)       # written by k_pde_finish
        grads = []
        if grad_targets is not None:
            grads = list(grad_targets)
        else:
            for p in self._pde_params():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                grads.append(p.grad)
        G = self._grads_struct_vel(grads)
        counters = torch.empty(_lib.NCOUNTERS, dtype=torch.int64, device=points.device)
        # no host_info: the call does not wait for the device; the kept count is out[1] / counters[4] (device side)
        if isinstance(weight, torch.Tensor):     # the weight lives in device memory (hipGraph replay: it decays every iteration)
            if not (weight.is_cuda and weight.dtype == torch.float32 and weight.numel() == 1):
                raise _lib.NvfiError("a device-side PDE weight must be a 1-element fp32 CUDA tensor")
            _lib.check(L.nvfi_pde_loss_dev(C.byref(desc), C.c_int64(P), _lib.ptr(points), _lib.ptr(t), _lib.ptr(weight), _lib.ptr(out),
                                           C.byref(G), _lib.ptr(ws), C.c_int64(ws.numel()), _lib.ptr(counters), _stream_ptr()))
        else:
            _lib.check(L.nvfi_pde_loss_ex(C.byref(desc), C.c_int64(P), _lib.ptr(points), _lib.ptr(t), C.c_float(float(weight)), _lib.ptr(out),
                                          C.byref(G), _lib.ptr(ws), C.c_int64(ws.numel()), _lib.ptr(counters), None, None, C.c_int64(0), None,
                                          _stream_ptr()))
        self.last_pde_out, self.last_pde_counters = out, counters
        self.last_pde_n_kept = None
        return out

    def _grads_struct_cached(self, ps):
        """nvfi_grads over the .grad tensors of the 31 render parameters (NULL for a parameter that is frozen or has no .grad): the ctypes
        struct is rebuilt only when a gradient pointer changed (a GradBucket / the arena keeps them fixed for the whole run)"""
        gs = [None if (p is None or p.grad is None or not p.requires_grad) else p.grad for p in ps]
        key = tuple(0 if g is None else g.data_ptr() for g in gs)
        c = _rt(self).get("_gs_cache")
        if c is None or c[0] != key:
            for g, p in zip(gs, ps):
                if g is not None and g.stride() != p.stride():
                    raise _lib.NvfiError("a parameter's .grad must share its memory layout (channels_last planes)")
            c = (key, self._grads_struct(gs))
            _rt(self)["_gs_cache"] = c
        return c[1]

    @torch.no_grad()
    def render_mse_backward_(self, t, ray_o, ray_d, target, white_bg=True, loss_scale=1.0, jitter=None, wait_before_backward=None):
        """Fused-driver form of one TRAINING render of train_nvfi.py:150-178 + its share of loss.backward() (train_nvfi.py:242):
        forward, `loss_scale * F.mse_loss(rgb_map, target)` and the backward of both, with the parameter gradients ACCUMULATED into p.grad
        (which must exist: nvfi_amd.dist.GradBucket, or zeros_like) - no autograd graph, no torch launch: nvfi_render_fwd_mse forms the loss and
        its gradient inside the composite kernel, nvfi_render_bwd_t takes it from there.  The counterpart of pde_loss_backward_ /
        regularizers_backward_ for the photometric term.  Returns (loss, rgb): a 0-dim device tensor with the UN-scaled mse and the (R,3) colours.
        wait_before_backward: a stream the backward half has to wait for (a driver that runs regularizers_backward_ - a plain read-modify-write of
        the plane gradients - on a side stream beside the forward half)."""
        L = _lib.lib()
        if not ray_o.is_cuda or not self.aabb.is_cuda:
            raise _lib.NvfiError("NVFi HIP kernels need the field and the rays on the GPU (no CPU fallback exists)")
        if self.mask_field is not None:
            raise NotImplementedError("render_mse_backward_ does not drive the mask branch")
        ray_o = ray_o.reshape(-1, 3).contiguous().float()
        ray_d = ray_d.reshape(-1, 3).contiguous().float()
        target = target.reshape(-1, 3).contiguous().float()
        R = ray_o.shape[0]
        dev = ray_o.device
        flags = _lib.NVFI_TRAIN | (_lib.NVFI_BWD_FORK if self.fork_backward else 0)
        if jitter is None:
            jitter = self._jitter(R, dev)          # the reference's CPU-generator draw (tensorf_base.py:302-306)
        if white_bg or bool(torch.rand((1,)) < 0.5):    # tensorf_keyframe.py:740
            flags |= _lib.NVFI_WHITE_BG
        t_dev = getattr(t, "dev", None)
        t = float(np.float32(float(t)))
        ps = self._render_params()
        desc = self._desc()
        S = desc.n_samples
        nbytes = C.c_int64(0)
        _lib.check(L.nvfi_render_workspace_bytes_t(C.byref(desc), C.c_int64(R), C.c_int(flags), C.c_float(t), C.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        a3, a1 = (3 * R + 63) // 64 * 64, (R + 63) // 64 * 64          # rgb | g_rgb | depth | acc | loss: one allocation, 256-byte aligned pieces
        out = torch.empty(2 * a3 + 2 * a1 + 64, device=dev)
        rgb, g_rgb, depth, acc, loss = out[:3 * R].view(R, 3), out[a3:a3 + 3 * R], out[2 * a3:2 * a3 + R], out[2 * a3 + a1:2 * a3 + a1 + R], out[2 * a3 + 2 * a1:]
        weights = torch.empty(R, S, device=dev)
        counters = torch.empty(_lib.NCOUNTERS, dtype=torch.int64, device=dev)
        st = _stream_ptr()
        _lib.check(L.nvfi_render_fwd_mse(C.byref(desc), C.c_int64(R), _lib.ptr(ray_o), _lib.ptr(ray_d), _lib.ptr(jitter), C.c_float(t), _lib.ptr(t_dev),
                                         C.c_int(flags), _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(acc), _lib.ptr(weights), _lib.ptr(ws), C.c_int64(ws.numel()),
                                         _lib.ptr(counters), _lib.ptr(target), C.c_float(float(loss_scale)), _lib.ptr(loss), _lib.ptr(g_rgb), st))
        self.last_counters = counters
        if wait_before_backward is not None:
            torch.cuda.current_stream().wait_stream(wait_before_backward)
        G = self._grads_struct_cached(ps)
        _lib.check(L.nvfi_render_bwd_t(C.byref(desc), C.c_int64(R), _lib.ptr(ray_o), _lib.ptr(ray_d), C.c_float(t), C.c_int(int(t_dev is not None)),
                                       C.c_int(flags), _lib.ptr(weights), _lib.ptr(g_rgb), None, None, None, C.byref(G), _lib.ptr(ws), C.c_int64(ws.numel()), st))
        return loss[0], rgb

    def _jitter(self, R, device):
        """Per-ray jitter drawn on the CPU generator like the reference (tensorf_base.py:302-306), staged through a small
        ring of pinned buffers so the upload is asynchronous."""
        ov = self.__dict__.get("jitter_override")
        if ov is not None:       # tests: explicit per-ray jitter instead of a draw
            return ov.to(device).reshape(-1).float().contiguous()
        ring = _rt(self).setdefault("_jit_ring", [])      # entries: [pinned buffer, event recorded after its last upload]
        idx = _rt(self).get("_jit_idx", 0)
        if len(ring) < 4:
            ring.append([torch.empty(max(R, 4096), 1).pin_memory(), None])
        slot = ring[idx % len(ring)]
        if slot[1] is not None:
            slot[1].synchronize()      # the asynchronous copy that last read this buffer must have executed before it is rewritten
        if slot[0].shape[0] < R:
            slot[0] = torch.empty(R, 1).pin_memory()
        _rt(self)["_jit_idx"] = idx + 1
        view = slot[0][:R]
        torch.rand(R, 1, out=view)
        out = view.to(device, non_blocking=True).reshape(-1)
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record()
        return out

    def _scratch(self, key, nbytes, device):
        """Reusable workspace for calls whose workspace does not have to outlive the call."""
        cache = _rt(self).setdefault("_scratch_cache", {})
        buf = cache.get(key)
        if buf is None or buf.numel() < nbytes or buf.device != device:
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            cache[key] = buf
        return buf

    # ------------------------------------------------------------------ per-iteration regularisers (next-row f-1; torch ops)
    def _reg_planes(self):
        return list(self.density_plane_space) + list(self.density_plane_time) + list(self.app_plane_space)

    def _reg_on_device(self, reg=None):
        from ..utils.tensorf_utils import is_reference_tvloss
        return self.aabb.is_cuda and not self.__dict__.get("regs_torch_ops", False) and self.density_n_comp[0] % 4 == 0 and self.app_n_comp[0] % 4 == 0 and (reg is None or is_reference_tvloss(reg))

    def density_L1(self):
        if self._reg_on_device():
            return _RegFn.apply(self, 0, *self._reg_planes())
        total = 0
        for i in range(3):
            total = total + torch.mean(torch.abs(self.density_plane_space[i])) + torch.mean(torch.abs(1 - self.density_plane_time[i]))
        return total

    def TV_loss_density(self, reg):
        if self._reg_on_device(reg):
            return _RegFn.apply(self, 1, *self._reg_planes()) * reg.TVLoss_weight
        total = 0
        for i in range(3):
            total = total + reg(self.density_plane_space[i]) * 1e-2 + ((reg(self.density_plane_time[i], t=True) * 1e-2) if self.num_keyframes > 1 else 0)
        return total

    def TV_loss_app(self, reg):
        if self._reg_on_device(reg):
            return _RegFn.apply(self, 2, *self._reg_planes()) * reg.TVLoss_weight
        total = 0
        for i in range(3):
            total = total + reg(self.app_plane_space[i]) * 1e-2
        return total

    @torch.no_grad()
    def regularizers_backward_(self, w_l1, w_tv_density=None, w_tv_app=None):
        """Fused form of `loss += w_l1*density_L1() + w_tv_density*TV_loss_density(reg) + w_tv_app*TV_loss_app(reg)`
        followed by backward: one pass per plane that ACCUMULATES the weighted gradients into p.grad and returns the
        three un-weighted loss values (device tensor [L1, TVd, TVa]).  Call it next to loss.backward()."""
        L = _lib.lib()
        ps = self._render_params()
        grads = []
        for p in ps[:9]:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
        for g, p in zip(grads, ps[:9]):
            if g.stride() != p.stride():
                raise _lib.NvfiError("plane .grad must share the channels_last layout of its parameter")
        G = self._grads_struct(grads + [None] * 22)
        out = torch.empty(3, device=ps[0].device)
        desc = self._desc()
        if isinstance(w_l1, torch.Tensor):       # the three weights as one fp32[3] CUDA tensor (hipGraph replay: they decay every iteration)
            if not (w_l1.is_cuda and w_l1.dtype == torch.float32 and w_l1.numel() == 3 and w_l1.is_contiguous()):
                raise _lib.NvfiError("device-side regulariser weights must be a contiguous fp32[3] CUDA tensor")
            _lib.check(L.nvfi_plane_regs_dev(C.byref(desc), _lib.ptr(w_l1), _lib.ptr(out), C.byref(G), _stream_ptr()))
            return out
        _lib.check(L.nvfi_plane_regs(C.byref(desc), C.c_float(w_l1), C.c_float(w_tv_density), C.c_float(w_tv_app), _lib.ptr(out),
                                     C.byref(G), _stream_ptr()))
        return out

    # ------------------------------------------------------------------ optimiser / checkpoint surface
    def get_optparam_groups(self, lr_init_spatialxyz=0.02, lr_init_network=0.001):
        """tensorf_keyframe.py:539-550"""
        gv = [{"params": self.density_plane_space, "lr": lr_init_spatialxyz},
              {"params": self.density_plane_time, "lr": lr_init_spatialxyz},
              {"params": self.app_plane_space, "lr": lr_init_spatialxyz},
              {"params": self.app_plane_time, "lr": lr_init_spatialxyz},
              {"params": self.basis_mat.parameters(), "lr": lr_init_network},
              {"params": self.basis_mat_density.parameters(), "lr": lr_init_network}]
        if isinstance(self.renderModule, nn.Module):
            gv += [{"params": self.renderModule.parameters(), "lr": lr_init_network}]
        if self.use_vel:
            gv += [{"params": self.vel.parameters(), "lr": lr_init_network}]
        return gv

    def get_kwargs(self):
        """tensorf_base.py:247-268"""
        kw = {"aabb": self.aabb, "gridSize": self.gridSize.tolist(), "density_n_comp": self.density_n_comp,
              "appearance_n_comp": self.app_n_comp, "app_dim": self.app_dim, "density_shift": self.density_shift,
              "alphaMask_thres": self.alphaMask_thres, "fea2denseAct": self.fea2denseAct, "near_far": self.near_far,
              "step_ratio": self.step_ratio, "shadingMode": self.shadingMode, "pos_pe": self.pos_pe, "view_pe": self.view_pe,
              "fea_pe": self.fea_pe, "featureC": self.featureC, "num_keyframes": self.num_keyframes}
        if self.alphaMask is not None:
            kw["alphaMask_grid"] = self.alphaMask.gridSize
        return kw

    # ------------------------------------------------------------------ occupancy grid maintenance (next-row f-3)
    @torch.no_grad()
    def compute_alpha(self, xyzt_locs, length=0.01, times=None, time_offset=None, transfer=False):
        """tensorf_keyframe.py:508-537 with per-point times (xyzt_locs[..., 3] raw t, [..., :3] world xyz)."""
        pts = self.normalize_coord(xyzt_locs[..., :3].reshape(-1, 3))
        t = xyzt_locs[..., -1:].reshape(-1, 1).float()
        ts = self.tmax / (self.num_keyframes - 1) if self.num_keyframes > 1 else 1
        base = torch.zeros_like(t) if transfer else torch.round((t / ts).clamp(0.0, self.num_keyframes - 1)) * ts
        prev = self.integrate_pos(pts, t, base) if self.use_vel else pts
        feat = self.compute_densityfeature(torch.cat([prev, self.normalize_time_coord(base)], dim=-1))
        sigma = self.feature2density(feat, {})
        return (1 - torch.exp(-sigma * length)).view(xyzt_locs.shape[:-1])

    @torch.no_grad()
    def getDenseAlpha(self, gridSize=None, transfer=False):
        """tensorf_keyframe.py:461-499: running maximum of alpha over the 60 frame times i/60 on a dense grid."""
        L = _lib.lib()
        dev = self.aabb.device
        samples = torch.stack(torch.meshgrid(torch.linspace(0, 1, gridSize[0]), torch.linspace(0, 1, gridSize[1]),
                                             torch.linspace(0, 1, gridSize[2]), indexing="ij"), -1).to(dev)
        dense_xyz = self.aabb[0] * (1 - samples) + self.aabb[1] * samples
        flat = dense_xyz.reshape(-1, 3).contiguous()
        N = flat.shape[0]
        alpha = torch.zeros(N, device=dev)
        desc = self._desc()
        nb = C.c_int64(0)
        _lib.check(L.nvfi_alpha_workspace_bytes(C.byref(desc), C.c_int64(N), C.byref(nb)))
        ws = self._scratch("alpha", nb.value, dev)
        for t in (np.linspace(0, 59, 60) / 60):
            _lib.check(L.nvfi_compute_alpha(C.byref(desc), C.c_int64(N), _lib.ptr(flat), C.c_float(float(np.float32(t))), C.c_int(int(transfer)),
                                            C.c_float(self._step_host), C.c_int(1), _lib.ptr(alpha), _lib.ptr(ws), C.c_int64(ws.numel()), _stream_ptr()))
        return alpha.view(gridSize[0], gridSize[1], gridSize[2]), dense_xyz

    @torch.no_grad()
    def updateAlphaMask(self, gridSize=(200, 200, 200), transfer=False):
        """tensorf_keyframe.py:379-405: dilate (3^3 max-pool), threshold, build the AlphaGridMask, return the tight aabb."""
        gridSize = [int(g) for g in gridSize]
        alpha, dense_xyz = self.getDenseAlpha(gridSize, transfer=transfer)
        dense_xyz = dense_xyz.transpose(0, 2).contiguous()
        alpha = alpha.clamp(0, 1).transpose(0, 2).contiguous()[None, None]
        alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1).view(gridSize[::-1])
        alpha[alpha >= self.alphaMask_thres] = 1
        alpha[alpha < self.alphaMask_thres] = 0
        self.alphaMask = AlphaGridMask(self.aabb.device, self.aabb, alpha)
        valid_xyz = dense_xyz[alpha > 0.5]
        return torch.stack((valid_xyz.amin(0), valid_xyz.amax(0)))

    @torch.no_grad()
    def shrink(self, new_aabb):
        """tensorf_keyframe.py:407-458: crop every plane to the voxel range of new_aabb."""
        xyz_min, xyz_max = new_aabb
        t_l, b_r = (xyz_min - self.aabb[0]) / self.units, (xyz_max - self.aabb[0]) / self.units
        t_l, b_r = torch.round(torch.round(t_l)).long(), torch.round(b_r).long() + 1
        b_r = torch.stack([b_r, self.gridSize]).amin(0)
        tl, br = t_l.tolist(), b_r.tolist()

        def crop(space, time):
            ns, nt = [], []
            for i in range(3):
                a, b = self.matModeSpace[i]
                c = self.matModeTime[i][0]
                ns.append(nn.Parameter(_cl(space[i].data[..., tl[b]:br[b], tl[a]:br[a]].clone())))
                nt.append(nn.Parameter(_cl(time[i].data[..., :, tl[c]:br[c]].clone())))
            return nn.ParameterList(ns), nn.ParameterList(nt)

        self.density_plane_space, self.density_plane_time = crop(self.density_plane_space, self.density_plane_time)
        self.app_plane_space, self.app_plane_time = crop(self.app_plane_space, self.app_plane_time)
        if self.alphaMask is not None and not torch.all(self.alphaMask.gridSize == self.gridSize):
            t_l_r, b_r_r = t_l / (self.gridSize - 1), (b_r - 1) / (self.gridSize - 1)
            correct = torch.zeros_like(new_aabb)
            correct[0] = (1 - t_l_r) * self.aabb[0] + t_l_r * self.aabb[1]
            correct[1] = (1 - b_r_r) * self.aabb[0] + b_r_r * self.aabb[1]
            new_aabb = correct
        newSize = b_r - t_l
        self.aabb.copy_(new_aabb)       # the surround-box gate keeps the INITIAL aabb (velocity_field.py:44 quirk)
        self.update_stepSize([int(v) for v in newSize.tolist()])

    # ------------------------------------------------------------------ grid maintenance (next-row f-3; torch ops)
    @torch.no_grad()
    def upsample_volume_grid(self, res_target, new_keyframes):
        """tensorf_keyframe.py:327-376: bilinear align_corners=True resampling of every plane."""
        self.num_keyframes = new_keyframes
        self.time_scale_factor = self.tmax / (self.num_keyframes - 1) if self.num_keyframes > 1 else 1

        def up(space, time):
            ns, nt = [], []
            for i in range(3):
                a, b = self.matModeSpace[i]
                c = self.matModeTime[i][0]
                ns.append(nn.Parameter(_cl(F.interpolate(space[i].data, size=(int(res_target[b]), int(res_target[a])), mode="bilinear", align_corners=True))))
                nt.append(nn.Parameter(_cl(F.interpolate(time[i].data, size=(self.num_keyframes, int(res_target[c])), mode="bilinear", align_corners=True))))
            return nn.ParameterList(ns), nn.ParameterList(nt)

        self.app_plane_space, self.app_plane_time = up(self.app_plane_space, self.app_plane_time)
        self.density_plane_space, self.density_plane_time = up(self.density_plane_space, self.density_plane_time)
        self.update_stepSize(res_target)
