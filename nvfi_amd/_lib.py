"""ctypes binding of libnvfi_hip.so (include/nvfi_hip.h).  The product path has NO CPU fallback:
a missing library or a non-GPU tensor raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("NVFI_LIB", os.path.join(HERE, "csrc", "libnvfi_hip.so"))   # NVFI_LIB: alternative build (experiments)
fp = C.c_void_p

NVFI_TRAIN, NVFI_WHITE_BG, NVFI_TRANSFER, NVFI_WANT_MASK, NVFI_BWD_FORK = 1, 2, 4, 8, 16
NCOUNTERS = 8


class FieldDesc(C.Structure):
    _fields_ = [
        ("G", C.c_int32 * 3), ("K", C.c_int32), ("Cd", C.c_int32), ("Ca", C.c_int32), ("app_dim", C.c_int32),
        ("n_samples", C.c_int32), ("use_vel", C.c_int32), ("gate_sur", C.c_int32), ("has_amask", C.c_int32), ("shading", C.c_int32), ("vel_fp16", C.c_int32),
        ("am_dims", C.c_int32 * 3),
        ("aabb", C.c_float * 6), ("near_", C.c_float), ("far_", C.c_float), ("step_size", C.c_float),
        ("density_shift", C.c_float), ("distance_scale", C.c_float), ("weight_thres", C.c_float),
        ("alpha_thres", C.c_float), ("tmax", C.c_float),
        ("gate_lo", C.c_float * 3), ("gate_hi", C.c_float * 3),
        ("dps", fp * 3), ("dpt", fp * 3), ("aps", fp * 3), ("apt", fp * 3),
        ("basis", fp), ("rW", fp * 3), ("rb", fp * 3),
        ("vW", fp * 6), ("vb", fp * 6), ("aW", fp * 6), ("ab", fp * 6),
        ("amask", fp),
        ("frags", fp),
    ]


class MaskDesc(C.Structure):
    _fields_ = [("n_layer", C.c_int32), ("n_dim", C.c_int32), ("mask_dim", C.c_int32), ("W", fp * 5), ("b", fp * 5)]


class AdamTensor(C.Structure):
    _fields_ = [("p", fp), ("g", fp), ("m", fp), ("v", fp), ("n", C.c_int64), ("lr", C.c_float)]


class MaskGrads(C.Structure):
    _fields_ = [("W", fp * 5), ("b", fp * 5)]


class Grads(C.Structure):
    _fields_ = [
        ("dps", fp * 3), ("dpt", fp * 3), ("aps", fp * 3), ("apt", fp * 3),
        ("basis", fp), ("rW", fp * 3), ("rb", fp * 3),
        ("vW", fp * 6), ("vb", fp * 6), ("aW", fp * 6), ("ab", fp * 6),
    ]


class DrawDesc(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("iteration", C.c_uint64), ("iteration_dev", fp),
                ("n_batches", C.c_int32), ("R", C.c_int64), ("n_pixels", C.c_int64),
                ("bundle_o", fp), ("bundle_d", fp), ("target_img", fp),
                ("rays_o", fp * 2), ("rays_d", fp * 2), ("target", fp * 2), ("pixel_ids", fp * 2),
                ("P", C.c_int64), ("aabb", C.c_float * 6), ("points", fp), ("t", fp)]


EXPORTS = [
    "nvfi_last_error", "nvfi_abi_version",
    "nvfi_render_workspace_bytes", "nvfi_render_workspace_bytes_t", "nvfi_render_fwd", "nvfi_render_bwd", "nvfi_render_fwd_t", "nvfi_render_bwd_t",
    "nvfi_pde_loss_dev", "nvfi_adam_step_dev", "nvfi_frag_cache_bytes", "nvfi_pack_frags", "nvfi_stream_capture_id", "nvfi_render_fwd_mse", "nvfi_draw_batch",
    "nvfi_pde_workspace_bytes", "nvfi_pde_loss", "nvfi_pde_loss_ex", "nvfi_pde_loss_split", "nvfi_plane_regs", "nvfi_plane_regs_dev", "nvfi_adam_step", "nvfi_mse", "nvfi_render_mask", "nvfi_render_export_masked", "nvfi_maskfield_workspace_bytes", "nvfi_maskfield_fwd", "nvfi_maskfield_bwd", "nvfi_sh_render", "nvfi_compute_alpha", "nvfi_gen_rays",
    "nvfi_vel_eval", "nvfi_vel_workspace_bytes", "nvfi_integrate_pos", "nvfi_density_at", "nvfi_app_at", "nvfi_render_mlp", "nvfi_app_workspace_bytes", "nvfi_alpha_workspace_bytes",
    "nvfi_comm_unique_id", "nvfi_comm_init", "nvfi_allreduce_grads", "nvfi_comm_destroy", "nvfi_selftest", "nvfi_debug_act", "nvfi_prof_enable", "nvfi_prof_collect", "nvfi_prof_nclasses",
]

_LIB = None


class NvfiError(RuntimeError):
    pass


def lib():
    """Load the HIP library; fail loudly if it has not been built (no fallback path exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO):
            raise NvfiError(f"{SO} is missing: build it with `python -m nvfi_amd.build` (needs hipcc). "
                            "There is no CPU fallback for the NVFi hot path.")
        L = C.CDLL(SO)
        L.nvfi_last_error.restype = C.c_char_p
        for name in EXPORTS:
            getattr(L, name)  # raises AttributeError on a missing symbol
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise NvfiError(f"libnvfi_hip error {rc}: {lib().nvfi_last_error().decode()}")


def capture_id(stream_handle):
    """id (> 0) of the hipGraph capture the stream takes part in, 0 when it is not capturing"""
    cid = C.c_uint64(0)
    check(lib().nvfi_stream_capture_id(C.c_void_p(stream_handle), C.byref(cid)))
    return cid.value


def ptr(t):
    """Device pointer of a torch tensor (or NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NvfiError("NVFi HIP kernels need tensors on the GPU (no CPU fallback exists)")
    return C.c_void_p(t.data_ptr())
