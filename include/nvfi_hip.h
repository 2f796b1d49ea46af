/*
 * nvfi_hip.h - C ABI of the MI355X-native NVFi hot path (libnvfi_hip.so, hand-written HIP for gfx950).
 *
 * The reference has no FFI layer: its hot path is Python calling ATen ops.  This ABI is what the
 * host-side mirror (nvfi_amd/models) binds through ctypes; each entry point names the reference
 * interface it replaces.  Plain pointers and sizes only: every pointer below is a DEVICE pointer
 * unless stated, `stream` is a hipStream_t passed as void*, the caller owns every buffer, nothing
 * is allocated inside (workspace sizes are queried first).  All calls are asynchronous on
 * `stream` - none of them waits for the device - with the documented exceptions that hand a value back to the host or are diagnostics:
 * nvfi_pde_loss_ex / nvfi_pde_loss_split when `host_info` is non-NULL, nvfi_prof_collect, nvfi_selftest, and nvfi_prof_enable(1) (it drops a
 * marker launch on the null stream and waits for it).
 * Return value: 0 = ok, otherwise an error code; nvfi_last_error() gives the text.
 *
 * Layouts: factor planes are CHANNEL-LAST, [H][W][C] fp32 (the physical layout of a torch
 * (1,C,H,W) tensor in torch.channels_last); Linear weights are (out,in) row-major as in torch.
 */
#ifndef NVFI_HIP_H
#define NVFI_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NVFI_ABI_VERSION 5

/* Field description: TensorVMKeyframeTimeKplane state that reaches the hot path
 * (reference models/tensorf_keyframe.py:37-134, models/tensorf_base.py:133-227). */
typedef struct nvfi_field_desc {
    int32_t G[3];            /* gridSize x,y,z */
    int32_t K;               /* num_keyframes */
    int32_t Cd, Ca, app_dim; /* 24, 48, 32 in every shipped config */
    int32_t n_samples;       /* nSamples (tensorf_base.py:223) */
    int32_t use_vel;         /* cfg.use_vel */
    int32_t gate_sur;        /* 0 VelocityAABB(eps), 1 VelocityAABBSur (velocity_field.py:21-51) */
    int32_t has_amask;       /* alphaMask present (used in eval only, tensorf_keyframe.py:656) */
    int32_t shading;         /* shadingMode (tensorf_base.py:184-197): 0 MLP_PE (app_dim 32 + MLPRender_PE), 1 SH (app_dim 27, SHRender: no MLP) */
    int32_t vel_fp16;        /* ABI v4, opt-in (0 = off, the default and the parity mode; 2 = as 1 but with fp32 products emulated by TWO binary16
                              * terms per operand, three MFMAs, ~2^-21 relative per product): 1 = every NO-GRAD back-advection (nvfi_integrate_pos,
                              * the warp of eval-mode renders, nvfi_compute_alpha) evaluates VelBasis with fp16-input MFMAs (weights and layer
                              * inputs rounded to binary16, fp32 accumulation) - the counterpart of the reference's autocast switch
                              * --disable_fp32 (train_nvfi.py:96,144) for inference; training renders, the PDE term and all gradients stay fp32.
                              * Bit 2 (+4, round 4, opt-in): the velocity warp of TRAINING renders runs its FORWARD on the one-term fp16-input MFMA too
                              * (pre-activations stashed in fp32; the adjoint and the weight gradients stay fp32 MFMA on those stashes - the
                              * arithmetic of a forward under autocast with an fp32 backward; the PDE term and the render MLP stay fp32).
                              * 3 = x6 (round 5): fp32 products of the hidden layers formed EXACTLY from three bfloat16 terms per operand on the 16-bit
                              * matrix pipe - same results as the fp32 MFMA kernels to rounding order, and since round 6 what 0 selects for every
                              * no-grad back-advection as well (eval renders and the PDE prefilter since round 5; nvfi_integrate_pos and
                              * nvfi_compute_alpha now).  Bit 3 (+8): keep the fp32 MFMA kernels for those two calls (A/B reference). */
    int32_t am_dims[3];      /* alpha volume W,H,D */
    float aabb[6];           /* min xyz, max xyz */
    float near_, far_, step_size;
    float density_shift, distance_scale, weight_thres, alpha_thres, tmax;
    float gate_lo[3], gate_hi[3]; /* velocity gate box in normalised coordinates */
    const float* dps[3];     /* density_plane_space[i]  [G_b][G_a][Cd] */
    const float* dpt[3];     /* density_plane_time[i]   [K][G_c][Cd]   */
    const float* aps[3];     /* app_plane_space[i]      [G_b][G_a][Ca] */
    const float* apt[3];     /* app_plane_time[i]       [K][G_c][Ca]   */
    const float* basis;      /* basis_mat.weight (app_dim, Ca) */
    const float* rW[3];      /* renderModule.mlp.{0,2,4}.weight */
    const float* rb[3];
    const float* vW[6];      /* vel_net.weight_net Linear weights (6 layers) */
    const float* vb[6];
    const float* aW[6];      /* vel_net.a_weight_net */
    const float* ab[6];
    const float* amask;      /* alpha volume (D,H,W) or NULL */
    const float* frags;      /* ABI v5: fragment cache written by nvfi_pack_frags for THESE weights, or NULL (the render / PDE calls then repack
                              * the weights into their own workspace, as in v1-v4) */
} nvfi_field_desc;

/* Gradient buffers, same shapes/layouts as the parameters; kernels ACCUMULATE (+=) into them.
 * A NULL pointer skips that gradient. */
typedef struct nvfi_grads {
    float* dps[3]; float* dpt[3]; float* aps[3]; float* apt[3];
    float* basis;
    float* rW[3]; float* rb[3];
    float* vW[6]; float* vb[6];
    float* aW[6]; float* ab[6];
} nvfi_grads;

#define NVFI_TRAIN     1  /* training mode: jitter used, alpha mask ignored, intermediates kept for backward */
#define NVFI_WHITE_BG  2  /* rgb += 1-acc  (white_bg or the random-white coin, tensorf_keyframe.py:740) */
#define NVFI_TRANSFER  4  /* transfer_vel: base time 0 (models/nvfi.py:30) */
#define NVFI_WANT_MASK 8  /* plan workspace room for nvfi_render_mask (mask_field attached) */
#define NVFI_BWD_FORK 16  /* nvfi_render_bwd at a keyframe time: the density half of the backward may run on a library-owned side stream beside the
                           * appearance half (joined before the call returns); for callers that drive a single stream */

/* counters written by nvfi_render_fwd (device int64[8]):
 * 0 valid samples V, 1 warped samples N, 2 appearance-masked samples M, 3 velocity-net evaluations, 7 device-time plan mismatch (nvfi_render_fwd_t) */
#define NVFI_NCOUNTERS 8

const char* nvfi_last_error(void);
int nvfi_abi_version(void);

/* ---- render: replaces NVFi.render_ray / TensorVMKeyframeTimeKplane.forward + render_pts
 *      (models/nvfi.py:27-31, models/tensorf_keyframe.py:613-755) for one chunk of R rays. */
int nvfi_render_workspace_bytes(const nvfi_field_desc* f, int64_t R, int flags, int64_t* bytes);
/* exact size for a given time t (extrapolated times need more RK2 steps, hence more stash) */
int nvfi_render_workspace_bytes_t(const nvfi_field_desc* f, int64_t R, int flags, float t, int64_t* bytes);
int nvfi_render_fwd(const nvfi_field_desc* f, int64_t R,
                    const float* rays_o, const float* rays_d, /* (R,3) */
                    const float* jitter,                     /* (R) u_r in [0,1) or NULL (eval) */
                    float t, int flags,
                    float* rgb, float* depth, float* acc,    /* (R,3) (R) (R) */
                    float* weights,                          /* (R,S) */
                    void* workspace, int64_t workspace_bytes,
                    int64_t* counters,                       /* device int64[NVFI_NCOUNTERS] or NULL */
                    void* stream);
/* backward of the call that filled `workspace` (autograd of models/tensorf_keyframe.py:613-755,
 * reference: loss.backward() at train_nvfi.py:242).  Upstream grads may be NULL (= zero). */
int nvfi_render_bwd(const nvfi_field_desc* f, int64_t R,
                    const float* rays_o, const float* rays_d, float t, int flags,
                    const float* weights,                    /* the (R,S) output of the forward */
                    const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_weights,
                    const nvfi_grads* grads,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* ---- the same two calls with the frame time in DEVICE memory (ABI v3), so that a training iteration can be captured once as a hipGraph and
 *      replayed with a new time every iteration (train_nvfi.py:150-158 draws a new frame per iteration): `t` still fixes the launch plan
 *      on the host - how many RK2 steps the warp takes, hence which kernels run and how the workspace is laid out - while the kernels
 *      take the time itself from *t_dev (keyframe row, step sizes and step times are derived on the device, same arithmetic).  *t_dev
 *      must lie in the same plan class as `t` (same number of RK2 steps back to its keyframe: true for all non-keyframe frame times of
 *      the shipped configs, and for all keyframe times); if it does not, the call renders `t` and sets counters[7] = 1.
 *      t_dev == NULL / t_on_device == 0: exactly nvfi_render_fwd / nvfi_render_bwd. */
int nvfi_render_fwd_t(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d, const float* jitter,
                      float t, const float* t_dev, int flags, float* rgb, float* depth, float* acc, float* weights,
                      void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream);
int nvfi_render_bwd_t(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d, float t, int t_on_device, int flags,
                      const float* weights, const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_weights,
                      const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, void* stream);

/* nvfi_render_fwd_t with the photometric loss of the training loop attached (ABI v5; train_nvfi.py:159,178: F.mse_loss(rgb_map, target)):
 * loss[0] = mean((rgb - target)^2) and g_rgb[i] = loss_scale * 2 (rgb[i] - target[i]) / (3 R) - the upstream gradient nvfi_render_bwd[_t]
 * takes - come out of the composite kernel of the same call (no nvfi_mse launch, no autograd node). */
int nvfi_render_fwd_mse(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d, const float* jitter,
                        float t, const float* t_dev, int flags, float* rgb, float* depth, float* acc, float* weights,
                        void* workspace, int64_t workspace_bytes, int64_t* counters,
                        const float* target /* (R,3) */, float loss_scale, float* loss /* device float[1] */, float* g_rgb /* (R,3) */, void* stream);

/* ---- fragment cache (ABI v5).  The MFMA kernels read the nn.Linear weights of the field - renderModule.mlp + basis_mat
 *      (models/tensorf_base.py:67-98, tensorf_keyframe.py:310), vel_net.weight_net / a_weight_net (models/velocity_field.py:60-67) - in
 *      fragment order.  nvfi_pack_frags writes every fragment set the render and PDE calls use into `cache` (nvfi_frag_cache_bytes bytes,
 *      caller-owned) in ONE launch; a descriptor whose `frags` points at it makes nvfi_render_fwd[_t] / nvfi_render_bwd[_t] /
 *      nvfi_pde_loss* read it instead of repacking the weights per call (7 launches per training iteration otherwise).  The caller repacks
 *      after every change of the weights (once per optimiser step) and orders the readers behind the repack. */
int nvfi_frag_cache_bytes(const nvfi_field_desc* f, int64_t* bytes);
int nvfi_pack_frags(const nvfi_field_desc* f, void* cache, int64_t cache_bytes, void* stream);
/* id of the hipGraph capture `stream` is part of (0: not capturing; forked streams of one capture share the id).  A caller that bakes the
 * cache pointer into a captured call must have a nvfi_pack_frags node in the SAME capture (or replay the graph that holds it first): the
 * Python mirror uses this id to refuse a cache that was packed outside the capture, whatever its key says. */
int nvfi_stream_capture_id(void* stream, uint64_t* id);

/* ---- PDE regulariser: replaces NVFi.get_vel_loss (models/nvfi.py:42-84) with explicit collocation
 *      points (world space (P,3)) and raw times (P).  out (device float[4]): loss, n_kept, sum div^2,
 *      sum transport^2.  grads: vW,vb,aW,ab are accumulated scaled by `loss_scale` when non-NULL. */
/* Workspace of the PDE call, caller-allocated and physically backed whatever the kept count turns out to be: 1.6 GiB at P = 32 768,
 * 5.5 GiB at 131 072, 10.7 GiB from P = 262 144 up (the candidates are processed in chunks of 262 144; the stash of a chunk is sized for
 * every candidate kept). */
int nvfi_pde_workspace_bytes(const nvfi_field_desc* f, int64_t P, int64_t* bytes);
int nvfi_pde_loss(const nvfi_field_desc* f, int64_t P, const float* points, const float* t,
                  float loss_scale, float* out, const nvfi_grads* grads,
                  void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream);

/* same, with optional diagnostics: kept (device uint8[P]), jac (device float[n_jac][6][4], rows 3..5 zero: the loss
 * does not use the acceleration Jacobian) for the first n_jac kept points; host_info (HOST int64[2], optional): the kept count and
 * the number of prefilter net evaluations - asking for it makes the call synchronise `stream` once (the reference's
 * `if xyzt.shape[0] == 0`, nvfi.py:66, is the same wait); with host_info == NULL the call is asynchronous: the kept count never
 * leaves the device (out[1], counters[4]) and the Jacobian / adjoint passes size themselves from it. */
int nvfi_pde_loss_ex(const nvfi_field_desc* f, int64_t P, const float* points, const float* t,
                     float loss_scale, float* out, const nvfi_grads* grads,
                     void* workspace, int64_t workspace_bytes, int64_t* counters,
                     uint8_t* kept_out, float* jac_out, int64_t n_jac, int64_t* host_info, void* stream);
/* As nvfi_pde_loss_ex with the call split over two streams: out[] (the value) is complete on `stream` after the Jacobian forward; the adjoint
 * pass and the weight gradients run on `bwd_stream`, ordered behind the forward by an event.  For the reference's loop, which waits for the
 * value right after the call (train_nvfi.py:233) and then differentiates the renders: the PDE adjoint overlaps with them.  The caller
 * orders its use of `grads` and the release of `workspace` behind `bwd_stream`.  P above one chunk (262 144): one stream. */
int nvfi_pde_loss_split(const nvfi_field_desc* f, int64_t P, const float* points, const float* t, float loss_scale,
                        float* out, const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, int64_t* counters,
                        uint8_t* kept_out, float* jac_out, int64_t n_jac, int64_t* host_info, void* stream, void* bwd_stream);

/* nvfi_pde_loss with the loss scale (train_nvfi.py:229-234: vel_reg_weight, decayed every iteration) read from device memory */
int nvfi_pde_loss_dev(const nvfi_field_desc* f, int64_t P, const float* points, const float* t, const float* loss_scale_dev,
                      float* out, const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream);

/* ---- mask branch of render_pts (models/tensorf_keyframe.py:673-676, 749-753) with the MaskField as train_segm.py:97-102
 *      builds it: 3 -> n_dim x n_layer (ReLU) -> mask_dim, softmax.  Call after nvfi_render_fwd with the same workspace:
 *      mask_map[R][mask_dim] = sum over appearance-masked samples of weight * softmax(MaskField(warped xyz)). Inference only. */
typedef struct nvfi_mask_desc {
    int32_t n_layer;         /* hidden layers (4) */
    int32_t n_dim;           /* hidden width (128) */
    int32_t mask_dim;        /* outputs (<= 32) */
    const float* W[5];       /* point_fc.{0..3}.weight (n_dim,3|n_dim), mask_fc.weight (mask_dim,n_dim) */
    const float* b[5];
} nvfi_mask_desc;
int nvfi_render_mask(const nvfi_field_desc* f, const nvfi_mask_desc* m, int64_t R, float t, int flags, const float* weights,
                     float* mask_map, void* workspace, int64_t workspace_bytes, void* stream);
/* appearance-masked samples of the nvfi_render_fwd call that filled `workspace` (same f, R, t, flags): warped keyframe positions
 * xyz_out (cap,3) and dense sample indices idx_out (cap) = ray * n_samples + sample; entries beyond counters[2] are left untouched.
 * With it the host builds the train-mode (differentiable) mask branch of render_pts (models/tensorf_keyframe.py:749-753) from
 * nvfi_maskfield_fwd / nvfi_maskfield_bwd: mask_map = sum_j weight_j * MaskField(xyz_j). */
int nvfi_render_export_masked(const nvfi_field_desc* f, int64_t R, float t, int flags, void* workspace, int64_t workspace_bytes,
                              int64_t cap, float* xyz_out, int64_t* idx_out, void* stream);
/* ---- MaskField on free points, forward and backward: the model train_segm.py:126-227 optimises (models/mask_field.py:68-83;
 *      xyz (N,3) -> softmax mask (N,mask_dim)).  mode & NVFI_MASK_TRAIN keeps the activations in `workspace` for nvfi_maskfield_bwd, which
 *      ACCUMULATES d loss / d W_l, b_l (l = point_fc.0..3, mask_fc) from g_mask = d loss / d mask (N,mask_dim); the points carry
 *      no gradient (the reference computes them under no_grad, train_segm.py:137-170). */
typedef struct nvfi_mask_grads { float* W[5]; float* b[5]; } nvfi_mask_grads;
int nvfi_maskfield_workspace_bytes(const nvfi_mask_desc* m, int64_t N, int train, int64_t* bytes);
#define NVFI_MASK_TRAIN 1   /* keep the activations in `workspace` for nvfi_maskfield_bwd */
#define NVFI_MASK_FP16  2   /* layer products on the fp16-input MFMA (weights and layer inputs rounded to fp16, fp32 accumulation,
                               fp32 bias / ReLU / softmax / stashes / weight gradients); default: exact fp32 MFMA */
int nvfi_maskfield_fwd(const nvfi_mask_desc* m, int64_t N, const float* xyz, float* mask_out, int mode,
                       void* workspace, int64_t workspace_bytes, void* stream);
int nvfi_maskfield_bwd(const nvfi_mask_desc* m, int64_t N, const float* g_mask, const nvfi_mask_grads* grads, int mode,
                       void* workspace, int64_t workspace_bytes, void* stream);
/* SHRender (models/tensorf_model_utils.py:292-296 with models/sh.py:87-110, degree 2): view (N,3), feat (N,27) -> rgb (N,3) */
int nvfi_sh_render(int64_t N, const float* view, const float* feat27, float* rgb, void* stream);

/* ---- per-iteration plane regularisers (next-row f-1): density_L1, TV_loss_density, TV_loss_app
 *      (models/tensorf_keyframe.py:188-231, utils/tensorf_utils.py:139-158) in one pass.  out3 (device float[3]) receives the
 *      UN-weighted values (L1, TV density, TV app) exactly as the reference functions return them; when grads is non-NULL the
 *      gradient of  w_l1*L1 + w_tv_density*TVd + w_tv_app*TVa  is accumulated into grads->dps/dpt/aps. */
int nvfi_plane_regs(const nvfi_field_desc* f, float w_l1, float w_tv_density, float w_tv_app, float* out3,
                    const nvfi_grads* grads, void* stream);
/* same with the three weights read from DEVICE memory (float[3]): the autograd backward of `weight * field.density_L1()` receives its
 * upstream gradient as a device scalar */
int nvfi_plane_regs_dev(const nvfi_field_desc* f, const float* w3_dev, float* out3, const nvfi_grads* grads, void* stream);

/* ---- outer optimiser step: torch.optim.Adam(betas, eps) without amsgrad / weight decay (train_nvfi.py:88-96, 243) over n_tensors
 *      parameter tensors in ONE launch.  `t` is a HOST array; p/g/m/v are device pointers (parameter, gradient, exp_avg, exp_avg_sq),
 *      n elements each, lr the tensor's learning rate; `step` counts from 1 (bias corrections); zero_grad != 0 clears g in the same pass. */
typedef struct nvfi_adam_tensor { float* p; float* g; float* m; float* v; int64_t n; float lr; } nvfi_adam_tensor;
/* F.mse_loss(x, target) of a training batch (train_nvfi.py:159,178) with its gradient in one launch: loss[0] = mean((x - target)^2),
 * grad[i] = 2 (x[i] - target[i]) / n.  One workgroup: n <= 2^22. */
int nvfi_mse(const float* x, const float* target, int64_t n, float* loss, float* grad, void* stream);
int nvfi_adam_step(const nvfi_adam_tensor* t, int n_tensors, float beta1, float beta2, float eps, int64_t step, int zero_grad, void* stream);

/* same step with the per-iteration scalars in DEVICE memory (hipGraph replay): hyper_dev[0] = 1/sqrt(1-beta2^step),
 * hyper_dev[1+k] = lr_k / (1-beta1^step) for tensor k of the table (the values nvfi_adam_step derives on the host; no tensor may be empty) */
int nvfi_adam_step_dev(const nvfi_adam_tensor* t, int n_tensors, float beta1, float beta2, float eps, const float* hyper_dev, int zero_grad, void* stream);

/* ---- building blocks used by train_segm-style callers and by the parity tests */
/* VelBasis.forward (velocity_field.py:69-75): xt (N,4) -> u (N,6)=(v,a); gated!=0: VelocityAABB[Sur].forward -> (N,3) in u (stride 6) */
int nvfi_vel_eval(const nvfi_field_desc* f, int64_t N, const float* xt, float* u6, int gated,
                  void* workspace, int64_t workspace_bytes, void* stream);
int nvfi_vel_workspace_bytes(const nvfi_field_desc* f, int64_t N, int64_t* bytes);
/* integrate_pos (tensorf_keyframe.py:575-611): x (N,3) normalised, t (N), base (N) -> xk (N,3) */
int nvfi_integrate_pos(const nvfi_field_desc* f, int64_t N, const float* x, const float* t, const float* base,
                       float* xk, void* workspace, int64_t workspace_bytes, void* stream);
/* compute_densityfeature + feature2density (tensorf_keyframe.py:233-272, 312-321): xyzt (N,4) -> feat (N), sigma (N) */
int nvfi_density_at(const nvfi_field_desc* f, int64_t N, const float* xyzt, float* feat, float* sigma, void* stream);
/* compute_appfeature + MLPRender_PE (tensorf_keyframe.py:274-310, tensorf_base.py:88-98): xyzt (N,4), view (N,3) -> rgb (N,3) */
int nvfi_app_workspace_bytes(const nvfi_field_desc* f, int64_t N, int64_t* bytes);
int nvfi_app_at(const nvfi_field_desc* f, int64_t N, const float* xyzt, const float* view, float* rgb,
                void* workspace, int64_t workspace_bytes, void* stream);
/* renderModule(pts, viewdirs, features) as a stand-alone call - MLPRender_PE.forward (tensorf_base.py:88-98) or, with shading = 1,
 * SHRender (tensorf_model_utils.py:292-296): xyz (N,3) normalised positions, view (N,3), features (N, app_dim) -> rgb (N,3).
 * Workspace: 2 x nvfi_app_workspace_bytes(f, N) is enough.  Forward only (inside a render the module is differentiated by nvfi_render_bwd). */
int nvfi_render_mlp(const nvfi_field_desc* f, int64_t N, const float* xyz, const float* view, const float* features, float* rgb,
                    void* workspace, int64_t workspace_bytes, void* stream);
/* ---- multi-GPU: rays and collocation points shard over one process per GPU; the ONLY data-path exchange is the sum of the
 *      flat fp32 gradient buffer (the reference has no distributed code: models/ is single-device).  RCCL over xGMI,
 *      loaded lazily with dlopen.  Bootstrap: rank 0 gets a 128-byte id, the host program distributes it, every rank inits.
 *      nvfi_allreduce_grads is in place and asynchronous on `stream`; average != 0 divides by the world size. */
#define NVFI_UNIQUE_ID_BYTES 128
typedef struct nvfi_comm nvfi_comm;
int nvfi_comm_unique_id(void* id128_host);
int nvfi_comm_init(nvfi_comm** comm, int world, int rank, const void* id128_host);
int nvfi_allreduce_grads(nvfi_comm* comm, float* flat_grads, int64_t count, int average, void* stream);
int nvfi_comm_destroy(nvfi_comm* comm);
/* per-kernel-class HIP-event timing for bench.py: enable, run, collect (host arrays of nvfi_prof_nclasses() entries).
 * classes: 0 rk2_fwd 1 rk2_bwd 2 app_fwd 3 app_bwd 4 wgrad 5 pde_fwd 6 pde_bwd 7 density_fwd 8 density_bwd 9 pde_prefilter */
int nvfi_prof_enable(int on);
int nvfi_prof_collect(double* total_ms, int64_t* count);
int nvfi_prof_nclasses(void);
/* compute_alpha (tensorf_keyframe.py:508-537) for a per-call time: xyz (N,3) WORLD coordinates -> normalise, snap to the keyframe
 * (base 0 when transfer), RK2 back-advect, density, alpha = 1-exp(-sigma*length); alpha_out[n] = max(alpha_out[n], alpha) when
 * accumulate_max != 0 (getDenseAlpha's running maximum over the 60 frame times, :476-497), plain store otherwise. */
int nvfi_alpha_workspace_bytes(const nvfi_field_desc* f, int64_t N, int64_t* bytes);
int nvfi_compute_alpha(const nvfi_field_desc* f, int64_t N, const float* xyz_world, float t, int transfer, float length,
                       int accumulate_max, float* alpha_out, void* workspace, int64_t workspace_bytes, void* stream);
/* Camera.get_ray_bundle + pixel selection (models/camera.py:112-138,159-172): pose (device float[12], row-major 3x4 c2w),
 * pixel ids (device int64[n], row-major y*W+x) -> rays_o (n,3), rays_d (n,3) */
int nvfi_gen_rays(const float* pose3x4, int H, int W, float focal, int64_t n, const int64_t* pixel_ids, float* rays_o, float* rays_d, void* stream);
/* ---- the random inputs of ONE training iteration in one launch (ABI v5): what train_nvfi.py:150-178 draws per render (a pixel batch of the
 *      posed image: Camera.sample_rays, models/camera.py:159-172) and what NVFi.get_vel_loss draws (models/nvfi.py:44-47: collocation points
 *      uniform in the box, times uniform in [0,1)), from a counter-based generator (Philox4x32-10 keyed by `seed`; counter = element, segment,
 *      iteration): same (seed, iteration) -> same batch, whatever the launch configuration.  Per batch b < n_batches: R pixel indices uniform in
 *      [0, n_pixels) -> rays_o/rays_d[b] gathered from the camera bundle (n_pixels,3), target[b] gathered from target_img (n_pixels,3) or, when
 *      that is NULL, uniform in [0,1) (synthetic benchmark targets); pixel_ids[b] (int64, optional) receives the indices.  P points + times.
 *      The R pixels of a batch are DISTINCT (np.random.choice(replace=False), camera.py:160): pixel r = perm_b(r), a keyed bijection of
 *      [0, n_pixels) (four-round Feistel network, cycle-walked); R > n_pixels is refused like the reference's draw raises. */
typedef struct nvfi_draw_desc {
    uint64_t seed, iteration;
    const uint64_t* iteration_dev;   /* optional: the iteration counter in DEVICE memory (hipGraph replay) */
    int32_t n_batches; int64_t R; int64_t n_pixels;
    const float* bundle_o; const float* bundle_d; const float* target_img;
    float* rays_o[2]; float* rays_d[2]; float* target[2]; int64_t* pixel_ids[2];
    int64_t P; float aabb[6]; float* points; float* t;
} nvfi_draw_desc;
int nvfi_draw_batch(const nvfi_draw_desc* d, void* stream);
/* The hand-rolled primitives the MLP engine uses instead of libm (engine.h: one-exp2/one-rcp sigmoid, Cody-Waite sin/cos), evaluated
 * element-wise so that tests can bound their error against float64: kind 0 sigmoid, 1 sin, 2 cos, 3 SiLU, 4 SiLU', 5 SiLU''. */
int nvfi_debug_act(int kind, int64_t n, const float* x, float* y, void* stream);
/* MFMA fragment-layout self test: returns max abs error of a 128x128 fp32 layer against a VALU loop (host float*) */
int nvfi_selftest(float* max_err_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif
