/*
 * nvfi_oracle.h - CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the NVFi render + physics-loss hot path, used only as the checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product path
 * (the .hip sources under nvfi_amd/csrc behind include/nvfi_hip.h) never links, imports or calls this file.
 *
 * Parity status: PINNED.  Every function here is checked against golden vectors produced by
 * importing the reference implementation (PyTorch CPU) in the build container
 * (tests/golden/make_golden.py -> tests/golden/ npz files, tests/test_oracle_golden.py).
 *
 * Layouts are the reference's logical ones: planes are (C,H,W) row-major (torch NCHW with N=1),
 * Linear weights are (out,in) row-major.
 */
#ifndef NVFI_ORACLE_H
#define NVFI_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t G[3];        /* gridSize x,y,z                      (tensorf_base.py:219) */
    int32_t K;           /* num_keyframes                       (tensorf_keyframe.py:41) */
    int32_t Cd, Ca, app_dim;
    int32_t n_samples;   /* nSamples                            (tensorf_base.py:223) */
    int32_t use_vel;     /* cfg.use_vel                         (tensorf_keyframe.py:92) */
    int32_t gate_sur;    /* 0: VelocityAABB(eps) 1: VelocityAABBSur (velocity_field.py:21-51) */
    int32_t has_amask;   /* alphaMask present (eval only)       (tensorf_keyframe.py:656) */
    int32_t shading;     /* 0: MLP_PE (MLPRender_PE)  1: SH (SHRender on 27 features; tensorf_base.py:196-197) */
    int32_t am_dims[3];  /* alpha volume W,H,D */
    float aabb[6];       /* min xyz, max xyz */
    float near_, far_, step_size;
    float density_shift, distance_scale, weight_thres, alpha_thres, tmax;
    float gate_lo[3], gate_hi[3];
    const float *dps[3], *dpt[3], *aps[3], *apt[3];
    const float *basis;                                   /* (app_dim, Ca) */
    const float *rW[3], *rb[3];                           /* MLPRender_PE 110-128-128-3 */
    const float *vW[6], *vb[6];                           /* VelBasis.weight_net   */
    const float *aW[6], *ab[6];                           /* VelBasis.a_weight_net */
    const float *amask;                                   /* (D,H,W) */
} orc_field_t;

typedef struct {
    float *dps[3], *dpt[3], *aps[3], *apt[3];
    float *basis;
    float *rW[3], *rb[3];
    float *vW[6], *vb[6];
    float *aW[6], *ab[6];
} orc_grads_t;

/* flags for orc_render_fwd */
#define ORC_TRAIN      1   /* training mode: jitter u is used, alpha mask ignored */
#define ORC_WHITE_BG   2   /* add (1-acc) to rgb (white_bg or the random-white coin) */
#define ORC_TRANSFER   4   /* transfer_vel: base time = 0 */

typedef struct orc_ctx orc_ctx_t;

int  orc_num_threads(void);
void orc_set_num_threads(int n);

/* a-3 TensorBase.sample_ray (tensorf_base.py:290-314). pts may be NULL. valid is uint8. */
void orc_sample_ray(const orc_field_t* f, int64_t R, const float* o, const float* d, const float* u,
                    float* pts, float* z, uint8_t* valid);
/* a-8 VelBasis.forward -> (v,a) (N,6); get_vel -> (N,3); a-7 gated velocity (N,3) */
void orc_vel_net(const orc_field_t* f, int64_t N, const float* xt, float* u6);
void orc_get_vel(const orc_field_t* f, int64_t N, const float* xt, float* v3);
void orc_vel_gated(const orc_field_t* f, int64_t N, const float* xt, float* v3);
/* a-6 integrate_pos with per-point t/base (N,1) (tensorf_keyframe.py:575-611) */
void orc_integrate_pos(const orc_field_t* f, int64_t N, const float* x, const float* t, const float* base, float* xk);
/* a-9/a-10/a-12 */
void orc_density_feature(const orc_field_t* f, int64_t N, const float* xyzt, float* feat);
void orc_app_feature(const orc_field_t* f, int64_t N, const float* xyzt, float* feat);
void orc_feature2density(const orc_field_t* f, int64_t N, const float* feat, float* sigma);
/* a-11 raw2alpha (tensorf_model_utils.py:186-197) */
void orc_raw2alpha(int64_t R, int64_t S, const float* sigma, const float* dist, float* alpha, float* weight);
/* a-13 MLPRender_PE.forward */
void orc_render_mlp(const orc_field_t* f, int64_t N, const float* pts, const float* view, const float* feat, float* rgb);
/* a-15 AlphaGridMask.sample_alpha */
void orc_sample_alpha(const orc_field_t* f, int64_t N, const float* xyz, float* alpha);
/* a-17 SHRender */
void orc_sh_render(int64_t N, const float* view, const float* feat27, float* rgb);

/* a-1/2/5/14: one render_ray call. weight is (R,S). Returns a context holding the intermediates
 * the backward needs when keep_ctx != 0 (free with orc_ctx_free), NULL otherwise. */
orc_ctx_t* orc_render_fwd(const orc_field_t* f, int64_t R, const float* o, const float* d, const float* u,
                          float t, int flags, float* rgb, float* depth, float* acc, float* weight,
                          int64_t* counters /* [4]: valid, warped, app-masked, rk2 evals; may be NULL */,
                          int keep_ctx);
/* a-18: backward of orc_render_fwd; upstream grads may be NULL (= zero). Grad buffers are ACCUMULATED into. */
void orc_render_bwd(orc_ctx_t* ctx, const float* g_rgb, const float* g_depth, const float* g_acc,
                    const float* g_weight, orc_grads_t* grads);
void orc_ctx_free(orc_ctx_t* ctx);

/* a-16 NVFi.get_vel_loss with explicit collocation (world-space points (P,3), raw t (P,1)).
 * Returns the loss (0 when nothing is kept); kept (P) uint8 and jac (n_jac,6,4) optional;
 * grads (vW,vb,aW,ab only) accumulated when non-NULL. n_kept_out optional. */
float orc_pde_loss(const orc_field_t* f, int64_t P, const float* points, const float* t,
                   uint8_t* kept, int64_t* n_kept_out, int64_t n_jac, float* jac, orc_grads_t* grads,
                   int64_t* rk2_evals_out);

/* a-19 / config 5: MaskField (models/mask_field.py:68-83 built as train_segm.py:97-102: 3 -> 128 x 4 ReLU -> K, softmax).
 * W[l] (out,in) row-major, l = point_fc.0..3, mask_fc.  mask (N,K).  When g_mask != NULL the gradients of
 * sum(mask * g_mask) wrt every W[l], b[l] are ACCUMULATED into gW[l], gb[l] (autograd of train_segm.py:195). */
void orc_maskfield(const float* const* W, const float* const* b, int mask_dim, int64_t N, const float* xyz, float* mask,
                   const float* g_mask, float* const* gW, float* const* gb);

/* next-row f-1 regularisers */
float orc_density_L1(const orc_field_t* f);
float orc_tv_density(const orc_field_t* f);
float orc_tv_app(const orc_field_t* f);

/* test switch: velocity-net forward with weights / layer inputs rounded to binary16 (the product's vel_fp16 inference mode) */
void orc_set_vel_fp16(int on);
float orc_f16_round(float x);
#ifdef __cplusplus
}
#endif
#endif
