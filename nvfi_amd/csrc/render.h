// render.h - argument blocks of the render kernels (render.hip)
#pragma once
#include <string.h>
#include "common.h"
#include "vel.h"

struct SampleArgs {
    nvfi_field_desc f;
    int64_t R;
    const float* o; const float* d; const float* u;
    int train;
    const int* inside;
    float4* xw; float* xpre; uint8_t* valid; int* cnt;
    uint8_t* rflag; int* cnt_r;   // valid AND inside the velocity gate: the samples the RK2 warp has to touch (NULL: not wanted)
    // k_sample_fill (sampling + both ordered compact lists in one launch): look-back status words (zero at launch), the lists, their totals
    unsigned long long* lb; int* vlist; int* rlist; int* total_v; int* total_r;
};

struct DensityArgs {
    nvfi_field_desc f;
    const int* count; int64_t n_direct;
    const int* list;
    const float4* xw;
    float tn; int per_point_t;
    const float* sched;       // optional device-side schedule record (common.h); NULL: the by-value tn
    float* xpre; float* feat_out; float* sigma_out;
    // backward
    const float* gxpre;
    nvfi_grads g;
    const uint8_t* mflag; const float4* gxw; float4* gxk;
};

struct WeightArgs {
    int64_t R; int S;
    const float* xpre; const float4* xw;
    float distance_scale, weight_thres, far_;
    float* weight; uint8_t* mflag; float* acc; float* depth; int* cnt_m;
    // k_weights_fill (weights + the ordered list of appearance-masked samples in one launch)
    unsigned long long* lb; int* off_m_out; int* mlist; int* total_m;
    // backward
    const int* off_m; const float4* rgbs; const float4* rgb_pre;
    const float* g_rgb; const float* g_depth; const float* g_acc; const float* g_weight;
    float* gxpre;
    int white_bg;
};

struct FinalArgs {
    int64_t R;
    const int* off_m; const int* mlist;
    const float* weight; const float4* rgbs; const float* acc;
    int white_bg;
    float4* rgb_pre; float* rgb;
    // the call's counters (k_counters' job) written by workgroup 0 of the same launch; NULL: not wanted
    const int* c; int nsteps; int64_t* counters_out; const float* sched;
    // nvfi_render_fwd_mse: F.mse_loss(rgb, target) and its gradient from the same launch (target == NULL: not wanted).  g_rgb_out[i] =
    // loss_scale * 2 (rgb[i] - target[i]) / (3 R); per-workgroup partial sums, summed in workgroup order by the last one to finish (ticket)
    const float* target; float* g_rgb_out; float* loss_out; float* partial; int* ticket; float loss_scale;
};

struct AppArgs {
    nvfi_field_desc f;
    RenderFrags W;
    const int* count; int64_t n_direct;
    const int* list;
    const float4* xw;
    float tn; int per_point_t; int S;
    const float* sched;
    const float* rays_d; const float* view_per_point;
    float4* rgbs; int rgb_dense;
    const float* feat48;   // (M,48) plane-product features of the masked samples by compact index (k_app_feat): no plane gather in k_app_fwd
    const float* feat_in;  // (N, app_dim) appearance features given by the caller (renderModule as a stand-alone call): no plane gather, no basis_mat
    float* stash_f; float* stash_b;
    unsigned* relu_mask;   // [tile][layer 1 | layer 2][lo | hi][64 lanes]: bit s of a lane = (hidden activation register s > 0), written by k_app_fwd<stash>,
                           // read by k_app_bwd instead of the 128 activation rows themselves (1 KB instead of 32 KB per tile)
    // backward
    nvfi_grads g;
    const float* g_rgb; const float4* rgb_pre; const float* weight;
    float4* gxw;
    float* gg;           // (M,48) per-sample channel gradients for k_plane_scatter (NULL: scatter in-kernel)
    int plane_tail;      // 1: coordinate gradients of the plane lookups (+ in-kernel scatter when gg is NULL) are computed here; 0: k_og does it (or nobody needs them)
};

struct ScatterArgs {
    nvfi_field_desc f;
    const int* count; const int* list;
    const float4* xw; float tn;
    const float* sched;
    const float* gxpre;   // density: one upstream gradient per sample
    const float* gg;      // appearance: (M,48)
    nvfi_grads g;
    int y0, gmax;         // LDS variant: first time row touched by this call, max grid extent
    int plane_mask;       // debug: bit p enables scattering into plane p (default 63)
};

__global__ void k_counters(const int* c, int nsteps, int64_t* out, const float* sched = nullptr);
__global__ void k_unpack_rgb(const float4* in, float* out, int64_t N);
__global__ void k_pack_xyz4(const float* in, float4* out, int64_t N);
int launch_vel_wgrad(const float* zst, const float* x0st, const float* gst, const int* count, int cap_tiles, int nrep,
                     int act_mode, float* slabs, int nslab, float* const* gW, float* const* gb, float scale, hipStream_t st, int fused_nslab = 0,
                     const WgradJobs* pre_w = nullptr, const ReduceJobs* pre_r = nullptr);
