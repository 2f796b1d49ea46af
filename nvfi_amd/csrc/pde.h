// pde.h - argument blocks and stash geometry of the PDE kernels (pde.hip)
#pragma once
#include "common.h"
#include "vel.h"

#define PDE_MAX_CLASS 64           // RK2 step-count buckets
#define PDE_CHUNK 262144           // kept points processed per pass (bounds the stash: 32 KB per point -> 8.4 GB of address space, touched only up to the kept count; sized for 288 GB of HBM)
#ifndef PDE_NSLAB
#define PDE_NSLAB 256
#endif

// per-tile stash rows (each row = 64 floats)
#define PDE_Z     0                          // weight_net pre-activations        5*64
#define PDE_ZD    (PDE_Z + 320)              // tangent pre-activations       4 x 5*64
#define PDE_X0    (PDE_ZD + 4 * 320)         // encoder slots                      16
#define PDE_X0D   (PDE_X0 + 16)              // encoder tangents               4 x 16
#define PDE_GA    (PDE_X0D + 64)             // adjoints: value, 4 tangents   5 x 336
#define PDE_ZA    (PDE_GA + 5 * 336)         // a_weight_net pre-activations      320   (the 4 x 320 second-derivative correction rows of the round-1 column kernels went with them in round 6)
#define PDE_GAA   (PDE_ZA + 320)             // a_weight_net adjoints             336
#define PDE_TILE_ROWS (PDE_GAA + 336)

struct PdePrepArgs {
    nvfi_field_desc f;
    int64_t P;
    const float* points; const float* t;
    float4* qorig; float4* xw; float* pt_t; float* pt_base;
    int* cls; int* rank; int* cls_count;
};

// x4 fragment repack jobs (pde_jet.hip): four consecutive K steps of a lane side by side, so one 16-byte load feeds four MFMA steps
struct X4Jobs { const float* src[12]; float* dst[12]; int MT[12]; int NS[12]; int n; };
#define X4_FLOATS(MT, NS) ((MT) * (((NS) + 3) / 4) * 256)
// v-net fragments in x4 order: forward f[0] (4 tiles x 14 steps), f[1..4] (4 x 64), f[5] (1 x 64); transposed t[1..4], t[5] (4 x 4)
#define VEL_X4_FLOATS (X4_FLOATS(4, 14) + 4 * X4_FLOATS(4, 64) + X4_FLOATS(1, 64) + 4 * X4_FLOATS(4, 64) + X4_FLOATS(4, 4))

// bookkeeping done by the last workgroup of k_pde_seeds (round 5; ticket == NULL: the separate k_pde_pass_count / _finish / _counters launches)
struct PdeTail {
    int* ticket;               // zero at launch
    int* dcount;               // k_pde_pass_count: [0] samples of this pass (whole 128-point groups), [8] the fused adjoint's queue word
    float* out;                // k_pde_finish: loss, n_kept, sum div^2, sum transport^2 (NULL: not the last pass of the call)
    int64_t* counters; const int* cls_count; int64_t P; int pre16;     // k_pde_counters (NULL: not wanted / not the last pass)
};

struct PdeJetArgs {
    PdeTail tail;
    VelFrags Wv, Wa;
    const float4* f4[6]; const float4* t4[6]; const float* bv[6];   // x4 fragments / bias fragments of weight_net (fused jet kernels)
    int jet_tiles;             // fused jet launches: workgroups [0, jet_tiles) are jet tiles, the rest the acceleration net's column
    const float4* qorig; const int* klist;
    int64_t first; const int* kcount; int64_t cap; int wgs;     // kcount: DEVICE count of kept points; this pass handles [first, first + cap)
    float* stash; float* seeds; float* wout; double* sums;
    float scale; const float* scale_dev;   // loss scale by value, or (non-NULL) read from device memory (hipGraph replay: a weight that decays every iteration)
    float* jac; int64_t n_jac;
    int x4;                    // k_pde_jet_fwd: layers 0..3 of z / zd_j in x4 stash blocks (engine.h: stash_st16_x4) - their only reader is then k_pde_fuse_bwd
};

// kept points of this pass: the launch grids are sized for the worst case (every candidate kept) and workgroups beyond the
// device-side count leave at once - the host never learns the count, so the call needs no synchronisation
__device__ __forceinline__ int pde_pass_count_of(const PdeJetArgs& a) {
    const int64_t c = (int64_t)(*a.kcount) - a.first;
    return c <= 0 ? 0 : (c > a.cap ? (int)a.cap : (int)c);
}
int launch_frag_x4(const X4Jobs& jobs, hipStream_t st);

// feature-split per-point RK2 (vel_split.hip): one 32-point tile per workgroup, x4 weight fragments straight from L2
struct SplitArgs {
    nvfi_field_desc f;
    const float4* f4[6]; const float* bv[6];
    const int* count; int64_t n_direct;      // device count of list entries (NULL -> n_direct)
    const int* list; float4* xw;             // list[i] -> point index (NULL: identity); positions updated in place
    const float* pt_t; const float* pt_base; int pt_by_list; float dt_max; int max_steps;
};
int launch_rk2_split(const SplitArgs& a, int64_t cap_points, int wide, hipStream_t st);
// the render warp on the same layout (uniform step sequence, optional training stash)
#define VEL_X4F_FLOATS (X4_FLOATS(4, 14) + 4 * X4_FLOATS(4, 64) + X4_FLOATS(1, 64))     // forward fragments only
struct SplitUniArgs { Rk2Args r; const float4* f4[6]; const float* bv[6]; };
int launch_rk2_split_uni(const SplitUniArgs& a, int64_t cap_samples, bool stash, hipStream_t st);
// the adjoint (vel_split.hip: k_rk2_split_bwd); t4[0] = T0 (1 tile x 64 steps), t4[1..4], t4[5] (4 tiles x 4 steps)
#define VEL_X4B_FLOATS (X4_FLOATS(1, 64) + 4 * X4_FLOATS(4, 64) + X4_FLOATS(4, 4))
struct SplitBwdArgs { Rk2Args r; const float4* t4[6]; };
int launch_rk2_split_bwd(const SplitBwdArgs& a, int64_t cap_samples, hipStream_t st);
int pack_vel_x4_bwd(const VelFrags& W, float* buf, const float4** t4, hipStream_t st);
// x4 copies of the forward fragments of a packed VelFrags into buf (VEL_X4F_FLOATS); fills f4[6]
int pack_vel_x4_fwd(const VelFrags& W, float* buf, const float4** f4, hipStream_t st);

// opt-in fp16 pre-pass of the prefilter (pre16.hip)
#define PRE16_IMAGE_BYTES 150528   // fp16 fragments of the six weight_net layers (144 KB) + fp32 biases: staged into LDS once per workgroup
struct Pre16Args {
    nvfi_field_desc f;
    void* img;                     // PRE16_IMAGE_BYTES of workspace
    void* img_lo;                  // split mode (split16band): PRE16_IMAGE_BYTES more for the second binary16 term of the weights; NULL otherwise
    int64_t P; const int* list;    // bucket order -> point index
    const float4* xw; float4* xout; uint8_t* near;
    const float* pt_t; const float* pt_base; float dt_max; int max_steps; float eps_gate;
};
int launch_pre16(const nvfi_field_desc* f, Pre16Args a, hipStream_t st);
// fp16-input inference back-advection (nvfi_field_desc.vel_fp16): P = capacity, count = optional device-side count, list = optional
// compact -> dense map; per-point mode reads pt_t / pt_base at the compact index, uniform mode the step schedule (by value or `sched`)
struct Rk16Args {
    nvfi_field_desc f;
    void* img;                     // 2 x PRE16_IMAGE_BYTES of workspace (the second half holds the lo image of the split mode, vel_fp16 = 2)
    void* img_lo;                  // set by the launcher
    int64_t P; const int* count; const int* list;
    const float4* xw; float4* xout; float* xout3;
    const float* pt_t; const float* pt_base; float dt_max; int max_steps;
    int nsteps; float dt[64]; float tcur[64]; const float* sched;
    // training stash (uniform mode, vel_fp16 bit 2): the forward's pre-activations / encoder slots / per-(step, sample) records in exactly the
    // layout k_rk2_split_uni<STASH> writes, so the fp32 adjoint (k_rk2_fuse_bwd / k_rk2_split_bwd) and k_wgrad_ring8 read them unchanged
    float* zst; float* x0st; float* rec; int64_t cap; int64_t cap_tiles;
};
int launch_rk2_inf16(const nvfi_field_desc* f, Rk16Args a, bool uniform, hipStream_t st, bool stash = false);
int launch_pde_band(const nvfi_field_desc* f, int64_t P, const int* perm, const float* sig, const uint8_t* near, float band, uint8_t* flags,
                    int* cnt, hipStream_t st);
int launch_pde_band_map(int64_t P, const int* bcount, const int* perm, int* blist, hipStream_t st);
// tiles: capacity in 32-point tiles (five weight_net columns each); anet_wgs: capacity in 128-point workgroups of the acceleration net
int launch_pde_jet_fwd(const PdeJetArgs& a, unsigned tiles, unsigned anet_wgs, hipStream_t st);
int launch_pde_jet_bwd(const PdeJetArgs& a, unsigned tiles, unsigned anet_wgs, hipStream_t st);
// pde_jet6.hip (round 6): the forward with the hidden layers on the 16-bit matrix pipe (x6img: the three bfloat16 weight images of x6.h)
int launch_pde_jet6_fwd(const PdeJetArgs& a, const void* x6img, unsigned tiles, unsigned anet_wgs, hipStream_t st);

// value adjoint with the correction term; no input gradient needed
template <int ACT, bool CORR>
__device__ __forceinline__ void velnet_value_backward(const VelFrags& W, float* lds_w, float* lds_b, int lane, const float* seed4,
                                                      const float* zst, const float* corr, float* gst) {
    float g[64];
    f32x16 acc[4];
    g[0] = seed4[0]; g[1] = seed4[1]; g[2] = seed4[2]; g[3] = seed4[3];
    {
        float* gw_rows = gst + (size_t)5 * 64 * REGF;
#pragma unroll
        for (int s = 0; s < 16; ++s) gw_rows[s * REGF + lane] = s < 4 ? g[s] : 0.f;
    }
    __syncthreads();
    stage_frag(lds_w, lds_b, W.t[5], VEL_T5, nullptr, 0);
    __syncthreads();
    acc_init<4>(acc, lds_b, 0, false);
    layer_mfma<4, 4>(lds_w, lane, g, acc);
#pragma unroll 1
    for (int l = 4; l >= 0; --l) {
        const float* zl = zst + (size_t)l * 64 * REGF;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int s = 16 * m + r;
                float v = act_d1<ACT>(zl[s * REGF + lane]) * acc[m][r];
                if (CORR) {
                    const float* c0 = corr + (size_t)l * 64 * REGF + s * REGF + lane;
                    v += (c0[0] + c0[(size_t)320 * REGF]) + (c0[(size_t)640 * REGF] + c0[(size_t)960 * REGF]);
                }
                g[s] = v;
            }
        stash_store<64>(gst + (size_t)l * 64 * REGF, lane, g);
        if (l >= 1) {
            __syncthreads();
            stage_frag(lds_w, lds_b, W.t[l], VEL_FH, nullptr, 0);
            __syncthreads();
            acc_init<4>(acc, lds_b, 0, false);
            layer_mfma<4, 64>(lds_w, lane, g, acc);
        }
    }
}


int launch_pde_wgrad(const float* stash, int ntiles, float* slabs, int* dcount, const nvfi_grads* G, hipStream_t st, int fused_nslab = 0, const float* fused_slabs = nullptr,
                     const float* fused_slabs_a = nullptr);
