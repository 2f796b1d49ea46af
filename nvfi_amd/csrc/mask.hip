// mask.hip - MaskField (reference models/mask_field.py:34-83 as train_segm.py:97-102 builds it: 3 -> 128 x 4 (ReLU) -> K,
// softmax) on free points, forward and backward: the per-iteration model of train_segm.py:126-227 (BASELINE config 5).
//
// Same fp32 MFMA sample-tile engine as the velocity and render MLPs (engine.h): a wave owns 32 points, activations stay in
// registers between layers, one layer's weight fragment is staged in LDS per workgroup.  Training stashes the post-ReLU
// activations (they are the B operands of the weight gradients and carry the ReLU mask of the adjoint pass) and the softmax
// output; the backward walks the transposed fragments and leaves the adjoints in the MFMA register layout for k_wgrad.
#include "common.h"
#include "engine16.h"
#include <string.h>

#define MK_F_ROWS (16 + 4 * 64 + 16)   // forward stash per tile: x0 (16) | h1..h4 (64 each) | softmax (16)
#define MK_B_ROWS (16 + 4 * 64)        // backward stash per tile: g_logits (16) | g_z4, g_z3, g_z2, g_z1 (64 each)
#define MK_NSLAB 256
#define MK_SLAB_FLOATS (128 * 128 + 128)
#define MK_F0 (4 * 2 * 64)
#define MK_FH (4 * 64 * 64)
#define MK_F4 (1 * 64 * 64)
#define MK_T4 (4 * 16 * 64)            // transposed last layer: rows = 128 hidden features, slots = 32 logit rows
#define MK_FRAG_FLOATS (MK_F0 + 3 * MK_FH + MK_F4 + 5 * 128 + MK_T4 + 3 * MK_FH)

struct MkFrags { const float* f[5]; const float* b[5]; const float* t[5]; };   // t[0] unused (no gradient wrt the points)
struct MkArgs {
    MkFrags W; int mask_dim; int64_t N;
    const float* xyz; float* out;        // (N,3) -> (N,mask_dim)
    const float* g_out;                  // backward: d loss / d mask (N,mask_dim)
    float* stash_f; float* stash_b;
    unsigned* relu_mask;                 // [tile][hidden layer 0..3][lo | hi][64 lanes]: signs of the post-ReLU activations (the adjoint reads these, not the rows)
};
#define MK_MASK_WORDS (4 * 128)

template <bool STASH>
__global__ __launch_bounds__(WG_THREADS, 2) void k_maskfield_fwd(MkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_w = lds; float* lds_b = lds + LDS_W_FLOATS;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave_id();
    const int64_t i = (int64_t)tile * TILE + (lane & 31);
    const bool active = i < a.N;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (active) { px = a.xyz[3 * i]; py = a.xyz[3 * i + 1]; pz = a.xyz[3 * i + 2]; }
    float* st = STASH ? a.stash_f + (size_t)tile * (MK_F_ROWS * REGF) : nullptr;
    float xa[64], xb[64];
    xb[0] = h ? py : px; xb[1] = h ? 0.f : pz;
    if (STASH) {
#pragma unroll
        for (int s = 0; s < 16; ++s) st[s * REGF + lane] = s < 2 ? xb[s] : 0.f;
    }
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[0], MK_F0, a.W.b[0], 128);
    __syncthreads();
    layer_tiles<4, 2>(lds_w, lds_b, true, lane, h, xb, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { xa[16 * m + r] = fmaxf(acc[r], 0.f); if (STASH) st[(16 + 16 * m + r) * REGF + lane] = xa[16 * m + r]; }
    });
    if (STASH) relu_mask_store(a.relu_mask + (size_t)tile * MK_MASK_WORDS, lane, xa);
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[1], MK_FH, a.W.b[1], 128);
    __syncthreads();
    layer_tiles<4, 64>(lds_w, lds_b, true, lane, h, xa, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { xb[16 * m + r] = fmaxf(acc[r], 0.f); if (STASH) st[(80 + 16 * m + r) * REGF + lane] = xb[16 * m + r]; }
    });
    if (STASH) relu_mask_store(a.relu_mask + (size_t)tile * MK_MASK_WORDS + 128, lane, xb);
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[2], MK_FH, a.W.b[2], 128);
    __syncthreads();
    layer_tiles<4, 64>(lds_w, lds_b, true, lane, h, xb, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { xa[16 * m + r] = fmaxf(acc[r], 0.f); if (STASH) st[(144 + 16 * m + r) * REGF + lane] = xa[16 * m + r]; }
    });
    if (STASH) relu_mask_store(a.relu_mask + (size_t)tile * MK_MASK_WORDS + 256, lane, xa);
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[3], MK_FH, a.W.b[3], 128);
    __syncthreads();
    layer_tiles<4, 64>(lds_w, lds_b, true, lane, h, xa, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { xb[16 * m + r] = fmaxf(acc[r], 0.f); if (STASH) st[(208 + 16 * m + r) * REGF + lane] = xb[16 * m + r]; }
    });
    if (STASH) relu_mask_store(a.relu_mask + (size_t)tile * MK_MASK_WORDS + 384, lane, xb);
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[4], MK_F4, a.W.b[4], 32);
    __syncthreads();
    float o[16];
    layer_tiles<1, 64>(lds_w, lds_b, true, lane, h, xb, [&](int, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = acc[r];
    });
    // softmax over the mask_dim logits of the point: rows (r&3)+8(r>>2)+4h live in this lane, the rest in lane^32
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; if (row < a.mask_dim) mx = fmaxf(mx, o[r]); }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; o[r] = row < a.mask_dim ? expf(o[r] - mx) : 0.f; sum += o[r]; }
    sum += __shfl_xor(sum, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float p = o[r] / sum;
        if (STASH) st[(272 + r) * REGF + lane] = p;
        if (active && row < a.mask_dim) a.out[(size_t)i * a.mask_dim + row] = p;
    }
}

// adjoint pass: g_logits = p * (g - sum_k g_k p_k), then g_z_l = (W_{l+1}^T g_z_{l+1}) * [h_l > 0]
__global__ __launch_bounds__(WG_THREADS, 2) void k_maskfield_bwd(MkArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_w = lds; float* lds_b = lds + LDS_W_FLOATS;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave_id();
    const int64_t i = (int64_t)tile * TILE + (lane & 31);
    const bool active = i < a.N;
    const float* stf = a.stash_f + (size_t)tile * (MK_F_ROWS * REGF);
    float* stb = a.stash_b + (size_t)tile * (MK_B_ROWS * REGF);
    float g[64];
    {
        float p[16], go[16], dot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            p[r] = stf[(272 + r) * REGF + lane];
            go[r] = (active && row < a.mask_dim) ? a.g_out[(size_t)i * a.mask_dim + row] : 0.f;
            dot += go[r] * p[r];
        }
        dot += __shfl_xor(dot, 32);
#pragma unroll
        for (int r = 0; r < 16; ++r) { g[r] = p[r] * (go[r] - dot); stb[r * REGF + lane] = g[r]; }
    }
    f32x16 acc[4];
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.t[4], MK_T4, nullptr, 0);
    __syncthreads();
    acc_init<4>(acc, lds_b, 0, false);
    layer_mfma<4, 16>(lds_w, lane, g, acc);
#pragma unroll 1
    for (int l = 3; l >= 1; --l) {
        // acc = gradient wrt h_{l+1}: mask with the SIGNS of the stashed activation (two words per lane instead of 64 rows), stash as
        // g_z_{l+1}, push through W_{l+1}^T
        float* gz = stb + (size_t)(16 + 64 * (3 - l)) * REGF;
        relu_mask_apply(a.relu_mask + (size_t)tile * MK_MASK_WORDS + 128 * l, lane, acc, g);
        stash_store<64>(gz, lane, g);
        __syncthreads();
        stage_frag(lds_w, lds_b, a.W.t[l], MK_FH, nullptr, 0);
        __syncthreads();
        acc_init<4>(acc, lds_b, 0, false);
        layer_mfma<4, 64>(lds_w, lane, g, acc);
    }
    {   // g_z1 (no further propagation: the points carry no gradient, train_segm.py:137-170 runs them under no_grad)
        float* gz = stb + (size_t)(16 + 64 * 3) * REGF;
        relu_mask_apply(a.relu_mask + (size_t)tile * MK_MASK_WORDS, lane, acc, g);
        stash_store<64>(gz, lane, g);
    }
}

// ---------------------------------------------------------------- fp16-input MFMA variant (BASELINE config 5: "fp16 MFMA MLP")
// Same data flow with v_mfma_f32_32x32x16_f16: weights and layer inputs are rounded to fp16, products accumulate in fp32,
// bias / ReLU / softmax and every stash stay fp32 (the weight gradients still run on the fp32 k_wgrad).  One MFMA covers 16 input
// features: lane (n, h) supplies k = 8h + j  <->  feature 16 s + 8 (j >> 2) + 4 h + (j & 3), which is exactly registers
// 8s .. 8s+7 of the previous layer's D-layout output - so, as in the fp32 engine, activations never leave registers and the
// permutation lives in the packed weight fragments.  32 MFMAs of 32 cycles per 128x128 layer instead of 256 of 64.
__host__ __device__ inline int feat16(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

struct Pack16Job { const float* W; h8_t* frag; int out, in, MT, NS, transposed; };
struct Pack16Jobs { Pack16Job j[12]; int n; };
__global__ void k_pack16(Pack16Jobs jobs) {
    const Pack16Job& J = jobs.j[blockIdx.x];
    const int total = J.MT * J.NS * 64;
    for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += gridDim.y * blockDim.x) {
        const int lane = idx & 63, ms = idx >> 6, sidx = ms % J.NS, m = ms / J.NS;
        const int row = 32 * m + (lane & 31), h = lane >> 5;
        h8_t v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = feat16(sidx, h, j);
            float w = 0.f;
            if (!J.transposed) { if (row < J.out && k < J.in) w = J.W[(size_t)row * J.in + k]; }      // rows = outputs, k = inputs
            else { if (row < J.in && k < J.out) w = J.W[(size_t)k * J.in + row]; }                    // rows = inputs, k = outputs
            v[j] = (_Float16)w;
        }
        J.frag[idx] = v;
    }
}

struct Mk16Frags { const h8_t* f[5]; const float* b[5]; const h8_t* t[5]; };
struct Mk16Args {
    Mk16Frags W; int mask_dim; int64_t N;
    const float* xyz; float* out; const float* g_out;
    float* stash_f; float* stash_b;
    unsigned* relu_mask;
};

__device__ __forceinline__ void stage16(h8_t* lds_w, const h8_t* __restrict__ frag, int n8, float* lds_b, const float* __restrict__ bias, int nb) {
    const float4* src = reinterpret_cast<const float4*>(frag);
    float4* dst = reinterpret_cast<float4*>(lds_w);
    for (int i = threadIdx.x; i < n8; i += WG_THREADS) dst[i] = src[i];
    if (threadIdx.x < 128) lds_b[threadIdx.x] = (bias && threadIdx.x < nb) ? bias[threadIdx.x] : 0.f;
}
// one layer: MT output tiles, NS k-steps; epi(m, acc)
template <int MT, int NS, class Epi>
__device__ __forceinline__ void layer16(const h8_t* lds_w, const float* lds_b, bool bias, int lane, int h, const h8_t* B, Epi epi) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias ? lds_b[32 * m + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) acc = MFMA16(lds_w[(m * NS + sidx) * 64 + lane], B[sidx], acc);
        epi(m, acc);
    }
}

template <bool STASH>
__global__ __launch_bounds__(WG_THREADS, 2) void k_maskfield_fwd16(Mk16Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    h8_t* lds_w = reinterpret_cast<h8_t*>(lds); float* lds_b = lds + LDS_W_FLOATS;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave_id();
    const int64_t i = (int64_t)tile * TILE + (lane & 31);
    const bool active = i < a.N;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (active) { px = a.xyz[3 * i]; py = a.xyz[3 * i + 1]; pz = a.xyz[3 * i + 2]; }
    float* st = STASH ? a.stash_f + (size_t)tile * (MK_F_ROWS * REGF) : nullptr;
    float xa[64], xb[64];
    if (STASH) {   // the point slots in the fp32 engine's layout (B operand of the layer-0 weight gradient)
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) st[s2 * REGF + lane] = s2 == 0 ? (h ? py : px) : (s2 == 1 ? (h ? 0.f : pz) : 0.f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) xb[k] = 0.f;
    if (h == 0) { xb[0] = px; xb[1] = py; xb[2] = pz; }     // features 0..2 = k 0..2 of lane half 0
    h8_t B[8];
    __syncthreads();
    stage16(lds_w, a.W.f[0], 4 * 1 * 64, lds_b, a.W.b[0], 128);
    __syncthreads();
    to_h8<1>(xb, B);
    layer16<4, 1>(lds_w, lds_b, true, lane, h, B, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { xa[16 * m + r] = fmaxf(acc[r], 0.f); if (STASH) st[(16 + 16 * m + r) * REGF + lane] = xa[16 * m + r]; }
    });
    if (STASH) relu_mask_store(a.relu_mask + (size_t)tile * MK_MASK_WORDS, lane, xa);
#pragma unroll 1
    for (int l = 1; l < 4; ++l) {
        __syncthreads();
        stage16(lds_w, a.W.f[l], 4 * 8 * 64, lds_b, a.W.b[l], 128);
        __syncthreads();
        to_h8<8>(xa, B);
        float* sl = STASH ? st + (size_t)(16 + 64 * l) * REGF : nullptr;
        layer16<4, 8>(lds_w, lds_b, true, lane, h, B, [&](int m, const f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { xb[16 * m + r] = fmaxf(acc[r], 0.f); if (STASH) sl[(16 * m + r) * REGF + lane] = xb[16 * m + r]; }
        });
        if (STASH) relu_mask_store(a.relu_mask + (size_t)tile * MK_MASK_WORDS + 128 * l, lane, xb);
#pragma unroll
        for (int k = 0; k < 64; ++k) xa[k] = xb[k];
    }
    __syncthreads();
    stage16(lds_w, a.W.f[4], 1 * 8 * 64, lds_b, a.W.b[4], a.mask_dim);
    __syncthreads();
    to_h8<8>(xa, B);
    float o[16];
    layer16<1, 8>(lds_w, lds_b, true, lane, h, B, [&](int, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = acc[r];
    });
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; if (row < a.mask_dim) mx = fmaxf(mx, o[r]); }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; o[r] = row < a.mask_dim ? expf(o[r] - mx) : 0.f; sum += o[r]; }
    sum += __shfl_xor(sum, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float p = o[r] / sum;
        if (STASH) st[(272 + r) * REGF + lane] = p;
        if (active && row < a.mask_dim) a.out[(size_t)i * a.mask_dim + row] = p;
    }
}

__global__ __launch_bounds__(WG_THREADS, 2) void k_maskfield_bwd16(Mk16Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    h8_t* lds_w = reinterpret_cast<h8_t*>(lds); float* lds_b = lds + LDS_W_FLOATS;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave_id();
    const int64_t i = (int64_t)tile * TILE + (lane & 31);
    const bool active = i < a.N;
    const float* stf = a.stash_f + (size_t)tile * (MK_F_ROWS * REGF);
    float* stb = a.stash_b + (size_t)tile * (MK_B_ROWS * REGF);
    float g[64], gn[64];
    {
        float p[16], go[16], dot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            p[r] = stf[(272 + r) * REGF + lane];
            go[r] = (active && row < a.mask_dim) ? a.g_out[(size_t)i * a.mask_dim + row] : 0.f;
            dot += go[r] * p[r];
        }
        dot += __shfl_xor(dot, 32);
#pragma unroll
        for (int r = 0; r < 16; ++r) { g[r] = p[r] * (go[r] - dot); stb[r * REGF + lane] = g[r]; }
    }
    h8_t B[8];
    __syncthreads();
    stage16(lds_w, a.W.t[4], 4 * 2 * 64, lds_b, nullptr, 0);
    __syncthreads();
    to_h8<2>(g, B);
    layer16<4, 2>(lds_w, lds_b, false, lane, h, B, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gn[16 * m + r] = acc[r];
    });
#pragma unroll 1
    for (int l = 3; l >= 1; --l) {
        float* gz = stb + (size_t)(16 + 64 * (3 - l)) * REGF;
        {
            const unsigned* mk = a.relu_mask + (size_t)tile * MK_MASK_WORDS + 128 * l;
            const unsigned mlo = mk[lane], mhi = mk[64 + lane];
#pragma unroll
            for (int k = 0; k < 64; ++k) g[k] = (((k < 32 ? mlo : mhi) >> (k & 31)) & 1u) ? gn[k] : 0.f;
        }
        stash_store<64>(gz, lane, g);
        __syncthreads();
        stage16(lds_w, a.W.t[l], 4 * 8 * 64, lds_b, nullptr, 0);
        __syncthreads();
        to_h8<8>(g, B);
        layer16<4, 8>(lds_w, lds_b, false, lane, h, B, [&](int m, const f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gn[16 * m + r] = acc[r];
        });
    }
    {
        float* gz = stb + (size_t)(16 + 64 * 3) * REGF;
        {
            const unsigned* mk = a.relu_mask + (size_t)tile * MK_MASK_WORDS;
            const unsigned mlo = mk[lane], mhi = mk[64 + lane];
#pragma unroll
            for (int k = 0; k < 64; ++k) g[k] = (((k < 32 ? mlo : mhi) >> (k & 31)) & 1u) ? gn[k] : 0.f;
        }
        stash_store<64>(gz, lane, g);
    }
}

__global__ void k_set_int(int* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }

struct MkPlan { float* frag; float* stash_f; float* stash_b; float* slabs; unsigned* relu; int* count; int64_t tiles; int64_t total; };
static void plan_mask(int64_t N, int train, void* ws, MkPlan* P) {
    Bump B{(char*)ws, 0, 0};
    P->tiles = (N + WG_SAMPLES - 1) / WG_SAMPLES * 4;
    P->frag = B.take<float>(MK_FRAG_FLOATS);
    P->count = B.take<int>(16);
    P->stash_f = P->stash_b = P->slabs = nullptr; P->relu = nullptr;
    if (train) {
        P->relu = B.take<unsigned>(P->tiles * (int64_t)MK_MASK_WORDS);
        P->stash_f = B.take<float>(P->tiles * (int64_t)(MK_F_ROWS * REGF));
        P->stash_b = B.take<float>(P->tiles * (int64_t)(MK_B_ROWS * REGF));
        P->slabs = B.take<float>((int64_t)MK_NSLAB * MK_SLAB_FLOATS * 5);
    }
    P->total = align_up(B.off, 256);
}

static int mask_check(const nvfi_mask_desc* m) {
    if (m->n_layer != 4 || m->n_dim != 128 || m->mask_dim < 1 || m->mask_dim > 32)
        return nvfi_fail(2, "mask field must be 3->128x4->mask_dim<=32 (train_segm.py:97-102); got n_layer=%d n_dim=%d mask_dim=%d", m->n_layer, m->n_dim, m->mask_dim);
    for (int l = 0; l < 5; ++l) if (!m->W[l] || !m->b[l]) return nvfi_fail(2, "mask field layer %d has no weight/bias pointer", l);
    return 0;
}

static int mask_frags(const nvfi_mask_desc* m, float* frag, MkFrags* W, bool transposed, hipStream_t st) {
    PackJobs jobs; jobs.n = 0;
    float* p = frag;
    for (int l = 0; l < 5; ++l) {
        PackJob& J = jobs.j[jobs.n++];
        memset(&J, 0, sizeof(J));
        const int MT = l < 4 ? 4 : 1, NS = l == 0 ? 2 : 64;
        J.W = m->W[l]; J.b = m->b[l]; J.frag = p; p += MT * NS * 64; J.bfrag = p; p += 128;
        J.out = l < 4 ? 128 : m->mask_dim; J.in = l == 0 ? 3 : 128; J.MT = MT; J.NS = NS;
        J.row_kind = RK_NATURAL; J.slot_kind = l == 0 ? SK_XYZ : SK_HIDDEN; J.transposed = 0; J.x4 = 0;
        W->f[l] = J.frag; W->b[l] = J.bfrag;
    }
    W->t[0] = nullptr;
    for (int l = 1; l < 5; ++l) {           // dgrad fragments: rows = input features of layer l, slots = its output rows
        PackJob& J = jobs.j[jobs.n++];
        memset(&J, 0, sizeof(J));
        const int NS = l < 4 ? 64 : 16;
        J.W = m->W[l]; J.b = nullptr; J.frag = p; p += 4 * NS * 64; J.bfrag = nullptr;
        J.out = l < 4 ? 128 : m->mask_dim; J.in = 128; J.MT = 4; J.NS = NS;
        J.row_kind = RK_NATURAL; J.slot_kind = SK_HIDDEN; J.transposed = 1; J.x4 = 0;
        W->t[l] = J.frag;
    }
    if (!transposed) jobs.n = 5;
    return launch_pack(jobs, st);
}

// fp16 fragments inside the same workspace region: forward l=0 (4x1), l=1..3 (4x8), l=4 (1x8); transposed l=1..3 (4x8), l=4 (4x2)
static void mask_frag16_ptrs(float* frag, Mk16Frags* W, const nvfi_mask_desc* m) {
    h8_t* p = reinterpret_cast<h8_t*>(frag);
    for (int l = 0; l < 5; ++l) { const int MT = l < 4 ? 4 : 1, NS = l == 0 ? 1 : 8; W->f[l] = p; p += MT * NS * 64; W->b[l] = m ? m->b[l] : nullptr; }
    W->t[0] = nullptr;
    for (int l = 1; l < 5; ++l) { const int NS = l < 4 ? 8 : 2; W->t[l] = p; p += 4 * NS * 64; }
}
static int mask_frags16(const nvfi_mask_desc* m, float* frag, Mk16Frags* W, bool transposed, hipStream_t st) {
    mask_frag16_ptrs(frag, W, m);
    Pack16Jobs jobs; jobs.n = 0;
    for (int l = 0; l < 5; ++l) {
        Pack16Job& J = jobs.j[jobs.n++];
        J.W = m->W[l]; J.frag = const_cast<h8_t*>(W->f[l]); J.out = l < 4 ? 128 : m->mask_dim; J.in = l == 0 ? 3 : 128;
        J.MT = l < 4 ? 4 : 1; J.NS = l == 0 ? 1 : 8; J.transposed = 0;
    }
    if (transposed)
        for (int l = 1; l < 5; ++l) {
            Pack16Job& J = jobs.j[jobs.n++];
            J.W = m->W[l]; J.frag = const_cast<h8_t*>(W->t[l]); J.out = l < 4 ? 128 : m->mask_dim; J.in = 128;
            J.MT = 4; J.NS = l < 4 ? 8 : 2; J.transposed = 1;
        }
    hipLaunchKernelGGL(k_pack16, dim3(jobs.n, 4), dim3(256), 0, st, jobs);
    LAUNCHCK();
    return 0;
}

static int mask_attrs() {
    static bool done = false;
    if (done) return 0;
    HIPCK(hipFuncSetAttribute((const void*)k_maskfield_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_maskfield_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_maskfield_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_maskfield_fwd16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_maskfield_fwd16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_maskfield_bwd16, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    done = true;
    return 0;
}

extern "C" int nvfi_maskfield_workspace_bytes(const nvfi_mask_desc* m, int64_t N, int train, int64_t* bytes) {
    if (mask_check(m)) return 2;
    MkPlan P;
    plan_mask(N > 0 ? N : 1, train, nullptr, &P);
    *bytes = P.total;
    return 0;
}

extern "C" int nvfi_maskfield_fwd(const nvfi_mask_desc* m, int64_t N, const float* xyz, float* mask_out, int mode,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
    const int train = mode & NVFI_MASK_TRAIN;
    hipStream_t st = (hipStream_t)stream;
    if (mask_check(m)) return 2;
    if (N <= 0) return 0;
    if (N >= (1ll << 31) - 256) return nvfi_fail(2, "too many points for one call");
    if (mask_attrs()) return 1;
    MkPlan P;
    plan_mask(N, train, workspace, &P);
    if (P.total > workspace_bytes) return nvfi_fail(4, "workspace too small");
    const unsigned wgs = (unsigned)((N + WG_SAMPLES - 1) / WG_SAMPLES);
    if (mode & NVFI_MASK_FP16) {
        Mk16Args h; memset(&h, 0, sizeof(h));
        if (mask_frags16(m, P.frag, &h.W, train != 0, st)) return 1;
        h.mask_dim = m->mask_dim; h.N = N; h.xyz = xyz; h.out = mask_out; h.stash_f = P.stash_f; h.stash_b = P.stash_b; h.relu_mask = P.relu;
        if (train) hipLaunchKernelGGL(k_maskfield_fwd16<true>, dim3(wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, h);
        else hipLaunchKernelGGL(k_maskfield_fwd16<false>, dim3(wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, h);
        LAUNCHCK();
        return 0;
    }
    MkArgs a; memset(&a, 0, sizeof(a));
    if (mask_frags(m, P.frag, &a.W, train != 0, st)) return 1;
    a.mask_dim = m->mask_dim; a.N = N; a.xyz = xyz; a.out = mask_out; a.stash_f = P.stash_f; a.stash_b = P.stash_b; a.relu_mask = P.relu;
    if (train) hipLaunchKernelGGL(k_maskfield_fwd<true>, dim3(wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, a);
    else hipLaunchKernelGGL(k_maskfield_fwd<false>, dim3(wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, a);
    LAUNCHCK();
    return 0;
}

extern "C" int nvfi_maskfield_bwd(const nvfi_mask_desc* m, int64_t N, const float* g_mask, const nvfi_mask_grads* grads, int mode,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (mask_check(m)) return 2;
    if (N <= 0) return 0;
    if (mask_attrs()) return 1;
    MkPlan P;
    plan_mask(N, 1, workspace, &P);
    if (P.total > workspace_bytes) return nvfi_fail(4, "workspace too small (was the forward run with train != 0?)");
    MkArgs a; memset(&a, 0, sizeof(a));
    PackJobs dummy; dummy.n = 0;
    {   // fragment pointers inside the workspace the forward packed (same layout as mask_frags)
        float* p = P.frag;
        for (int l = 0; l < 5; ++l) { const int MT = l < 4 ? 4 : 1, NS = l == 0 ? 2 : 64; a.W.f[l] = p; p += MT * NS * 64; a.W.b[l] = p; p += 128; }
        for (int l = 1; l < 5; ++l) { const int NS = l < 4 ? 64 : 16; a.W.t[l] = p; p += 4 * NS * 64; }
    }
    a.mask_dim = m->mask_dim; a.N = N; a.g_out = g_mask; a.stash_f = P.stash_f; a.stash_b = P.stash_b; a.relu_mask = P.relu;
    const unsigned wgs = (unsigned)((N + WG_SAMPLES - 1) / WG_SAMPLES);
    if (mode & NVFI_MASK_FP16) {
        Mk16Args h; memset(&h, 0, sizeof(h));
        mask_frag16_ptrs(P.frag, &h.W, m);
        h.mask_dim = m->mask_dim; h.N = N; h.g_out = g_mask; h.stash_f = P.stash_f; h.stash_b = P.stash_b; h.relu_mask = P.relu;
        hipLaunchKernelGGL(k_maskfield_bwd16, dim3(wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, h);
    } else
    hipLaunchKernelGGL(k_maskfield_bwd, dim3(wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, a);
    hipLaunchKernelGGL(k_set_int, dim3(1), dim3(64), 0, st, P.count, (int)N);
    LAUNCHCK();
    // weight gradients: dW_l = g_z_l^T h_{l-1}, db_l = sum g_z_l  (split-K MFMA over the stashed tiles, engine.hip)
    WgradJobs wj; wj.n = 0; ReduceJobs rj; rj.n = 0;
    const size_t fs = MK_F_ROWS * REGF, bs = MK_B_ROWS * REGF;
    for (int l = 0; l < 5; ++l) {
        if (!grads->W[l] && !grads->b[l]) continue;
        WgradJob& J = wj.j[wj.n];
        memset(&J, 0, sizeof(J));
        // adjoint rows: l = 4 -> g_logits (16 regs); l = 3..0 -> g_z_{l+1} at 16 + 64*(3-l)
        J.A = l == 4 ? P.stash_b : P.stash_b + (size_t)(16 + 64 * (3 - l)) * REGF; J.a_regs = l == 4 ? 16 : 64; J.a_tile_stride = bs;
        // inputs: l = 0 -> the point slots (16 regs); l >= 1 -> h_l at 16 + 64*(l-1)
        J.B = l == 0 ? P.stash_f : P.stash_f + (size_t)(16 + 64 * (l - 1)) * REGF; J.b_regs = l == 0 ? 16 : 64; J.b_tile_stride = fs;
        J.bmode = BM_RAW; J.count = P.count; J.cap_tiles = (int)P.tiles; J.nrep = 1;
        J.slabs = P.slabs + (size_t)wj.n * MK_NSLAB * MK_SLAB_FLOATS; J.nslab = MK_NSLAB;
        ReduceJob& Q = rj.j[rj.n++];
        memset(&Q, 0, sizeof(Q));
        Q.slabs = J.slabs; Q.nslab = MK_NSLAB; Q.MTA = J.a_regs / 16; Q.KTB = J.b_regs / 16; Q.gW = grads->W[l]; Q.gb = grads->b[l];
        Q.out = l < 4 ? 128 : m->mask_dim; Q.in = l == 0 ? 3 : 128; Q.row_kind = RK_NATURAL; Q.slot_kind = l == 0 ? SK_XYZ : SK_HIDDEN; Q.scale = 1.f;
        ++wj.n;
    }
    return launch_wgrad(wj, rj, st);
}
