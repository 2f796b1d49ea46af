// engine.h - fp32 MFMA "sample-tile" MLP engine for gfx950 (CDNA4).
//
// Design (MI355X-first, not a translation of the reference's nn.Linear stack):
//   * a wave owns a tile of 32 samples; lanes l and l+32 (h = l>>5) hold the same sample.
//   * a Linear layer is D[out][sample] = W[out][k] * X[k][sample] on v_mfma_f32_32x32x2_f32
//     (exact fp32, bitwise an fmaf chain).  Activations never leave registers between layers:
//     the MFMA C/D layout (row = (r&3)+8(r>>2)+4h, col = lane&31) of layer l IS the B-operand
//     layout of layer l+1 once the contraction index is walked in that order, so the weights are
//     pre-permuted ("fragment order") instead: frag[(m*NS+s)*64 + lane] = W[32m + (lane&31)][slot(s,h)].
//   * fragments of one layer (<= 64 KB) are staged in LDS once per workgroup (4 waves = 128
//     samples) and read with conflict-free ds_read_b32; two workgroups per CU overlap staging,
//     activation VALU work and MFMA.
//   * weight gradients are a separate split-K MFMA kernel over stashed tiles (k_wgrad), transposing
//     sample<->feature through padded LDS images.
//
// Reference semantics implemented with this engine: VelBasis weight_net / a_weight_net
// (models/velocity_field.py:54-98), MLPRender_PE + basis_mat (models/tensorf_base.py:67-98,
// models/tensorf_keyframe.py:310).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#define WG_THREADS 256
#define TILE 32                       // samples per wave
#define WG_SAMPLES 128                // 4 waves
#define LDS_W_FLOATS 16384            // 64 KB fragment buffer
#define LDS_B_FLOATS 128
#define ENGINE_LDS_BYTES ((LDS_W_FLOATS + LDS_B_FLOATS) * 4)
#define REGF (64)                     // floats per stash register row (one per lane)
// Stash rows are written once and read back once or twice, hundreds of MB later: their stores and their loads in the adjoint kernels are
// non-temporal (`global_store_dword ... nt` / `global_load_dword ... nt`), so that they stream past L2 / MALL instead of evicting the
// planes and weight fragments (measured: k_app_fwd<stash> -8 %, k_pde_jet_bwd -5 %, step 5.065 -> 4.96 ms).
// -DNVFI_STASH_TEMPORAL keeps plain accesses (bisecting).
#ifdef NVFI_STASH_TEMPORAL
#define STASH_ST(p, v) ((p) = (v))
#define STASH_LD(p) (p)
#else
#define STASH_ST(p, v) __builtin_nontemporal_store((float)(v), &(p))
#define STASH_LD(p) __builtin_nontemporal_load(&(p))
#endif

// ---------------------------------------------------------------- layout maps
// D-layout: register index s (= 16*tile + r) and half h  ->  row index
__host__ __device__ inline int dmap(int s, int h) {
    int m = s >> 4, r = s & 15;
    return 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
}
// inverse: row -> p = 2*s + h
__host__ __device__ inline int dinv_p(int rho) {
    int m = rho >> 5, q = rho & 31;
    int h = (q >> 2) & 1;
    int r = (q & 3) | ((q >> 3) << 2);
    return 2 * (16 * m + r) + h;
}

enum { SK_HIDDEN = 0, SK_VEL_IN = 1, SK_RENDER_IN = 2, SK_XYZ = 3 };
// slot p = 2*s + h  ->  logical input feature index of the layer (or -1: unused slot)
__host__ __device__ inline int slot_logical(int kind, int p) {
    int s = p >> 1, h = p & 1;
    if (kind == SK_HIDDEN) return dmap(s, h);
    if (kind == SK_VEL_IN) {          // PositionEncoder(3) on (x,y,z,t), base_network.py:42-54
        if (s == 0) return h;         // x | y
        if (s == 1) return 2 + h;     // z | t
        if (s < 14) { int k = (s - 2) >> 2, c = (s - 2) & 3; return (h ? 8 : 4) + 8 * k + c; }  // sin | cos
        return -1;
    }
    if (kind == SK_XYZ) return p < 3 ? p : -1;   // MaskField input: (x | y), (z | -)
    // SK_RENDER_IN: [feat32 | view3 | pts3 | sin(pts)18 | cos(pts)18 | sin(view)18 | cos(view)18]
    if (s < 16) return dmap(s, h);
    if (s < 19) return (h ? 35 : 32) + (s - 16);
    if (s < 37) return (h ? 56 : 38) + (s - 19);
    if (s < 55) return (h ? 92 : 74) + (s - 37);
    return -1;
}
enum { RK_NATURAL = 0, RK_VEL_IN = 1, RK_RENDER_IN = 2 };
__host__ __device__ inline int row_logical(int kind, int rho) {
    if (kind == RK_NATURAL) return rho;
    return slot_logical(kind == RK_VEL_IN ? SK_VEL_IN : SK_RENDER_IN, dinv_p(rho));
}

// ---------------------------------------------------------------- fragment packing
struct PackJob {
    const float* W;      // (out,in) row-major
    const float* b;      // (out) or NULL
    float* frag;         // MT*NS*64 floats
    float* bfrag;        // MT*32 floats or NULL
    int out, in, MT, NS;
    int row_kind, slot_kind;
    int transposed;      // 0: rows=out features, slots=in features ; 1 (dgrad): rows=in features, slots=out features
    int x4;              // 1: write the fragment in x4 order (four consecutive K steps of a lane side by side, MT * ceil(NS/4) * 256 floats,
                         //    zero beyond NS) straight from W - what k_frag_x4 (pde_jet.hip) makes of a packed fragment
};
#define MAX_PACK_JOBS 40
struct PackJobs { PackJob j[MAX_PACK_JOBS]; int n; };
// every fragment set of a field in one launch (nvfi_pack_frags, frags.hip): render MLP 8, two velocity nets 16 + 16, their x4 copies 17
#define MAX_PACK_JOBS_ALL 60
struct PackJobsAll { PackJob j[MAX_PACK_JOBS_ALL]; int n; };

__global__ void k_pack(PackJobs jobs);

// ---------------------------------------------------------------- activations
// SiLU and its derivatives share one sigmoid built from the hardware exp2/rcp instructions (v_exp_f32, v_rcp_f32:
// ~1 ulp each; |rel err| of sigmoid <~ 2e-7 + |z|*6e-8), an order of magnitude cheaper than libm expf + IEEE divide.
// Define NVFI_ACCURATE_ACT to fall back to expf / true division when bisecting a parity question.
__device__ __forceinline__ float fast_sigmoid(float z) {
#ifdef NVFI_ACCURATE_ACT
    return 1.f / (1.f + expf(-z));
#else
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * z));
#endif
}
// sin(a) (want_cos = 0) or cos(a) (want_cos = 1) for the positional encodings.  libm sincosf carries a branch-free
// Payne-Hanek reduction (~200 instructions, and every lane needs only one of the two values); the arguments here are
// coordinates times <= 32, so a 3-constant Cody-Waite reduction by pi/2 and the degree-7/8 minimax kernels suffice:
// <= 1.6 ulp, |abs err| < 1e-7 for |a| <= 400 and <= 3.1 ulp up to 4000 (checked against float64 on 2M points per
// decade); the reduction degrades gradually beyond ~1e4, where the velocity gate / box test has long zeroed the result.
// (A libm fallback branch is not an option: the compiler if-converts it and evaluates both paths for every lane.)
__device__ __forceinline__ float trig_sel(float a, int want_cos) {
#ifdef NVFI_ACCURATE_ACT
    return want_cos ? cosf(a) : sinf(a);
#else
    const float kf = rintf(a * 0.636619772367581f);
    const int k = (int)kf + want_cos;                                 // cos(a) = sin(a + pi/2)
    float r = __builtin_fmaf(kf, -1.5707963705062866f, a);            // pi/2 = hi + mid + lo
    r = __builtin_fmaf(kf, 4.371138828673793e-08f, r);
    r = __builtin_fmaf(kf, 1.7763568394002505e-15f, r);
    const float z = r * r;
    float sp = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    sp = __builtin_fmaf(sp * z, r, r);
    float cp = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    cp = __builtin_fmaf(cp * z, z, __builtin_fmaf(z, -0.5f, 1.0f));
    const float v = (k & 1) ? cp : sp;
    return (k & 2) ? -v : v;
#endif
}
template <int ACT> __device__ __forceinline__ float act_f(float z);
template <> __device__ __forceinline__ float act_f<0>(float z) { return z > 0.f ? z : 0.f; }
template <> __device__ __forceinline__ float act_f<1>(float z) { return z * fast_sigmoid(z); }
template <int ACT> __device__ __forceinline__ float act_d1(float z);
template <> __device__ __forceinline__ float act_d1<0>(float z) { return z > 0.f ? 1.f : 0.f; }
template <> __device__ __forceinline__ float act_d1<1>(float z) {
    float s = fast_sigmoid(z);
    return s * (1.f + z * (1.f - s));
}
template <int ACT> __device__ __forceinline__ float act_d2(float z);
template <> __device__ __forceinline__ float act_d2<0>(float) { return 0.f; }
template <> __device__ __forceinline__ float act_d2<1>(float z) {
    float s = fast_sigmoid(z);
    return s * (1.f - s) * (2.f + z * (1.f - 2.f * s));
}
// first and second derivative together (one sigmoid)
template <int ACT> __device__ __forceinline__ void act_d12(float z, float& d1, float& d2) {
    if (ACT == 0) { d1 = z > 0.f ? 1.f : 0.f; d2 = 0.f; return; }
    float s = fast_sigmoid(z);
    d1 = s * (1.f + z * (1.f - s));
    d2 = s * (1.f - s) * (2.f + z * (1.f - 2.f * s));
}

// Pairs of values on the packed-fp32 VALU (v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of fp32 per instruction slot).  Same
// operations in the same order as the scalar forms above - bit-identical results - with about half the issue slots for the
// polynomial part; the two transcendentals per value stay scalar.  Epilogues of the one-wave-per-SIMD kernels are not hidden
// behind MFMAs, so their VALU count is kernel time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fast_sigmoid2(f32x2 z) {
    f32x2 s; s.x = fast_sigmoid(z.x); s.y = fast_sigmoid(z.y);
    return s;
}
template <int ACT> __device__ __forceinline__ void act_d12_2(f32x2 z, f32x2& d1, f32x2& d2) {
    if (ACT == 0) { d1.x = z.x > 0.f ? 1.f : 0.f; d1.y = z.y > 0.f ? 1.f : 0.f; d2 = (f32x2)(0.f); return; }
    const f32x2 s = fast_sigmoid2(z);
    const f32x2 oms = 1.f - s;
    d1 = s * (1.f + z * oms);
    d2 = s * oms * (2.f + z * (1.f - 2.f * s));
}
template <int ACT> __device__ __forceinline__ f32x2 act_d1_2(f32x2 z) {
    if (ACT == 0) { f32x2 d; d.x = z.x > 0.f ? 1.f : 0.f; d.y = z.y > 0.f ? 1.f : 0.f; return d; }
    const f32x2 s = fast_sigmoid2(z);
    return s * (1.f + z * (1.f - s));
}

// ---------------------------------------------------------------- LDS staging + MFMA layer
__device__ __forceinline__ void stage_frag(float* lds_w, float* lds_b, const float* __restrict__ frag, int nfloats,
                                           const float* __restrict__ bfrag, int nb) {
    const float4* src = reinterpret_cast<const float4*>(frag);
    float4* dst = reinterpret_cast<float4*>(lds_w);
#ifdef NVFI_EXP_NOSTAGE   // timing experiment only: results are garbage
    if (nfloats < 0)
#endif
    for (int i = threadIdx.x; i < (nfloats >> 2); i += WG_THREADS) dst[i] = src[i];
    if (threadIdx.x < nb) lds_b[threadIdx.x] = bfrag ? bfrag[threadIdx.x] : 0.f;
}

template <int MT>
__device__ __forceinline__ void acc_init(f32x16* acc, const float* lds_b, int h, bool use_bias) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = use_bias ? lds_b[32 * m + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
}

template <int MT, int NS>
__device__ __forceinline__ void layer_mfma(const float* lds_w, int lane, const float* x, f32x16* acc) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = MFMA32(lds_w[(m * NS + s) * 64 + lane], x[s], acc[m]);
    }
}

// Tile-major layer: the MT output tiles are produced one after the other (64 back-to-back MFMAs on one accumulator:
// the 32x32x2 f32 MFMA's dependent latency equals its issue interval), and `epi(m, acc)` - activation, stash stores,
// derivative loads - of a finished tile is independent work the scheduler overlaps with the next tile's MFMAs.
// Only 16 accumulator registers are live, which pays for a second activation array (ping-pong x -> xn).
// `defer` (optional): the last tile's accumulator is handed back instead of being passed to `epi` - the caller runs that
// epilogue AFTER it has issued the next layer's staging loads.  (vmcnt counts stores too on gfx9/CDNA: staging loads issued
// behind the last tile's 16 stash stores waited for their L2 acknowledgement - 28 % of the wave cycles of the training forward.)
template <int MT, int NS, class Epi>
__device__ __forceinline__ void layer_tiles(const float* lds_w, const float* lds_b, bool bias, int lane, int h, const float* x, Epi epi,
                                            f32x16* defer = nullptr) {
    f32x16 prev;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias ? lds_b[32 * m + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
        // region: MFMAs of tile m  +  epilogue of tile m-1 (independent: the scheduler interleaves them);
        // the sched_barrier keeps the epilogue loads/temporaries of later tiles from being hoisted (register pressure)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            acc = MFMA32(lds_w[(m * NS + s) * 64 + lane], x[s], acc);
        }
        if (m > 0) {
            epi(m - 1, prev);
            // ask for the interleave explicitly: per MFMA one LDS operand read and a few of the epilogue's VALU / store ops
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read (next A operand)
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU (epilogue of the previous tile)
                if ((s & 3) == 0) __builtin_amdgcn_sched_group_barrier(0x030, 1, 0);   // one VMEM op (stash store / z load)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        prev = acc;
    }
    if (defer) *defer = prev; else epi(MT - 1, prev);
}

// store / load a block of NR stash registers (one 256-B row per register)
// x4 stash blocks: rows 4k .. 4k+3 of a 16-row group as ONE 16-byte access per lane - element (row R, lane L) at (R >> 2) * 256 + 4 L + (R & 3)
// instead of R * 64 + L, i.e. a permutation inside the 1 KiB the four row-major rows occupy: a quarter of the VMEM instructions of a stash
// writer / reader.  Only for rows whose EVERY reader knows the layout (the hidden layers' pre-activations once the fused adjoint kernels are
// their only consumers: the ring kernel's DMA wants row-major rows).
typedef float f32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stash_st16_x4(float* base16, int lane, const f32x16& v) {
    f32x4s* p = reinterpret_cast<f32x4s*>(base16) + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4s q = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
#ifdef NVFI_STASH_TEMPORAL
        p[k * 64] = q;
#else
        __builtin_nontemporal_store(q, p + k * 64);
#endif
    }
}
__device__ __forceinline__ void stash_st16_x4(float* base16, int lane, const float (&v)[16]) {
    f32x4s* p = reinterpret_cast<f32x4s*>(base16) + lane;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4s q = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
#ifdef NVFI_STASH_TEMPORAL
        p[k * 64] = q;
#else
        __builtin_nontemporal_store(q, p + k * 64);
#endif
    }
}
// rows 4 k .. 4 k + 3 of an x4 block group for this lane (the reader's side of stash_st16_x4)
__device__ __forceinline__ void stash_ld4_x4(const float* base16, int lane, int k, float* v4) {
    const f32x4s* p = reinterpret_cast<const f32x4s*>(base16) + lane + k * 64;
#ifdef NVFI_STASH_TEMPORAL
    const f32x4s q = *p;
#else
    const f32x4s q = __builtin_nontemporal_load(p);
#endif
    v4[0] = q[0]; v4[1] = q[1]; v4[2] = q[2]; v4[3] = q[3];
}
template <int NR>
__device__ __forceinline__ void stash_store(float* base, int lane, const float* v) {
#pragma unroll
    for (int s = 0; s < NR; ++s) STASH_ST(base[s * REGF + lane], v[s]);
}
template <int MT>
__device__ __forceinline__ void stash_store_acc(float* base, int lane, const f32x16* acc) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) STASH_ST(base[(16 * m + r) * REGF + lane], acc[m][r]);
}

// ReLU masks of a hidden layer for the backward pass: bit s of the lane's pair of words = (activation register s > 0).  The adjoint of a
// ReLU layer needs only the signs - 1 KB per tile and layer instead of the 32 KB of activation rows (which stay: the weight gradient contracts them)
__device__ __forceinline__ void relu_mask_store(unsigned* dst, int lane, const float* x) {
    unsigned lo = 0u, hi = 0u;
#pragma unroll
    for (int s = 0; s < 32; ++s) { lo |= x[s] > 0.f ? 1u << s : 0u; hi |= x[32 + s] > 0.f ? 1u << s : 0u; }
    dst[lane] = lo; dst[64 + lane] = hi;
}
// g[s] = bit s of the mask ? acc[s] : 0   (acc in four 16-register tiles)
__device__ __forceinline__ void relu_mask_apply(const unsigned* mk, int lane, const f32x16* acc, float* g) {
    const unsigned mlo = mk[lane], mhi = mk[64 + lane];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[16 * m + r] = (((m < 2 ? mlo : mhi) >> ((16 * m + r) & 31)) & 1u) ? acc[m][r] : 0.f;
}

// ---------------------------------------------------------------- velocity-basis nets
// fragments of one 6-layer net (28-128-128-128-128-128-6), forward and transposed (dgrad)
struct VelFrags {
    const float* f[6];   // forward fragments: L0 MT4 NS14 | L1-4 MT4 NS64 | L5 MT1 NS64
    const float* b[6];   // bias fragments (MT*32)
    const float* t[6];   // transposed: T0 MT1 NS64 (rows = input slots) | T1-4 MT4 NS64 | T5 MT4 NS4
};
#define VEL_F0 (4 * 14 * 64)
#define VEL_FH (4 * 64 * 64)
#define VEL_F5 (1 * 64 * 64)
#define VEL_T0 (1 * 64 * 64)
#define VEL_T5 (4 * 4 * 64)
#define VEL_FRAG_FLOATS (VEL_F0 + 4 * VEL_FH + VEL_F5 + 6 * 128 + VEL_T0 + 4 * VEL_FH + VEL_T5)

// stash geometry for one (eval, tile): z: 5 layers x 64 regs, x0: 16 regs
#define VEL_Z_REGS (5 * 64)
#define VEL_X0_REGS 16

__device__ __forceinline__ float comp4(const float4& q, int c) { return c == 0 ? q.x : (c == 1 ? q.y : (c == 2 ? q.z : q.w)); }

// PositionEncoder slots for this lane (x[0..13]); regs 14,15 zero
__device__ __forceinline__ void vel_encode_slots(const float4& q, int h, float* x) {
    x[0] = h ? q.y : q.x;
    x[1] = h ? q.w : q.z;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            x[2 + 4 * k + c] = trig_sel(comp4(q, c) * (float)(1 << k), h);
        }
    x[14] = 0.f; x[15] = 0.f;
}

// One forward evaluation of a VelBasis weight net for the wave's 32 samples.
// Must be called by all 4 waves of the workgroup (contains barriers).
// out4: lane h=0 -> w0..w3 ; lane h=1 -> w4,w5,0,0.
// zst: this (eval,tile)'s z stash (VEL_Z_REGS rows) or NULL; x0st: x0 stash (16 rows) or NULL.
// DEFER: run each layer's last-tile epilogue (its stash stores) behind the next layer's staging loads - pays only when a stash
// is written (the 16 accumulator registers it keeps live cost spills in the stash-less kernels).
template <int ACT, bool DEFER = false>
__device__ __forceinline__ void velnet_forward(const VelFrags& W, float* lds_w, float* lds_b, int lane,
                                               const float4& q, float* zst, float* x0st, float* out4) {
    const int h = lane >> 5;
    float xa[64], xb[64];
    f32x16 last;
    f32x16* const dp = DEFER ? &last : nullptr;
    vel_encode_slots(q, h, xb);
    if (x0st) stash_store<16>(x0st, lane, xb);
    __syncthreads();
    stage_frag(lds_w, lds_b, W.f[0], VEL_F0, W.b[0], 128);
    __syncthreads();
    auto epi0 = [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (zst) zst[(16 * m + r) * REGF + lane] = acc[r];
            xa[16 * m + r] = act_f<ACT>(acc[r]);
        }
    };
    layer_tiles<4, 14>(lds_w, lds_b, true, lane, h, xb, epi0, dp);
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        const int l = 1 + 2 * it;
        float* z0 = zst ? zst + (size_t)(l - 1) * 64 * REGF : nullptr;
        float* z1 = zst ? zst + (size_t)l * 64 * REGF : nullptr;
        float* z2 = zst ? zst + (size_t)(l + 1) * 64 * REGF : nullptr;
        __syncthreads();
        stage_frag(lds_w, lds_b, W.f[l], VEL_FH, W.b[l], 128);
        if (DEFER) {   // deferred epilogue of the previous layer's last tile (layer l-1 wrote xa)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (z0) z0[(48 + r) * REGF + lane] = last[r];
                xa[48 + r] = act_f<ACT>(last[r]);
            }
        }
        __syncthreads();
        layer_tiles<4, 64>(lds_w, lds_b, true, lane, h, xa, [&](int m, const f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (z1) z1[(16 * m + r) * REGF + lane] = acc[r];
                xb[16 * m + r] = act_f<ACT>(acc[r]);
            }
        }, dp);
        __syncthreads();
        stage_frag(lds_w, lds_b, W.f[l + 1], VEL_FH, W.b[l + 1], 128);
        if (DEFER) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (z1) z1[(48 + r) * REGF + lane] = last[r];
                xb[48 + r] = act_f<ACT>(last[r]);
            }
        }
        __syncthreads();
        layer_tiles<4, 64>(lds_w, lds_b, true, lane, h, xb, [&](int m, const f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (z2) z2[(16 * m + r) * REGF + lane] = acc[r];
                xa[16 * m + r] = act_f<ACT>(acc[r]);
            }
        }, dp);
    }
    float* z4 = zst ? zst + (size_t)4 * 64 * REGF : nullptr;
    __syncthreads();
    stage_frag(lds_w, lds_b, W.f[5], VEL_F5, W.b[5], 32);
    if (DEFER) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (z4) z4[(48 + r) * REGF + lane] = last[r];
            xa[48 + r] = act_f<ACT>(last[r]);
        }
    }
    __syncthreads();
    layer_tiles<1, 64>(lds_w, lds_b, true, lane, h, xa, [&](int, const f32x16& acc) {
        out4[0] = acc[0]; out4[1] = acc[1]; out4[2] = acc[2]; out4[3] = acc[3];
    });
}

// both halves get the full 6-vector
__device__ __forceinline__ void gather6(const float* out4, int h, float* w) {
    float p0 = __shfl_xor(out4[0], 32), p1 = __shfl_xor(out4[1], 32), p2 = __shfl_xor(out4[2], 32), p3 = __shfl_xor(out4[3], 32);
    if (h == 0) { w[0] = out4[0]; w[1] = out4[1]; w[2] = out4[2]; w[3] = out4[3]; w[4] = p0; w[5] = p1; }
    else        { w[0] = p0; w[1] = p1; w[2] = p2; w[3] = p3; w[4] = out4[0]; w[5] = out4[1]; }
}
// inverse: 6-vector (same in both halves) -> the D-layout tile-0 registers 0..3 of this lane
__device__ __forceinline__ void scatter6(const float* w, int h, float* r4) {
    if (h == 0) { r4[0] = w[0]; r4[1] = w[1]; r4[2] = w[2]; r4[3] = w[3]; }
    else        { r4[0] = w[4]; r4[1] = w[5]; r4[2] = 0.f; r4[3] = 0.f; }
}

// v = sum_i w_i b_i(x)  (velocity_field.py:77-93)
__device__ __forceinline__ void vel_from_w(const float* w, float x, float y, float z, float* v) {
    v[0] = w[0] - w[4] * z + w[5] * y;
    v[1] = w[1] + w[3] * z - w[5] * x;
    v[2] = w[2] - w[3] * y + w[4] * x;
}
__device__ __forceinline__ void acc_from_w(const float* aw, float x, float y, float z, float* a) {
    a[0] = aw[0] - aw[4] * x - aw[5] * x;
    a[1] = aw[1] - aw[3] * y - aw[5] * y;
    a[2] = aw[2] - aw[3] * z - aw[4] * z;
}

// (The adjoint of one evaluation lives with its kernels: vel_split.hip - feature-split, unfused - and vel_fuse.hip - the persistent kernel with the weight
// gradients; the one-tile-per-wave forms of rounds 1-2, velnet_backward / velnet_backward_p on a double-buffered fragment pipe, were retired in round 6.)

// adjoint of the PositionEncoder slots: ge (16 regs of this lane) + the forward slots x0 -> (gx,gy,gz,gt) summed over both halves
__device__ __forceinline__ float4 vel_encode_bwd(const float* ge, const float* x0, int h) {
    // lane h=0 holds (x | z) raw and the sin slots; lane h=1 holds (y | t) raw and the cos slots.
    float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float mine = x0[2 + 4 * k + c];               // sin (h=0) or cos (h=1)
            float other = __shfl_xor(mine, 32);           // cos (h=0) or sin (h=1)
            float fr = (float)(1 << k);
            // d sin = fr*cos ; d cos = -fr*sin
            g[c] += (h ? -fr * other : fr * other) * ge[2 + 4 * k + c];
        }
    float r0 = ge[0], r1 = ge[1];
    // h=0: r0 -> x, r1 -> z ; h=1: r0 -> y, r1 -> t
    float gx = g[0] + (h ? 0.f : r0), gy = g[1] + (h ? r0 : 0.f), gz = g[2] + (h ? 0.f : r1), gt = g[3] + (h ? r1 : 0.f);
    gx += __shfl_xor(gx, 32); gy += __shfl_xor(gy, 32); gz += __shfl_xor(gz, 32); gt += __shfl_xor(gt, 32);
    return make_float4(gx, gy, gz, gt);
}

// ---------------------------------------------------------------- weight-gradient kernel (split-K over tiles)
// G[pA][pB] = sum_tiles sum_j A[pA][j] * B[pB][j]   (p-space: p = 2*reg + h)
enum { BM_RAW = 0, BM_SILU = 1, BM_RELU = 2, BM_SILU_TAN = 3, BM_RELU_TAN = 4 };
struct WgradJob {
    const float* A; size_t a_tile_stride; int a_regs;      // rows of 64 floats; a_regs multiple of 16
    const float* B; const float* B2; size_t b_tile_stride; int b_regs; int bmode;
    const int* count; int cap_tiles;  // number of valid samples (device), tile capacity
    int nrep; size_t a_rep_stride, b_rep_stride, b2_rep_stride;   // replicate over e.g. (step,eval) blocks
    float* slabs;                      // [nslab][ (32*MTA) * (32*KTB) + 32*MTA ]
    int nslab;
};
#define MAX_WGRAD_JOBS 24
struct WgradJobs { WgradJob j[MAX_WGRAD_JOBS]; int n; };

// slab reduce + un-permute into the logical gradient tensors
struct ReduceJob {
    const float* slabs; int nslab; int MTA, KTB;
    const float* slabs2; int nslab2;   // optional second slab set reduced into the same target
    float* gW; float* gb; int out, in;
    int row_kind, slot_kind;     // maps of the forward layer: rows of G = out rows (p-space of D layout), cols = input slots
    float scale;
};
struct ReduceJobs { ReduceJob j[MAX_WGRAD_JOBS]; int n; };
__global__ void k_wgrad_reduce(ReduceJobs jobs);

// ---------------------------------------------------------------- helpers
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
