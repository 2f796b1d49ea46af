// vel_split.hip - per-point RK2 back-advection (reference models/tensorf_keyframe.py:575-611 around the gated VelBasis of
// models/velocity_field.py:21-98) with ONE 32-point tile per workgroup and the network's FEATURES split over the four waves.
//
// k_rk2_fwd (vel.hip) gives every wave its own tile: best throughput per staged weight byte, but a tile's latency is the whole
// network on one SIMD (~100 k cycles per evaluation), which is what a SHORT list of points pays however few they are - the fp32
// re-evaluation list behind the fp16 pre-pass of the PDE prefilter (pre16.hip) is such a list.  Here wave w owns output rows
// [32w, 32w + 32) of every hidden layer (one MFMA tile, the x4 weight fragments of pde_jet.hip read straight from L2 into
// registers one layer ahead), the four 32-row slices meet in a 16 KB LDS exchange buffer between layers, and the 128 -> 6 output
// layer is contracted by one wave (rotating with the workgroup index) and broadcast through LDS.  A tile's latency drops ~3.5x.
//
// Every accumulator sees the same operands in the same K order as in engine.h's layer_tiles, and every wave carries the same
// replicated RK2 state, so the result is bit-identical to k_rk2_fwd<false, false>.
#include <stdlib.h>
#include "common.h"
#include "vel.h"
#include "pde.h"

#define SPLIT_XCH_F4 (16 * 64)              // [s/4][lane] float4: one layer's 128 features x 32 points
#define SPLIT_LDS_BYTES(NT) ((NT) * (SPLIT_XCH_F4 * 16 + 4 * 64 * 4) + 6 * 128 * 4)

template <int NS4>
__device__ __forceinline__ void split_load(const float4* __restrict__ a4, int lane, float4* wq) {
#pragma unroll
    for (int g = 0; g < NS4; ++g) wq[g] = a4[g * 64 + lane];
}
template <int NS4, int NT, int XS>
__device__ __forceinline__ void split_mfma(const float4* wq, const float (&x)[NT][XS], f32x16* acc) {
#pragma unroll
    for (int g = 0; g < NS4; ++g) {
        const float av[4] = {wq[g].x, wq[g].y, wq[g].z, wq[g].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = MFMA32(av[k], x[t][4 * g + k], acc[t]);
    }
}

// the same with the B operands streamed from the exchange buffer (one 16-byte LDS read per tile and 4 K-steps): no register copy of
// the layer input, so four tiles fit a wave and every 16-byte weight load feeds 16 MFMAs
template <int NT>
__device__ __forceinline__ void split_mfma_lds(const float4* wq, const float4* xl, int tile0, f32x16* acc) {
    float4 b[NT], bn[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[t] = xl[((tile0 + t) * 16) * 64];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g + 1 < 16) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[t] = xl[((tile0 + t) * 16 + g + 1) * 64];
        }
        const float av[4] = {wq[g].x, wq[g].y, wq[g].z, wq[g].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float bv = k == 0 ? b[t].x : (k == 1 ? b[t].y : (k == 2 ? b[t].z : b[t].w));
                acc[t] = MFMA32(av[k], bv, acc[t]);
            }
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = bn[t];
    }
}

// one gated-velocity network evaluation of the workgroup's NT tiles; all four waves return the same out4 (lane h=0: w0..w3,
// h=1: w4, w5) per tile
// STASH (training render): pre-activations z (5 layers x 64 rows, wave w = rows 16w..16w+15 of each layer) and the encoder slots go to
// the per-(evaluation, tile) stash in exactly the layout of velnet_forward (engine.h), so k_rk2_bwd / k_wgrad read it unchanged
template <int NT, bool STASH = false>
__device__ __forceinline__ void velnet_split(const float4* const* f4, float4* xch, float* bc, int w, int owner, int lane, int h,
                                             const float4* q, float4* wq, const float* lb, float (&out4)[NT][4],
                                             float* const* zst = nullptr, float* const* x0st = nullptr) {
    f32x16 acc[NT];
    const float4* xl = xch + lane;
    {
        float in0[NT][16];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            vel_encode_slots(q[t], h, in0[t]);
            if (STASH && w == t) stash_store<16>(x0st[t], lane, in0[t]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = lb[32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
        }
        split_mfma<4, NT, 16>(wq, in0, acc);             // wq holds layer 0 (loaded by the caller / the previous evaluation)
    }
    const int mine = (w - owner) & 3;                    // the tile whose 128 -> 6 output layer this wave contracts (if < NT)
#pragma unroll 1
    for (int l = 0; l < 5; ++l) {
        // the next layer's weights start their trip from L2 now; they land behind the epilogue and the exchange
        if (l < 4) split_load<16>(f4[l + 1] + (size_t)w * 16 * 64, lane, wq);
        else if (mine < NT) split_load<16>(f4[5], lane, wq);
        if (STASH) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) STASH_ST(zst[t][(size_t)(l * 64 + 16 * w + r) * REGF + lane], acc[t][r]);
        }
        __syncthreads();                                 // the previous layer's readers of the exchange buffer are done
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                xch[(t * 16 + 4 * w + k) * 64 + lane] = make_float4(act_f<1>(acc[t][4 * k]), act_f<1>(acc[t][4 * k + 1]), act_f<1>(acc[t][4 * k + 2]),
                                                                    act_f<1>(acc[t][4 * k + 3]));
        __syncthreads();
        if (l < 4) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = lb[128 * (l + 1) + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
            split_mfma_lds<NT>(wq, xl, 0, acc);
        }
    }
    // 128 -> 6 output layer: tile t is contracted by wave (owner + t) & 3 alone (one accumulator, K in layer_tiles' order)
    if (mine < NT) {
        f32x16 ao[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) ao[0][r] = lb[128 * 5 + (r & 3) + 8 * (r >> 2) + 4 * h];
        split_mfma_lds<1>(wq, xl, mine, ao);
#pragma unroll
        for (int r = 0; r < 4; ++r) bc[(mine * 4 + r) * 64 + lane] = ao[0][r];
    }
    // layer 0 of the NEXT evaluation (the caller stops using wq before that)
    split_load<4>(f4[0] + (size_t)w * 4 * 64, lane, wq);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) out4[t][r] = bc[(t * 4 + r) * 64 + lane];
}

// ---------------------------------------------------------------- the same evaluation with the 128 -> 6 OUTPUT layer on the vector pipe
// On the matrix pipe the output layer is 64 MFMAs of a 32-row tile that holds 6 useful rows, issued by one wave per tile while the
// other waves of the workgroup wait at the barrier behind it: 10 % of an evaluation's critical path for 0.5 % of its arithmetic.
// Here every lane multiplies the 16 activations of the last hidden layer it already holds in registers (its point i, features
// 32 w + (r & 3) + 8 (r >> 2) + 4 h) with the matching 16 x 6 output weights (LDS image [w][h][r][8], 4 KB), the two halves meet by
// one cross-lane add, the four waves' partial sums by one trip through LDS (8 KB, added in wave order by every wave: replicated state
// stays bit-identical across the workgroup), and the last hidden layer's activations never go to the exchange buffer.
// fp32 FMAs in a fixed order instead of the MFMA's products and sums: the same arithmetic, another rounding (1e-7 relative).
#define SPLIT_VOUT_LDS_BYTES(NT) ((NT) * (SPLIT_XCH_F4 * 16 + 4 * 2 * 32 * 16) + 6 * 128 * 4 + 4 * 2 * 16 * 8 * 4)
template <int NT, bool STASH = false>
__device__ __forceinline__ void velnet_split_vout(const float4* const* f4, float4* xch, float4* part, const float4* w5l, int w, int lane, int h,
                                                  const float4* q, float4* wq, const float* lb, float (&out6)[NT][6],
                                                  float* const* zst = nullptr, float* const* x0st = nullptr) {
    f32x16 acc[NT];
    const float4* xl = xch + lane;
    {
        float in0[NT][16];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            vel_encode_slots(q[t], h, in0[t]);
            if (STASH && w == t) stash_store<16>(x0st[t], lane, in0[t]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = lb[32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
        }
        split_mfma<4, NT, 16>(wq, in0, acc);             // wq holds layer 0 (loaded by the caller / the previous evaluation)
    }
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        split_load<16>(f4[l + 1] + (size_t)w * 16 * 64, lane, wq);
        if (STASH) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) STASH_ST(zst[t][(size_t)(l * 64 + 16 * w + r) * REGF + lane], acc[t][r]);
        }
        __syncthreads();                                 // the previous layer's readers of the exchange buffer are done
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                xch[(t * 16 + 4 * w + k) * 64 + lane] = make_float4(act_f<1>(acc[t][4 * k]), act_f<1>(acc[t][4 * k + 1]), act_f<1>(acc[t][4 * k + 2]),
                                                                    act_f<1>(acc[t][4 * k + 3]));
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = lb[128 * (l + 1) + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
        split_mfma_lds<NT>(wq, xl, 0, acc);
    }
    // layer 0 of the NEXT evaluation (the caller stops using wq before that)
    split_load<4>(f4[0] + (size_t)w * 4 * 64, lane, wq);
    if (STASH) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) STASH_ST(zst[t][(size_t)(4 * 64 + 16 * w + r) * REGF + lane], acc[t][r]);
    }
    float p[NT][6];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int o = 0; o < 6; ++o) p[t][o] = 0.f;
    const float4* wl = w5l + (w * 2 + h) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if ((r & 1) == 0) __builtin_amdgcn_sched_barrier(0);     // weight reads in groups of two rows (all 32 at once cost 128 registers)
        const float4 wa = wl[2 * r], wb = wl[2 * r + 1];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float av = act_f<1>(acc[t][r]);
            p[t][0] = __builtin_fmaf(av, wa.x, p[t][0]); p[t][1] = __builtin_fmaf(av, wa.y, p[t][1]); p[t][2] = __builtin_fmaf(av, wa.z, p[t][2]);
            p[t][3] = __builtin_fmaf(av, wa.w, p[t][3]); p[t][4] = __builtin_fmaf(av, wb.x, p[t][4]); p[t][5] = __builtin_fmaf(av, wb.y, p[t][5]);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int o = 0; o < 6; ++o) p[t][o] += __shfl_xor(p[t][o], 32);
        if (h == 0) {
            part[((t * 4 + w) * 2 + 0) * 32 + lane] = make_float4(p[t][0], p[t][1], p[t][2], p[t][3]);
            part[((t * 4 + w) * 2 + 1) * 32 + lane] = make_float4(p[t][4], p[t][5], 0.f, 0.f);
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int o = 0; o < 6; ++o) out6[t][o] = lb[128 * 5 + o];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            __builtin_amdgcn_sched_barrier(0);                        // (all sixteen reads of a two-tile workgroup at once cost 64 registers)
            const float4 A = part[((t * 4 + ww) * 2 + 0) * 32 + (lane & 31)], B = part[((t * 4 + ww) * 2 + 1) * 32 + (lane & 31)];
            out6[t][0] += A.x; out6[t][1] += A.y; out6[t][2] += A.z; out6[t][3] += A.w; out6[t][4] += B.x; out6[t][5] += B.y;
        }
    }
}

template <int NT, bool VOUT, bool STASH = false>
__device__ __forceinline__ void velnet_any(const float4* const* f4, float4* xch, float4* part, float* bc, const float4* w5l, int w, int owner, int lane,
                                           int h, const float4* q, float4* wq, const float* lb, float (&out6)[NT][6],
                                           float* const* zst = nullptr, float* const* x0st = nullptr) {
    if (VOUT) velnet_split_vout<NT, STASH>(f4, xch, part, w5l, w, lane, h, q, wq, lb, out6, zst, x0st);
    else {
        float o4[NT][4];
        velnet_split<NT, STASH>(f4, xch, bc, w, owner, lane, h, q, wq, lb, o4, zst, x0st);
#pragma unroll
        for (int t = 0; t < NT; ++t) gather6(o4[t], h, out6[t]);
    }
}

// VOUT: the output layer on the vector pipe (default); false: on the matrix pipe like every other layer (NVFI_SPLIT_VOUT=0: bit-identical
// to k_rk2_fwd<false, false> of vel.hip)
#ifndef SPLIT_WG_PER_CU_1
#define SPLIT_WG_PER_CU_1 4     // one-tile workgroups: 126 registers, four per CU (three: prefilter 1.31 instead of 1.29 ms)
#endif
template <int NT, bool VOUT>
__global__ __launch_bounds__(WG_THREADS, NT == 1 ? SPLIT_WG_PER_CU_1 : (NT == 2 ? 3 : 2)) void k_rk2_split(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* xch = reinterpret_cast<float4*>(lds);
    float4* part = xch + NT * SPLIT_XCH_F4;               // VOUT: [tile][wave][2][32 points] partial output sums; else the 4 x 64 broadcast rows
    float* bc = reinterpret_cast<float*>(part);
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int owner = blockIdx.x & 3;
    const int count = a.count ? *a.count : (int)a.n_direct;
    if ((int)blockIdx.x * NT * TILE >= count) return;
    bool active[NT]; int n[NT]; float x[NT], y[NT], z[NT], zw[NT], tcur[NT], off[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int i = (blockIdx.x * NT + t) * TILE + (lane & 31);
        active[t] = i < count;
        n[t] = active[t] ? (a.list ? a.list[i] : i) : 0;
        const float4 q0 = active[t] ? a.xw[n[t]] : zero4();
        x[t] = q0.x; y[t] = q0.y; z[t] = q0.z; zw[t] = q0.w;
        const int ti = a.pt_by_list ? n[t] : i;
        tcur[t] = active[t] ? a.pt_t[ti] : 0.f;
        off[t] = active[t] ? tcur[t] - a.pt_base[ti] : 0.f;
    }
    float4 wq[16];
    split_load<4>(a.f4[0] + (size_t)w * 4 * 64, lane, wq);
    // the six bias vectors live in LDS for the whole kernel (3 KB; rows beyond a layer's width read as the fragment's zero padding)
    float* lb = VOUT ? reinterpret_cast<float*>(part + NT * 4 * 2 * 32) : bc + NT * 4 * 64;
    for (int k = threadIdx.x; k < 6 * 128; k += WG_THREADS) lb[k] = (k & 127) < (k < 640 ? 128 : 32) ? a.bv[k >> 7][k & 127] : 0.f;
    // ... and (VOUT) the output layer's weights in the order the lanes hold the last hidden layer: [w][h][r][8] (6 outputs + 2 zeros)
    float* w5f = lb + 6 * 128;
    if (VOUT)
        for (int k = threadIdx.x; k < 4 * 2 * 16 * 8; k += WG_THREADS) {
            const int o = k & 7, r = (k >> 3) & 15, hh = (k >> 7) & 1, ww = k >> 8;
            w5f[k] = o < 6 ? a.f.vW[5][o * 128 + 32 * ww + (r & 3) + 8 * (r >> 2) + 4 * hh] : 0.f;
        }
    const float4* w5l = reinterpret_cast<const float4*>(w5f);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < a.max_steps; ++s) {
        bool live[NT], any = false;
        float dt[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            live[t] = active[t] && fabsf(off[t]) > 0.f;
            any = any || live[t];
            const float m = fminf(fabsf(off[t]), a.dt_max);
            dt[t] = off[t] > 0.f ? m : (off[t] < 0.f ? -m : 0.f);
        }
        if (!__any(any)) break;                           // the same decision in all four waves (replicated state)
        float o6[NT][6], px[NT], py[NT], pz[NT];
        float4 q[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) q[t] = make_float4(x[t], y[t], z[t], tcur[t]);
        velnet_any<NT, VOUT>(a.f4, xch, part, bc, w5l, w, owner, lane, h, q, wq, lb, o6);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v1[3];
            const float* w1 = o6[t];
            vel_from_w(w1, x[t], y[t], z[t], v1);
            if (gated_out(a.f, x[t], y[t], z[t])) { v1[0] = v1[1] = v1[2] = 0.f; }
            const float hdt = 0.5f * dt[t];
            px[t] = x[t] - hdt * v1[0]; py[t] = y[t] - hdt * v1[1]; pz[t] = z[t] - hdt * v1[2];
            q[t] = make_float4(px[t], py[t], pz[t], tcur[t] - hdt);
        }
        velnet_any<NT, VOUT>(a.f4, xch, part, bc, w5l, w, owner, lane, h, q, wq, lb, o6);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v2[3];
            const float* w2 = o6[t];
            vel_from_w(w2, px[t], py[t], pz[t], v2);
            if (gated_out(a.f, px[t], py[t], pz[t])) { v2[0] = v2[1] = v2[2] = 0.f; }
            const float nx = x[t] - dt[t] * v2[0], ny = y[t] - dt[t] * v2[1], nz = z[t] - dt[t] * v2[2];
            const bool rej = a.f.gate_sur && gated_out(a.f, nx, ny, nz);   // tensorf_keyframe.py:603-605
            if (live[t] && !rej) { x[t] = nx; y[t] = ny; z[t] = nz; }
            if (live[t]) { off[t] = off[t] - dt[t]; tcur[t] = tcur[t] - dt[t]; }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (active[t] && h == 0 && w == 0) a.xw[n[t]] = make_float4(x[t], y[t], z[t], zw[t]);
}

// ---------------------------------------------------------------- render warp: every sample takes the same (dt_s, t_s) sequence
// (k_rk2_fwd<true, STASH> of vel.hip on the feature-split layout; same stash, same records, same numbers)
template <int NT, bool STASH, bool VOUT>
__device__ __forceinline__ void rk2_split_uni_body(const SplitUniArgs& a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* xch = reinterpret_cast<float4*>(lds);
    float4* part = xch + NT * SPLIT_XCH_F4;
    float* bc = reinterpret_cast<float*>(part);
    const Rk2Args& ra = a.r;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int owner = blockIdx.x & 3;
    const int count = *ra.count;
    // whole 128-sample groups, as k_rk2_fwd: the adjoint and weight-gradient kernels walk every tile of the last, ragged group
    if ((int)blockIdx.x * NT * TILE >= (count + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES) return;
    bool active[NT]; int n[NT], idx[NT]; float x[NT], y[NT], z[NT], zw[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        idx[t] = (blockIdx.x * NT + t) * TILE + (lane & 31);
        active[t] = idx[t] < count;
        n[t] = active[t] ? ra.list[idx[t]] : 0;
        const float4 q0 = active[t] ? ra.xw[n[t]] : zero4();
        x[t] = q0.x; y[t] = q0.y; z[t] = q0.z; zw[t] = q0.w;
    }
    float4 wq[16];
    split_load<4>(a.f4[0] + (size_t)w * 4 * 64, lane, wq);
    float* lb = VOUT ? reinterpret_cast<float*>(part + NT * 4 * 2 * 32) : bc + NT * 4 * 64;
    for (int k = threadIdx.x; k < 6 * 128; k += WG_THREADS) lb[k] = (k & 127) < (k < 640 ? 128 : 32) ? a.bv[k >> 7][k & 127] : 0.f;
    float* w5f = lb + 6 * 128;
    if (VOUT)
        for (int k = threadIdx.x; k < 4 * 2 * 16 * 8; k += WG_THREADS) {
            const int o = k & 7, r = (k >> 3) & 15, hh = (k >> 7) & 1, ww = k >> 8;
            w5f[k] = o < 6 ? ra.f.vW[5][o * 128 + 32 * ww + (r & 3) + 8 * (r >> 2) + 4 * hh] : 0.f;
        }
    const float4* w5l = reinterpret_cast<const float4*>(w5f);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < ra.nsteps; ++s) {
        const float dt = RK_DT(ra, s), tcur = RK_TC(ra, s), hdt = 0.5f * dt;
        float* z1[NT]; float* z2[NT]; float* x1[NT]; float* x2[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t tile = (size_t)blockIdx.x * NT + t;
            const size_t e1 = (size_t)(2 * s) * ra.cap_tiles + tile, e2 = (size_t)(2 * s + 1) * ra.cap_tiles + tile;
            z1[t] = STASH ? ra.zst + e1 * (VEL_Z_REGS * REGF) : nullptr; z2[t] = STASH ? ra.zst + e2 * (VEL_Z_REGS * REGF) : nullptr;
            x1[t] = STASH ? ra.x0st + e1 * (VEL_X0_REGS * REGF) : nullptr; x2[t] = STASH ? ra.x0st + e2 * (VEL_X0_REGS * REGF) : nullptr;
        }
        float o6[NT][6], px[NT], py[NT], pz[NT], w1[NT][6];
        bool g1[NT];
        float4 q[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) q[t] = make_float4(x[t], y[t], z[t], tcur);
        velnet_any<NT, VOUT, STASH>(a.f4, xch, part, bc, w5l, w, owner, lane, h, q, wq, lb, o6, z1, x1);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v1[3];
#pragma unroll
            for (int k = 0; k < 6; ++k) w1[t][k] = o6[t][k];
            vel_from_w(w1[t], x[t], y[t], z[t], v1);
            g1[t] = gated_out(ra.f, x[t], y[t], z[t]);
            if (g1[t]) { v1[0] = v1[1] = v1[2] = 0.f; }
            px[t] = x[t] - hdt * v1[0]; py[t] = y[t] - hdt * v1[1]; pz[t] = z[t] - hdt * v1[2];
            q[t] = make_float4(px[t], py[t], pz[t], tcur - hdt);
        }
        velnet_any<NT, VOUT, STASH>(a.f4, xch, part, bc, w5l, w, owner, lane, h, q, wq, lb, o6, z2, x2);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v2[3];
            const float* w2 = o6[t];
            vel_from_w(w2, px[t], py[t], pz[t], v2);
            const bool g2 = gated_out(ra.f, px[t], py[t], pz[t]);
            if (g2) { v2[0] = v2[1] = v2[2] = 0.f; }
            const float nx = x[t] - dt * v2[0], ny = y[t] - dt * v2[1], nz = z[t] - dt * v2[2];
            const bool rej = ra.f.gate_sur && gated_out(ra.f, nx, ny, nz);   // tensorf_keyframe.py:603-605
            if (STASH && active[t] && h == 0 && w == (t & 3)) {
                float* rc = ra.rec + (size_t)s * RK_NF * ra.cap + idx[t];
                rc[0 * ra.cap] = x[t]; rc[1 * ra.cap] = y[t]; rc[2 * ra.cap] = z[t];
                rc[3 * ra.cap] = px[t]; rc[4 * ra.cap] = py[t]; rc[5 * ra.cap] = pz[t];
#pragma unroll
                for (int k = 0; k < 6; ++k) { rc[(6 + k) * ra.cap] = w1[t][k]; rc[(12 + k) * ra.cap] = w2[k]; }
                rc[18 * ra.cap] = __int_as_float((g1[t] ? 1 : 0) | (g2 ? 2 : 0) | (rej ? 4 : 0));
            }
            if (active[t] && !rej) { x[t] = nx; y[t] = ny; z[t] = nz; }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (active[t] && h == 0 && w == 0) ra.xw[n[t]] = make_float4(x[t], y[t], z[t], zw[t]);
}
template <int NT, bool STASH, bool VOUT>
__global__ __launch_bounds__(WG_THREADS, 2) void k_rk2_split_uni(SplitUniArgs a) { rk2_split_uni_body<NT, STASH, VOUT>(a); }

int launch_rk2_split_uni(const SplitUniArgs& a, int64_t cap_samples, bool stash, hipStream_t st) {
    const int64_t tiles = (cap_samples + TILE - 1) / TILE;
    if (tiles <= 0) return 0;
    ProfScope ps(PK_RK2_FWD, st);
    // NVFI_SPLIT_UNI_NT=1 (experiment): one tile per workgroup, output layer on the vector pipe
    static int unt = -1;
    if (unt < 0) unt = 2;      // (the NVFI_SPLIT_UNI_NT / _VOUT / _BWD_NT / NVFI_SPLIT_NT / _VOUT sweep knobs of rounds 3-4 were retired in round 6: these are the swept optima)
    if (unt == 1) {
        const dim3 g1((unsigned)tiles), b1(WG_THREADS);
        if (stash) hipLaunchKernelGGL((k_rk2_split_uni<1, true, true>), g1, b1, SPLIT_VOUT_LDS_BYTES(1), st, a);
        else hipLaunchKernelGGL((k_rk2_split_uni<1, false, true>), g1, b1, SPLIT_VOUT_LDS_BYTES(1), st, a);
        LAUNCHCK();
        return 0;
    }
    const dim3 g((unsigned)((tiles + 1) / 2)), b(WG_THREADS);
    // NVFI_SPLIT_UNI_VOUT=1 (opt-in): the output layer on the vector pipe as in the prefilter (velnet_split_vout).  Alone the kernel gains 5.5 %
    // (0.475 -> 0.447 ms per step, 0.59 -> 0.63 of the fp32 MFMA peak), but the allocator then takes 228 / 204 registers instead of 179 / 150, the
    // gather / scatter kernels of the other chains no longer fit beside its two waves per SIMD, and the three-stream step gains nothing
    // (5.04 against 5.07 ms): off by default, where it is the numbers of k_rk2_fwd<true, STASH> bit for bit
    static int vout = -1;
    if (vout < 0) vout = 0;
    if (vout) {
        if (stash) hipLaunchKernelGGL((k_rk2_split_uni<2, true, true>), g, b, SPLIT_VOUT_LDS_BYTES(2), st, a);
        else hipLaunchKernelGGL((k_rk2_split_uni<2, false, true>), g, b, SPLIT_VOUT_LDS_BYTES(2), st, a);
    } else {
        if (stash) hipLaunchKernelGGL((k_rk2_split_uni<2, true, false>), g, b, SPLIT_LDS_BYTES(2), st, a);
        else hipLaunchKernelGGL((k_rk2_split_uni<2, false, false>), g, b, SPLIT_LDS_BYTES(2), st, a);
    }
    LAUNCHCK();
    return 0;
}

// ---------------------------------------------------------------- RK2 adjoint of the render warp on the same layout
// (k_rk2_bwd of vel.hip: same recurrence, same stash rows, same K order per accumulator -> the same adjoint stash bit for bit).
// Wave w owns rows [32w, 32w + 32) of every layer's INPUT gradient (one tile of the transposed weights, x4 fragments from L2), loads
// only its own 16 z rows and stores only its own 16 adjoint rows per layer and tile; the 128 -> 28 input layer of tile t is
// contracted by wave (owner + t) & 3 and its 16 slot gradients are broadcast through LDS.  36 KB of LDS instead of the 128 KB
// double-buffered fragment pipe of k_rk2_bwd: the kernel shares a CU with whatever the other streams run.
#define SPLIT_BWD_LDS_BYTES(NT) ((NT) * (SPLIT_XCH_F4 * 16 + 16 * 64 * 4))

template <int NT>
__device__ __forceinline__ void velnet_split_bwd(const float4* const* t4, float4* xch, float* bc, int w, int owner, int lane,
                                                 const float (&gw4)[NT][4], const float* const* zst, float* const* gst, float4* wq,
                                                 float (&ge)[NT][16]) {
    f32x16 acc[NT];
    float zp[NT][16];
    const float4* xl = xch + lane;
    const int mine = (w - owner) & 3;
    wq[0] = t4[5][(size_t)w * 64 + lane];                 // T5: 4 tiles x 1 group of 4 K-steps
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) zp[t][r] = STASH_LD(zst[t][(size_t)(4 * 64 + 16 * w + r) * REGF + lane]);
        if (w == t) {                                     // adjoint of the 6 outputs: B operand of the output layer's weight gradient
            float* gw_rows = gst[t] + (size_t)5 * 64 * REGF;
#pragma unroll
            for (int r = 0; r < 16; ++r) gw_rows[r * REGF + lane] = r < 4 ? gw4[t][r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
    {
        const float av[4] = {wq[0].x, wq[0].y, wq[0].z, wq[0].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = MFMA32(av[k], gw4[t][k], acc[t]);
    }
#pragma unroll 1
    for (int l = 4; l >= 0; --l) {
        if (l >= 1) split_load<16>(t4[l] + (size_t)w * 16 * 64, lane, wq);
        else if (mine < NT) split_load<16>(t4[0], lane, wq);
        float g[NT][16];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                g[t][r] = acc[t][r] * act_d1<1>(zp[t][r]);
                STASH_ST(gst[t][(size_t)(l * 64 + 16 * w + r) * REGF + lane], g[t][r]);
            }
            if (l >= 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) zp[t][r] = STASH_LD(zst[t][(size_t)((l - 1) * 64 + 16 * w + r) * REGF + lane]);
            }
        }
        __syncthreads();                                  // the previous layer's readers of the exchange buffer are done
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                xch[(t * 16 + 4 * w + k) * 64 + lane] = make_float4(g[t][4 * k], g[t][4 * k + 1], g[t][4 * k + 2], g[t][4 * k + 3]);
        __syncthreads();
        if (l >= 1) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            split_mfma_lds<NT>(wq, xl, 0, acc);
        }
    }
    if (mine < NT) {
        f32x16 ao[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) ao[0][r] = 0.f;
        split_mfma_lds<1>(wq, xl, mine, ao);
#pragma unroll
        for (int r = 0; r < 16; ++r) bc[(mine * 16 + r) * 64 + lane] = ao[0][r];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ge[t][r] = bc[(t * 16 + r) * 64 + lane];
}

template <int NT>
__global__ __launch_bounds__(WG_THREADS, 2) void k_rk2_split_bwd(SplitBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* xch = reinterpret_cast<float4*>(lds);
    float* bc = lds + NT * SPLIT_XCH_F4 * 4;
    const Rk2Args& ra = a.r;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int owner = blockIdx.x & 3;
    const int count = *ra.count;
    if ((int)blockIdx.x * NT * TILE >= (count + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES) return;
    bool active[NT]; int idx[NT]; float g3[NT][3];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        idx[t] = (blockIdx.x * NT + t) * TILE + (lane & 31);
        active[t] = idx[t] < count;
        const float4 gin = active[t] ? ra.gxk[ra.list[idx[t]]] : zero4();   // upstream gradient of the warped position
        g3[t][0] = gin.x; g3[t][1] = gin.y; g3[t][2] = gin.z;
    }
    float4 wq[16];
#pragma unroll 1
    for (int s = ra.nsteps - 1; s >= 0; --s) {
        const float dt = RK_DT(ra, s), tcur = RK_TC(ra, s);
        float gacc[NT][3], gup[NT][3];
        bool g1[NT], g2[NT], rej[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float* rc = ra.rec + (size_t)s * RK_NF * ra.cap + (active[t] ? idx[t] : 0);
            const int flags = active[t] ? __float_as_int(rc[18 * ra.cap]) : 7;
            g1[t] = flags & 1; g2[t] = flags & 2; rej[t] = flags & 4;
#pragma unroll
            for (int c = 0; c < 3; ++c) { gacc[t][c] = 0.f; gup[t][c] = g3[t][c]; }
        }
#pragma unroll 1
        for (int e = 1; e >= 0; --e) {
            const int po = e ? 3 : 0, wo = e ? 12 : 6;
            const float coef = e ? -dt : -0.5f * dt;
            const float te = e ? tcur - 0.5f * dt : tcur;
            float r4[NT][4], gloc[NT][3], ge[NT][16];
            const float* zs[NT]; float* gs[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float* rc = ra.rec + (size_t)s * RK_NF * ra.cap + (active[t] ? idx[t] : 0);
                const bool gate = e ? g2[t] : g1[t];
                const size_t es = (size_t)(2 * s + e) * ra.cap_tiles + (size_t)blockIdx.x * NT + t;
                zs[t] = ra.zst + es * (VEL_Z_REGS * REGF); gs[t] = ra.gst + es * (VEL_G_REGS * REGF);
                float p[3], wv[6], gv[3], gw[6];
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c] = active[t] ? rc[(po + c) * ra.cap] : 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) wv[k] = active[t] ? rc[(wo + k) * ra.cap] : 0.f;
                const bool on = active[t] && !rej[t] && !gate;
#pragma unroll
                for (int c = 0; c < 3; ++c) gv[c] = on ? coef * gup[t][c] : 0.f;
                gw[0] = gv[0]; gw[1] = gv[1]; gw[2] = gv[2];
                gw[3] = p[2] * gv[1] - p[1] * gv[2];
                gw[4] = -p[2] * gv[0] + p[0] * gv[2];
                gw[5] = p[1] * gv[0] - p[0] * gv[1];
                gloc[t][0] = -wv[5] * gv[1] + wv[4] * gv[2];
                gloc[t][1] = wv[5] * gv[0] - wv[3] * gv[2];
                gloc[t][2] = -wv[4] * gv[0] + wv[3] * gv[1];
                scatter6(gw, h, r4[t]);
            }
            velnet_split_bwd<NT>(a.t4, xch, bc, w, owner, lane, r4, zs, gs, wq, ge);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float* rc = ra.rec + (size_t)s * RK_NF * ra.cap + (active[t] ? idx[t] : 0);
                float p[3], x0[16];
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c] = active[t] ? rc[(po + c) * ra.cap] : 0.f;
                vel_encode_slots(make_float4(p[0], p[1], p[2], te), h, x0);
                const float4 gq = vel_encode_bwd(ge[t], x0, h);
                gloc[t][0] += gq.x; gloc[t][1] += gq.y; gloc[t][2] += gq.z;
#pragma unroll
                for (int c = 0; c < 3; ++c) { gacc[t][c] += gloc[t][c]; gup[t][c] = gloc[t][c]; }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (active[t] && !rej[t]) {
#pragma unroll
                for (int c = 0; c < 3; ++c) g3[t][c] = g3[t][c] + gacc[t][c] + 0.f;
            }
    }
}

int launch_rk2_split_bwd(const SplitBwdArgs& a, int64_t cap_samples, hipStream_t st) {
    const int64_t tiles = (cap_samples + TILE - 1) / TILE;
    if (tiles <= 0) return 0;
    ProfScope ps(PK_RK2_BWD, st);
    static int nt = -1;
    if (nt < 0) nt = 2;
    if (nt == 1) hipLaunchKernelGGL(k_rk2_split_bwd<1>, dim3((unsigned)tiles), dim3(WG_THREADS), SPLIT_BWD_LDS_BYTES(1), st, a);
    else hipLaunchKernelGGL(k_rk2_split_bwd<2>, dim3((unsigned)((tiles + 1) / 2)), dim3(WG_THREADS), SPLIT_BWD_LDS_BYTES(2), st, a);
    LAUNCHCK();
    return 0;
}

// wide = 0: one tile per workgroup (shortest latency: short lists); wide = 1: NVFI_SPLIT_NT tiles per workgroup share every weight load
int launch_rk2_split(const SplitArgs& a, int64_t cap_points, int wide, hipStream_t st) {
    const int64_t tiles = (cap_points + TILE - 1) / TILE;
    if (tiles <= 0) return 0;
    ProfScope ps(PK_PDE_PREFILTER, st);
    static int nt = -1, vout = 1;
    if (nt < 0) {
        // NVFI_SPLIT_NT: tiles per workgroup of the wide launch.  With the output layer on the vector pipe one tile is the default: alone it is
        // 1.5 % slower than two (1.32 against 1.30 ms; every weight load feeds one tile), but its 130-register workgroups leave the render
        // chains' kernels more room beside it - the three-stream step is 1 % faster (5.04 against 5.10 ms).  NVFI_SPLIT_VOUT=0: matrix pipe.
        nt = 1; vout = 1;
        HIPCK(hipFuncSetAttribute((const void*)k_rk2_split<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SPLIT_VOUT_LDS_BYTES(4)));
        HIPCK(hipFuncSetAttribute((const void*)k_rk2_split<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SPLIT_VOUT_LDS_BYTES(4)));
    }
    const int n = wide ? nt : 1;
    const dim3 b(WG_THREADS);
    // (both variants get the larger LDS image of the vector-pipe form: three workgroups per CU either way)
#define SPLIT_GO(NT_) do { const dim3 g((unsigned)((tiles + NT_ - 1) / NT_));                                                              \
        if (vout) hipLaunchKernelGGL((k_rk2_split<NT_, true>), g, b, SPLIT_VOUT_LDS_BYTES(NT_), st, a);                                   \
        else hipLaunchKernelGGL((k_rk2_split<NT_, false>), g, b, SPLIT_VOUT_LDS_BYTES(NT_), st, a); } while (0)
    if (n == 1) SPLIT_GO(1); else if (n == 4) SPLIT_GO(4); else SPLIT_GO(2);
#undef SPLIT_GO
    LAUNCHCK();
    return 0;
}
