// vel_x6.hip - the per-point RK2 back-advection of vel_split.hip (reference models/tensorf_keyframe.py:575-611 around the gated VelBasis
// of models/velocity_field.py:21-98) with the hidden layers' fp32 products formed on the 16-BIT matrix pipe - exactly.
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate (64 FLOP / clk / SIMD) and nothing issues beside it - not from another wave, not
// from its own (tools/probes/dual_pipe_probe3.hip) - so the fp32 kernels sit at 0.6-0.75 of a 157 TFLOP/s roof.  v_mfma_f32_32x32x16_bf16
// delivers 16x the multiply-adds per cycle and leaves 4-5 issue slots per instruction to the issuing wave's own VALU work.
//
// How (the "x6" scheme; the 6-product analogue of cuBLAS' BF16x9 fp32 emulation):
//   every fp32 operand is split into THREE bfloat16 terms, x = x1 + x2 + x3 with x1 = rn_bf16(x), x2 = rn_bf16(x - x1), x3 = x - x1 - x2:
//   the residuals are exact in fp32 (|x - x1| <= 2^-9 |x| has at most 16 significant bits, the second residual at most 8), bfloat16 has
//   the exponent range of fp32, so the three 8-bit terms carry the 24-bit significand EXACTLY - no scaling, no subnormal terms (a binary16
//   version of this file lost first terms below 6.1e-5 to the subnormal range: 4.7e-6 outliers against float64);
//   a product w x = sum_ij w_i x_j is formed from the SIX term products of relative magnitude >= 2^-18 - each the exact product of two 8-bit
//   numbers, added into an fp32 accumulator by the MFMA - w1x1 | w1x2 + w2x1 | w2x2 + w1x3 + w3x1: three accumulators, one per magnitude
//   class, added once per layer, small ones first.  The dropped products (w2x3, w3x2, w3x3) are <= 2^-26 |w x|: a quarter of the rounding
//   error of ONE fp32 product.
//   What is left is the rounding of the fp32 accumulation itself - with the small terms summed apart from the large ones it is SMALLER
//   than the error of a K = 128 fp32 dot product accumulated in sequence (tests/studies/x6_accuracy_study.py; on the device:
//   tests/test_gpu_x6.py compares both kernels against float64).  Encoder, biases, SiLU, the 128 -> 6 output layer (vector pipe, fp32
//   FMAs as in velnet_split_vout), gates and the RK2 recurrence are the fp32 code of vel_split.hip.
//
// Layout: feature split as in k_rk2_split - one workgroup of four waves per NT point tiles, wave w owns output rows [32 w, 32 w + 32) of
// every hidden layer; the weights' three bfloat16 images ([layer][row tile][K step][lane] x 8 halves = the A operand of one MFMA per 16
// bytes) stream from L2 two K steps ahead; the layer input travels between the waves through LDS already split: every lane writes the
// six 16-byte B operands (3 terms x 2 K steps) of its 16 outputs and reads 3 x 8 per tile and layer.
#include <stdlib.h>
#include <utility>
#include "common.h"
#include "vel.h"
#include "pde.h"
#include "engine16.h"
#include "x6.h"

// ---------------------------------------------------------------- packing: three bfloat16 images of layers 0..4 (x6_pack_body, x6.h)
__global__ __launch_bounds__(256) void k_pack_x6(X6PackArgs a) { x6_pack_body(a, blockIdx.x * blockDim.x + threadIdx.x); }
int launch_pack_x6(const float* const* W, void* img, hipStream_t st, void* imgT) {
    X6PackArgs pk;
    for (int l = 0; l < 5; ++l) pk.W[l] = W[l];
    pk.img = reinterpret_cast<b8_t*>(img); pk.imgT = reinterpret_cast<b8_t*>(imgT);
    hipLaunchKernelGGL(k_pack_x6, dim3((X6_H8 + 255) / 256), dim3(256), 0, st, pk);
    LAUNCHCK();
    return 0;
}

// ---------------------------------------------------------------- three-term split of 8 activations -> the B operands of one K step
// By TRUNCATION (NVFI_X6_ROUND_SPLIT undefined): t1 = the upper 16 bits of x, r = x - t1 (<= 16 significant bits, exact), t2 = the upper 16
// bits of r, t3 = r - t2 (<= 8 significant bits: its upper 16 bits ARE the value) - x = t1 + t2 + t3 exactly, like the rounded split the
// weights get at pack time (x6.h split3), for 3 v_perm_b32 + 4 v_and_b32 + 4 v_sub_f32 per PAIR of values instead of 6 conversions, 4 shifts
// / masks and 4 subtractions: the epilogue is VALU time the matrix pipe waits for.  The terms are up to twice as large as rounded ones
// (|t2| < 2^-7 |x|, |t3| < 2^-15 |x|), so the three dropped term products (w2 x3, w3 x2, w3 x3: < 2^-22 of the product) are too; the error
// against float64 stays at the fp32 kernels' (tests/test_gpu_x6.py).
__device__ __forceinline__ void split3_8(const float* v, b8_t& b1, b8_t& b2, b8_t& b3) {
#ifdef NVFI_X6_ROUND_SPLIT
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 t1, t2, t3;
        split3(v[j], t1, t2, t3);
        b1[j] = t1; b2[j] = t2; b3[j] = t3;
    }
#else
    unsigned p1[4], p2[4], p3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float xa = v[2 * j], xb = v[2 * j + 1];
        const unsigned ua = __float_as_uint(xa), ub = __float_as_uint(xb);
        p1[j] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                       // (upper half of xb) << 16 | upper half of xa
        const float ra = xa - __uint_as_float(ua & 0xffff0000u), rb = xb - __uint_as_float(ub & 0xffff0000u);
        const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
        p2[j] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
        const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
        p3[j] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 q1 = {p1[0], p1[1], p1[2], p1[3]}, q2 = {p2[0], p2[1], p2[2], p2[3]}, q3 = {p3[0], p3[1], p3[2], p3[3]};
    b1 = __builtin_bit_cast(b8_t, q1); b2 = __builtin_bit_cast(b8_t, q2); b3 = __builtin_bit_cast(b8_t, q3);
#endif
}

// the six term products of one K step for one tile: a0 += A1 B1 (the leading terms: the sum the fp32 MFMA forms, in the same order) ;
// a1 += A1 B2 + A2 B2 + A2 B1 + A1 B3 + A3 B1 (everything <= 2^-8 of it: rounding among these is 2^-32 of the result).  Two accumulators since
// round 5 late (three before: a separate one for the 2^-16 class bought nothing measurable and cost 16 registers and 16 adds per drain)
__device__ __forceinline__ void x6_step(const b8_t& A1, const b8_t& A2, const b8_t& A3, const b8_t& B1, const b8_t& B2, const b8_t& B3,
                                        f32x16& a0, f32x16& a1) {
    a0 = MFMA16B(A1, B1, a0);
    a1 = MFMA16B(A1, B2, a1);
    a1 = MFMA16B(A2, B2, a1);
    a1 = MFMA16B(A2, B1, a1);
    a1 = MFMA16B(A1, B3, a1);
    a1 = MFMA16B(A3, B1, a1);
}

// LDS exchange image of one tile: [term][K step 0..7][lane] h8
#define X6_XCH_H8 (3 * 8 * 64)
#define X6_LDS_BYTES(NT) ((NT) * (X6_XCH_H8 * 16 + 4 * 2 * 32 * 16) + 6 * 128 * 4 + 4 * 2 * 16 * 8 * 4)

// one gated-velocity network evaluation of the workgroup's NT tiles (velnet_split_vout of vel_split.hip with x6 hidden layers)
//
// Schedule of one layer: the tile's whole layer input (8 K steps x 3 terms = 24 operands, 96 registers) is read from LDS, the A operands
// (weights) rotate through four register sets refilled from L2 behind the MFMAs of a later step, and the layer ends with the drain - the
// VALU read of the three accumulators (the compiler puts the s_nop the hardware needs in front of it) - behind which empty asm statements
// keep every operand register of the layer alive.  The pins and the opaque step on the layer-0 operands are belt and braces from the hunt
// described at the bottom of this file (they cost nothing); they are NOT what fixed it - tools/probes/mfma_hazard_probe.hip shows that
// overwriting an MFMA's A or B registers straight behind its issue is harmless on gfx950.
// STASH (training render): the fp32 pre-activations z of the five hidden layers (wave w = rows 16 w .. 16 w + 15 of each layer) and the encoder
// slots go to the per-(evaluation, tile) stash in the layout of k_rk2_split_uni<STASH> (vel_split.hip): the fp32 adjoint kernels read it unchanged
template <int NT, bool STASH = false, bool X4 = false>     // X4 (with STASH): Rk2Args::z_x4, the z rows of layers 0..3 as x4 stash blocks
__device__ __forceinline__ void velnet_x6(const b8_t* __restrict__ img, b8_t* xch, float4* part, const float4* w5l, int w, int lane, int h,
                                          const float4* q, const float* lb, float (&out6)[NT][6], float* const* zst = nullptr, float* const* x0st = nullptr) {
    const b8_t* W1 = img; const b8_t* W2 = img + X6_H8; const b8_t* W3 = img + 2 * X6_H8;
    f32x16 a0[NT], a1[NT];
    b8_t Bf[NT][8][3];
    // ---- layer 0: every wave encodes the point itself (28 inputs in 16 slots per lane half = 2 K steps)
    // v: the fp32 pre-activations of the layer just finished; every layer ends with the drain (the read of its accumulators) and the pins
    float v[NT][16];
    {
        b8_t A1[2], A2[2], A3[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int o = X6_L0 + (w * 2 + s) * 64 + lane;
            A1[s] = W1[o]; A2[s] = W2[o]; A3[s] = W3[o];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float x0[16];
            vel_encode_slots(q[t], h, x0);
            if (STASH && w == t) stash_store<16>(x0st[t], lane, x0);
            split3_8(x0, Bf[t][0][0], Bf[t][0][1], Bf[t][0][2]);
            split3_8(x0 + 8, Bf[t][1][0], Bf[t][1][1], Bf[t][1][2]);
            // one register tuple per operand, alive until the pin behind the drain
#pragma unroll
            for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(Bf[t][s][0]), "+v"(Bf[t][s][1]), "+v"(Bf[t][s][2]));
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[t][r] = lb[32 * w + (r & 3) + 8 * (r >> 2) + 4 * h]; a1[t][r] = 0.f; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) x6_step(A1[s], A2[s], A3[s], Bf[t][s][0], Bf[t][s][1], Bf[t][s][2], a0[t], a1[t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[t][r] = a1[t][r] + a0[t][r];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) asm volatile("" :: "v"(Bf[t][s][0]), "v"(Bf[t][s][1]), "v"(Bf[t][s][2]), "v"(v[t][0]), "v"(v[t][15]));
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" :: "v"(A1[s]), "v"(A2[s]), "v"(A3[s]), "v"(v[0][0]), "v"(v[0][15]));
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        // the next layer's first three K steps start their trip from L2 now; they land behind the epilogue and the exchange
        const b8_t* P1 = W1 + X6_LH(l + 1) + (w * 8) * 64 + lane;
        const b8_t* P2 = W2 + X6_LH(l + 1) + (w * 8) * 64 + lane;
        const b8_t* P3 = W3 + X6_LH(l + 1) + (w * 8) * 64 + lane;
        b8_t A1[4], A2[4], A3[4];
#pragma unroll
        for (int s = 0; s < 3; ++s) { A1[s] = P1[s * 64]; A2[s] = P2[s * 64]; A3[s] = P3[s * 64]; }
        if (STASH) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (X4) stash_st16_x4(zst[t] + (size_t)(l * 64 + 16 * w) * REGF, lane, v[t]);      // (four 16-byte stores instead of sixteen rows)
                else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) STASH_ST(zst[t][(size_t)(l * 64 + 16 * w + r) * REGF + lane], v[t][r]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[t][r] = act_f<1>(v[t][r]);
        __syncthreads();                                 // the previous layer's readers of the exchange buffer are done
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                b8_t b1, b2, b3;
                split3_8(v[t] + 8 * k, b1, b2, b3);
                b8_t* dst = xch + (size_t)t * X6_XCH_H8 + (2 * w + k) * 64 + lane;
                dst[0] = b1; dst[8 * 64] = b2; dst[16 * 64] = b3;
            }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const b8_t* src = xch + (size_t)t * X6_XCH_H8 + s * 64 + lane;
                Bf[t][s][0] = src[0]; Bf[t][s][1] = src[8 * 64]; Bf[t][s][2] = src[16 * 64];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[t][r] = lb[128 * (l + 1) + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h]; a1[t][r] = 0.f; }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t) x6_step(A1[s & 3], A2[s & 3], A3[s & 3], Bf[t][s][0], Bf[t][s][1], Bf[t][s][2], a0[t], a1[t]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 3 < 8) { A1[(s + 3) & 3] = P1[(s + 3) * 64]; A2[(s + 3) & 3] = P2[(s + 3) * 64]; A3[(s + 3) & 3] = P3[(s + 3) * 64]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        // drain: the sums of the three magnitude classes, small ones first - the read of the accumulators is the point behind which the
        // layer's MFMAs have completed; the pins keep every operand register of the layer untouched up to here
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[t][r] = a1[t][r] + a0[t][r];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 8; ++s) asm volatile("" :: "v"(Bf[t][s][0]), "v"(Bf[t][s][1]), "v"(Bf[t][s][2]), "v"(v[t][0]), "v"(v[t][15]));
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" :: "v"(A1[s]), "v"(A2[s]), "v"(A3[s]), "v"(v[0][0]), "v"(v[0][15]));
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- 128 -> 6 output layer on the vector pipe (velnet_split_vout): fp32 FMAs over the 16 activations each lane holds
    float p[NT][6];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int o = 0; o < 6; ++o) p[t][o] = 0.f;
    const float4* wl = w5l + (w * 2 + h) * 32;
    float (&zl)[NT][16] = v;
    if (STASH) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) STASH_ST(zst[t][(size_t)(4 * 64 + 16 * w + r) * REGF + lane], zl[t][r]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if ((r & 1) == 0) __builtin_amdgcn_sched_barrier(0);
        const float4 wa = wl[2 * r], wb = wl[2 * r + 1];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float av = act_f<1>(zl[t][r]);
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
            __builtin_amdgcn_sched_barrier(0);
            const float4 A = part[((t * 4 + ww) * 2 + 0) * 32 + (lane & 31)], B = part[((t * 4 + ww) * 2 + 1) * 32 + (lane & 31)];
            out6[t][0] += A.x; out6[t][1] += A.y; out6[t][2] += A.z; out6[t][3] += A.w; out6[t][4] += B.x; out6[t][5] += B.y;
        }
    }
}

// the recurrence of k_rk2_split<NT, true> (vel_split.hip), per-point times
template <int NT>
__global__ __launch_bounds__(WG_THREADS, NT == 1 ? 2 : 1) void k_rk2_x6(X6Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    b8_t* xch = reinterpret_cast<b8_t*>(lds);
    float4* part = reinterpret_cast<float4*>(xch + NT * X6_XCH_H8);
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    float* lb = reinterpret_cast<float*>(part + NT * 4 * 2 * 32);
    for (int k = threadIdx.x; k < 6 * 128; k += WG_THREADS) lb[k] = (k & 127) < (k < 640 ? 128 : 6) ? a.f.vb[k >> 7][k & 127] : 0.f;      // (the raw bias vectors)
    float* w5f = lb + 6 * 128;
    for (int k = threadIdx.x; k < 4 * 2 * 16 * 8; k += WG_THREADS) {
        const int o = k & 7, r = (k >> 3) & 15, hh = (k >> 7) & 1, ww = k >> 8;
        w5f[k] = o < 6 ? a.f.vW[5][o * 128 + 32 * ww + (r & 3) + 8 * (r >> 2) + 4 * hh] : 0.f;
    }
    const float4* w5l = reinterpret_cast<const float4*>(w5f);
    const b8_t* img = reinterpret_cast<const b8_t*>(a.img);
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
        velnet_x6<NT>(img, xch, part, w5l, w, lane, h, q, lb, o6);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v1[3];
            vel_from_w(o6[t], x[t], y[t], z[t], v1);
            if (gated_out(a.f, x[t], y[t], z[t])) { v1[0] = v1[1] = v1[2] = 0.f; }
            const float hdt = 0.5f * dt[t];
            px[t] = x[t] - hdt * v1[0]; py[t] = y[t] - hdt * v1[1]; pz[t] = z[t] - hdt * v1[2];
            q[t] = make_float4(px[t], py[t], pz[t], tcur[t] - hdt);
        }
        velnet_x6<NT>(img, xch, part, w5l, w, lane, h, q, lb, o6);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v2[3];
            vel_from_w(o6[t], px[t], py[t], pz[t], v2);
            if (gated_out(a.f, px[t], py[t], pz[t])) { v2[0] = v2[1] = v2[2] = 0.f; }
            const float nx = x[t] - dt[t] * v2[0], ny = y[t] - dt[t] * v2[1], nz = z[t] - dt[t] * v2[2];
            const bool rej = a.f.gate_sur && gated_out(a.f, nx, ny, nz);   // tensorf_keyframe.py:603-605
            if (live[t] && !rej) { x[t] = nx; y[t] = ny; z[t] = nz; }
            if (live[t]) { off[t] = off[t] - dt[t]; tcur[t] = tcur[t] - dt[t]; }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
        if (active[t] && h == 0 && w == 0) {
            if (a.xout3) { float* o = a.xout3 + 3 * (size_t)n[t]; o[0] = x[t]; o[1] = y[t]; o[2] = z[t]; }
            else a.xw[n[t]] = make_float4(x[t], y[t], z[t], zw[t]);
        }
}

// ---------------------------------------------------------------- render warp: every sample takes the same (dt_s, t_s) sequence
// (rk2_split_uni_body of vel_split.hip on the x6 evaluation: same compact list, same in-place update, same stash and records)
template <int NT, bool STASH, bool X4 = false>
__global__ __launch_bounds__(WG_THREADS, NT == 1 ? 2 : 1) void k_rk2_x6_uni(X6UniArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    b8_t* xch = reinterpret_cast<b8_t*>(lds);
    float4* part = reinterpret_cast<float4*>(xch + NT * X6_XCH_H8);
    const Rk2Args& ra = a.r;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = *ra.count;
    // whole 128-sample groups when stashing: the adjoint and weight-gradient kernels walk every tile of the last, ragged group
    if ((int)blockIdx.x * NT * TILE >= (STASH ? (count + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES : count)) return;
    bool active[NT]; int n[NT], idx[NT]; float x[NT], y[NT], z[NT], zw[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        idx[t] = (blockIdx.x * NT + t) * TILE + (lane & 31);
        active[t] = idx[t] < count;
        n[t] = active[t] ? ra.list[idx[t]] : 0;
        const float4 q0 = active[t] ? ra.xw[n[t]] : zero4();
        x[t] = q0.x; y[t] = q0.y; z[t] = q0.z; zw[t] = q0.w;
    }
    float* lb = reinterpret_cast<float*>(part + NT * 4 * 2 * 32);
    for (int k = threadIdx.x; k < 6 * 128; k += WG_THREADS) lb[k] = (k & 127) < (k < 640 ? 128 : 6) ? ra.f.vb[k >> 7][k & 127] : 0.f;
    float* w5f = lb + 6 * 128;
    for (int k = threadIdx.x; k < 4 * 2 * 16 * 8; k += WG_THREADS) {
        const int o = k & 7, r = (k >> 3) & 15, hh = (k >> 7) & 1, ww = k >> 8;
        w5f[k] = o < 6 ? ra.f.vW[5][o * 128 + 32 * ww + (r & 3) + 8 * (r >> 2) + 4 * hh] : 0.f;
    }
    const float4* w5l = reinterpret_cast<const float4*>(w5f);
    const b8_t* img = reinterpret_cast<const b8_t*>(a.img);
    const int nsteps = ra.sched ? __float_as_int(ra.sched[2]) : ra.nsteps;
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
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
        velnet_x6<NT, STASH, X4>(img, xch, part, w5l, w, lane, h, q, lb, o6, z1, x1);
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
        velnet_x6<NT, STASH, X4>(img, xch, part, w5l, w, lane, h, q, lb, o6, z2, x2);
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

// TWO WORKGROUPS PER CU, AND NO PACKED-FP32 VALU CODE (round 5).  As first built - compiled like every other file - these kernels
// gave run-to-run differences on the same input in 0.1-0.7 % of the tiles when two workgroups shared a CU: per-layer dumps put the first
// difference in the positional encoder of a step's SECOND evaluation, one slot (the x component, one frequency) of ONE of the four waves
// wrong in lanes 48..63, as if its argument had been a small constant.  Never with one workgroup per CU (8 x 524 288 points bit-identical),
// never in the fp32 kernels (which run two per CU with the same encoder).  Delta debugging on the kernel itself (only run-to-run equality
// is tested, so parts can be cut out freely): gone without the MFMAs, gone with a trig-free encoder, still there without the LDS exchange,
// without the output stage, without the range reduction or the v_cndmask selects - and gone as soon as the encoder's arguments are made
// opaque one by one, which keeps the SLP vectoriser from pairing two evaluations into v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (SGPR
// pair, op_sel).  So nvfi_amd/build.py compiles this file with -fno-slp-vectorize: no v_pk_* instruction is left in these kernels, the
// repeats are bit-identical at two workgroups per CU (tests/test_gpu_x6.py: 8 x 524 288 points; the render-warp determinism tests), and
// the speed is the same (the epilogues hide behind the other workgroup's MFMAs).  The mechanism is NOT understood: none of the pieces
// reproduces in isolation (tools/probes/mfma_hazard_probe.hip, nine experiments incl. packed fp32 beside this MFMA pattern at 1, 2 and 4
// waves per SIMD: all clean).  Asking for more than half of the CU's LDS keeps a second workgroup off the CU - the other
// configuration known to be clean (about 7 % of the step slower; the switch for it, NVFI_X6_ONE_WG, was retired in round 6).
#define X6_ONE_WG_LDS (84 * 1024)
// (per device: hipFuncSetAttribute applies to the device that is current - one process per GPU never sees a second one, a host that drives
// several devices from one process does)
template <typename K>
static int x6_set_lds(K kernel) {
    HIPCK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, X6_ONE_WG_LDS));
    return 0;
}
static size_t x6_lds_nt1() { return (size_t)X6_LDS_BYTES(1); }      // (NVFI_X6_ONE_WG=1 asked for X6_ONE_WG_LDS here: retired in round 6, the fence in build.py is the fix)

int launch_rk2_x6_uni(const X6UniArgs& a, int64_t cap_samples, bool stash, hipStream_t st) {
    const int64_t tiles = (cap_samples + TILE - 1) / TILE;
    if (tiles <= 0) return 0;
    static DeviceOnce once;
    if (once.run([] { return (x6_set_lds(k_rk2_x6_uni<1, true, true>) || x6_set_lds(k_rk2_x6_uni<1, true>) || x6_set_lds(k_rk2_x6_uni<1, false>)) ? 1 : 0; })) return 1;
    ProfScope ps(PK_RK2_FWD, st);
    // NVFI_X6W_UNI: 1 (default) eval renders on the one-wave-per-tile kernel of vel_x6w.hip (bit-identical; an 800 x 800 test frame 134 -> 120 ms),
    // 2 training renders too (same stash and records; no faster there: 0.35 against 0.37 ms, the stash stores are not hidden), 0 neither
    static int wuni = -1;
    if (wuni < 0) { const char* e = getenv("NVFI_X6W_UNI"); wuni = e ? atoi(e) : 1; }
    if ((wuni >= 1 && !stash) || wuni >= 2) return launch_rk2_x6w_uni(a, cap_samples, stash, st);
    // (round 6: the two-tiles-per-workgroup variants - NVFI_X6_NT=2 - are retired: never a default, 8 % slower)
    if (stash && a.r.z_x4) hipLaunchKernelGGL((k_rk2_x6_uni<1, true, true>), dim3((unsigned)tiles), dim3(WG_THREADS), x6_lds_nt1(), st, a);
    else if (stash) hipLaunchKernelGGL((k_rk2_x6_uni<1, true>), dim3((unsigned)tiles), dim3(WG_THREADS), x6_lds_nt1(), st, a);
    else hipLaunchKernelGGL((k_rk2_x6_uni<1, false>), dim3((unsigned)tiles), dim3(WG_THREADS), x6_lds_nt1(), st, a);
    LAUNCHCK();
    return 0;
}

int launch_rk2_x6(const X6Args& a, int64_t cap_points, hipStream_t st) {
    const int64_t tiles = (cap_points + TILE - 1) / TILE;
    if (tiles <= 0) return 0;
    static DeviceOnce once;
    if (once.run([] { return x6_set_lds(k_rk2_x6<1>) ? 1 : 0; })) return 1;
    // default since round 5 (late): one wave per tile, the epilogue in the MFMAs' VALU slots (vel_x6w.hip; bit-identical results, the bench
    // prefilter 0.89 -> 0.83 ms).  NVFI_X6W=0: the four-waves-per-tile kernel below
    static int x6w = -1;
    if (x6w < 0) { const char* e = getenv("NVFI_X6W"); x6w = e ? atoi(e) : 1; }
    // round 6: a SMALL call is latency-bound - its evaluations are a serial chain per tile, 60 k cycles each on one wave (x6w) against ~17.5 k with
    // the tile's four row tiles on four SIMDs (k_rk2_x6<1>; 35 k with two workgroups per CU) - so up to NVFI_X6W_MIN_TILES the four-wave kernel
    // runs; same products in the same order, identical bits (tests/test_gpu_x6.py).  Default 4096, measured on the PDE prefilter (trajectories of up
    // to 20 evaluations): 1024 tiles (the strong-scaling shard) 0.41 -> 0.19 ms, 4096 tiles 0.45 -> 0.44 ms, 8192 tiles (the full batch) one wave per
    // tile wins (0.80 against 0.89 ms).  train_segm's integrate_pos (3 000-30 000 occupied points, 40-60 evaluations deep): 1.03 -> 0.46 ms.
    static int min_tiles = -1;
    if (min_tiles < 0) { const char* e = getenv("NVFI_X6W_MIN_TILES"); min_tiles = e ? atoi(e) : 4096; }
    if (x6w && tiles > min_tiles) return launch_rk2_x6w(a, cap_points, st);
    hipLaunchKernelGGL(k_rk2_x6<1>, dim3((unsigned)tiles), dim3(WG_THREADS), x6_lds_nt1(), st, a);
    LAUNCHCK();
    return 0;
}
