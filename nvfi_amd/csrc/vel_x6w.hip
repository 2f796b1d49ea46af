// x6, one wave per 32-point tile (round 5, late): the velocity net of vel_x6.hip with the WHOLE 128-wide layer in one wave.
//
// Why.  k_rk2_x6 (vel_x6.hip) splits a layer's 128 output rows over the four waves of a workgroup.  Counters say where its time goes
// (profiles/r05_pmc_sq.csv): per wave and evaluation ~250 MFMAs (8 k cycles) stand against ~1 500 VALU instructions - every wave repeats the
// encoder, the RK2 glue and its own epilogue - and on this part a VALU instruction of one wave never overlaps a matrix instruction of ANOTHER
// wave (dual_pipe_probe3): the two add up, the matrix pipe is 40 % busy, and a second workgroup per CU only hides latencies.  Inside ONE wave
// about four VALU instructions per 16-bit MFMA are free.  So: one wave owns a tile and all four 32-row tiles of every layer; while the 48
// MFMAs of row tile m run, the wave's VALU slots carry the epilogue of row tile m - 1 (drain -> SiLU -> truncation split -> the two K steps of
// the NEXT layer's B operand, which in this layout are the lane's own: nothing is exchanged BETWEEN waves and there is no barrier; the layer
// output travels through 24 KB of LDS of the wave's own only because two register arrays of 96 do not fit); the epilogue of a layer's last
// row tile rides under K steps 0..5 of the next layer's first tile and is complete before K step 6 needs it.  The A operands (weights) are one
// linear stream of 136 entries (layer 0: 8, layers 1..4: 32 each) through an eight-slot register ring, seven entries ahead.
// Same products, same order of accumulation as k_rk2_x6 / k_rk2_x6_uni: results are bit-identical (tests/test_gpu_x6.py); DESIGN.md 4.8.2 has the
// counters of both kernels and what was tried.
// Round 6 (probes, per-tile stamps, two alternative kernels built and dropped): a tile's time is the wave's ISSUE time - 1.33 k cycles of
// epilogue VALU + 0.6 k for its 24 weight loads + the MFMAs' own issue = 2.27 k against 1.54 k of matrix time; the output layer's sums ride in the layer-4
// epilogues, the MFMA is pinned first in its slot, an evaluation sends for its successor's first ring entries before its uncovered tail.
#include <stdlib.h>
#include <utility>
#include "common.h"
#include "vel.h"
#include "pde.h"
#include "engine16.h"
#include "x6.h"

typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
typedef const b8_t __attribute__((address_space(1))) * x6w_gptr;      // (behind the opaque step a generic pointer would load through FLAT)
#define X6W_ENTRIES 136
#define X6W_RING 8                            // weight-stream ring: entry en + 7 is requested behind entry en's MFMAs (7 K steps = 1 300 cycles ahead:
                                             // one wave per SIMD has nobody to hide an L2 round trip behind; 3 ahead stalled every K step)
#define X6W_OB_H8 (8 * 3 * 64)               // per wave: the layer output on its way to the next layer's input registers, [K step][term][lane]
#define X6W_LDS_BYTES ((6 * 128 + 4 * 2 * 16 * 8) * 4 + 4 * X6W_OB_H8 * 16)

#ifdef X6W_TIMING                     // shader-clock stamps at the tile boundaries of one evaluation of one wave (printed at the end of k_rk2_x6w)
__device__ unsigned long long x6w_ts[32];
#define X6W_STAMP(c, k) do { if ((c).stamp) x6w_ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define X6W_STAMP(c, k) do { } while (0)
#endif
// the stream loads: per-lane 64-bit pointers (global_load, the default) - X6W_BUFLD (timing probe): buffer loads, SGPR resource + the lane's offset in one
// VGPR + a running SGPR offset: the same real cycles per evaluation (46.4 k against 46.6 k) - a weight load costs its ~25 cycles on the return path,
// not in the address form
#if defined(X6W_BUFLD)
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
#define X6W_OPAQUE(c) asm volatile("" : "+s"((c).so))
#define X6W_LD(c, P, E) __builtin_bit_cast(b8_t, __builtin_amdgcn_raw_buffer_load_b128((c).R##P, (int)(c).vo + (E) * 1024, (c).so, 0))
#define X6W_PTR(img, lane) ((x6w_gptr)(img))
#else
#define X6W_OPAQUE(c) asm volatile("" : "+v"((c).W1), "+v"((c).W2), "+v"((c).W3))
#define X6W_LD(c, P, E) ((c).P[(E) * 64])
#define X6W_PTR(img, lane) ((x6w_gptr)((img) + (lane)))
#endif
struct X6W {
    int stamp;
    unsigned vo;                   // lane * 16
#ifdef X6W_BUFLD
    __amdgpu_buffer_rsrc_t RW1, RW2, RW3; int so;
#endif
    x6w_gptr W1, W2, W3;           // per lane, RUNNING: the base of the current group of four stream entries (entry x of the lane at [(x & 3) * 64],
                                   // a 13-bit immediate offset); advanced by 4 KB once per four entries and made opaque, so that the compiler
                                   // neither re-derives 136 x 3 64-bit addresses from the kernel argument nor keeps them (the first build spilled 265)
    const float* lb;               // LDS: raw biases [6][128]
    const float4* w5l;             // LDS: output-layer weights [m][h][r][8]
    b8_t* ob;                      // LDS, this wave's: + lane
    float* zp;                     // STASH: the z rows of the current row tile (+ lane), RUNNING like the weight stream: 4 KB per row tile
    int lane, h;
};

template <int... Is, class F> __device__ __forceinline__ void x6w_for(std::integer_sequence<int, Is...>, F f) { (f(std::integral_constant<int, Is>{}), ...); }

// Register budget (one wave per SIMD, 512 registers): the layer input `in` (8 K steps x 3 terms = 96) lives in registers for the four row tiles
// that read it; the layer OUTPUT goes through 24 KB of LDS per wave (six 16-byte writes per row tile) and is pulled into `in` K step by K
// step as the last row tile of the layer releases them - one array, one LDS buffer, no ping-pong (two register arrays spilled 574 registers).
struct X6WEpi {                    // transient state of one pending epilogue
    float rr[16];
    unsigned pk[3][8];
    // LAST: the row tile's share of the 128 -> 6 output layer, formed in the same micro-slots (round 6)
    float p[6];                    // its partial sums: a chain over the tile's 16 activations, as velnet_x6 forms them
    float4 wa, wb;                 // output-layer weights of the activation whose turn is next (LDS broadcast reads, two micro-slots ahead; four ahead: no faster)
    float out[6];                  // the evaluation's outputs: bias, then the row tiles in order
    float* zst;                    // STASH: the pending tile's sixteen z rows (+ lane): pre-activation r is stored at u = 2 r, in the MFMAs' shadow
};
// piece I (0..35) of the epilogue of a row tile: v = its 16 pre-activations; K steps m2, m2 + 1 of the layer output
// The epilogue of a row tile as 38 micro-slots of about one transcendental + four plain VALU instructions each - what one 16-bit MFMA leaves
// room for in the same wave (dual_pipe_probe3; a slot with two transcendentals or six plain instructions costs ~45 cycles instead of 34):
//   u = 2 r      E_r: t = exp2(-log2(e) z_r)            u = 2 r + 1   R_r: s = rcp(1 + t)            u = 2 r + 2   M_r: z_r s       (act_f<1>'s arithmetic)
//   pair p = (2 p, 2 p + 1), complete at u = 4 p + 4:  u = 4 p + 5 .. 4 p + 8: the truncation split in four parts (3, 2, 3, 3 instructions)
//   u = 22, u = 37: the three 16-byte LDS writes of K step m2 / m2 + 1 of the layer output
// LAST (last hidden layer): no split - activation r enters the tile's six output-layer sums behind its SiLU: two FMAs per micro-slot over slots 3..45
// (weights of r + 1 requested behind them), the two lane halves added at u = 46, 47 (v_permlane32_swap), the tile added to the outputs at u = 47.  (Rounds 5-6: the
// activations went to LDS as floats and one pass behind the last tile did all 384 FMAs outside any MFMA's shadow: 6 % of an evaluation.)
// p + (p of the lane 32 away): v_permlane32_swap (gfx950) leaves the low half's value in both halves of one register and the high half's in the other;
// their sum is __shfl_xor(p, 32)'s bit for bit (an fp32 add commutes) without ds_bpermute's LDS round trip in the middle of an epilogue
__device__ __forceinline__ float x6w_add_halves(float p) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// STASH: 0 no stash | 1 row-major z rows | 2 x4 blocks for the row tiles of layers 0..3 (Rk2Args::z_x4); PX4: the PENDING tile is one of those
template <int U, bool LAST, int STASH = 0, bool PX4 = false>
__device__ __forceinline__ void x6w_micro(const X6W& c, float (&v)[16], X6WEpi& e, int m2) {
    if constexpr (STASH && U < 32 && (U & 1) == 0) {       // (v[r] is still the pre-activation: SiLU's last step comes at u = 2 r + 2)
        if constexpr (PX4) {                               // rows 4 k .. 4 k + 3 as one 16-byte store at u = 8 k (row 4 k changes at u = 8 k + 2)
            if constexpr ((U & 7) == 0) {
                const f32x4s q = {v[U >> 1], v[(U >> 1) + 1], v[(U >> 1) + 2], v[(U >> 1) + 3]};
                __builtin_nontemporal_store(q, reinterpret_cast<f32x4s*>(e.zst - c.lane) + c.lane + (U >> 3) * 64);
            }
        } else STASH_ST(e.zst[(U >> 1) * REGF], v[U >> 1]);
    }
#ifndef X6W_PROBE_NO_SILU            // (timing probe: identity activation)
    if constexpr (U < 32) {
        constexpr int r = U >> 1;
        if constexpr ((U & 1) == 0) e.rr[r] = __builtin_amdgcn_exp2f(-1.44269504088896341f * v[r]);
        else e.rr[r] = __builtin_amdgcn_rcpf(1.f + e.rr[r]);
    }
    if constexpr (U >= 2 && U <= 32 && (U & 1) == 0) { constexpr int r = (U - 2) >> 1; v[r] = v[r] * e.rr[r]; }
#endif
    if constexpr (LAST) {
        const float4* wl = c.w5l + ((m2 >> 1) * 2 + c.h) * 32;
        if constexpr (U == 0) {
#pragma unroll
            for (int o = 0; o < 6; ++o) e.p[o] = 0.f;
        }
        // 48 FMA pairs g = 3 r + j (j = 0: outputs 0, 1; 1: outputs 2, 3; 2: outputs 4, 5) over slots 3..45, two pairs in every eighth slot: slot(g) = 3 + g - g / 8
        // (>= 2 r + 3: activation r is complete); the 48-slot form evens out what 38 slots carried (2.69 k cycles for such a tile against 2.3 k for one with a split)
        if constexpr (U == 1) e.wa = wl[0];
        if constexpr (U == 2) e.wb = wl[1];
        x6w_for(std::make_integer_sequence<int, 48>{}, [&](auto Gc) {
            constexpr int g = decltype(Gc)::value, r = g / 3, j = g % 3;
            if constexpr (3 + g - g / 8 == U) {
                if constexpr (j == 0) { e.p[0] = __builtin_fmaf(v[r], e.wa.x, e.p[0]); e.p[1] = __builtin_fmaf(v[r], e.wa.y, e.p[1]); }
                if constexpr (j == 1) {
                    e.p[2] = __builtin_fmaf(v[r], e.wa.z, e.p[2]); e.p[3] = __builtin_fmaf(v[r], e.wa.w, e.p[3]);
                    if constexpr (r < 15) e.wa = wl[2 * (r + 1)];
                }
                if constexpr (j == 2) {
                    e.p[4] = __builtin_fmaf(v[r], e.wb.x, e.p[4]); e.p[5] = __builtin_fmaf(v[r], e.wb.y, e.p[5]);
                    if constexpr (r < 15) e.wb = wl[2 * (r + 1) + 1];
                }
            }
        });
        if constexpr (U == 46) {
#pragma unroll
            for (int o = 0; o < 3; ++o) e.p[o] = x6w_add_halves(e.p[o]);
        }
        if constexpr (U == 47) {
#pragma unroll
            for (int o = 3; o < 6; ++o) e.p[o] = x6w_add_halves(e.p[o]);
#pragma unroll
            for (int o = 0; o < 6; ++o) e.out[o] += e.p[o];
        }
    } else {
        if constexpr (U >= 5 && U <= 36) {
            constexpr int p = (U - 5) >> 2, part = (U - 5) & 3;
#ifdef X6W_PROBE_NO_SPLIT            // (timing probe: one instruction per pair instead of eleven)
            if constexpr (part == 0) {
                e.pk[0][p] = __builtin_amdgcn_perm(__float_as_uint(v[2 * p + 1]), __float_as_uint(v[2 * p]), 0x07060302u);
                e.pk[1][p] = e.pk[0][p]; e.pk[2][p] = e.pk[0][p];
            }
#else
            if constexpr (part == 0) {
                const unsigned ua = __float_as_uint(v[2 * p]), ub = __float_as_uint(v[2 * p + 1]);
                e.pk[0][p] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
                e.rr[2 * p] = v[2 * p] - __uint_as_float(ua & 0xffff0000u);
            } else if constexpr (part == 1) {
                e.rr[2 * p + 1] = v[2 * p + 1] - __uint_as_float(__float_as_uint(v[2 * p + 1]) & 0xffff0000u);
            } else if constexpr (part == 2) {
                const unsigned va = __float_as_uint(e.rr[2 * p]), vb = __float_as_uint(e.rr[2 * p + 1]);
                e.pk[1][p] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
                e.rr[2 * p] = e.rr[2 * p] - __uint_as_float(va & 0xffff0000u);
            } else {
                const float sb = e.rr[2 * p + 1] - __uint_as_float(__float_as_uint(e.rr[2 * p + 1]) & 0xffff0000u);
                e.pk[2][p] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(e.rr[2 * p]), 0x07060302u);
            }
#endif
        }
        if constexpr (U == 22 || U == 37) {
            constexpr int k = U == 22 ? 0 : 1;
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                const u32x4w q = {e.pk[term][4 * k], e.pk[term][4 * k + 1], e.pk[term][4 * k + 2], e.pk[term][4 * k + 3]};
                c.ob[((size_t)(m2 + k) * 3 + term) * 64] = __builtin_bit_cast(b8_t, q);
            }
        }
    }
}
template <int U0, int N, bool LAST, int STASH = 0, bool PX4 = false>
__device__ __forceinline__ void x6w_micros(const X6W& c, float (&v)[16], X6WEpi& e, int m2) {
    x6w_for(std::make_integer_sequence<int, N>{}, [&](auto Uc) {
        constexpr int U = U0 + decltype(Uc)::value;
        if constexpr (U < (LAST ? 48 : 38)) x6w_micro<U, LAST, STASH, PX4>(c, v, e, m2);
    });
}
__device__ __forceinline__ void x6w_load_in(const X6W& c, b8_t (&in)[8][3], int s) {
    in[s][0] = c.ob[((size_t)s * 3 + 0) * 64]; in[s][1] = c.ob[((size_t)s * 3 + 1) * 64]; in[s][2] = c.ob[((size_t)s * 3 + 2) * 64];
}

// MFMA j (0..5) of K step s: x6_step's order; the first K step of a row tile starts a1 from the constant 0 (a0 holds the bias)
template <int J, bool FIRST>
__device__ __forceinline__ void x6w_mfma(const b8_t& A1, const b8_t& A2, const b8_t& A3, const b8_t (&B)[3], f32x16& a0, f32x16& a1) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (J == 0) a0 = MFMA16B(A1, B[0], a0);
    else if constexpr (J == 1) a1 = MFMA16B(A1, B[1], FIRST ? zero : a1);
    else if constexpr (J == 2) a1 = MFMA16B(A2, B[1], a1);
    else if constexpr (J == 3) a1 = MFMA16B(A2, B[0], a1);
    else if constexpr (J == 4) a1 = MFMA16B(A1, B[2], a1);
    else a1 = MFMA16B(A3, B[0], a1);
}

// KIND of a row tile: 0 layer-0 tile (2 K steps on X0; 3-4 micro-slots per MFMA slot) | 1 first tile of a hidden layer (the pending epilogue is the
// previous layer's last tile: its 38 micro-slots in slots 0..29, its K steps 6, 7 pulled into `in` at slot 30, read at slot 36) | 2 middle tile |
// 3 last tile of a hidden layer that has a successor (releases in[s] K step by K step and pulls the next layer's input in behind) | 4 last tile
// of the last hidden layer
template <int E0, int KIND, int LROW, bool HAVE_PE, bool PE_LAST, int STASH = 0>
__device__ __forceinline__ void x6w_tile(X6W& c, b8_t (&A1)[X6W_RING], b8_t (&A2)[X6W_RING], b8_t (&A3)[X6W_RING], const b8_t (&X0)[2][3], b8_t (&in)[8][3],
                                         float (&pv)[16], X6WEpi& e, int pm2, float (&nv)[16], f32x16& bias) {
    constexpr int NS = KIND == 0 ? 2 : 8;
    f32x16 a0 = bias, a1;                  // the bias rows of this tile were read from LDS a tile ago
    x6w_for(std::make_integer_sequence<int, NS * 6>{}, [&](auto Ic) {
        constexpr int I = decltype(Ic)::value, s = I / 6, j = I % 6, en = E0 + s;
        if constexpr (KIND == 0) x6w_mfma<j, s == 0>(A1[en % X6W_RING], A2[en % X6W_RING], A3[en % X6W_RING], X0[s], a0, a1);
        else x6w_mfma<j, s == 0>(A1[en % X6W_RING], A2[en % X6W_RING], A3[en % X6W_RING], in[s], a0, a1);
        // the MFMA first in its slot: left to itself the scheduler puts it last in one region and first in the next - two MFMAs back to back behind
        // two epilogue slots in a row (round 6: -1.2 % per evaluation)
        __builtin_amdgcn_sched_barrier(0);
        // the stream WRAPS: behind the last tile's K steps the ring takes entries 0..6 of the next evaluation (136 = 17 x 8: the slots they belong in), in the
        // MFMAs' shadow and a whole uncovered tail ahead of their use (the last evaluation's are wasted and harmless)
        if constexpr (j == 5 && en + X6W_RING - 1 < X6W_ENTRIES + X6W_RING - 1) {
            constexpr int x = en + X6W_RING - 1;
            if constexpr ((x & 3) == 0) {
#ifndef X6W_PROBE_SAME_ENTRIES      // (timing probe: every wave re-reads the first four entries - 12 KB that stay in the CU's L1)
#ifdef X6W_BUFLD
                c.so += (x == X6W_ENTRIES) ? -(X6W_ENTRIES / 4 - 1) * 4096 : 4096;
#else
                constexpr int adv = (x == X6W_ENTRIES) ? -(X6W_ENTRIES / 4 - 1) * 256 : 256;
                c.W1 += adv; c.W2 += adv; c.W3 += adv;
#endif
#endif
                X6W_OPAQUE(c);
            }
#if defined(X6W_PROBE_QUARTER_LOADS)     // (timing probe: a quarter of the weight loads; the other ring slots keep what they hold)
            if constexpr ((x & 3) == 0) { A1[x % X6W_RING] = X6W_LD(c, W1, 0); A2[x % X6W_RING] = X6W_LD(c, W2, 0); A3[x % X6W_RING] = X6W_LD(c, W3, 0); }
#elif defined(X6W_PROBE_LDS_A)           // (timing probe: the A operands from LDS - this wave's output buffer, garbage - instead of L1)
            A1[x % X6W_RING] = c.ob[((x & 7) * 3 + 0) * 64]; A2[x % X6W_RING] = c.ob[((x & 7) * 3 + 1) * 64]; A3[x % X6W_RING] = c.ob[((x & 7) * 3 + 2) * 64];
#else
            A1[x % X6W_RING] = X6W_LD(c, W1, x & 3); A2[x % X6W_RING] = X6W_LD(c, W2, x & 3); A3[x % X6W_RING] = X6W_LD(c, W3, x & 3);
#endif
        }
#ifndef X6W_PROBE_NO_EPILOGUE          // (timing probe; NOT the MFMA stream alone: without the epilogue the earlier tiles' MFMAs are dead code and go too)
        if constexpr (HAVE_PE) {
            if constexpr (KIND == 0) {                        // 12 slots: 4 micro-slots in the first two, 3 in the others
                if constexpr (I < 2) x6w_micros<4 * I, 4, PE_LAST, STASH, (STASH == 2 && LROW >= 32 && LROW - 32 < 512)>(c, pv, e, pm2);
                else x6w_micros<8 + 3 * (I - 2), 3, PE_LAST, STASH, (STASH == 2 && LROW >= 32 && LROW - 32 < 512)>(c, pv, e, pm2);
            } else if constexpr (KIND == 1) {                 // done by slot 29: two micro-slots in each of the first eight
                if constexpr (I < 8) x6w_micros<2 * I, 2, PE_LAST, STASH, (STASH == 2 && LROW >= 32 && LROW - 32 < 512)>(c, pv, e, pm2);
                else if constexpr (I < 30) x6w_micros<I + 8, 1, PE_LAST, STASH, (STASH == 2 && LROW >= 32 && LROW - 32 < 512)>(c, pv, e, pm2);
            } else if constexpr (I < (PE_LAST ? 48 : 38)) x6w_micros<I, 1, PE_LAST, STASH, (STASH == 2 && LROW >= 32 && LROW - 32 < 512)>(c, pv, e, pm2);
        }
#endif
        if constexpr (KIND == 1 && I == 30) { x6w_load_in(c, in, 6); x6w_load_in(c, in, 7); }
        if constexpr (I == 1 && LROW + 32 < 640) {     // the next row tile's bias (behind the first MFMA, which has just consumed this one's)
#pragma unroll
            for (int r = 0; r < 16; ++r) bias[r] = c.lb[LROW + 32 + (r & 3) + 8 * (r >> 2) + 4 * c.h];
        }
        if constexpr (KIND == 3) {
            if constexpr (j == 5 && s < 4) x6w_load_in(c, in, s);
            if constexpr (I == 40) { x6w_load_in(c, in, 4); x6w_load_in(c, in, 5); }      // (K step 5 of the output is written in micro-slot 37)
        }
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int r = 0; r < 16; ++r) nv[r] = a1[r] + a0[r];
    if constexpr (STASH) {                     // the fp32 pre-activations, rows (l * 64 + 16 m + r) of the evaluation's stash: k_rk2_split_uni<STASH>'s layout
        if constexpr (KIND == 4) {             // (the evaluation's last tile has no successor whose MFMAs could cover its stores)
#pragma unroll
            for (int r = 0; r < 16; ++r) STASH_ST(c.zp[r * REGF], nv[r]);
        } else e.zst = c.zp;                   // round 6: stored from the epilogue's micro-slots (sixteen stores in a row here were 400 uncovered cycles per tile)
        c.zp += 16 * REGF;
        asm volatile("" : "+v"(c.zp));
    }
    X6W_STAMP(c, E0 < 8 ? 1 + E0 / 2 : 5 + (E0 - 8) / 8);
    __builtin_amdgcn_sched_barrier(0);
}

// one evaluation of the net for the wave's 32 points
// the first seven entries of the weight stream into the ring at kernel start; every evaluation then requests its successor's behind its own last K steps
// (x6w_tile: the stream wraps) - requested at the head of the evaluation the first tile waited 0.6-1 k cycles for them.
struct X6WRing { b8_t A1[X6W_RING], A2[X6W_RING], A3[X6W_RING]; };
__device__ __forceinline__ void x6w_prime(const X6W& c0, X6WRing& R) {
    X6W c = c0;
    X6W_OPAQUE(c);
#pragma unroll
    for (int en = 0; en < X6W_RING - 1; ++en) {
#ifdef X6W_BUFLD
        if (en == 4) c.so += 4096;
#else
        if (en == 4) { c.W1 += 256; c.W2 += 256; c.W3 += 256; }
#endif
        R.A1[en] = X6W_LD(c, W1, en & 3); R.A2[en] = X6W_LD(c, W2, en & 3); R.A3[en] = X6W_LD(c, W3, en & 3);
    }
}
template <int STASH = 0>
__device__ __forceinline__ void velnet_x6w(const X6W& c0, X6WRing& R, const float4& q, float (&out6)[6], float* zst = nullptr, float* x0st = nullptr) {
    X6W c = c0;
    X6W_STAMP(c, 0);
    c.zp = STASH ? zst + c0.lane : nullptr;                    // (the stream pointers run through one evaluation)
    X6W_OPAQUE(c);
    b8_t X0[2][3], in[8][3];
    b8_t (&A1)[X6W_RING] = R.A1, (&A2)[X6W_RING] = R.A2, (&A3)[X6W_RING] = R.A3;
    X6WEpi e;
    float va[16], vb[16];
    {
        float x0[16];
#ifdef X6W_PROBE_NO_ENC              // (timing probe: no sines / cosines)
#pragma unroll
        for (int k = 0; k < 16; ++k) x0[k] = q.x + (float)k * q.y;
#else
        vel_encode_slots(q, c.h, x0);
#endif
        if (STASH) stash_store<16>(x0st, c.lane, x0);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            unsigned p1[4], p2[4], p3[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xa = x0[8 * k + 2 * j], xb = x0[8 * k + 2 * j + 1];
                const unsigned ua = __float_as_uint(xa), ub = __float_as_uint(xb);
                p1[j] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
                const float ra = xa - __uint_as_float(ua & 0xffff0000u), rb = xb - __uint_as_float(ub & 0xffff0000u);
                const unsigned wa = __float_as_uint(ra), wb = __float_as_uint(rb);
                p2[j] = __builtin_amdgcn_perm(wb, wa, 0x07060302u);
                const float sa = ra - __uint_as_float(wa & 0xffff0000u), sb = rb - __uint_as_float(wb & 0xffff0000u);
                p3[j] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
            }
            const u32x4w q1 = {p1[0], p1[1], p1[2], p1[3]}, q2 = {p2[0], p2[1], p2[2], p2[3]}, q3 = {p3[0], p3[1], p3[2], p3[3]};
            X0[k][0] = __builtin_bit_cast(b8_t, q1); X0[k][1] = __builtin_bit_cast(b8_t, q2); X0[k][2] = __builtin_bit_cast(b8_t, q3);
        }
    }
    // (the ring holds entries 0..6: x6w_prime; the stream pointers continue behind them)
#ifdef X6W_BUFLD
    c.so += 4096;
#else
    c.W1 += 256; c.W2 += 256; c.W3 += 256;
#endif
    f32x16 bias;
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = c.lb[(r & 3) + 8 * (r >> 2) + 4 * c.h];
#pragma unroll
    for (int o = 0; o < 6; ++o) e.out[o] = c.lb[128 * 5 + o];
    // layer 0 (28 -> 128): four row tiles of two K steps
    x6w_tile<0, 0, 0, false, false, STASH>(c, A1, A2, A3, X0, in, va, e, 0, va, bias);
    x6w_tile<2, 0, 32, true, false, STASH>(c, A1, A2, A3, X0, in, va, e, 0, vb, bias);
    x6w_tile<4, 0, 64, true, false, STASH>(c, A1, A2, A3, X0, in, vb, e, 2, va, bias);
    x6w_tile<6, 0, 96, true, false, STASH>(c, A1, A2, A3, X0, in, va, e, 4, vb, bias);
#pragma unroll
    for (int s = 0; s < 6; ++s) x6w_load_in(c, in, s);
    // layers 1..3: first | middle | middle | last-with-successor
    x6w_tile<8, 1, 128, true, false, STASH>(c, A1, A2, A3, X0, in, vb, e, 6, va, bias);
    x6w_tile<16, 2, 160, true, false, STASH>(c, A1, A2, A3, X0, in, va, e, 0, vb, bias);
    x6w_tile<24, 2, 192, true, false, STASH>(c, A1, A2, A3, X0, in, vb, e, 2, va, bias);
    x6w_tile<32, 3, 224, true, false, STASH>(c, A1, A2, A3, X0, in, va, e, 4, vb, bias);
    x6w_tile<40, 1, 256, true, false, STASH>(c, A1, A2, A3, X0, in, vb, e, 6, va, bias);
    x6w_tile<48, 2, 288, true, false, STASH>(c, A1, A2, A3, X0, in, va, e, 0, vb, bias);
    x6w_tile<56, 2, 320, true, false, STASH>(c, A1, A2, A3, X0, in, vb, e, 2, va, bias);
    x6w_tile<64, 3, 352, true, false, STASH>(c, A1, A2, A3, X0, in, va, e, 4, vb, bias);
    x6w_tile<72, 1, 384, true, false, STASH>(c, A1, A2, A3, X0, in, vb, e, 6, va, bias);
    x6w_tile<80, 2, 416, true, false, STASH>(c, A1, A2, A3, X0, in, va, e, 0, vb, bias);
    x6w_tile<88, 2, 448, true, false, STASH>(c, A1, A2, A3, X0, in, vb, e, 2, va, bias);
    x6w_tile<96, 3, 480, true, false, STASH>(c, A1, A2, A3, X0, in, va, e, 4, vb, bias);
    // layer 4: its activations go to LDS as floats (rows [m][4][lane] float4 of the same buffer, free once in[6..7] are in registers)
    x6w_tile<104, 1, 512, true, false, STASH>(c, A1, A2, A3, X0, in, vb, e, 6, va, bias);
    x6w_tile<112, 2, 544, true, true, STASH>(c, A1, A2, A3, X0, in, va, e, 0, vb, bias);
    x6w_tile<120, 2, 576, true, true, STASH>(c, A1, A2, A3, X0, in, vb, e, 2, va, bias);
    x6w_tile<128, 4, 608, true, true, STASH>(c, A1, A2, A3, X0, in, va, e, 4, vb, bias);
    // ---- 128 -> 6: fp32 FMAs in velnet_x6's order (per row tile: a chain over its 16 activations, the two lane halves added, then the tiles in order).
    // Row tiles 0..2 were summed in their epilogues' micro-slots (x6w_micro<LAST>); the last one has no MFMAs behind it
#ifdef X6W_PROBE_NO_OUT              // (timing probe: no output layer)
#pragma unroll
    for (int o = 0; o < 6; ++o) out6[o] = e.out[o];
    out6[0] += vb[0]; out6[1] += vb[5];
    return;
#endif
    {
        float p[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float4* wl = c.w5l + (3 * 2 + c.h) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 wa = wl[2 * r], wb = wl[2 * r + 1];
            const float av = act_f<1>(vb[r]);
            p[0] = __builtin_fmaf(av, wa.x, p[0]); p[1] = __builtin_fmaf(av, wa.y, p[1]); p[2] = __builtin_fmaf(av, wa.z, p[2]);
            p[3] = __builtin_fmaf(av, wa.w, p[3]); p[4] = __builtin_fmaf(av, wb.x, p[4]); p[5] = __builtin_fmaf(av, wb.y, p[5]);
        }
#pragma unroll
        for (int o = 0; o < 6; ++o) out6[o] = e.out[o] + x6w_add_halves(p[o]);
    }
    X6W_STAMP(c, 21);
}

// the recurrence of k_rk2_x6 (vel_x6.hip), one wave per tile
__global__ __launch_bounds__(WG_THREADS, 1) void k_rk2_x6w(X6Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lb = lds;
    float* w5f = lb + 6 * 128;
    b8_t* obase = reinterpret_cast<b8_t*>(w5f + 4 * 2 * 16 * 8);
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = a.count ? *a.count : (int)a.n_direct;
    if ((int)blockIdx.x * 4 * TILE >= count) return;
    for (int k = threadIdx.x; k < 6 * 128; k += WG_THREADS) lb[k] = (k & 127) < (k < 640 ? 128 : 6) ? a.f.vb[k >> 7][k & 127] : 0.f;
    for (int k = threadIdx.x; k < 4 * 2 * 16 * 8; k += WG_THREADS) {
        const int o = k & 7, r = (k >> 3) & 15, hh = (k >> 7) & 1, ww = k >> 8;
        w5f[k] = o < 6 ? a.f.vW[5][o * 128 + 32 * ww + (r & 3) + 8 * (r >> 2) + 4 * hh] : 0.f;
    }
    __syncthreads();
    const int tile = blockIdx.x * 4 + wv;
    if (tile * TILE >= count) return;                      // (no barrier behind this point: the waves are independent)
    const int i = tile * TILE + (lane & 31);
    const bool active = i < count;
    const int n = active ? (a.list ? a.list[i] : i) : 0;
    const float4 q0 = active ? a.xw[n] : zero4();
    float x = q0.x, y = q0.y, z = q0.z;
    const float zw = q0.w;
    const int ti = a.pt_by_list ? n : i;
    float tcur = active ? a.pt_t[ti] : 0.f;
    float off = active ? tcur - a.pt_base[ti] : 0.f;
    X6W c; c.stamp = 0;
    const b8_t* img = reinterpret_cast<const b8_t*>(a.img);
    c.W1 = X6W_PTR(img, lane); c.W2 = X6W_PTR(img + X6_H8, lane); c.W3 = X6W_PTR(img + 2 * X6_H8, lane); c.vo = (unsigned)lane * 16u;
#ifdef X6W_BUFLD
    c.so = 0;
    c.RW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<b8_t*>(img), 0, 0x7fffffff, 0x00020000);
    c.RW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<b8_t*>(img + X6_H8), 0, 0x7fffffff, 0x00020000);
    c.RW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<b8_t*>(img + 2 * X6_H8), 0, 0x7fffffff, 0x00020000);
#endif
    c.lb = lb; c.w5l = reinterpret_cast<const float4*>(w5f); c.ob = obase + (size_t)wv * X6W_OB_H8 + lane; c.lane = lane; c.h = h;
    X6WRing ring;
    x6w_prime(c, ring);
#pragma unroll 1
    for (int s = 0; s < a.max_steps; ++s) {
        const bool live = active && fabsf(off) > 0.f;
        const float mm = fminf(fabsf(off), a.dt_max);
        const float dt = off > 0.f ? mm : (off < 0.f ? -mm : 0.f);
        if (!__any(live)) break;
        const float hdt = 0.5f * dt;
        float px = x, py = y, pz = z;
        float o6[6];
        // two evaluations through ONE copy of the network code (it is ~30 KB of instructions)
#pragma unroll 1
        for (int ev = 0; ev < 2; ++ev) {
            const float4 q = make_float4(px, py, pz, ev ? tcur - hdt : tcur);
#ifdef X6W_TIMING
            c.stamp = (blockIdx.x == 700 && wv == 1 && s == 2 && ev == 0) ? 1 : 0;
#endif
            velnet_x6w(c, ring, q, o6);
            if (ev == 0) {
                float v1[3];
                vel_from_w(o6, x, y, z, v1);
                if (gated_out(a.f, x, y, z)) { v1[0] = v1[1] = v1[2] = 0.f; }
                px = x - hdt * v1[0]; py = y - hdt * v1[1]; pz = z - hdt * v1[2];
            }
        }
        float v2[3];
        vel_from_w(o6, px, py, pz, v2);
        if (gated_out(a.f, px, py, pz)) { v2[0] = v2[1] = v2[2] = 0.f; }
        const float nx = x - dt * v2[0], ny = y - dt * v2[1], nz = z - dt * v2[2];
        const bool rej = a.f.gate_sur && gated_out(a.f, nx, ny, nz);   // tensorf_keyframe.py:603-605
        if (live && !rej) { x = nx; y = ny; z = nz; }
        if (live) { off = off - dt; tcur = tcur - dt; }
    }
#ifdef X6W_TIMING
    if (blockIdx.x == 700 && wv == 1 && lane == 0) {
        printf("[x6w timing] tiles (layer 0: 4, layers 1-4: 4 each), then the last row tile's sums:");
        for (int k = 1; k <= 21; ++k) printf(" %llu", x6w_ts[k] - x6w_ts[k - 1]);
        printf(" | total %llu\n", x6w_ts[21] - x6w_ts[0]);
    }
#endif
    if (active && h == 0) {
        if (a.xout3) { float* o = a.xout3 + 3 * (size_t)n; o[0] = x; o[1] = y; o[2] = z; }
        else a.xw[n] = make_float4(x, y, z, zw);
    }
}


// (per device: hipFuncSetAttribute applies to the device that is current - one process per GPU never sees a second one, a host that drives
// several devices from one process does; common.h: DeviceOnce)
template <typename K>
static int x6w_set_lds(K kernel) {
    HIPCK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, X6W_LDS_BYTES));
    return 0;
}
// ---------------------------------------------------------------- render warp (uniform schedule; k_rk2_x6_uni of vel_x6.hip, one wave per tile)
// A workgroup = four consecutive tiles = one 128-sample group of the stash geometry (training: the adjoint walks whole groups)
template <int STASH>
__global__ __launch_bounds__(WG_THREADS, 1) void k_rk2_x6w_uni(X6UniArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lb = lds;
    float* w5f = lb + 6 * 128;
    b8_t* obase = reinterpret_cast<b8_t*>(w5f + 4 * 2 * 16 * 8);
    const Rk2Args& ra = a.r;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = *ra.count;
    if ((int)blockIdx.x * WG_SAMPLES >= count) return;
    for (int k = threadIdx.x; k < 6 * 128; k += WG_THREADS) lb[k] = (k & 127) < (k < 640 ? 128 : 6) ? ra.f.vb[k >> 7][k & 127] : 0.f;
    for (int k = threadIdx.x; k < 4 * 2 * 16 * 8; k += WG_THREADS) {
        const int o = k & 7, r = (k >> 3) & 15, hh = (k >> 7) & 1, ww = k >> 8;
        w5f[k] = o < 6 ? ra.f.vW[5][o * 128 + 32 * ww + (r & 3) + 8 * (r >> 2) + 4 * hh] : 0.f;
    }
    __syncthreads();
    const size_t tile = (size_t)blockIdx.x * 4 + wv;
    if (!STASH && (int)(tile * TILE) >= count) return;
    const int idx = (int)tile * TILE + (lane & 31);
    const bool active = idx < count;
    const int n = active ? ra.list[idx] : 0;
    const float4 q0 = active ? ra.xw[n] : zero4();
    float x = q0.x, y = q0.y, z = q0.z;
    const float zw = q0.w;
    X6W c; c.stamp = 0;
    const b8_t* img = reinterpret_cast<const b8_t*>(a.img);
    c.W1 = X6W_PTR(img, lane); c.W2 = X6W_PTR(img + X6_H8, lane); c.W3 = X6W_PTR(img + 2 * X6_H8, lane); c.vo = (unsigned)lane * 16u;
#ifdef X6W_BUFLD
    c.so = 0;
    c.RW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<b8_t*>(img), 0, 0x7fffffff, 0x00020000);
    c.RW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<b8_t*>(img + X6_H8), 0, 0x7fffffff, 0x00020000);
    c.RW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<b8_t*>(img + 2 * X6_H8), 0, 0x7fffffff, 0x00020000);
#endif
    c.lb = lb; c.w5l = reinterpret_cast<const float4*>(w5f); c.ob = obase + (size_t)wv * X6W_OB_H8 + lane; c.lane = lane; c.h = h; c.zp = nullptr;
    const int nsteps = ra.sched ? __float_as_int(ra.sched[2]) : ra.nsteps;
    X6WRing ring;
    x6w_prime(c, ring);
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        const float dt = RK_DT(ra, s), tcur = RK_TC(ra, s), hdt = 0.5f * dt;
        float px = x, py = y, pz = z;
        float o6[6], w1[6];
        bool g1 = false;
#pragma unroll 1
        for (int ev = 0; ev < 2; ++ev) {
            const size_t e = (size_t)(2 * s + ev) * ra.cap_tiles + tile;
            const float4 q = make_float4(px, py, pz, ev ? tcur - hdt : tcur);
            velnet_x6w<STASH>(c, ring, q, o6, STASH ? ra.zst + e * (VEL_Z_REGS * REGF) : nullptr, STASH ? ra.x0st + e * (VEL_X0_REGS * REGF) : nullptr);
            if (ev == 0) {
                float v1[3];
#pragma unroll
                for (int k = 0; k < 6; ++k) w1[k] = o6[k];
                vel_from_w(w1, x, y, z, v1);
                g1 = gated_out(ra.f, x, y, z);
                if (g1) { v1[0] = v1[1] = v1[2] = 0.f; }
                px = x - hdt * v1[0]; py = y - hdt * v1[1]; pz = z - hdt * v1[2];
            }
        }
        float v2[3];
        vel_from_w(o6, px, py, pz, v2);
        const bool g2 = gated_out(ra.f, px, py, pz);
        if (g2) { v2[0] = v2[1] = v2[2] = 0.f; }
        const float nx = x - dt * v2[0], ny = y - dt * v2[1], nz = z - dt * v2[2];
        const bool rej = ra.f.gate_sur && gated_out(ra.f, nx, ny, nz);   // tensorf_keyframe.py:603-605
        if (STASH && active && h == 0) {
            float* rc = ra.rec + (size_t)s * RK_NF * ra.cap + idx;
            rc[0 * ra.cap] = x; rc[1 * ra.cap] = y; rc[2 * ra.cap] = z;
            rc[3 * ra.cap] = px; rc[4 * ra.cap] = py; rc[5 * ra.cap] = pz;
#pragma unroll
            for (int k = 0; k < 6; ++k) { rc[(6 + k) * ra.cap] = w1[k]; rc[(12 + k) * ra.cap] = o6[k]; }
            rc[18 * ra.cap] = __int_as_float((g1 ? 1 : 0) | (g2 ? 2 : 0) | (rej ? 4 : 0));
        }
        if (active && !rej) { x = nx; y = ny; z = nz; }
    }
    if (active && h == 0) ra.xw[n] = make_float4(x, y, z, zw);
}

int launch_rk2_x6w_uni(const X6UniArgs& a, int64_t cap_samples, bool stash, hipStream_t st) {
    const int64_t groups = (cap_samples + WG_SAMPLES - 1) / WG_SAMPLES;
    if (groups <= 0) return 0;
    static DeviceOnce once;
    if (once.run([] { return (x6w_set_lds(k_rk2_x6w_uni<2>) || x6w_set_lds(k_rk2_x6w_uni<1>) || x6w_set_lds(k_rk2_x6w_uni<0>)) ? 1 : 0; })) return 1;
    if (stash && a.r.z_x4) hipLaunchKernelGGL(k_rk2_x6w_uni<2>, dim3((unsigned)groups), dim3(WG_THREADS), X6W_LDS_BYTES, st, a);
    else if (stash) hipLaunchKernelGGL(k_rk2_x6w_uni<1>, dim3((unsigned)groups), dim3(WG_THREADS), X6W_LDS_BYTES, st, a);
    else hipLaunchKernelGGL(k_rk2_x6w_uni<0>, dim3((unsigned)groups), dim3(WG_THREADS), X6W_LDS_BYTES, st, a);
    LAUNCHCK();
    return 0;
}

int launch_rk2_x6w(const X6Args& a, int64_t cap_points, hipStream_t st) {
    const int64_t tiles = (cap_points + TILE - 1) / TILE;
    if (tiles <= 0) return 0;
    static DeviceOnce once;
    if (once.run([] { return x6w_set_lds(k_rk2_x6w); })) return 1;
    hipLaunchKernelGGL(k_rk2_x6w, dim3((unsigned)((tiles + 3) / 4)), dim3(WG_THREADS), X6W_LDS_BYTES, st, a);
    LAUNCHCK();
    return 0;
}
