// pde_fuse.hip - adjoint of the PDE regulariser's Jacobian program WITH the velocity net's hidden-layer weight gradients formed in the
// same kernel (reference: the second-order backward of the functorch Jacobian inside NVFi.get_vel_loss, models/nvfi.py:68-84, through
// VelBasis.weight_net, models/velocity_field.py:58-67).
//
// k_pde_jet_bwd (pde_jet.hip) writes the layer gradients of all five columns (value, d/dx, d/dy, d/dz, d/dt) to the stash - 1 680 rows
// per 32-point tile - and k_wgrad_ring8 reads them back together with z / zd_j (the z image five times per layer) to contract
//     G_l = sum_points [ gz_l (x) SiLU(z_{l-1})  +  sum_j gzd_l^j (x) (SiLU'(z_{l-1}) zd_{l-1}^j) ].
// Here ONE persistent workgroup of TWELVE waves per CU does both, in the role split of vel_fuse.hip:
//
//   * waves 0-3 ("adjoint" waves, one per SIMD, raised priority): wave w owns rows [32w, 32w+32) of every layer.  The five columns of a
//     tile are processed ONE AT A TIME per layer (k_pde_jet_bwd keeps all five in 400 registers of a two-waves-per-SIMD workgroup; at
//     three waves per SIMD a wave has 168): phase (l, 0) is the value column's dgrad - its epilogue forms SiLU(z_l) (the layer input of
//     the value column), SiLU'(z_l), SiLU''(z_l) ONCE per layer and keeps the two derivatives in registers -, phases (l, 1..4) are the
//     tangent columns: gzd_l^j = SiLU' * ga_j to LDS, the tangent layer input SiLU' * zd_l^j to LDS, and the second-derivative correction
//     SiLU'' * zd_l^j * ga_j accumulated into the value column's gradient, which leaves for LDS with the last tangent column;
//   * waves 4-11 ("contraction" waves, two per SIMD) hold the 4 x 16 output tiles of the four 128 x 128 gradients for the whole launch
//     (128 accumulator registers) and contract the pair (layer gradient of column c at layer l+1, layer input of column c at layer l) one
//     phase behind the adjoint waves;
//   * one barrier per phase, 21 per tile.  LDS: six rotating images for the tangent columns' gradients (an image is written in phase t,
//     read as the dgrad's B operand in phase t+5 and contracted in phase t+6; there are four tangent writes in five phases), one for the
//     value column's (written with the last tangent column, read in the next two phases), two alternating images of layer inputs: 9 x 16.5 KB;
//   * at the end every workgroup writes one slab per hidden layer in k_wgrad_ring8's format: k_wgrad_reduce is unchanged.
//   * SECOND HALF of the launch: the ReLU acceleration net (a_weight_net, value column only).  A workgroup that has run out of weight_net
//     tiles writes its four slabs, clears the accumulators and takes acceleration-net tiles from a device-side queue: the same roles, five
//     phases per tile (three rotating gradient images), the four hidden-layer gradients of a_weight_net in the same registers.  A tile costs a
//     fifth of a weight_net tile, so the queue levels the one-tile imbalance of the first half (1 056 tiles over 256 workgroups are 4.1
//     rounds: 224 workgroups would idle for the fifth) - and the acceleration net's adjoint no longer needs a launch of its own at half occupancy.
// Still through the stash and k_wgrad_ring8: the two edge layers of both nets (28 -> 128: gz_0 is stored as before; 128 -> 6: the seed rows).
//
// Numerics: every ga / gzd is the number k_pde_jet_bwd forms (same operands, same K order); the value column's gradient adds the four
// corrections one after the other instead of pairwise, and a weight gradient is summed over points in another order than
// k_wgrad_ring8's - differences of the order of two fp32 summation orders.
#include <stdlib.h>
#include <stdio.h>
#include "common.h"
#include "vel.h"
#include "pde.h"
#include "fuse.h"

#define PF_THREADS 768
#define PF_NT 6                                       // rotating images of the tangent columns' layer gradients
#define PF_V PF_NT                                    // image of the value column's layer gradient
#define PF_Y0 (PF_NT + 1)                             // two alternating images of layer inputs
#define PF_IMAGES (PF_NT + 3)
#define PF_LDS_BYTES (PF_IMAGES * FUSE_XB * 16 + 64)   // + the two queue words of the acceleration-net half
#ifndef PF_P0_FIRST
#define PF_P0_FIRST 4                                 // tangent columns whose layer-4 rows phase 0 requests together with z_4
#endif

// -DPF_TIMING: workgroup 0 accumulates shader-clock intervals (adjoint wave 0: [0] phase 0 work, [1] value-phase dgrad, [2] value-phase
// epilogue, [3] tangent-phase dgrad, [4] tangent-phase epilogue, [8..12] the barrier waits behind them; contraction wave 0: [16] work,
// [17] barrier waits; [31] tiles)
#ifdef PF_TIMING
struct PfT { unsigned long long pt[32]; unsigned long long t0; };
#define PT_NOW() __builtin_amdgcn_s_memtime()
#define PT_ADD(T_, slot) do { const unsigned long long n_ = PT_NOW(); (T_).pt[slot] += n_ - (T_).t0; (T_).t0 = n_; } while (0)
#define PT_MFMA_DONE(acc) asm volatile("s_nop 0" :: "v"((acc)[0]))
#else
struct PfT { };
#define PT_ADD(T_, slot) do { } while (0)
#define PT_MFMA_DONE(acc) do { } while (0)
#endif

__device__ __forceinline__ int pf_count(const PdeFuseArgs& a) {
    const int64_t c = (int64_t)(*a.kcount) - a.first;
    return c <= 0 ? 0 : (c > a.cap ? (int)a.cap : (int)c);
}
// opaque_u (fuse.h) as a VOLATILE asm: the row-block bases of a tile are formed where they are used - left to loop-invariant code motion, the
// tile-independent halves of ~60 of them are hoisted out of the persistent loop, overflow the SGPR file and come back as v_readlane / scratch reloads
__device__ __forceinline__ gcfp pf_base(const float* p) { gcfp q = (gcfp)p; asm volatile("" : "+s"(q)); return q; }
__device__ __forceinline__ gfp pf_base(float* p) { gfp q = (gfp)p; asm volatile("" : "+s"(q)); return q; }
__device__ __forceinline__ int pf_inc(int s, int by) { s += by; return s >= PF_NT ? s - PF_NT : s; }

// ---------------------------------------------------------------- adjoint waves
struct PfA {
    float4* S;           // the nine images
    int w, lane, pos;    // pos: float4 index of this lane inside a row group (h * 33 + sample)
    bool x4;             // the forward's hidden-layer rows are x4 stash blocks
};

// sixteen stash rows of this wave: row-major (one dword per row and lane) or - the hidden layers the forward wrote as x4 blocks
// (engine.h: stash_st16_x4; PdeFuseArgs::x4) - four 16-byte accesses
__device__ __forceinline__ void pf_rows16(const float* base16, int lane, bool x4, float (&v)[16]) {
    if (x4) {
        const __attribute__((address_space(1))) f32x4s* q = (const __attribute__((address_space(1))) f32x4s*)pf_base(base16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4s t = __builtin_nontemporal_load(q + k * 64 + lane);
            v[4 * k] = t[0]; v[4 * k + 1] = t[1]; v[4 * k + 2] = t[2]; v[4 * k + 3] = t[3];
        }
    } else {
        gcfp zp = pf_base(base16);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = STASH_LD(zp[r * REGF + lane]);
    }
}

// dgrad of one column: acc = T^T fragment (registers) x layer gradient image (LDS)
__device__ __forceinline__ void pf_dgrad(const PfA& A, int img, const f32x4v (&wq)[16], f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float4* Xr = A.S + img * FUSE_XB + A.pos;
    float4 b = Xr[0], bn;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g + 1 < 16) bn = Xr[(g + 1) * 2 * FUSE_HR];
        const float a4[4] = {wq[g].x, wq[g].y, wq[g].z, wq[g].w};
        acc = MFMA32(a4[0], b.x, acc); acc = MFMA32(a4[1], b.y, acc); acc = MFMA32(a4[2], b.z, acc); acc = MFMA32(a4[3], b.w, acc);
        b = bn;
    }
}
// this wave's 16 registers of an image
__device__ __forceinline__ void pf_put(const PfA& A, int img, const float (&v)[16]) {
    float4* Xw = A.S + img * FUSE_XB + (4 * A.w) * 2 * FUSE_HR + A.pos;
#pragma unroll
    for (int k = 0; k < 4; ++k) Xw[k * 2 * FUSE_HR] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
}

// address of this wave's first float4 of an image (its four row groups follow at a stride of 2 * FUSE_HR)
__device__ __forceinline__ float4* pf_rows(const PfA& A, int img) { return A.S + img * FUSE_XB + (4 * A.w) * 2 * FUSE_HR + A.pos; }

// value column at layer L (3..0): reads image PF_V, writes the layer input SiLU(z_L) to image yimg; leaves SiLU', SiLU'' and the first term
// of the value column's gradient in registers
template <int L>
__device__ __forceinline__ void pf_value_phase(const PfA& A, const float* T, const f32x4v (&wq)[16], int yimg,
                                               float (&d1)[16], float (&d2)[16], float (&gzv)[16], PfT& TT) {
    float zr[16];
    pf_rows16(T + (size_t)(PDE_Z + L * 64 + 16 * A.w) * REGF, A.lane, A.x4, zr);
    f32x16 acc;
    pf_dgrad(A, PF_V, wq, acc);
    PT_MFMA_DONE(acc); PT_ADD(TT, 1);
    float4* Yw = pf_rows(A, yimg);
#pragma unroll
    for (int k = 0; k < 4; ++k) {                         // four registers at a time: the float4 leaves as soon as it is complete
        float a4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = 4 * k + c;
            const float z = zr[r], s = fast_sigmoid(z), oms = 1.f - s;
            a4[c] = z * s;
            d1[r] = s * (1.f + z * oms);
            d2[r] = s * oms * (2.f + z * (1.f - 2.f * s));
            gzv[r] = d1[r] * acc[r];
        }
        Yw[k * 2 * FUSE_HR] = make_float4(a4[0], a4[1], a4[2], a4[3]);
    }
    PT_ADD(TT, 2);
}

// tangent column J (1..4) at layer L (3..0): reads image rimg; writes gzd to image wimg (L > 0) or to the stash (L = 0: the A operand of
// the input layer's weight gradient), the tangent layer input to image yimg; LAST (J = 4): the value column's gradient leaves too, and the
// next layer's weights start their trip from L2 behind the last MFMA that reads the current ones
template <int L, bool LAST>
__device__ __forceinline__ void pf_tangent_phase(const PfA& A, const PdeFuseArgs& a, float* T, f32x4v (&wq)[16], int J, int rimg, int wimg, int yimg,
                                                 const float (&d1)[16], const float (&d2)[16], float (&gzv)[16], PfT& TT) {
    float zd[16];
    pf_rows16(T + (size_t)(PDE_ZD + 320 * (J - 1) + L * 64 + 16 * A.w) * REGF, A.lane, A.x4, zd);
    f32x16 acc;
    pf_dgrad(A, rimg, wq, acc);
    PT_MFMA_DONE(acc); PT_ADD(TT, 3);
    if (LAST && L >= 1) {
        asm volatile("" :: "v"(acc[0]));
        split_load16(a.t4[L] + (size_t)A.w * 16 * 64, A.lane, wq);
    }
    float4* Yw = pf_rows(A, yimg);
    float4* Xw = pf_rows(A, wimg);
    float4* Vw = pf_rows(A, PF_V);
    gfp gp = pf_base(T + (size_t)(PDE_GA + 336 * J + 16 * A.w) * REGF);
    gfp gv = pf_base(T + (size_t)(PDE_GA + 16 * A.w) * REGF);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float g4[4], a4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = 4 * k + c;
            const float ga = acc[r];
            g4[c] = d1[r] * ga;
            a4[c] = d1[r] * zd[r];
            gzv[r] = gzv[r] + d2[r] * zd[r] * ga;
        }
        Yw[k * 2 * FUSE_HR] = make_float4(a4[0], a4[1], a4[2], a4[3]);
        if (L > 0) {
            Xw[k * 2 * FUSE_HR] = make_float4(g4[0], g4[1], g4[2], g4[3]);
            if (LAST) Vw[k * 2 * FUSE_HR] = make_float4(gzv[4 * k], gzv[4 * k + 1], gzv[4 * k + 2], gzv[4 * k + 3]);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) STASH_ST(gp[(4 * k + c) * REGF + A.lane], g4[c]);
            if (LAST) {
#pragma unroll
                for (int c = 0; c < 4; ++c) STASH_ST(gv[(4 * k + c) * REGF + A.lane], gzv[4 * k + c]);
            }
        }
    }
    PT_ADD(TT, 4);
}

template <int L>
__device__ __forceinline__ void pf_layer(const PfA& A, const PdeFuseArgs& a, float* T, f32x4v (&wq)[16], int& rs,
                                         float (&d1)[16], float (&d2)[16], float (&gzv)[16], PfT& TT) {
    // phases tau = 1 + 5 (3 - L) + k; the layer inputs alternate between the two Y images by the parity of tau
    constexpr int tau0 = 1 + 5 * (3 - L);
    pf_value_phase<L>(A, T, wq, PF_Y0 + (tau0 & 1), d1, d2, gzv, TT);
    FUSE_BAR(); PT_ADD(TT, 9);
#pragma unroll 1
    for (int k = 1; k <= 3; ++k) {
        pf_tangent_phase<L, false>(A, a, T, wq, k, rs, pf_inc(rs, 4), PF_Y0 + ((tau0 + k) & 1), d1, d2, gzv, TT);
        rs = pf_inc(rs, 1);
        FUSE_BAR(); PT_ADD(TT, 11);
    }
    pf_tangent_phase<L, true>(A, a, T, wq, 4, rs, pf_inc(rs, 4), PF_Y0 + ((tau0 + 4) & 1), d1, d2, gzv, TT);
    rs = pf_inc(rs, 1);
    FUSE_BAR(); PT_ADD(TT, 12);
}

__device__ __forceinline__ void pf_role_adjoint(const PdeFuseArgs& a, float4* S, int w, int lane, int ntiles) {
    PfA A; A.S = S; A.w = w; A.lane = lane; A.x4 = a.x4 != 0;
    const int h = lane >> 5, j = lane & 31;
    A.pos = h * FUSE_HR + j;
    const int G = gridDim.x;
    const size_t cs = a.cap;
    int rs = 0;                                           // tangent image the next tangent phase reads
    f32x4v wq[16];
    float d1[16], d2[16], gzv[16];
    PfT TT;
#ifdef PF_TIMING
    for (int k = 0; k < 32; ++k) TT.pt[k] = 0;
    TT.t0 = PT_NOW();
#endif
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        float* T = a.stash + (size_t)tile * PDE_TILE_ROWS * REGF;
        const int i = tile * TILE + j;
        const bool ok = i < (int)a.cap;
        // per-tile opaque copies of the launch-invariant bases: everything derived from them (20 seed rows, 17 fragment quarters) is formed
        // where it is used instead of being hoisted out of the persistent loop into SGPRs the kernel does not have
        PdeFuseArgs al = a;
        {
            gcfp sp_ = pf_base(a.seeds); al.seeds = (const float*)sp_;
#pragma unroll
            for (int l = 1; l <= 5; ++l) { gcfp tp_ = pf_base(reinterpret_cast<const float*>(a.t4[l])); al.t4[l] = reinterpret_cast<const float4*>((const float*)tp_); }
        }
        size_t csl = cs; asm volatile("" : "+s"(csl));
        // ---- phase 0: the five columns at layer 4 (6 -> 128: four MFMAs per column), all from registers
        {
            f32x4v w5;
            {
                gcf4p b5 = (gcf4p)(al.t4[5] + (size_t)w * 64);
                asm("" : "+s"(b5));
                w5 = b5[lane];
            }
            // adjoint seeds of the 6 outputs per column, in D-layout registers 0..3: lanes of half h hold outputs 4h .. 4h+3 (k_pde_seeds)
            // (loaded unconditionally: row sbase + 4h + k off a wave-uniform base per (column, k) with ONE lane offset - a predicated load
            // becomes a branch per value, a lane-dependent row a hoisted and spilled offset per value -; the two rows past a column's six that
            // the upper half reads belong to the next column (36 rows are allocated), they and the points beyond the capacity are zeroed afterwards)
            float sd[5][4];
            {
                const int lo = (ok ? i : 0) + (h ? 4 * (int)csl : 0);
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const int sbase = c == 0 ? 0 : 6 * c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gcfp sp = pf_base(al.seeds + (size_t)(sbase + k) * csl);
                        sd[c][k] = sp[lo];
                    }
                }
#pragma unroll
                for (int c = 0; c < 5; ++c)
#pragma unroll
                    for (int k = 0; k < 4; ++k) sd[c][k] = (ok && (k < 2 || h == 0)) ? sd[c][k] : 0.f;
            }
            // every stash row of the phase is requested up front (96 registers in flight; the derivative / gradient registers are not live
            // yet and the weight fragment is requested behind the phase): ONE round trip to HBM instead of five dependent ones
            float zr[16], zd[4][16];
            {
                gcfp zp = pf_base(T + (size_t)(PDE_Z + 4 * 64 + 16 * w) * REGF);
#pragma unroll
                for (int r = 0; r < 16; ++r) zr[r] = STASH_LD(zp[r * REGF + lane]);
#pragma unroll
                for (int k = 0; k < PF_P0_FIRST; ++k) {
                    gcfp zq = pf_base(T + (size_t)(PDE_ZD + 320 * k + 4 * 64 + 16 * w) * REGF);
#pragma unroll
                    for (int r = 0; r < 16; ++r) zd[k][r] = STASH_LD(zq[r * REGF + lane]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // rows gw of each column's adjoint image (A operand of the output layer's weight gradient): wave c & 3 stores column c
#pragma unroll
            for (int c = 0; c < 5; ++c)
                if (w == (c & 3)) {
                    gfp gw_rows = pf_base(T + (size_t)(PDE_GA + 336 * c + 320) * REGF);
#pragma unroll
                    for (int s = 0; s < 16; ++s) gw_rows[s * REGF + lane] = s < 4 ? sd[c][s] : 0.f;
                }
            const float a4[4] = {w5.x, w5.y, w5.z, w5.w};
            {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = MFMA32(a4[k], sd[0][k], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float z = zr[r], s = fast_sigmoid(z), oms = 1.f - s;
                    d1[r] = s * (1.f + z * oms);
                    d2[r] = s * oms * (2.f + z * (1.f - 2.f * s));
                    gzv[r] = d1[r] * acc[r];
                }
            }
            // (the rows of the last tangent columns follow once z_4 has been consumed: all 96 at once do not fit the register budget)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = PF_P0_FIRST; k < 4; ++k) {
                gcfp zq = pf_base(T + (size_t)(PDE_ZD + 320 * k + 4 * 64 + 16 * w) * REGF);
#pragma unroll
                for (int r = 0; r < 16; ++r) zd[k][r] = STASH_LD(zq[r * REGF + lane]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 1; k <= 4; ++k) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                // one column at a time: an MFMA is a pure register operation that the optimiser forms wherever it likes - all four products
                // first, 64 registers - unless its operands come out of a statement it cannot move
                asm volatile("" : "+v"(sd[k][0]), "+v"(sd[k][1]), "+v"(sd[k][2]), "+v"(sd[k][3]) :: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = MFMA32(a4[q], sd[k][q], acc);
                float4* Xw = pf_rows(A, pf_inc(rs, k - 1));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float g4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = 4 * q + c;
                        const float ga = acc[r];
                        g4[c] = d1[r] * ga;
                        gzv[r] = gzv[r] + d2[r] * zd[k - 1][r] * ga;
                    }
                    Xw[q * 2 * FUSE_HR] = make_float4(g4[0], g4[1], g4[2], g4[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            pf_put(A, PF_V, gzv);
            __builtin_amdgcn_sched_barrier(0);              // (the fragment's 64 registers are free only now)
            split_load16(al.t4[4] + (size_t)w * 16 * 64, lane, wq);
            PT_ADD(TT, 0);
            FUSE_BAR(); PT_ADD(TT, 8);
        }
        pf_layer<3>(A, al, T, wq, rs, d1, d2, gzv, TT);
        pf_layer<2>(A, al, T, wq, rs, d1, d2, gzv, TT);
        pf_layer<1>(A, al, T, wq, rs, d1, d2, gzv, TT);
        pf_layer<0>(A, al, T, wq, rs, d1, d2, gzv, TT);
#ifdef PF_TIMING
        TT.pt[31] += 1;
#endif
    }
#ifdef PF_TIMING
    if (a.timing && blockIdx.x == 0 && w == 0 && lane == 0)
        for (int k = 0; k < 16; ++k) a.timing[k] = TT.pt[k];
    if (a.timing && blockIdx.x == 0 && w == 0 && lane == 0) a.timing[31] = TT.pt[31];
#endif
}

// ---- acceleration net (ReLU, value column): one tile = phase 0 (6 -> 128 from the seeds) + one phase per hidden layer
// gradient images rotate over three of the tangent images (c, c+1, c+2 mod 3; c advances by one per tile), layer inputs alternate as above
__device__ __forceinline__ int pf_inc3(int s, int by) { s += by; return s >= 3 ? s - 3 : s; }

template <int L>
__device__ __forceinline__ void pf_accel_phase(const PfA& A, const PdeFuseArgs& al, float* T, f32x4v (&wq)[16], int rimg, int wimg, int yimg) {
    float zr[16];
    {
        gcfp zp = pf_base(T + (size_t)(PDE_ZA + L * 64 + 16 * A.w) * REGF);
#pragma unroll
        for (int r = 0; r < 16; ++r) zr[r] = STASH_LD(zp[r * REGF + A.lane]);
    }
    f32x16 acc;
    pf_dgrad(A, rimg, wq, acc);
    if (L >= 1) {
        asm volatile("" :: "v"(acc[0]));
        split_load16(al.ta4[L] + (size_t)A.w * 16 * 64, A.lane, wq);
    }
    float4* Yw = pf_rows(A, yimg);
    float4* Xw = pf_rows(A, wimg);
    gfp gp = pf_base(T + (size_t)(PDE_GAA + 16 * A.w) * REGF);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float g4[4], a4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float z = zr[4 * k + c];
            g4[c] = (z > 0.f ? 1.f : 0.f) * acc[4 * k + c];         // act_d1<0>(z) * acc, as velnet_value_backward<0> forms it
            a4[c] = z > 0.f ? z : 0.f;
        }
        Yw[k * 2 * FUSE_HR] = make_float4(a4[0], a4[1], a4[2], a4[3]);
        if (L > 0) Xw[k * 2 * FUSE_HR] = make_float4(g4[0], g4[1], g4[2], g4[3]);
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) STASH_ST(gp[(4 * k + c) * REGF + A.lane], g4[c]);
        }
    }
}

// one acceleration-net tile; c = this tile's first gradient image
__device__ __forceinline__ void pf_accel_tile(const PfA& A, const PdeFuseArgs& a, int tile, int c, f32x4v (&wq)[16]) {
    const int w = A.w, lane = A.lane, h = lane >> 5, j = lane & 31;
    float* T = a.stash + (size_t)tile * PDE_TILE_ROWS * REGF;
    const int i = tile * TILE + j;
    const bool ok = i < (int)a.cap;
    PdeFuseArgs al = a;
    {
        gcfp sp_ = pf_base(a.seeds); al.seeds = (const float*)sp_;
#pragma unroll
        for (int l = 1; l <= 5; ++l) { gcfp tp_ = pf_base(reinterpret_cast<const float*>(a.ta4[l])); al.ta4[l] = reinterpret_cast<const float4*>((const float*)tp_); }
    }
    size_t csl = a.cap; asm volatile("" : "+s"(csl));
    // ---- phase 0
    {
        f32x4v w5;
        {
            gcf4p b5 = (gcf4p)(al.ta4[5] + (size_t)w * 64);
            asm("" : "+s"(b5));
            w5 = b5[lane];
        }
        // seeds of the six outputs (rows 30..35): lanes of half h hold outputs 4h .. 4h+3 in D-layout registers 0..3; the upper half owns only
        // two - for k = 2, 3 it re-reads the lower half's rows (36, 37 do not exist) and is zeroed
        float sd[4];
        {
            const int ii = ok ? i : 0, lo = ii + (h ? 4 * (int)csl : 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                gcfp sp = pf_base(al.seeds + (size_t)(30 + k) * csl);
                const float v = sp[k < 2 ? lo : ii];
                sd[k] = (ok && (k < 2 || h == 0)) ? v : 0.f;
            }
        }
        float zr[16];
        {
            gcfp zp = pf_base(T + (size_t)(PDE_ZA + 4 * 64 + 16 * w) * REGF);
#pragma unroll
            for (int r = 0; r < 16; ++r) zr[r] = STASH_LD(zp[r * REGF + lane]);
        }
        if (w == 0) {
            gfp gw_rows = pf_base(T + (size_t)(PDE_GAA + 320) * REGF);
#pragma unroll
            for (int s = 0; s < 16; ++s) gw_rows[s * REGF + lane] = s < 4 ? sd[s] : 0.f;
        }
        const float a4[4] = {w5.x, w5.y, w5.z, w5.w};
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = MFMA32(a4[k], sd[k], acc);
        float4* Xw = pf_rows(A, c);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float g4[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) g4[cc] = (zr[4 * q + cc] > 0.f ? 1.f : 0.f) * acc[4 * q + cc];
            Xw[q * 2 * FUSE_HR] = make_float4(g4[0], g4[1], g4[2], g4[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        split_load16(al.ta4[4] + (size_t)w * 16 * 64, lane, wq);
        FUSE_BAR();
    }
    pf_accel_phase<3>(A, al, T, wq, c, pf_inc3(c, 1), PF_Y0 + 1); FUSE_BAR();
    pf_accel_phase<2>(A, al, T, wq, pf_inc3(c, 1), pf_inc3(c, 2), PF_Y0 + 0); FUSE_BAR();
    pf_accel_phase<1>(A, al, T, wq, pf_inc3(c, 2), c, PF_Y0 + 1); FUSE_BAR();
    pf_accel_phase<0>(A, al, T, wq, c, c, PF_Y0 + 0); FUSE_BAR();
}

// the queue of acceleration-net tiles: adjoint wave 0 takes the NEXT tile's index during phase 0 of the current one and parks it in LDS
// (word n & 1 for the n-th tile of this workgroup); every wave reads it behind the tile's last barrier
#define PF_QWORD(lds_f, n) ((volatile int*)((lds_f) + PF_IMAGES * FUSE_XB * 4))[(n) & 1]

__device__ __forceinline__ void pf_role_adjoint_accel(const PdeFuseArgs& a, float* lds, int w, int lane, int ntiles) {
    PfA A; A.S = reinterpret_cast<float4*>(lds); A.w = w; A.lane = lane; A.x4 = false;
    A.pos = (lane >> 5) * FUSE_HR + (lane & 31);
    f32x4v wq[16];
    // (no queue - NVFI_DETERMINISTIC=1 -: the static share of the first half, so that every slab sums the same tiles in the same order in every run)
    const int G = gridDim.x;
    if (w == 0 && lane == 0) PF_QWORD(lds, 0) = a.queue ? atomicAdd(a.queue, 1) : (int)blockIdx.x;
    FUSE_BAR();                                           // the transition barrier: the first index is visible, the contraction waves have flushed weight_net's slabs
    int c = 0;
#pragma unroll 1
    for (int n = 0;; ++n) {
        const int tile = __builtin_amdgcn_readfirstlane(PF_QWORD(lds, n));
        if (tile >= ntiles) break;
        if (w == 0 && lane == 0) PF_QWORD(lds, n + 1) = a.queue ? atomicAdd(a.queue, 1) : (int)blockIdx.x + (n + 1) * G;
        pf_accel_tile(A, a, tile, c, wq);
        c = pf_inc3(c, 1);
    }
}

// ---------------------------------------------------------------- contraction waves
// (the barrier names the accumulators as in/out operands: see FUSE_BAR_G in vel_fuse.hip)
#define PF_BAR_G() do { __builtin_amdgcn_sched_barrier(0); PT_ADD(TT, 16);                                              \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(G0a), "+v"(G0b), "+v"(G1a), "+v"(G1b), "+v"(G2a), "+v"(G2b), "+v"(G3a), "+v"(G3b) :: "memory"); \
        PT_ADD(TT, 17); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PF_XBF (FUSE_XB * 4)               // floats per image
// G[L][t] += sum over the tile's 32 points of g[32 ob + row][s] * a[32 (ib0 + t) + col][s]; BIAS: the value column's gradient also sums into the bias
#define PF_CONTRACT(L, XIMG, YIMG, BIAS)                                                                             \
    do {                                                                                                             \
        const float* xa_ = Sf + (XIMG) * PF_XBF + ob * FUSE_TF + o; const float* yb_ = Sf + (YIMG) * PF_XBF + ib0 * FUSE_TF + o; \
        _Pragma("unroll") for (int st = 0; st < 16; ++st) {                                                          \
            const float av_ = xa_[8 * st], b0_ = yb_[8 * st], b1_ = yb_[8 * st + FUSE_TF];                            \
            G##L##a = MFMA32(av_, b0_, G##L##a); G##L##b = MFMA32(av_, b1_, G##L##b);                               \
            if (BIAS) asm("v_add_f32 %0, %0, %1" : "+v"(bs##L) : "v"(av_));                                          \
        }                                                                                                            \
    } while (0)
// the contraction waves' intervals of layer L, one phase behind the adjoint waves: value pair, tangent pairs 1..3; the pair of the fourth
// tangent column is contracted in the first interval of the next layer (or of the next tile)
#define PF_C_LAYER(L, TAU0)                                                                                          \
    do {                                                                                                             \
        PF_CONTRACT(L, PF_V, PF_Y0 + ((TAU0) & 1), 1); PF_BAR_G();                                                    \
        PF_CONTRACT(L, rs, PF_Y0 + (((TAU0) + 1) & 1), 0); rs = pf_inc(rs, 1); PF_BAR_G();                            \
        PF_CONTRACT(L, rs, PF_Y0 + (((TAU0) + 2) & 1), 0); rs = pf_inc(rs, 1); PF_BAR_G();                            \
        PF_CONTRACT(L, rs, PF_Y0 + (((TAU0) + 3) & 1), 0); rs = pf_inc(rs, 1); PF_BAR_G();                            \
    } while (0)

// one slab per layer and workgroup, in k_wgrad_ring8's format (rows / columns in p-space, bias sums behind the 128 x 128 block)
// (wave-uniform row-group bases + ONE lane offset: with per-lane 64-bit addresses the flush in the middle of the kernel spills hundreds of registers)
#define PF_FLUSH(SLABS, L)                                                                                           \
    do {                                                                                                             \
        float* Sl = (SLABS) + (size_t)(L) * a.layer_stride + (size_t)blockIdx.x * a.slab_floats;                     \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
            gfp Sq = pf_base(Sl + (size_t)(32 * ob + 8 * q) * 128 + 32 * ib0);                                       \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
                Sq[c * 128 + flo] = G##L##a[4 * q + c];                                                              \
                Sq[c * 128 + 32 + flo] = G##L##b[4 * q + c];                                                         \
            }                                                                                                        \
        }                                                                                                            \
        if ((v & 1) == 0) {                                                                                          \
            float bsum = bs##L; bsum += __shfl_xor(bsum, 32);                                                        \
            if (kk == 0) Sl[(size_t)128 * 128 + 32 * ob + i] = bsum;                                                 \
        }                                                                                                            \
    } while (0)

// ACCEL = false: weight_net's tiles (static share: tile = workgroup + k * grid); true: the acceleration net's (queue).  Two instances run
// one after the other, each with its own accumulators from zero to its flush (one body that flushed, cleared and went on spilled all 128)
template <bool ACCEL>
__device__ __forceinline__ void pf_role_contract(const PdeFuseArgs& a, const float* Sf, int v, int lane, int ntiles) {
    const int i = lane & 31, kk = lane >> 5;
    const int ob = v >> 1, ib0 = 2 * (v & 1);
    // float offset of (row i of a 32-row tile, sample kk) in an image
    const int o = ((i >> 3) * 2 + (i & 1)) * (FUSE_HR * 4) + ((i >> 1) & 3) + 4 * kk;      // MFMA step st contracts sample 2 st + kk: + 8 st floats
    const int flo = 4 * kk * 128 + i;                     // lane offset of a slab row: row (r & 3) + 8 (r >> 2) + 4 kk of the wave's 32, column i
    f32x16 G0a, G0b, G1a, G1b, G2a, G2b, G3a, G3b;
    float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { G0a[r] = 0.f; G0b[r] = 0.f; G1a[r] = 0.f; G1b[r] = 0.f; G2a[r] = 0.f; G2b[r] = 0.f; G3a[r] = 0.f; G3b[r] = 0.f; }
    PfT TT;
#ifdef PF_TIMING
    for (int k = 0; k < 32; ++k) TT.pt[k] = 0;
    TT.t0 = PT_NOW();
#endif
    bool pending = false;
    if (!ACCEL) {
        const int G = gridDim.x;
        int rs = 0;                                       // tangent image of the next tangent pair
        // pending: the fourth tangent pair of the previous tile's layer 0 (image rs, Y image 0)
#pragma unroll 1
        for (int tile = blockIdx.x; tile < ntiles; tile += G) {
            if (pending) { PF_CONTRACT(0, rs, PF_Y0 + 0, 0); rs = pf_inc(rs, 1); }
            PF_BAR_G();                                       // phase 0 of the adjoint waves
            PF_BAR_G();                                       // (3, 0)
            PF_C_LAYER(3, 1);
            PF_CONTRACT(3, rs, PF_Y0 + ((1 + 4) & 1), 0); rs = pf_inc(rs, 1); PF_BAR_G();       // (2, 0)
            PF_C_LAYER(2, 6);
            PF_CONTRACT(2, rs, PF_Y0 + ((6 + 4) & 1), 0); rs = pf_inc(rs, 1); PF_BAR_G();       // (1, 0)
            PF_C_LAYER(1, 11);
            PF_CONTRACT(1, rs, PF_Y0 + ((11 + 4) & 1), 0); rs = pf_inc(rs, 1); PF_BAR_G();      // (0, 0)
            PF_C_LAYER(0, 16);
            pending = true;
        }
        if (pending) PF_CONTRACT(0, rs, PF_Y0 + 0, 0);
#ifdef PF_TIMING
        if (a.timing && blockIdx.x == 0 && v == 0 && lane == 0) { a.timing[16] = TT.pt[16]; a.timing[17] = TT.pt[17]; }
#endif
        PF_FLUSH(a.slabs, 0); PF_FLUSH(a.slabs, 1); PF_FLUSH(a.slabs, 2); PF_FLUSH(a.slabs, 3);
    } else {
        PF_BAR_G();                                       // the transition barrier (weight_net's slabs are on their way, the first queue index is visible)
        int c = 0;
#pragma unroll 1
        for (int n = 0;; ++n) {
            const int tile = __builtin_amdgcn_readfirstlane(PF_QWORD(Sf, n));
            if (tile >= ntiles) break;
            // pairs (gradient image of layer l + 1, layer input of layer l), all with the bias sums; c = this tile's first gradient image
            if (pending) PF_CONTRACT(0, pf_inc3(c, 2), PF_Y0 + 0, 1);      // the previous tile's last pair: its image c_prev = c - 1
            PF_BAR_G();                                   // phase 0
            PF_BAR_G();                                   // layer 3
            PF_CONTRACT(3, c, PF_Y0 + 1, 1); PF_BAR_G();                    // layer 2
            PF_CONTRACT(2, pf_inc3(c, 1), PF_Y0 + 0, 1); PF_BAR_G();        // layer 1
            PF_CONTRACT(1, pf_inc3(c, 2), PF_Y0 + 1, 1); PF_BAR_G();        // layer 0
            pending = true;
            c = pf_inc3(c, 1);
        }
        if (pending) PF_CONTRACT(0, pf_inc3(c, 2), PF_Y0 + 0, 1);
        PF_FLUSH(a.slabs_a, 0); PF_FLUSH(a.slabs_a, 1); PF_FLUSH(a.slabs_a, 2); PF_FLUSH(a.slabs_a, 3);
    }
}

__global__ __launch_bounds__(PF_THREADS) void k_pde_fuse_bwd(PdeFuseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = __builtin_amdgcn_readfirstlane(pf_count(a));
    // whole 128-point groups, as the forward stashed them
    const int ntiles = (count + WG_SAMPLES - 1) / WG_SAMPLES * (WG_SAMPLES / TILE);
    if (wave < 4) {
        __builtin_amdgcn_s_setprio(3);
        pf_role_adjoint(a, reinterpret_cast<float4*>(lds), wave, lane, ntiles);
        if (a.do_accel) pf_role_adjoint_accel(a, lds, wave, lane, ntiles);
    } else {
        pf_role_contract<false>(a, lds, wave - 4, lane, ntiles);
        if (a.do_accel) pf_role_contract<true>(a, lds, wave - 4, lane, ntiles);
    }
}

int launch_pde_fuse_bwd(const PdeFuseArgs& a, int64_t cap_points, int max_slabs, int* nslab_out, hipStream_t st) {
    *nslab_out = 0;
    const int64_t tiles = (cap_points + TILE - 1) / TILE;
    if (tiles <= 0) return 0;
    static int ncu_dev[64] = {0};                          // per device: the attribute and the CU count belong to the current device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!ncu_dev[dev]) {
        hipDeviceProp_t prop;
        int n = 256;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
        HIPCK(hipFuncSetAttribute((const void*)k_pde_fuse_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS_BYTES));
        ncu_dev[dev] = n;
    }
    const int ncu = ncu_dev[dev];
    int G = ncu < max_slabs ? ncu : max_slabs;           // one persistent workgroup per CU (leaving 8-32 CUs to the other streams: no gain)
    if ((int64_t)G > tiles) G = (int)tiles;
#ifdef PF_TIMING
    static unsigned long long* tbuf = nullptr; static int shots = 0;
    if (!tbuf) { HIPCK(hipMalloc(&tbuf, 32 * 8)); }
    PdeFuseArgs b = a; b.timing = tbuf;
    HIPCK(hipMemsetAsync(tbuf, 0, 32 * 8, st));
    hipLaunchKernelGGL(k_pde_fuse_bwd, dim3((unsigned)G), dim3(PF_THREADS), PF_LDS_BYTES, st, b);
    if (++shots % 8 == 0 && shots <= 64) {
        unsigned long long hh[32];
        HIPCK(hipStreamSynchronize(st));
        HIPCK(hipMemcpy(hh, tbuf, sizeof(hh), hipMemcpyDeviceToHost));
        const double n = hh[31] ? (double)hh[31] : 1.0;
        fprintf(stderr, "[pde fuse timing] tiles %llu | per tile: P0 %.0f (+wait %.0f) | value dgrad %.0f epi %.0f wait %.0f | tangent dgrad %.0f epi %.0f wait %.0f / %.0f | contraction work %.0f wait %.0f\n",
                hh[31], hh[0] / n, hh[8] / n, hh[1] / n, hh[2] / n, hh[9] / n, hh[3] / n, hh[4] / n, hh[11] / n, hh[12] / n, hh[16] / n, hh[17] / n);
    }
#else
    hipLaunchKernelGGL(k_pde_fuse_bwd, dim3((unsigned)G), dim3(PF_THREADS), PF_LDS_BYTES, st, a);
#endif
    LAUNCHCK();
    *nslab_out = G;
    return 0;
}
