// wgrad_ring.hip - split-K weight-gradient kernel with an LDS ring filled by global_load_lds_dwordx4 (gfx950 LDS-DMA).
//
//   G[pA][pB] = sum over sample tiles of A[pA][j] * B'[pB][j]        (p-space rows of the stash images, B' = bmode(B, B2))
//
// Same contraction, slab format and reduce kernel as the k_wgrad of rounds 1-2 (retired in round 6); what changes is how the operands
// reach the MFMAs and how the work is shared.  k_wgrad loads both stash images of an item straight into the registers of ONE wave per
// SIMD (two alternating operand sets, the A image read by both column-part workers) and gives every job its own workers.  Here:
//   * ONE workgroup of EIGHT waves per CU (two per SIMD: one wave's activation / address / DMA-issue instructions run under the
//     other's MFMAs) walks EVERY job of the launch with the same share of each job's items - all workgroups do identical work whatever
//     the mix of 128x128, 32x128 and 128x32 jobs - and writes slab g of every job;
//   * every byte of a stash image is fetched ONCE per item, by LDS-DMA: a wave-instruction moves 1 KiB (8 rows of 128 B) without
//     touching a VGPR, into a ring of RING_MAX_SLOTS item slots (2 since the end of round 3; 4 before - 96 KB per CU in flight - was no faster): the next item
//     is in flight behind the one being contracted, across the single barrier per item (raw s_barrier + counted s_waitcnt vmcnt: a
//     __syncthreads() would drain the DMA queue).  The DMA is inline asm with an SGPR base + a per-lane 32-bit offset that is constant
//     for the whole job, so an item costs ~30 scalar instructions per wave;
//   * the LDS image is row-major [row p][32 samples] with the eight 16-byte chunks of a row XOR-permuted by (p >> 1) & 7 - applied on
//     the SOURCE address, the DMA destination is lane-linear - so that the MFMA operand reads (lane (i, h) reads samples 16h+4v..+3 of
//     row i: ds_read_b128) are conflict-free in every 16-lane service group of the instruction (SQ_LDS_BANK_CONFLICT = 0);
//   * wave (q, hk): q = w & 3 picks the column tile of B (or the row tile of A when B has a single tile), hk = w >> 2 the half of the
//     tile's 32 samples it contracts: every B value is activated exactly once; the two half-sums meet in LDS once per job;
//   * jobs that accumulate into the same gradient (the value column and the four tangent columns of a PDE layer) are CHAINED: the
//     accumulators carry over and one slab is written for both.
// Reference semantics: the weight gradients of VelBasis (models/velocity_field.py:60-67), MLPRender_PE + basis_mat
// (models/tensorf_base.py:88-98, models/tensorf_keyframe.py:310) and MaskField (models/mask_field.py:68-83) under autograd.
#include "engine.h"
#include "common.h"
#include <stdlib.h>
#include <string.h>

#define RING_ROW_BYTES 128                 // one stash row of a tile: 32 samples
#define RING_TILE_BYTES 4096               // 32 rows
#ifndef RING_MAX_SLOTS
#define RING_MAX_SLOTS 2                   // ring slots of one item: the item being contracted + one in flight (64 KB; 96 KB for the tangent jobs).  Measured
                                           // in the step, -D variants: 4 slots (128 KB) 1.128-1.137 ms per step for the four launches, 3 slots 1.127, 2 slots 1.104-1.110
                                           // (5 slots like 4): the window is not what limits the kernel, and 64 KB of LDS stay free for the kernels of the other streams
#endif

// LDS-DMA of 1 KiB per wave-instruction: lane L's 16 bytes at base + voff land at lds_dst + 16 L (base: SGPR pair, lds_dst: wave-uniform
// LDS byte address).  Inline asm on purpose: hipcc waits vmcnt(0) in front of the next ds_read whenever one of ITS global_load_lds is
// outstanding (the builtin form of this loop drained the ring every item - checked in the .s); an asm DMA is outside its bookkeeping,
// the counted waits below are ours.  M0 is written and restored inside the statement (the compiler does not preserve it around asm).
__device__ __forceinline__ void glds16(const char* base, int voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
__device__ __forceinline__ void vmcnt_le(int n) {
    // counted wait with a compile-time ladder (the immediate must be a constant)
    switch (n) {
        case 0: VMCNT(0); break;
        case 4: VMCNT(4); break;
        case 5: VMCNT(5); break;
        case 6: VMCNT(6); break;
        case 8: VMCNT(8); break;
        default: VMCNT(0); break;
    }
}

struct RingAcc { f32x16 acc[4]; float bsum; };

// MTA row tiles of A, KTB column tiles of B (32 rows each); BM as in k_wgrad.  `fresh`: start from zero accumulators; `flush`: write the slab.
template <int MTA, int KTB, int BM>
__device__ __forceinline__ void wgrad_ring8(const WgradJob& J, int g, int G, char* lds, RingAcc& R, bool fresh, bool flush) {
    constexpr bool tan = BM == BM_SILU_TAN || BM == BM_RELU_TAN;
    constexpr int NA = 4 * MTA, NB = 4 * KTB, NP = NA + NB * (tan ? 2 : 1);     // 1 KiB pieces per item: A | B | B2, consecutive in a slot
    constexpr int A_BYTES = MTA * RING_TILE_BYTES, B_BYTES = KTB * RING_TILE_BYTES;
    constexpr int SLOT_BYTES = NP * 1024;
    constexpr int NSLOT = SLOT_BYTES * RING_MAX_SLOTS <= 128 * 1024 + 16 * 1024 * (RING_MAX_SLOTS == 3) ? RING_MAX_SLOTS : 3;
    constexpr int LPI = (NP + 7) / 8;                       // DMA instructions per wave and item (short waves re-issue a piece: same bytes, same place)
    constexpr int NACC = KTB == 4 ? MTA : 1;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int q = w & 3, hk = w >> 2;
    const int kt = KTB == 4 ? q : (KTB == 2 ? (q & 1) : 0);
    const int mt0 = KTB == 1 ? q : 0;
    const bool active = KTB == 2 ? (q < 2) : true;
    const int a_rows = 32 * MTA, b_rows = 32 * KTB;

    int count = *J.count;
    int ntiles = (count + TILE - 1) / TILE;
    if (ntiles > J.cap_tiles) ntiles = J.cap_tiles;
    const int nitems = J.nrep * ntiles;
    const int n_my = g < nitems ? (nitems - g + G - 1) / G : 0;

    if (fresh) {
        R.bsum = 0.f;
#pragma unroll
        for (int m = 0; m < NACC; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) R.acc[m][r] = 0.f;
    }

    // DMA source offset of this lane inside a 1 KiB piece (8 rows): lane L -> row L >> 3, chunk position L & 7 holds chunk
    // (L & 7) ^ ((row >> 1) & 7); a wave's pieces ((w + 8 j) mod NP, NP even) all have the parity of w, so (row >> 1) & 7 = (4 w + (L >> 4)) & 7
    const int swz_ld = (4 * (w & 1) + (lane >> 4)) & 7;
    const int src_off = (lane >> 3) * RING_ROW_BYTES + (((lane & 7) ^ swz_ld) << 4);
    // operand reads: row (32 t + i), chunk c at position c ^ ((i >> 1) & 7)
    const int swz_rd = (i >> 1) & 7;
    const unsigned lds_base = lds_addr(lds);
    // per wave and job: which image each of its pieces belongs to, its offset inside the image (+ the lane offset), its place in the slot
    int voff[LPI], img[LPI], soff[LPI];
#pragma unroll
    for (int j = 0; j < LPI; ++j) {
        int piece = w + 8 * j;
        if ((LPI * 8 != NP) && piece >= NP) piece -= NP;
        img[j] = piece < NA ? 0 : (piece < NA + NB ? 1 : 2);
        const int in_img = piece - (img[j] == 0 ? 0 : (img[j] == 1 ? NA : NA + NB));
        voff[j] = in_img * 1024 + src_off;
        soff[j] = piece * 1024;
    }
    // the item walk: item = g + k G -> (rep, tile); the image bases advance by G tiles per item and wrap into the next rep
    const size_t a_step = (size_t)G * J.a_tile_stride * 4, b_step = (size_t)G * J.b_tile_stride * 4;
    const long a_wrap = ((long)J.a_rep_stride - (long)ntiles * (long)J.a_tile_stride) * 4;
    const long b_wrap = ((long)J.b_rep_stride - (long)ntiles * (long)J.b_tile_stride) * 4;
    const long b2_wrap = tan ? ((long)J.b2_rep_stride - (long)ntiles * (long)J.b_tile_stride) * 4 : 0;
    int nxt_tile = g, nxt_left = n_my;                         // the next item to issue (tile inside its rep), items left to issue
    const char* pa = reinterpret_cast<const char*>(J.A);
    const char* pb = reinterpret_cast<const char*>(J.B);
    const char* pb2 = reinterpret_cast<const char*>(tan ? J.B2 : J.B);
    if (n_my > 0) {
        while (nxt_tile >= ntiles) { nxt_tile -= ntiles; pa += (size_t)J.a_rep_stride * 4; pb += (size_t)J.b_rep_stride * 4; if (tan) pb2 += (size_t)J.b2_rep_stride * 4; }
        pa += (size_t)nxt_tile * J.a_tile_stride * 4; pb += (size_t)nxt_tile * J.b_tile_stride * 4; if (tan) pb2 += (size_t)nxt_tile * J.b_tile_stride * 4;
    }
    auto issue = [&](int slot) {
        // (issued unconditionally so that the counted waits stay static: past the last item the last one is simply read again)
        const unsigned S = __builtin_amdgcn_readfirstlane(lds_base + slot * SLOT_BYTES);
#pragma unroll
        for (int j = 0; j < LPI; ++j) {
            const char* base = img[j] == 0 ? pa : (img[j] == 1 ? pb : pb2);
            glds16(base, voff[j], S + soff[j]);
        }
        if (nxt_left > 1) {
            --nxt_left;
            nxt_tile += G; pa += a_step; pb += b_step; if (tan) pb2 += b_step;
            while (nxt_tile >= ntiles) { nxt_tile -= ntiles; pa += a_wrap; pb += b_wrap; if (tan) pb2 += b2_wrap; }
        }
    };

    // row tile of accumulator m: rotated by q when a wave owns all row tiles, so that the tile whose bias sums this wave owns is always
    // a4[0] (a select over the wave index made the compiler spill the operand array to scratch)
    int rt[NACC];
#pragma unroll
    for (int m = 0; m < NACC; ++m) rt[m] = KTB == 4 ? ((MTA == 4 ? q + m : m) & (MTA - 1)) : mt0;
    // bias sums: the tangent columns carry no bias term, and in a chain the reduce kernel takes the bias from the FIRST job's slabs only
    const bool bias_owner = fresh && !tan && (KTB == 4 ? (q < MTA) : (KTB == 1 ? true : q == 0));
    auto compute = [&](int slot) {
        const char* S = lds + slot * SLOT_BYTES;
        const char* Ab = S + i * RING_ROW_BYTES;
        const char* Bb = S + A_BYTES + (32 * kt + i) * RING_ROW_BYTES;
#pragma unroll
        for (int vv = 0; vv < 2; ++vv) {
            const int pos = ((4 * h + 2 * hk + vv) ^ swz_rd) << 4;
            const float4 t4 = *reinterpret_cast<const float4*>(Bb + pos);
            float b[4] = {t4.x, t4.y, t4.z, t4.w};
            if (BM == BM_SILU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) b[c] = act_f<1>(b[c]);
            } else if (BM == BM_RELU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) b[c] = act_f<0>(b[c]);
            } else if (tan) {
                const float4 u4 = *reinterpret_cast<const float4*>(Bb + B_BYTES + pos);
                const float z2[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) b[c] = (BM == BM_SILU_TAN ? act_d1<1>(b[c]) : act_d1<0>(b[c])) * z2[c];
            }
            float4 a4[NACC];
#pragma unroll
            for (int m = 0; m < NACC; ++m) a4[m] = *reinterpret_cast<const float4*>(Ab + rt[m] * RING_TILE_BYTES + pos);
            if (bias_owner) R.bsum += (a4[0].x + a4[0].y) + (a4[0].z + a4[0].w);
            if (active) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int m = 0; m < NACC; ++m) {
                        const float av = c == 0 ? a4[m].x : c == 1 ? a4[m].y : c == 2 ? a4[m].z : a4[m].w;
                        R.acc[m] = MFMA32(av, b[c], R.acc[m]);
                    }
            }
        }
    };

    if (n_my > 0) {
#pragma unroll
        for (int s = 0; s < NSLOT - 1; ++s) issue(s);
        int slot = 0;
#pragma unroll 1
        for (int k = 0; k < n_my; ++k) {
            vmcnt_le(LPI * (NSLOT - 2));                     // this wave's pieces of item k have landed
            __builtin_amdgcn_s_barrier();                    // ... and everybody else's; all waves are done reading the slot refilled next
            int nslot = slot + NSLOT - 1; nslot = nslot >= NSLOT ? nslot - NSLOT : nslot;
#ifndef RING_EXP_NODMA          // timing experiments only (wrong gradients): -DRING_EXP_NODMA / -DRING_EXP_NOCOMPUTE
            issue(nslot);
#endif
#ifndef RING_EXP_NOCOMPUTE
            compute(slot);
#endif
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
        VMCNT(0);                                            // the tail prefetches must land before the LDS is reused
    }
    __builtin_amdgcn_s_barrier();
    if (!flush) return;
    // the two sample halves of a tile meet in LDS (once per slab): waves 4..7 park their sums, waves 0..3 add and write the slab
    float* X = reinterpret_cast<float*>(lds) + (size_t)q * (NACC * 16 + 1) * 64 + lane;
    if (hk == 1) {
#pragma unroll
        for (int m = 0; m < NACC; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) X[(m * 16 + r) * 64] = R.acc[m][r];
        X[NACC * 16 * 64] = R.bsum;
    }
    __syncthreads();
    if (hk == 0) {
        float* S = J.slabs + (size_t)g * ((size_t)a_rows * b_rows + a_rows);
        if (active) {
#pragma unroll
            for (int m = 0; m < NACC; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * rt[m] + (r & 3) + 8 * (r >> 2) + 4 * h;
                    S[(size_t)row * b_rows + 32 * kt + i] = R.acc[m][r] + X[(m * 16 + r) * 64];
                }
        }
        const bool bias_slot = KTB == 4 ? (q < MTA) : (KTB == 1 ? true : q == 0);
        if (bias_slot) {
            float bs = R.bsum + X[NACC * 16 * 64];
            bs += __shfl_xor(bs, 32);
            if (h == 0) S[(size_t)a_rows * b_rows + 32 * rt[0] + i] = bs;
        }
    }
    __syncthreads();      // the next job's prologue refills this LDS
}

extern __shared__ __attribute__((aligned(1024))) char ring_lds[];

// chain[j] bit 0: job j continues the accumulators of job j - 1; bit 1: job j hands its accumulators to job j + 1 (no slab written)
struct RingChain { unsigned char c[MAX_WGRAD_JOBS]; };

__global__ __launch_bounds__(512, 1) void k_wgrad_ring8(WgradJobs jobs, RingChain chain) {
    const int g = blockIdx.x, G = gridDim.x;
    RingAcc R;
#pragma unroll 1
    for (int jn = 0; jn < jobs.n; ++jn) {
        const WgradJob& J = jobs.j[jn];
        const bool fresh = !(chain.c[jn] & 1), flush = !(chain.c[jn] & 2);
#define RING8_MODES(MT_, KT_)                                                                         \
        switch (J.bmode) {                                                                            \
            case BM_RAW: wgrad_ring8<MT_, KT_, BM_RAW>(J, g, G, ring_lds, R, fresh, flush); break;    \
            case BM_SILU: wgrad_ring8<MT_, KT_, BM_SILU>(J, g, G, ring_lds, R, fresh, flush); break;  \
            case BM_RELU: wgrad_ring8<MT_, KT_, BM_RELU>(J, g, G, ring_lds, R, fresh, flush); break;  \
            case BM_SILU_TAN: wgrad_ring8<MT_, KT_, BM_SILU_TAN>(J, g, G, ring_lds, R, fresh, flush); break; \
            default: wgrad_ring8<MT_, KT_, BM_RELU_TAN>(J, g, G, ring_lds, R, fresh, flush); break;   \
        }
        if (J.a_regs == 64 && J.b_regs == 64) { RING8_MODES(4, 4) }
        else if (J.a_regs == 16 && J.b_regs == 64) { RING8_MODES(1, 4) }
        else if (J.a_regs == 64 && J.b_regs == 16) wgrad_ring8<4, 1, BM_RAW>(J, g, G, ring_lds, R, fresh, flush);
        else if (J.a_regs == 16 && J.b_regs == 32) wgrad_ring8<1, 2, BM_RAW>(J, g, G, ring_lds, R, fresh, flush);
    }
}

static int ring_slot_bytes(const WgradJob& J) {
    const bool tanm = J.bmode == BM_SILU_TAN || J.bmode == BM_RELU_TAN;
    return (J.a_regs / 16 + (J.b_regs / 16) * (tanm ? 2 : 1)) * RING_TILE_BYTES;
}

// Called by launch_wgrad (engine.hip): every job's nslab is the CAPACITY of its slab buffer on entry and the number of slabs written
// on return (bj / br are updated; the caller launches k_wgrad_reduce over br).
int launch_wgrad_ring(WgradJobs& bj, ReduceJobs& br, hipStream_t st) {
    if (bj.n == 0) return 0;
    static int ncu = 0, want = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t prop;
        ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        want = ncu;                                             // workgroups of a launch: one per CU (192 of 256 measured +1.5 % in round 3, within noise since)
        HIPCK(hipFuncSetAttribute((const void*)k_wgrad_ring8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    int G = want, lds_need = 68 * 1024;                         // (the end-of-job exchange of the two sample halves needs 66.5 KB)
    for (int i = 0; i < bj.n; ++i) {
        const WgradJob& J = bj.j[i];
        const bool ok = (J.a_regs == 64 && J.b_regs == 64) || (J.a_regs == 16 && J.b_regs == 64) || (J.a_regs == 64 && J.b_regs == 16 && J.bmode == BM_RAW) ||
                        (J.a_regs == 16 && J.b_regs == 32 && J.bmode == BM_RAW);
        if (!ok) return nvfi_fail(5, "k_wgrad_ring8: unsupported tile shape a_regs=%d b_regs=%d bmode=%d", J.a_regs, J.b_regs, J.bmode);
        G = J.nslab < G ? J.nslab : G;
        const int sb = ring_slot_bytes(J);
        const int need = sb * (sb * RING_MAX_SLOTS <= 128 * 1024 + 16 * 1024 * (RING_MAX_SLOTS == 3) ? RING_MAX_SLOTS : 3);
        lds_need = need > lds_need ? need : lds_need;
    }
    RingChain ch; memset(&ch, 0, sizeof(ch));
    for (int i = 0; i < bj.n; ++i) {
        WgradJob& J = bj.j[i];
        for (int k = 0; k < br.n; ++k) {
            if (br.j[k].slabs == J.slabs) br.j[k].nslab = G;
            if (br.j[k].slabs2 == J.slabs) br.j[k].nslab2 = G;
        }
        J.nslab = G;
    }
    // chain consecutive jobs of one shape whose slabs a reduce job adds into the same gradient (slabs + slabs2): one slab for both
    for (int i = 0; i + 1 < bj.n; ++i) {
        const WgradJob &J0 = bj.j[i], &J1 = bj.j[i + 1];
        if (J0.a_regs != J1.a_regs || J0.b_regs != J1.b_regs || (ch.c[i] & 1)) continue;
        bool pair = false;
        for (int k = 0; k < br.n; ++k) pair = pair || (br.j[k].slabs == J0.slabs && br.j[k].slabs2 == J1.slabs);
        if (!pair) continue;
        ch.c[i] |= 2; ch.c[i + 1] |= 1;
        const float* first = J0.slabs;
        for (int k = 0; k < br.n; ++k) {       // every reader of the first slab set (weights: slabs + slabs2; bias: slabs alone) reads the second
            ReduceJob& Q = br.j[k];
            if (Q.slabs == first) { Q.slabs = J1.slabs; Q.nslab = G; if (Q.slabs2 == J1.slabs) { Q.slabs2 = nullptr; Q.nslab2 = 0; } }
        }
    }
    hipLaunchKernelGGL(k_wgrad_ring8, dim3(G), dim3(512), (size_t)lds_need, st, bj, ch);
    LAUNCHCK();
    return 0;
}
