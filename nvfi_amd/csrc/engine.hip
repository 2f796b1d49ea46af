// engine.hip - fragment packing, split-K weight-gradient kernel, slab reduce.  See engine.h.
#include "engine.h"
#include "common.h"
#include "x6.h"
#include <stdlib.h>
#include <string.h>

__device__ __forceinline__ void pack_job(const PackJob& J, int y, int ny) {
    const int NS4 = (J.NS + 3) >> 2;
    const int total = J.x4 ? J.MT * NS4 * 256 : J.MT * J.NS * 64;
    for (int idx = y * blockDim.x + threadIdx.x; idx < total; idx += ny * blockDim.x) {
        int lane, s, m;
        if (J.x4) { const int k = idx & 3, ms = idx >> 8; lane = (idx >> 2) & 63; m = ms / NS4; s = 4 * (ms - m * NS4) + k; }
        else { const int ms = idx >> 6; lane = idx & 63; m = ms / J.NS; s = ms - m * J.NS; }
        int i = lane & 31, h = lane >> 5;
        int rowL = row_logical(J.row_kind, 32 * m + i);
        int colL = slot_logical(J.slot_kind, 2 * s + h);
        float v = 0.f;
        if (s < J.NS) {
            if (!J.transposed) {
                if (rowL >= 0 && rowL < J.out && colL >= 0 && colL < J.in) v = J.W[(size_t)rowL * J.in + colL];
            } else {
                if (rowL >= 0 && rowL < J.in && colL >= 0 && colL < J.out) v = J.W[(size_t)colL * J.in + rowL];
            }
        }
        J.frag[idx] = v;
    }
    if (J.bfrag && y == 0) {
        for (int rho = threadIdx.x; rho < J.MT * 32; rho += blockDim.x) {
            int rl = row_logical(J.row_kind, rho);
            J.bfrag[rho] = (J.b && rl >= 0 && rl < J.out) ? J.b[rl] : 0.f;
        }
    }
}
__global__ void k_pack(PackJobs jobs) { pack_job(jobs.j[blockIdx.x], blockIdx.y, gridDim.y); }
// ... and, behind the fp32 fragment jobs, the three bfloat16 images of the x6 kernels (x6.h): workgroups [jobs.n, jobs.n + X6_PACK_WGX) x 8
#define X6_PACK_WGX ((X6_H8 + 8 * 256 - 1) / (8 * 256))
__global__ void k_pack_all(PackJobsAll jobs, X6PackArgs x6) {
    if ((int)blockIdx.x < jobs.n) pack_job(jobs.j[blockIdx.x], blockIdx.y, gridDim.y);
    else x6_pack_body(x6, (((int)blockIdx.x - jobs.n) * 8 + (int)blockIdx.y) * 256 + (int)threadIdx.x);
}
__global__ void k_zero_words(unsigned* p, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
int launch_zero(void* p, int64_t bytes, hipStream_t st) {
    if (bytes <= 0) return 0;
    if (bytes & 3) return nvfi_fail(2, "launch_zero: %lld bytes is not a multiple of 4", (long long)bytes);
    const int64_t n = bytes / 4;
    hipLaunchKernelGGL(k_zero_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(p), n);
    LAUNCHCK();
    return 0;
}
int launch_pack_all(const PackJobsAll& jobs, const X6PackArgs* x6, hipStream_t st) {
    if (jobs.n == 0 && !x6) return 0;
    X6PackArgs xa; memset(&xa, 0, sizeof(xa));
    if (x6) xa = *x6;
    hipLaunchKernelGGL(k_pack_all, dim3(jobs.n + (x6 ? X6_PACK_WGX : 0), 8), dim3(256), 0, st, jobs, xa);
    LAUNCHCK();
    return 0;
}

// G[pA][pB] = sum over tiles of A[pA][j] B[pB][j]  (p-space rows of the stash images).
// No LDS, no barriers: the MFMA contraction index is the SAMPLE, and a stash row already holds the 32
// samples of one p contiguously, so lane (i,h) reads row i, samples 16h..16h+15 (4 x 16-byte loads) and
// MFMA step st pairs sample 16h+st of both operands.  Each wave is an independent worker that owns MT
// row tiles x KT column tiles of G for a strided subset of the sample tiles and writes its own slab part;
// k_wgrad_reduce sums the slabs and un-permutes into the logical gradient tensors.
//
// Measured facts that shape the loop (tools/probes/):
//  * the kernel is bound by the LATENCY of the stash reads, not their bandwidth or pattern (cache-hot and fully
//    coalesced variants run at the same speed), so the operands of the next item must be in flight while MFMAs run;
//  * a wave does not issue its own VALU instructions in the shadow of its own MFMAs (64 cycles per 32x32x2 with nothing
//    between, +5 cycles per interleaved VALU instruction; a second wave on the SIMD hides only half of that), so the
//    non-MFMA work per item is kept minimal: wave-uniform addressing in SGPRs, activations only on the B side.
// One wave per SIMD with the whole 512-register file: two complete operand sets alternate (explicitly, so the register
// allocator cannot merge them) - while the 128 MFMAs of one item read one set, the loads of the next item fill the other.
// (Quarter-tile rolling refills at 2 waves/SIMD were tried: 4x2 tiles spill, 2x2 tiles were 20 % slower overall.  Half-tile items
// with four rotating 48-register sets - no AGPR<->VGPR copies left in the loop, three items in flight - measured 6 % slower than
// this whole-tile loop: neither the copies nor the prefetch depth is what limits it.)
#ifndef WGRAD_MT
#define WGRAD_MT 4        // row tiles per worker for 128-row A images
#endif
template <int MT, int KTW, bool TAN>
struct WgradOps { float4 a[MT][4]; float4 b[KTW][4]; float4 b2[TAN ? KTW : 1][4]; };

template <int MT, int KTW, bool TAN>
__device__ __forceinline__ void wgrad_load(WgradOps<MT, KTW, TAN>& o, const WgradJob& J, int item, int ntiles, const int (&aoff)[MT], const int (&boff)[KTW]) {
    const int rep = item / ntiles, tile = item - rep * ntiles;
    const float* At = J.A + (size_t)rep * J.a_rep_stride + (size_t)tile * J.a_tile_stride;
    const float* Bt = J.B + (size_t)rep * J.b_rep_stride + (size_t)tile * J.b_tile_stride;
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
        for (int v = 0; v < 4; ++v) o.b[kt][v] = *reinterpret_cast<const float4*>(Bt + boff[kt] + 4 * v);
    if (TAN) {
        const float* B2t = J.B2 + (size_t)rep * J.b2_rep_stride + (size_t)tile * J.b_tile_stride;
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
            for (int v = 0; v < 4; ++v) o.b2[kt][v] = *reinterpret_cast<const float4*>(B2t + boff[kt] + 4 * v);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) o.a[mt][v] = *reinterpret_cast<const float4*>(At + aoff[mt] + 4 * v);
}

template <int MT, int KTW, int BM>   // MT: row tiles of A, KTW: column tiles of B owned by one worker, BM: how B is formed from the stash
__device__ __forceinline__ void wgrad_worker(const WgradJob& J, int worker, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const int kparts = (J.b_regs >> 4) / KTW, mparts = (J.a_regs >> 4) / MT;   // workers per slab = mparts * kparts
    const int kpart = worker % kparts, mpart = (worker / kparts) % mparts, wslot = worker / (kparts * mparts);
    if (wslot >= J.nslab) return;
    const int a_rows = 2 * J.a_regs, b_rows = 2 * J.b_regs;
    int count = *J.count;
    int ntiles = (count + TILE - 1) / TILE;
    if (ntiles > J.cap_tiles) ntiles = J.cap_tiles;
    const int nitems = J.nrep * ntiles;
    constexpr bool tan = BM == BM_SILU_TAN || BM == BM_RELU_TAN;
    f32x16 acc[MT][KTW];
    float bsum[MT];            // bias sums (column-part worker 0 only)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        bsum[mt] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][kt][r] = 0.f;
    }
    // float offset of (row p, sample 16h) inside a tile image: p = 2*reg + hh -> reg*64 + hh*32
    int aoff[MT], boff[KTW];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { const int p = 32 * (mpart * MT + mt) + i; aoff[mt] = (p >> 1) * 64 + (p & 1) * 32 + 16 * h; }
#pragma unroll
    for (int kt = 0; kt < KTW; ++kt) { const int p = 32 * (kpart * KTW + kt) + i; boff[kt] = (p >> 1) * 64 + (p & 1) * 32 + 16 * h; }
    auto consume = [&](const WgradOps<MT, KTW, tan>& cur) {
        float b[KTW][16];
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 t4 = cur.b[kt][v];
                float z[4] = {t4.x, t4.y, t4.z, t4.w};
                if (BM == BM_SILU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) z[c] = act_f<1>(z[c]);
                } else if (BM == BM_RELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) z[c] = act_f<0>(z[c]);
                } else if (tan) {
                    const float4 u4 = cur.b2[kt][v];
                    const float z2[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) z[c] = (BM == BM_SILU_TAN ? act_d1<1>(z[c]) : act_d1<0>(z[c])) * z2[c];
                }
                b[kt][4 * v] = z[0]; b[kt][4 * v + 1] = z[1]; b[kt][4 * v + 2] = z[2]; b[kt][4 * v + 3] = z[3];
            }
        }
        if (kpart == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int v = 0; v < 4; ++v) bsum[mt] += (cur.a[mt][v].x + cur.a[mt][v].y) + (cur.a[mt][v].z + cur.a[mt][v].w);
        }
#pragma unroll
        for (int st = 0; st < 16; ++st)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float4 a4 = cur.a[mt][st >> 2];
                const float av = (st & 3) == 0 ? a4.x : (st & 3) == 1 ? a4.y : (st & 3) == 2 ? a4.z : a4.w;
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt) acc[mt][kt] = MFMA32(av, b[kt][st], acc[mt][kt]);
            }
    };
    WgradOps<MT, KTW, tan> o0, o1;
    if (wslot < nitems) {
        // prefetches are unconditional (the last one re-reads a valid item) so that the loop is straight-line code and the
        // compiler's vmcnt bookkeeping leaves exactly the newest set in flight
        // single-exit loop: with `break`s inside, the compiler copied all 128 accumulators AGPR -> VGPR in every trip for the exit paths
        const int last = nitems - 1, stp = J.nslab;
        const int n_my = (nitems - wslot + stp - 1) / stp;
        auto it_of = [&](int k) { const int it = wslot + k * stp; return it < last ? it : last; };
        wgrad_load<MT, KTW, tan>(o0, J, it_of(0), ntiles, aoff, boff);
#pragma unroll 1
        for (int k = 0; k < n_my; k += 2) {
            wgrad_load<MT, KTW, tan>(o1, J, it_of(k + 1), ntiles, aoff, boff);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (the scheduler would sink it to save registers)
            consume(o0);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < n_my) {
                wgrad_load<MT, KTW, tan>(o0, J, it_of(k + 2), ntiles, aoff, boff);
                __builtin_amdgcn_sched_barrier(0);
                consume(o1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float* S = J.slabs + (size_t)wslot * ((size_t)a_rows * b_rows + a_rows);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (mpart * MT + mt) + (r & 3) + 8 * (r >> 2) + 4 * h;
                S[(size_t)row * b_rows + 32 * (kpart * KTW + kt) + i] = acc[mt][kt][r];
            }
        if (kpart == 0) {
            float bs = bsum[mt];
            bs += __shfl_xor(bs, 32);
            if (h == 0) S[(size_t)a_rows * b_rows + 32 * (mpart * MT + mt) + i] = bs;
        }
    }
}

__global__ __launch_bounds__(WG_THREADS, 1) void k_wgrad(WgradJobs jobs) {
    const WgradJob& J = jobs.j[blockIdx.y];
    const int worker = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: item addressing stays in SGPRs
    const int lane = threadIdx.x & 63;
#define WGRAD_MODES(MT_, KT_)                                                             \
    switch (J.bmode) {                                                                    \
        case BM_RAW: wgrad_worker<MT_, KT_, BM_RAW>(J, worker, lane); break;              \
        case BM_SILU: wgrad_worker<MT_, KT_, BM_SILU>(J, worker, lane); break;            \
        case BM_RELU: wgrad_worker<MT_, KT_, BM_RELU>(J, worker, lane); break;            \
        case BM_SILU_TAN: wgrad_worker<MT_, (MT_ > 2 ? 1 : KT_), BM_SILU_TAN>(J, worker, lane); break;    \
        default: wgrad_worker<MT_, (MT_ > 2 ? 1 : KT_), BM_RELU_TAN>(J, worker, lane); break;             \
    }
    if (J.a_regs == 64 && J.b_regs == 64) { WGRAD_MODES(WGRAD_MT, 2) }
    else if (J.a_regs == 16 && J.b_regs == 64) { WGRAD_MODES(1, 2) }   // 2 workers per slab
    else if (J.a_regs == 64 && J.b_regs == 16) wgrad_worker<WGRAD_MT, 1, BM_RAW>(J, worker, lane);
    else if (J.a_regs == 16 && J.b_regs == 32) wgrad_worker<1, 2, BM_RAW>(J, worker, lane);
}

#ifndef REDUCE_BLOCKS
#define REDUCE_BLOCKS 260
#endif
// 128 G entries per workgroup as 32 float4 columns; 8 thread groups each sum an eighth of the slabs (16-byte loads, eight in flight per
// thread: the 4-byte, one-at-a-time version ran at 2.8 TB/s), then thread (g < 4, q) finishes component g of column q
__global__ __launch_bounds__(256) void k_wgrad_reduce(ReduceJobs jobs) {
    __shared__ float4 part[8][32];
    const ReduceJob& J = jobs.j[blockIdx.y];
    const int a_rows = 32 * J.MTA, b_rows = 32 * J.KTB;
    const size_t slab = (size_t)a_rows * b_rows + a_rows;       // (a multiple of 32 floats)
    const int total = a_rows * b_rows + a_rows;
    const int q = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int start = J.gW ? 0 : a_rows * b_rows;   // bias-only job: skip the weight entries
    for (int base = start + blockIdx.x * 128; base < total; base += gridDim.x * 128) {
        const int idx4 = base + 4 * q;
        float4 s = zero4();
        if (idx4 < total) {
            auto sum = [&](const float* sl, int ns) {
                int k = g;
                for (; k + 56 < ns; k += 64) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ld4(sl + (size_t)(k + 8 * u) * slab + idx4);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
                }
                for (; k < ns; k += 8) { const float4 v = ld4(sl + (size_t)k * slab + idx4); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            };
            sum(J.slabs, J.nslab);
            if (J.slabs2) sum(J.slabs2, J.nslab2);
        }
        __syncthreads();
        part[g][q] = s;
        __syncthreads();
        const int idx = idx4 + g;
        if (g < 4 && idx4 < total) {
            const float* pp = reinterpret_cast<const float*>(&part[0][q]) + g;
            float r = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) r += pp[u * 32 * 4];
            int pA, pB = -1;
            if (idx < a_rows * b_rows) { pA = idx / b_rows; pB = idx - pA * b_rows; }
            else pA = idx - a_rows * b_rows;
            // G rows/cols are p-space indices (p = 2*reg + h) of the A / B stash images
            const int o = row_logical(J.row_kind, dmap(pA >> 1, pA & 1));
            if (o >= 0 && o < J.out) {
                if (pB >= 0) {
                    const int in = slot_logical(J.slot_kind, pB);
                    // atomic: launches on different streams (the two renders and the PDE term of a step) may accumulate into the same gradient
                    if (in >= 0 && in < J.in && J.gW) atomicAdd(&J.gW[(size_t)o * J.in + in], J.scale * r);
                } else if (J.gb) atomicAdd(&J.gb[o], J.scale * r);
            }
        }
    }
}

static_assert(sizeof(WgradJobs) <= 4000 && sizeof(ReduceJobs) <= 4000 && sizeof(PackJobs) <= 4000 && sizeof(PackJobsAll) + sizeof(X6PackArgs) <= 4000, "kernel argument blocks must stay under 4 KB");

// ---------------------------------------------------------------- host launchers (kept in this TU: no relocatable device code needed)
int launch_pack(const PackJobs& jobs, hipStream_t st) {
    if (jobs.n == 0) return 0;
    hipLaunchKernelGGL(k_pack, dim3(jobs.n, 8), dim3(256), 0, st, jobs);
    LAUNCHCK();
    return 0;
}
int launch_wgrad(const WgradJobs& wj, const ReduceJobs& rj, hipStream_t st) {
    if (wj.n == 0) {          // every slab set was written by a fused adjoint kernel: only the reduce is left
        if (rj.n) { ProfScope ps(PK_WGRAD, st); hipLaunchKernelGGL(k_wgrad_reduce, dim3(REDUCE_BLOCKS, rj.n), dim3(256), 0, st, rj); LAUNCHCK(); }
        return 0;
    }
    // Each job's nslab is the capacity of its slab buffer.
    // NVFI_WGRAD_WAVES caps the number of workers and shares them between the jobs in proportion to their work; unset, every
    // job gets as many slabs as its buffer holds (measured best: many short workers balance the ragged job mix).
    static int resident = 0;
    if (!resident) {
        const char* e = getenv("NVFI_WGRAD_WAVES");
        resident = (e && atoi(e) > 0) ? atoi(e) : (1 << 28);
    }
    WgradJobs bj = wj; ReduceJobs br = rj;
    // default: the LDS-ring kernel (wgrad_ring.hip); NVFI_WGRAD=engine keeps the register-operand kernel of rounds 1-2
    static int ring = -1;
    if (ring < 0) { const char* e = getenv("NVFI_WGRAD"); ring = (e && !strcmp(e, "engine")) ? 0 : 1; }
    if (ring) {
        ProfScope ps(PK_WGRAD, st);
        if (launch_wgrad_ring(bj, br, st)) return 1;
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(REDUCE_BLOCKS, br.n), dim3(256), 0, st, br);
        LAUNCHCK();
        return 0;
    }
    double cost[MAX_WGRAD_JOBS], total = 0.0; int wps[MAX_WGRAD_JOBS];
    for (int i = 0; i < bj.n; ++i) {
        const WgradJob& J = bj.j[i];
        const bool ok = (J.a_regs == 64 && J.b_regs == 64) || (J.a_regs == 16 && J.b_regs == 64) || (J.a_regs == 64 && J.b_regs == 16) || (J.a_regs == 16 && J.b_regs == 32);
        if (!ok) return nvfi_fail(5, "k_wgrad: unsupported tile shape a_regs=%d b_regs=%d", J.a_regs, J.b_regs);
        // tangent jobs carry a second B stream and ~12 VALU instructions per B value (act'(z) * zd): 4 row tiles x ONE column tile per
        // worker, so every B tile is activated by exactly one worker (2 x 2 tiles activated each B tile twice; 4 x 2 spills)
        const bool tanm = J.bmode == BM_SILU_TAN || J.bmode == BM_RELU_TAN;
        const int mt = J.a_regs == 64 ? WGRAD_MT : 1, ktw = (J.b_regs == 16 || (tanm && J.a_regs == 64)) ? 1 : 2;
        wps[i] = ((J.a_regs >> 4) / mt) * ((J.b_regs >> 4) / ktw);
        cost[i] = (double)(J.nrep > 0 ? J.nrep : 1) * wps[i] * (mt + ktw * (J.B2 ? 2 : 1));
        total += cost[i];
    }
    int nworkers = 0;
    for (int i = 0; i < bj.n; ++i) {
        WgradJob& J = bj.j[i];
        int ns = (int)((double)resident * cost[i] / total / wps[i]);
        ns = ns < 1 ? 1 : ns; ns = ns > J.nslab ? J.nslab : ns;
        for (int k = 0; k < br.n; ++k) {
            if (br.j[k].slabs == J.slabs) br.j[k].nslab = ns;
            if (br.j[k].slabs2 == J.slabs) br.j[k].nslab2 = ns;
        }
        J.nslab = ns;
        const int w = ns * wps[i];
        nworkers = w > nworkers ? w : nworkers;
    }
    ProfScope ps(PK_WGRAD, st);
    hipLaunchKernelGGL(k_wgrad, dim3((nworkers + 3) / 4, bj.n), dim3(WG_THREADS), 0, st, bj);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(REDUCE_BLOCKS, br.n), dim3(256), 0, st, br);
    LAUNCHCK();
    return 0;
}
