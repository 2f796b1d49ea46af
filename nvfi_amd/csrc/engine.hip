// engine.hip - fragment packing, split-K weight-gradient kernel, slab reduce.  See engine.h.
#include "engine.h"
#include "common.h"

__global__ void k_pack(PackJobs jobs) {
    const PackJob& J = jobs.j[blockIdx.x];
    const int total = J.MT * J.NS * 64;
    for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += gridDim.y * blockDim.x) {
        int lane = idx & 63, ms = idx >> 6;
        int s = ms % J.NS, m = ms / J.NS;
        int i = lane & 31, h = lane >> 5;
        int rowL = row_logical(J.row_kind, 32 * m + i);
        int colL = slot_logical(J.slot_kind, 2 * s + h);
        float v = 0.f;
        if (!J.transposed) {
            if (rowL >= 0 && rowL < J.out && colL >= 0 && colL < J.in) v = J.W[(size_t)rowL * J.in + colL];
        } else {
            if (rowL >= 0 && rowL < J.in && colL >= 0 && colL < J.out) v = J.W[(size_t)colL * J.in + rowL];
        }
        J.frag[idx] = v;
    }
    if (J.bfrag && blockIdx.y == 0) {
        for (int rho = threadIdx.x; rho < J.MT * 32; rho += blockDim.x) {
            int rl = row_logical(J.row_kind, rho);
            J.bfrag[rho] = (J.b && rl >= 0 && rl < J.out) ? J.b[rl] : 0.f;
        }
    }
}

// G[pA][pB] = sum over tiles of A[pA][j] B[pB][j]  (p-space rows of the stash images).
// No LDS, no barriers: the MFMA contraction index is the SAMPLE, and a stash row already holds the 32
// samples of one p contiguously, so lane (i,h) reads row i, samples 16h..16h+15 (4 x 16-byte loads) and
// MFMA step st pairs sample 16h+st of both operands.  Each wave is an independent worker that owns MT
// row tiles x KT column tiles of G for a strided subset of the sample tiles and writes its own slab part;
// k_wgrad_reduce sums the slabs and un-permutes into the logical gradient tensors.
#ifndef WGRAD_MT
#define WGRAD_MT 4        // row tiles per worker for 128-row A images (4: 2 workers/slab, 2 waves/SIMD; 2: 4 workers/slab, 3 waves/SIMD)
#endif
#define WGRAD_WAVES (WGRAD_MT == 4 ? 2 : 3)
template <int MT, int KTW>   // MT: row tiles of A, KTW: column tiles of B owned by one worker
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
    f32x16 acc[MT][KTW];
    float bsum[MT];
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
    for (int item = wslot; item < nitems; item += J.nslab) {
        const int rep = item / ntiles, tile = item - rep * ntiles;
        const float* At = J.A + (size_t)rep * J.a_rep_stride + (size_t)tile * J.a_tile_stride;
        const float* Bt = J.B + (size_t)rep * J.b_rep_stride + (size_t)tile * J.b_tile_stride;
        const float* B2t = J.B2 ? J.B2 + (size_t)rep * J.b2_rep_stride + (size_t)tile * J.b_tile_stride : nullptr;
        float b[KTW][16];
#pragma unroll
        for (int kt = 0; kt < KTW; ++kt) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 t4 = *reinterpret_cast<const float4*>(Bt + boff[kt] + 4 * v);
                float z[4] = {t4.x, t4.y, t4.z, t4.w};
                if (J.bmode == BM_SILU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) z[c] = act_f<1>(z[c]);
                } else if (J.bmode == BM_RELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) z[c] = act_f<0>(z[c]);
                } else if (J.bmode == BM_SILU_TAN || J.bmode == BM_RELU_TAN) {
                    const float4 u4 = *reinterpret_cast<const float4*>(B2t + boff[kt] + 4 * v);
                    const float z2[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) z[c] = (J.bmode == BM_SILU_TAN ? act_d1<1>(z[c]) : act_d1<0>(z[c])) * z2[c];
                }
                b[kt][4 * v] = z[0]; b[kt][4 * v + 1] = z[1]; b[kt][4 * v + 2] = z[2]; b[kt][4 * v + 3] = z[3];
            }
        }
        // A operand in groups of two row tiles (keeps the live set under 256 VGPRs at 2 waves/SIMD)
        constexpr int MG = MT >= 2 ? 2 : 1;
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += MG) {
            float a[MG][16];
#pragma unroll
            for (int mm = 0; mm < MG; ++mm) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float4 t4 = *reinterpret_cast<const float4*>(At + aoff[m0 + mm] + 4 * v);
                    a[mm][4 * v] = t4.x; a[mm][4 * v + 1] = t4.y; a[mm][4 * v + 2] = t4.z; a[mm][4 * v + 3] = t4.w;
                }
                if (kpart == 0) {
#pragma unroll
                    for (int v = 0; v < 16; ++v) bsum[m0 + mm] += a[mm][v];
                }
            }
#ifdef NVFI_EXP_WGRAD_NOMFMA   // timing experiment: loads only (keep the operands alive)
#pragma unroll
            for (int mm = 0; mm < MG; ++mm)
#pragma unroll
                for (int kt = 0; kt < KTW; ++kt)
#pragma unroll
                    for (int st = 0; st < 16; ++st) acc[m0 + mm][kt][st] += a[mm][st] * b[kt][st];
#else
#pragma unroll
            for (int st = 0; st < 16; ++st)
#pragma unroll
                for (int mm = 0; mm < MG; ++mm)
#pragma unroll
                    for (int kt = 0; kt < KTW; ++kt) acc[m0 + mm][kt] = MFMA32(a[mm][st], b[kt][st], acc[m0 + mm][kt]);
#endif
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
            float bs = bsum[mt] + __shfl_xor(bsum[mt], 32);
            if (h == 0) S[(size_t)a_rows * b_rows + 32 * (mpart * MT + mt) + i] = bs;
        }
    }
}

__global__ __launch_bounds__(WG_THREADS, WGRAD_WAVES) void k_wgrad(WgradJobs jobs) {
    const WgradJob& J = jobs.j[blockIdx.y];
    const int worker = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (J.a_regs == 64 && J.b_regs == 64) wgrad_worker<WGRAD_MT, 2>(J, worker, lane);
    else if (J.a_regs == 16 && J.b_regs == 64) wgrad_worker<1, 2>(J, worker, lane);   // 2 workers per slab
    else if (J.a_regs == 64 && J.b_regs == 16) wgrad_worker<WGRAD_MT, 1>(J, worker, lane);
    else if (J.a_regs == 16 && J.b_regs == 32) wgrad_worker<1, 2>(J, worker, lane);
}

// 64 G entries per workgroup; 4 threads per entry each sum a quarter of the slabs
__global__ __launch_bounds__(256) void k_wgrad_reduce(ReduceJobs jobs) {
    __shared__ float part[4][64];
    const ReduceJob& J = jobs.j[blockIdx.y];
    const int a_rows = 32 * J.MTA, b_rows = 32 * J.KTB;
    const size_t slab = (size_t)a_rows * b_rows + a_rows;
    const int total = a_rows * b_rows + a_rows;
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int start = J.gW ? 0 : a_rows * b_rows;   // bias-only job: skip the weight entries
    for (int base = start + blockIdx.x * 64; base < total; base += gridDim.x * 64) {
        const int idx = base + e;
        float s = 0.f;
        if (idx < total) {
            for (int k = g; k < J.nslab; k += 4) s += J.slabs[(size_t)k * slab + idx];
            if (J.slabs2) for (int k = g; k < J.nslab2; k += 4) s += J.slabs2[(size_t)k * slab + idx];
        }
        __syncthreads();
        part[g][e] = s;
        __syncthreads();
        if (g == 0 && idx < total) {
            s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
            int pA, pB = -1;
            if (idx < a_rows * b_rows) { pA = idx / b_rows; pB = idx - pA * b_rows; }
            else pA = idx - a_rows * b_rows;
            // G rows/cols are p-space indices (p = 2*reg + h) of the A / B stash images
            const int o = row_logical(J.row_kind, dmap(pA >> 1, pA & 1));
            if (o >= 0 && o < J.out) {
                if (pB >= 0) {
                    const int in = slot_logical(J.slot_kind, pB);
                    if (in >= 0 && in < J.in && J.gW) J.gW[(size_t)o * J.in + in] += J.scale * s;
                } else if (J.gb) J.gb[o] += J.scale * s;
            }
        }
    }
}

static_assert(sizeof(WgradJobs) <= 4000 && sizeof(ReduceJobs) <= 4000 && sizeof(PackJobs) <= 4000, "kernel argument blocks must stay under 4 KB");

// ---------------------------------------------------------------- host launchers (kept in this TU: no relocatable device code needed)
int launch_pack(const PackJobs& jobs, hipStream_t st) {
    if (jobs.n == 0) return 0;
    hipLaunchKernelGGL(k_pack, dim3(jobs.n, 8), dim3(256), 0, st, jobs);
    LAUNCHCK();
    return 0;
}
int launch_wgrad(const WgradJobs& wj, const ReduceJobs& rj, hipStream_t st) {
    if (wj.n == 0) return 0;
    int nworkers = 0;
    for (int i = 0; i < wj.n; ++i) {
        const WgradJob& J = wj.j[i];
        const bool ok = (J.a_regs == 64 && J.b_regs == 64) || (J.a_regs == 16 && J.b_regs == 64) || (J.a_regs == 64 && J.b_regs == 16) || (J.a_regs == 16 && J.b_regs == 32);
        if (!ok) return nvfi_fail(5, "k_wgrad: unsupported tile shape a_regs=%d b_regs=%d", J.a_regs, J.b_regs);
        const int w = J.nslab * (J.b_regs == 64 ? 2 : 1) * (J.a_regs == 64 ? 4 / WGRAD_MT : 1);
        nworkers = w > nworkers ? w : nworkers;
    }
    ProfScope ps(PK_WGRAD, st);
    hipLaunchKernelGGL(k_wgrad, dim3((nworkers + 3) / 4, wj.n), dim3(WG_THREADS), 0, st, wj);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(260, rj.n), dim3(256), 0, st, rj);
    LAUNCHCK();
    return 0;
}
