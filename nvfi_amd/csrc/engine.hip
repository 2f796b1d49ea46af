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

// G[pA][pB] = sum over tiles of A[pA][j] B[pB][j]; one slab per workgroup, reduced by k_wgrad_reduce.
__global__ __launch_bounds__(WG_THREADS) void k_wgrad(WgradJobs jobs) {
    const WgradJob& J = jobs.j[blockIdx.y];
    if ((int)blockIdx.x >= J.nslab) return;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int a_rows = 2 * J.a_regs, b_rows = 2 * J.b_regs;
    float* la = lds;
    float* lb = lds + a_rows * 33;
    const int KTB = b_rows >> 5, NTT = (a_rows >> 5) * KTB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    int count = *J.count;
    int ntiles = (count + TILE - 1) / TILE;
    if (ntiles > J.cap_tiles) ntiles = J.cap_tiles;
    const int nitems = J.nrep * ntiles;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float bsum = 0.f;
    int aoff[4], boff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int tt = w + 4 * q;
        int m = tt / KTB, kt = tt % KTB;
        aoff[q] = (32 * m + i) * 33 + h;
        boff[q] = (32 * kt + i) * 33 + h;
    }
    for (int item = blockIdx.x; item < nitems; item += J.nslab) {
        int rep = item / ntiles, tile = item - rep * ntiles;
        const float* At = J.A + (size_t)rep * J.a_rep_stride + (size_t)tile * J.a_tile_stride;
        const float* Bt = J.B + (size_t)rep * J.b_rep_stride + (size_t)tile * J.b_tile_stride;
        const float* B2t = J.B2 ? J.B2 + (size_t)rep * J.b2_rep_stride + (size_t)tile * J.b_tile_stride : nullptr;
        __syncthreads();
        for (int e = tid; e < J.a_regs * 64; e += WG_THREADS) {
            int reg = e >> 6, ln = e & 63;
            la[(2 * reg + (ln >> 5)) * 33 + (ln & 31)] = At[e];
        }
        for (int e = tid; e < J.b_regs * 64; e += WG_THREADS) {
            int reg = e >> 6, ln = e & 63;
            float v = Bt[e];
            switch (J.bmode) {
                case BM_SILU: v = act_f<1>(v); break;
                case BM_RELU: v = act_f<0>(v); break;
                case BM_SILU_TAN: v = act_d1<1>(v) * B2t[e]; break;
                case BM_RELU_TAN: v = act_d1<0>(v) * B2t[e]; break;
                default: break;
            }
            lb[(2 * reg + (ln >> 5)) * 33 + (ln & 31)] = v;
        }
        __syncthreads();
        if (tid < a_rows) {
            float s = 0.f;
#pragma unroll 8
            for (int j = 0; j < 32; ++j) s += la[tid * 33 + j];
            bsum += s;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (w + 4 * q < NTT) {
#pragma unroll
                for (int st = 0; st < 16; ++st) acc[q] = MFMA32(la[aoff[q] + 2 * st], lb[boff[q] + 2 * st], acc[q]);
            }
        }
    }
    float* S = J.slabs + (size_t)blockIdx.x * ((size_t)a_rows * b_rows + a_rows);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int tt = w + 4 * q;
        if (tt < NTT) {
            int m = tt / KTB, kt = tt % KTB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
                S[(size_t)row * b_rows + 32 * kt + i] = acc[q][r];
            }
        }
    }
    if (tid < a_rows) S[(size_t)a_rows * b_rows + tid] = bsum;
}

__global__ void k_wgrad_reduce(ReduceJobs jobs) {
    const ReduceJob& J = jobs.j[blockIdx.y];
    const int a_rows = 32 * J.MTA, b_rows = 32 * J.KTB;
    const size_t slab = (size_t)a_rows * b_rows + a_rows;
    const int total = a_rows * b_rows + a_rows;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int pA, pB = -1;
        if (idx < a_rows * b_rows) { pA = idx / b_rows; pB = idx - pA * b_rows; }
        else pA = idx - a_rows * b_rows;
        // G rows/cols are p-space indices (p = 2*reg + h) of the A / B stash images
        int o = row_logical(J.row_kind, dmap(pA >> 1, pA & 1));
        if (o < 0 || o >= J.out) continue;
        float* dst;
        if (pB >= 0) {
            int in = slot_logical(J.slot_kind, pB);
            if (in < 0 || in >= J.in || !J.gW) continue;
            dst = J.gW + (size_t)o * J.in + in;
        } else {
            if (!J.gb) continue;
            dst = J.gb + o;
        }
        float s = 0.f;
        for (int k = 0; k < J.nslab; ++k) s += J.slabs[(size_t)k * slab + idx];
        *dst += J.scale * s;
    }
}

// ---------------------------------------------------------------- host launchers (kept in this TU: no relocatable device code needed)
int launch_pack(const PackJobs& jobs, hipStream_t st) {
    if (jobs.n == 0) return 0;
    hipLaunchKernelGGL(k_pack, dim3(jobs.n, 8), dim3(256), 0, st, jobs);
    LAUNCHCK();
    return 0;
}
int launch_wgrad(const WgradJobs& wj, const ReduceJobs& rj, hipStream_t st) {
    if (wj.n == 0) return 0;
    static bool attr = false;
    const int lds_bytes = 2 * 128 * 33 * 4;
    if (!attr) {
        HIPCK(hipFuncSetAttribute((const void*)k_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr = true;
    }
    int nslab = 0;
    for (int i = 0; i < wj.n; ++i) nslab = wj.j[i].nslab > nslab ? wj.j[i].nslab : nslab;
    ProfScope ps(PK_WGRAD, st);
    hipLaunchKernelGGL(k_wgrad, dim3(nslab, wj.n), dim3(WG_THREADS), lds_bytes, st, wj);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(32, rj.n), dim3(256), 0, st, rj);
    LAUNCHCK();
    return 0;
}
