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

// Weight gradients: G[pA][pB] = sum over tiles of A[pA][j] B[pB][j]  (p-space rows of the stash images).  The contraction index of the MFMAs is
// the SAMPLE; a stash row already holds the 32 samples of one p contiguously.  The split-K kernel is k_wgrad_ring8 (wgrad_ring.hip): workers own
// slabs of G over a strided subset of the sample tiles, k_wgrad_reduce below sums the slabs and un-permutes into the logical gradient tensors.
// (The register-operand kernel k_wgrad of rounds 1-2 - one wave per SIMD, two alternating operand sets in 512 registers, latency-bound on the stash
// reads - was retired in round 6: same numbers to rounding, same time, docs/experiments_r03.md item 2 keeps its measurements.)

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
    // Each job's nslab is the capacity of its slab buffer; the launcher sets the number of slabs written.
    WgradJobs bj = wj; ReduceJobs br = rj;
    ProfScope ps(PK_WGRAD, st);
    if (launch_wgrad_ring(bj, br, st)) return 1;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(REDUCE_BLOCKS, br.n), dim3(256), 0, st, br);
    LAUNCHCK();
    return 0;
}
