// scatter.hip - plane-gradient scatter through sorted plane tiles, and the (sample, channel-quad) gather kernels.
//
// Backward of the factor-plane lookups (autograd of models/tensorf_keyframe.py:233-310: F.grid_sample on the six planes of the
// density / appearance factorisation): every sample adds  w_tap * d(loss)/d(value_p[c])  to 4 texels x C channels of each plane.
//
// Measured on MI355X (tools/probes/lds_atomic_probe.hip):
//   * device-scope fp32 atomics run at ~2.6e11 lane-atomics/s for the whole chip whatever the occupancy (they are executed
//     past the XCD L2s), so 4.3e7 taps per call are a 0.17 ms floor;
//   * LDS fp32 atomics (ds_add_f32) take ~190 cycles per 64-lane instruction per CU (one lane every 3 cycles) - 50x slower than
//     a plain LDS read-add-write;
//   * thread-per-sample gathers with a loop over channel quads are latency bound (one memory round trip per quad and plane).
// Hence:
//   k_og          lanes = (sample, channel quad): the 24 tap loads of a lane are unconditional (clamped addresses, masked by a
//                 select) and all in flight together; writes og[i][p][c] = d(loss)/d(value_p[c]) and, for the density branch,
//                 the coordinate gradients (replaces the per-sample loop of k_density_bwd).
//   k_tile_hist (+ the scan, in its last workgroup) / k_tile_fill   counting sort of the samples by TxT-texel tile, once per space plane.
//   k_tile_scatter_mfma (default, NVFI_SCATTER=mfma)   ONE WAVE per (plane, 4x4-texel tile, chunk of <= 128 samples): the scatter of
//                 an item is the small GEMM  G[texel][channel] = sum_s W[texel][s] * og[s][channel]  on v_mfma_f32_16x16x4_f32 (exact
//                 fp32 products, fp32 accumulation): 25 tile + apron texels = two 16-row tiles, the 2x5 strip of the paired time
//                 plane a third, 24 / 48 channels = 2 / 3 column tiles.  The tile gradient lives in accumulators and is flushed
//                 with global atomics from registers; the time strips of the 4 waves of a workgroup - consecutive items, which share
//                 tile columns because the bins of planes 0 and 2 are numbered column by column - are summed in LDS first.
//                 49 / 44 us per call (C = 48 / 24) against 104 / 92 us for the LDS kernel on the stationary bench workload.
//   k_tile_scatter (NVFI_SCATTER=lds, and grids with more than 8192 4x4 tiles)   ONE WAVE per (plane, 8x8-texel tile, chunk of
//                 samples): the tile (+1 texel apron) of the space plane and the matching
//                 strip of the paired time plane live in LDS and are updated with PLAIN read-add-write - a single wave executes
//                 its LDS instructions in order and the lanes of one instruction ((channel, x-tap)) never collide - then flushed
//                 with one contiguous run of global atomics per tile row.  Global atomics drop from 576 per sample to
//                 (tiles x apron) per chunk.  ~50 VALU instructions (4 cycles each) per sample and 24 channels bound it.
// Pairing of time planes: the time plane whose spatial axis is one of the space plane's axes shares that axis' tap index, so its
// two touched rows restricted to the tile are a (T+1)-texel strip:  plane 0 (x,y) <-> time plane 5 (x), plane 1 (x,z) <-> time
// plane 3 (z), plane 2 (y,z) <-> time plane 4 (y).
#include "common.h"
#include "render.h"
#include "scatter.h"

#define TT SCATTER_T
#define TW (TT + 1)

// ---------------------------------------------------------------- branch-free bilinear taps
struct Tap {
    int o[4];      // float offsets of the 4 taps (texel * C), 0 when masked
    float w[4];    // nw, ne, sw, se (un-masked: the loaded value is zeroed instead, as grid_sample's zero padding)
    bool m[4];
    float fw, fe, fn, fs;
};
__device__ __forceinline__ void tap_setup(const Bl& b, int C, Tap& t) {
    const int o0 = b.base * C, oW = b.W * C;
    t.m[0] = b.m0; t.m[1] = b.m1; t.m[2] = b.m2; t.m[3] = b.m3;
    t.o[0] = b.m0 ? o0 : 0; t.o[1] = b.m1 ? o0 + C : 0; t.o[2] = b.m2 ? o0 + oW : 0; t.o[3] = b.m3 ? o0 + oW + C : 0;
    t.w[0] = b.e * b.s; t.w[1] = b.w * b.s; t.w[2] = b.e * b.n; t.w[3] = b.w * b.n;
    t.fw = b.w; t.fe = b.e; t.fn = b.n; t.fs = b.s;
}
__device__ __forceinline__ float4 sel4(bool m, const float4& v) { return m ? v : zero4(); }

// ---------------------------------------------------------------- k_og
// LPS lanes per sample (8 for C=24, 16 for C=48), C/4 of them active.
template <int C, bool COORD>
__global__ __launch_bounds__(256) void k_og(OgArgs a) {
    constexpr int LPS = C == 24 ? 8 : 16, NQ = C / 4, SPW = 256 / LPS;
    const nvfi_field_desc& f = a.f;
    const int count = *a.count;
    const int sub = threadIdx.x % LPS;
    const int i = blockIdx.x * SPW + threadIdx.x / LPS;
    if (blockIdx.x * SPW >= count) return;
    const bool act = i < count && sub < NQ;
    const int ic = i < count ? i : count - 1;
    const int n = a.list[ic];
    const float4 q = a.xw[n];
    const int qd = sub < NQ ? sub : 0;
    float4 g4;
    if (C == 24) { const float g = a.gxpre[n]; g4 = make_float4(g, g, g, g); }
    else g4 = *reinterpret_cast<const float4*>(a.gg + (size_t)ic * 48 + 4 * qd);
    Bl b[6];
    plane_setups(f, q.x, q.y, q.z, SCHED_TN(a), b);
    const float* pl[6];
#pragma unroll
    for (int p = 0; p < 3; ++p) { pl[p] = C == 24 ? f.dps[p] : f.aps[p]; pl[3 + p] = C == 24 ? f.dpt[p] : f.apt[p]; }
    float4 v[6][4];
    Tap t[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        tap_setup(b[p], C, t[p]);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[p][k] = ld4(pl[p] + t[p].o[k] + 4 * qd);
    }
    float4 val[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[p][k] = sel4(t[p].m[k], v[p][k]);
        val[p].x = v[p][0].x * t[p].w[0] + v[p][1].x * t[p].w[1] + v[p][2].x * t[p].w[2] + v[p][3].x * t[p].w[3];
        val[p].y = v[p][0].y * t[p].w[0] + v[p][1].y * t[p].w[1] + v[p][2].y * t[p].w[2] + v[p][3].y * t[p].w[3];
        val[p].z = v[p][0].z * t[p].w[0] + v[p][1].z * t[p].w[1] + v[p][2].z * t[p].w[2] + v[p][3].z * t[p].w[3];
        val[p].w = v[p][0].w * t[p].w[0] + v[p][1].w * t[p].w[1] + v[p][2].w * t[p].w[2] + v[p][3].w * t[p].w[3];
    }
    // o[p] = g * prod_{k != p} val[k]  (prefix / suffix products)
    float4 L[6], Rr[6], o[6];
    L[0] = g4;
#pragma unroll
    for (int p = 1; p < 6; ++p) L[p] = make_float4(L[p - 1].x * val[p - 1].x, L[p - 1].y * val[p - 1].y, L[p - 1].z * val[p - 1].z, L[p - 1].w * val[p - 1].w);
    Rr[5] = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int p = 4; p >= 0; --p) Rr[p] = make_float4(Rr[p + 1].x * val[p + 1].x, Rr[p + 1].y * val[p + 1].y, Rr[p + 1].z * val[p + 1].z, Rr[p + 1].w * val[p + 1].w);
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        o[p] = make_float4(L[p].x * Rr[p].x, L[p].y * Rr[p].y, L[p].z * Rr[p].z, L[p].w * Rr[p].w);
        if (act && a.og) *reinterpret_cast<float4*>(a.og + ((size_t)i * 6 + p) * C + 4 * qd) = o[p];
    }
    if (COORD) {
        float g3[3] = {0.f, 0.f, 0.f};
        if (act) {
            float gx[6], gy[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const float4 &v0 = v[p][0], &v1 = v[p][1], &v2 = v[p][2], &v3 = v[p][3], &g = o[p];
                const float s = t[p].fs, nn = t[p].fn, e = t[p].fe, w = t[p].fw;
                gx[p] = ((v1.x - v0.x) * s + (v3.x - v2.x) * nn) * g.x + ((v1.y - v0.y) * s + (v3.y - v2.y) * nn) * g.y +
                        ((v1.z - v0.z) * s + (v3.z - v2.z) * nn) * g.z + ((v1.w - v0.w) * s + (v3.w - v2.w) * nn) * g.w;
                gy[p] = ((v2.x - v0.x) * e + (v3.x - v1.x) * w) * g.x + ((v2.y - v0.y) * e + (v3.y - v1.y) * w) * g.y +
                        ((v2.z - v0.z) * e + (v3.z - v1.z) * w) * g.z + ((v2.w - v0.w) * e + (v3.w - v1.w) * w) * g.w;
            }
            float mx, my;
            plane_mults(f, 0, mx, my); g3[0] += gx[0] * mx; g3[1] += gy[0] * my;
            plane_mults(f, 1, mx, my); g3[0] += gx[1] * mx; g3[2] += gy[1] * my;
            plane_mults(f, 2, mx, my); g3[1] += gx[2] * mx; g3[2] += gy[2] * my;
            plane_mults(f, 3, mx, my); g3[2] += gx[3] * mx;
            plane_mults(f, 4, mx, my); g3[1] += gx[4] * mx;
            plane_mults(f, 5, mx, my); g3[0] += gx[5] * mx;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 1; d < LPS; d <<= 1) g3[c] += __shfl_xor(g3[c], d);
        if (C == 48) {
            if (i < count && sub == 0 && a.gxw_acc) {
                const float4 g0 = a.gxw_acc[n];
                a.gxw_acc[n] = make_float4(g0.x + g3[0], g0.y + g3[1], g0.z + g3[2], 0.f);
            }
        } else if (i < count && sub == 0 && a.gxk) {
            const float4 ga = (a.mflag && a.mflag[n]) ? a.gxw[n] : zero4();   // appearance-branch part (masked samples only)
            a.gxk[n] = make_float4(ga.x + g3[0], ga.y + g3[1], ga.z + g3[2], 0.f);   // dense (per sample): the RK2 adjoint walks its own list
        }
    }
}

// ---------------------------------------------------------------- k_app_feat
// The 48-channel appearance feature (product of the six plane samples, tensorf_keyframe.py:274-310) of every masked sample, with the lanes
// of k_og: 16 per sample, 12 active, one channel quad each - a wave instruction covers the 192 contiguous bytes of a texel for four
// samples.  Inside k_app_fwd (lane = sample, two quads per pass, six passes) the same taps cost six dependent round trips and used 32 of
// every 128 bytes fetched per pass; feat[i][48] (compact index) is read back there with six 16-byte loads per lane.
__global__ __launch_bounds__(256) void k_app_feat(OgArgs a) {
    constexpr int LPS = 16, NQ = 12, SPW = 256 / LPS;
    const nvfi_field_desc& f = a.f;
    const int count = *a.count;
    const int sub = threadIdx.x % LPS;
    const int i = blockIdx.x * SPW + threadIdx.x / LPS;
    if (blockIdx.x * SPW >= count) return;
    const bool act = i < count && sub < NQ;
    const int ic = i < count ? i : count - 1;
    const int n = a.list[ic];
    const float4 q = a.xw[n];
    const int qd = sub < NQ ? sub : 0;
    Bl b[6];
    plane_setups(f, q.x, q.y, q.z, SCHED_TN(a), b);
    const float* pl[6] = {f.aps[0], f.aps[1], f.aps[2], f.apt[0], f.apt[1], f.apt[2]};
    float4 v[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) v[p] = bl_sample4(pl[p], f.Ca, b[p], qd);
    if (act) {
        float4 o;       // (same association as the in-kernel gather of k_app_fwd)
        o.x = ((v[0].x * v[1].x) * v[2].x) * ((v[3].x * v[4].x) * v[5].x);
        o.y = ((v[0].y * v[1].y) * v[2].y) * ((v[3].y * v[4].y) * v[5].y);
        o.z = ((v[0].z * v[1].z) * v[2].z) * ((v[3].z * v[4].z) * v[5].z);
        o.w = ((v[0].w * v[1].w) * v[2].w) * ((v[3].w * v[4].w) * v[5].w);
        *reinterpret_cast<float4*>(a.og + (size_t)i * 48 + 4 * qd) = o;
    }
}
int launch_app_feat(const OgArgs& oa, int64_t N, hipStream_t st) {
    hipLaunchKernelGGL(k_app_feat, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, st, oa);
    LAUNCHCK();
    return 0;
}

// ---------------------------------------------------------------- k_density_q
// compute_densityfeature + density_shift (+ softplus) with lanes = (sample, channel quad): 8 lanes per sample, 6 active.
__global__ __launch_bounds__(256) void k_density_q(DensityArgs a) {
    const nvfi_field_desc& f = a.f;
    const int count = a.count ? *a.count : (int)a.n_direct;
    if (blockIdx.x * 32 >= count) return;
    const int sub = threadIdx.x & 7;
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool in = i < count;
    const int ic = in ? i : count - 1;
    const int n = a.list ? a.list[ic] : ic;
    const float4 q = a.xw[n];
    const float tn = a.per_point_t ? q.w : SCHED_TN(a);
    const int qd = sub < 6 ? sub : 0;
    Bl b[6];
    plane_setups(f, q.x, q.y, q.z, tn, b);
    const float* pl[6] = {f.dps[0], f.dps[1], f.dps[2], f.dpt[0], f.dpt[1], f.dpt[2]};
    float4 v[6][4];
    Tap t[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        tap_setup(b[p], 24, t[p]);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[p][k] = ld4(pl[p] + t[p].o[k] + 4 * qd);
    }
    float4 val[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[p][k] = sel4(t[p].m[k], v[p][k]);
        val[p].x = v[p][0].x * t[p].w[0] + v[p][1].x * t[p].w[1] + v[p][2].x * t[p].w[2] + v[p][3].x * t[p].w[3];
        val[p].y = v[p][0].y * t[p].w[0] + v[p][1].y * t[p].w[1] + v[p][2].y * t[p].w[2] + v[p][3].y * t[p].w[3];
        val[p].z = v[p][0].z * t[p].w[0] + v[p][1].z * t[p].w[1] + v[p][2].z * t[p].w[2] + v[p][3].z * t[p].w[3];
        val[p].w = v[p][0].w * t[p].w[0] + v[p][1].w * t[p].w[1] + v[p][2].w * t[p].w[2] + v[p][3].w * t[p].w[3];
    }
    float sum = 0.f;
    if (sub < 6) {
        sum += ((val[0].x * val[1].x) * val[2].x) * ((val[3].x * val[4].x) * val[5].x);
        sum += ((val[0].y * val[1].y) * val[2].y) * ((val[3].y * val[4].y) * val[5].y);
        sum += ((val[0].z * val[1].z) * val[2].z) * ((val[3].z * val[4].z) * val[5].z);
        sum += ((val[0].w * val[1].w) * val[2].w) * ((val[3].w * val[4].w) * val[5].w);
    }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) sum += __shfl_xor(sum, d);
    if (in && sub == 0) {
        if (a.feat_out) a.feat_out[n] = sum;
        if (a.xpre) a.xpre[n] = sum + f.density_shift;
        if (a.sigma_out) a.sigma_out[n] = softplus_f(sum + f.density_shift);
    }
}
int launch_density_q(const DensityArgs& da, int64_t N, hipStream_t st) {
    hipLaunchKernelGGL(k_density_q, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, st, da);
    LAUNCHCK();
    return 0;
}

// ---------------------------------------------------------------- tile sort
__device__ __forceinline__ int tile_of(float gx, float gy, int W, int H, int ntx, int nty, int T, bool colmajor) {
    float x = (gx + 1.f) * ((float)(W - 1) / 2.f), y = (gy + 1.f) * ((float)(H - 1) / 2.f);
    float xf = floorf(x), yf = floorf(y);
    xf = fminf(fmaxf(xf, 0.f), (float)(W - 1)); yf = fminf(fmaxf(yf, 0.f), (float)(H - 1));
    if (!(xf == xf)) xf = 0.f;
    if (!(yf == yf)) yf = 0.f;
    const int tx = (int)xf / T, ty = (int)yf / T;
    return colmajor ? tx * nty + ty : ty * ntx + tx;
}
__device__ __forceinline__ void bins_of(const TileGeom& g, const float4& q, int* b) {
    b[0] = tile_of(q.x, q.y, g.G[0], g.G[1], g.ntx[0], g.nty[0], g.T, g.colmajor != 0);
    b[1] = g.boff[1] + tile_of(q.x, q.z, g.G[0], g.G[2], g.ntx[1], g.nty[1], g.T, false);
    b[2] = g.boff[2] + tile_of(q.y, q.z, g.G[1], g.G[2], g.ntx[2], g.nty[2], g.T, g.colmajor != 0);
}

// (round 5: blockIdx.y selects one of up to two sort jobs - the valid-sample list and the appearance-masked list of a render's backward are both
//  known after the forward, so their counting sorts share the two launches)
__device__ void tile_scan_block(const TileSortArgs& a);
__global__ __launch_bounds__(512) void k_tile_hist(TileSortArgs2 a2) {
    const TileSortArgs& a = a2.j[blockIdx.y];
    extern __shared__ int sh[];
    int* h = sh;
    const int nb = a.g.nbins;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) h[k] = 0;
    __syncthreads();
    const int count = *a.count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float4 q = a.xw[a.list[i]];
        int b[3];
        bins_of(a.g, q, b);
        atomicAdd(&h[b[0]], 1); atomicAdd(&h[b[1]], 1); atomicAdd(&h[b[2]], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += blockDim.x) if (h[k]) atomicAdd(&a.hist[k], h[k]);
    // The last workgroup to arrive scans.  No fence: the histogram adds are agent-scope atomics (performed at the coherence point, not in a
    // CU-local cache) that have completed - s_waitcnt vmcnt(0) - before the workgroup's ticket is drawn, and the scan reads the counts with
    // agent-scope atomic loads.  (A __threadfence() per workgroup here - an L2 write-back / invalidate on every XCD - cost the step 10 %.)
    __shared__ int last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(&a.hist[nb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1);
    __syncthreads();
    if (last) tile_scan_block(a);
}

// one workgroup: tile starts (exclusive scan of the bin counts), chunk items, cursors; clears the histogram (and the ticket) for the next
// call.  Runs in the LAST workgroup of k_tile_hist to finish (ticket in hist[nbins]): one launch less per counting sort.
#define SCAN_PAD(k) ((k) + ((k) >> 4))      // LDS position of bin k: a thread's 16 consecutive bins sit 17 words from its neighbour's
__device__ void tile_scan_block(const TileSortArgs& a) {
    // every thread owns SCAN_PT consecutive bins: one block-wide scan of the per-thread sums instead of nbins / 512 dependent rounds.
    // Global memory is touched in coalesced rows only (bin = thread + 512 j), transposed through LDS: with thread-strided accesses
    // every wave instruction of this single workgroup touched 32 cache lines.
    constexpr int SCAN_PT = (SCATTER_MAX_BINS_MFMA + 511) / 512;
    static_assert(SCAN_PT == 16, "SCAN_PAD assumes 16 bins per thread");
    extern __shared__ int sh[];                  // [0, P): counts, then first samples; [P, 2P): first items;  P = SCAN_PAD(nbins) + 1
    __shared__ int wsum[2][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nb = a.g.nbins, chunk = a.g.chunk, csh = a.g.chunk_shift;     // chunk = 1 << csh
    int* sa = sh; int* sb = sh + SCAN_PAD(nb) + 1;
    __syncthreads();                             // (the histogram in sh[] is dead: every thread is past its flush)
#pragma unroll
    for (int j = 0; j < SCAN_PT; ++j) {
        const int k = tid + 512 * j;
        if (k < nb) sa[SCAN_PAD(k)] = __hip_atomic_load(&a.hist[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int k0 = tid * SCAN_PT;
    int c[SCAN_PT];
    int s0 = 0, s1 = 0;
#pragma unroll
    for (int j = 0; j < SCAN_PT; ++j) {
        c[j] = k0 + j < nb ? sa[17 * tid + j] : 0;
        s0 += c[j]; s1 += (c[j] + chunk - 1) >> csh;
    }
    int i0 = s0, i1 = s1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t0 = __shfl_up(i0, o), t1 = __shfl_up(i1, o); if (lane >= o) { i0 += t0; i1 += t1; } }
    if (lane == 63) { wsum[0][w] = i0; wsum[1][w] = i1; }
    __syncthreads();
    int w0 = 0, w1 = 0, t0 = 0, t1 = 0;
    for (int j = 0; j < (int)(blockDim.x >> 6); ++j) { if (j < w) { w0 += wsum[0][j]; w1 += wsum[1][j]; } t0 += wsum[0][j]; t1 += wsum[1][j]; }
    int start = w0 + i0 - s0, istart = w1 + i1 - s1;
#pragma unroll
    for (int j = 0; j < SCAN_PT; ++j) {
        if (k0 + j < nb) { sa[17 * tid + j] = start; sb[17 * tid + j] = istart; }
        start += c[j]; istart += (c[j] + chunk - 1) >> csh;
    }
    __syncthreads();
    // (the item list itself is written by the fill kernel's many workgroups from start / istart)
#pragma unroll
    for (int j = 0; j < SCAN_PT; ++j) {
        const int k = tid + 512 * j;
        if (k < nb) {
            const int st = sa[SCAN_PAD(k)];
            a.cursor[k] = st; a.start[k] = st; a.istart[k] = sb[SCAN_PAD(k)];
            a.hist[k] = 0;
        }
    }
    if (tid == 0) { a.start[nb] = t0; a.istart[nb] = t1; *a.nitems = t1; a.hist[nb] = 0; }
}

__global__ __launch_bounds__(512) void k_tile_fill(TileSortArgs2 a2) {
    const TileSortArgs& a = a2.j[blockIdx.y];
    extern __shared__ int sh[];
    const int nb = a.g.nbins;
    int* cnt = sh; int* bas = sh + nb;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) cnt[k] = 0;
    __syncthreads();
    const int count = *a.count;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int b[3] = {0, 0, 0}, r[3] = {0, 0, 0};
    float4 qq = zero4();
    if (i < count) {
        const float4 q = a.xw[a.list[i]];
        qq = q;
        bins_of(a.g, q, b);
#pragma unroll
        for (int p = 0; p < 3; ++p) r[p] = atomicAdd(&cnt[b[p]], 1);
    }
    __syncthreads();
    {   // one returning atomic per touched bin: all of a thread's requests are in flight before the first result is stored
        constexpr int PT = (SCATTER_MAX_BINS_MFMA + 511) / 512;
        int rr[PT];
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int k = threadIdx.x + j * 512;
            const int c = k < nb ? cnt[k] : 0;
            rr[j] = c ? atomicAdd(&a.cursor[k], c) : 0;
        }
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int k = threadIdx.x + j * 512;
            if (k < nb) bas[k] = rr[j];
        }
    }
    __syncthreads();
    // items of bins blockIdx.x, blockIdx.x + gridDim.x, ...: (bin, first sample, samples, 0) per chunk
    for (int k = blockIdx.x + threadIdx.x * gridDim.x; k < nb; k += gridDim.x * blockDim.x) {
        const int s0 = a.start[k], c = a.start[k + 1] - s0, chunk = a.g.chunk;
        int4* it = a.items + a.istart[k];
        for (int q = 0; q * chunk < c; ++q) it[q] = make_int4(k, s0 + q * chunk, min(chunk, c - q * chunk), 0);
    }
    if (i < count) {
        // record = (sample index, u, v): the two in-plane coordinates of the pass, so the scatter needs no second and third lookup
        a.sorted[bas[b[0]] + r[0]] = make_float4(__int_as_float(i), qq.x, qq.y, 0.f);
        a.sorted[bas[b[1]] + r[1]] = make_float4(__int_as_float(i), qq.x, qq.z, 0.f);
        a.sorted[bas[b[2]] + r[2]] = make_float4(__int_as_float(i), qq.y, qq.z, 0.f);
    }
}

// ---------------------------------------------------------------- tile scatter: one workgroup per item, one private tile per wave
__device__ __forceinline__ float rl_f(float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); }

#ifndef TS_WAVES
#define TS_WAVES 4
#endif
#define TS_U 16      // per-sample gradient rows in flight per wave (rolling prefetch)
static_assert(SCATTER_CHUNK <= 64 * TS_WAVES, "a wave prepares at most 64 samples");
#define TS_SP (TW * TW * 24)
#define TS_TM (2 * TW * 24)
// A wave's share of an item is at most 64 samples, so lane s first prepares sample s on its own (tap addresses inside the LDS tile,
// the four bilinear weights, the tap masks) and the sample loop only broadcasts those ten values (v_readlane) - the per-sample
// instruction count, which every lane of the wave pays, drops ~4x against computing the taps in the loop.  Masked taps are
// redirected to a per-lane dummy word so the read-add-write stays branch-free.
template <int CT>
__global__ __launch_bounds__(64 * TS_WAVES) void k_tile_scatter(TileScatterArgs a) {
    __shared__ float sp_all[TS_WAVES][TS_SP + TS_TM + 64];
    if ((int)blockIdx.x >= *a.nitems) return;
    const nvfi_field_desc& f = a.f;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* sp = sp_all[wv];                 // [0, TS_SP) space tile, [TS_SP, TS_SP+TS_TM) time strip, then 64 dummy words
    const int4 item = a.items[blockIdx.x];
    const int c0 = blockIdx.y * 24;
    const int p = item.x >= a.geo.boff[2] ? 2 : (item.x >= a.geo.boff[1] ? 1 : 0);
    const int tile = item.x - a.geo.boff[p];
    const int ty = tile / a.geo.ntx[p], tx = tile - ty * a.geo.ntx[p];
    const int ox = tx * TT, oy = ty * TT;
    const int ia = p == 2 ? 1 : 0, ib = p == 0 ? 1 : 2;       // matModeSpace axes of plane p
    const int W = f.G[ia], H = f.G[ib];
    const int tp = p == 0 ? 5 : (p == 1 ? 3 : 4);              // paired time plane
    const bool t_on_u = p != 1;                                // the time plane's spatial axis is the u (x-tap) axis of plane p, else v
    const int Wt = t_on_u ? W : H, ot = t_on_u ? ox : oy;
    float* gsp = CT == 24 ? a.g.dps[p] : a.g.aps[p];
    float* gtm = CT == 24 ? a.g.dpt[tp - 3] : a.g.apt[tp - 3];
    for (int k = lane; k < TS_SP + TS_TM + 64; k += 64) sp[k] = 0.f;
    const int ch = lane >> 1, dx = lane & 1;
    const bool lane_on = lane < 48;
    const int chs = lane_on ? ch : 0;
    const int dummy = TS_SP + TS_TM + lane;
    // this wave's share of the item (<= 64 samples): lane s prepares sample s
    const int per = (((item.z + TS_WAVES - 1) / TS_WAVES) + TS_U - 1) / TS_U * TS_U;
    const int w_lo = min(item.z, wv * per), nw = min(item.z, w_lo + per) - w_lo;
    int il = 0, as = 0, at = 0;             // sample index; packed (tile address << 4 | tap masks) for the space tile / time strip
    float ws[4] = {0.f, 0.f, 0.f, 0.f}, wt[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane < nw) {
        const float4 rec = a.sorted[item.y + w_lo + lane];
        il = __float_as_int(rec.x);
        const float uu = rec.y, vv = rec.z;
        Bl b, bt;
        int x0, y0, xt, yt;
        bl_setup_xy(uu, vv, W, H, b, x0, y0);
        bl_setup_xy(t_on_u ? uu : vv, SCHED_TN(a), Wt, f.K, bt, xt, yt);
        ws[0] = b.e * b.s; ws[1] = b.w * b.s; ws[2] = b.e * b.n; ws[3] = b.w * b.n;
        wt[0] = bt.e * bt.s; wt[1] = bt.w * bt.s; wt[2] = bt.e * bt.n; wt[3] = bt.w * bt.n;
        const int ms = (b.m0 ? 1 : 0) | (b.m1 ? 2 : 0) | (b.m2 ? 4 : 0) | (b.m3 ? 8 : 0);
        const int mt = (bt.m0 ? 1 : 0) | (bt.m1 ? 2 : 0) | (bt.m2 ? 4 : 0) | (bt.m3 ? 8 : 0);
        // any un-masked tap implies -1 <= x0-ox <= T-1 and -1 <= y0-oy <= T-1 (tile_of clamps the same floor), so the biased
        // address below is non-negative whenever it is used
        as = ((((y0 - oy) * TW + (x0 - ox)) * 24 + 1024) << 4) | ms;
        at = (((xt - ot) * 24 + 1024) << 4) | mt;
    }
    float gs[TS_U], gt[TS_U];
    auto issue = [&](int s, int slot) {
        const int sc = s < nw ? s : 0;
        const int i = __builtin_amdgcn_readlane(il, sc);
        gs[slot] = a.og[((size_t)i * 6 + p) * CT + c0 + chs];
        gt[slot] = a.og[((size_t)i * 6 + tp) * CT + c0 + chs];
    };
    const int loff = dx * 24 + ch - 1024;
    if (nw > 0) {
#pragma unroll
        for (int j = 0; j < TS_U; ++j) issue(j, j);
#pragma unroll 1
        for (int s0 = 0; s0 < nw; s0 += TS_U) {
#pragma unroll
            for (int j = 0; j < TS_U; ++j) {
                const int s = s0 + j;
                const float g_s = gs[j], g_t = gt[j];
                issue(s + TS_U, j);
                if (s < nw) {
                    const int pas = __builtin_amdgcn_readlane(as, s), pat = __builtin_amdgcn_readlane(at, s);
                    const float w00 = rl_f(ws[0], s), w01 = rl_f(ws[1], s), w10 = rl_f(ws[2], s), w11 = rl_f(ws[3], s);
                    const float t00 = rl_f(wt[0], s), t01 = rl_f(wt[1], s), t10 = rl_f(wt[2], s), t11 = rl_f(wt[3], s);
                    const int a0 = (pas >> 4) + loff, b0 = (pat >> 4) + loff + TS_SP;
                    const bool m0 = lane_on && ((pas >> dx) & 1), m1 = lane_on && ((pas >> (2 + dx)) & 1);
                    const bool n0 = lane_on && ((pat >> dx) & 1), n1 = lane_on && ((pat >> (2 + dx)) & 1);
                    const int i0 = m0 ? a0 : dummy, i1 = m1 ? a0 + TW * 24 : dummy;
                    const int j0 = n0 ? b0 : dummy, j1 = n1 ? b0 + TW * 24 : dummy;
                    const float r0 = sp[i0], r1 = sp[i1], q0 = sp[j0], q1 = sp[j1];
                    // (i0 == i1 or j0 == j1 only on the dummy word, whose value is never used)
                    sp[i0] = r0 + (dx ? w01 : w00) * g_s;
                    sp[i1] = r1 + (dx ? w11 : w10) * g_s;
                    sp[j0] = q0 + (dx ? t01 : t00) * g_t;
                    sp[j1] = q1 + (dx ? t11 : t10) * g_t;
                }
            }
        }
    }
    __syncthreads();
    // flush the sum of the waves' private tiles: contiguous runs of (texel, channel) per tile row
    if (gsp) {
        for (int k = threadIdx.x; k < TW * TW * 24; k += 64 * TS_WAVES) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < TS_WAVES; ++w) v += sp_all[w][k];
            if (v != 0.f) {
                const int c = k % 24, tx2 = (k / 24) % TW, ty2 = k / (24 * TW);
                const int X = ox + tx2, Y = oy + ty2;
                if (X < W && Y < H) atomicAdd(gsp + ((size_t)Y * W + X) * CT + c0 + c, v);
            }
        }
    }
    if (gtm) {
        for (int k = threadIdx.x; k < 2 * TW * 24; k += 64 * TS_WAVES) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < TS_WAVES; ++w) v += sp_all[w][TS_SP + k];
            if (v != 0.f) {
                const int c = k % 24, tx2 = (k / 24) % TW, r = k / (24 * TW);
                const int X = ot + tx2, Y = SCHED_Y0(a) + r;
                if (X < Wt && Y >= 0 && Y < f.K) atomicAdd(gtm + ((size_t)Y * Wt + X) * CT + c0 + c, v);
            }
        }
    }
}

// ---------------------------------------------------------------- MFMA tile scatter: one wave per item, the tile gradient in accumulators
// The scatter of one item is a small GEMM: G[texel][channel] = sum over the item's samples of W[texel][sample] * og[sample][channel], where
// W holds the bilinear weight of the sample at the texel (four non-zeros per column).  With 4x4-texel tiles the tile and its apron are 25
// texels - two 16-row tiles of v_mfma_f32_16x16x4_f32 - and the matching 2x5 strip of the paired time plane is a third; 24 (48) channels
// are 2 (3) column tiles, so a step of 4 samples is 6 (9) MFMAs of 32 cycles against ~200 VALU cycles per sample and 24 channels of the
// LDS read-add-write kernel above.  The products are exact fp32 (same e*s weights, fp32 accumulation); only the order of the sum differs,
// as it does between two runs of the LDS kernel.  Lane s prepares sample s once: the five x- and five y-weights of the tile columns / rows
// (zeros except at the tap pair; the zero-padding masks folded in), the same for the time strip, and the sample index, as a 19-word record
// in LDS; in the loop lane (k = lane / 16, row = lane % 16) reads the two factors of W[row][4 step + k] for each row tile (conflict-free:
// odd record stride) and the og row of its sample straight from global memory (16 lanes = 64 contiguous bytes), two steps in flight.
// No workgroup barrier: the four waves of a workgroup own separate items, records and accumulators and flush with global atomics from
// registers (16 lanes = 16 consecutive channels of one texel).
#ifndef MS_WAVES
#define MS_WAVES 4      // (2: same; 8: +4 us per call, 16: +10 us - the workgroup waits for its slowest item at the barrier)
#endif
// timing experiments (tools/r03_scatter_bisect.sh): MSX_NOLOOP=1 skips the MFMA loop, MSX_NOFLUSH=1/2/3 skips all / the time / the space atomics
#ifndef MSX_NOLOOP
#define MSX_NOLOOP 0
#endif
#ifndef MSX_NOFLUSH
#define MSX_NOFLUSH 0
#endif
#define MS_W (SCATTER_T_MFMA + 1)
#define MS_REC 19              // wx[5] wy[5] wtx[5] wty[2] index zero
typedef float ms_f4 __attribute__((ext_vector_type(4)));

template <int CT>
__global__ __launch_bounds__(64 * MS_WAVES) void k_tile_scatter_mfma(TileScatterArgs a) {
    constexpr int NCT = (CT + 15) / 16;
    constexpr int T = SCATTER_T_MFMA;
    static_assert(10 * 16 * NCT <= 64 * MS_REC, "a wave's time strip is parked in its record area");
    __shared__ float rec_all[MS_WAVES][64 * MS_REC];
    __shared__ int col_of[MS_WAVES];                       // strip id (plane, tile column / row) of each wave's item, -1: none
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int itx = blockIdx.x * MS_WAVES + wv;
    const int nitems = *a.nitems;
    if ((int)blockIdx.x * MS_WAVES >= nitems) return;      // (whole workgroup)
    const nvfi_field_desc& f = a.f;
    float* rec = rec_all[wv];
    const bool live = itx < nitems;
    const int4 item = live ? a.items[itx] : make_int4(0, 0, 0, 0);
    const int p = item.x >= a.geo.boff[2] ? 2 : (item.x >= a.geo.boff[1] ? 1 : 0);
    const int tile = item.x - a.geo.boff[p];
    const bool t_on_u = p != 1;                                // the paired time plane runs along the u axis of plane p, else along v
    int tx, ty;
    if (t_on_u && a.geo.colmajor) { tx = tile / a.geo.nty[p]; ty = tile - tx * a.geo.nty[p]; }
    else { ty = tile / a.geo.ntx[p]; tx = tile - ty * a.geo.ntx[p]; }
    const int ox = tx * T, oy = ty * T;
    const int ia = p == 2 ? 1 : 0, ib = p == 0 ? 1 : 2;       // matModeSpace axes of plane p
    const int W = f.G[ia], H = f.G[ib];
    const int tp = p == 0 ? 5 : (p == 1 ? 3 : 4);              // paired time plane
    const int Wt = t_on_u ? W : H, ot = t_on_u ? ox : oy;
    const int y0t = SCHED_Y0(a);
    float* gsp = CT == 24 ? a.g.dps[p] : a.g.aps[p];
    float* gtm = CT == 24 ? a.g.dpt[tp - 3] : a.g.apt[tp - 3];
    const int n = lane & 15, k = lane >> 4;
    // record words of this lane's A rows: space texels n and 16 + n (row-major 5x5), time texel n (row-major 2x5); word 18 is zero
    const int ax0 = n % MS_W, ay0 = 5 + n / MS_W;
    const int ax1 = 16 + n < 25 ? (16 + n) % MS_W : 18, ay1 = 16 + n < 25 ? 5 + (16 + n) / MS_W : 18;
    const int atx = n < 10 ? 10 + n % MS_W : 18, aty = n < 10 ? 15 + n / MS_W : 18;
    ms_f4 acc[3][NCT];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[r][c] = ms_f4{0.f, 0.f, 0.f, 0.f};
    const float* ogp = a.og + (size_t)p * CT + n;
    const int dtp = (tp - p) * CT;
#pragma unroll 1
    for (int b0 = 0; b0 < item.z; b0 += 64) {
        const int nb = min(64, item.z - b0);
        {   // lane s prepares sample s of the batch
            float wxa = 0.f, wxb = 0.f, wya = 0.f, wyb = 0.f, txa = 0.f, txb = 0.f, tya = 0.f, tyb = 0.f;
            int sx = -8, sy = -8, st = -8, il = 0;
            if (lane < nb) {
                const float4 r = a.sorted[item.y + b0 + lane];
                il = __float_as_int(r.x);
                Bl b, bt;
                int x0, y0, xt, yt;
                bl_setup_xy(r.y, r.z, W, H, b, x0, y0);
                bl_setup_xy(t_on_u ? r.y : r.z, SCHED_TN(a), Wt, f.K, bt, xt, yt);
                wxa = (x0 >= 0 && x0 < W) ? b.e : 0.f; wxb = (x0 + 1 >= 0 && x0 + 1 < W) ? b.w : 0.f;
                wya = (y0 >= 0 && y0 < H) ? b.s : 0.f; wyb = (y0 + 1 >= 0 && y0 + 1 < H) ? b.n : 0.f;
                txa = (xt >= 0 && xt < Wt) ? bt.e : 0.f; txb = (xt + 1 >= 0 && xt + 1 < Wt) ? bt.w : 0.f;
                tya = (yt >= 0 && yt < f.K) ? bt.s : 0.f; tyb = (yt + 1 >= 0 && yt + 1 < f.K) ? bt.n : 0.f;
                // (yt is the launch's time row y0t whenever a time weight is non-zero: same floor, same clamp)
                sx = x0 - ox; sy = y0 - oy; st = xt - ot;
            }
            float* q = rec + lane * MS_REC;
#pragma unroll
            for (int j = 0; j < MS_W; ++j) {
                q[j] = j == sx ? wxa : (j == sx + 1 ? wxb : 0.f);
                q[5 + j] = j == sy ? wya : (j == sy + 1 ? wyb : 0.f);
                q[10 + j] = j == st ? txa : (j == st + 1 ? txb : 0.f);
            }
            q[15] = tya; q[16] = tyb; q[17] = __int_as_float(il); q[18] = 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nst = ((nb + 15) >> 4) << 2;         // steps of 4 samples, a multiple of 4 (records past the batch hold zero weights)
        auto load_b = [&](int s, float (&bp)[NCT], float (&bt)[NCT]) {
            const int sc = s < nst ? s : nst - 1;
            const int i = __float_as_int(rec[(4 * sc + k) * MS_REC + 17]);
            const float* q = ogp + (size_t)i * (6 * CT);
#pragma unroll
            for (int c = 0; c < NCT; ++c) { bp[c] = q[16 * c]; bt[c] = q[dtp + 16 * c]; }
        };
        auto step = [&](int s, const float (&bp)[NCT], const float (&bt)[NCT]) {
            const float* r = rec + (4 * s + k) * MS_REC;
            const float a0 = r[ax0] * r[ay0], a1 = r[ax1] * r[ay1], at = r[atx] * r[aty];
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bp[c], acc[0][c], 0, 0, 0);
                acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bp[c], acc[1][c], 0, 0, 0);
                acc[2][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, bt[c], acc[2][c], 0, 0, 0);
            }
        };
        // four operand sets rotate: the og rows of step s + 3 are requested before the MFMAs of step s are issued
        float bp0[NCT], bt0[NCT], bp1[NCT], bt1[NCT], bp2[NCT], bt2[NCT], bp3[NCT], bt3[NCT];
        load_b(0, bp0, bt0);
        load_b(1, bp1, bt1);
        load_b(2, bp2, bt2);
#pragma unroll 1
        for (int s = 0; s < (MSX_NOLOOP ? 0 : nst); s += 4) {
            load_b(s + 3, bp3, bt3);
            __builtin_amdgcn_sched_barrier(0);
            step(s, bp0, bt0);
            __builtin_amdgcn_sched_barrier(0);
            load_b(s + 4, bp0, bt0);
            __builtin_amdgcn_sched_barrier(0);
            step(s + 1, bp1, bt1);
            __builtin_amdgcn_sched_barrier(0);
            load_b(s + 5, bp1, bt1);
            __builtin_amdgcn_sched_barrier(0);
            step(s + 2, bp2, bt2);
            __builtin_amdgcn_sched_barrier(0);
            load_b(s + 6, bp2, bt2);
            __builtin_amdgcn_sched_barrier(0);
            step(s + 3, bp3, bt3);
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // (the next batch overwrites the records)
    }
    // park the time strip (10 texels x 16 NCT channels) in the wave's own record area: the workgroup sums the strips of waves that
    // share a tile column before they go to memory - consecutive items do (column-major bins), and the 2 x Wt texels of a time plane
    // would otherwise take one atomic per item, all on the same few cache lines
    if (live) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = 4 * k + v;
            if (r < 10) {
#pragma unroll
                for (int c = 0; c < NCT; ++c) rec[r * (16 * NCT) + 16 * c + n] = acc[2][c][v];
            }
        }
    }
    if (lane == 0) col_of[wv] = live && item.z > 0 ? (p << 16) | (t_on_u ? tx : ty) : -1;
    // space tile: straight from the accumulators; output register v of lane (k, n) is row 4 k + v, channel 16 c + n of each 16 x 16 tile
    if (live && MSX_NOFLUSH != 1 && MSX_NOFLUSH != 3 && gsp) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int r = 16 * rt + 4 * k + v;
                const int X = ox + r % MS_W, Y = oy + r / MS_W;
                if (r < 25 && X < W && Y < H) {
                    float* g = gsp + ((size_t)Y * W + X) * CT + n;
#pragma unroll
                    for (int c = 0; c < NCT; ++c) {
                        const float val = acc[rt][c][v];
                        if (16 * c + n < CT && val != 0.f) atomicAdd(g + 16 * c, val);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (MSX_NOFLUSH == 1 || MSX_NOFLUSH == 2) return;
    // thread (texel r, channel c) walks the waves in order and sends one sum per run of equal strip ids
    for (int tid = threadIdx.x; tid < 10 * 16 * NCT; tid += 64 * MS_WAVES) {
        const int c = tid % (16 * NCT), r = tid / (16 * NCT);
        float sum = 0.f;
        int cur = -1;
        auto flush = [&]() {
            if (cur < 0 || sum == 0.f || c >= CT) return;
            const int pp = cur >> 16, o = (cur & 0xffff) * T;
            const int tpp = pp == 0 ? 5 : (pp == 1 ? 3 : 4);
            const int Wtt = pp == 2 ? f.G[1] : (pp == 1 ? f.G[2] : f.G[0]);       // plane 0: x, plane 1 (x,z): z, plane 2 (y,z): y
            float* g = CT == 24 ? a.g.dpt[tpp - 3] : a.g.apt[tpp - 3];
            const int X = o + r % MS_W, Y = y0t + r / MS_W;
            if (g && X < Wtt && Y >= 0 && Y < f.K) atomicAdd(g + ((size_t)Y * Wtt + X) * CT + c, sum);
        };
        for (int w = 0; w < MS_WAVES; ++w) {
            const int id = col_of[w];
            if (id != cur) { flush(); cur = id; sum = 0.f; }
            if (id >= 0) sum += rec_all[w][tid];
        }
        flush();
    }
    (void)gtm; (void)Wt;
}

// ---------------------------------------------------------------- host
// NVFI_SCATTER=mfma (default) | lds: the MFMA kernel on 4x4-texel tiles, or the LDS read-add-write kernel on 8x8-texel tiles
static bool scatter_mfma() { static int u = -1; if (u < 0) { const char* e = getenv("NVFI_SCATTER"); u = (e && !strcmp(e, "lds")) ? 0 : 1; } return u != 0; }
static int tile_geom_T(const nvfi_field_desc* f, TileGeom* g, int T, int chunk) {
    const int ia[3] = {0, 0, 1}, ib[3] = {1, 2, 2};
    int off = 0;
    for (int p = 0; p < 3; ++p) {
        g->ntx[p] = (f->G[ia[p]] + T - 1) / T;
        g->nty[p] = (f->G[ib[p]] + T - 1) / T;
        g->boff[p] = off;
        off += g->ntx[p] * g->nty[p];
    }
    g->nbins = off; g->T = T; g->chunk = chunk; g->chunk_shift = 31 - __builtin_clz(chunk); g->colmajor = T == SCATTER_T_MFMA ? 1 : 0;
    for (int c = 0; c < 3; ++c) g->G[c] = f->G[c];
    return off;
}
int tile_geom(const nvfi_field_desc* f, TileGeom* g) {
    if (scatter_mfma() && tile_geom_T(f, g, SCATTER_T_MFMA, SCATTER_CHUNK_MFMA) <= SCATTER_MAX_BINS_MFMA) return 0;
    return tile_geom_T(f, g, SCATTER_T, SCATTER_CHUNK) <= SCATTER_MAX_BINS ? 0 : 1;
}

static_assert((SCATTER_CHUNK & (SCATTER_CHUNK - 1)) == 0 && (SCATTER_CHUNK_MFMA & (SCATTER_CHUNK_MFMA - 1)) == 0, "chunks are powers of two");
int64_t tile_items_cap(const TileGeom& g, int64_t N) { return 3 * ((N + g.chunk - 1) / g.chunk) + g.nbins; }

void plan_tile_scatter(Bump& B, const nvfi_field_desc* f, int64_t N, TileWork* w) {
    tile_geom(f, &w->g);
    if (!w->hist) w->hist = B.take<int>(w->g.nbins + 64);      // (render.hip takes the histograms itself, next to its counters)
    w->cursor = B.take<int>(w->g.nbins); w->nitems = B.take<int>(4);   // hist[nbins]: ticket of k_tile_hist's last-workgroup scan
    w->start = B.take<int>(w->g.nbins + 1); w->istart = B.take<int>(w->g.nbins + 1);
    w->items = B.take<int4>(tile_items_cap(w->g, N));
    w->sorted = B.take<float4>(3 * N);
    w->og = B.take<float>(N * 6 * 48);
    w->cap_items = tile_items_cap(w->g, N);
}

// histogram storage (and its ticket, hist[nbins]) must be zero before the first k_tile_hist of a workspace - the forward's fill of the
// counters covers it (render.hip: plan_render) - and the scan re-zeroes both after every use

// k_tile_hist's scan uses up to 70 KB of dynamic LDS (8192 bins): raised once, from the entry points (never inside a stream capture)
int ensure_scatter_attrs() {
    static bool done = false;
    if (done) return 0;
    HIPCK(hipFuncSetAttribute((const void*)k_tile_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    done = true;
    return 0;
}

int launch_og(const nvfi_field_desc* f, const OgArgs& oa, int C, bool coord, int64_t N, hipStream_t st) {
    const int spw = C == 24 ? 32 : 16;
    const unsigned blocks = (unsigned)((N + spw - 1) / spw);
    if (C == 24) {
        if (coord) hipLaunchKernelGGL((k_og<24, true>), dim3(blocks), dim3(256), 0, st, oa);
        else hipLaunchKernelGGL((k_og<24, false>), dim3(blocks), dim3(256), 0, st, oa);
    } else {
        if (coord) hipLaunchKernelGGL((k_og<48, true>), dim3(blocks), dim3(256), 0, st, oa);
        else hipLaunchKernelGGL((k_og<48, false>), dim3(blocks), dim3(256), 0, st, oa);
    }
    LAUNCHCK();
    return 0;
}

// counting sort of one or two compact lists by plane tile (w[j], count[j], list[j]; same geometry, same positions xw): two launches
int launch_tile_sort(const TileWork* const* w, const int* const* count, const int* const* list, int njobs, const float4* xw, int64_t N, hipStream_t st) {
    TileSortArgs2 s2; memset(&s2, 0, sizeof(s2));
    for (int j = 0; j < njobs; ++j) {
        TileSortArgs& sa = s2.j[j];
        const TileWork& t = *w[j];
        sa.g = t.g; sa.count = count[j]; sa.list = list[j]; sa.xw = xw; sa.hist = t.hist; sa.cursor = t.cursor; sa.items = t.items; sa.nitems = t.nitems; sa.sorted = t.sorted; sa.start = t.start; sa.istart = t.istart;
    }
    const int nb = w[0]->g.nbins;
    unsigned hb = (unsigned)((N + 511) / 512); if (hb > 512) hb = 512;
    const size_t hist_lds = sizeof(int) * 2 * (size_t)(nb + (nb >> 4) + 1);      // the histogram, then the scan's two transposed rows
    hipLaunchKernelGGL(k_tile_hist, dim3(hb, njobs), dim3(512), hist_lds, st, s2);
    hipLaunchKernelGGL(k_tile_fill, dim3((unsigned)((N + 511) / 512), njobs), dim3(512), sizeof(int) * 2 * nb, st, s2);
    LAUNCHCK();
    return 0;
}

// sorted: the list was already sorted into w by launch_tile_sort (round 5: both lists of a backward in one pair of launches)
int launch_tile_scatter(const nvfi_field_desc* f, const TileWork& w, const int* count, const int* list, const float4* xw, float tn,
                        const nvfi_grads& g, int C, int64_t N, hipStream_t st, const float* sched, bool sorted) {
    if (!sorted) {
        const TileWork* wp[1] = {&w}; const int* cp[1] = {count}; const int* lp[1] = {list};
        if (launch_tile_sort(wp, cp, lp, 1, xw, N, st)) return 1;
    }
    TileScatterArgs ta; memset(&ta, 0, sizeof(ta));
    ta.f = *f; ta.geo = w.g; ta.items = w.items; ta.nitems = w.nitems; ta.sorted = w.sorted; ta.list = list; ta.xw = xw; ta.og = w.og; ta.tn = tn;
    ta.g = g; ta.sched = sched;
    const float y = (tn + 1.f) * ((float)(f->K - 1) / 2.f);
    float yf = floorf(y);
    yf = fminf(fmaxf(yf, -4.f), (float)f->K + 2.f);
    ta.y0 = (int)yf;
    if (w.g.T == SCATTER_T_MFMA) {
        const unsigned wgs = (unsigned)((w.cap_items + MS_WAVES - 1) / MS_WAVES);
        if (C == 24) hipLaunchKernelGGL(k_tile_scatter_mfma<24>, dim3(wgs), dim3(64 * MS_WAVES), 0, st, ta);
        else hipLaunchKernelGGL(k_tile_scatter_mfma<48>, dim3(wgs), dim3(64 * MS_WAVES), 0, st, ta);
    } else if (C == 24) hipLaunchKernelGGL(k_tile_scatter<24>, dim3((unsigned)w.cap_items, 1), dim3(64 * TS_WAVES), 0, st, ta);
    else hipLaunchKernelGGL(k_tile_scatter<48>, dim3((unsigned)w.cap_items, 2), dim3(64 * TS_WAVES), 0, st, ta);
    LAUNCHCK();
    return 0;
}
