// render.hip - ray marching through the keyframe K-plane field: sampling, density gather, volume
// weights, appearance MLP (fp32 MFMA), composite; and the hand-written backward of all of it.
//
// Reference semantics: models/tensorf_base.py:290-314 (sample_ray), models/tensorf_keyframe.py:233-325
// (feature lookups, softplus), models/tensorf_model_utils.py:176-197 (PE, raw2alpha),
// models/tensorf_base.py:67-98 (MLPRender_PE), models/tensorf_keyframe.py:641-755 (render_pts).
//
// Work decomposition (MI355X-first): the reference's boolean-mask gather/scatter with a host sync
// per mask becomes on-device compaction (per-ray counts -> one-block scan -> ordered fill), so no
// kernel launch depends on a host-visible count.  Gathers read channel-last planes with 16-byte
// loads; per-ray scans/reductions are wave-level; the MLP contractions run on the MFMA engine.
#include "common.h"
#include "render.h"
#include "fuse.h"
#include "scatter.h"
#include "pde.h"
#include "frags.h"
#include "x6.h"
#include <stdlib.h>
#include <mutex>

// ================================================================ sampling + compaction
__global__ void k_any_inside(nvfi_field_desc f, int64_t R, const float* __restrict__ o, int* flag) {
    // tensorf_base.py:294: ((aabb0 <= o) & (o <= aabb1)).any() over every coordinate of every ray
    bool hit = false;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < R * 3; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % 3);
        float v = o[i];
        if (f.aabb[c] <= v && v <= f.aabb[3 + c]) hit = true;
    }
    if (__any(hit) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__device__ __forceinline__ float ray_tmin(const nvfi_field_desc& f, bool inside, const float* o, const float* d) {
    if (inside) return f.near_;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float vec = d[c] == 0.f ? 1e-6f : d[c];
        float ra = (f.aabb[3 + c] - o[c]) / vec;
        float rb = (f.aabb[c] - o[c]) / vec;
        m = fmaxf(m, fminf(ra, rb));
    }
    return fminf(fmaxf(m, f.near_), f.far_);
}

__device__ __forceinline__ float alpha_lookup(const nvfi_field_desc& f, float x, float y, float z) {
    // AlphaGridMask.sample_alpha: trilinear, align_corners=True, zeros padding (tensorf_model_utils.py:433-439)
    const int W = f.am_dims[0], H = f.am_dims[1], D = f.am_dims[2];
    float ix = (x + 1.f) * ((float)(W - 1) / 2.f), iy = (y + 1.f) * ((float)(H - 1) / 2.f), iz = (z + 1.f) * ((float)(D - 1) / 2.f);
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    float wx = ix - fx, wy = iy - fy, wz = iz - fz;
    fx = fminf(fmaxf(fx, -4.f), W + 2.f); fy = fminf(fmaxf(fy, -4.f), H + 2.f); fz = fminf(fmaxf(fz, -4.f), D + 2.f);
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    float s = 0.f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
                if (xi < 0 || xi >= W || yi < 0 || yi >= H || zi < 0 || zi >= D) continue;
                float w = (dx ? wx : 1.f - wx) * (dy ? wy : 1.f - wy) * (dz ? wz : 1.f - wz);
                s += f.amask[((size_t)zi * H + yi) * W + xi] * w;
            }
    return s;
}

// one wave per ray
__global__ __launch_bounds__(256) void k_sample(SampleArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const nvfi_field_desc& f = a.f;
    const int S = f.n_samples;
    float o[3] = {a.o[3 * r], a.o[3 * r + 1], a.o[3 * r + 2]};
    float d[3] = {a.d[3 * r], a.d[3 * r + 1], a.d[3 * r + 2]};
    const float tmin = ray_tmin(f, *a.inside != 0, o, d);
    const float u = (a.train && a.u) ? a.u[r] : 0.f;
    int cnt = 0, cntr = 0;
    for (int j0 = 0; j0 < S; j0 += 64) {
        const int j = j0 + lane;
        bool ok = false, mv = false;
        if (j < S) {
            float rng = (float)j + u;
            float step = f.step_size * rng;
            float z = tmin + step;
            float p[3], xn[3];
            ok = true;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                p[c] = o[c] + d[c] * z;
                if (f.aabb[c] > p[c] || p[c] > f.aabb[3 + c]) ok = false;
                xn[c] = norm_coord(f, c, p[c]);
            }
            if (ok && f.has_amask && !a.train) ok = alpha_lookup(f, xn[0], xn[1], xn[2]) > 0.f;
            const int64_t n = r * S + j;
            a.xw[n] = make_float4(xn[0], xn[1], xn[2], z);
            a.xpre[n] = XPRE_INVALID;
            a.valid[n] = ok ? 1 : 0;
            // a sample outside the velocity gate never moves (v = 0 there, velocity_field.py:28-33,46-51): the warp skips it
            mv = ok && !gated_out(f, xn[0], xn[1], xn[2]);
            if (a.rflag) a.rflag[n] = mv ? 1 : 0;
        }
        cnt += __popcll(__ballot(ok));
        cntr += __popcll(__ballot(mv));
    }
    if (lane == 0) { a.cnt[r] = cnt; if (a.cnt_r) a.cnt_r[r] = cntr; }
}

// ordered fill of the compact list: list[off[r] + rank] = dense index.  The exclusive scan of the per-group counts rides in the same
// launch: a workgroup (4 groups) sums the counts of every group before its own - n <= a few thousand ints out of L2 - instead of reading
// the result of a separate one-workgroup scan kernel (one launch less per compaction: 3 per render, 2 per PDE call); it also writes
// off[] for its groups (k_final_fwd / k_weights_bwd read the per-ray offsets of the masked list), the last one off[n] and *total_out.
__global__ __launch_bounds__(256) void k_fill(int64_t R, int S, const uint8_t* __restrict__ flags, const int* __restrict__ cnt, int* __restrict__ off,
                                              int* __restrict__ list, int* total_out) {
    __shared__ int part[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * 4;
    int s = 0;
    for (int64_t i = threadIdx.x; i < r0; i += 256) s += cnt[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) part[w] = s;
    __syncthreads();
    int base = (part[0] + part[1]) + (part[2] + part[3]);
    const int64_t r = r0 + w;
    for (int k = 0; k < w; ++k) base += (r0 + k < R) ? cnt[r0 + k] : 0;
    if (r >= R) return;
    if (lane == 0) {
        off[r] = base;
        if (r == R - 1) { const int tot = base + cnt[r]; off[R] = tot; *total_out = tot; }
    }
    for (int j0 = 0; j0 < S; j0 += 64) {
        const int j = j0 + lane;
        bool ok = j < S && flags[r * S + j];
        unsigned long long b = __ballot(ok);
        if (ok) list[base + __popcll(b & ((1ull << lane) - 1ull))] = (int)(r * S + j);
        base += __popcll(b);
    }
}

// ---------------------------------------------------------------- round 5: the same lists without the second (and third) launch
// NVFI_FUSED_LAUNCH (default 1): the producers of the flags place the list entries themselves (look-back, common.h); 0 keeps the
// count + k_fill launches of rounds 1-4.  Same flags, same order: the lists are identical entry for entry.
// the training warp's kernel family and what it means for the stash layout (forward and backward must agree, so both ask here):
// NVFI_RK2_X6 (default 1): the x6 kernels; NVFI_RK2_FUSE (default 1): the adjoint + hidden-layer weight gradients in one persistent kernel - then the z
// rows of layers 0..3 have ONE reader and travel as x4 stash blocks (a quarter of the stash instructions on both sides; NVFI_RK2_X4=0: row-major)
static bool warp_x6_on() { static int v = -1; if (v < 0) { const char* e = getenv("NVFI_RK2_X6"); v = e ? atoi(e) : 1; } return v != 0; }
static bool rk2_fuse_on() { static int v = -1; if (v < 0) { const char* e = getenv("NVFI_RK2_FUSE"); v = e ? atoi(e) : 1; } return v != 0; }
static bool warp_stash_x4(const nvfi_field_desc* f) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("NVFI_RK2_X4"); v = e ? atoi(e) : 1; }
    return v != 0 && warp_x6_on() && rk2_fuse_on() && !(f->vel_fp16 & 4);
}
bool fused_launch() { static int u = -1; if (u < 0) { const char* e = getenv("NVFI_FUSED_LAUNCH"); u = e ? atoi(e) : 1; } return u != 0; }

// k_sample + the two k_fill launches behind it
__global__ __launch_bounds__(256) void k_sample_fill(SampleArgs a) {
    __shared__ int cv[4], cr[4];
    __shared__ unsigned long long excl_sh;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + w;
    const bool ron = r < a.R;
    const nvfi_field_desc& f = a.f;
    const int S = f.n_samples;
    int cnt = 0, cntr = 0;
    if (ron) {
        float o[3] = {a.o[3 * r], a.o[3 * r + 1], a.o[3 * r + 2]};
        float d[3] = {a.d[3 * r], a.d[3 * r + 1], a.d[3 * r + 2]};
        const float tmin = ray_tmin(f, *a.inside != 0, o, d);
        const float u = (a.train && a.u) ? a.u[r] : 0.f;
        for (int j0 = 0; j0 < S; j0 += 64) {
            const int j = j0 + lane;
            bool ok = false, mv = false;
            if (j < S) {
                float rng = (float)j + u;
                float step = f.step_size * rng;
                float z = tmin + step;
                float p[3], xn[3];
                ok = true;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    p[c] = o[c] + d[c] * z;
                    if (f.aabb[c] > p[c] || p[c] > f.aabb[3 + c]) ok = false;
                    xn[c] = norm_coord(f, c, p[c]);
                }
                if (ok && f.has_amask && !a.train) ok = alpha_lookup(f, xn[0], xn[1], xn[2]) > 0.f;
                const int64_t n = r * S + j;
                a.xw[n] = make_float4(xn[0], xn[1], xn[2], z);
                a.xpre[n] = XPRE_INVALID;
                a.valid[n] = ok ? 1 : 0;
                mv = ok && !gated_out(f, xn[0], xn[1], xn[2]);
                if (a.rflag) a.rflag[n] = mv ? 1 : 0;
            }
            cnt += __popcll(__ballot(ok));
            cntr += __popcll(__ballot(mv));
        }
    }
    if (lane == 0) { cv[w] = cnt; cr[w] = cntr; }
    __syncthreads();
    if (w == 0) {
        const unsigned long long agg = ((unsigned long long)((cv[0] + cv[1]) + (cv[2] + cv[3])) << 31) | (unsigned long long)((cr[0] + cr[1]) + (cr[2] + cr[3]));
        const unsigned long long e = lb_exclusive(a.lb, (int)blockIdx.x, agg);
        if (lane == 0) {
            excl_sh = e;
            if (blockIdx.x == gridDim.x - 1) {
                const unsigned long long tot = e + agg;
                *a.total_v = (int)(tot >> 31);
                if (a.rflag) *a.total_r = (int)(tot & 0x7fffffffull);
            }
        }
    }
    __syncthreads();
    if (!ron) return;
    int base_v = (int)(excl_sh >> 31), base_r = (int)(excl_sh & 0x7fffffffull);
    for (int k = 0; k < w; ++k) { base_v += cv[k]; base_r += cr[k]; }
    // (the flags were written by this very lane above)
    for (int j0 = 0; j0 < S; j0 += 64) {
        const int j = j0 + lane;
        const bool ok = j < S && a.valid[r * S + j];
        const unsigned long long b = __ballot(ok);
        if (ok) a.vlist[base_v + __popcll(b & ((1ull << lane) - 1ull))] = (int)(r * S + j);
        base_v += __popcll(b);
        if (a.rflag) {
            const bool mv = j < S && a.rflag[r * S + j];
            const unsigned long long bm = __ballot(mv);
            if (mv) a.rlist[base_r + __popcll(bm & ((1ull << lane) - 1ull))] = (int)(r * S + j);
            base_r += __popcll(bm);
        }
    }
}

// scan + ordered fill for n_groups groups of 64 flags (used by the PDE prefilter)
int launch_scan_fill(const int* cnt, int* off, int64_t ngroups, int* total, const uint8_t* flags, int* list, hipStream_t st) {
    if (ngroups <= 0) return launch_zero(total, sizeof(int), st);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, st, ngroups, 64, flags, cnt, off, list, total);
    LAUNCHCK();
    return 0;
}

// ================================================================ density
// (forward: k_density_q in scatter.hip - lanes = sample x channel quad)

// backward: gxpre -> plane grads (atomics) + coordinate grads
__global__ __launch_bounds__(256) void k_density_bwd(DensityArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int count = *a.count;
    if (i >= count) return;
    const nvfi_field_desc& f = a.f;
    const int n = a.list[i];
    const float4 q = a.xw[n];
    const float gf = a.gxpre[n];
    Bl b[6];
    plane_setups(f, q.x, q.y, q.z, SCHED_TN(a), b);
    float gx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gy[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* pl[6] = {f.dps[0], f.dps[1], f.dps[2], f.dpt[0], f.dpt[1], f.dpt[2]};
    float* gp[6] = {a.g.dps[0], a.g.dps[1], a.g.dps[2], a.g.dpt[0], a.g.dpt[1], a.g.dpt[2]};
    const int nq = f.Cd >> 2;
    for (int q4 = 0; q4 < nq; ++q4) {
        float4 v[6];
#pragma unroll
        for (int p = 0; p < 6; ++p) v[p] = bl_sample4(pl[p], f.Cd, b[p], q4);
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            float4 o = make_float4(gf, gf, gf, gf);
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k != p) { o.x *= v[k].x; o.y *= v[k].y; o.z *= v[k].z; o.w *= v[k].w; }
            bl_backward4(pl[p], gp[p], f.Cd, b[p], q4, o, gx[p], gy[p]);
        }
    }
    float g3[3] = {0.f, 0.f, 0.f};
    {
        float mx, my;
        plane_mults(f, 0, mx, my); g3[0] += gx[0] * mx; g3[1] += gy[0] * my;
        plane_mults(f, 1, mx, my); g3[0] += gx[1] * mx; g3[2] += gy[1] * my;
        plane_mults(f, 2, mx, my); g3[1] += gx[2] * mx; g3[2] += gy[2] * my;
        plane_mults(f, 3, mx, my); g3[2] += gx[3] * mx;
        plane_mults(f, 4, mx, my); g3[1] += gx[4] * mx;
        plane_mults(f, 5, mx, my); g3[0] += gx[5] * mx;
    }
    if (a.gxk) {
        float4 ga = a.mflag[n] ? a.gxw[n] : zero4();   // appearance-branch part (masked samples only)
        a.gxk[n] = make_float4(ga.x + g3[0], ga.y + g3[1], ga.z + g3[2], 0.f);   // dense (per sample): the RK2 adjoint walks its own list
    }
}

// ================================================================ volume weights (raw2alpha) + composites
// one wave per ray; 64-sample segments with a carried transmittance
__global__ __launch_bounds__(256) void k_weights_fwd(WeightArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int S = a.S;
    float carry = 1.f, accs = 0.f, dep = 0.f;
    int cnt = 0;
    for (int j0 = 0; j0 < S; j0 += 64) {
        const int j = j0 + lane;
        const bool in = j < S;
        const int64_t n = r * S + j;
        float sig = 0.f, dist = 0.f, z = 0.f;
        if (in) {
            sig = softplus_f(a.xpre[n]);
            z = a.xw[n].w;
            if (j + 1 < S) dist = (a.xw[n + 1].w - z) * a.distance_scale;
        }
        const float al = 1.f - expf(-sig * dist);
        const float fct = 1.f - al + 1e-10f;
        float p = fct;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(p, o); if (lane >= o) p *= t; }
        float ex = __shfl_up(p, 1);
        if (lane == 0) ex = 1.f;
        const float T = carry * ex;
        const float w = al * T;
        carry = carry * __shfl(p, 63);
        const bool m = in && w > a.weight_thres;
        if (in) { a.weight[n] = w; a.mflag[n] = m ? 1 : 0; accs += w; dep += w * z; }
        cnt += __popcll(__ballot(m));
    }
    accs = wave_sum(accs); dep = wave_sum(dep);
    if (lane == 0) {
        a.acc[r] = accs;
        a.depth[r] = dep + (1.f - accs) * a.far_;
        a.cnt_m[r] = cnt;
    }
}

// the call's counters for the caller (device-side totals -> int64[8]); by k_counters, or by workgroup 0 of k_final_fwd (round 5)
__device__ __forceinline__ void counters_body(const int* c, int nsteps, int64_t* out, const float* sched) {
    out[0] = c[0];
    out[1] = nsteps > 0 ? c[3] : 0;
    out[2] = c[1];
    out[3] = (int64_t)(nsteps > 0 ? c[3] : 0) * 2 * nsteps;
    out[4] = out[5] = out[6] = 0;
    out[7] = sched ? __float_as_int(sched[3]) : 0;      // 1: the device-side time did not fit the planned RK2 step count (the planned time was rendered)
}

// k_weights_fwd + the k_fill launch behind it: the ordered list of appearance-masked samples (weight > rayMarch_weight_thres) and the per-ray
// offsets into it (k_final_fwd / k_weights_bwd read off_m) from the same launch
__global__ __launch_bounds__(256) void k_weights_fill(WeightArgs a) {
    __shared__ int cm[4];
    __shared__ unsigned long long excl_sh;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + w;
    const bool ron = r < a.R;
    const int S = a.S;
    int cnt = 0;
    if (ron) {
        float carry = 1.f, accs = 0.f, dep = 0.f;
        for (int j0 = 0; j0 < S; j0 += 64) {
            const int j = j0 + lane;
            const bool in = j < S;
            const int64_t n = r * S + j;
            float sig = 0.f, dist = 0.f, z = 0.f;
            if (in) {
                sig = softplus_f(a.xpre[n]);
                z = a.xw[n].w;
                if (j + 1 < S) dist = (a.xw[n + 1].w - z) * a.distance_scale;
            }
            const float al = 1.f - expf(-sig * dist);
            const float fct = 1.f - al + 1e-10f;
            float p = fct;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(p, o); if (lane >= o) p *= t; }
            float ex = __shfl_up(p, 1);
            if (lane == 0) ex = 1.f;
            const float T = carry * ex;
            const float wgt = al * T;
            carry = carry * __shfl(p, 63);
            const bool m = in && wgt > a.weight_thres;
            if (in) { a.weight[n] = wgt; a.mflag[n] = m ? 1 : 0; accs += wgt; dep += wgt * z; }
            cnt += __popcll(__ballot(m));
        }
        accs = wave_sum(accs); dep = wave_sum(dep);
        if (lane == 0) {
            a.acc[r] = accs;
            a.depth[r] = dep + (1.f - accs) * a.far_;
        }
    }
    if (lane == 0) cm[w] = cnt;
    __syncthreads();
    if (w == 0) {
        const unsigned long long agg = (unsigned long long)((cm[0] + cm[1]) + (cm[2] + cm[3]));
        const unsigned long long e = lb_exclusive(a.lb, (int)blockIdx.x, agg);
        if (lane == 0) {
            excl_sh = e;
            if (blockIdx.x == gridDim.x - 1) { const int tot = (int)(e + agg); a.off_m_out[a.R] = tot; *a.total_m = tot; }
        }
    }
    __syncthreads();
    if (!ron) return;
    int base = (int)excl_sh;
    for (int k = 0; k < w; ++k) base += cm[k];
    if (lane == 0) a.off_m_out[r] = base;
    for (int j0 = 0; j0 < S; j0 += 64) {
        const int j = j0 + lane;
        const bool ok = j < S && a.mflag[r * S + j];
        const unsigned long long b = __ballot(ok);
        if (ok) a.mlist[base + __popcll(b & ((1ull << lane) - 1ull))] = (int)(r * S + j);
        base += __popcll(b);
    }
}

__global__ __launch_bounds__(256) void k_final_fwd(FinalArgs a) {
    __shared__ float red[4];
    __shared__ int last;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    const bool ron = r < a.R;
    float se = 0.f;
    if (ron) {
        const int b0 = a.off_m[r], b1 = a.off_m[r + 1];
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        for (int i = b0 + lane; i < b1; i += 64) {
            const float w = a.weight[a.mlist[i]];
            const float4 c = a.rgbs[i];
            c0 += w * c.x; c1 += w * c.y; c2 += w * c.z;
        }
        c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
        if (lane == 0) {
            if (a.white_bg) { float bg = 1.f - a.acc[r]; c0 += bg; c1 += bg; c2 += bg; }
            a.rgb_pre[r] = make_float4(c0, c1, c2, 0.f);
            const float o0 = fminf(fmaxf(c0, 0.f), 1.f), o1 = fminf(fmaxf(c1, 0.f), 1.f), o2 = fminf(fmaxf(c2, 0.f), 1.f);
            a.rgb[3 * r] = o0; a.rgb[3 * r + 1] = o1; a.rgb[3 * r + 2] = o2;
            if (a.target) {
                const float inv = 1.f / (float)(3 * a.R);
                const float d0 = o0 - a.target[3 * r], d1 = o1 - a.target[3 * r + 1], d2 = o2 - a.target[3 * r + 2];
                a.g_rgb_out[3 * r] = a.loss_scale * (2.f * d0 * inv); a.g_rgb_out[3 * r + 1] = a.loss_scale * (2.f * d1 * inv); a.g_rgb_out[3 * r + 2] = a.loss_scale * (2.f * d2 * inv);
                se = (d0 * d0 + d1 * d1) + d2 * d2;
            }
        }
    }
    if (a.counters_out && blockIdx.x == 0 && threadIdx.x == 0) counters_body(a.c, a.nsteps, a.counters_out, a.sched);
    if (!a.target) return;
    if (lane == 0) red[wv] = se;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.partial + blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last = __hip_atomic_fetch_add(a.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last && wv == 0) {      // the last workgroup sums the partials in workgroup order: the value does not depend on which one that is
        float t = 0.f;
        for (int k = lane; k < (int)gridDim.x; k += 64) t += __hip_atomic_load(a.partial + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = wave_sum(t);
        if (lane == 0) { *a.loss_out = t * (1.f / (float)(3 * a.R)); *a.ticket = 0; }
    }
}

// backward of composites + raw2alpha: produces d/d(xpre) per sample
__global__ __launch_bounds__(256) void k_weights_bwd(WeightArgs a) {
    __shared__ float carries[4][17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    if (r >= a.R) return;
    const int S = a.S;
    const int nseg = (S + 63) >> 6;
    // upstream
    float gr[3] = {0.f, 0.f, 0.f};
    if (a.g_rgb) {
        const float4 pre = a.rgb_pre[r];
        const float pv[3] = {pre.x, pre.y, pre.z};
#pragma unroll
        for (int c = 0; c < 3; ++c) gr[c] = (pv[c] >= 0.f && pv[c] <= 1.f) ? a.g_rgb[3 * r + c] : 0.f;
    }
    const float gd = a.g_depth ? a.g_depth[r] : 0.f, ga = a.g_acc ? a.g_acc[r] : 0.f;
    const float bgsum = a.white_bg ? (gr[0] + gr[1] + gr[2]) : 0.f;
    // pass 1: carried transmittance at the start of each segment
    float carry = 1.f;
    for (int sg = 0; sg < nseg; ++sg) {
        const int j = sg * 64 + lane;
        const int64_t n = r * S + j;
        float sig = 0.f, dist = 0.f;
        if (j < S) {
            sig = softplus_f(a.xpre[n]);
            if (j + 1 < S) dist = (a.xw[n + 1].w - a.xw[n].w) * a.distance_scale;
        }
        const float al = 1.f - expf(-sig * dist);
        float p = 1.f - al + 1e-10f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(p, o); if (lane >= o) p *= t; }
        if (lane == 0) carries[wv][sg] = carry;
        carry = carry * __shfl(p, 63);
    }
    // pass 2: reverse segments
    float suffix = 0.f;   // sum_{i>j} gw_i w_i over later segments
    int mrank_end = a.off_m[r + 1];
    for (int sg = nseg - 1; sg >= 0; --sg) {
        const int j = sg * 64 + lane;
        const bool in = j < S;
        const int64_t n = r * S + j;
        float sig = 0.f, dist = 0.f, z = 0.f, xp = XPRE_INVALID;
        bool m = false;
        if (in) {
            xp = a.xpre[n];
            sig = softplus_f(xp);
            z = a.xw[n].w;
            if (j + 1 < S) dist = (a.xw[n + 1].w - z) * a.distance_scale;
            m = a.mflag[n] != 0;
        }
        const float al = 1.f - expf(-sig * dist);
        const float fct = 1.f - al + 1e-10f;
        float p = fct;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(p, o); if (lane >= o) p *= t; }
        float ex = __shfl_up(p, 1);
        if (lane == 0) ex = 1.f;
        const float T = carries[wv][sg] * ex;
        const float w = al * T;
        // colour of masked samples comes from the compact list (ray-ordered)
        const unsigned long long mb = __ballot(m);
        const int seg_cnt = __popcll(mb);
        float gw = -bgsum + ga + gd * (z - a.far_) + ((a.g_weight && in) ? a.g_weight[n] : 0.f);
        if (m) {
            const int mi = mrank_end - seg_cnt + __popcll(mb & ((1ull << lane) - 1ull));
            const float4 c = a.rgbs[mi];
            gw += gr[0] * c.x + gr[1] * c.y + gr[2] * c.z;
        }
        mrank_end -= seg_cnt;
        if (!in) gw = 0.f;
        // suffix sums within the segment: s_j = sum_{i>j} gw_i w_i
        float v = gw * w, inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { float t = __shfl_down(inc, o); if (lane + o < 64) inc += t; }
        const float suf = suffix + (inc - v);
        suffix = suffix + __shfl(inc, 0);
        if (in) {
            const float galpha = gw * T - suf / fct;
            const float gsig = galpha * dist * (1.f - al);
            a.gxpre[n] = gsig * (xp > 20.f ? 1.f : sigmoid_f(xp));
        }
    }
}

// ================================================================ appearance (MFMA)
// Per-lane scratch in the (idle) weight LDS region: row k of thread tid lives at scr[k*256 + tid].
// Loop-computed values (gathers, sincos) go through it so that the big register arrays keep static indices.
#define SCR_OFF (8 * 256)

__device__ __forceinline__ void app_gather_to_scratch(const nvfi_field_desc& f, const Bl* b, int h, float* scr) {
    // lane (j,h) holds channels 4*(2a+h)+c, a=0..5  (= the B-operand layout of the basis_mat layer)
#pragma unroll 1
    for (int a6 = 0; a6 < 6; ++a6) {
        const int q4 = 2 * a6 + h;
        float4 s0 = bl_sample4(f.aps[0], f.Ca, b[0], q4), s1 = bl_sample4(f.aps[1], f.Ca, b[1], q4), s2 = bl_sample4(f.aps[2], f.Ca, b[2], q4);
        float4 t0 = bl_sample4(f.apt[0], f.Ca, b[3], q4), t1 = bl_sample4(f.apt[1], f.Ca, b[4], q4), t2 = bl_sample4(f.apt[2], f.Ca, b[5], q4);
        scr[(4 * a6 + 0) * 256 + threadIdx.x] = ((s0.x * s1.x) * s2.x) * ((t0.x * t1.x) * t2.x);
        scr[(4 * a6 + 1) * 256 + threadIdx.x] = ((s0.y * s1.y) * s2.y) * ((t0.y * t1.y) * t2.y);
        scr[(4 * a6 + 2) * 256 + threadIdx.x] = ((s0.z * s1.z) * s2.z) * ((t0.z * t1.z) * t2.z);
        scr[(4 * a6 + 3) * 256 + threadIdx.x] = ((s0.w * s1.w) * s2.w) * ((t0.w * t1.w) * t2.w);
    }
}

template <bool STASH>
__global__ __launch_bounds__(WG_THREADS, 2) void k_app_fwd(AppArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_w = lds; float* lds_b = lds + LDS_W_FLOATS;
    const nvfi_field_desc& f = a.f;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int count = a.count ? *a.count : (int)a.n_direct;
    if ((int)(blockIdx.x * WG_SAMPLES) >= count) return;
    const int tile = blockIdx.x * 4 + wave_id();
    const int i = tile * TILE + (lane & 31);
    const bool active = i < count;
    const int n = active ? (a.list ? a.list[i] : i) : 0;
    float4 q = active ? a.xw[n] : zero4();
    const float tn = a.per_point_t ? q.w : SCHED_TN(a);
    float vd[3] = {0.f, 0.f, 0.f};
    if (active) {
        const float* vp = a.view_per_point ? a.view_per_point + 3 * (size_t)n : a.rays_d + 3 * (size_t)(n / a.S);
        vd[0] = vp[0]; vd[1] = vp[1]; vd[2] = vp[2];
    }
    float x[64];
    float* scr = lds_w + SCR_OFF;
    if (a.feat48) {
        // lane (j,h) holds channels 4*(2a+h)+c, a=0..5, of its sample
        const float* fp = a.feat48 + (size_t)(active ? i : 0) * 48 + 4 * h;
#pragma unroll
        for (int a6 = 0; a6 < 6; ++a6) {
            const float4 v = active ? ld4(fp + 8 * a6) : zero4();
            x[4 * a6 + 0] = v.x; x[4 * a6 + 1] = v.y; x[4 * a6 + 2] = v.z; x[4 * a6 + 3] = v.w;
        }
    } else {
        if (!a.feat_in) {
            Bl b[6];
            plane_setups(f, q.x, q.y, q.z, tn, b);
            app_gather_to_scratch(f, b, h, scr);
        }
#pragma unroll
        for (int s = 0; s < 24; ++s) x[s] = a.feat_in ? 0.f : scr[s * 256 + threadIdx.x];
    }
    float* st = STASH ? a.stash_f + (size_t)tile * (APP_F_ROWS * REGF) : nullptr;
    if (STASH) {
#pragma unroll
        for (int s = 0; s < 32; ++s) STASH_ST(st[s * REGF + lane], s < 24 ? x[s] : 0.f);
    }
    // positional encodings (tensorf_model_utils.py:176-183) -> scratch rows 0..35 (sin|cos selected by h)
#pragma unroll 1
    for (int e = 0; e < 18; ++e) {
        const int c = e / 6, k = e - 6 * c;
        const float fr = (float)(1 << k);
        const float pc = c == 0 ? q.x : (c == 1 ? q.y : q.z);
        const float vc = c == 0 ? vd[0] : (c == 1 ? vd[1] : vd[2]);
        scr[e * 256 + threadIdx.x] = trig_sel(pc * fr, h);
        scr[(18 + e) * 256 + threadIdx.x] = trig_sel(vc * fr, h);
    }
    // basis_mat: 48 -> 32 (no bias); its fragment occupies LDS rows below SCR_OFF
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.fb, RF_B, nullptr, 0);
    __syncthreads();
    f32x16 o1[1];
    acc_init<1>(o1, lds_b, h, false);
    if (!a.feat_in) layer_mfma<1, 24>(lds_w, lane, x, o1);
    else {      // features from the caller, in the D layout of the basis tile: register r of lane (n, h) is feature (r&3) + 8(r>>2) + 4h
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            o1[0][r] = (active && row < f.app_dim) ? a.feat_in[(size_t)n * f.app_dim + row] : 0.f;
        }
    }
    if (f.shading == 1) {
        // SHRender (tensorf_model_utils.py:292-296, sh.py:87-110): the 27 features are rows (r&3)+8(r>>2)+4h of the basis tile, split over the
        // lane pair (l, l+32); colour c = relu(sum_k SH_k(viewdir) feat[9c + k] + 0.5).  No MLP, no positional encodings.
        float sh[9];
        sh_bases9(vd, sh);
        float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < 27) part[row / 9] += sh[row % 9] * o1[0][r];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) part[c] += __shfl_xor(part[c], 32);
        if (active && h == 0) {
            float4 c = make_float4(fmaxf(part[0] + 0.5f, 0.f), fmaxf(part[1] + 0.5f, 0.f), fmaxf(part[2] + 0.5f, 0.f), 0.f);
            a.rgbs[a.rgb_dense ? n : i] = c;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = o1[0][r];
    x[16] = h ? q.x : vd[0]; x[17] = h ? q.y : vd[1]; x[18] = h ? q.z : vd[2];
#pragma unroll
    for (int e = 0; e < 18; ++e) { x[19 + e] = scr[e * 256 + threadIdx.x]; x[37 + e] = scr[(18 + e) * 256 + threadIdx.x]; }
#pragma unroll
    for (int s = 55; s < 64; ++s) x[s] = 0.f;
    if (STASH) stash_store<64>(st + 32 * REGF, lane, x);
    f32x16 acc[4];
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f1, RF_1, a.W.b1, 128);
    __syncthreads();
    acc_init<4>(acc, lds_b, h, true);
    layer_mfma<4, 55>(lds_w, lane, x, acc);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[16 * m + r] = fmaxf(acc[m][r], 0.f);
    if (STASH) stash_store<64>(st + 96 * REGF, lane, x);
    if (STASH) relu_mask_store(a.relu_mask + (size_t)tile * 256, lane, x);
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f2, RF_2, a.W.b2, 128);
    __syncthreads();
    acc_init<4>(acc, lds_b, h, true);
    layer_mfma<4, 64>(lds_w, lane, x, acc);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[16 * m + r] = fmaxf(acc[m][r], 0.f);
    if (STASH) stash_store<64>(st + 160 * REGF, lane, x);
    if (STASH) relu_mask_store(a.relu_mask + (size_t)tile * 256 + 128, lane, x);
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f3, RF_3, a.W.b3, 32);
    __syncthreads();
    acc_init<1>(o1, lds_b, h, true);
    layer_mfma<1, 64>(lds_w, lane, x, o1);
    if (active && h == 0) {
        float4 c = make_float4(sigmoid_f(o1[0][0]), sigmoid_f(o1[0][1]), sigmoid_f(o1[0][2]), 0.f);
        a.rgbs[a.rgb_dense ? n : i] = c;
    }
}

// (Round 3 built and measured a PERSISTENT form of this kernel - the whole 145.5 KB forward image resident in LDS, eight independent waves
//  per CU walking their own tiles with no barrier, bit-identical results: 0.372 ms per step against 0.300 ms for this kernel.  With
//  2 700 tiles on 2 048 waves a third of the waves run two tiles back to back while the rest idle, a wave keeps only 12 taps in flight
//  beside the MLP's registers, and the launch owns the CU.  Dropped)
static int launch_app_fwd(const AppArgs& aa, int64_t cap_samples, bool stash, hipStream_t st) {
    const unsigned wgs = (unsigned)((cap_samples + WG_SAMPLES - 1) / WG_SAMPLES);
    if (wgs == 0) return 0;
    if (stash) hipLaunchKernelGGL(k_app_fwd<true>, dim3(wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, aa);
    else hipLaunchKernelGGL(k_app_fwd<false>, dim3(wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, aa);
    LAUNCHCK();
    return 0;
}

// backward of the appearance branch for masked samples
__global__ __launch_bounds__(WG_THREADS, 2) void k_app_bwd(AppArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_w = lds; float* lds_b = lds + LDS_W_FLOATS;
    const nvfi_field_desc& f = a.f;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int count = *a.count;
    if ((int)(blockIdx.x * WG_SAMPLES) >= count) return;
    const int tile = blockIdx.x * 4 + wave_id();
    const int i = tile * TILE + (lane & 31);
    const bool active = i < count;
    const int n = active ? a.list[i] : 0;
    const float* stf = a.stash_f + (size_t)tile * (APP_F_ROWS * REGF);
    float* stb = a.stash_b + (size_t)tile * (APP_B_ROWS * REGF);
    float g[64];
    f32x16 acc[4];
    float gpts[3];
    if (f.shading == 1) {
        // SHRender backward: d pre_c = w * gr_c where the stored colour is positive (relu'), d feat[9c + k] = SH_k(viewdir) * d pre_c
        float gpre[3] = {0.f, 0.f, 0.f};
        float vd[3] = {0.f, 0.f, 0.f};
        if (active) {
            const int r = n / a.S;
            const float* vp = a.rays_d + 3 * (size_t)r;
            vd[0] = vp[0]; vd[1] = vp[1]; vd[2] = vp[2];
            if (a.g_rgb) {
                const float4 pre = a.rgb_pre[r];
                const float pv[3] = {pre.x, pre.y, pre.z};
                const float4 c = a.rgbs[i];
                const float cv[3] = {c.x, c.y, c.z};
                const float w = a.weight[n];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float gr = (pv[k] >= 0.f && pv[k] <= 1.f) ? a.g_rgb[3 * (size_t)r + k] : 0.f;
                    gpre[k] = cv[k] > 0.f ? w * gr : 0.f;
                }
            }
        }
        float sh[9];
        sh_bases9(vd, sh);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            acc[0][r] = row < 27 ? sh[row % 9] * gpre[row / 9] : 0.f;
        }
        gpts[0] = gpts[1] = gpts[2] = 0.f;
    } else {
    // seeds: go_c = w * gr_c * c(1-c) in rows 0..2 of a D tile (lane h=0 regs 0..2)
    {
        float go[3] = {0.f, 0.f, 0.f};
        if (active && h == 0 && a.g_rgb) {
            const int r = n / a.S;
            const float4 pre = a.rgb_pre[r];
            const float pv[3] = {pre.x, pre.y, pre.z};
            const float4 c = a.rgbs[i];
            const float cv[3] = {c.x, c.y, c.z};
            const float w = a.weight[n];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float gr = (pv[k] >= 0.f && pv[k] <= 1.f) ? a.g_rgb[3 * (size_t)r + k] : 0.f;
                go[k] = w * gr * cv[k] * (1.f - cv[k]);
            }
        }
        g[0] = go[0]; g[1] = go[1]; g[2] = go[2]; g[3] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) STASH_ST(stb[s * REGF + lane], s < 3 ? g[s] : 0.f);
    }
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.t3, RT_3, nullptr, 0);
    __syncthreads();
    acc_init<4>(acc, lds_b, 0, false);
    layer_mfma<4, 4>(lds_w, lane, g, acc);
    {
        const unsigned* mk = a.relu_mask + (size_t)tile * 256 + 128;
        const unsigned mlo = mk[lane], mhi = mk[64 + lane];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[16 * m + r] = (((m < 2 ? mlo : mhi) >> ((16 * m + r) & 31)) & 1u) ? acc[m][r] : 0.f;
    }
    stash_store<64>(stb + 16 * REGF, lane, g);
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.t2, RT_2, nullptr, 0);
    __syncthreads();
    acc_init<4>(acc, lds_b, 0, false);
    layer_mfma<4, 64>(lds_w, lane, g, acc);
    {
        const unsigned* mk = a.relu_mask + (size_t)tile * 256;
        const unsigned mlo = mk[lane], mhi = mk[64 + lane];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[16 * m + r] = (((m < 2 ? mlo : mhi) >> ((16 * m + r) & 31)) & 1u) ? acc[m][r] : 0.f;
    }
    stash_store<64>(stb + 80 * REGF, lane, g);
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.t1, RT_1, nullptr, 0);
    __syncthreads();
    acc_init<4>(acc, lds_b, 0, false);
    layer_mfma<4, 64>(lds_w, lane, g, acc);
    // acc = gradient wrt the 110 input slots (RENDER_IN layout)
    {
        const float* xin = stf + 32 * REGF;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = h ? acc[1][c] : 0.f;          // slots 16..18: h=1 holds raw pts
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int sl = 19 + c * 6 + k;
                const float mine = STASH_LD(xin[sl * REGF + lane]);
                const float other = __shfl_xor(mine, 32);
                const float fr = (float)(1 << k);
                s += (h ? -fr * other : fr * other) * acc[sl >> 4][sl & 15];
            }
            s += __shfl_xor(s, 32);
            gpts[c] = s;
        }
    }
    }   // MLP_PE
    // gfeat (tile 0) -> stash, then basis^T -> gg (48 channels in gather layout)
#pragma unroll
    for (int r = 0; r < 16; ++r) { g[r] = acc[0][r]; STASH_ST(stb[(144 + r) * REGF + lane], g[r]); }
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.tb, RT_B, nullptr, 0);
    __syncthreads();
    f32x16 gg[2];
    acc_init<2>(gg, lds_b, 0, false);
    layer_mfma<2, 16>(lds_w, lane, g, gg);
    // plane backward for this lane's 24 channels (channel grads via LDS scratch -> non-unrolled loop)
    float* scr = lds_w + SCR_OFF;
    __syncthreads();
#pragma unroll
    for (int s2 = 0; s2 < 24; ++s2) scr[s2 * 256 + threadIdx.x] = active ? gg[s2 >> 4][s2 & 15] : 0.f;
    if (active && a.gg) {   // per-sample channel gradients for the channel-parallel scatter kernel: gg[i][4*(2a+h)+c]
#pragma unroll
        for (int a6 = 0; a6 < 6; ++a6) {
            const int s0 = 4 * a6;
            *reinterpret_cast<float4*>(a.gg + (size_t)i * 48 + 4 * (2 * a6 + h)) =
                make_float4(gg[s0 >> 4][s0 & 15], gg[s0 >> 4][(s0 & 15) + 1], gg[s0 >> 4][(s0 & 15) + 2], gg[s0 >> 4][(s0 & 15) + 3]);
        }
    }
    if (!a.plane_tail) {   // the plane part of the coordinate gradient is added by k_og<48, true> (scatter.hip) when it is needed at all
        if (active && h == 0) a.gxw[n] = make_float4(gpts[0], gpts[1], gpts[2], 0.f);
        return;
    }
    float4 q = active ? a.xw[n] : zero4();
    Bl b[6];
    plane_setups(f, q.x, q.y, q.z, SCHED_TN(a), b);
    float gx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gy[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int a6 = 0; a6 < 6; ++a6) {
        const int q4 = 2 * a6 + h;
        const float4 gq = make_float4(scr[(4 * a6) * 256 + threadIdx.x], scr[(4 * a6 + 1) * 256 + threadIdx.x],
                                      scr[(4 * a6 + 2) * 256 + threadIdx.x], scr[(4 * a6 + 3) * 256 + threadIdx.x]);
        float4 v[6];
        v[0] = bl_sample4(f.aps[0], f.Ca, b[0], q4); v[1] = bl_sample4(f.aps[1], f.Ca, b[1], q4); v[2] = bl_sample4(f.aps[2], f.Ca, b[2], q4);
        v[3] = bl_sample4(f.apt[0], f.Ca, b[3], q4); v[4] = bl_sample4(f.apt[1], f.Ca, b[4], q4); v[5] = bl_sample4(f.apt[2], f.Ca, b[5], q4);
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            float4 o = gq;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k != p) { o.x *= v[k].x; o.y *= v[k].y; o.z *= v[k].z; o.w *= v[k].w; }
            const float* pl = p == 0 ? f.aps[0] : p == 1 ? f.aps[1] : p == 2 ? f.aps[2] : p == 3 ? f.apt[0] : p == 4 ? f.apt[1] : f.apt[2];
            float* gp = p == 0 ? a.g.aps[0] : p == 1 ? a.g.aps[1] : p == 2 ? a.g.aps[2] : p == 3 ? a.g.apt[0] : p == 4 ? a.g.apt[1] : a.g.apt[2];
            bl_backward4(pl, (active && !a.gg) ? gp : nullptr, f.Ca, b[p], q4, o, gx[p], gy[p]);
        }
    }
    float g3[3] = {0.f, 0.f, 0.f};
    {
        float mx, my;
        plane_mults(f, 0, mx, my); g3[0] += gx[0] * mx; g3[1] += gy[0] * my;
        plane_mults(f, 1, mx, my); g3[0] += gx[1] * mx; g3[2] += gy[1] * my;
        plane_mults(f, 2, mx, my); g3[1] += gx[2] * mx; g3[2] += gy[2] * my;
        plane_mults(f, 3, mx, my); g3[2] += gx[3] * mx;
        plane_mults(f, 4, mx, my); g3[1] += gx[4] * mx;
        plane_mults(f, 5, mx, my); g3[0] += gx[5] * mx;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) g3[c] += __shfl_xor(g3[c], 32);
    if (active && h == 0) a.gxw[n] = make_float4(g3[0] + gpts[0], g3[1] + gpts[1], g3[2] + gpts[2], 0.f);
}


// ================================================================ plane-gradient scatter (channel-parallel)
// One wave walks a few samples; lanes are CHANNELS (x the two x-taps for 24 channels), so every atomic
// instruction adds a contiguous run of one or two texel vectors (96..192 B) instead of 64 scattered words:
// ~14x fewer cache-line atomic operations than one-thread-per-sample scattering.
#define SCATTER_SPW 8
__device__ __forceinline__ float dpp_xor1(float v) { return __shfl_xor(v, 1); }

// DET: the gradient pointers address int64 shadow planes and every contribution is added as a fixed-point integer (2^50 per unit):
// integer addition is associative, so the sums are bit-identical whatever order the atomics arrive in (NVFI_DETERMINISTIC=1).
#define DET_SCALE 1125899906842624.0      /* 2^50: +-8192 of range, 8.9e-16 of resolution */
template <bool DET>
__device__ __forceinline__ void grad_add(float* g, size_t idx, float v) {
    // (a single contribution saturates at the int64 range instead of wrapping; running SUMS beyond +-8192 still wrap - test mode)
    if (DET) atomicAdd(reinterpret_cast<unsigned long long*>(g) + idx, (unsigned long long)__double2ll_rn(fmin(fmax((double)v * DET_SCALE, -9.2e18), 9.2e18)));
    else atomicAdd(g + idx, v);
}
__global__ void k_det_finish(const long long* __restrict__ shadow, float* __restrict__ g, int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) g[i] += (float)((double)shadow[i] * (1.0 / DET_SCALE));
}
template <int C, bool DET = false>
__global__ __launch_bounds__(256) void k_plane_scatter(ScatterArgs a) {
    const int lane = threadIdx.x & 63;
    const int wg = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int count = *a.count;
    const int i0 = wg * SCATTER_SPW;
    if (i0 >= count) return;
    const nvfi_field_desc& f = a.f;
    const float* pl[6]; float* gp[6];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        pl[p] = C == 24 ? f.dps[p] : f.aps[p]; pl[3 + p] = C == 24 ? f.dpt[p] : f.apt[p];
        gp[p] = C == 24 ? a.g.dps[p] : a.g.aps[p]; gp[3 + p] = C == 24 ? a.g.dpt[p] : a.g.apt[p];
    }
    const int ch = C == 24 ? (lane >> 1) : lane;
    const int dx0 = C == 24 ? (lane & 1) : 0;
    const bool lane_on = lane < 48;
#pragma unroll 1
    for (int k = 0; k < SCATTER_SPW; ++k) {
        const int i = i0 + k;
        if (i >= count) break;
        const int n = __builtin_amdgcn_readfirstlane(a.list[i]);
        const float4 q = a.xw[n];
        Bl b[6];
        plane_setups(f, q.x, q.y, q.z, SCHED_TN(a), b);
        float gch;
        if (C == 24) gch = a.gxpre[n];
        else gch = lane_on ? a.gg[(size_t)i * 48 + ch] : 0.f;
        float val[6];
        if (C == 24) {
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const bool my0 = dx0 ? b[p].m1 : b[p].m0, my1 = dx0 ? b[p].m3 : b[p].m2;
                const float wx = dx0 ? b[p].w : b[p].e;
                const size_t o0 = (size_t)(b[p].base + dx0) * C + ch, o1 = o0 + (size_t)b[p].W * C;
                const float v0 = (lane_on && my0) ? pl[p][o0] : 0.f, v1 = (lane_on && my1) ? pl[p][o1] : 0.f;
                const float part = v0 * (wx * b[p].s) + v1 * (wx * b[p].n);
                val[p] = part + dpp_xor1(part);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const size_t o0 = (size_t)b[p].base * C + ch, o1 = o0 + (size_t)b[p].W * C;
                const float v0 = (lane_on && b[p].m0) ? pl[p][o0] : 0.f, v1 = (lane_on && b[p].m1) ? pl[p][o0 + C] : 0.f;
                const float v2 = (lane_on && b[p].m2) ? pl[p][o1] : 0.f, v3 = (lane_on && b[p].m3) ? pl[p][o1 + C] : 0.f;
                val[p] = v0 * (b[p].e * b[p].s) + v1 * (b[p].w * b[p].s) + v2 * (b[p].e * b[p].n) + v3 * (b[p].w * b[p].n);
            }
        }
        // prefix/suffix products: other_p = prod_{k != p} val[k]
        float L[6], Rr[6];
        L[0] = gch; 
#pragma unroll
        for (int p = 1; p < 6; ++p) L[p] = L[p - 1] * val[p - 1];
        Rr[5] = 1.f;
#pragma unroll
        for (int p = 4; p >= 0; --p) Rr[p] = Rr[p + 1] * val[p + 1];
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            if (!gp[p] || !lane_on || !((a.plane_mask >> p) & 1)) continue;
            const float o = L[p] * Rr[p];
            if (C == 24) {
                const bool my0 = dx0 ? b[p].m1 : b[p].m0, my1 = dx0 ? b[p].m3 : b[p].m2;
                const float wx = dx0 ? b[p].w : b[p].e;
                const size_t o0 = (size_t)(b[p].base + dx0) * C + ch, o1 = o0 + (size_t)b[p].W * C;
                if (my0) grad_add<DET>(gp[p], o0, (wx * b[p].s) * o);
                if (my1) grad_add<DET>(gp[p], o1, (wx * b[p].n) * o);
            } else {
                const size_t o0 = (size_t)b[p].base * C + ch, o1 = o0 + (size_t)b[p].W * C;
                if (b[p].m0) grad_add<DET>(gp[p], o0, (b[p].e * b[p].s) * o);
                if (b[p].m1) grad_add<DET>(gp[p], o0 + C, (b[p].w * b[p].s) * o);
                if (b[p].m2) grad_add<DET>(gp[p], o1, (b[p].e * b[p].n) * o);
                if (b[p].m3) grad_add<DET>(gp[p], o1 + C, (b[p].w * b[p].n) * o);
            }
        }
    }
}


// Variant with LDS-privatised TIME planes.  The time coordinate is a per-call scalar, so every sample scatters into the
// same two rows of the three time planes (2 x G x C floats each): per-workgroup LDS accumulators absorb that contention
// and are flushed once; space planes keep the coalesced global atomics.  A workgroup handles 24 channels starting at c0
// of planes with CT channels per texel (density: CT=24, one group; appearance: CT=48, two groups on blockIdx.y).
template <int CT>
__global__ __launch_bounds__(1024) void k_plane_scatter_lds(ScatterArgs a) {
    extern __shared__ __attribute__((aligned(16))) float acc_lds[];   // [3 planes][2 rows][gmax][24]
    const nvfi_field_desc& f = a.f;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int c0 = blockIdx.y * 24;
    const int gmax = a.gmax;
    for (int k = threadIdx.x; k < 6 * gmax * 24; k += blockDim.x) acc_lds[k] = 0.f;
    __syncthreads();
    const int count = *a.count;
    const float* pl[6]; float* gp[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        pl[p] = CT == 24 ? f.dps[p] : f.aps[p]; pl[3 + p] = CT == 24 ? f.dpt[p] : f.apt[p];
        gp[p] = CT == 24 ? a.g.dps[p] : a.g.aps[p];
    }
    const int ch = lane >> 1, dx0 = lane & 1;
    const bool lane_on = lane < 48;
    const int wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x * nwv + wv), wave_total = gridDim.x * nwv;
    // two samples per trip: the dependent chain list -> position -> taps is pure latency, so both chains are issued together
    constexpr int U = 2;
    // each wave walks a contiguous run of the (ray-ordered) list: neighbouring samples share texels, so their atomics
    // stay in one wave / one XCD's L2 instead of bouncing the same lines between XCDs
#ifdef NVFI_EXP_SCATTER_STRIDED
    const int i_lo = wave_global, i_hi = count, i_step = U * wave_total, u_step = wave_total;
#else
    const int chunk = (count + wave_total - 1) / wave_total;
    const int i_lo = wave_global * chunk, i_hi = min(count, i_lo + chunk), i_step = U, u_step = 1;
#endif
#pragma unroll 1
    for (int i0 = i_lo; i0 < i_hi; i0 += i_step) {
        float4 q[U]; float gch[U]; bool on[U]; float o[U][6];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * u_step;
            on[u] = i < i_hi;
            const int n = __builtin_amdgcn_readfirstlane(a.list[on[u] ? i : i0]);
            q[u] = a.xw[n];
            if (CT == 24) gch[u] = a.gxpre[n];
            else gch[u] = (lane_on && on[u]) ? a.gg[(size_t)i * 48 + c0 + ch] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            Bl b[6];
            plane_setups(f, q[u].x, q[u].y, q[u].z, SCHED_TN(a), b);
            float val[6];
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                const bool my0 = dx0 ? b[p].m1 : b[p].m0, my1 = dx0 ? b[p].m3 : b[p].m2;
                const float wx = dx0 ? b[p].w : b[p].e;
                const size_t o0 = (size_t)(b[p].base + dx0) * CT + c0 + ch, o1 = o0 + (size_t)b[p].W * CT;
                const float v0 = (lane_on && my0) ? pl[p][o0] : 0.f, v1 = (lane_on && my1) ? pl[p][o1] : 0.f;
                const float part = v0 * (wx * b[p].s) + v1 * (wx * b[p].n);
                val[p] = part + dpp_xor1(part);
            }
            float L[6], Rr[6];
            L[0] = gch[u];
#pragma unroll
            for (int p = 1; p < 6; ++p) L[p] = L[p - 1] * val[p - 1];
            Rr[5] = 1.f;
#pragma unroll
            for (int p = 4; p >= 0; --p) Rr[p] = Rr[p + 1] * val[p + 1];
#pragma unroll
            for (int p = 0; p < 6; ++p) o[u][p] = L[p] * Rr[p];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!(lane_on && on[u])) continue;
            Bl b[6];
            plane_setups(f, q[u].x, q[u].y, q[u].z, SCHED_TN(a), b);
#pragma unroll
            for (int p = 0; p < 3; ++p) {       // space planes: coalesced global atomics
                if (!gp[p]) continue;
                const bool my0 = dx0 ? b[p].m1 : b[p].m0, my1 = dx0 ? b[p].m3 : b[p].m2;
                const float wx = dx0 ? b[p].w : b[p].e;
                const size_t o0 = (size_t)(b[p].base + dx0) * CT + c0 + ch, o1 = o0 + (size_t)b[p].W * CT;
                if (my0) atomicAdd(gp[p] + o0, (wx * b[p].s) * o[u][p]);
                if (my1) atomicAdd(gp[p] + o1, (wx * b[p].n) * o[u][p]);
            }
#pragma unroll
            for (int p = 3; p < 6; ++p) {       // time planes: workgroup-private LDS rows (y0, y0+1 are call constants)
                const bool my0 = dx0 ? b[p].m1 : b[p].m0, my1 = dx0 ? b[p].m3 : b[p].m2;
                const float wx = dx0 ? b[p].w : b[p].e;
                const int x = b[p].base - SCHED_Y0(a) * b[p].W + dx0;        // column inside the row
                float* r0 = acc_lds + ((size_t)((p - 3) * 2 + 0) * gmax + x) * 24 + ch;
                if (my0) atomicAdd(r0, (wx * b[p].s) * o[u][p]);
                if (my1) atomicAdd(r0 + (size_t)gmax * 24, (wx * b[p].n) * o[u][p]);
            }
        }
    }
    __syncthreads();
    // flush the private rows
    const int Gc[3] = {f.G[2], f.G[1], f.G[0]};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float* g = CT == 24 ? a.g.dpt[p] : a.g.apt[p];
        if (!g) continue;
        for (int dy = 0; dy < 2; ++dy) {
            const int y = SCHED_Y0(a) + dy;
            if (y < 0 || y >= f.K) continue;
            for (int k = threadIdx.x; k < Gc[p] * 24; k += blockDim.x) {
                const int x = k / 24, c = k - 24 * x;
                const float v = acc_lds[((size_t)(p * 2 + dy) * gmax + x) * 24 + c];
                if (v != 0.f) atomicAdd(g + ((size_t)y * Gc[p] + x) * CT + c0 + c, v);
            }
        }
    }
}

// Optional side stream for the plane-gradient scatters (NVFI_SIDE_STREAM=1): they are bound by L2 atomics and leave the MFMA
// pipes idle, so the backward can fork them next to the weight-gradient / RK2-adjoint kernels and join before returning.
// Off by default: it gained 2.5 % while those kernels ran two workgroups per CU, but since they own a CU each (one wave per
// SIMD with the whole register file, engine.h: FragPipe) a scatter wave cannot co-reside with them and the fork only splits CUs.
struct SideStream {
    hipStream_t s = nullptr; hipEvent_t fork[2] = {nullptr, nullptr}, join = nullptr; int state = -1;
    int get() {
        if (state >= 0) return state;
        const char* e = getenv("NVFI_SIDE_STREAM");
        state = (e && atoi(e) != 0) ? 1 : 0;
        if (state) {
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) state = 0;
            for (int i = 0; i < 2 && state; ++i) if (hipEventCreateWithFlags(&fork[i], hipEventDisableTiming) != hipSuccess) state = 0;
            if (state && hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) state = 0;
        }
        return state;
    }
};
static SideStream g_side;
// NVFI_BWD_FORK (flags bit 16), for a caller that drives ONE stream: parts of the render backward run on a library-owned stream, with their
// own tile-sort workspace / slab region, joined before the call returns.
//   keyframe time: the two halves - appearance (k_app_bwd, k_og<48>, tile scatter, render-MLP weight gradients) and density (k_weights_bwd,
//     k_og<24>, tile scatter) - share nothing but inputs: the density half runs beside the appearance half;
//   non-keyframe time: the coordinate gradients chain k_app_bwd -> k_og<48> -> k_og<24> -> RK2 adjoint -> velocity-net weight gradients; the
//     plane scatters of both branches and the render-MLP weight gradients hang off that chain and run beside it.
// A caller that already overlaps several renders / the PDE term on its own streams (bench.py's fused driver) leaves the bit off.
struct ForkStream {
    hipStream_t s = nullptr; hipEvent_t fork = nullptr, fork2 = nullptr, join = nullptr; int state = -1;
    std::once_flag once;
    int get() {
        // created on the device that is current at the first call, once (autograd runs backward nodes on per-device worker threads)
        std::call_once(once, [this] {
            int st = 1;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) st = 0;
            if (st && hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) st = 0;
            if (st && hipEventCreateWithFlags(&fork2, hipEventDisableTiming) != hipSuccess) st = 0;
            if (st && hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) st = 0;
            state = st;
        });
        return state;
    }
};
// one fork stream per device ordinal: a second device (or a second field on another device) in the same process gets its own stream and
// events instead of launching its forked half on the first device's
#define NVFI_MAX_DEVICES 16
static ForkStream g_forks[NVFI_MAX_DEVICES];
static ForkStream& fork_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NVFI_MAX_DEVICES) dev = 0;
    return g_forks[dev];
}
#define g_fork (fork_of_current_device())

// NVFI_DETERMINISTIC=1: bit-reproducible plane gradients for tests.  The sorted-tile path sums in an order that
// depends on atomic cursors; this mode takes the plain atomic scatter instead and accumulates in fixed point (k_plane_scatter<C, true>).
static bool det_mode() { static int d = -1; if (d < 0) { const char* e = getenv("NVFI_DETERMINISTIC"); d = (e && atoi(e) != 0) ? 1 : 0; } return d != 0; }
static int64_t plane_elems(const nvfi_field_desc* f, int64_t* off /* [12]: dps[3] dpt[3] aps[3] apt[3] */) {
    const int A[3] = {0, 0, 1}, Bx[3] = {1, 2, 2}, Cc[3] = {2, 1, 0};
    int64_t n = 0;
    for (int i = 0; i < 3; ++i) { off[i] = n; n += (int64_t)f->G[A[i]] * f->G[Bx[i]] * f->Cd; }
    for (int i = 0; i < 3; ++i) { off[3 + i] = n; n += (int64_t)f->K * f->G[Cc[i]] * f->Cd; }
    for (int i = 0; i < 3; ++i) { off[6 + i] = n; n += (int64_t)f->G[A[i]] * f->G[Bx[i]] * f->Ca; }
    for (int i = 0; i < 3; ++i) { off[9 + i] = n; n += (int64_t)f->K * f->G[Cc[i]] * f->Ca; }
    return n;
}
static int launch_scatter_det(ScatterArgs& sa, int C, int64_t N, hipStream_t st) {
    const unsigned sc_blocks = (unsigned)((N + 4 * SCATTER_SPW - 1) / (4 * SCATTER_SPW));
    if (C == 24) hipLaunchKernelGGL((k_plane_scatter<24, true>), dim3(sc_blocks), dim3(256), 0, st, sa);
    else hipLaunchKernelGGL((k_plane_scatter<48, true>), dim3(sc_blocks), dim3(256), 0, st, sa);
    LAUNCHCK();
    return 0;
}
// plane-gradient scatter: LDS-privatised time rows when they fit, plain channel-parallel atomics otherwise
static int launch_scatter(const nvfi_field_desc* f, ScatterArgs& sa, int C, int64_t N, float tn, hipStream_t st) {
    int gmax = f->G[0] > f->G[1] ? f->G[0] : f->G[1];
    gmax = gmax > f->G[2] ? gmax : f->G[2];
    const size_t lds = (size_t)6 * gmax * 24 * sizeof(float);
    if (lds <= 150 * 1024) {
        static bool attr = false;
        if (!attr) {
            HIPCK(hipFuncSetAttribute((const void*)k_plane_scatter_lds<24>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            HIPCK(hipFuncSetAttribute((const void*)k_plane_scatter_lds<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            attr = true;
        }
        // row y0 of the time planes: same arithmetic as bl_setup on the per-call time coordinate
        const float y = (tn + 1.f) * ((float)(f->K - 1) / 2.f);
        float yf = floorf(y);
        yf = fminf(fmaxf(yf, -4.f), (float)f->K + 2.f);
        sa.y0 = (int)yf; sa.gmax = gmax;
        if (C == 24) hipLaunchKernelGGL(k_plane_scatter_lds<24>, dim3(256, 1), dim3(1024), lds, st, sa);
        else hipLaunchKernelGGL(k_plane_scatter_lds<48>, dim3(256, 2), dim3(1024), lds, st, sa);
    } else {
        const unsigned sc_blocks = (unsigned)((N + 4 * SCATTER_SPW - 1) / (4 * SCATTER_SPW));
        if (C == 24) hipLaunchKernelGGL(k_plane_scatter<24>, dim3(sc_blocks), dim3(256), 0, st, sa);
        else hipLaunchKernelGGL(k_plane_scatter<48>, dim3(sc_blocks), dim3(256), 0, st, sa);
    }
    LAUNCHCK();
    return 0;
}

// ================================================================ host: fragment jobs, launches, ABI
int pack_render_frags(const nvfi_field_desc* f, float* buf, RenderFrags* out, PackJobs* jobs) {
    float* p = buf;
    auto take = [&](int n) { float* r = p; p += n; return r; };
    float* fb = take(RF_B); float* f1 = take(RF_1); float* f2 = take(RF_2); float* f3 = take(RF_3);
    float* b1 = take(128); float* b2 = take(128); float* b3 = take(32);
    float* t3 = take(RT_3); float* t2 = take(RT_2); float* t1 = take(RT_1); float* tb = take(RT_B);
    auto add = [&](const float* W, const float* b, float* frag, float* bfrag, int o, int in, int MT, int NS, int rk, int sk, int tr) {
        if (jobs->n >= MAX_PACK_JOBS) return 1;
        PackJob& J = jobs->j[jobs->n++];
        J.W = W; J.b = b; J.frag = frag; J.bfrag = bfrag; J.out = o; J.in = in; J.MT = MT; J.NS = NS;
        J.row_kind = rk; J.slot_kind = sk; J.transposed = tr; J.x4 = 0;
        return 0;
    };
    int rc = 0;
    rc |= add(f->basis, nullptr, fb, nullptr, f->app_dim, f->Ca, 1, 24, RK_NATURAL, SK_HIDDEN, 0);
    if (f->shading == 0) {      // SH shading has no render MLP (tensorf_base.py:196-197)
        rc |= add(f->rW[0], f->rb[0], f1, b1, 128, 110, 4, 55, RK_NATURAL, SK_RENDER_IN, 0);
        rc |= add(f->rW[1], f->rb[1], f2, b2, 128, 128, 4, 64, RK_NATURAL, SK_HIDDEN, 0);
        rc |= add(f->rW[2], f->rb[2], f3, b3, 3, 128, 1, 64, RK_NATURAL, SK_HIDDEN, 0);
        rc |= add(f->rW[2], nullptr, t3, nullptr, 3, 128, 4, 4, RK_NATURAL, SK_HIDDEN, 1);
        rc |= add(f->rW[1], nullptr, t2, nullptr, 128, 128, 4, 64, RK_NATURAL, SK_HIDDEN, 1);
        rc |= add(f->rW[0], nullptr, t1, nullptr, 128, 110, 4, 64, RK_RENDER_IN, SK_HIDDEN, 1);
    }
    rc |= add(f->basis, nullptr, tb, nullptr, f->app_dim, f->Ca, 2, 16, RK_NATURAL, SK_HIDDEN, 1);
    if (rc) return nvfi_fail(3, "too many pack jobs");
    out->fb = fb; out->f1 = f1; out->b1 = b1; out->f2 = f2; out->b2 = b2; out->f3 = f3; out->b3 = b3;
    out->t3 = t3; out->t2 = t2; out->t1 = t1; out->tb = tb;
    return 0;
}

static int check_desc(const nvfi_field_desc* f) {
    if (f->Cd != 24 || f->Ca != 48 || f->app_dim != (f->shading == 1 ? 27 : 32) || f->shading < 0 || f->shading > 1)
        return nvfi_fail(2, "unsupported component counts Cd=%d Ca=%d app_dim=%d shading=%d (kernels are built for 24/48/32 with MLP_PE, 24/48/27 with SH)", f->Cd, f->Ca, f->app_dim, f->shading);
    if (f->n_samples < 1 || f->n_samples > 1024) return nvfi_fail(2, "n_samples=%d outside [1,1024]", f->n_samples);
    return 0;
}

// number of RK2 steps and their (dt, t) sequence for a per-call scalar t (tensorf_keyframe.py:575-609)
static int rk_schedule(const nvfi_field_desc* f, float t, int flags, float* base_out, float* dts, float* tcs) {
    float base = (flags & NVFI_TRANSFER) ? 0.f : snap_base(*f, t);
    *base_out = base;
    if (!f->use_vel || is_close(t, base)) return 0;
    float dtm = dt_max_of(*f), off = t - base, tc = t;
    int n = 0;
    while (fabsf(off) > 0.f) {
        if (n >= MAX_RK_STEPS) return -1;
        float m = fabsf(off) < dtm ? fabsf(off) : dtm;
        float dt = off > 0.f ? m : -m;
        dts[n] = dt; tcs[n] = tc;
        off = off - dt; tc = tc - dt;
        ++n;
    }
    return n;
}

// Device-side schedule (hipGraph replay): same arithmetic as rk_schedule / norm_time / the y0 of the LDS scatter variants, from a time
// held in device memory.  The launch plan (number of RK2 steps -> which kernels run, stash sizes) was fixed on the host from `t_plan`;
// if the device time implies a different step count the record falls back to the plan's schedule and raises sched[3] (mirrored into
// counters[7] by k_counters) - results are then those of t_plan, never undefined.
struct SchedArgs {
    nvfi_field_desc f; const float* t_dev; int flags; int nsteps_plan; float tn_plan; float dt_plan[4]; float tc_plan[4]; float* sched;
};
__device__ void sched_body(const SchedArgs& a) {
    const nvfi_field_desc& f = a.f;
    float* S = a.sched;
    const float t = *a.t_dev;
    const float base = (a.flags & NVFI_TRANSFER) ? 0.f : snap_base(f, t);
    int n = 0;
    bool bad = false;
    if (f.use_vel && !is_close(t, base)) {
        const float dtm = dt_max_of(f);
        float off = t - base, tc = t;
        while (fabsf(off) > 0.f) {
            if (n >= MAX_RK_STEPS) { bad = true; break; }
            const float m = fabsf(off) < dtm ? fabsf(off) : dtm;
            const float dt = off > 0.f ? m : -m;
            S[SCHED_DT + n] = dt; S[SCHED_TC + n] = tc;
            off = off - dt; tc = tc - dt;
            ++n;
        }
    }
    float tn = f.use_vel ? norm_time(f, base) : norm_time(f, t);
    if (bad || n != a.nsteps_plan) {          // not the planned launch shape: render the planned time instead, and say so
        bad = true;
        n = a.nsteps_plan;
        for (int s = 0; s < n && s < 4; ++s) { S[SCHED_DT + s] = a.dt_plan[s]; S[SCHED_TC + s] = a.tc_plan[s]; }
        tn = a.tn_plan;
    }
    const float y = (tn + 1.f) * ((float)(f.K - 1) / 2.f);
    float yf = floorf(y);
    yf = fminf(fmaxf(yf, -4.f), (float)f.K + 2.f);
    S[0] = tn; S[1] = __int_as_float((int)yf); S[2] = __int_as_float(n); S[3] = __int_as_float(bad ? 1 : 0);
}
__global__ void k_sched(SchedArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    sched_body(a);
}

// The head of a render call in ONE workgroup: clears the call's counters / sort histograms / look-back words (the forward's memset), derives the
// device-side schedule when the frame time lives in device memory (k_sched), and tests the ray origins against the box (k_any_inside,
// tensorf_base.py:294).  Used for R <= PROLOGUE_MAX_RAYS; larger calls keep the three launches.
#define PROLOGUE_MAX_RAYS 8192
struct PrologueArgs { SchedArgs sc; int do_sched; int64_t R; const float* o; int* zero_from; int64_t zero_ints; int* inside; };
__global__ __launch_bounds__(256) void k_prologue(PrologueArgs a) {
    __shared__ int hit_any;
    if (threadIdx.x == 0) hit_any = 0;
    int4* z4 = reinterpret_cast<int4*>(a.zero_from);         // (256-byte aligned, a multiple of 256 bytes)
    for (int64_t k = threadIdx.x; k < a.zero_ints / 4; k += 256) z4[k] = make_int4(0, 0, 0, 0);
    __syncthreads();
    const nvfi_field_desc& f = a.sc.f;
    bool hit = false;
    for (int64_t i = threadIdx.x; i < a.R * 3; i += 256) {
        const int c = (int)(i % 3);
        const float v = a.o[i];
        if (f.aabb[c] <= v && v <= f.aabb[3 + c]) hit = true;
    }
    if (__any(hit) && (threadIdx.x & 63) == 0) hit_any = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        *a.inside = hit_any;       // (inside the zeroed range: written after the clear, by the same workgroup)
        if (a.do_sched) sched_body(a.sc);
    }
}


struct RenderPlan {
    int64_t N, cap_tiles;
    int nsteps;
    float* sched;       // device-side schedule record (SCHED_FLOATS), written by k_sched when the call passes a device time
    int* counters;      // [0] V, [1] M, [2] inside flag
    int *cnt_v, *off_v, *cnt_m, *off_m, *vlist, *mlist, *cnt_r, *off_r, *rlist;
    uint8_t *valid, *mflag, *rflag;
    float4 *xw, *rgbs, *rgb_pre, *gxw, *gxk;
    float *xpre, *gxpre;
    float *vel_frag, *render_frag, *vel_x4, *vel_x4b; void* img16; void* x6img; void* x6imgT;
    TileWork tw2; float* slabs2;
    float *app_f, *app_b, *zst, *x0st, *rec, *gst, *gg, *maskv, *mask_frag;
    unsigned* app_relu;
    float *slabs;
    long long* shadow;         // NVFI_DETERMINISTIC: int64 fixed-point images of the 12 plane gradients
    int64_t zero_bytes;        // counters .. end of the sort histograms / look-back words: zeroed by the forward's single fill (or k_prologue)
    unsigned long long *lb_s, *lb_w;   // look-back status words of k_sample_fill / k_weights_fill
    float* mse_part;
    TileWork tw; bool tiles;   // sorted-tile plane scatter (scatter.hip); tiles = false: grid too large, atomic scatter instead
    int64_t total;
};
#define NSLAB_MAX 256        // (1024 while NVFI_NSLAB could be swept: 0.6 GB of slab workspace nobody wrote)
#define NSLAB 256            // slab capacity of a weight-gradient job (one per CU; the NVFI_NSLAB sweep of round 2 was retired in round 6)
#define SLAB_FLOATS (128 * 128 + 128)

static bool use_tiles() { static int u = -1; if (u < 0) { const char* e = getenv("NVFI_SCATTER_TILES"); u = e ? atoi(e) : 1; } return u != 0; }
static void plan_render(const nvfi_field_desc* f, int64_t R, int flags, int nsteps, void* ws, RenderPlan* P) {
    Bump B{(char*)ws, 0, 0};
    const int64_t N = R * f->n_samples;
    const bool train = flags & NVFI_TRAIN;
    P->N = N; P->nsteps = nsteps;
    P->cap_tiles = (N + WG_SAMPLES - 1) / WG_SAMPLES * 4;   // whole workgroups: every wave of an active workgroup owns a stash tile
    const int64_t off_counters = align_up(B.off, 256);
    P->counters = B.take<int>(16);
    // the histograms (+ tickets) of the backward's two counting sorts sit right behind the counters: the forward's one fill zeroes all
    // three (the scans re-zero the histograms after every use, so the backward needs no fill of its own)
    P->tiles = train && tile_geom(f, &P->tw.g) == 0 && use_tiles() && !det_mode();
    P->tw.hist = P->tw2.hist = nullptr;
    if (P->tiles) { P->tw2.g = P->tw.g; P->tw.hist = B.take<int>(P->tw.g.nbins + 64); P->tw2.hist = B.take<int>(P->tw.g.nbins + 64); }
    // ... and so do the look-back words of the two fused compactions (k_sample_fill, k_weights_fill): one per workgroup of 4 rays
    const int64_t ray_wgs = (R + 3) / 4;
    P->lb_s = B.take<unsigned long long>(ray_wgs); P->lb_w = B.take<unsigned long long>(ray_wgs);
    P->mse_part = B.take<float>(ray_wgs);        // nvfi_render_fwd_mse: per-workgroup partial sums of k_final_fwd (ticket: counters[8])
    P->zero_bytes = align_up(B.off, 256) - off_counters;
    P->sched = B.take<float>(SCHED_FLOATS);
    P->cnt_v = B.take<int>(R); P->off_v = B.take<int>(R + 1);
    P->cnt_m = B.take<int>(R); P->off_m = B.take<int>(R + 1);
    P->vlist = B.take<int>(N); P->mlist = B.take<int>(N);
    P->valid = B.take<uint8_t>(N); P->mflag = B.take<uint8_t>(N);
    P->cnt_r = P->off_r = P->rlist = nullptr; P->rflag = nullptr;
    if (nsteps > 0) { P->cnt_r = B.take<int>(R); P->off_r = B.take<int>(R + 1); P->rlist = B.take<int>(N); P->rflag = B.take<uint8_t>(N); }
    P->xw = B.take<float4>(N + 1); P->rgbs = B.take<float4>(N); P->rgb_pre = B.take<float4>(R);
    P->xpre = B.take<float>(N);
    P->vel_frag = B.take<float>(VEL_FRAG_FLOATS);
    P->vel_x4 = nsteps > 0 ? B.take<float>(VEL_X4F_FLOATS) : nullptr;
    P->img16 = (nsteps > 0 && ((!train && (f->vel_fp16 & 3)) || (train && (f->vel_fp16 & 4)))) ? (void*)B.take<float4>(2 * PRE16_IMAGE_BYTES / 16) : nullptr;   // fp16 images (hi, lo)
    P->vel_x4b = (nsteps > 0 && train) ? B.take<float>(VEL_X4B_FLOATS) : nullptr;
    P->x6img = nsteps > 0 ? (void*)B.take<float>(X6_IMAGE_BYTES / 4) : nullptr;      // the x6 images when the descriptor carries no fragment cache
    P->x6imgT = (nsteps > 0 && train) ? (void*)B.take<float>(X6_IMAGE_BYTES / 4) : nullptr;      // ... and their transposes (the adjoint's dgrad, vel_fuse.hip)
    P->render_frag = B.take<float>(RENDER_FRAG_FLOATS);
    P->maskv = (flags & NVFI_WANT_MASK) ? B.take<float>(N * 32) : nullptr;
    P->mask_frag = (flags & NVFI_WANT_MASK) ? B.take<float>(64 * 1024) : nullptr;
    P->app_relu = nullptr;
    P->app_f = P->app_b = P->zst = P->x0st = P->rec = P->gst = P->slabs = nullptr;
    P->gxw = P->gxk = nullptr; P->gxpre = nullptr; P->gg = nullptr;
    if (train) {
        P->gxw = B.take<float4>(N); P->gxk = B.take<float4>(N); P->gxpre = B.take<float>(N);
        P->gg = B.take<float>(N * 48);
        P->app_f = B.take<float>(P->cap_tiles * (int64_t)(APP_F_ROWS * REGF));
        P->app_b = B.take<float>(P->cap_tiles * (int64_t)(APP_B_ROWS * REGF));
        P->app_relu = B.take<unsigned>(P->cap_tiles * (int64_t)256);
        P->slabs = B.take<float>((int64_t)NSLAB_MAX * SLAB_FLOATS * 6);
        P->shadow = nullptr;
        if (det_mode()) { int64_t off[12]; P->shadow = B.take<long long>(plane_elems(f, off)); }
        if (P->tiles) plan_tile_scatter(B, f, N, &P->tw);
        if (P->tiles) plan_tile_scatter(B, f, N, &P->tw2);     // NVFI_BWD_FORK: the density half of the backward sorts / scatters beside the appearance half
        P->slabs2 = nsteps > 0 ? B.take<float>((int64_t)NSLAB_MAX * SLAB_FLOATS * 6) : nullptr;   // ... and the velocity-net slabs beside the render-MLP slabs
        if (nsteps > 0) {
            const int64_t nev = 2 * (int64_t)nsteps;
            P->zst = B.take<float>(nev * P->cap_tiles * (int64_t)(VEL_Z_REGS * REGF));
            P->x0st = B.take<float>(nev * P->cap_tiles * (int64_t)(VEL_X0_REGS * REGF));
            P->gst = B.take<float>(nev * P->cap_tiles * (int64_t)(VEL_G_REGS * REGF));
            P->rec = B.take<float>((int64_t)nsteps * RK_NF * N);
        }
    }
    P->total = align_up(B.off, 256);
}

extern "C" int nvfi_render_workspace_bytes(const nvfi_field_desc* f, int64_t R, int flags, int64_t* bytes) {
    if (check_desc(f)) return 2;
    // t-independent upper bound: a training call may need up to 2 RK2 steps (|t-base| <= dt_max up to
    // rounding); extrapolated times need more and are re-planned by the caller via nvfi_render_workspace_bytes_t.
    RenderPlan P;
    plan_render(f, R, flags, (f->use_vel && (flags & NVFI_TRAIN)) ? 2 : 0, nullptr, &P);
    *bytes = P.total;
    return 0;
}
extern "C" int nvfi_render_workspace_bytes_t(const nvfi_field_desc* f, int64_t R, int flags, float t, int64_t* bytes) {
    if (check_desc(f)) return 2;
    float base, dts[MAX_RK_STEPS], tcs[MAX_RK_STEPS];
    int ns = rk_schedule(f, t, flags, &base, dts, tcs);
    if (ns < 0) return nvfi_fail(2, "t=%g needs more than %d RK2 steps", t, MAX_RK_STEPS);
    RenderPlan P;
    plan_render(f, R, flags, ns, nullptr, &P);
    *bytes = P.total;
    return 0;
}

static int ensure_render_attrs() {
    static bool done = false;
    if (done) return 0;
    HIPCK(hipFuncSetAttribute((const void*)k_app_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_app_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_app_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    if (ensure_scatter_attrs()) return 1;
    done = true;
    return 0;
}

extern "C" int nvfi_render_fwd(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d,
                               const float* jitter, float t, int flags, float* rgb, float* depth, float* acc,
                               float* weights, void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream) {
    return nvfi_render_fwd_t(f, R, rays_o, rays_d, jitter, t, nullptr, flags, rgb, depth, acc, weights, workspace, workspace_bytes, counters, stream);
}

static int render_fwd_impl(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d,
                           const float* jitter, float t, const float* t_dev, int flags, float* rgb, float* depth, float* acc,
                           float* weights, void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream,
                           const float* target, float loss_scale, float* loss, float* g_rgb);
extern "C" int nvfi_render_fwd_t(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d,
                                 const float* jitter, float t, const float* t_dev, int flags, float* rgb, float* depth, float* acc,
                                 float* weights, void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream) {
    return render_fwd_impl(f, R, rays_o, rays_d, jitter, t, t_dev, flags, rgb, depth, acc, weights, workspace, workspace_bytes, counters, stream, nullptr, 1.f, nullptr, nullptr);
}
extern "C" int nvfi_render_fwd_mse(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d,
                                   const float* jitter, float t, const float* t_dev, int flags, float* rgb, float* depth, float* acc,
                                   float* weights, void* workspace, int64_t workspace_bytes, int64_t* counters,
                                   const float* target, float loss_scale, float* loss, float* g_rgb, void* stream) {
    if (!target || !loss || !g_rgb) return nvfi_fail(2, "nvfi_render_fwd_mse: target, loss and g_rgb must be non-NULL");
    return render_fwd_impl(f, R, rays_o, rays_d, jitter, t, t_dev, flags, rgb, depth, acc, weights, workspace, workspace_bytes, counters, stream, target, loss_scale, loss, g_rgb);
}
static int render_fwd_impl(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d,
                           const float* jitter, float t, const float* t_dev, int flags, float* rgb, float* depth, float* acc,
                           float* weights, void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream,
                           const float* target, float loss_scale, float* loss, float* g_rgb) {
    hipStream_t st = (hipStream_t)stream;
    if (check_desc(f)) return 2;
    if (R <= 0) return 0;
    if (R * (int64_t)f->n_samples >= (1ll << 31) - 64) return nvfi_fail(2, "R*S too large for one call; chunk the rays");
    if (ensure_render_attrs() || ensure_lds_attrs()) return 1;
    const bool train = flags & NVFI_TRAIN;
    float base, dts[MAX_RK_STEPS], tcs[MAX_RK_STEPS];
    const int nsteps = rk_schedule(f, t, flags, &base, dts, tcs);
    if (nsteps < 0) return nvfi_fail(2, "t=%g needs more than %d RK2 steps", t, MAX_RK_STEPS);
    RenderPlan P;
    plan_render(f, R, flags, nsteps, workspace, &P);
    if (P.total > workspace_bytes) return nvfi_fail(4, "workspace too small: need %lld bytes, got %lld", (long long)P.total, (long long)workspace_bytes);
    const int S = f->n_samples;
    const int64_t N = P.N;
    const float tn = f->use_vel ? norm_time(*f, base) : norm_time(*f, t);
    const bool fl = fused_launch();
    const bool pro = fl && R <= PROLOGUE_MAX_RAYS;      // one-workgroup prologue: clear + schedule + origin test
    const float* sched = nullptr;
    if (t_dev && nsteps > 4) return nvfi_fail(2, "a device-side time supports plans of up to 4 RK2 steps (t=%g needs %d)", t, nsteps);
    if (pro) {
        PrologueArgs pa; memset(&pa, 0, sizeof(pa));
        pa.sc.f = *f; pa.R = R; pa.o = rays_o; pa.zero_from = P.counters; pa.zero_ints = P.zero_bytes / 4; pa.inside = P.counters + 2;
        if (t_dev) {
            pa.do_sched = 1;
            pa.sc.t_dev = t_dev; pa.sc.flags = flags; pa.sc.nsteps_plan = nsteps; pa.sc.tn_plan = tn; pa.sc.sched = P.sched;
            for (int s = 0; s < nsteps; ++s) { pa.sc.dt_plan[s] = dts[s]; pa.sc.tc_plan[s] = tcs[s]; }
            sched = P.sched;
        }
        hipLaunchKernelGGL(k_prologue, dim3(1), dim3(256), 0, st, pa);
    } else {
        if (launch_zero(P.counters, P.zero_bytes, st)) return 1;
        if (t_dev) {
            SchedArgs sc; memset(&sc, 0, sizeof(sc));
            sc.f = *f; sc.t_dev = t_dev; sc.flags = flags; sc.nsteps_plan = nsteps; sc.tn_plan = tn; sc.sched = P.sched;
            for (int s = 0; s < nsteps; ++s) { sc.dt_plan[s] = dts[s]; sc.tc_plan[s] = tcs[s]; }
            hipLaunchKernelGGL(k_sched, dim3(1), dim3(64), 0, st, sc);
            sched = P.sched;
        }
    }
    // fragments (weights change every optimiser step: repack per call, ~0.3 MB)
    // (round 5: or not at all - a descriptor that carries the field's fragment cache, nvfi_pack_frags, points the kernels at it)
    PackJobs jobs; jobs.n = 0;
    VelFrags VW; RenderFrags RW;
    FragCache FC; const bool cached = f->frags != nullptr;
    if (cached) frag_cache_layout(f->frags, &FC);
    if (f->use_vel && nsteps > 0) { if (pack_vel_frags(f->vW, f->vb, cached ? FC.vel : P.vel_frag, &VW, &jobs)) return 3; }
    if (pack_render_frags(f, cached ? FC.render : P.render_frag, &RW, &jobs)) return 3;
    if (!cached && launch_pack(jobs, st)) return 1;
    const unsigned ray_blocks = (unsigned)((R + 3) / 4);
    // sampling
    if (!pro) hipLaunchKernelGGL(k_any_inside, dim3(64), dim3(256), 0, st, *f, R, rays_o, P.counters + 2);
    SampleArgs sa; memset(&sa, 0, sizeof(sa));
    sa.f = *f; sa.R = R; sa.o = rays_o; sa.d = rays_d; sa.u = jitter; sa.train = train; sa.inside = P.counters + 2;
    sa.xw = P.xw; sa.xpre = P.xpre; sa.valid = P.valid; sa.cnt = P.cnt_v; sa.rflag = P.rflag; sa.cnt_r = P.cnt_r;
    if (fl) {
        sa.lb = P.lb_s; sa.vlist = P.vlist; sa.rlist = P.rlist; sa.total_v = P.counters + 0; sa.total_r = P.counters + 3;
        hipLaunchKernelGGL(k_sample_fill, dim3(ray_blocks), dim3(256), 0, st, sa);
    } else {
        hipLaunchKernelGGL(k_sample, dim3(ray_blocks), dim3(256), 0, st, sa);
        hipLaunchKernelGGL(k_fill, dim3(ray_blocks), dim3(256), 0, st, R, S, P.valid, P.cnt_v, P.off_v, P.vlist, P.counters + 0);
        if (nsteps > 0) {   // second compact list: the valid samples inside the velocity gate (counters[3])
            hipLaunchKernelGGL(k_fill, dim3(ray_blocks), dim3(256), 0, st, R, S, P.rflag, P.cnt_r, P.off_r, P.rlist, P.counters + 3);
        }
    }
    LAUNCHCK();
    // velocity warp back to the keyframe
    if (nsteps > 0) {
        Rk2Args ra; memset(&ra, 0, sizeof(ra));
        ra.f = *f; ra.Wv = VW; ra.count = P.counters + 3; ra.list = P.rlist; ra.xw = P.xw; ra.xout = nullptr;
        ra.nsteps = nsteps; ra.sched = sched;
        for (int s = 0; s < nsteps; ++s) { ra.dt[s] = dts[s]; ra.tcur[s] = tcs[s]; }
        ra.zst = P.zst; ra.x0st = P.x0st; ra.rec = P.rec; ra.gst = P.gst; ra.cap = N; ra.cap_tiles = P.cap_tiles;
        // fp32 form: the feature-split layout of vel_split.hip (the one-tile-per-wave k_rk2_fwd<uniform> of vel.hip, NVFI_RK2_SPLIT=0 - same stash, same
        // numbers bit for bit - was retired in round 6)
        // round 5: the warp on the x6 evaluation (vel_x6.hip: the hidden layers' fp32 products formed exactly from three bfloat16 terms per operand
        // on the 16-bit matrix pipe; same stash / records for the fp32 adjoint) unless NVFI_RK2_X6=0 or an fp16-input mode is asked for
        const int vf = f->vel_fp16 & 3;
        if ((warp_x6_on() && !(train && (f->vel_fp16 & 4)) && (train || vf == 0 || vf == 3)) || (!train && vf == 3)) {
            ra.z_x4 = (train && warp_stash_x4(f)) ? 1 : 0;
            X6UniArgs xa; xa.r = ra; xa.img = cached ? FC.vel_x6 : P.x6img;
            if (!cached && launch_pack_x6(f->vW, P.x6img, st)) return 1;
            if (launch_rk2_x6_uni(xa, N, train, st)) return 1;
        } else if (((f->vel_fp16 & 3) && !train) || ((f->vel_fp16 & 4) && train)) {
            // opt-in fp16-input modes (pre16.hip): eval-mode renders (bits 0-1), and - bit 2 - the FORWARD of a training render's warp, which
            // writes the same stash as k_rk2_split_uni<STASH> (the adjoint and the weight gradients stay fp32 MFMA)
            Rk16Args h; memset(&h, 0, sizeof(h));
            h.img = P.img16; h.P = N; h.count = P.counters + 3; h.list = P.rlist; h.xw = P.xw; h.xout = P.xw; h.nsteps = nsteps; h.sched = sched;
            h.zst = P.zst; h.x0st = P.x0st; h.rec = P.rec; h.cap = N; h.cap_tiles = P.cap_tiles;
            for (int s = 0; s < nsteps; ++s) { h.dt[s] = dts[s]; h.tcur[s] = tcs[s]; }
            if (launch_rk2_inf16(f, h, true, st, train)) return 1;
        } else {
            SplitUniArgs ua; ua.r = ra;
            if (cached) x4f_pointers(FC.vel_x4f, ua.f4);
            else if (pack_vel_x4_fwd(VW, P.vel_x4, ua.f4, st)) return 1;
            for (int l = 0; l < 6; ++l) ua.bv[l] = VW.b[l];
            if (launch_rk2_split_uni(ua, N, train, st)) return 1;
        }
    }
    // density
    DensityArgs da; memset(&da, 0, sizeof(da));
    da.f = *f; da.count = P.counters + 0; da.list = P.vlist; da.xw = P.xw; da.xpre = P.xpre; da.tn = tn; da.sched = sched;
    { ProfScope ps(PK_DENSITY_FWD, st); if (launch_density_q(da, N, st)) return 1; }
    // weights
    WeightArgs wa; memset(&wa, 0, sizeof(wa));
    wa.R = R; wa.S = S; wa.xpre = P.xpre; wa.xw = P.xw; wa.distance_scale = f->distance_scale; wa.weight_thres = f->weight_thres;
    wa.far_ = f->far_; wa.weight = weights; wa.mflag = P.mflag; wa.acc = acc; wa.depth = depth; wa.cnt_m = P.cnt_m;
    if (fl) {
        wa.lb = P.lb_w; wa.off_m_out = P.off_m; wa.mlist = P.mlist; wa.total_m = P.counters + 1;
        hipLaunchKernelGGL(k_weights_fill, dim3(ray_blocks), dim3(256), 0, st, wa);
    } else {
        hipLaunchKernelGGL(k_weights_fwd, dim3(ray_blocks), dim3(256), 0, st, wa);
        hipLaunchKernelGGL(k_fill, dim3(ray_blocks), dim3(256), 0, st, R, S, P.mflag, P.cnt_m, P.off_m, P.mlist, P.counters + 1);
    }
    LAUNCHCK();
    // appearance
    AppArgs aa; memset(&aa, 0, sizeof(aa));
    aa.f = *f; aa.W = RW; aa.count = P.counters + 1; aa.list = P.mlist; aa.xw = P.xw; aa.tn = tn; aa.S = S; aa.sched = sched;
    aa.rays_d = rays_d; aa.rgbs = P.rgbs; aa.stash_f = P.app_f; aa.relu_mask = P.app_relu;
    const unsigned app_wgs = (unsigned)((N + WG_SAMPLES - 1) / WG_SAMPLES);
    {
        ProfScope ps(PK_APP_FWD, st);
        // train: the plane-product features of the masked samples from their own gather kernel (parked in gg, which only the backward writes)
        if (train && f->Ca == 48) {
            OgArgs oa; memset(&oa, 0, sizeof(oa));
            oa.f = *f; oa.count = P.counters + 1; oa.list = P.mlist; oa.xw = P.xw; oa.tn = tn; oa.sched = sched; oa.og = P.gg;
            if (launch_app_feat(oa, N, st)) return 1;
            aa.feat48 = P.gg;
        }
        if (launch_app_fwd(aa, N, train, st)) return 1;
    }
    // composite
    FinalArgs fa; fa.R = R; fa.off_m = P.off_m; fa.mlist = P.mlist; fa.weight = weights; fa.rgbs = P.rgbs; fa.acc = acc;
    fa.white_bg = (flags & NVFI_WHITE_BG) ? 1 : 0; fa.rgb_pre = P.rgb_pre; fa.rgb = rgb;
    fa.c = P.counters; fa.nsteps = nsteps; fa.counters_out = fl ? counters : nullptr; fa.sched = sched;
    fa.target = target; fa.g_rgb_out = g_rgb; fa.loss_out = loss; fa.partial = P.mse_part; fa.ticket = P.counters + 8; fa.loss_scale = loss_scale;
    hipLaunchKernelGGL(k_final_fwd, dim3(ray_blocks), dim3(256), 0, st, fa);
    LAUNCHCK();
    if (counters && !fl) {
        hipLaunchKernelGGL(k_counters, dim3(1), dim3(64), 0, st, P.counters, nsteps, counters, sched);
        LAUNCHCK();
    }
    return 0;
}

extern "C" int nvfi_render_bwd(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d, float t,
                               int flags, const float* weights, const float* g_rgb, const float* g_depth,
                               const float* g_acc, const float* g_weights, const nvfi_grads* grads, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    return nvfi_render_bwd_t(f, R, rays_o, rays_d, t, 0, flags, weights, g_rgb, g_depth, g_acc, g_weights, grads, workspace, workspace_bytes, stream);
}

extern "C" int nvfi_render_bwd_t(const nvfi_field_desc* f, int64_t R, const float* rays_o, const float* rays_d, float t, int t_on_device,
                                 int flags, const float* weights, const float* g_rgb, const float* g_depth,
                                 const float* g_acc, const float* g_weights, const nvfi_grads* grads, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (check_desc(f)) return 2;
    if (R <= 0) return 0;
    if (!(flags & NVFI_TRAIN)) return nvfi_fail(2, "nvfi_render_bwd needs the workspace of a NVFI_TRAIN forward");
    if (ensure_render_attrs() || ensure_lds_attrs()) return 1;
    float base, dts[MAX_RK_STEPS], tcs[MAX_RK_STEPS];
    const int nsteps = rk_schedule(f, t, flags, &base, dts, tcs);
    RenderPlan P;
    plan_render(f, R, flags, nsteps, workspace, &P);
    if (P.total > workspace_bytes) return nvfi_fail(4, "workspace too small");
    const int S = f->n_samples;
    const int64_t N = P.N;
    const float tn = f->use_vel ? norm_time(*f, base) : norm_time(*f, t);
    const float* sched = t_on_device ? P.sched : nullptr;     // the record the forward's k_sched left in the workspace
    const unsigned ray_blocks = (unsigned)((R + 3) / 4);
    const bool side = g_side.get() != 0 && !P.tiles && !det_mode();   // the tile scatter reuses one og buffer for both branches: same stream
    const bool want_aplanes0 = grads->aps[0] || grads->apt[0], want_dplanes0 = grads->dps[0] || grads->dpt[0];
    const bool forkable = (flags & NVFI_BWD_FORK) && P.tiles && want_aplanes0 && want_dplanes0 && !det_mode() && g_fork.get() != 0;
    const bool fork = forkable && nsteps == 0, fork2 = forkable && nsteps > 0;
    hipStream_t sd = st;                                     // stream of k_weights_bwd / k_og<24> (keyframe fork: the side stream)
    hipStream_t s_atail = st, s_dtail = st;                  // streams of the appearance tail (scatter, render-MLP weight gradients) and of the density scatter
    if (fork) { sd = g_fork.s; s_dtail = g_fork.s; }
    if (fork2) { s_atail = g_fork.s; s_dtail = g_fork.s; }
    // round 5: both counting sorts of the backward (masked list -> P.tw, valid list -> P.tw2) in ONE pair of launches, up front - the lists and the
    // positions are the forward's; the density branch then always sorts into tw2 and keeps sharing the og buffer unless it runs on the side stream
    const bool fl = fused_launch();
    const bool want_asort = P.tiles && want_aplanes0, want_dsort = P.tiles && want_dplanes0;
    const bool presort = fl && (want_asort || want_dsort);
    TileWork twd_v = (fork || fork2 || presort) ? P.tw2 : P.tw;
    if (presort && !(fork || fork2)) twd_v.og = P.tw.og;
    const TileWork& twd = twd_v;
    if (presort) {
        const TileWork* wp[2]; const int* cp[2]; const int* lp[2]; int nj = 0;
        if (want_asort) { wp[nj] = &P.tw; cp[nj] = P.counters + 1; lp[nj] = P.mlist; ++nj; }
        if (want_dsort) { wp[nj] = &twd; cp[nj] = P.counters + 0; lp[nj] = P.vlist; ++nj; }
        ProfScope ps(PK_DENSITY_SCATTER, st);
        if (launch_tile_sort(wp, cp, lp, nj, P.xw, N, st)) return 1;
    }
    if (fork) { HIPCK(hipEventRecord(g_fork.fork, st)); HIPCK(hipStreamWaitEvent(g_fork.s, g_fork.fork, 0)); }     // (behind the sorts)
    // deterministic mode: the scatters add fixed-point integers into int64 shadow planes; k_det_finish folds them into the gradients
    nvfi_grads gdet = *grads;
    int64_t det_off[12]; int64_t det_n = 0;
    if (det_mode()) {
        det_n = plane_elems(f, det_off);
        if (launch_zero(P.shadow, det_n * (int64_t)sizeof(long long), st)) return 1;
        for (int i = 0; i < 3; ++i) {
            if (gdet.dps[i]) gdet.dps[i] = reinterpret_cast<float*>(P.shadow + det_off[i]);
            if (gdet.dpt[i]) gdet.dpt[i] = reinterpret_cast<float*>(P.shadow + det_off[3 + i]);
            if (gdet.aps[i]) gdet.aps[i] = reinterpret_cast<float*>(P.shadow + det_off[6 + i]);
            if (gdet.apt[i]) gdet.apt[i] = reinterpret_cast<float*>(P.shadow + det_off[9 + i]);
        }
    }
    // fragments were packed by the forward into the same workspace
    VelFrags VW; RenderFrags RW; PackJobs dummy; dummy.n = 0;
    FragCache FC; const bool cached = f->frags != nullptr;      // (... or live in the field's fragment cache: the same weights, the caller's contract)
    if (cached) frag_cache_layout(f->frags, &FC);
    if (f->use_vel && nsteps > 0) pack_vel_frags(f->vW, f->vb, cached ? FC.vel : P.vel_frag, &VW, &dummy);
    dummy.n = 0;
    pack_render_frags(f, cached ? FC.render : P.render_frag, &RW, &dummy);
    bool forked = false;
    // appearance branch
    AppArgs aa; memset(&aa, 0, sizeof(aa));
    aa.f = *f; aa.W = RW; aa.count = P.counters + 1; aa.list = P.mlist; aa.xw = P.xw; aa.tn = tn; aa.S = S; aa.sched = sched;
    aa.rays_d = rays_d; aa.rgbs = P.rgbs; aa.stash_f = P.app_f; aa.relu_mask = P.app_relu; aa.stash_b = P.app_b; aa.g = *grads;
    aa.g_rgb = g_rgb; aa.rgb_pre = P.rgb_pre; aa.weight = weights; aa.gxw = P.gxw; aa.gg = P.gg;
    aa.plane_tail = P.tiles ? 0 : 1;
    const unsigned app_wgs = (unsigned)((N + WG_SAMPLES - 1) / WG_SAMPLES);
    { ProfScope ps(PK_APP_BWD, st); hipLaunchKernelGGL(k_app_bwd, dim3(app_wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, aa); }
    const bool want_aplanes = grads->aps[0] || grads->apt[0];
    if (P.tiles) {
        if (want_aplanes || nsteps > 0) {
            // one pass over the masked samples: per-plane value gradients (og) for the tile scatter and, at non-keyframe times,
            // the plane part of the coordinate gradients (they only feed the RK2 adjoint)
            ProfScope ps(PK_APP_SCATTER, st);
            OgArgs oa; memset(&oa, 0, sizeof(oa));
            oa.f = *f; oa.count = P.counters + 1; oa.list = P.mlist; oa.xw = P.xw; oa.tn = tn; oa.sched = sched; oa.gg = P.gg; oa.og = want_aplanes ? P.tw.og : nullptr;
            oa.gxw_acc = nsteps > 0 ? P.gxw : nullptr;
            if (launch_og(f, oa, 48, nsteps > 0, N, st)) return 1;
            if (fork2) { HIPCK(hipEventRecord(g_fork.fork, st)); HIPCK(hipStreamWaitEvent(g_fork.s, g_fork.fork, 0)); }
            if (want_aplanes) {
                if (launch_tile_scatter(f, P.tw, P.counters + 1, P.mlist, P.xw, tn, *grads, 48, N, s_atail, sched, presort)) return 1;
            }
        }
    } else if (want_aplanes) {
        ScatterArgs sa; memset(&sa, 0, sizeof(sa));
        sa.f = *f; sa.count = P.counters + 1; sa.list = P.mlist; sa.xw = P.xw; sa.tn = tn; sa.sched = sched; sa.gg = P.gg; sa.g = det_mode() ? gdet : *grads; sa.plane_mask = 63;
        hipStream_t ss = st;
        if (side) { HIPCK(hipEventRecord(g_side.fork[0], st)); HIPCK(hipStreamWaitEvent(g_side.s, g_side.fork[0], 0)); ss = g_side.s; forked = true; }
        ProfScope ps(PK_APP_SCATTER, ss);
        if (det_mode() ? launch_scatter_det(sa, 48, N, ss) : launch_scatter(f, sa, 48, N, tn, ss)) return 1;
    }
    LAUNCHCK();
    // render-MLP weight gradients
    // (round 5: at a non-keyframe time on one stream their ring / reduce jobs ride in the velocity net's two launches at the end of the call)
    WgradJobs mlp_wj; mlp_wj.n = 0; ReduceJobs mlp_rj; mlp_rj.n = 0;
    const bool merge_wgrad = fl && nsteps > 0 && !fork2;
    {
        WgradJobs& wj = mlp_wj; ReduceJobs& rj = mlp_rj;
        const size_t fs = APP_F_ROWS * REGF, bs = APP_B_ROWS * REGF;
        auto add = [&](const float* A, int a_regs, const float* B, int b_regs, float* slabs, float* gW, float* gb, int out, int in, int sk) {
            WgradJob& J = wj.j[wj.n++];
            memset(&J, 0, sizeof(J));
            J.A = A; J.a_tile_stride = bs; J.a_regs = a_regs; J.B = B; J.B2 = nullptr; J.b_tile_stride = fs; J.b_regs = b_regs;
            J.bmode = BM_RAW; J.count = P.counters + 1; J.cap_tiles = (int)P.cap_tiles; J.nrep = 1; J.a_rep_stride = 0; J.b_rep_stride = 0;
            J.slabs = slabs; J.nslab = NSLAB;
            ReduceJob& Q = rj.j[rj.n++];
            memset(&Q, 0, sizeof(Q));
            Q.slabs = slabs; Q.nslab = NSLAB; Q.MTA = a_regs / 16; Q.KTB = b_regs / 16; Q.gW = gW; Q.gb = gb; Q.out = out; Q.in = in;
            Q.row_kind = RK_NATURAL; Q.slot_kind = sk; Q.scale = 1.f;
        };
        float* sl = P.slabs;
        if (grads->rW[2] || grads->rb[2]) add(P.app_b + 0, 16, P.app_f + 160 * REGF, 64, sl + 0 * (size_t)NSLAB_MAX * SLAB_FLOATS, grads->rW[2], grads->rb[2], 3, 128, SK_HIDDEN);
        if (grads->rW[1] || grads->rb[1]) add(P.app_b + 16 * REGF, 64, P.app_f + 96 * REGF, 64, sl + 1 * (size_t)NSLAB_MAX * SLAB_FLOATS, grads->rW[1], grads->rb[1], 128, 128, SK_HIDDEN);
        if (grads->rW[0] || grads->rb[0]) add(P.app_b + 80 * REGF, 64, P.app_f + 32 * REGF, 64, sl + 2 * (size_t)NSLAB_MAX * SLAB_FLOATS, grads->rW[0], grads->rb[0], 128, 110, SK_RENDER_IN);
        if (grads->basis) add(P.app_b + 144 * REGF, 16, P.app_f + 0, 32, sl + 3 * (size_t)NSLAB_MAX * SLAB_FLOATS, grads->basis, nullptr, f->app_dim, f->Ca, SK_HIDDEN);
        if (!merge_wgrad && launch_wgrad(wj, rj, s_atail)) return 1;
    }
    // composites + raw2alpha
    WeightArgs wa; memset(&wa, 0, sizeof(wa));
    wa.R = R; wa.S = S; wa.xpre = P.xpre; wa.xw = P.xw; wa.distance_scale = f->distance_scale; wa.weight_thres = f->weight_thres;
    wa.far_ = f->far_; wa.mflag = P.mflag; wa.off_m = P.off_m; wa.rgbs = P.rgbs; wa.rgb_pre = P.rgb_pre;
    wa.g_rgb = g_rgb; wa.g_depth = g_depth; wa.g_acc = g_acc; wa.g_weight = g_weights; wa.gxpre = P.gxpre;
    wa.white_bg = (flags & NVFI_WHITE_BG) ? 1 : 0;
    hipLaunchKernelGGL(k_weights_bwd, dim3(ray_blocks), dim3(256), 0, sd, wa);
    // density planes + coordinate grads
    DensityArgs da; memset(&da, 0, sizeof(da));
    da.f = *f; da.count = P.counters + 0; da.list = P.vlist; da.xw = P.xw; da.xpre = P.xpre; da.tn = tn; da.sched = sched;
    da.gxpre = P.gxpre; memset(&da.g, 0, sizeof(da.g)); da.mflag = P.mflag; da.gxw = P.gxw; da.gxk = nsteps > 0 ? P.gxk : nullptr;
    const bool want_dplanes = grads->dps[0] || grads->dpt[0];
    if (P.tiles) {
        // one pass over the samples: per-plane value gradients (og) for the tile scatter and, at non-keyframe times, the coordinate gradients
        if (nsteps > 0 || want_dplanes) {
            ProfScope ps(PK_DENSITY_BWD, sd);
            OgArgs oa; memset(&oa, 0, sizeof(oa));
            oa.f = *f; oa.count = P.counters + 0; oa.list = P.vlist; oa.xw = P.xw; oa.tn = tn; oa.sched = sched; oa.gxpre = P.gxpre; oa.og = want_dplanes ? twd.og : nullptr;
            oa.mflag = P.mflag; oa.gxw = P.gxw; oa.gxk = nsteps > 0 ? P.gxk : nullptr;
            if (launch_og(f, oa, 24, nsteps > 0, N, sd)) return 1;
        }
        if (fork2) { HIPCK(hipEventRecord(g_fork.fork2, st)); HIPCK(hipStreamWaitEvent(g_fork.s, g_fork.fork2, 0)); }
        if (want_dplanes) {
            ProfScope ps(PK_DENSITY_SCATTER, s_dtail);
            if (launch_tile_scatter(f, twd, P.counters + 0, P.vlist, P.xw, tn, *grads, 24, N, s_dtail, sched, presort)) return 1;
        }
        if (fork) { HIPCK(hipEventRecord(g_fork.join, g_fork.s)); HIPCK(hipStreamWaitEvent(st, g_fork.join, 0)); }
    } else if (nsteps > 0) { ProfScope ps(PK_DENSITY_BWD, st); hipLaunchKernelGGL(k_density_bwd, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, da); }
    if (!P.tiles && want_dplanes) {
        ScatterArgs sa; memset(&sa, 0, sizeof(sa));
        sa.f = *f; sa.count = P.counters + 0; sa.list = P.vlist; sa.xw = P.xw; sa.tn = tn; sa.sched = sched; sa.gxpre = P.gxpre; sa.g = det_mode() ? gdet : *grads; sa.plane_mask = 63;
        hipStream_t ss = st;
        if (side) { HIPCK(hipEventRecord(g_side.fork[1], st)); HIPCK(hipStreamWaitEvent(g_side.s, g_side.fork[1], 0)); ss = g_side.s; forked = true; }
        ProfScope ps(PK_DENSITY_SCATTER, ss);
        if (det_mode() ? launch_scatter_det(sa, 24, N, ss) : launch_scatter(f, sa, 24, N, tn, ss)) return 1;
    }
    if (det_mode()) {
        float* const real[12] = {grads->dps[0], grads->dps[1], grads->dps[2], grads->dpt[0], grads->dpt[1], grads->dpt[2],
                                 grads->aps[0], grads->aps[1], grads->aps[2], grads->apt[0], grads->apt[1], grads->apt[2]};
        for (int k = 0; k < 12; ++k) {
            if (!real[k]) continue;
            const int64_t n = (k + 1 < 12 ? det_off[k + 1] : det_n) - det_off[k];
            hipLaunchKernelGGL(k_det_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P.shadow + det_off[k], real[k], n);
        }
        LAUNCHCK();
    }
    LAUNCHCK();
    // RK2 adjoint + velocity-net weight gradients
    if (nsteps > 0) {
        Rk2Args ra; memset(&ra, 0, sizeof(ra));
        ra.f = *f; ra.Wv = VW; ra.count = P.counters + 3; ra.list = P.rlist; ra.xw = P.xw;
        ra.nsteps = nsteps; ra.sched = sched;
        for (int s = 0; s < nsteps; ++s) { ra.dt[s] = dts[s]; ra.tcur[s] = tcs[s]; }
        ra.zst = P.zst; ra.x0st = P.x0st; ra.rec = P.rec; ra.gst = P.gst; ra.cap = N; ra.cap_tiles = P.cap_tiles; ra.gxk = P.gxk;
        // NVFI_RK2_FUSE (default 1): vel_fuse.hip - the adjoint AND the four 128 x 128 weight gradients in one persistent kernel (no g_1..g_4
        // stash, no second pass over the z stash); 0: k_rk2_split_bwd + k_wgrad_ring8 over the full adjoint stash
        const int fuse = rk2_fuse_on() ? 1 : 0;
        ra.z_x4 = warp_stash_x4(f) ? 1 : 0;                              // (what the forward wrote: the same predicate)
        float* vslabs = (fork2 || merge_wgrad) ? P.slabs2 : P.slabs;      // (merged launches: the render MLP's slabs in P.slabs are still live)
        int fused_nslab = 0;
        if (fuse) {
            FuseBwdArgs fa; memset(&fa, 0, sizeof(fa));
            fa.r = ra; fa.slabs = vslabs; fa.layer_stride = (int64_t)NSLAB * SLAB_FLOATS; fa.slab_floats = SLAB_FLOATS;
            if (cached) { x4b_pointers(FC.vel_x4b, fa.t4); fa.imgT = FC.vel_x6t; }
            else {
                if (pack_vel_x4_bwd(VW, P.vel_x4b, fa.t4, st)) return 1;
                if (launch_pack_x6(f->vW, P.x6img, st, P.x6imgT)) return 1;
                fa.imgT = P.x6imgT;
            }
            if (launch_rk2_fuse_bwd(fa, N, NSLAB, &fused_nslab, st)) return 1;
        } else {              // vel_split.hip (k_rk2_bwd of vel.hip, NVFI_RK2_SPLIT_BWD=0 - the same adjoint stash bit for bit - was retired in round 6)
            SplitBwdArgs ba; ba.r = ra;
            if (cached) x4b_pointers(FC.vel_x4b, ba.t4);
            else if (pack_vel_x4_bwd(VW, P.vel_x4b, ba.t4, st)) return 1;
            if (launch_rk2_split_bwd(ba, N, st)) return 1;
        }
        if (launch_vel_wgrad(P.zst, P.x0st, P.gst, P.counters + 3, (int)P.cap_tiles, 2 * nsteps, BM_SILU, vslabs, NSLAB,
                             grads->vW, grads->vb, 1.f, st, fused_nslab, merge_wgrad ? &mlp_wj : nullptr, merge_wgrad ? &mlp_rj : nullptr)) return 1;
    }
    if (fork2) { HIPCK(hipEventRecord(g_fork.join, g_fork.s)); HIPCK(hipStreamWaitEvent(st, g_fork.join, 0)); }
    if (forked) { HIPCK(hipEventRecord(g_side.join, g_side.s)); HIPCK(hipStreamWaitEvent(st, g_side.join, 0)); }
    return 0;
}

// velocity-net weight gradients from the (z, x0, g) stashes of nrep evaluations
// fused_nslab > 0: the slabs of the four hidden layers were already written (fused_nslab of them each) by k_rk2_fuse_bwd - only the two
// edge layers are contracted here, the reduce covers all six
int launch_vel_wgrad(const float* zst, const float* x0st, const float* gst, const int* count, int cap_tiles, int nrep,
                     int act_mode, float* slabs, int nslab, float* const* gW, float* const* gb, float scale, hipStream_t st, int fused_nslab,
                     const WgradJobs* pre_w, const ReduceJobs* pre_r) {
    WgradJobs wj; wj.n = 0; ReduceJobs rj; rj.n = 0;
    if (pre_w) wj = *pre_w;          // jobs of the same call that share the two launches (the render MLP's, render.hip)
    if (pre_r) rj = *pre_r;
    const size_t zs = VEL_Z_REGS * REGF, gs = VEL_G_REGS * REGF, xs = VEL_X0_REGS * REGF;
    for (int l = 0; l < 6; ++l) {
        if (!gW[l] && !gb[l]) continue;
        if (fused_nslab > 0 && l >= 1 && l <= 4) {
            ReduceJob& Q = rj.j[rj.n++];
            memset(&Q, 0, sizeof(Q));
            Q.slabs = slabs + (size_t)l * nslab * SLAB_FLOATS; Q.nslab = fused_nslab; Q.MTA = 4; Q.KTB = 4; Q.gW = gW[l]; Q.gb = gb[l];
            Q.out = 128; Q.in = 128; Q.row_kind = RK_NATURAL; Q.slot_kind = SK_HIDDEN; Q.scale = scale;
            continue;
        }
        WgradJob& J = wj.j[wj.n++];
        memset(&J, 0, sizeof(J));
        J.A = gst + (size_t)l * 64 * REGF; J.a_tile_stride = gs; J.a_regs = l < 5 ? 64 : 16; J.a_rep_stride = (size_t)cap_tiles * gs;
        if (l == 0) { J.B = x0st; J.b_tile_stride = xs; J.b_regs = 16; J.bmode = BM_RAW; J.b_rep_stride = (size_t)cap_tiles * xs; }
        else { J.B = zst + (size_t)(l - 1) * 64 * REGF; J.b_tile_stride = zs; J.b_regs = 64; J.bmode = act_mode; J.b_rep_stride = (size_t)cap_tiles * zs; }
        J.B2 = nullptr; J.count = count; J.cap_tiles = cap_tiles; J.nrep = nrep;
        J.slabs = slabs + (size_t)l * nslab * SLAB_FLOATS; J.nslab = nslab;
        ReduceJob& Q = rj.j[rj.n++];
        memset(&Q, 0, sizeof(Q));
        Q.slabs = J.slabs; Q.nslab = nslab; Q.MTA = J.a_regs / 16; Q.KTB = J.b_regs / 16; Q.gW = gW[l]; Q.gb = gb[l];
        Q.out = l < 5 ? 128 : 6; Q.in = l == 0 ? 28 : 128; Q.row_kind = RK_NATURAL; Q.slot_kind = l == 0 ? SK_VEL_IN : SK_HIDDEN; Q.scale = scale;
    }
    return launch_wgrad(wj, rj, st);
}

__global__ void k_counters(const int* c, int nsteps, int64_t* out, const float* sched) {
    if (threadIdx.x == 0) counters_body(c, nsteps, out, sched);
}

// ================================================================ building blocks
extern "C" int nvfi_density_at(const nvfi_field_desc* f, int64_t N, const float* xyzt, float* feat, float* sigma, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (check_desc(f)) return 2;
    if (N <= 0) return 0;
    DensityArgs da; memset(&da, 0, sizeof(da));
    da.f = *f; da.count = nullptr; da.n_direct = N; da.list = nullptr; da.xw = reinterpret_cast<const float4*>(xyzt);
    da.per_point_t = 1; da.feat_out = feat; da.sigma_out = sigma; da.xpre = nullptr;
    return launch_density_q(da, N, st);
}

extern "C" int nvfi_app_workspace_bytes(const nvfi_field_desc* f, int64_t N, int64_t* bytes) {
    (void)f;
    Bump B{nullptr, 0, 0};
    B.take<float>(RENDER_FRAG_FLOATS); B.take<float4>(N > 0 ? N : 0);
    *bytes = align_up(B.off, 256);
    return 0;
}
extern "C" int nvfi_app_at(const nvfi_field_desc* f, int64_t N, const float* xyzt, const float* view, float* rgb,
                           void* workspace, int64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (check_desc(f)) return 2;
    if (N <= 0) return 0;
    if (ensure_render_attrs()) return 1;
    Bump B{(char*)workspace, 0, 0};
    float* frag = B.take<float>(RENDER_FRAG_FLOATS);
    float4* out4 = B.take<float4>(N);
    if (B.off > workspace_bytes) return nvfi_fail(4, "workspace too small: need %lld", (long long)B.off);
    PackJobs jobs; jobs.n = 0; RenderFrags RW;
    if (pack_render_frags(f, frag, &RW, &jobs)) return 3;
    if (launch_pack(jobs, st)) return 1;
    AppArgs aa; memset(&aa, 0, sizeof(aa));
    aa.f = *f; aa.W = RW; aa.count = nullptr; aa.n_direct = N; aa.list = nullptr; aa.xw = reinterpret_cast<const float4*>(xyzt);
    aa.per_point_t = 1; aa.S = 1; aa.view_per_point = view; aa.rgbs = out4; aa.rgb_dense = 1;
    if (launch_app_fwd(aa, N, false, st)) return 1;
    hipLaunchKernelGGL(k_unpack_rgb, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, out4, rgb, N);
    LAUNCHCK();
    return 0;
}
// renderModule(pts, viewdirs, features) as a stand-alone call (tensorf_base.py:88-98 / tensorf_model_utils.py:292-296): the appearance
// features are the caller's, only the positional encodings + MLP (or the SH epilogue) of k_app_fwd run.  xyz: (N,3) normalised positions.
extern "C" int nvfi_render_mlp(const nvfi_field_desc* f, int64_t N, const float* xyz, const float* view, const float* features, float* rgb,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (check_desc(f)) return 2;
    if (N <= 0) return 0;
    if (ensure_render_attrs()) return 1;
    Bump B{(char*)workspace, 0, 0};
    float* frag = B.take<float>(RENDER_FRAG_FLOATS);
    float4* out4 = B.take<float4>(N);
    float4* xw = B.take<float4>(N);
    if (B.off > workspace_bytes) return nvfi_fail(4, "workspace too small: need %lld", (long long)B.off);
    PackJobs jobs; jobs.n = 0; RenderFrags RW;
    if (pack_render_frags(f, frag, &RW, &jobs)) return 3;
    if (launch_pack(jobs, st)) return 1;
    hipLaunchKernelGGL(k_pack_xyz4, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, xyz, xw, N);
    AppArgs aa; memset(&aa, 0, sizeof(aa));
    aa.f = *f; aa.W = RW; aa.count = nullptr; aa.n_direct = N; aa.list = nullptr; aa.xw = xw;
    aa.per_point_t = 1; aa.S = 1; aa.view_per_point = view; aa.rgbs = out4; aa.rgb_dense = 1; aa.feat_in = features;
    if (launch_app_fwd(aa, N, false, st)) return 1;
    hipLaunchKernelGGL(k_unpack_rgb, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, out4, rgb, N);
    LAUNCHCK();
    return 0;
}
__global__ void k_pack_xyz4(const float* in, float4* out, int64_t N) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < N) out[i] = make_float4(in[3 * i], in[3 * i + 1], in[3 * i + 2], 0.f);
}
__global__ void k_unpack_rgb(const float4* in, float* out, int64_t N) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < N) { float4 v = in[i]; out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z; }
}


// ================================================================ a-19 mask branch (inference)
struct MaskFrags { const float* f[5]; const float* b[5]; };
struct MaskArgs {
    MaskFrags W; int mask_dim;
    const int* count; const int* list; const float4* xw;
    float* maskv;        // (M, 32) softmax outputs per masked sample
    int64_t R; const int* off_m; const float* weight; float* mask_map;
};

__global__ __launch_bounds__(WG_THREADS, 2) void k_mask_fwd(MaskArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_w = lds; float* lds_b = lds + LDS_W_FLOATS;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int count = *a.count;
    if ((int)(blockIdx.x * WG_SAMPLES) >= count) return;
    const int tile = blockIdx.x * 4 + wave_id();
    const int i = tile * TILE + (lane & 31);
    const bool active = i < count;
    const float4 q = active ? a.xw[a.list[i]] : zero4();
    float xa[64], xb[64];
    xb[0] = h ? q.y : q.x; xb[1] = h ? 0.f : q.z;
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[0], 4 * 2 * 64, a.W.b[0], 128);
    __syncthreads();
    layer_tiles<4, 2>(lds_w, lds_b, true, lane, h, xb, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xa[16 * m + r] = fmaxf(acc[r], 0.f);
    });
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[1], 4 * 64 * 64, a.W.b[1], 128);
    __syncthreads();
    layer_tiles<4, 64>(lds_w, lds_b, true, lane, h, xa, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xb[16 * m + r] = fmaxf(acc[r], 0.f);
    });
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[2], 4 * 64 * 64, a.W.b[2], 128);
    __syncthreads();
    layer_tiles<4, 64>(lds_w, lds_b, true, lane, h, xb, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xa[16 * m + r] = fmaxf(acc[r], 0.f);
    });
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[3], 4 * 64 * 64, a.W.b[3], 128);
    __syncthreads();
    layer_tiles<4, 64>(lds_w, lds_b, true, lane, h, xa, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xb[16 * m + r] = fmaxf(acc[r], 0.f);
    });
    __syncthreads();
    stage_frag(lds_w, lds_b, a.W.f[4], 1 * 64 * 64, a.W.b[4], 32);
    __syncthreads();
    float o[16];
    layer_tiles<1, 64>(lds_w, lds_b, true, lane, h, xb, [&](int, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = acc[r];
    });
    // softmax over the mask_dim logits of the sample: rows (r&3)+8(r>>2)+4h live in this lane, the rest in lane^32
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; if (row < a.mask_dim) mx = fmaxf(mx, o[r]); }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; o[r] = row < a.mask_dim ? expf(o[r] - mx) : 0.f; sum += o[r]; }
    sum += __shfl_xor(sum, 32);
    if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int row = (r & 3) + 8 * (r >> 2) + 4 * h; if (row < a.mask_dim) a.maskv[(size_t)i * 32 + row] = o[r] / sum; }
    }
}
// mask_map[r][k] = sum_j w_j mask_j[k] over the ray's masked samples (tensorf_keyframe.py:753)
__global__ __launch_bounds__(256) void k_mask_final(MaskArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.R) return;
    const int b0 = a.off_m[r], b1 = a.off_m[r + 1];
    // lane = (entry parity, channel): 2 entries per pass x 32 channels
    const int k = lane & 31, e0 = lane >> 5;
    float s = 0.f;
    for (int i = b0 + e0; i < b1; i += 2) s += a.weight[a.list[i]] * (k < a.mask_dim ? a.maskv[(size_t)i * 32 + k] : 0.f);
    s += __shfl_xor(s, 32);
    if (lane < a.mask_dim) a.mask_map[r * a.mask_dim + lane] = s;
}

extern "C" int nvfi_render_mask(const nvfi_field_desc* f, const nvfi_mask_desc* m, int64_t R, float t, int flags, const float* weights,
                                float* mask_map, void* workspace, int64_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (check_desc(f)) return 2;
    if (m->n_layer != 4 || m->n_dim != 128 || m->mask_dim < 1 || m->mask_dim > 32)
        return nvfi_fail(2, "mask field must be 3->128x4->mask_dim<=32 (train_segm.py:97-102); got n_layer=%d n_dim=%d mask_dim=%d", m->n_layer, m->n_dim, m->mask_dim);
    if (R <= 0) return 0;
    float base, dts[MAX_RK_STEPS], tcs[MAX_RK_STEPS];
    const int nsteps = rk_schedule(f, t, flags, &base, dts, tcs);
    RenderPlan P;
    plan_render(f, R, flags, nsteps < 0 ? 0 : nsteps, workspace, &P);
    if (P.total > workspace_bytes) return nvfi_fail(4, "workspace too small");
    static bool attr = false;
    if (!attr) { HIPCK(hipFuncSetAttribute((const void*)k_mask_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES)); attr = true; }
    if (!P.mask_frag) return nvfi_fail(2, "nvfi_render_mask needs a workspace planned with NVFI_WANT_MASK in flags");
    float* frag = P.mask_frag;
    PackJobs jobs; jobs.n = 0;
    MaskArgs a; memset(&a, 0, sizeof(a));
    float* p = frag;
    for (int l = 0; l < 5; ++l) {
        PackJob& J = jobs.j[jobs.n++];
        const int MT = l < 4 ? 4 : 1, NS = l == 0 ? 2 : 64;
        J.W = m->W[l]; J.b = m->b[l]; J.frag = p; p += MT * NS * 64; J.bfrag = p; p += 128;
        J.out = l < 4 ? 128 : m->mask_dim; J.in = l == 0 ? 3 : 128; J.MT = MT; J.NS = NS;
        J.row_kind = RK_NATURAL; J.slot_kind = l == 0 ? SK_XYZ : SK_HIDDEN; J.transposed = 0; J.x4 = 0;
        a.W.f[l] = J.frag; a.W.b[l] = J.bfrag;
    }
    if (launch_pack(jobs, st)) return 1;
    a.mask_dim = m->mask_dim; a.count = P.counters + 1; a.list = P.mlist; a.xw = P.xw;
    a.R = R; a.off_m = P.off_m; a.weight = weights; a.mask_map = mask_map;
    if (!P.maskv) return nvfi_fail(2, "nvfi_render_mask needs a workspace planned with NVFI_WANT_MASK in flags");
    a.maskv = P.maskv;
    const unsigned wgs = (unsigned)((P.N + WG_SAMPLES - 1) / WG_SAMPLES);
    hipLaunchKernelGGL(k_mask_fwd, dim3(wgs), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, a);
    hipLaunchKernelGGL(k_mask_final, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, a);
    LAUNCHCK();
    return 0;
}

// ---- appearance-masked samples of the call that filled `workspace`: warped keyframe position (xyz_out (M,3)) and dense sample index
//      r * S + j (idx_out (M)); at most `cap` entries are written (the true count is counters[2] of nvfi_render_fwd).  Lets the host
//      mirror build the DIFFERENTIABLE mask branch (tensorf_keyframe.py:749-753 in train mode) out of MaskField's own fwd/bwd kernels.
__global__ void k_export_masked(const int* count, int64_t cap, const int* list, const float4* xw, float* xyz, int64_t* idx) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t n = *count < cap ? *count : cap;
    if (i >= n) return;
    const int s = list[i];
    const float4 q = xw[s];
    xyz[3 * i] = q.x; xyz[3 * i + 1] = q.y; xyz[3 * i + 2] = q.z;
    idx[i] = s;
}
extern "C" int nvfi_render_export_masked(const nvfi_field_desc* f, int64_t R, float t, int flags, void* workspace, int64_t workspace_bytes,
                                         int64_t cap, float* xyz_out, int64_t* idx_out, void* stream) {
    if (check_desc(f)) return 2;
    if (R <= 0 || cap <= 0) return 0;
    float base, dts[MAX_RK_STEPS], tcs[MAX_RK_STEPS];
    const int nsteps = rk_schedule(f, t, flags, &base, dts, tcs);
    RenderPlan P;
    plan_render(f, R, flags, nsteps < 0 ? 0 : nsteps, workspace, &P);
    if (P.total > workspace_bytes) return nvfi_fail(4, "workspace too small");
    hipLaunchKernelGGL(k_export_masked, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P.counters + 1, cap, P.mlist, P.xw, xyz_out, idx_out);
    LAUNCHCK();
    return 0;
}

// ================================================================ a-17 SHRender (degree 2 real SH, relu(sum + 0.5))
__global__ void k_sh_render(int64_t N, const float* __restrict__ view, const float* __restrict__ ft, float* __restrict__ rgb) {
    const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float x = view[3 * n], y = view[3 * n + 1], z = view[3 * n + 2];
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    float b[9];
    b[0] = C0; b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
    b[4] = C2[0] * xy; b[5] = C2[1] * yz; b[6] = C2[2] * (2.0f * zz - xx - yy); b[7] = C2[3] * xz; b[8] = C2[4] * (xx - yy);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += b[k] * ft[27 * n + 9 * c + k];
        s += 0.5f;
        rgb[3 * n + c] = s > 0.f ? s : 0.f;
    }
}
extern "C" int nvfi_sh_render(int64_t N, const float* view, const float* feat27, float* rgb, void* stream) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(k_sh_render, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, view, feat27, rgb);
    LAUNCHCK();
    return 0;
}
