// common.h - shared device helpers for the NVFi hot-path kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/nvfi_hip.h"
#include "engine.h"

// ---------------------------------------------------------------- error plumbing (host)
int nvfi_fail(int code, const char* fmt, ...);
#define HIPCK(expr)                                                                              \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) return nvfi_fail(100 + (int)_e, "%s failed: %s (%s:%d)", #expr,   \
                                               hipGetErrorString(_e), __FILE__, __LINE__);       \
    } while (0)
#define LAUNCHCK() HIPCK(hipGetLastError())

// per-device one-time set-up of a launcher (hipFuncSetAttribute applies to the CURRENT device): `apply` runs under a lock until it has
// succeeded once on the device - a failed or racing first call leaves nothing marked as done
#ifdef __cplusplus
#include <mutex>
struct DeviceOnce {
    std::mutex mu;
    bool done[64] = {false};
    template <class F> int run(F apply) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        std::lock_guard<std::mutex> g(mu);
        if (done[dev]) return 0;
        const int rc = apply();
        if (rc == 0) done[dev] = true;
        return rc;
    }
};
#endif

// optional per-kernel-class HIP-event timing (bench.py): events are recorded on the launch stream
enum { PK_RK2_FWD = 0, PK_RK2_BWD, PK_APP_FWD, PK_APP_BWD, PK_WGRAD, PK_PDE_FWD, PK_PDE_BWD, PK_DENSITY_FWD, PK_DENSITY_BWD,
       PK_PDE_PREFILTER, PK_DENSITY_SCATTER, PK_APP_SCATTER, PK_OTHER, PK_COUNT };
void prof_begin(int cls, hipStream_t st);
void prof_end(int cls, hipStream_t st);
struct ProfScope { int c; hipStream_t s; ProfScope(int cls, hipStream_t st) : c(cls), s(st) { prof_begin(c, s); } ~ProfScope() { prof_end(c, s); } };

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// simple bump allocator over the caller's workspace
struct Bump {
    char* base; int64_t off; int64_t cap;
    template <typename T> T* take(int64_t n) {
        off = align_up(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * (int64_t)sizeof(T);
        return p;
    }
};

// ---------------------------------------------------------------- device-side time schedule
// A render normally gets its frame time as a host scalar, from which the host derives the keyframe time row (tn), the RK2 step
// sequence (dt, t) and passes them by value.  For hipGraph replay the same quantities can instead be produced ON THE DEVICE from a
// time held in device memory (k_sched, render.hip) into a small record in the call's workspace; every kernel argument block carries
// an optional pointer to that record and prefers it over its by-value copy.  Layout (floats):
//   [0] tn   [1] y0 (int bits: first time row of the LDS scatter variants)   [2] nsteps (int bits)   [3] mismatch flag (int bits)
//   [8 + s] dt of RK2 step s     [8 + 64 + s] start time of step s
#define SCHED_DT 8
#define SCHED_TC (8 + 64)
#define SCHED_FLOATS (8 + 2 * 64)
#define SCHED_TN(a) ((a).sched ? (a).sched[0] : (a).tn)
#define SCHED_Y0(a) ((a).sched ? __float_as_int((a).sched[1]) : (a).y0)
#define RK_DT(a, s) ((a).sched ? (a).sched[SCHED_DT + (s)] : (a).dt[s])
#define RK_TC(a, s) ((a).sched ? (a).sched[SCHED_TC + (s)] : (a).tcur[s])

// ---------------------------------------------------------------- field math (device)
#define XPRE_INVALID (-1.0e30f)

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // F.softplus beta=1 thr=20

// tensorf_base.py:241-242
__device__ __forceinline__ float norm_coord(const nvfi_field_desc& f, int c, float p) {
    float size = f.aabb[3 + c] - f.aabb[c];
    float inv = 2.0f / size;
    return (p - f.aabb[c]) * inv - 1.f;
}
__host__ __device__ inline float norm_time(const nvfi_field_desc& f, float t) {  // tensorf_keyframe.py:501-506
    if (f.K == 1 || f.tmax == 0.f) return t * 0.f;
    return t * 2.f / f.tmax - 1.f;
}
__host__ __device__ inline float time_scale(const nvfi_field_desc& f) {
    return f.K > 1 ? (float)((double)f.tmax / (double)(f.K - 1)) : 1.f;
}
__host__ __device__ inline float snap_base(const nvfi_field_desc& f, float t) {   // tensorf_keyframe.py:646-654
    float ts = time_scale(f);
    float q = t / ts;
    float hi = (float)(f.K - 1);
    if (q < 0.f) q = 0.f;
    if (q > hi) q = hi;
    return rintf(q) * ts;
}
__host__ __device__ inline bool is_close(float a, float b) { return fabsf(a - b) <= 1e-8f + fabsf(1e-5f * b); }
__host__ __device__ inline float dt_max_of(const nvfi_field_desc& f) {
    return f.K > 1 ? (float)(0.5 * (double)f.tmax / (double)(f.K - 1)) : 1.f;
}

__device__ __forceinline__ bool gated_out(const nvfi_field_desc& f, float x, float y, float z) {
    return x < f.gate_lo[0] || x > f.gate_hi[0] || y < f.gate_lo[1] || y > f.gate_hi[1] || z < f.gate_lo[2] || z > f.gate_hi[2];
}

// bilinear tap set (ATen grid_sampler_2d, align_corners=True, zeros padding) on a channel-last plane
struct Bl {
    int base;          // texel index of (y0,x0)
    int W;
    float w, e, n, s;  // w = x-floor(x), e = 1-w, n = y-floor(y), s = 1-n
    bool m0, m1, m2, m3;
};
__device__ __forceinline__ void bl_setup_xy(float gx, float gy, int W, int H, Bl& b, int& x0, int& y0) {
    float x = (gx + 1.f) * ((float)(W - 1) / 2.f);
    float y = (gy + 1.f) * ((float)(H - 1) / 2.f);
    float xf = floorf(x), yf = floorf(y);
    b.w = x - xf; b.e = 1.f - b.w;
    b.n = y - yf; b.s = 1.f - b.n;
    xf = fminf(fmaxf(xf, -4.f), (float)W + 2.f);
    yf = fminf(fmaxf(yf, -4.f), (float)H + 2.f);
    if (!(xf == xf)) xf = -4.f;
    if (!(yf == yf)) yf = -4.f;
    x0 = (int)xf; y0 = (int)yf;
    bool xi0 = x0 >= 0 && x0 < W, xi1 = x0 + 1 >= 0 && x0 + 1 < W;
    bool yi0 = y0 >= 0 && y0 < H, yi1 = y0 + 1 >= 0 && y0 + 1 < H;
    b.m0 = xi0 && yi0; b.m1 = xi1 && yi0; b.m2 = xi0 && yi1; b.m3 = xi1 && yi1;
    b.base = y0 * W + x0;
    b.W = W;
}
__device__ __forceinline__ void bl_setup(float gx, float gy, int W, int H, Bl& b) {
    int x0, y0;
    bl_setup_xy(gx, gy, W, H, b, x0, y0);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// 4 consecutive channels (float4 index q4) of the bilinear sample; C = channels per texel
__device__ __forceinline__ float4 bl_sample4(const float* __restrict__ plane, int C, const Bl& b, int q4) {
    // unconditional loads (a masked tap reads texel 0 and is zeroed by a select): the four taps - and, in the callers' loops, the
    // taps of all six planes - are in flight together instead of one exec-masked branch and wait per tap
    const float* p = plane + (size_t)b.base * C + 4 * q4;
    const float* safe = plane + 4 * q4;
    float4 v0 = ld4(b.m0 ? p : safe);
    float4 v1 = ld4(b.m1 ? p + C : safe);
    float4 v2 = ld4(b.m2 ? p + (size_t)b.W * C : safe);
    float4 v3 = ld4(b.m3 ? p + (size_t)b.W * C + C : safe);
    v0 = b.m0 ? v0 : zero4(); v1 = b.m1 ? v1 : zero4(); v2 = b.m2 ? v2 : zero4(); v3 = b.m3 ? v3 : zero4();
    float nw = b.e * b.s, ne = b.w * b.s, sw = b.e * b.n, se = b.w * b.n;
    float4 r;
    r.x = v0.x * nw + v1.x * ne + v2.x * sw + v3.x * se;
    r.y = v0.y * nw + v1.y * ne + v2.y * sw + v3.y * se;
    r.z = v0.z * nw + v1.z * ne + v2.z * sw + v3.z * se;
    r.w = v0.w * nw + v1.w * ne + v2.w * sw + v3.w * se;
    return r;
}
// backward for 4 channels: g = upstream grads of the 4 sampled values. Scatters into gplane (if non-null),
// accumulates the un-scaled coordinate gradients.
__device__ __forceinline__ void bl_backward4(const float* __restrict__ plane, float* __restrict__ gplane, int C, const Bl& b,
                                             int q4, const float4& g, float& gx, float& gy) {
    const size_t o0 = (size_t)b.base * C + 4 * q4;
    const size_t o1 = o0 + C, o2 = o0 + (size_t)b.W * C, o3 = o2 + C;
    float4 v0 = b.m0 ? ld4(plane + o0) : zero4();
    float4 v1 = b.m1 ? ld4(plane + o1) : zero4();
    float4 v2 = b.m2 ? ld4(plane + o2) : zero4();
    float4 v3 = b.m3 ? ld4(plane + o3) : zero4();
    gx += ((v1.x - v0.x) * b.s + (v3.x - v2.x) * b.n) * g.x + ((v1.y - v0.y) * b.s + (v3.y - v2.y) * b.n) * g.y +
          ((v1.z - v0.z) * b.s + (v3.z - v2.z) * b.n) * g.z + ((v1.w - v0.w) * b.s + (v3.w - v2.w) * b.n) * g.w;
    gy += ((v2.x - v0.x) * b.e + (v3.x - v1.x) * b.w) * g.x + ((v2.y - v0.y) * b.e + (v3.y - v1.y) * b.w) * g.y +
          ((v2.z - v0.z) * b.e + (v3.z - v1.z) * b.w) * g.z + ((v2.w - v0.w) * b.e + (v3.w - v1.w) * b.w) * g.w;
    if (gplane) {
        float nw = b.e * b.s, ne = b.w * b.s, sw = b.e * b.n, se = b.w * b.n;
        if (b.m0) { atomicAdd(gplane + o0, nw * g.x); atomicAdd(gplane + o0 + 1, nw * g.y); atomicAdd(gplane + o0 + 2, nw * g.z); atomicAdd(gplane + o0 + 3, nw * g.w); }
        if (b.m1) { atomicAdd(gplane + o1, ne * g.x); atomicAdd(gplane + o1 + 1, ne * g.y); atomicAdd(gplane + o1 + 2, ne * g.z); atomicAdd(gplane + o1 + 3, ne * g.w); }
        if (b.m2) { atomicAdd(gplane + o2, sw * g.x); atomicAdd(gplane + o2 + 1, sw * g.y); atomicAdd(gplane + o2 + 2, sw * g.z); atomicAdd(gplane + o2 + 3, sw * g.w); }
        if (b.m3) { atomicAdd(gplane + o3, se * g.x); atomicAdd(gplane + o3 + 1, se * g.y); atomicAdd(gplane + o3 + 2, se * g.z); atomicAdd(gplane + o3 + 3, se * g.w); }
    }
}

// plane geometry: matModeSpace = [0,1],[0,2],[1,2]; matModeTime first axis = 2,1,0 (tensorf_keyframe.py:39-40)
__device__ __forceinline__ void plane_setups(const nvfi_field_desc& f, float x, float y, float z, float tn, Bl* b) {
    bl_setup(x, y, f.G[0], f.G[1], b[0]);
    bl_setup(x, z, f.G[0], f.G[2], b[1]);
    bl_setup(y, z, f.G[1], f.G[2], b[2]);
    bl_setup(z, tn, f.G[2], f.K, b[3]);
    bl_setup(y, tn, f.G[1], f.K, b[4]);
    bl_setup(x, tn, f.G[0], f.K, b[5]);
}
// coordinate-gradient multipliers (W-1)/2, (H-1)/2 for plane i
__device__ __forceinline__ void plane_mults(const nvfi_field_desc& f, int i, float& mx, float& my) {
    const int a[6] = {0, 0, 1, 2, 1, 0}, bb[3] = {1, 2, 2};
    mx = (float)(f.G[a[i]] - 1) / 2.f;
    my = i < 3 ? (float)(f.G[bb[i]] - 1) / 2.f : (float)(f.K - 1) / 2.f;
}

// degree-2 real SH bases at an (un-normalised) direction (models/sh.py:87-110)
__device__ __forceinline__ void sh_bases9(const float* d, float* b) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float x = d[0], y = d[1], z = d[2];
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[0] = C0; b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
    b[4] = C2[0] * xy; b[5] = C2[1] * yz; b[6] = C2[2] * (2.0f * zz - xx - yy); b[7] = C2[3] * xz; b[8] = C2[4] * (xx - yy);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------- single-pass ordered compaction (decoupled look-back)
// The ordered compact lists of a render / PDE call (valid samples, gate-inside samples, appearance-masked samples, kept points) used to take
// two launches each: per-group counts, then k_fill (scan of the counts + ballot-ranked fill).  With a look-back the kernel that produces the
// flags also places them: a workgroup publishes the count of its groups in a status word, sums the words of the workgroups in front of it and
// fills its part of the list in the same launch - the list is the one k_fill wrote, entry for entry.
//   status[b] = state << 62 | value   (state 0: nothing yet, 1: the workgroup's own count, 2: the inclusive prefix up to and including b)
// value may pack two counts (31 bits each): every partial sum stays below 2^31 per field because the totals do (R * S < 2^31).
// The status array must be zero at launch.  Workgroups are numbered by blockIdx.x: the dispatcher starts workgroups of a grid in index order,
// so every word a workgroup waits for belongs to a workgroup that is already resident (the launch-order assumption rocPRIM's look-back scan
// makes for its static tile order); the spin is an agent-scope atomic load, the words themselves carry all the data that is exchanged.
#define LB_MASK ((1ull << 62) - 1ull)
__device__ __forceinline__ unsigned long long lb_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lb_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
// Called by ONE whole wave of workgroup b (every lane, same arguments): publishes `agg`, returns the sum of the values of workgroups 0..b-1.
__device__ __forceinline__ unsigned long long lb_exclusive(unsigned long long* status, int b, unsigned long long agg) {
    const int lane = threadIdx.x & 63;
    if (b == 0) { if (lane == 0) lb_store(status, (2ull << 62) | agg); return 0ull; }
    if (lane == 0) lb_store(status + b, (1ull << 62) | agg);
    unsigned long long excl = 0ull;
    for (int j = b - 1;; j -= 64) {
        const int idx = j - lane;
        unsigned long long v;
        do { v = idx >= 0 ? lb_load(status + idx) : (2ull << 62); } while (__any((v >> 62) == 0ull));
        const unsigned long long incl = __ballot((v >> 62) == 2ull);     // (a lane in front of workgroup 0 reads as an inclusive prefix of zero)
        const int first = incl ? __ffsll((long long)incl) - 1 : 64;
        excl += wave_sum_u64(lane <= first ? (v & LB_MASK) : 0ull);
        if (incl) break;
    }
    if (lane == 0) lb_store(status + b, (2ull << 62) | (excl + agg));
    return excl;
}

// fragment sets (device pointers into the workspace)
struct RenderFrags {
    const float* fb;            // basis fwd: MT1 NS24
    const float* f1; const float* b1;  // L1 MT4 NS55
    const float* f2; const float* b2;  // L2 MT4 NS64
    const float* f3; const float* b3;  // L3 MT1 NS64
    const float* t3;            // T3: MT4 NS4
    const float* t2;            // T2: MT4 NS64
    const float* t1;            // T1: MT4 NS64 (rows = RENDER_IN slots)
    const float* tb;            // Tbasis: MT2 NS16
};
#define RF_B (1 * 24 * 64)
#define RF_1 (4 * 55 * 64)
#define RF_2 (4 * 64 * 64)
#define RF_3 (1 * 64 * 64)
#define RT_3 (4 * 4 * 64)
#define RT_2 (4 * 64 * 64)
#define RT_1 (4 * 64 * 64)
#define RT_B (2 * 16 * 64)
#define RENDER_FRAG_FLOATS (RF_B + RF_1 + RF_2 + RF_3 + 128 + 128 + 32 + RT_3 + RT_2 + RT_1 + RT_B)

// app stash rows per tile: g 32 | x_in 64 | h1 64 | h2 64   (forward) ; go 16 | gz2 64 | gz1 64 | gfeat 16 (backward)
#define APP_F_ROWS (32 + 64 + 64 + 64)
#define APP_B_ROWS (16 + 64 + 64 + 16)

int pack_vel_frags(const float* const* W, const float* const* b, float* buf, VelFrags* out, PackJobs* jobs);
int pack_render_frags(const nvfi_field_desc* f, float* buf, RenderFrags* out, PackJobs* jobs);
int launch_pack(const PackJobs& jobs, hipStream_t st);
// zero `bytes` (a multiple of 4) at p with a KERNEL.  Round 6: hipMemsetAsync is not used on any path that can be captured - as the root node of a
// captured hipGraph (the PDE term's own graph of the multi-rank step) its replay left the workspace's histogram words uncleared / wrote elsewhere:
// k_pde_bucket then scattered through stale offsets ("Write access to a read-only page", two ranks on one GPU; tools/run_cap_debug2.py).
int launch_zero(void* p, int64_t bytes, hipStream_t st);
int launch_scan_fill(const int* cnt, int* off, int64_t ngroups, int* total, const uint8_t* flags, int* list, hipStream_t st);
int launch_wgrad(const WgradJobs& wj, const ReduceJobs& rj, hipStream_t st);
int launch_wgrad_ring(WgradJobs& bj, ReduceJobs& br, hipStream_t st);   // wgrad_ring.hip
int ensure_lds_attrs();
bool fused_launch();     // NVFI_FUSED_LAUNCH (default 1): round 5's fused small launches (render.hip)
