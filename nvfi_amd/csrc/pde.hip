// pde.hip - velocity PDE regulariser: divergence + transport residual of the VelBasis field.
//
// Reference semantics: NVFi.get_vel_loss (models/nvfi.py:42-84):
//   occupancy prefilter (normalise, snap to keyframe, RK2 back-advect, density, alpha >= thres),
//   J = d(v,a)/d(x,y,z,t) of the un-gated vel_net (functorch vmap(jacrev)), div = tr J[:3,:3],
//   transport = J[:3,:3] v + J[:3,3] - a, loss = 5 mean(div^2) + 0.1 mean(transport^2),
//   and loss.backward() through the Jacobian (second order).
//
// MI355X design: the Jacobian is computed in FORWARD mode (value + 4 tangents pushed through the
// MFMA engine, activations register-resident), the backward is the hand-written reverse of that
// tangent program (4 tangent-adjoint passes feeding a second-derivative correction into the
// value-adjoint pass), and weight gradients are split-K MFMA over the stashed tiles (k_wgrad).
// Points are bucketed by RK2 step count so workgroups of the prefilter are homogeneous.
#include <stdlib.h>
#include "common.h"
#include "vel.h"
#include "render.h"
#include "pde.h"
#include "fuse.h"
#include "scatter.h"
#include "frags.h"
#include "x6.h"

// ---------------------------------------------------------------- prefilter
__global__ __launch_bounds__(256) void k_pde_prep(PdePrepArgs a) {
    __shared__ int hist[PDE_MAX_CLASS];
    __shared__ int gbase[PDE_MAX_CLASS];
    if (threadIdx.x < PDE_MAX_CLASS) hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool on = i < a.P;
    const nvfi_field_desc& f = a.f;
    int ns = 0, lr = 0;
    if (on) {
        const float t = a.t[i];
        const float base = snap_base(f, t);
        float xn[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) xn[c] = norm_coord(f, c, a.points[3 * i + c]);
        a.qorig[i] = make_float4(xn[0], xn[1], xn[2], t);
        a.xw[i] = make_float4(xn[0], xn[1], xn[2], norm_time(f, base));
        // A point outside the velocity gate never moves (velocity_field.py:28-33,46-51: v = 0 there, so x_mid = x and the step leaves
        // x where it is - it stays outside for every later step): class 0, no network evaluations.  Exact, not an approximation.
        const bool frozen = gated_out(f, xn[0], xn[1], xn[2]);
        a.pt_t[i] = t; a.pt_base[i] = frozen ? t : base;
        // number of RK2 steps this point will take (same fp32 recurrence as the integrator)
        const float dtm = dt_max_of(f);
        float off = frozen ? 0.f : t - base;
        while (fabsf(off) > 0.f && ns < PDE_MAX_CLASS - 1) {
            float m = fminf(fabsf(off), dtm);
            off = off - (off > 0.f ? m : -m);
            ++ns;
        }
        a.cls[i] = ns;
        lr = atomicAdd(&hist[ns], 1);          // rank inside the workgroup (LDS atomic)
    }
    __syncthreads();
    if (threadIdx.x < PDE_MAX_CLASS && hist[threadIdx.x] > 0) gbase[threadIdx.x] = atomicAdd(&a.cls_count[threadIdx.x], hist[threadIdx.x]);
    __syncthreads();
    if (on) a.rank[i] = gbase[ns] + lr;
}
// bucket points by step count: perm[class_off[c] + rank] = i
__global__ __launch_bounds__(256) void k_pde_bucket(int64_t P, const int* cls, const int* rank, const int* cls_count, int* perm,
                                                    float* pt_t_perm, float* pt_base_perm, const float* pt_t, const float* pt_base) {
    __shared__ int off[PDE_MAX_CLASS];
    if (threadIdx.x == 0) {
        // most steps first: long-running workgroups start early
        int s = 0;
        for (int c = PDE_MAX_CLASS - 1; c >= 0; --c) { off[c] = s; s += cls_count[c]; }
    }
    __syncthreads();
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int p = off[cls[i]] + rank[i];
    perm[p] = (int)i;
    pt_t_perm[p] = pt_t[i]; pt_base_perm[p] = pt_base[i];
}
// density at the warped point (k_density_q, scatter.hip) -> alpha -> keep flag (nvfi.py:56-64); one wave = 64 consecutive points
__global__ __launch_bounds__(256) void k_pde_keep(nvfi_field_desc f, int64_t P, const float* sig, uint8_t* flags, int* cnt) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < P) {
        const float alpha = 1.f - expf(-sig[i] * 0.01f * 25.f);
        keep = alpha >= f.alpha_thres;
        flags[i] = keep ? 1 : 0;
    }
    const unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && i < P) cnt[i >> 6] = __popcll(b);
}

// k_pde_keep + the k_fill launch behind it (round 5): a wave = 64 consecutive points, a workgroup publishes its kept count and places its
// entries of the ordered kept list itself (look-back, common.h) - the list k_fill wrote, entry for entry
__global__ __launch_bounds__(256) void k_pde_keep_fill(nvfi_field_desc f, int64_t P, const float* sig, uint8_t* flags, unsigned long long* lb, int* klist, int* kcount) {
    __shared__ int ck[4];
    __shared__ unsigned long long excl_sh;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < P) {
        const float alpha = 1.f - expf(-sig[i] * 0.01f * 25.f);
        keep = alpha >= f.alpha_thres;
        flags[i] = keep ? 1 : 0;
    }
    const unsigned long long b = __ballot(keep);
    if (lane == 0) ck[w] = __popcll(b);
    __syncthreads();
    if (w == 0) {
        const unsigned long long agg = (unsigned long long)((ck[0] + ck[1]) + (ck[2] + ck[3]));
        const unsigned long long e = lb_exclusive(lb, (int)blockIdx.x, agg);
        if (lane == 0) {
            excl_sh = e;
            if (blockIdx.x == gridDim.x - 1) *kcount = (int)(e + agg);
        }
    }
    __syncthreads();
    int base = (int)excl_sh;
    for (int k = 0; k < w; ++k) base += ck[k];
    if (keep) klist[base + __popcll(b & ((1ull << lane) - 1ull))] = (int)i;
}

// (The Jacobian passes - value + four tangent columns forward, their adjoints with the second-derivative corrections - live in pde_jet.hip,
// pde_jet6.hip and pde_fuse.hip; the column-parallel forms that stood here in rounds 1-5 were retired in round 6.)
#define pde_pass_count pde_pass_count_of
// Round 5: the call's tiny bookkeeping kernels ride in k_pde_seeds.  Every workgroup of the launch draws a ticket when it is done (its
// partial sums are agent-scope atomics that have completed - s_waitcnt vmcnt(0) - before the ticket, as in k_tile_hist); the LAST one does the
// work of k_pde_pass_count (the sample count of this pass for the weight-gradient kernels, the fused adjoint's queue word), and - in the last
// pass of the call - of k_pde_finish (the value) and k_pde_counters.  One thread, a few dozen loads.
__device__ __forceinline__ void pde_tail(const PdeJetArgs& a, int count) {
    if (threadIdx.x != 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const PdeTail& T = a.tail;
    if (__hip_atomic_fetch_add(T.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (int)gridDim.x - 1) return;
    __hip_atomic_store(T.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (the next pass of the call draws from zero again)
    if (T.dcount) {
        *T.dcount = (count + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES;
        T.dcount[8] = 0;
    }
    if (T.out) {
        const int64_t nk = *a.kcount;
        const double sd = __hip_atomic_load(a.sums + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), st = __hip_atomic_load(a.sums + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        T.out[0] = nk > 0 ? (float)(5.0 * sd / (double)nk + 0.1 * st / (3.0 * (double)nk)) : 0.f;
        T.out[1] = (float)nk; T.out[2] = (float)sd; T.out[3] = (float)st;
    }
    if (T.counters) {
        int64_t evals = 0;
        for (int c = 0; c < PDE_MAX_CLASS; ++c) evals += 2ll * c * T.cls_count[c];
        int64_t* c8 = T.counters;
        c8[0] = 0; c8[1] = T.P; c8[2] = 0; c8[3] = evals; c8[4] = *a.kcount; c8[5] = T.pre16 ? a.kcount[1] : 0; c8[6] = c8[7] = 0;
    }
}
// K3: per-point residuals (nvfi.py:74-83), loss partial sums and adjoint seeds
__global__ __launch_bounds__(256) void k_pde_seeds(PdeJetArgs a) {
    __shared__ float red[8];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int count = pde_pass_count(a);
    if ((int)(blockIdx.x * 256) >= count) { if (a.tail.ticket) pde_tail(a, count); return; }
    const bool active = i < count;
    const size_t cs = a.cap;
    const int capc = (count + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES;   // the jet workgroups cover whole 128-point groups
    const float inv_n = 1.f / (float)(*a.kcount);
    float sd = 0.f, st = 0.f;
    if (i < capc) {
        float w[6], wd[4][6], aw[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            w[k] = a.wout[(size_t)k * cs + i]; aw[k] = a.wout[(size_t)(30 + k) * cs + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) wd[j][k] = a.wout[(size_t)(6 + 6 * j + k) * cs + i];
        }
        float4 q = active ? a.qorig[a.klist[a.first + i]] : zero4();
        const float x = q.x, y = q.y, z = q.z;
        float v[3], ac[3], Jv[3][4];
        vel_from_w(w, x, y, z, v);
        acc_from_w(aw, x, y, z, ac);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            Jv[0][j] = wd[j][0] - wd[j][4] * z + wd[j][5] * y;
            Jv[1][j] = wd[j][1] + wd[j][3] * z - wd[j][5] * x;
            Jv[2][j] = wd[j][2] - wd[j][3] * y + wd[j][4] * x;
        }
        Jv[0][2] += -w[4]; Jv[0][1] += w[5];
        Jv[1][2] += w[3];  Jv[1][0] += -w[5];
        Jv[2][1] += -w[3]; Jv[2][0] += w[4];
        const float div = Jv[0][0] + Jv[1][1] + Jv[2][2];
        float tr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) tr[c] = Jv[c][0] * v[0] + Jv[c][1] * v[1] + Jv[c][2] * v[2] + Jv[c][3] - ac[c];
        float* sp = a.seeds + i;
        if (active) {
            sd = div * div; st = tr[0] * tr[0] + tr[1] * tr[1] + tr[2] * tr[2];
            const float lscale = a.scale_dev ? *a.scale_dev : a.scale;
            const float gdiv = lscale * 10.f * div * inv_n;
            float gtr[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gtr[c] = lscale * 0.2f * tr[c] * inv_n / 3.f;
            float gJ[3][4], gv[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { gJ[c][k] = gtr[c] * v[k]; gv[k] += gtr[c] * Jv[c][k]; }
                gJ[c][3] = gtr[c];
                gJ[c][c] += gdiv;
            }
            sp[0 * cs] = gv[0]; sp[1 * cs] = gv[1]; sp[2 * cs] = gv[2];
            sp[3 * cs] = z * gv[1] - y * gv[2] + gJ[1][2] - gJ[2][1];
            sp[4 * cs] = -z * gv[0] + x * gv[2] - gJ[0][2] + gJ[2][0];
            sp[5 * cs] = y * gv[0] - x * gv[1] + gJ[0][1] - gJ[1][0];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sp[(6 + 6 * j + 0) * cs] = gJ[0][j]; sp[(6 + 6 * j + 1) * cs] = gJ[1][j]; sp[(6 + 6 * j + 2) * cs] = gJ[2][j];
                sp[(6 + 6 * j + 3) * cs] = z * gJ[1][j] - y * gJ[2][j];
                sp[(6 + 6 * j + 4) * cs] = -z * gJ[0][j] + x * gJ[2][j];
                sp[(6 + 6 * j + 5) * cs] = y * gJ[0][j] - x * gJ[1][j];
            }
            const float ga[3] = {-gtr[0], -gtr[1], -gtr[2]};
            sp[30 * cs] = ga[0]; sp[31 * cs] = ga[1]; sp[32 * cs] = ga[2];
            sp[33 * cs] = -y * ga[1] - z * ga[2];
            sp[34 * cs] = -x * ga[0] - z * ga[2];
            sp[35 * cs] = -x * ga[0] - y * ga[1];
            if (a.jac && (a.first + i) < a.n_jac) {
                float* jp = a.jac + (size_t)(a.first + i) * 24;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) jp[c * 4 + j] = Jv[c][j];
#pragma unroll
                for (int k = 12; k < 24; ++k) jp[k] = 0.f;
            }
        } else {
            // ragged tail of the last workgroup: zero seeds so that every adjoint stash row is zero
            for (int k = 0; k < 36; ++k) sp[(size_t)k * cs] = 0.f;
        }
    }
    sd = wave_sum(sd); st = wave_sum(st);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = sd; red[4 + (threadIdx.x >> 6)] = st; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(a.sums + 0, (double)(red[0] + red[1] + red[2] + red[3]));
        atomicAdd(a.sums + 1, (double)(red[4] + red[5] + red[6] + red[7]));
    }
    if (a.tail.ticket) pde_tail(a, count);
}

// tiny helpers that keep the bookkeeping on the device (no synchronisation on the launch stream)
// per-pass sample count for k_wgrad (whole 128-point groups: the ragged rows of the last group are zero)
__global__ void k_pde_pass_count(const int* kcount, int64_t first, int64_t cap, int* dcount) {
    if (threadIdx.x == 0) {
        const int64_t c = (int64_t)(*kcount) - first;
        const int n = c <= 0 ? 0 : (c > cap ? (int)cap : (int)c);
        *dcount = (n + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES;
        dcount[8] = 0;          // k_pde_fuse_bwd's queue of acceleration-net tiles
    }
}
// counters[1] = candidates, [3] = prefilter net evaluations (2 per RK2 step, from the step-class histogram), [4] = kept points
// [5] = points re-evaluated in fp32 behind the opt-in fp16 pre-pass (pre16.hip), 0 otherwise
__global__ void k_pde_counters(const int* cls_count, const int* kcount, int64_t P, int pre16, int64_t* c8) {
    if (threadIdx.x == 0) {
        int64_t evals = 0;
        for (int c = 0; c < PDE_MAX_CLASS; ++c) evals += 2ll * c * cls_count[c];
        c8[0] = 0; c8[1] = P; c8[2] = 0; c8[3] = evals; c8[4] = *kcount; c8[5] = pre16 ? kcount[1] : 0; c8[6] = c8[7] = 0;
    }
}
__global__ void k_pde_finish(const double* sums, const int* kcount, float* out) {
    if (threadIdx.x == 0) {
        const int64_t nk = *kcount;
        const double sd = sums[0], st = sums[1];
        out[0] = nk > 0 ? (float)(5.0 * sd / (double)nk + 0.1 * st / (3.0 * (double)nk)) : 0.f;
        out[1] = (float)nk; out[2] = (float)sd; out[3] = (float)st;
    }
}

// ---------------------------------------------------------------- host
struct PdePlan {
    float4 *qorig, *xw;
    float *pt_t, *pt_base, *pt_t_perm, *pt_base_perm;
    int *cls, *rank, *cls_count, *perm, *cnt, *off, *klist, *kcount, *dcount;
    uint8_t* flags;
    float* sig;     // density at the warped points (prefilter)
    float4* xw16; uint8_t* near; int* blist; int* bcount; void* img16; void* img16lo;   // fp16 pre-pass (pre16.hip)
    void* x6img;                                                                       // x6 prefilter (vel_x6.hip)
    double* sums; unsigned long long* lb; int64_t zero_bytes;
    float *vel_frag, *a_frag, *vel_x4, *a_x4, *stash, *seeds, *wout, *slabs;
    int64_t chunk, total;
};
static void plan_pde(int64_t P, void* ws, PdePlan* L) {
    Bump B{(char*)ws, 0, 0};
    const int64_t nw = (P + 63) / 64;
    L->qorig = B.take<float4>(P); L->xw = B.take<float4>(P);
    L->pt_t = B.take<float>(P); L->pt_base = B.take<float>(P); L->pt_t_perm = B.take<float>(P); L->pt_base_perm = B.take<float>(P);
    L->cls = B.take<int>(P); L->rank = B.take<int>(P);
    // one fill clears: the class histogram + counts (PDE_MAX_CLASS + 16 ints: [+0] kept, [+1] band, [+2] ticket of k_pde_seeds' tail), the four
    // double loss sums and the look-back words of k_pde_keep_fill (one per workgroup of 256 points)
    L->cls_count = B.take<int>(PDE_MAX_CLASS + 16); L->sums = reinterpret_cast<double*>(L->cls_count + PDE_MAX_CLASS + 16);
    B.off += 4 * (int64_t)sizeof(double);
    L->lb = reinterpret_cast<unsigned long long*>(L->cls_count + PDE_MAX_CLASS + 16 + 8);
    B.off += ((P + 255) / 256) * (int64_t)sizeof(unsigned long long);
    L->zero_bytes = (PDE_MAX_CLASS + 16) * (int64_t)sizeof(int) + 4 * (int64_t)sizeof(double) + ((P + 255) / 256) * (int64_t)sizeof(unsigned long long);
    L->perm = B.take<int>(P);
    L->cnt = B.take<int>(nw); L->off = B.take<int>(nw + 1); L->klist = B.take<int>(P); L->kcount = L->cls_count + PDE_MAX_CLASS;
    L->flags = B.take<uint8_t>(nw * 64);
    L->sig = B.take<float>(nw * 64);
    L->xw16 = B.take<float4>(P); L->near = B.take<uint8_t>(nw * 64); L->blist = B.take<int>(P); L->bcount = L->cls_count + PDE_MAX_CLASS + 1;
    L->img16 = B.take<float4>(PRE16_IMAGE_BYTES / 16);
    L->img16lo = B.take<float4>(PRE16_IMAGE_BYTES / 16);
    L->x6img = B.take<float4>(X6_IMAGE_BYTES / 16);
    L->dcount = B.take<int>(16);
    L->vel_frag = B.take<float>(VEL_FRAG_FLOATS); L->a_frag = B.take<float>(VEL_FRAG_FLOATS);
    L->vel_x4 = B.take<float>(VEL_X4_FLOATS);
    L->a_x4 = B.take<float>(4 * X4_FLOATS(4, 64) + X4_FLOATS(4, 4));      // transposed fragments of a_weight_net (pde_fuse.hip)
    L->chunk = P < PDE_CHUNK ? (P + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES : PDE_CHUNK;
    L->stash = B.take<float>(L->chunk / TILE * (int64_t)PDE_TILE_ROWS * REGF);
    L->seeds = B.take<float>(36 * L->chunk);
    L->wout = B.take<float>(36 * L->chunk);
    L->slabs = B.take<float>((int64_t)PDE_NSLAB * (128 * 128 + 128) * 18);
    L->total = align_up(B.off, 256);
}
extern "C" int nvfi_pde_workspace_bytes(const nvfi_field_desc* f, int64_t P, int64_t* bytes) {
    (void)f;
    PdePlan L; plan_pde(P, nullptr, &L);
    *bytes = L.total;
    return 0;
}

static int ensure_pde_attrs() {
    static bool done = false;
    if (done) return 0;
    done = true;
    return 0;
}

static int pde_loss_impl(const nvfi_field_desc* f, int64_t P, const float* points, const float* t, float loss_scale, const float* loss_scale_dev,
                         float* out, const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, int64_t* counters,
                         uint8_t* kept_out, float* jac_out, int64_t n_jac, int64_t* host_info, void* stream);
// nvfi_pde_loss_split: the stream of the adjoint + weight-gradient half of the call in progress (NULL: everything on `stream`)
static thread_local hipStream_t t_bwd_stream = nullptr;
static thread_local hipEvent_t t_split_ev = nullptr;

extern "C" int nvfi_pde_loss_ex(const nvfi_field_desc* f, int64_t P, const float* points, const float* t, float loss_scale,
                                float* out, const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, int64_t* counters,
                                uint8_t* kept_out, float* jac_out, int64_t n_jac, int64_t* host_info, void* stream) {
    return pde_loss_impl(f, P, points, t, loss_scale, nullptr, out, grads, workspace, workspace_bytes, counters, kept_out, jac_out, n_jac, host_info, stream);
}

/* The VALUE of the loss (out[]) is complete on `stream` after the Jacobian forward; the adjoint pass and the weight gradients - a third of the
 * call's time, needed only by the caller's backward - run on `bwd_stream`, ordered behind the forward by an event.  For a caller that waits
 * for the value on `stream` right after the call (train_nvfi.py:233 `if loss_vel > 0`) and then issues other work there (the renders'
 * backward), the two overlap.  The caller orders its use of the gradients (and the release of the workspace) behind `bwd_stream`.
 * A candidate count above one chunk (262 144) falls back to one stream: the chunks share the stash. */
extern "C" int nvfi_pde_loss_split(const nvfi_field_desc* f, int64_t P, const float* points, const float* t, float loss_scale,
                                   float* out, const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, int64_t* counters,
                                   uint8_t* kept_out, float* jac_out, int64_t n_jac, int64_t* host_info, void* stream, void* bwd_stream) {
    if (!t_split_ev) HIPCK(hipEventCreateWithFlags(&t_split_ev, hipEventDisableTiming));
    t_bwd_stream = (bwd_stream && bwd_stream != stream) ? (hipStream_t)bwd_stream : nullptr;
    const int rc = pde_loss_impl(f, P, points, t, loss_scale, nullptr, out, grads, workspace, workspace_bytes, counters, kept_out, jac_out, n_jac, host_info, stream);
    t_bwd_stream = nullptr;
    return rc;
}

extern "C" int nvfi_pde_loss_dev(const nvfi_field_desc* f, int64_t P, const float* points, const float* t, const float* loss_scale_dev,
                                 float* out, const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream) {
    if (!loss_scale_dev) return nvfi_fail(2, "nvfi_pde_loss_dev: loss_scale_dev is NULL");
    return pde_loss_impl(f, P, points, t, 1.f, loss_scale_dev, out, grads, workspace, workspace_bytes, counters, nullptr, nullptr, 0, nullptr, stream);
}

static int pde_loss_impl(const nvfi_field_desc* f, int64_t P, const float* points, const float* t, float loss_scale, const float* loss_scale_dev,
                         float* out, const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, int64_t* counters,
                         uint8_t* kept_out, float* jac_out, int64_t n_jac, int64_t* host_info, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (P <= 0) return nvfi_fail(2, "P must be positive");
    if (P >= (1ll << 31) - 256) return nvfi_fail(2, "P too large");
    if (!f->use_vel) return nvfi_fail(2, "PDE loss needs use_vel");
    if (ensure_pde_attrs() || ensure_lds_attrs()) return 1;
    PdePlan L; plan_pde(P, workspace, &L);
    if (L.total > workspace_bytes) return nvfi_fail(4, "workspace too small: need %lld bytes, got %lld", (long long)L.total, (long long)workspace_bytes);
    if (launch_zero(L.cls_count, L.zero_bytes, st)) return 1;     // histogram / counts, loss sums, look-back words: one fill (a kernel, not hipMemsetAsync: common.h)
    const bool fl = fused_launch();
    PackJobs jobs; jobs.n = 0;
    VelFrags VW, AW;
    FragCache FC; const bool cached = f->frags != nullptr;       // round 5: the field's fragment cache (nvfi_pack_frags) instead of a repack per call
    if (cached) frag_cache_layout(f->frags, &FC);
    if (pack_vel_frags(f->vW, f->vb, cached ? FC.vel : L.vel_frag, &VW, &jobs)) return 3;
    if (pack_vel_frags(f->aW, f->ab, cached ? FC.anet : L.a_frag, &AW, &jobs)) return 3;
    if (!cached && launch_pack(jobs, st)) return 1;
    // fused jet kernels (pde_jet.hip / pde_jet6.hip): x4 copies of the v-net fragments.  (The column-parallel Jacobian kernels of round 1 - one workgroup
    // per (tile, column), NVFI_PDE_JET=0 - were retired in round 6.)
    // NVFI_PDE_FUSE (default 1): pde_fuse.hip - weight_net's Jacobian adjoint and its four 128 x 128 weight gradients in one persistent kernel
    // (no gz_1..gz_4 stash, no second pass over the z / zd stash); 0: k_pde_jet_bwd + k_wgrad_ring8 over the full adjoint stash
    static int pde_fuse = -1;
    if (pde_fuse < 0) { const char* e = getenv("NVFI_PDE_FUSE"); pde_fuse = e ? atoi(e) : 1; }
    // prefilter mode: x6 (default, round 5: vel_x6.hip - the fp32 products of the hidden layers formed exactly on the 16-bit matrix pipe) | fp32 (the
    // feature-split fp32 MFMA kernel of vel_split.hip, the default of rounds 2-4; "engine32", k_rk2_fwd of vel.hip with the same numbers bit for
    // bit and ~4 % slower, was retired in round 6) | fp16band (pre16.hip: fp16-input pass + fp32 re-evaluation of the unsafe points)
    // split16band (opt-in, pre16.hip): the pre-pass with fp32 products emulated by two binary16 terms per operand (three fp16 MFMAs), and a
    // band 100 x narrower than fp16band's in front of the same fp32 re-evaluation
    static int pre16 = -1; static float band16 = 0.1f, eps16 = 2e-3f;
    if (pre16 < 0) {
        const char* e = getenv("NVFI_PDE_PREFILTER");
        if (e && strcmp(e, "fp16band") && strcmp(e, "split16band") && strcmp(e, "fp32") && strcmp(e, "split32") && strcmp(e, "x6"))
            return nvfi_fail(2, "NVFI_PDE_PREFILTER must be fp32, x6, fp16band or split16band");
        // 2 = split kernel ("split32" = "fp32"); 4 = x6 (vel_x6.hip: fp32 products formed exactly from three binary16 terms per operand)
        // round 5: x6 is the default - its error against float64 is not larger than the fp32 MFMA kernel's (tests/test_gpu_x6.py) and it is 1.35x faster
        pre16 = !e ? 4 : (!strcmp(e, "fp16band") ? 1 : (!strcmp(e, "split16band") ? 3 : (!strcmp(e, "x6") ? 4 : 2)));
        if (pre16 == 3) { band16 = 1e-3f; eps16 = 2e-5f; }
    }
    const float4* f4[6] = {nullptr}; const float4* t4[6] = {nullptr};
    if (cached) { x4f_pointers(FC.vel_x4f, f4); x4b_pointers(FC.vel_x4b, t4); }
    else {
        X4Jobs xj; xj.n = 0;
        float* p = L.vel_x4;
        auto add = [&](const float* src, int MT, int NS, const float4** slot) {
            xj.src[xj.n] = src; xj.dst[xj.n] = p; xj.MT[xj.n] = MT; xj.NS[xj.n] = NS; ++xj.n;
            *slot = reinterpret_cast<const float4*>(p);
            p += X4_FLOATS(MT, NS);
        };
        add(VW.f[0], 4, 14, &f4[0]);
        for (int l = 1; l <= 4; ++l) add(VW.f[l], 4, 64, &f4[l]);
        add(VW.f[5], 1, 64, &f4[5]);
        for (int l = 1; l <= 4; ++l) add(VW.t[l], 4, 64, &t4[l]);
        add(VW.t[5], 4, 4, &t4[5]);
        if (launch_frag_x4(xj, st)) return 1;
    }
    const float4* ta4[6] = {nullptr};
    if (cached) a_x4b_pointers(FC.a_x4b, ta4);
    else if (pde_fuse && grads) {
        X4Jobs xj; xj.n = 0;
        float* p = L.a_x4;
        for (int l = 1; l <= 5; ++l) {
            const int NS = l < 5 ? 64 : 4;
            xj.src[xj.n] = AW.t[l]; xj.dst[xj.n] = p; xj.MT[xj.n] = 4; xj.NS[xj.n] = NS; ++xj.n;
            ta4[l] = reinterpret_cast<const float4*>(p);
            p += X4_FLOATS(4, NS);
        }
        if (launch_frag_x4(xj, st)) return 1;
    }
    const unsigned pb = (unsigned)((P + 255) / 256);
    PdePrepArgs pa; pa.f = *f; pa.P = P; pa.points = points; pa.t = t; pa.qorig = L.qorig; pa.xw = L.xw; pa.pt_t = L.pt_t; pa.pt_base = L.pt_base;
    pa.cls = L.cls; pa.rank = L.rank; pa.cls_count = L.cls_count;
    hipLaunchKernelGGL(k_pde_prep, dim3(pb), dim3(256), 0, st, pa);
    hipLaunchKernelGGL(k_pde_bucket, dim3(pb), dim3(256), 0, st, P, L.cls, L.rank, L.cls_count, L.perm, L.pt_t_perm, L.pt_base_perm, L.pt_t, L.pt_base);
    LAUNCHCK();
    // RK2 back-advection in bucket order (per-point times)
    const int64_t nw = (P + 63) / 64;
    Rk2Args ra; memset(&ra, 0, sizeof(ra));
    ra.f = *f; ra.Wv = VW; ra.xw = L.xw; ra.xout = nullptr; ra.dt_max = dt_max_of(*f); ra.max_steps = PDE_MAX_CLASS;
    DensityArgs da; memset(&da, 0, sizeof(da));
    da.f = *f; da.per_point_t = 1; da.sigma_out = L.sig;
    SplitArgs sa; memset(&sa, 0, sizeof(sa));
    sa.f = *f; sa.xw = L.xw; sa.dt_max = ra.dt_max; sa.max_steps = PDE_MAX_CLASS;
    for (int l = 0; l < 6; ++l) { sa.f4[l] = f4[l]; sa.bv[l] = VW.b[l]; }
    if (pre16 == 2) {
        sa.count = nullptr; sa.n_direct = P; sa.list = L.perm; sa.pt_t = L.pt_t_perm; sa.pt_base = L.pt_base_perm;
        if (launch_rk2_split(sa, P, 1, st)) return 1;
        da.n_direct = P; da.xw = L.xw;
        if (launch_density_q(da, P, st)) return 1;
    } else if (pre16 == 4) {
        X6Args xa; memset(&xa, 0, sizeof(xa));
        xa.f = *f; xa.img = cached ? FC.vel_x6 : L.x6img; xa.n_direct = P; xa.list = L.perm; xa.xw = L.xw; xa.pt_t = L.pt_t_perm; xa.pt_base = L.pt_base_perm;
        xa.dt_max = ra.dt_max; xa.max_steps = PDE_MAX_CLASS;
        if (!cached && launch_pack_x6(f->vW, L.x6img, st)) return 1;
        { ProfScope ps(PK_PDE_PREFILTER, st); if (launch_rk2_x6(xa, P, st)) return 1; }
        da.n_direct = P; da.xw = L.xw;
        if (launch_density_q(da, P, st)) return 1;
    } else {
        // opt-in (pre16.hip): fp16-input pass over every candidate, then the fp32 kernel for the points whose decision is not safe
        Pre16Args qa; memset(&qa, 0, sizeof(qa));
        qa.f = *f; qa.img = L.img16; qa.img_lo = pre16 == 3 ? L.img16lo : nullptr; qa.P = P; qa.list = L.perm; qa.xw = L.xw; qa.xout = L.xw16; qa.near = L.near;
        qa.pt_t = L.pt_t_perm; qa.pt_base = L.pt_base_perm; qa.dt_max = ra.dt_max; qa.max_steps = PDE_MAX_CLASS; qa.eps_gate = eps16;
        if (launch_pre16(f, qa, st)) return 1;
        da.n_direct = P; da.xw = L.xw16;
        if (launch_density_q(da, P, st)) return 1;
        if (launch_pde_band(f, P, L.perm, L.sig, L.near, band16, L.flags, L.cnt, st)) return 1;
        launch_scan_fill(L.cnt, L.off, nw, L.bcount, L.flags, L.blist, st);
        if (launch_pde_band_map(P, L.bcount, L.perm, L.blist, st)) return 1;
        // short list, long trajectories: the feature-split kernel (a tile's latency, not the chip's throughput, bounds this pass)
        sa.count = L.bcount; sa.list = L.blist; sa.pt_t = L.pt_t; sa.pt_base = L.pt_base; sa.pt_by_list = 1;
        if (launch_rk2_split(sa, P, 0, st)) return 1;
        da.count = L.bcount; da.list = L.blist; da.xw = L.xw;
        if (launch_density_q(da, P, st)) return 1;
    }
    if (fl) hipLaunchKernelGGL(k_pde_keep_fill, dim3(pb), dim3(256), 0, st, *f, P, L.sig, L.flags, L.lb, L.klist, L.kcount);
    else {
        hipLaunchKernelGGL(k_pde_keep, dim3(pb), dim3(256), 0, st, *f, P, L.sig, L.flags, L.cnt);
        launch_scan_fill(L.cnt, L.off, nw, L.kcount, L.flags, L.klist, st);
    }
    LAUNCHCK();
    // The kept count stays on the device (L.kcount): the jet passes are launched with worst-case grids and size themselves from it.
    // Only a caller that asks for host_info (diagnostics; the Python mirror's `return 0.` decision, nvfi.py:66) pays a synchronisation.
    // A split call (nvfi_pde_loss_split: the reference's loop) waits LATER - once the Jacobian forward and the value are queued on `stream` and the
    // adjoint half on the other one: the host's wait then covers the prefilter AND the forward (the caller's own wait for the value, `if loss_vel > 0`,
    // returns at once) and the device does not idle between the prefilter and the jets while the host wakes up (30-140 us per iteration in a trace).
    const bool defer_host = host_info && t_bwd_stream && P <= L.chunk && grads;
    // (static storage, per host thread: an error return between the copy being queued and the wait must not leave the copy a dead stack frame
    // to write into - ADVICE r4)
    static thread_local int hcnt_early[PDE_MAX_CLASS + 16];
    bool early_copy = false;
    auto read_host_info = [&]() -> int {
        int hcnt[PDE_MAX_CLASS + 16];
        if (!early_copy) HIPCK(hipMemcpyAsync(hcnt, L.cls_count, sizeof(hcnt), hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
        if (early_copy) memcpy(hcnt, hcnt_early, sizeof(hcnt));
        int64_t evals = 0;
        for (int c = 0; c < PDE_MAX_CLASS; ++c) evals += 2ll * c * hcnt[c];
        host_info[0] = hcnt[PDE_MAX_CLASS]; host_info[1] = evals;
        return 0;
    };
    if (host_info && !defer_host) { if (read_host_info()) return 1; }
    if (kept_out) HIPCK(hipMemcpyAsync(kept_out, L.flags, (size_t)P, hipMemcpyDeviceToDevice, st));
    bool finished = false;
    for (int64_t first = 0; first < P; first += L.chunk) {
        const int64_t cap = P - first < L.chunk ? (P - first + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES : L.chunk;
        PdeJetArgs ja; memset(&ja, 0, sizeof(ja));
        ja.Wv = VW; ja.Wa = AW; ja.qorig = L.qorig; ja.klist = L.klist; ja.first = first; ja.kcount = L.kcount; ja.cap = cap;
        ja.stash = L.stash; ja.seeds = L.seeds; ja.sums = L.sums; ja.scale = loss_scale; ja.scale_dev = loss_scale_dev; ja.jac = jac_out; ja.n_jac = n_jac;
        const unsigned wgs = (unsigned)(cap / WG_SAMPLES);
        ja.wgs = (int)wgs;
        ja.wout = L.wout;
        for (int l = 0; l < 6; ++l) { ja.f4[l] = f4[l]; ja.t4[l] = t4[l]; ja.bv[l] = VW.b[l]; }
        // the fused adjoint is the only reader of the hidden layers' pre-activations: they travel as x4 stash blocks (row-major for the unfused adjoint)
        const bool fuse_ok = pde_fuse && grads && grads->vW[1] && grads->vW[2] && grads->vW[3] && grads->vW[4];
        ja.x4 = fuse_ok ? 1 : 0;
        {
            ProfScope ps(PK_PDE_FWD, st);
            {
                // all five weight_net columns of a tile in one workgroup; the ReLU acceleration net keeps its column kernel
                // (the launch carries the acceleration net's value column as trailing workgroups: they fill the tail of the jet tiles)
                // NVFI_PDE_JET_X6 (default 1, round 6): pde_jet6.hip - the same program with the hidden layers' products on the 16-bit matrix pipe
                static int jet_x6 = -1;
                if (jet_x6 < 0) { const char* e = getenv("NVFI_PDE_JET_X6"); jet_x6 = e ? atoi(e) : 1; }
                if (jet_x6) {
                    if (!cached && pre16 != 4 && first == 0 && launch_pack_x6(f->vW, L.x6img, st)) return 1;      // (the x6 prefilter has packed it already)
                    if (launch_pde_jet6_fwd(ja, cached ? FC.vel_x6 : L.x6img, (unsigned)(cap / TILE), wgs, st)) return 1;
                } else if (launch_pde_jet_fwd(ja, (unsigned)(cap / TILE), wgs, st)) return 1;
            }
            if (fl) {
                // the last workgroup of k_pde_seeds does the bookkeeping launches' work (pde_tail); out / counters in the last pass of the call only
                const bool last_pass = first + L.chunk >= P;
                ja.tail.ticket = L.cls_count + PDE_MAX_CLASS + 2; ja.tail.dcount = grads ? L.dcount : nullptr;
                ja.tail.out = last_pass ? out : nullptr;
                ja.tail.counters = last_pass ? counters : nullptr; ja.tail.cls_count = L.cls_count; ja.tail.P = P; ja.tail.pre16 = (pre16 == 1 || pre16 == 3) ? 1 : 0;
                if (last_pass) finished = true;
            }
            hipLaunchKernelGGL(k_pde_seeds, dim3((unsigned)(cap / 256 + 1)), dim3(256), 0, st, ja);
        }
        if (grads) {
            hipStream_t sb = st;
            if (t_bwd_stream && P <= L.chunk) {      // split call: the value is finished on st first, the adjoint half follows on the other stream
                if (!fl) hipLaunchKernelGGL(k_pde_finish, dim3(1), dim3(64), 0, st, L.sums, L.kcount, out);
                // the host's copy of the counts is queued IN FRONT of the adjoint half: k_pde_fuse_bwd owns every CU for ~0.4 ms (12 waves x 168
                // registers, 152 KB of LDS per workgroup) and a copy kernel queued behind its start waits for its first workgroup to retire - the
                // caller's `if loss_vel > 0` then returned 0.43 ms late (trace of the drop-in loop)
                if (defer_host) { HIPCK(hipMemcpyAsync(hcnt_early, L.cls_count, sizeof(int) * (PDE_MAX_CLASS + 16), hipMemcpyDeviceToHost, st)); early_copy = true; }
                HIPCK(hipEventRecord(t_split_ev, st));
                HIPCK(hipStreamWaitEvent(t_bwd_stream, t_split_ev, 0));
                sb = t_bwd_stream;
                finished = true;
            }
            int fused_nslab = 0, fused_accel = 0;
            float* fused_slabs = L.slabs + (size_t)14 * PDE_NSLAB * (128 * 128 + 128);      // sets 14..17 (launch_pde_wgrad uses at most 10 beside them)
            float* fused_slabs_a = L.slabs + (size_t)10 * PDE_NSLAB * (128 * 128 + 128);    // sets 10..13 (at most 6 ring jobs are left then)
            // (before the adjoint: the fused kernel's queue word is cleared by the same launch)
            if (!fl) hipLaunchKernelGGL(k_pde_pass_count, dim3(1), dim3(64), 0, sb, L.kcount, first, cap, L.dcount);
            {
                ProfScope ps(PK_PDE_BWD, sb);
                if (fuse_ok) {
                    // pde_fuse.hip: the adjoint of weight_net's five columns AND its four hidden-layer weight gradients in one persistent kernel;
                    // the acceleration net's adjoint keeps k_pde_jet_bwd's trailing workgroups (a launch with zero jet tiles)
                    PdeFuseArgs fa; memset(&fa, 0, sizeof(fa));
                    for (int l = 0; l < 6; ++l) { fa.t4[l] = t4[l]; fa.ta4[l] = ta4[l]; }
                    fa.kcount = L.kcount; fa.first = first; fa.cap = cap; fa.stash = L.stash; fa.seeds = L.seeds; fa.x4 = ja.x4;
                    fa.slabs = fused_slabs; fa.layer_stride = (int64_t)PDE_NSLAB * (128 * 128 + 128); fa.slab_floats = 128 * 128 + 128;
                    // ... and, when all four hidden-layer gradients of a_weight_net are wanted, its adjoint + those gradients as the kernel's second
                    // half (tiles from a device-side queue); otherwise the acceleration net's adjoint keeps k_pde_jet_bwd's trailing workgroups
                    fa.do_accel = (grads->aW[1] && grads->aW[2] && grads->aW[3] && grads->aW[4]) ? 1 : 0;
                    fa.slabs_a = fused_slabs_a;
                    static int det = -1;
                    if (det < 0) { const char* e = getenv("NVFI_DETERMINISTIC"); det = (e && atoi(e) != 0) ? 1 : 0; }
                    fa.queue = det ? nullptr : L.dcount + 8;
                    if (!fa.do_accel && launch_pde_jet_bwd(ja, 0, wgs, sb)) return 1;
                    // a split call (the reference's loop: the caller waits for the value - `if loss_vel > 0` - while this half runs on the other
                    // stream) leaves 8 CUs to the caller's comparison / copy kernels: a persistent workgroup owns its CU, and behind 256 of them
                    // that wait was 0.35 ms
                    const int max_wgs = sb != st ? PDE_NSLAB - 8 : PDE_NSLAB;
                    if (launch_pde_fuse_bwd(fa, cap, max_wgs, &fused_nslab, sb)) return 1;
                    fused_accel = fa.do_accel;
                } else if (launch_pde_jet_bwd(ja, (unsigned)(cap / TILE), wgs, sb)) return 1;
            }
            LAUNCHCK();
            if (launch_pde_wgrad(L.stash, (int)(cap / TILE), L.slabs, L.dcount, grads, sb, fused_nslab, fused_slabs, fused_accel ? fused_slabs_a : nullptr)) return 1;
            if (defer_host) { if (read_host_info()) return 1; }
        }
        LAUNCHCK();
    }
    if (!finished) hipLaunchKernelGGL(k_pde_finish, dim3(1), dim3(64), 0, st, L.sums, L.kcount, out);
    LAUNCHCK();
    if (counters && !fl) {
        hipLaunchKernelGGL(k_pde_counters, dim3(1), dim3(64), 0, st, L.cls_count, L.kcount, P, pre16 == 1 || pre16 == 3, counters);
        LAUNCHCK();
    }
    return 0;
}

extern "C" int nvfi_pde_loss(const nvfi_field_desc* f, int64_t P, const float* points, const float* t, float loss_scale, float* out,
                             const nvfi_grads* grads, void* workspace, int64_t workspace_bytes, int64_t* counters, void* stream) {
    return nvfi_pde_loss_ex(f, P, points, t, loss_scale, out, grads, workspace, workspace_bytes, counters, nullptr, nullptr, 0, nullptr, stream);
}

// weight gradients of both nets from one chunk's stash
// fused_nslab > 0: the slabs of weight_net's four hidden layers were already written (fused_nslab of them each, value + tangent columns
// summed) by k_pde_fuse_bwd - only its two edge layers and the acceleration net are contracted here, the reduce covers everything
// (fused_slabs_a: likewise for a_weight_net, when the kernel's second half ran)
int launch_pde_wgrad(const float* stash, int ntiles, float* slabs, int* dcount, const nvfi_grads* G, hipStream_t st, int fused_nslab, const float* fused_slabs,
                     const float* fused_slabs_a) {
    const size_t ts = (size_t)PDE_TILE_ROWS * REGF;
    const size_t slab = (size_t)PDE_NSLAB * (128 * 128 + 128);
    // k_wgrad reads the sample count from device memory (dcount, set by k_pde_pass_count: no host sync); ntiles is the capacity
    WgradJobs wj; wj.n = 0; ReduceJobs rj; rj.n = 0;
    for (int net = 0; net < 2; ++net)
        for (int l = 0; l < 6; ++l) {
            float* gW = net == 0 ? G->vW[l] : G->aW[l];
            float* gb = net == 0 ? G->vb[l] : G->ab[l];
            if (!gW && !gb) continue;
            if (fused_nslab > 0 && l >= 1 && l <= 4 && (net == 0 || fused_slabs_a)) {
                ReduceJob& Q = rj.j[rj.n++];
                memset(&Q, 0, sizeof(Q));
                Q.slabs = (net == 0 ? fused_slabs : fused_slabs_a) + (size_t)(l - 1) * slab; Q.nslab = fused_nslab; Q.MTA = 4; Q.KTB = 4; Q.gW = gW; Q.gb = gb;
                Q.out = 128; Q.in = 128; Q.row_kind = RK_NATURAL; Q.slot_kind = SK_HIDDEN; Q.scale = 1.f;
                continue;
            }
            const int npass = net == 0 ? 2 : 1;    // weight_net: value column + 4 tangent columns
            float* sl[2] = {nullptr, nullptr};
            for (int pass = 0; pass < npass; ++pass) {
                WgradJob& J = wj.j[wj.n];
                memset(&J, 0, sizeof(J));
                J.a_tile_stride = ts; J.b_tile_stride = ts; J.a_regs = l < 5 ? 64 : 16; J.count = dcount; J.cap_tiles = ntiles;
                J.slabs = slabs + (size_t)wj.n * slab; J.nslab = PDE_NSLAB;
                sl[pass] = J.slabs;
                ++wj.n;
                if (net == 1) {               // a_weight_net
                    J.A = stash + (size_t)(PDE_GAA + l * 64) * REGF; J.nrep = 1;
                    if (l == 0) { J.B = stash + (size_t)PDE_X0 * REGF; J.b_regs = 16; J.bmode = BM_RAW; }
                    else { J.B = stash + (size_t)(PDE_ZA + (l - 1) * 64) * REGF; J.b_regs = 64; J.bmode = BM_RELU; }
                } else if (pass == 0) {       // value column of weight_net
                    J.A = stash + (size_t)(PDE_GA + l * 64) * REGF; J.nrep = 1;
                    if (l == 0) { J.B = stash + (size_t)PDE_X0 * REGF; J.b_regs = 16; J.bmode = BM_RAW; }
                    else { J.B = stash + (size_t)(PDE_Z + (l - 1) * 64) * REGF; J.b_regs = 64; J.bmode = BM_SILU; }
                } else {                      // 4 tangent columns (no bias term)
                    J.A = stash + (size_t)(PDE_GA + 336 + l * 64) * REGF; J.nrep = 4; J.a_rep_stride = (size_t)336 * REGF;
                    if (l == 0) { J.B = stash + (size_t)PDE_X0D * REGF; J.b_regs = 16; J.bmode = BM_RAW; J.b_rep_stride = (size_t)16 * REGF; }
                    else {
                        J.B = stash + (size_t)(PDE_Z + (l - 1) * 64) * REGF; J.b_rep_stride = 0;
                        J.B2 = stash + (size_t)(PDE_ZD + (l - 1) * 64) * REGF; J.b2_rep_stride = (size_t)320 * REGF;
                        J.b_regs = 64; J.bmode = BM_SILU_TAN;
                    }
                }
            }
            const int a_regs = l < 5 ? 64 : 16, b_regs = l == 0 ? 16 : 64;
            // weights: value (+ tangent) slabs; bias: value slabs only (tangent columns carry no bias term)
            ReduceJob& Q = rj.j[rj.n++];
            memset(&Q, 0, sizeof(Q));
            Q.slabs = sl[0]; Q.nslab = PDE_NSLAB; Q.slabs2 = sl[1]; Q.nslab2 = sl[1] ? PDE_NSLAB : 0;
            Q.MTA = a_regs / 16; Q.KTB = b_regs / 16; Q.gW = gW; Q.gb = nullptr;
            Q.out = l < 5 ? 128 : 6; Q.in = l == 0 ? 28 : 128; Q.row_kind = RK_NATURAL; Q.slot_kind = l == 0 ? SK_VEL_IN : SK_HIDDEN; Q.scale = 1.f;
            if (gb) {
                ReduceJob& Qb = rj.j[rj.n++];
                Qb = Q; Qb.slabs2 = nullptr; Qb.nslab2 = 0; Qb.gW = nullptr; Qb.gb = gb;
            }
        }
    if (launch_wgrad(wj, rj, st)) return 1;
    return 0;
}
