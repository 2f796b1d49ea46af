// vel.hip - velocity-field kernels: VelBasis evaluation and the RK2 back-advection (forward + adjoint).
//
// Reference semantics: models/velocity_field.py:21-98 (VelocityAABB[Sur], VelBasis),
// models/tensorf_keyframe.py:575-611 (integrate_pos).  One workgroup = 4 waves = 128 samples; each
// wave runs the 6-layer MLP of its 32 samples on fp32 MFMA with register-resident activations
// (engine.h).  RK2 is fused around the two network evaluations of a step.
#include "common.h"
#include "vel.h"

// ---------------------------------------------------------------- plain evaluation (VelBasis.forward / gated get_vel)
__global__ __launch_bounds__(WG_THREADS, 2) void k_vel_eval(VelEvalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_w = lds; float* lds_b = lds + LDS_W_FLOATS;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave_id();
    const int64_t i = tile * TILE + (lane & 31);
    const bool active = i < a.N;
    float4 q = active ? reinterpret_cast<const float4*>(a.xt)[i] : zero4();
    float o4[4], w[6];
    velnet_forward<1>(a.Wv, lds_w, lds_b, lane, q, nullptr, nullptr, o4);
    gather6(o4, h, w);
    float v[3];
    vel_from_w(w, q.x, q.y, q.z, v);
    if (a.gated) {
        if (gated_out(a.f, q.x, q.y, q.z)) { v[0] = v[1] = v[2] = 0.f; }
        if (active && h == 0) { a.u6[6 * i] = v[0]; a.u6[6 * i + 1] = v[1]; a.u6[6 * i + 2] = v[2]; }
        return;
    }
    float aw[6], acc3[3];
    velnet_forward<0>(a.Wa, lds_w, lds_b, lane, q, nullptr, nullptr, o4);
    gather6(o4, h, aw);
    acc_from_w(aw, q.x, q.y, q.z, acc3);
    if (active && h == 0) {
        a.u6[6 * i] = v[0]; a.u6[6 * i + 1] = v[1]; a.u6[6 * i + 2] = v[2];
        a.u6[6 * i + 3] = acc3[0]; a.u6[6 * i + 4] = acc3[1]; a.u6[6 * i + 5] = acc3[2];
    }
}

// ---------------------------------------------------------------- RK2 forward
// UNIFORM: every sample takes the same (dt_s, t_s) sequence (render path, t is a per-call scalar);
// otherwise per-point t/base (integrate_pos API, PDE prefilter) with a workgroup-uniform loop.
// (the stash-writing form of the training warp and the adjoint k_rk2_bwd that stood here were retired in round 6: vel_split.hip / vel_x6.hip /
// vel_fuse.hip own the training path; this kernel serves the fp32 escape of integrate_pos / compute_alpha and the band re-evaluation)
template <bool UNIFORM>
__global__ __launch_bounds__(WG_THREADS, 2) void k_rk2_fwd(Rk2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_w = lds; float* lds_b = lds + LDS_W_FLOATS;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int count = a.count ? *a.count : (int)a.n_direct;
    const int wg_first = blockIdx.x * WG_SAMPLES;
    if (wg_first >= count) return;                       // workgroup-uniform exit
    const int tile = blockIdx.x * 4 + wave_id();
    const int i = tile * TILE + (lane & 31);
    const bool active = i < count;
    const int n = active ? (a.list ? a.list[i] : i) : 0;
    float4 q0 = active ? a.xw[n] : zero4();
    float x = q0.x, y = q0.y, z = q0.z;
    float off = 0.f, tcur = 0.f;
    if (!UNIFORM) {
        const int ti = a.pt_by_list ? n : i;
        tcur = active ? a.pt_t[ti] : 0.f;
        off = active ? tcur - a.pt_base[ti] : 0.f;
    }
    const int nsteps = UNIFORM ? a.nsteps : a.max_steps;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        float dt;
        if (UNIFORM) { dt = RK_DT(a, s); tcur = RK_TC(a, s); }
        else {
            bool unfinished = fabsf(off) > 0.f;
            if (!__syncthreads_or(unfinished)) break;
            float m = fminf(fabsf(off), a.dt_max);
            dt = off > 0.f ? m : (off < 0.f ? -m : 0.f);
        }
        const bool live = active && (UNIFORM || fabsf(off) > 0.f);
        float o4[4], w1[6], w2[6], v1[3], v2[3];
        // v1 = vel(x, t)
        velnet_forward<1>(a.Wv, lds_w, lds_b, lane, make_float4(x, y, z, tcur), nullptr, nullptr, o4);
        gather6(o4, h, w1);
        vel_from_w(w1, x, y, z, v1);
        const bool g1 = gated_out(a.f, x, y, z);
        if (g1) { v1[0] = v1[1] = v1[2] = 0.f; }
        // midpoint
        const float hdt = 0.5f * dt;
        const float px = x - hdt * v1[0], py = y - hdt * v1[1], pz = z - hdt * v1[2];
        const float tm = tcur - hdt;
        velnet_forward<1>(a.Wv, lds_w, lds_b, lane, make_float4(px, py, pz, tm), nullptr, nullptr, o4);
        gather6(o4, h, w2);
        vel_from_w(w2, px, py, pz, v2);
        const bool g2 = gated_out(a.f, px, py, pz);
        if (g2) { v2[0] = v2[1] = v2[2] = 0.f; }
        const float nx = x - dt * v2[0], ny = y - dt * v2[1], nz = z - dt * v2[2];
        const bool rej = a.f.gate_sur && gated_out(a.f, nx, ny, nz);   // tensorf_keyframe.py:603-605
        if (live && !rej) { x = nx; y = ny; z = nz; }
        if (!UNIFORM) {
            if (live) { off = off - dt; tcur = tcur - dt; }
        }
    }
    if (active && h == 0) {
        if (a.xout) { a.xout[3 * (size_t)i] = x; a.xout[3 * (size_t)i + 1] = y; a.xout[3 * (size_t)i + 2] = z; }
        else a.xw[n] = make_float4(x, y, z, q0.w);
    }
}

// ---------------------------------------------------------------- host-side helpers
int ensure_lds_attrs() {
    static bool done = false;
    if (done) return 0;
    HIPCK(hipFuncSetAttribute((const void*)k_vel_eval, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_rk2_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_rk2_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    done = true;
    return 0;
}

int launch_vel_eval(const VelEvalArgs& a, hipStream_t st) {
    if (ensure_lds_attrs()) return 1;
    int64_t nwg = (a.N + WG_SAMPLES - 1) / WG_SAMPLES;
    if (nwg <= 0) return 0;
    hipLaunchKernelGGL(k_vel_eval, dim3((unsigned)nwg), dim3(WG_THREADS), ENGINE_LDS_BYTES, st, a);
    LAUNCHCK();
    return 0;
}
int launch_rk2_fwd(const Rk2Args& a, int64_t cap_samples, bool uniform, hipStream_t st) {
    if (ensure_lds_attrs()) return 1;
    int64_t nwg = (cap_samples + WG_SAMPLES - 1) / WG_SAMPLES;
    if (nwg <= 0) return 0;
    dim3 g((unsigned)nwg), b(WG_THREADS);
    ProfScope ps(uniform ? PK_RK2_FWD : PK_PDE_PREFILTER, st);
    if (uniform) hipLaunchKernelGGL((k_rk2_fwd<true>), g, b, ENGINE_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((k_rk2_fwd<false>), g, b, ENGINE_LDS_BYTES, st, a);
    LAUNCHCK();
    return 0;
}

// fragment packing jobs for one VelBasis net
int pack_vel_frags(const float* const* W, const float* const* b, float* buf, VelFrags* out, PackJobs* jobs) {
    float* p = buf;
    auto add = [&](const float* Wl, const float* bl, float* frag, float* bfrag, int o, int in, int MT, int NS, int rk, int sk, int tr) {
        if (jobs->n >= MAX_PACK_JOBS) return 1;
        PackJob& J = jobs->j[jobs->n++];
        J.W = Wl; J.b = bl; J.frag = frag; J.bfrag = bfrag; J.out = o; J.in = in; J.MT = MT; J.NS = NS;
        J.row_kind = rk; J.slot_kind = sk; J.transposed = tr; J.x4 = 0;
        return 0;
    };
    float* f0 = p; p += VEL_F0;
    float* fh[4]; for (int l = 0; l < 4; ++l) { fh[l] = p; p += VEL_FH; }
    float* f5 = p; p += VEL_F5;
    float* bf[6]; for (int l = 0; l < 6; ++l) { bf[l] = p; p += 128; }
    float* t0 = p; p += VEL_T0;
    float* th[4]; for (int l = 0; l < 4; ++l) { th[l] = p; p += VEL_FH; }
    float* t5 = p; p += VEL_T5;
    int rc = 0;
    rc |= add(W[0], b[0], f0, bf[0], 128, 28, 4, 14, RK_NATURAL, SK_VEL_IN, 0);
    for (int l = 1; l <= 4; ++l) rc |= add(W[l], b[l], fh[l - 1], bf[l], 128, 128, 4, 64, RK_NATURAL, SK_HIDDEN, 0);
    rc |= add(W[5], b[5], f5, bf[5], 6, 128, 1, 64, RK_NATURAL, SK_HIDDEN, 0);
    rc |= add(W[0], nullptr, t0, nullptr, 128, 28, 1, 64, RK_VEL_IN, SK_HIDDEN, 1);
    for (int l = 1; l <= 4; ++l) rc |= add(W[l], nullptr, th[l - 1], nullptr, 128, 128, 4, 64, RK_NATURAL, SK_HIDDEN, 1);
    rc |= add(W[5], nullptr, t5, nullptr, 6, 128, 4, 4, RK_NATURAL, SK_HIDDEN, 1);
    if (rc) return nvfi_fail(3, "too many pack jobs");
    out->f[0] = f0; out->t[0] = t0; out->f[5] = f5; out->t[5] = t5;
    for (int l = 1; l <= 4; ++l) { out->f[l] = fh[l - 1]; out->t[l] = th[l - 1]; }
    for (int l = 0; l < 6; ++l) out->b[l] = bf[l];
    return 0;
}
