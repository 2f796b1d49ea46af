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
template <bool UNIFORM, bool STASH>
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
        float* zst1 = nullptr; float* zst2 = nullptr; float* x0s1 = nullptr; float* x0s2 = nullptr;
        if (STASH) {
            size_t e1 = ((size_t)(2 * s) * a.cap_tiles + tile), e2 = ((size_t)(2 * s + 1) * a.cap_tiles + tile);
            zst1 = a.zst + e1 * (VEL_Z_REGS * REGF); zst2 = a.zst + e2 * (VEL_Z_REGS * REGF);
            x0s1 = a.x0st + e1 * (VEL_X0_REGS * REGF); x0s2 = a.x0st + e2 * (VEL_X0_REGS * REGF);
        }
        float o4[4], w1[6], w2[6], v1[3], v2[3];
        // v1 = vel(x, t)
        velnet_forward<1, STASH>(a.Wv, lds_w, lds_b, lane, make_float4(x, y, z, tcur), zst1, x0s1, o4);
        gather6(o4, h, w1);
        vel_from_w(w1, x, y, z, v1);
        const bool g1 = gated_out(a.f, x, y, z);
        if (g1) { v1[0] = v1[1] = v1[2] = 0.f; }
        // midpoint
        const float hdt = 0.5f * dt;
        const float px = x - hdt * v1[0], py = y - hdt * v1[1], pz = z - hdt * v1[2];
        const float tm = tcur - hdt;
        velnet_forward<1, STASH>(a.Wv, lds_w, lds_b, lane, make_float4(px, py, pz, tm), zst2, x0s2, o4);
        gather6(o4, h, w2);
        vel_from_w(w2, px, py, pz, v2);
        const bool g2 = gated_out(a.f, px, py, pz);
        if (g2) { v2[0] = v2[1] = v2[2] = 0.f; }
        const float nx = x - dt * v2[0], ny = y - dt * v2[1], nz = z - dt * v2[2];
        const bool rej = a.f.gate_sur && gated_out(a.f, nx, ny, nz);   // tensorf_keyframe.py:603-605
        if (STASH && active && h == 0) {
            float* rc = a.rec + (size_t)s * RK_NF * a.cap + i;
            rc[0 * a.cap] = x; rc[1 * a.cap] = y; rc[2 * a.cap] = z;
            rc[3 * a.cap] = px; rc[4 * a.cap] = py; rc[5 * a.cap] = pz;
#pragma unroll
            for (int k = 0; k < 6; ++k) { rc[(6 + k) * a.cap] = w1[k]; rc[(12 + k) * a.cap] = w2[k]; }
            rc[18 * a.cap] = __int_as_float((g1 ? 1 : 0) | (g2 ? 2 : 0) | (rej ? 4 : 0));
        }
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

// ---------------------------------------------------------------- RK2 adjoint (render backward)
__global__ __launch_bounds__(WG_THREADS, 1) void k_rk2_bwd(Rk2Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int count = *a.count;
    if ((int)(blockIdx.x * WG_SAMPLES) >= count) return;
    FragPipe pipe; pipe.init(lds);
    pipe.issue(a.Wv.t[5], VEL_T5);
    pipe.cur = 1; pipe.commit();          // first fragment into buffer 0
    const int tile = blockIdx.x * 4 + wave_id();
    const int i = tile * TILE + (lane & 31);
    const bool active = i < count;
    float4 gin = active ? a.gxk[a.list[i]] : zero4();     // upstream gradient of the warped position, stored per sample
    float g3[3] = {gin.x, gin.y, gin.z};
#pragma unroll 1
    for (int s = a.nsteps - 1; s >= 0; --s) {
        const float dt = RK_DT(a, s), tcur = RK_TC(a, s);
        const float* rc = a.rec + (size_t)s * RK_NF * a.cap + (active ? i : 0);
        const int flags = active ? __float_as_int(rc[18 * a.cap]) : 7;
        const bool g1 = flags & 1, g2 = flags & 2, rej = flags & 4;
        const size_t e1 = ((size_t)(2 * s) * a.cap_tiles + tile), e2 = ((size_t)(2 * s + 1) * a.cap_tiles + tile);
        // The two network evaluations of the step are walked in reverse by ONE loop body (e = 1: v2 = vel(pmid, tmid),
        // x_new = x - dt*v2;  e = 0: v1 = vel(x, t), pmid = x - dt/2*v1); records are re-read around the MLP pass
        // instead of being kept live - the two 64-register adjoint arrays need the room.
        float gacc[3] = {0.f, 0.f, 0.f};        // gpm after e=1, then + gx1
        float gup[3] = {g3[0], g3[1], g3[2]};   // upstream of the evaluation being processed
#pragma unroll 1
        for (int e = 1; e >= 0; --e) {
            const int po = e ? 3 : 0, wo = e ? 12 : 6;
            const float coef = e ? -dt : -0.5f * dt;
            const float te = e ? tcur - 0.5f * dt : tcur;
            const bool gate = e ? g2 : g1;
            const size_t es = ((size_t)(2 * s + e) * a.cap_tiles + tile);
            float p[3], w[6], gv[3], gw[6], r4[4], gloc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = active ? rc[(po + c) * a.cap] : 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) w[k] = active ? rc[(wo + k) * a.cap] : 0.f;
            const bool on = active && !rej && !gate;
#pragma unroll
            for (int c = 0; c < 3; ++c) gv[c] = on ? coef * gup[c] : 0.f;
            gw[0] = gv[0]; gw[1] = gv[1]; gw[2] = gv[2];
            gw[3] = p[2] * gv[1] - p[1] * gv[2];
            gw[4] = -p[2] * gv[0] + p[0] * gv[2];
            gw[5] = p[1] * gv[0] - p[0] * gv[1];
            gloc[0] = -w[5] * gv[1] + w[4] * gv[2];
            gloc[1] = w[5] * gv[0] - w[3] * gv[2];
            gloc[2] = -w[4] * gv[0] + w[3] * gv[1];
            scatter6(gw, h, r4);
            float ge[16];
            velnet_backward_p<1>(a.Wv, pipe, lane, r4, a.zst + es * (VEL_Z_REGS * REGF), a.gst + es * (VEL_G_REGS * REGF), ge, a.Wv.t[5], VEL_T5);
            float x0[16];
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = active ? rc[(po + c) * a.cap] : 0.f;
            vel_encode_slots(make_float4(p[0], p[1], p[2], te), h, x0);
            const float4 gq = vel_encode_bwd(ge, x0, h);
            gloc[0] += gq.x; gloc[1] += gq.y; gloc[2] += gq.z;
#pragma unroll
            for (int c = 0; c < 3; ++c) { gacc[c] += gloc[c]; gup[c] = gloc[c]; }
        }
        float gpm[3] = {gacc[0], gacc[1], gacc[2]}, gx1[3] = {0.f, 0.f, 0.f};
        if (active && !rej) {
#pragma unroll
            for (int c = 0; c < 3; ++c) g3[c] = g3[c] + gpm[c] + gx1[c];
        }
    }
}

// ---------------------------------------------------------------- host-side helpers
int ensure_lds_attrs() {
    static bool done = false;
    if (done) return 0;
    HIPCK(hipFuncSetAttribute((const void*)k_vel_eval, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_rk2_fwd<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_rk2_fwd<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_rk2_fwd<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_rk2_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, ENGINE2_LDS_BYTES));
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
int launch_rk2_fwd(const Rk2Args& a, int64_t cap_samples, bool uniform, bool stash, hipStream_t st) {
    if (ensure_lds_attrs()) return 1;
    int64_t nwg = (cap_samples + WG_SAMPLES - 1) / WG_SAMPLES;
    if (nwg <= 0) return 0;
    dim3 g((unsigned)nwg), b(WG_THREADS);
    ProfScope ps(uniform ? PK_RK2_FWD : PK_PDE_PREFILTER, st);
    if (uniform && stash) hipLaunchKernelGGL((k_rk2_fwd<true, true>), g, b, ENGINE_LDS_BYTES, st, a);
    else if (uniform) hipLaunchKernelGGL((k_rk2_fwd<true, false>), g, b, ENGINE_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((k_rk2_fwd<false, false>), g, b, ENGINE_LDS_BYTES, st, a);
    LAUNCHCK();
    return 0;
}
int launch_rk2_bwd(const Rk2Args& a, int64_t cap_samples, hipStream_t st) {
    if (ensure_lds_attrs()) return 1;
    int64_t nwg = (cap_samples + WG_SAMPLES - 1) / WG_SAMPLES;
    if (nwg <= 0) return 0;
    ProfScope ps(PK_RK2_BWD, st);
    hipLaunchKernelGGL(k_rk2_bwd, dim3((unsigned)nwg), dim3(WG_THREADS), ENGINE2_LDS_BYTES, st, a);
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
