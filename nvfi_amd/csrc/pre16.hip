// pre16.hip - OPT-IN fp16-input pre-pass of the PDE occupancy prefilter (NVFI_PDE_PREFILTER=fp16band; fp32 stays the default).
//
// The prefilter (reference models/nvfi.py:50-64) only DECIDES which collocation points take part in the PDE loss: RK2
// back-advection (models/tensorf_keyframe.py:575-611), density at the warped point, alpha >= alphaMask_thres.  The decision is
// insensitive to small position errors except for points whose alpha lies next to the threshold or whose trajectory passes next
// to a face of the velocity gate (velocity_field.py:28-33,46-51), where a rounding difference flips a branch.  So:
//   1. k_rk2_pre16 back-advects EVERY candidate with v_mfma_f32_32x32x16_f16 (weights and layer inputs rounded to fp16, fp32
//      accumulation; encoder, SiLU, basis combination and the RK2 arithmetic in fp32) and flags the points that came within
//      `eps_gate` of a gate face (or of the rejection box of VelocityAABBSur) at any stage;
//   2. pde.hip evaluates the density there, puts the flagged points and those with |alpha / thres - 1| <= band on a list,
//      re-runs the fp32 kernel (k_rk2_fwd, vel.hip) for that list only and takes their density from the fp32 position.
// Every kept/dropped decision therefore comes either from the fp32 path or from an alpha that is further from the threshold
// than the band (measured fp16 deviation of alpha: <= 1.2 % on 3 x 10^6 points, tests/studies/prefilter_fp16_study.py; default
// band 10 %).  The kept points' jets, loss and gradients are computed in fp32 from the ORIGINAL coordinates as before.
//
// MI355X layout: in fp16 the whole 6-layer net is 144 KB of fragments + 3 KB of biases, so it is staged into the CU's 160 KB LDS
// ONCE per workgroup and there is no barrier after that: 8 waves (2 per SIMD) walk their own 32-point tiles independently, one
// wave's SiLU/encoder VALU work overlapping the other's MFMAs.  The kernel is bound by the SiLU transcendentals, not by MFMA.
#include "common.h"
#include "vel.h"
#include "pde.h"
#include "engine16.h"

#define P16_THREADS 512
#define SPLIT_SCALE 2048.f                           // split mode: the second binary16 term of an operand is stored x 2^11
#define P16_L0 0                                   // 4 tiles x 2 K-steps x 64 lanes (h8 units)
#define P16_LH(l) (512 + ((l) - 1) * 2048)          // l = 1..4: 4 x 8 x 64
#define P16_L5 (512 + 4 * 2048)                     // 1 x 8 x 64
#define P16_H8 (512 + 4 * 2048 + 512)
static_assert(P16_H8 * 16 + 6 * 128 * 4 == PRE16_IMAGE_BYTES, "image size");

// ---------------------------------------------------------------- packing: the LDS image, built in global memory once per call
struct Pack16VelArgs { const float* W[6]; const float* b[6]; h8_t* img; h8_t* img_lo; };
__global__ __launch_bounds__(256) void k_pack_vel16(Pack16VelArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < P16_H8) {
        int l, local, NS, out, in, kind;
        if (idx < 512) { l = 0; local = idx; NS = 2; out = 128; in = 28; kind = SK_VEL_IN; }
        else if (idx < P16_L5) { l = 1 + (idx - 512) / 2048; local = (idx - 512) % 2048; NS = 8; out = 128; in = 128; kind = SK_HIDDEN; }
        else { l = 5; local = idx - P16_L5; NS = 8; out = 6; in = 128; kind = SK_HIDDEN; }
        const int lane = local & 63, ms = local >> 6, s = ms % NS, m = ms / NS;
        const int row = 32 * m + (lane & 31), h = lane >> 5;
        const float* W = a.W[l];
        h8_t v, vlo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int feat = slot_logical(kind, 2 * (8 * s + j) + h);     // register 8s + j of lane half h (engine.h)
            float w = 0.f;
            if (row < out && feat >= 0 && feat < in) w = W[(size_t)row * in + feat];
            v[j] = (_Float16)w;
            vlo[j] = (_Float16)((w - (float)v[j]) * SPLIT_SCALE);           // second binary16 term of the weight (split mode)
        }
        a.img[idx] = v;
        if (a.img_lo) a.img_lo[idx] = vlo;
    }
    if (idx < 6 * 128) {
        const int l = idx >> 7, row = idx & 127;
        float* bias = reinterpret_cast<float*>(a.img + P16_H8);
        bias[idx] = row < (l < 5 ? 128 : 6) ? a.b[l][row] : 0.f;
    }
}

// ---------------------------------------------------------------- one layer: MT output tiles, NS K-steps; the epilogue of tile
// m - 1 is issued behind the MFMAs of tile m (independent work for the scheduler, as in engine.h's layer_tiles).
//
// SPLIT (NVFI_PDE_PREFILTER=split16band): fp32 products emulated on the fp16 matrix pipe with TWO binary16 terms per operand,
//   x = xh + xl / 2048,  w = wh + wl / 2048   (xh = rn16(x), xl = rn16((x - xh) 2048): |x - xh - xl/2048| <= 2^-22 |x|, likewise w)
//   w x  ~  wh xh + (wh xl + wl xh) / 2048     - three MFMAs; the dropped wl xl term and the two representation errors are each <= 2^-22 |w x|,
// accumulated in fp32 like the fp32 MFMA accumulates its exact products: a relative error of ~2^-21 per product against the 2^-24 of one fp32
// rounding, i.e. the same order as the difference between two fp32 summation orders.  The hi fragments live in LDS as before; the lo
// fragments (a second 147 KB image) are read from L2 one row tile ahead (16 bytes per lane and MFMA: ~14 B / clk / CU).  Measured with the
// split arithmetic in place: 0.54 ms for the 1.0e6 evaluations of the prefilter against 0.305 ms for the one-term kernel and 1.31 ms for fp32 MFMA.
template <int MT, int NS, bool SPLIT, class Epi>
__device__ __forceinline__ void layer16p(const h8_t* w, const h8_t* __restrict__ wlo, const float* bias, int lane, int h, const h8_t* B, Epi epi) {
    f32x16 prev;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x16 acc;
        h8_t L[SPLIT ? NS : 1];
        if (SPLIT) {
            __builtin_amdgcn_sched_barrier(0);      // the lo fragments of ONE row tile in flight (hoisted over the four tiles they cost 128 registers and spill)
#pragma unroll
            for (int s = 0; s < NS; ++s) L[s] = wlo[(m * NS + s) * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias[32 * m + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
        for (int s = 0; s < NS; ++s) acc = MFMA16(w[(m * NS + s) * 64 + lane], B[s], acc);
        if (SPLIT) {
            f32x16 lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) lo[r] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) lo = MFMA16(w[(m * NS + s) * 64 + lane], B[NS + s], lo);
#pragma unroll
            for (int s = 0; s < NS; ++s) lo = MFMA16(L[s], B[s], lo);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(lo[r], 1.f / SPLIT_SCALE, acc[r]);
        }
        if (SPLIT) { epi(m, acc); __builtin_amdgcn_sched_barrier(0); continue; }   // (no deferred epilogue: its 16 registers are needed)
        if (m > 0) epi(m - 1, prev);
        prev = acc;
    }
    if (!SPLIT) epi(MT - 1, prev);
}
// one activation value -> its place in the B operands of the NEXT layer: register g (= 16 m + r of the D layout) is element g & 7 of K step
// g >> 3; split mode stores the second binary16 term NS K-steps further on
template <int NS, bool SPLIT>
__device__ __forceinline__ void put_h8(h8_t* B, int g, float v) {
    const _Float16 hi = (_Float16)v;
    B[g >> 3][g & 7] = hi;
    if (SPLIT) B[NS + (g >> 3)][g & 7] = (_Float16)((v - (float)hi) * SPLIT_SCALE);
}

// gated VelBasis.get_vel weights (velocity_field.py:77-93) of the wave's 32 points; out4 as velnet_forward (engine.h).
// The activations live ONLY as MFMA operands: the epilogue of a layer writes act(z) straight into the (hi [, lo]) binary16 operand registers
// of the next layer - two operand sets alternate, no fp32 activation array exists (the split mode needs its registers for the second terms).
// zst / x0st (training stash, or NULL): the fp32 pre-activations of the five hidden layers (row 64 l + 16 m + r of the tile's z block - the D
// layout of a 32x32 MFMA is the same for fp16 and fp32 inputs, so this IS the engine's stash layout) and the encoder slots
template <bool SPLIT = false>
__device__ __forceinline__ void velnet16(const h8_t* W, const h8_t* Wlo, const float* bias, int lane, int h, const float4& q, float* out4,
                                         float* zst = nullptr, float* x0st = nullptr) {
    constexpr int NB = SPLIT ? 16 : 8;
    h8_t Ba[NB], Bb[NB];
    {
        float x0[16];
        vel_encode_slots(q, h, x0);
        if (x0st) stash_store<16>(x0st, lane, x0);
#pragma unroll
        for (int g = 0; g < 16; ++g) put_h8<2, SPLIT>(Ba, g, x0[g]);
    }
    layer16p<4, 2, SPLIT>(W + P16_L0, Wlo + P16_L0, bias, lane, h, Ba, [&](int m, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { if (zst) STASH_ST(zst[(size_t)(16 * m + r) * REGF + lane], acc[r]); put_h8<8, SPLIT>(Bb, 16 * m + r, act_f<1>(acc[r])); }
    });
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        const int l = 1 + 2 * it;
        float* z1 = zst ? zst + (size_t)l * 64 * REGF : nullptr;
        float* z2 = zst ? zst + (size_t)(l + 1) * 64 * REGF : nullptr;
        layer16p<4, 8, SPLIT>(W + P16_LH(l), Wlo + P16_LH(l), bias + 128 * l, lane, h, Bb, [&](int m, const f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { if (z1) STASH_ST(z1[(size_t)(16 * m + r) * REGF + lane], acc[r]); put_h8<8, SPLIT>(Ba, 16 * m + r, act_f<1>(acc[r])); }
        });
        layer16p<4, 8, SPLIT>(W + P16_LH(l + 1), Wlo + P16_LH(l + 1), bias + 128 * (l + 1), lane, h, Ba, [&](int m, const f32x16& acc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { if (z2) STASH_ST(z2[(size_t)(16 * m + r) * REGF + lane], acc[r]); put_h8<8, SPLIT>(Bb, 16 * m + r, act_f<1>(acc[r])); }
        });
    }
    layer16p<1, 8, SPLIT>(W + P16_L5, Wlo + P16_L5, bias + 128 * 5, lane, h, Bb, [&](int, const f32x16& acc) {
        out4[0] = acc[0]; out4[1] = acc[1]; out4[2] = acc[2]; out4[3] = acc[3];
    });
}

__device__ __forceinline__ bool near_gate(const nvfi_field_desc& f, float eps, float x, float y, float z) {
    return fabsf(x - f.gate_lo[0]) < eps || fabsf(x - f.gate_hi[0]) < eps || fabsf(y - f.gate_lo[1]) < eps || fabsf(y - f.gate_hi[1]) < eps ||
           fabsf(z - f.gate_lo[2]) < eps || fabsf(z - f.gate_hi[2]) < eps;
}

// same recurrence as k_rk2_fwd<false, false> (vel.hip), per wave instead of per workgroup
template <bool SPLIT>
__global__ __launch_bounds__(P16_THREADS, 1) void k_rk2_pre16(Pre16Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const float4* src = reinterpret_cast<const float4*>(a.img);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < PRE16_IMAGE_BYTES / 16; i += P16_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    const h8_t* W = reinterpret_cast<const h8_t*>(lds);
    const h8_t* Wlo = reinterpret_cast<const h8_t*>(a.img_lo);
    const float* bias = lds + P16_H8 * 4;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * (P16_THREADS / 64) + (threadIdx.x >> 6);
    const int64_t i = tile * TILE + (lane & 31);
    if (tile * TILE >= a.P) return;                     // wave-uniform; no barrier follows
    const bool active = i < a.P;
    const int n = active ? a.list[i] : 0;
    const float4 q0 = active ? a.xw[n] : zero4();
    float x = q0.x, y = q0.y, z = q0.z;
    float tcur = active ? a.pt_t[i] : 0.f;
    float off = active ? tcur - a.pt_base[i] : 0.f;
    bool near = false;
#pragma unroll 1
    for (int s = 0; s < a.max_steps; ++s) {
        const bool live = fabsf(off) > 0.f;
        if (!__any(live)) break;
        const float m = fminf(fabsf(off), a.dt_max);
        const float dt = off > 0.f ? m : (off < 0.f ? -m : 0.f);
        float o4[4], w1[6], w2[6], v1[3], v2[3];
        velnet16<SPLIT>(W, Wlo, bias, lane, h, make_float4(x, y, z, tcur), o4);
        gather6(o4, h, w1);
        vel_from_w(w1, x, y, z, v1);
        if (gated_out(a.f, x, y, z)) { v1[0] = v1[1] = v1[2] = 0.f; }
        const float hdt = 0.5f * dt;
        const float px = x - hdt * v1[0], py = y - hdt * v1[1], pz = z - hdt * v1[2];
        velnet16<SPLIT>(W, Wlo, bias, lane, h, make_float4(px, py, pz, tcur - hdt), o4);
        gather6(o4, h, w2);
        vel_from_w(w2, px, py, pz, v2);
        if (gated_out(a.f, px, py, pz)) { v2[0] = v2[1] = v2[2] = 0.f; }
        const float nx = x - dt * v2[0], ny = y - dt * v2[1], nz = z - dt * v2[2];
        const bool rej = a.f.gate_sur && gated_out(a.f, nx, ny, nz);
        if (live) {
            // the fp16 trajectory drifts from the fp32 one step by step, so the guard distance grows with the number of steps walked
            // (round-2 advice: a fixed eps_gate has no error bound on a 10-step trajectory)
            const float eg = a.eps_gate * (float)(s + 1);
            near = near || near_gate(a.f, eg, x, y, z) || near_gate(a.f, eg, px, py, pz) ||
                   (a.f.gate_sur && near_gate(a.f, eg, nx, ny, nz));
            if (!rej) { x = nx; y = ny; z = nz; }
            off = off - dt; tcur = tcur - dt;
        }
    }
    if (active && h == 0) { a.xout[n] = make_float4(x, y, z, q0.w); a.near[n] = near ? 1 : 0; }
}

// ---------------------------------------------------------------- opt-in fp16-input INFERENCE mode of the velocity field
// nvfi_field_desc.vel_fp16 (off by default; the reference's counterpart is its autocast switch --disable_fp32, train_nvfi.py:96,144):
// every no-grad back-advection - integrate_pos (train_segm.py:150-166 walks up to 30 RK2 steps per point), the render warp of
// eval-mode renders, getDenseAlpha's 60 frame times - evaluates VelBasis on v_mfma_f32_32x32x16_f16 (weights and layer inputs rounded
// to binary16, fp32 accumulation; encoder, SiLU, basis combination, gates and the RK2 recurrence in fp32).  Same wave-per-tile kernel
// as the prefilter pre-pass above, with the two time modes of k_rk2_fwd: per-point (t, base) or a uniform step schedule.
// Training renders, the PDE regulariser and every gradient stay fp32.  tests/test_gpu_vel_fp16.py checks it against a CPU restatement
// in the same arithmetic (binary16-rounded operands, fp32 accumulation).
// STASH (uniform mode only): the training render's warp - the z / x0 stashes and the per-(step, sample) records of k_rk2_split_uni<STASH>
// (vel_split.hip), whole 128-sample groups (the adjoint and the weight-gradient kernels walk every tile of the last, ragged group)
template <bool UNI, bool SPLIT, bool STASH = false>
__global__ __launch_bounds__(P16_THREADS, 1) void k_rk2_inf16(Rk16Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const float4* src = reinterpret_cast<const float4*>(a.img);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int i = threadIdx.x; i < PRE16_IMAGE_BYTES / 16; i += P16_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    const h8_t* W = reinterpret_cast<const h8_t*>(lds);
    const h8_t* Wlo = reinterpret_cast<const h8_t*>(a.img_lo);
    const float* bias = lds + P16_H8 * 4;
    const int lane = threadIdx.x & 63, h = lane >> 5;
    int64_t P = a.P;
    if (a.count) { const int64_t c = *a.count; P = c < P ? c : P; }
    const int64_t tile = (int64_t)blockIdx.x * (P16_THREADS / 64) + (threadIdx.x >> 6);
    const int64_t i = tile * TILE + (lane & 31);
    if (tile * TILE >= (STASH ? (P + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES : P)) return;     // wave-uniform; no barrier follows
    const bool active = i < P;
    const int64_t n = active ? (a.list ? (int64_t)a.list[i] : i) : 0;
    const float4 q0 = active ? a.xw[n] : zero4();
    float x = q0.x, y = q0.y, z = q0.z;
    float tcur = (!UNI && active) ? a.pt_t[i] : 0.f;
    float off = (!UNI && active) ? tcur - a.pt_base[i] : 0.f;
    const int nsteps = UNI ? (a.sched ? __float_as_int(a.sched[2]) : a.nsteps) : a.max_steps;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        float dt;
        bool live;
        if (UNI) { dt = RK_DT(a, s); tcur = RK_TC(a, s); live = active; }
        else {
            live = fabsf(off) > 0.f;
            if (!__any(live)) break;
            const float m = fminf(fabsf(off), a.dt_max);
            dt = off > 0.f ? m : (off < 0.f ? -m : 0.f);
        }
        float o4[4], w1[6], w2[6], v1[3], v2[3];
        float *z1 = nullptr, *z2 = nullptr, *x1 = nullptr, *x2 = nullptr;
        if (STASH) {
            const size_t e1 = (size_t)(2 * s) * a.cap_tiles + tile, e2 = (size_t)(2 * s + 1) * a.cap_tiles + tile;
            z1 = a.zst + e1 * (VEL_Z_REGS * REGF); z2 = a.zst + e2 * (VEL_Z_REGS * REGF);
            x1 = a.x0st + e1 * (VEL_X0_REGS * REGF); x2 = a.x0st + e2 * (VEL_X0_REGS * REGF);
        }
        velnet16<SPLIT>(W, Wlo, bias, lane, h, make_float4(x, y, z, tcur), o4, z1, x1);
        gather6(o4, h, w1);
        vel_from_w(w1, x, y, z, v1);
        const bool g1 = gated_out(a.f, x, y, z);
        if (g1) { v1[0] = v1[1] = v1[2] = 0.f; }
        const float hdt = 0.5f * dt;
        const float px = x - hdt * v1[0], py = y - hdt * v1[1], pz = z - hdt * v1[2];
        velnet16<SPLIT>(W, Wlo, bias, lane, h, make_float4(px, py, pz, tcur - hdt), o4, z2, x2);
        gather6(o4, h, w2);
        vel_from_w(w2, px, py, pz, v2);
        const bool g2 = gated_out(a.f, px, py, pz);
        if (g2) { v2[0] = v2[1] = v2[2] = 0.f; }
        const float nx = x - dt * v2[0], ny = y - dt * v2[1], nz = z - dt * v2[2];
        const bool rej = a.f.gate_sur && gated_out(a.f, nx, ny, nz);
        if (STASH && active && h == 0) {
            float* rc = a.rec + (size_t)s * RK_NF * a.cap + i;
            rc[0 * a.cap] = x; rc[1 * a.cap] = y; rc[2 * a.cap] = z;
            rc[3 * a.cap] = px; rc[4 * a.cap] = py; rc[5 * a.cap] = pz;
#pragma unroll
            for (int k = 0; k < 6; ++k) { rc[(6 + k) * a.cap] = w1[k]; rc[(12 + k) * a.cap] = w2[k]; }
            rc[18 * a.cap] = __int_as_float((g1 ? 1 : 0) | (g2 ? 2 : 0) | (rej ? 4 : 0));
        }
        if (live) {
            if (!rej) { x = nx; y = ny; z = nz; }
            if (!UNI) { off = off - dt; tcur = tcur - dt; }
        }
    }
    if (active && h == 0) {
        if (a.xout3) { a.xout3[3 * n] = x; a.xout3[3 * n + 1] = y; a.xout3[3 * n + 2] = z; }
        else a.xout[n] = make_float4(x, y, z, q0.w);
    }
}

int launch_rk2_inf16(const nvfi_field_desc* f, Rk16Args a, bool uniform, hipStream_t st, bool stash) {
    static bool attr = false;
    if (!attr) {
        HIPCK(hipFuncSetAttribute((const void*)k_rk2_inf16<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PRE16_IMAGE_BYTES));
        HIPCK(hipFuncSetAttribute((const void*)k_rk2_inf16<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PRE16_IMAGE_BYTES));
        HIPCK(hipFuncSetAttribute((const void*)k_rk2_inf16<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PRE16_IMAGE_BYTES));
        HIPCK(hipFuncSetAttribute((const void*)k_rk2_inf16<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PRE16_IMAGE_BYTES));
        HIPCK(hipFuncSetAttribute((const void*)k_rk2_inf16<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PRE16_IMAGE_BYTES));
        attr = true;
    }
    if (a.P <= 0) return 0;
    const bool split = !stash && (f->vel_fp16 & 3) == 2;      // two binary16 terms per operand (fp32 products emulated): the lo image follows the hi image
    a.img_lo = split ? (char*)a.img + PRE16_IMAGE_BYTES : nullptr;
    Pack16VelArgs pk;
    for (int l = 0; l < 6; ++l) { pk.W[l] = f->vW[l]; pk.b[l] = f->vb[l]; }
    pk.img = reinterpret_cast<h8_t*>(a.img);
    pk.img_lo = reinterpret_cast<h8_t*>(a.img_lo);
    hipLaunchKernelGGL(k_pack_vel16, dim3((P16_H8 + 255) / 256), dim3(256), 0, st, pk);
    a.f = *f;
    const int64_t tiles = stash ? (a.P + WG_SAMPLES - 1) / WG_SAMPLES * (WG_SAMPLES / TILE) : (a.P + TILE - 1) / TILE;
    const unsigned wgs = (unsigned)((tiles + P16_THREADS / 64 - 1) / (P16_THREADS / 64));
    ProfScope ps(PK_RK2_FWD, st);
    if (stash) {
        if (!uniform) return nvfi_fail(2, "the fp16-input training stash exists for the uniform (render) warp only");
        hipLaunchKernelGGL((k_rk2_inf16<true, false, true>), dim3(wgs), dim3(P16_THREADS), PRE16_IMAGE_BYTES, st, a);
    } else if (uniform) {
        if (split) hipLaunchKernelGGL((k_rk2_inf16<true, true>), dim3(wgs), dim3(P16_THREADS), PRE16_IMAGE_BYTES, st, a);
        else hipLaunchKernelGGL((k_rk2_inf16<true, false>), dim3(wgs), dim3(P16_THREADS), PRE16_IMAGE_BYTES, st, a);
    } else {
        if (split) hipLaunchKernelGGL((k_rk2_inf16<false, true>), dim3(wgs), dim3(P16_THREADS), PRE16_IMAGE_BYTES, st, a);
        else hipLaunchKernelGGL((k_rk2_inf16<false, false>), dim3(wgs), dim3(P16_THREADS), PRE16_IMAGE_BYTES, st, a);
    }
    LAUNCHCK();
    return 0;
}

// alpha at the fp16-warped point: flags = 1 for the points whose decision is left to the fp32 pass.  Walks the points in BUCKET
// order (perm: most RK2 steps first), so the compacted list keeps the workgroups of the fp32 pass homogeneous in step count.
__global__ __launch_bounds__(256) void k_pde_band(nvfi_field_desc f, int64_t P, const int* perm, const float* sig, const uint8_t* near,
                                                  float band, uint8_t* flags, int* cnt) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool b = false;
    if (i < P) {
        const int n = perm[i];
        const float alpha = 1.f - expf(-sig[n] * 0.01f * 25.f);
        b = near[n] || fabsf(alpha - f.alpha_thres) <= band * f.alpha_thres || !(alpha == alpha);
        flags[i] = b ? 1 : 0;
    }
    const unsigned long long m = __ballot(b);
    if ((threadIdx.x & 63) == 0 && i < P) cnt[i >> 6] = __popcll(m);
}

int launch_pre16(const nvfi_field_desc* f, Pre16Args a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIPCK(hipFuncSetAttribute((const void*)k_rk2_pre16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PRE16_IMAGE_BYTES));
        HIPCK(hipFuncSetAttribute((const void*)k_rk2_pre16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PRE16_IMAGE_BYTES));
        attr = true;
    }
    Pack16VelArgs pk;
    for (int l = 0; l < 6; ++l) { pk.W[l] = f->vW[l]; pk.b[l] = f->vb[l]; }
    pk.img = reinterpret_cast<h8_t*>(a.img);
    pk.img_lo = reinterpret_cast<h8_t*>(a.img_lo);
    hipLaunchKernelGGL(k_pack_vel16, dim3((P16_H8 + 255) / 256), dim3(256), 0, st, pk);
    const int64_t tiles = (a.P + TILE - 1) / TILE;
    const unsigned wgs = (unsigned)((tiles + P16_THREADS / 64 - 1) / (P16_THREADS / 64));
    {
        ProfScope ps(PK_PDE_PREFILTER, st);
        if (a.img_lo) hipLaunchKernelGGL(k_rk2_pre16<true>, dim3(wgs), dim3(P16_THREADS), PRE16_IMAGE_BYTES, st, a);
        else hipLaunchKernelGGL(k_rk2_pre16<false>, dim3(wgs), dim3(P16_THREADS), PRE16_IMAGE_BYTES, st, a);
    }
    LAUNCHCK();
    return 0;
}
int launch_pde_band(const nvfi_field_desc* f, int64_t P, const int* perm, const float* sig, const uint8_t* near, float band, uint8_t* flags,
                    int* cnt, hipStream_t st) {
    hipLaunchKernelGGL(k_pde_band, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, *f, P, perm, sig, near, band, flags, cnt);
    LAUNCHCK();
    return 0;
}
// bucket positions -> point indices, in place
__global__ __launch_bounds__(256) void k_pde_band_map(const int* bcount, const int* perm, int* blist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *bcount) blist[i] = perm[blist[i]];
}
int launch_pde_band_map(int64_t P, const int* bcount, const int* perm, int* blist, hipStream_t st) {
    hipLaunchKernelGGL(k_pde_band_map, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, bcount, perm, blist);
    LAUNCHCK();
    return 0;
}
