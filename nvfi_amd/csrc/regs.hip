// regs.hip - per-iteration plane regularisers: density_L1 + TV_loss_density + TV_loss_app
// in ONE memory-bound pass per plane that also accumulates the weighted gradients.
//
// Reference semantics: TensorVMKeyframeTimeKplane.density_L1 / TV_loss_density / TV_loss_app
// (models/tensorf_keyframe.py:188-231) with utils.tensorf_utils.TVLoss (utils/tensorf_utils.py:139-158; t=True
// multiplies the time-axis term by 3).  Planes are channel-last [H][W][C]: the x neighbour is +-C floats away,
// the y neighbour +-W*C, so every load is coalesced across the channel/x index.
#include "common.h"

struct RegJob {
    const float* p; float* g;
    int H, W, C;
    int l1_mode;        // 0 none, 1 mean|v|, 2 mean|1-v|
    float l1_val;       // 1/n            (loss contribution scale)
    float l1_grad;      // w_l1/n         (gradient scale)
    float h_val, w_val; // TV value scales: 2*1e-2*hmul/count_h , 2*1e-2/count_w
    float tv_w;         // TV weight (gradient scale multiplies the value scales)
    int tv_slot;        // 1: TV density, 2: TV app, 0: no TV
};
struct RegJobs { RegJob j[12]; int n; float* out; const float* wdev; };   // wdev (optional): device float[3] multipliers of the (L1, TV density, TV app) gradients

// One thread handles 4 consecutive channels of a texel (C is a multiple of 4, so they share x and y): 16-byte loads of the
// texel and its four neighbours, 32-bit index arithmetic (the largest plane has 1.9 M elements).
__global__ __launch_bounds__(1024) void k_plane_regs(RegJobs jobs) {
    __shared__ float red[2][16];
    const RegJob& J = jobs.j[blockIdx.y];
    const int total4 = (J.H * J.W * J.C) >> 2;
    const int rowf = J.W * J.C;
    float l1 = 0.f, tv = 0.f;
    const float m_l1 = jobs.wdev ? jobs.wdev[0] : 1.f, m_tv = (jobs.wdev && J.tv_slot) ? jobs.wdev[J.tv_slot] : 1.f;
    for (int i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += gridDim.x * blockDim.x) {
        const int idx = i4 << 2;
        const int y = idx / rowf;
        const int x = (idx - y * rowf) / J.C;
        const float4 v4 = *reinterpret_cast<const float4*>(J.p + idx);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        if (J.l1_mode == 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { l1 += fabsf(v[c]); g[c] += m_l1 * J.l1_grad * (v[c] > 0.f ? 1.f : (v[c] < 0.f ? -1.f : 0.f)); }
        } else if (J.l1_mode == 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float u = 1.f - v[c]; l1 += fabsf(u); g[c] -= m_l1 * J.l1_grad * (u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f)); }
        }
        if (J.tv_slot) {
            float gh[4] = {0.f, 0.f, 0.f, 0.f}, gw[4] = {0.f, 0.f, 0.f, 0.f};
            if (y + 1 < J.H) {
                const float4 n4 = *reinterpret_cast<const float4*>(J.p + idx + rowf);
                const float n[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float d = n[c] - v[c]; tv += J.h_val * d * d; gh[c] -= d; }
            }
            if (y > 0) {
                const float4 n4 = *reinterpret_cast<const float4*>(J.p + idx - rowf);
                const float n[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) gh[c] += v[c] - n[c];
            }
            if (x + 1 < J.W) {
                const float4 n4 = *reinterpret_cast<const float4*>(J.p + idx + J.C);
                const float n[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float d = n[c] - v[c]; tv += J.w_val * d * d; gw[c] -= d; }
            }
            if (x > 0) {
                const float4 n4 = *reinterpret_cast<const float4*>(J.p + idx - J.C);
                const float n[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) gw[c] += v[c] - n[c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) g[c] += m_tv * J.tv_w * 2.f * (J.h_val * gh[c] + J.w_val * gw[c]);
        }
        if (J.g) {
            float4 o = *reinterpret_cast<float4*>(J.g + idx);
            o.x += g[0]; o.y += g[1]; o.z += g[2]; o.w += g[3];
            *reinterpret_cast<float4*>(J.g + idx) = o;
        }
    }
    l1 = wave_sum(l1); tv = wave_sum(tv);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = l1; red[1][threadIdx.x >> 6] = tv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s0 = 0.f, s1 = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { s0 += red[0][w]; s1 += red[1][w]; }
        if (J.l1_mode) atomicAdd(jobs.out + 0, J.l1_val * s0);
        if (J.tv_slot) atomicAdd(jobs.out + J.tv_slot, s1);
    }
}

static int plane_regs(const nvfi_field_desc* f, float w_l1, float w_tv_density, float w_tv_app, const float* wdev, float* out3,
                      const nvfi_grads* grads, void* stream);
extern "C" int nvfi_plane_regs(const nvfi_field_desc* f, float w_l1, float w_tv_density, float w_tv_app, float* out3,
                               const nvfi_grads* grads, void* stream) {
    return plane_regs(f, w_l1, w_tv_density, w_tv_app, nullptr, out3, grads, stream);
}
// weights in DEVICE memory (w3_dev: float[3]): the backward of `w * density_L1()` etc. under autograd hands over its upstream
// gradient as a device scalar; reading it here avoids a host synchronisation
extern "C" int nvfi_plane_regs_dev(const nvfi_field_desc* f, const float* w3_dev, float* out3, const nvfi_grads* grads, void* stream) {
    if (!w3_dev) return nvfi_fail(2, "nvfi_plane_regs_dev: w3_dev is NULL");
    return plane_regs(f, 1.f, 1.f, 1.f, w3_dev, out3, grads, stream);
}
static int plane_regs(const nvfi_field_desc* f, float w_l1, float w_tv_density, float w_tv_app, const float* wdev, float* out3,
                      const nvfi_grads* grads, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    RegJobs jobs; jobs.n = 0; jobs.out = out3; jobs.wdev = wdev;
    if (launch_zero(out3, 3 * sizeof(float), st)) return 1;
    const int A[3] = {0, 0, 1}, Bx[3] = {1, 2, 2}, Cc[3] = {2, 1, 0};
    auto add = [&](const float* p, float* g, int H, int W, int C, int l1_mode, int tv_slot, float hmul, float tvw) {
        RegJob& J = jobs.j[jobs.n++];
        J.p = p; J.g = g; J.H = H; J.W = W; J.C = C; J.l1_mode = l1_mode; J.tv_slot = tv_slot; J.tv_w = tvw;
        const double n = (double)H * W * C;
        J.l1_val = (float)(1.0 / n); J.l1_grad = (float)(w_l1 / n);
        const double ch = (double)C * (H - 1) * W, cw = (double)C * H * (W - 1);
        J.h_val = H > 1 ? (float)(2.0 * 1e-2 * hmul / ch) : 0.f;
        J.w_val = W > 1 ? (float)(2.0 * 1e-2 / cw) : 0.f;
    };
    for (int i = 0; i < 3; ++i) {
        add(f->dps[i], grads ? grads->dps[i] : nullptr, f->G[Bx[i]], f->G[A[i]], f->Cd, 1, 1, 1.f, w_tv_density);
        add(f->dpt[i], grads ? grads->dpt[i] : nullptr, f->K, f->G[Cc[i]], f->Cd, 2, f->K > 1 ? 1 : 0, 3.f, w_tv_density);
        add(f->aps[i], grads ? grads->aps[i] : nullptr, f->G[Bx[i]], f->G[A[i]], f->Ca, 0, 2, 1.f, w_tv_app);
    }
    if ((f->Cd & 3) || (f->Ca & 3)) return nvfi_fail(2, "nvfi_plane_regs: channel counts must be multiples of 4");
    // few, fat workgroups: every workgroup ends with two atomics on the same three floats (~11 ns each when contended)
    hipLaunchKernelGGL(k_plane_regs, dim3(96, jobs.n), dim3(1024), 0, st, jobs);
    LAUNCHCK();
    return 0;
}
