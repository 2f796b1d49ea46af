// optim.hip - one multi-tensor Adam launch for the outer optimiser step (train_nvfi.py:88-96,243: torch.optim.Adam(betas=(0.9,0.99))
// over the per-group learning rates of get_optparam_groups).  Same update rule as torch.optim.Adam without amsgrad / weight decay:
//   m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// HBM-bound: 28 B per parameter (p, g, m, v read; p, m, v written), +4 B when the gradient is zeroed in the same pass.
#include "common.h"
#include <string.h>

#define ADAM_MAX_T 48
struct AdamT { float* p; float* g; float* m; float* v; int64_t n; float step_size; int vec; };
struct AdamArgs { AdamT t[ADAM_MAX_T]; int n; float b1, b2, eps, inv_sqrt_bc2; int zero_grad;
                  const float* hyper; int hyper_base; };   // hyper (device, optional): [0] 1/sqrt(1-b2^t), [1+k] lr_k/(1-b1^t) of tensor k

__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
    const AdamT& T = a.t[blockIdx.y];
    const float b1 = a.b1, b2 = a.b2, eps = a.eps;
    const float isb = a.hyper ? a.hyper[0] : a.inv_sqrt_bc2, ss = a.hyper ? a.hyper[1 + a.hyper_base + blockIdx.y] : T.step_size;
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    if (T.vec) {
        const int64_t n4 = T.n >> 2;
        float4* p4 = reinterpret_cast<float4*>(T.p); float4* g4 = reinterpret_cast<float4*>(T.g);
        float4* m4 = reinterpret_cast<float4*>(T.m); float4* v4 = reinterpret_cast<float4*>(T.v);
        for (int64_t i = tid; i < n4; i += nth) {
            float4 p = p4[i], g = g4[i], m = m4[i], v = v4[i];
            float* pp = &p.x; float* gg = &g.x; float* mm = &m.x; float* vv = &v.x;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mm[c] = b1 * mm[c] + (1.f - b1) * gg[c];
                vv[c] = b2 * vv[c] + (1.f - b2) * gg[c] * gg[c];
                const float denom = sqrtf(vv[c]) * isb + eps;
                pp[c] -= ss * mm[c] / denom;
            }
            p4[i] = p; m4[i] = m; v4[i] = v;
            if (a.zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        for (int64_t i = tid; i < T.n; i += nth) {
            const float g = T.g[i];
            const float m = b1 * T.m[i] + (1.f - b1) * g;
            const float v = b2 * T.v[i] + (1.f - b2) * g * g;
            const float denom = sqrtf(v) * isb + eps;
            T.p[i] -= ss * m / denom;
            T.m[i] = m; T.v[i] = v;
            if (a.zero_grad) T.g[i] = 0.f;
        }
    }
}

static int adam_impl(const nvfi_adam_tensor* t, int n_tensors, float beta1, float beta2, float eps, int64_t step, const float* hyper, int zero_grad, void* stream);

extern "C" int nvfi_adam_step(const nvfi_adam_tensor* t, int n_tensors, float beta1, float beta2, float eps, int64_t step, int zero_grad,
                              void* stream) {
    if (step < 1) return nvfi_fail(2, "nvfi_adam_step: step counts from 1");
    return adam_impl(t, n_tensors, beta1, beta2, eps, step, nullptr, zero_grad, stream);
}

extern "C" int nvfi_adam_step_dev(const nvfi_adam_tensor* t, int n_tensors, float beta1, float beta2, float eps, const float* hyper_dev,
                                  int zero_grad, void* stream) {
    if (!hyper_dev) return nvfi_fail(2, "nvfi_adam_step_dev: hyper_dev is NULL");
    for (int k = 0; k < n_tensors; ++k) if (t[k].n <= 0) return nvfi_fail(2, "nvfi_adam_step_dev: tensor %d is empty (hyper_dev is indexed by table position)", k);
    return adam_impl(t, n_tensors, beta1, beta2, eps, 1, hyper_dev, zero_grad, stream);
}

static int adam_impl(const nvfi_adam_tensor* t, int n_tensors, float beta1, float beta2, float eps, int64_t step, const float* hyper, int zero_grad, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n_tensors <= 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    for (int base = 0; base < n_tensors;) {
        AdamArgs a; memset(&a, 0, sizeof(a));
        a.hyper = hyper; a.hyper_base = base;
        a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2)); a.zero_grad = zero_grad;
        int64_t nmax = 0;
        int k = base;
        for (; k < n_tensors && a.n < ADAM_MAX_T; ++k) {
            if (!t[k].p || !t[k].g || !t[k].m || !t[k].v) return nvfi_fail(2, "nvfi_adam_step: tensor %d has a NULL pointer", k);
            if (t[k].n <= 0) continue;
            AdamT& T = a.t[a.n++];
            T.p = t[k].p; T.g = t[k].g; T.m = t[k].m; T.v = t[k].v; T.n = t[k].n;
            T.step_size = (float)((double)t[k].lr / bc1);
            const uintptr_t al = (uintptr_t)T.p | (uintptr_t)T.g | (uintptr_t)T.m | (uintptr_t)T.v;
            T.vec = ((al & 15) == 0 && (T.n & 3) == 0) ? 1 : 0;
            nmax = T.n > nmax ? T.n : nmax;
        }
        base = k;      // running index: empty tensors are skipped without being counted against the batch
        if (a.n == 0) continue;
        int64_t bx = (nmax / 4 + 255) / 256;
        if (bx < 1) bx = 1;
        if (bx > 2048) bx = 2048;
        hipLaunchKernelGGL(k_adam, dim3((unsigned)bx, a.n), dim3(256), 0, st, a);
        LAUNCHCK();
    }
    return 0;
}


// ---------------------------------------------------------------- photometric loss of the training loop (train_nvfi.py:159,178:
// F.mse_loss(rgb_map, target)): value and d(loss)/d(x) in ONE launch.  torch issues three launches forward (square-difference,
// reduce, divide) and three backward (two fills, one scaled difference) for 6 144 numbers; on a step that is a chain of ~45 dependent
// launches each of them is ~5 us of critical path.  One workgroup: the batch of a training step is a few thousand values.
__global__ __launch_bounds__(1024) void k_mse(const float* __restrict__ x, const float* __restrict__ y, int64_t n, float* loss, float* grad) {
    __shared__ float part[16];
    const float inv = 1.f / (float)n;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = x[i] - y[i];
        s += d * d;
        grad[i] = 2.f * d * inv;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += part[k];
        *loss = t * inv;
    }
}
extern "C" int nvfi_mse(const float* x, const float* target, int64_t n, float* loss, float* grad, void* stream) {
    if (n <= 0) return nvfi_fail(2, "nvfi_mse: n must be positive");
    if (n > (1 << 22)) return nvfi_fail(2, "nvfi_mse: one-workgroup kernel, n <= 4194304 (use torch.nn.functional.mse_loss for images)");
    hipLaunchKernelGGL(k_mse, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, target, n, loss, grad);
    LAUNCHCK();
    return 0;
}
