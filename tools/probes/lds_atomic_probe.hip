// LDS / global float-atomic throughput probe (gfx950).  hipcc --offload-arch=gfx950 -O3 lds_atomic_probe.hip -o lds_atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>   // 0 ds_add_f32 consecutive, 1 ds_add stride 24 (48 lanes active, (ch,dx) layout), 2 same address, 3 plain read-add-write (no atomic), 4 ds_add random texel rows
__global__ __launch_bounds__(256) void k_lds(int iters, float* out, long long* cyc) {
    extern __shared__ float lds[];
    const int N = 8192;
    for (int k = threadIdx.x; k < N; k += blockDim.x) lds[k] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long t0 = __builtin_readcyclecounter();
    unsigned s = 12345u + wv * 977u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        int base = (s >> 8) % (N - 256);
        int idx;
        if (MODE == 0) idx = base + lane;
        else if (MODE == 1) idx = base + ((lane & 1) * 24 + (lane >> 1));
        else if (MODE == 2) idx = base;
        else if (MODE == 3) idx = wv * 64 + lane;
        else idx = (base + (lane >> 3) * 997) % (N - 64) + (lane & 7) * 4;
        if (MODE == 3) { lds[idx] = lds[idx] + 1.f; }
        else if (MODE == 1) { if (lane < 48) atomicAdd(&lds[idx], 1.f); }
        else atomicAdd(&lds[idx], 1.f);
    }
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float sum = 0.f;
    for (int k = threadIdx.x; k < N; k += blockDim.x) sum += lds[k];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int MODE>   // global float atomics: 0 = 48 lanes contiguous 192 B at a random texel, 1 = 64 lanes contiguous, 2 = 4 lanes/clk style: random per lane
__global__ __launch_bounds__(256) void k_glb(int iters, float* buf, size_t n) {
    const int lane = threadIdx.x & 63;
    unsigned s = 12345u + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 977u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        size_t base = ((size_t)(s >> 4) * 24) % (n - 4096);
        if (MODE == 0) { if (lane < 48) atomicAdd(buf + base + lane, 1.f); }
        else if (MODE == 1) atomicAdd(buf + base + lane, 1.f);
        else { unsigned r = s ^ (lane * 2654435761u); atomicAdd(buf + (size_t)(r >> 2) % n, 1.f); }
    }
}

template <typename F> float timeit(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms;
}

int main() {
    float* out; long long* cyc; CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 1024 * 8));
    const int iters = 4000;
    const char* names[5] = {"ds_add consecutive 64", "ds_add (ch,dx) 48 lanes", "ds_add same address", "plain rmw (no atomic)", "ds_add 8 texels x 8 lanes stride 4"};
    for (int m = 0; m < 5; ++m) {
        for (int wg = 1; wg <= 4; wg *= 4) {
            auto f = [&]() {
                dim3 g(256 * wg), b(256);
                size_t sh = 8192 * 4;
                if (m == 0) hipLaunchKernelGGL(k_lds<0>, g, b, sh, 0, iters, out, cyc);
                if (m == 1) hipLaunchKernelGGL(k_lds<1>, g, b, sh, 0, iters, out, cyc);
                if (m == 2) hipLaunchKernelGGL(k_lds<2>, g, b, sh, 0, iters, out, cyc);
                if (m == 3) hipLaunchKernelGGL(k_lds<3>, g, b, sh, 0, iters, out, cyc);
                if (m == 4) hipLaunchKernelGGL(k_lds<4>, g, b, sh, 0, iters, out, cyc);
            };
            float ms = timeit(f);
            long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            // per CU: wg workgroups x 4 waves x iters instructions
            printf("%-36s wg/CU=%d: %.3f ms, %.1f cycles (wave clock) per wave-instr per CU-slot, %.2f ns per instr per CU\n", names[m], wg, ms,
                   (double)c / iters, ms * 1e6 / ((double)iters * 4 * wg));
        }
    }
    const size_t n = (size_t)199 * 199 * 24 * 3;
    float* buf; CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
    const char* gn[3] = {"global atomic 48 lanes contiguous", "global atomic 64 lanes contiguous", "global atomic random per lane"};
    for (int m = 0; m < 3; ++m) {
        for (int wg = 1; wg <= 8; wg *= 2) {
            const int it2 = 500;
            auto f = [&]() {
                dim3 g(256 * wg), b(256);
                if (m == 0) hipLaunchKernelGGL(k_glb<0>, g, b, 0, 0, it2, buf, n);
                if (m == 1) hipLaunchKernelGGL(k_glb<1>, g, b, 0, 0, it2, buf, n);
                if (m == 2) hipLaunchKernelGGL(k_glb<2>, g, b, 0, 0, it2, buf, n);
            };
            float ms = timeit(f);
            double lanes = (double)256 * wg * 4 * it2 * (m == 0 ? 48 : 64);
            printf("%-36s wg/CU=%d: %.3f ms, %.3g lane-atomics/s\n", gn[m], wg, ms, lanes / (ms * 1e-3));
        }
    }
    return 0;
}
