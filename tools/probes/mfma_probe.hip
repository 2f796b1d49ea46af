// Micro-probe: issue rate of v_mfma_f32_32x32x2_f32 and the shader clock under full-chip MFMA load.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k_probe(float* out, long long* clk, int iters) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    long long c0 = __builtin_readcyclecounter();          // s_memtime: shader clock
    long long r0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    long long c1 = __builtin_readcyclecounter();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
static void run(const char* name, int blocks, int iters) {
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&clk, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<NACC>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long* h = (long long*)malloc(sizeof(long long) * 2 * blocks);
    hipMemcpy(h, clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; rt += h[2 * b + 1]; }
    cyc /= blocks; rt /= blocks;
    const double n_mfma = (double)iters * 4 * NACC;
    const double flops = n_mfma * 4096.0 * blocks * 4;
    printf("%-28s blocks=%d  memtime ticks/mfma=%.2f  realtime ns/mfma=%.2f  memtime:realtime=%.3f  kernel %.3f ms  %.1f TFLOP/s\n",
           name, blocks, cyc / n_mfma, rt * 10.0 / n_mfma, cyc / (rt > 0 ? rt : 1), ms, flops / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(clk); free(h);
}

int main() {
    run<8>("8 acc, 1 WG/CU", 256, 4000);
    run<8>("8 acc, 2 WG/CU", 512, 4000);
    run<2>("2 acc, 1 WG/CU", 256, 16000);
    run<4>("4 acc, 1 WG/CU", 256, 8000);
    run<8>("8 acc, 1 WG on 1 CU", 1, 4000);
    run<8>("8 acc, 4 WG/CU", 1024, 4000);
    return 0;
}
