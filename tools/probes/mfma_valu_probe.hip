// Micro-probe: do independent VALU instructions of the SAME wave issue in the shadow of a 64-cycle v_mfma_f32_32x32x2_f32?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_probe mfma_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>   // KIND 0: v_add_f32 (independent regs), 1: v_exp_f32, 2: v_accvgpr_read, 3: dependent v_add chain feeding B
__global__ __launch_bounds__(1024) void k_probe(float* out, long long* clk, int iters) {
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    float a = (threadIdx.x & 255) * 1e-3f, b = blockIdx.x * 1e-3f;
    float t[8];
    for (int k = 0; k < 8; ++k) t[k] = a + k;
    long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const int k = k8 & 3;
            float bb = b;
            if (KIND == 3) {
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[0]) : "v"(a));
                bb = t[0];
            }
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[k], 0, 0, 0);
            if (KIND != 3) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(t[v & 7]) : "v"(a));
                    if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(t[v & 7]));
                    if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(t[v & 7]) : "v"(a));
                }
            }
        }
    }
    long long c1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < 4; ++k) { for (int r = 0; r < 16; ++r) s += acc[k][r]; s += t[k] + t[k + 4]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int NV, int KIND>
static void run(const char* name, int wgs_per_cu = 1) {
    const int iters = 2000, blocks = 256, threads = 256 * wgs_per_cu;   // one workgroup per CU: its waves are co-resident by construction
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * blocks * 1024);
    hipMalloc(&clk, sizeof(long long) * blocks);
    hipLaunchKernelGGL((k_probe<NV, KIND>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    long long* h = (long long*)malloc(sizeof(long long) * blocks);
    hipMemcpy(h, clk, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < blocks; ++b) cyc += h[b];
    cyc /= blocks;
    printf("%-34s NV=%2d  waves/SIMD=%d  ticks per MFMA per wave = %.1f  -> per SIMD %.1f\n", name, NV, wgs_per_cu, cyc / (iters * 8.0), cyc / (iters * 8.0) / wgs_per_cu);
    hipFree(out); hipFree(clk); free(h);
}

int main() {
    run<0, 0>("no VALU");
    run<4, 0>("independent v_add after MFMA");
    run<8, 0>("independent v_add after MFMA");
    run<12, 0>("independent v_add after MFMA");
    run<16, 0>("independent v_add after MFMA");
    run<24, 0>("independent v_add after MFMA");
    run<2, 1>("v_exp after MFMA");
    run<4, 1>("v_exp after MFMA");
    run<8, 1>("v_exp after MFMA");
    run<4, 2>("v_mul+v_add pairs after MFMA");
    run<8, 2>("v_mul+v_add pairs after MFMA");
    run<2, 3>("dependent chain feeding B");
    run<6, 3>("dependent chain feeding B");
    run<12, 3>("dependent chain feeding B");
    printf("-- two and three waves per SIMD --\n");
    run<0, 0>("no VALU", 2);
    run<4, 0>("independent v_add after MFMA", 2);
    run<8, 0>("independent v_add after MFMA", 2);
    run<12, 0>("independent v_add after MFMA", 2);
    run<16, 0>("independent v_add after MFMA", 2);
    run<24, 0>("independent v_add after MFMA", 2);
    run<6, 3>("dependent chain feeding B", 2);
    run<12, 3>("dependent chain feeding B", 2);
    run<8, 0>("independent v_add after MFMA", 3);
    run<16, 0>("independent v_add after MFMA", 3);
    run<24, 0>("independent v_add after MFMA", 3);
    run<12, 3>("dependent chain feeding B", 3);
    run<24, 0>("independent v_add after MFMA", 4);
    return 0;
}
