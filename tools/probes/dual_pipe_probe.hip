// Micro-probe: do the two fp32 pipes of a CDNA4 SIMD run side by side?  Waves that only issue v_mfma_f32_32x32x2_f32 (matrix pipe,
// 64 cycles each) share a SIMD with waves that only issue v_pk_fma_f32 (vector pipe, 4 cycles per wave-instruction, the weight pair in
// SGPRs as a lane-is-a-sample GEMM would hold it).  Every role gets a fixed amount of work; the kernel's HIP-event time is measured
// for each role ALONE and for both TOGETHER: perfect overlap would give max(alone), none gives their sum.
// (Per-wave timestamps are NOT a usable measure here: the waves of a workgroup do not start and end together, and summing
// flops / own-duration over waves that share a SIMD double-counts - it reported "280 TFLOP/s" for four matrix waves per SIMD.)
// build: hipcc --offload-arch=gfx950 -O3 -o dual_pipe_probe dual_pipe_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// waves [0, nm_waves) of a workgroup are matrix waves, the rest vector waves (consecutive waves land on consecutive SIMDs)
// SRC: 0 = weight pair in SGPRs, 1 = all operands in VGPRs
template <int SRC>
__global__ __launch_bounds__(1024) void k_dual(float* out, int nm_waves, int it_m, int it_v, float w0, float w1) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool matrix = wave < nm_waves;
    float a = (threadIdx.x & 63) * 1e-3f, b = blockIdx.x * 1e-3f;
    float s = 0.f;
    if (matrix) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        for (int it = 0; it < it_m; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 3], 0, 0, 0);
        }
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    } else {
        f32x2 acc[16];
        for (int k = 0; k < 16; ++k) acc[k] = f32x2{a + k, b + k};
        f32x2 x = {a, b};
        f32x2 w = {w0, w1};            // kernel arguments: uniform, live in SGPRs
        f32x2 wv = {w0 + a, w1 + a};
        for (int it = 0; it < it_v; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                if (SRC == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k & 15]) : "s"(w), "v"(x));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k & 15]) : "v"(wv), "v"(x));
            }
        }
        for (int k = 0; k < 16; ++k) s += acc[k][0] + acc[k][1];
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SRC>
static float launch(float* out, int nm, int nv, int it_m, int it_v) {    // nm matrix waves and nv vector waves PER SIMD; ms of the launch
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_dual<SRC>, dim3(256), dim3(64 * 4 * (nm + nv)), 0, 0, out, 4 * nm, it_m, it_v, 1.0001f, 0.9999f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

template <int SRC>
static void run(float* out, int nm, int nv) {
    // ~2 ms per role at the nominal rates: a matrix iteration is 8 x 64 cycles, a vector iteration 32 x 4 cycles
    const int it_m = 10000 / nm, it_v = 40000 / nv;
    const double fl_m = 256.0 * 4 * nm * it_m * 8.0 * 4096.0, fl_v = 256.0 * 4 * nv * it_v * 32.0 * 256.0;
    const float tm = launch<SRC>(out, nm, 0, it_m, it_v), tv = launch<SRC>(out, 0, nv, it_m, it_v), tb = launch<SRC>(out, nm, nv, it_m, it_v);
    const float lo = tm > tv ? tm : tv, hi = tm + tv;
    printf("%s weights, %d matrix + %d vector waves per SIMD:  matrix alone %.2f ms (%5.1f TFLOP/s)   vector alone %.2f ms (%5.1f TFLOP/s)   "
           "together %.2f ms (%5.1f TFLOP/s)   overlap %.2f  (1 = max of the two, 0 = their sum)\n",
           SRC == 0 ? "SGPR" : "VGPR", nm, nv, tm, fl_m / tm * 1e-9, tv, fl_v / tv * 1e-9, tb, (fl_m + fl_v) / tb * 1e-9, (hi - tb) / (hi - lo));
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    run<0>(out, 1, 1); run<0>(out, 1, 2); run<0>(out, 2, 2); run<0>(out, 1, 3); run<0>(out, 3, 1);
    run<1>(out, 1, 1); run<1>(out, 1, 2);
    hipFree(out);
    return 0;
}
