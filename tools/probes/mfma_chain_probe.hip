// Micro-probe, round 5: issue interval of v_mfma_f32_32x32x16_bf16 when consecutive MFMAs accumulate into the SAME registers (SrcC = the
// previous vDst), into 2, 3, 4 accumulators in rotation, and in the x6 K-step pattern (a0 once, a1 five times).  One wave per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain_probe mfma_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define M(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)
template <int PAT>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (PAT == 1) { M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); M(c0); }
        if (PAT == 2) { M(c0); M(c1); M(c0); M(c1); M(c0); M(c1); M(c0); M(c1); M(c0); M(c1); M(c0); M(c1); }
        if (PAT == 3) { M(c0); M(c1); M(c2); M(c0); M(c1); M(c2); M(c0); M(c1); M(c2); M(c0); M(c1); M(c2); }
        if (PAT == 4) { M(c0); M(c1); M(c2); M(c3); M(c0); M(c1); M(c2); M(c3); M(c0); M(c1); M(c2); M(c3); }
        if (PAT == 5) { M(c0); M(c1); M(c1); M(c1); M(c1); M(c1); M(c0); M(c1); M(c1); M(c1); M(c1); M(c1); }          // x6 K step, two accumulators
        if (PAT == 6) { M(c0); M(c1); M(c2); M(c1); M(c2); M(c2); M(c0); M(c1); M(c2); M(c1); M(c2); M(c2); }          // x6 K step, three accumulators (round-5 order)
        if (PAT == 7) { M(c0); M(c2); M(c1); M(c3); M(c1); M(c3); M(c1); M(c3); M(c1); M(c3); M(c1); M(c3); }          // two row tiles interleaved: (a0,b0),(a1,b1)x5
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int PAT> void run(const char* what, float* out, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-64s %6.1f cycles per MFMA (clock64)\n", what, (double)h / (iters * 12.0));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<4>("four accumulators in rotation", out, cyc);
    run<3>("three in rotation", out, cyc);
    run<2>("two alternating", out, cyc);
    run<1>("one accumulator (every MFMA depends on the previous one)", out, cyc);
    run<6>("x6 K step, three accumulators (a0 a1 a2 a1 a2 a2)", out, cyc);
    run<5>("x6 K step, two accumulators (a0 a1 a1 a1 a1 a1)", out, cyc);
    run<7>("two row tiles interleaved (a0 b0 a1 b1 a1 b1 ...)", out, cyc);
    return 0;
}
