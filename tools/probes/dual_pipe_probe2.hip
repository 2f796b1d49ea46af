// Micro-probe, round 4: WHICH vector instructions run beside fp32 MFMAs of another wave on the same SIMD?  dual_pipe_probe.hip (round 3)
// only tried v_pk_fma_f32 - the packed-fp32 FMA that the data sheet prices like the fp32 matrix pipe - and found the sum of the two times.
// Here the vector role issues one of: v_pk_fma_f32, plain v_fma_f32, v_exp_f32 (quarter rate), v_add_u32 (integer), ds_write_b128 +
// ds_read_b128 (LDS), and the matrix role one of: fp32 32x32x2 with four independent accumulators, the same with ONE accumulator (a
// dependent chain, as a dgrad with one output tile issues it), fp16 32x32x8.  Every role gets a fixed amount of work; HIP-event time of
// each role ALONE and of both TOGETHER: perfect overlap gives max(alone), none gives their sum.
// build: hipcc --offload-arch=gfx950 -O3 -o dual_pipe_probe2 dual_pipe_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// MOP: 0 fp32 MFMA x 4 accumulators, 1 fp32 MFMA x 1 accumulator, 2 fp16 MFMA 32x32x8 x 4 accumulators
// VOP: 0 v_pk_fma_f32, 1 v_fma_f32, 2 v_exp_f32, 3 v_add_u32, 4 LDS write + read (b128)
template <int MOP, int VOP>
__global__ __launch_bounds__(1024) void k_dual(float* out, int nm_waves, int it_m, int it_v, float w0, float w1) {
    __shared__ f32x4 lds[1024];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool matrix = wave < nm_waves;
    float a = (threadIdx.x & 63) * 1e-3f, b = blockIdx.x * 1e-3f;
    float s = 0.f;
    if (matrix) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        f16x4 ha = {(_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b};
        for (int it = 0; it < it_m; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (MOP == 0) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 3], 0, 0, 0);
                else if (MOP == 1) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                else acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x8f16(ha, ha, acc[k & 3], 0, 0, 0);
            }
        }
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    } else {
        f32x2 acc[16];
        for (int k = 0; k < 16; ++k) acc[k] = f32x2{a + k, b + k};
        f32x2 x = {a, b};
        f32x2 wv = {w0 + a, w1 + a};
        unsigned ia[16]; float fa[16];
        for (int k = 0; k < 16; ++k) { ia[k] = threadIdx.x + k; fa[k] = a + 0.5f * k; }
        f32x4 q = {a, b, a, b};
        volatile f32x4* lpv = lds + threadIdx.x;
        for (int it = 0; it < it_v; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                if (VOP == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k & 15]) : "v"(wv), "v"(x));
                else if (VOP == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fa[k & 15]) : "v"(a), "v"(b));
                else if (VOP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(fa[k & 15]));
                else if (VOP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[k & 15]) : "v"(ia[(k + 1) & 15]));
                else if ((k & 7) == 0) { *lpv = q; q = *lpv; }
            }
        }
        for (int k = 0; k < 16; ++k) s += acc[k][0] + acc[k][1] + ia[k] + fa[k];
        s += q.x;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MOP, int VOP>
static float launch(float* out, int nm, int nv, int it_m, int it_v) {    // nm matrix waves and nv vector waves PER SIMD; ms of the launch
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_dual<MOP, VOP>), dim3(256), dim3(64 * 4 * (nm + nv)), 0, 0, out, 4 * nm, it_m, it_v, 1.0001f, 0.9999f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

template <int MOP, int VOP>
static void run(float* out, int nm, int nv) {
    static const char* mn[] = {"fp32 mfma x4 acc", "fp32 mfma x1 acc", "fp16 mfma x4 acc"};
    static const char* vn[] = {"v_pk_fma_f32", "v_fma_f32", "v_exp_f32", "v_add_u32", "ds_write+read b128"};
    const int it_m = (MOP == 2 ? 20000 : 10000) / nm, it_v = (VOP == 2 ? 10000 : (VOP == 4 ? 20000 : 40000)) / nv;
    const float tm = launch<MOP, VOP>(out, nm, 0, it_m, it_v), tv = launch<MOP, VOP>(out, 0, nv, it_m, it_v), tb = launch<MOP, VOP>(out, nm, nv, it_m, it_v);
    const float lo = tm > tv ? tm : tv, hi = tm + tv;
    printf("%-17s + %-19s %d + %d waves/SIMD: matrix alone %.2f ms  vector alone %.2f ms  together %.2f ms  overlap %.2f\n",
           mn[MOP], vn[VOP], nm, nv, tm, tv, tb, (hi - tb) / (hi - lo));
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    run<0, 0>(out, 1, 1); run<0, 1>(out, 1, 1); run<0, 2>(out, 1, 1); run<0, 3>(out, 1, 1); run<0, 4>(out, 1, 1);
    run<1, 1>(out, 1, 1); run<1, 2>(out, 1, 1); run<1, 4>(out, 1, 1);
    run<0, 1>(out, 1, 2); run<0, 2>(out, 2, 1);
    run<2, 0>(out, 1, 1); run<2, 1>(out, 1, 1); run<2, 2>(out, 1, 1); run<2, 4>(out, 1, 1);
    // matrix role alone, one accumulator vs four, beside ANOTHER matrix wave: who gets the pipe?
    hipFree(out);
    return 0;
}
