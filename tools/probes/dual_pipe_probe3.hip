// Micro-probe, round 5 (verdict item 7): does anything issue beside a MATRIX instruction of another wave on the same SIMD - for the fp32
// MFMA (32x32x2) and for the 16-bit MFMAs the kernels actually use (v_mfma_f32_32x32x16_f16 / _bf16)?
// dual_pipe_probe2 (round 4) answered "no, times add" for every pairing; the reviewer's objections, taken one by one here:
//   (i)   it used the 32x32x8 f16 instruction: here 32x32x16 f16 and bf16;
//   (ii)  both roles had about equal length, so a placement artefact (two matrix waves on one SIMD, two vector waves on another) would
//         look like "no overlap": here the vector role runs at 0.5x, 1x and 2x the matrix role's time, and
//   (iii) every wave reports the SIMD it ran on (s_getreg_b32 HW_REG_HW_ID, SIMD_ID = bits 5:4): the table prints, per pairing, how
//         many SIMDs of workgroup 0 held exactly one wave of each role.
// A second experiment measures the same thing INSIDE one wave (one wave per SIMD): a stream of MFMAs with F independent VALU / LDS
// instructions between consecutive MFMAs - the cost per filler is what an epilogue interleaved by hand would pay.
// build: hipcc --offload-arch=gfx950 -O3 -o dual_pipe_probe3 dual_pipe_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MOP: 0 fp32 32x32x2 | 1 f16 32x32x16 | 2 bf16 32x32x16       (four independent accumulators)
template <int MOP>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[4], int k, float a, float b, const f16x8& ha, const bf16x8& ba) {
    if (MOP == 0) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 3], 0, 0, 0);
    else if (MOP == 1) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, ha, acc[k & 3], 0, 0, 0);
    else acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, ba, acc[k & 3], 0, 0, 0);
}
// VOP: 0 v_fma_f32 | 1 v_exp_f32 | 2 v_and_b32 + v_sub_f32 (the operand split of a bf16x3 decomposition) | 3 v_perm_b32 | 4 ds_read_b128
template <int VOP>
__device__ __forceinline__ void valu_step(float (&fa)[16], unsigned (&ia)[16], int k, float a, float b, volatile f32x4* lp, f32x4& q) {
    if (VOP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fa[k & 15]) : "v"(a), "v"(b));
    else if (VOP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(fa[k & 15]));
    else if (VOP == 2) { asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(ia[k & 15]) : "v"(fa[k & 15])); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(fa[k & 15]) : "v"(ia[k & 15])); }
    else if (VOP == 3) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(ia[k & 15]) : "v"(ia[(k + 1) & 15]), "v"(ia[(k + 2) & 15]), "v"(0x07060302u));
    else { q = *lp; }
}

__device__ __forceinline__ int simd_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return (v >> 4) & 3; }

template <int MOP, int VOP>
__global__ __launch_bounds__(1024) void k_dual(float* out, int* where, int nm_waves, int it_m, int it_v) {
    __shared__ f32x4 lds[1024];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool matrix = wave < nm_waves;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) where[wave] = simd_id() | (matrix ? 16 : 0);
    float a = (threadIdx.x & 63) * 1e-3f, b = blockIdx.x * 1e-3f, s = 0.f;
    lds[threadIdx.x] = f32x4{a, b, a, b};
    __syncthreads();
    if (matrix) {
        f32x16 acc[4];
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        f16x8 ha; bf16x8 ba;
        for (int k = 0; k < 8; ++k) { ha[k] = (_Float16)(a + k); ba[k] = (__bf16)(b + k); }
        for (int it = 0; it < it_m; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) mfma_step<MOP>(acc, k, a, b, ha, ba);
        }
        for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    } else {
        float fa[16]; unsigned ia[16];
        for (int k = 0; k < 16; ++k) { ia[k] = threadIdx.x + k; fa[k] = a + 0.5f * k; }
        f32x4 q = {a, b, a, b};
        volatile f32x4* lp = lds + threadIdx.x;
        for (int it = 0; it < it_v; ++it) {
#pragma unroll
            for (int k = 0; k < 32; ++k) valu_step<VOP>(fa, ia, k, a, b, lp, q);
        }
        for (int k = 0; k < 16; ++k) s += ia[k] + fa[k];
        s += q.x;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave per SIMD: MFMA stream with F fillers between consecutive MFMAs
template <int MOP, int VOP, int F>
__global__ __launch_bounds__(256) void k_inter(float* out, int it) {
    __shared__ f32x4 lds[256];
    float a = (threadIdx.x & 63) * 1e-3f, b = blockIdx.x * 1e-3f, s = 0.f;
    lds[threadIdx.x] = f32x4{a, b, a, b};
    __syncthreads();
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    f16x8 ha; bf16x8 ba;
    for (int k = 0; k < 8; ++k) { ha[k] = (_Float16)(a + k); ba[k] = (__bf16)(b + k); }
    float fa[16]; unsigned ia[16];
    for (int k = 0; k < 16; ++k) { ia[k] = threadIdx.x + k; fa[k] = a + 0.5f * k; }
    f32x4 q = {a, b, a, b};
    volatile f32x4* lp = lds + threadIdx.x;
    for (int i = 0; i < it; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mfma_step<MOP>(acc, k, a, b, ha, ba);
#pragma unroll
            for (int f = 0; f < F; ++f) valu_step<VOP>(fa, ia, k * F + f, a, b, lp, q);
        }
    }
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    for (int k = 0; k < 16; ++k) s += ia[k] + fa[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + q.x;
}

static float timed(void (*launch)(void*), void* ctx) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0); launch(ctx); hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

struct Ctx { float* out; int* where; int nm, nv, it_m, it_v; };
template <int MOP, int VOP> static void go(void* p) {
    Ctx* c = (Ctx*)p;
    hipLaunchKernelGGL((k_dual<MOP, VOP>), dim3(256), dim3(64 * 4 * (c->nm + c->nv)), 0, 0, c->out, c->where, 4 * c->nm, c->it_m, c->it_v);
}
static const char* MN[] = {"fp32 32x32x2", "f16 32x32x16", "bf16 32x32x16"};
static const char* VN[] = {"v_fma_f32", "v_exp_f32", "v_and+v_sub (split)", "v_perm_b32", "ds_read_b128"};

template <int MOP, int VOP>
static void run(float* out, int* where, float vscale) {
    // iteration counts that give each role ~1 ms alone (8 MFMAs / 32 vector instructions per iteration)
    const int it_m = MOP == 0 ? 4000 : 8000;
    const int base_v = VOP == 1 ? 8000 : (VOP == 4 ? 4000 : 16000);
    Ctx c{out, where, 1, 0, it_m, (int)(base_v * vscale)};
    const float tm = timed(go<MOP, VOP>, &c);
    c.nm = 0; c.nv = 1;
    const float tv = timed(go<MOP, VOP>, &c);
    c.nm = 1; c.nv = 1;
    const float tb = timed(go<MOP, VOP>, &c);
    int h[8];
    hipMemcpy(h, where, sizeof(h), hipMemcpyDeviceToHost);
    int mixed = 0;
    for (int sd = 0; sd < 4; ++sd) {
        int m = 0, v = 0;
        for (int w = 0; w < 8; ++w) if ((h[w] & 3) == sd) { if (h[w] & 16) ++m; else ++v; }
        if (m == 1 && v == 1) ++mixed;
    }
    const float lo = tm > tv ? tm : tv, hi = tm + tv;
    printf("%-14s + %-20s matrix alone %.3f ms  vector alone %.3f ms  together %.3f ms  overlap %.2f   SIMDs of workgroup 0 with one wave of each role: %d/4\n",
           MN[MOP], VN[VOP], tm, tv, tb, (hi - tb) / (hi - lo), mixed);
}

template <int MOP, int VOP, int F> static void goi(void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL((k_inter<MOP, VOP, F>), dim3(256), dim3(256), 0, 0, c->out, c->it_m); }
template <int MOP, int VOP>
static void run_inter(float* out) {
    Ctx c{out, nullptr, 0, 0, MOP == 0 ? 4000 : 8000, 0};
    const float t0 = timed(goi<MOP, VOP, 0>, &c), t1 = timed(goi<MOP, VOP, 1>, &c), t2 = timed(goi<MOP, VOP, 2>, &c), t4 = timed(goi<MOP, VOP, 4>, &c), t8 = timed(goi<MOP, VOP, 8>, &c);
    const double per = t0 * 1e-3 * 2.4e9 / (c.it_m * 8.0);
    printf("%-14s + F x %-20s one wave/SIMD: %.1f cycles per MFMA alone (at 2.4 GHz); with F = 1 / 2 / 4 / 8 fillers per MFMA: x%.2f  x%.2f  x%.2f  x%.2f\n",
           MN[MOP], VN[VOP], per, t1 / t0, t2 / t0, t4 / t0, t8 / t0);
}

int main() {
    float* out; int* where;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipMalloc(&where, sizeof(int) * 64);
    printf("== two roles on one SIMD (one matrix wave + one vector wave per SIMD, 256 workgroups of 8 waves); overlap 1 = max(alone), 0 = sum\n");
    for (float vs : {0.5f, 1.f, 2.f}) {
        printf("-- vector role at %.1fx its nominal length\n", vs);
        run<0, 0>(out, where, vs); run<0, 1>(out, where, vs); run<0, 4>(out, where, vs);
        run<1, 0>(out, where, vs); run<1, 1>(out, where, vs); run<1, 2>(out, where, vs); run<1, 3>(out, where, vs); run<1, 4>(out, where, vs);
        run<2, 0>(out, where, vs); run<2, 1>(out, where, vs); run<2, 2>(out, where, vs); run<2, 3>(out, where, vs); run<2, 4>(out, where, vs);
    }
    printf("== fillers inside one wave's MFMA stream (one wave per SIMD)\n");
    run_inter<0, 0>(out); run_inter<0, 1>(out); run_inter<0, 4>(out);
    run_inter<1, 0>(out); run_inter<1, 1>(out); run_inter<1, 2>(out); run_inter<1, 3>(out); run_inter<1, 4>(out);
    run_inter<2, 0>(out); run_inter<2, 2>(out); run_inter<2, 3>(out); run_inter<2, 4>(out);
    hipFree(out); hipFree(where);
    return 0;
}
