// probe (round 6): (1) does v_dot2c_f32_bf16 form x.lo * 1 + x.hi * 1 + acc exactly in fp32?  yes (3.6e-7 against float64).
// (2) k_bitcast / k_elems below: the row sum of a b8_t operand written two ways.  hipcc 7.2 (-O3, gfx950) compiles
//     fdot2(bit_cast<bf16x2>(bit_cast<u32x4>(v)[i]), one, acc) to FOUR v_dot2c on the SAME register (it even loads one dword only) - a miscompile;
//     the element-wise form {v[2i], v[2i+1]} uses the four registers.  Check: llvm-objdump -d | grep v_dot2c.   hipcc --offload-arch=gfx950 -O3 dot2_bf16_probe.hip -o dot2_probe && ./dot2_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* x, int n, float* o) {
    float acc = 0.f;
    const bf2 one = {(__bf16)1.0f, (__bf16)1.0f};
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x[i * 64 + threadIdx.x]), one, acc, false);
    o[threadIdx.x] = acc;
}
typedef __bf16 pb8_t __attribute__((ext_vector_type(8)));
typedef unsigned pu4 __attribute__((ext_vector_type(4)));
__global__ void k_bitcast(const pb8_t* x, float* o) {
    const pb8_t v = x[threadIdx.x];
    const pu4 u = __builtin_bit_cast(pu4, v);
    const bf2 one = {(__bf16)1.0f, (__bf16)1.0f};
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, u[i]), one, acc, false);
    o[threadIdx.x] = acc;
}
__global__ void k_elems(const pb8_t* x, float* o) {
    const pb8_t v = x[threadIdx.x];
    const bf2 one = {(__bf16)1.0f, (__bf16)1.0f};
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const bf2 p = {v[2 * i], v[2 * i + 1]}; acc = __builtin_amdgcn_fdot2_f32_bf16(p, one, acc, false); }
    o[threadIdx.x] = acc;
}
int main() {
    const int n = 24;
    unsigned* h = (unsigned*)malloc(n * 64 * 4);
    double ref[64] = {0};
    srand(1);
    for (int i = 0; i < n; ++i)
        for (int l = 0; l < 64; ++l) {
            float a = ((rand() % 2001) - 1000) * 1e-6f * (i % 3 == 0 ? 1.f : (i % 3 == 1 ? 3e-3f : 1e-5f)), b = ((rand() % 2001) - 1000) * 1e-6f;
            unsigned ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
            ua &= 0xffff0000u; ub &= 0xffff0000u;
            float fa, fb; memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
            h[i * 64 + l] = (ub & 0xffff0000u) | (ua >> 16);
            ref[l] += (double)fa + (double)fb;
        }
    unsigned* d; float* o; hipMalloc(&d, n * 64 * 4); hipMalloc(&o, 64 * 4);
    hipMemcpy(d, h, n * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, o);
    float ho[64]; hipMemcpy(ho, o, 64 * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int l = 0; l < 64; ++l) { double e = fabs(ho[l] - ref[l]) / (fabs(ref[l]) + 1e-12); if (e > worst) worst = e; }
    printf("lane 0: got %.9g ref %.9g ; worst relative error over 64 lanes %.3e\n", ho[0], ref[0], worst);
    // (2): 64 lanes x 8 bf16 values = the first 4 dwords per lane of a fresh buffer
    unsigned hb[64 * 4]; double r2[64];
    for (int l = 0; l < 64; ++l) { r2[l] = 0; for (int i = 0; i < 4; ++i) { unsigned w = h[(i * 7 % n) * 64 + l]; hb[l * 4 + i] = w; unsigned lo = w << 16, hi = w & 0xffff0000u; float fl, fh; memcpy(&fl, &lo, 4); memcpy(&fh, &hi, 4); r2[l] += (double)fl + (double)fh; } }
    unsigned* db; hipMalloc(&db, sizeof(hb)); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
        if (which == 0) hipLaunchKernelGGL(k_bitcast, dim3(1), dim3(64), 0, 0, (const pb8_t*)db, o);
        else hipLaunchKernelGGL(k_elems, dim3(1), dim3(64), 0, 0, (const pb8_t*)db, o);
        hipMemcpy(ho, o, 64 * 4, hipMemcpyDeviceToHost);
        double w2 = 0; for (int l = 0; l < 64; ++l) { double e = fabs(ho[l] - r2[l]) / (fabs(r2[l]) + 1e-12); if (e > w2) w2 = e; }
        printf("%s: worst relative error of the 8-element row sum %.3e\n", which == 0 ? "k_bitcast (bit_cast of an extracted dword)" : "k_elems (element-wise pairs)", w2);
    }
    return 0;
}
