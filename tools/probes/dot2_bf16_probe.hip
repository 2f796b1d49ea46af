// probe (round 6): does v_dot2c_f32_bf16 form x.lo * 1 + x.hi * 1 + acc exactly in fp32?   hipcc --offload-arch=gfx950 -O3 dot2_bf16_probe.hip -o dot2_probe && ./dot2_probe
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
    return 0;
}
