// Micro-probe, round 6: the ISA window of the round-5 x6 glitch, lifted out of k_rk2_x6<1>.
//
// In the build WITH packed-fp32 code one encoder slot of a step's second evaluation came out wrong in lanes 48..63, rarely and only at two
// workgroups per CU.  That slot is the first argument reduction of p_x = x - hdt * v1x, and p_x is formed by the only packed arithmetic of the RK2
// glue (disassembly of k_rk2_x6<1>):
//     v_cndmask_b32     v3, v1, 0, s[8:9]                                           (the velocity gate)
//     v_cndmask_b32     v2, v6, 0, s[8:9]
//     v_pk_mul_f32      v[2:3], v[0:1], v[2:3] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]     (hdt broadcast from the LOW half of a pair with a stale high half)
//     v_pk_add_f32      v[94:95], v[32:33], v[2:3] op_sel:[1,0] op_sel_hi:[0,1]              (cross-wise add: p_x lands in the HIGH half)
//     v_mul_f32         v1, 0x3f22f983, v95                                         (2/pi * p_x: the encoder's range reduction)
// This probe runs exactly that sequence (one inline-asm block, fixed registers) ITER times per wave on lane-dependent data, between dependent chains
// of v_mfma_f32_32x32x16_bf16 (the x6 accumulate pattern: 6 per K step), and compares every result bit for bit with the same arithmetic done by plain
// v_mul_f32 / v_add_f32.  Variants: MF = MFMAs in front of the window (0 / 1 / 6), a lane-dependent number of wait states between them and the window
// (s_nop 0..7 by wave, so that the two waves of a SIMD drift against each other), 1 / 2 / 4 waves per SIMD (dynamic LDS sets the occupancy).
// Output per variant: windows executed, mismatches, and the histogram of mismatching lanes in quarters (0-15 | 16-31 | 32-47 | 48-63).
// build: hipcc --offload-arch=gfx950 -O3 -o pk_opsel_probe pk_opsel_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define MFMA "v_mfma_f32_32x32x16_bf16 v[108:123], v[104:107], v[100:103], v[108:123]\n"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
             "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "vcc", "memory"
// window: inputs %2 = hdt, %3 = stale, %4 = g_lo (v6 role), %5 = g_hi (v1 role), %6 = x_lo (v32), %7 = x_hi (v33), mask in vcc; outputs %0 = v94 (lo), %1 = 2/pi * v95
#define WINDOW \
    "v_mov_b32 v90, %2\n v_mov_b32 v91, %3\n v_mov_b32 v96, %6\n v_mov_b32 v97, %7\n" \
    "v_cndmask_b32 v93, 0, %5, vcc\n" \
    "v_cndmask_b32 v92, 0, %4, vcc\n" \
    "v_pk_mul_f32 v[92:93], v[90:91], v[92:93] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n" \
    "v_pk_add_f32 v[94:95], v[96:97], v[92:93] op_sel:[1,0] op_sel_hi:[0,1]\n" \
    "v_mul_f32 %1, 0x3f22f983, v95\n" \
    "v_mov_b32 %0, v94\n"

template <int MF>
__global__ __launch_bounds__(256) void k_probe(unsigned long long* stats, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    (void)lds;
    unsigned long long bad = 0, done = 0;
    const unsigned bb = 0x3f803f80u, aa = 0x3f803f80u;
    asm volatile("v_mov_b32 v100, %0\n v_mov_b32 v101, %0\n v_mov_b32 v102, %0\n v_mov_b32 v103, %0\n v_mov_b32 v104, %1\n v_mov_b32 v105, %1\n v_mov_b32 v106, %1\n v_mov_b32 v107, %1\n"
                 "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n"
                 "v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n"
                 :: "v"(bb), "v"(aa) : CLOB);
    for (int it = 0; it < iters; ++it) {
        // lane- and iteration-dependent data (no two lanes alike, so a value from another lane or another pass is a mismatch)
        const float hdt = 0.0125f + 1e-4f * (float)((it * 7 + wave) & 31);
        const float stale = 3.0f + (float)lane;
        const float glo = 0.37f + 0.011f * (float)lane + 0.001f * (float)(it & 15), ghi = -0.81f + 0.013f * (float)lane;
        const float xlo = -0.6f + 0.017f * (float)lane, xhi = 0.45f - 0.009f * (float)lane + 0.002f * (float)(it & 7);
        const bool gate_open = ((lane * 5 + it) % 11) != 0;
        float r94, r95;
        const unsigned long long m = __ballot(gate_open);
        // expected: plain instructions, same roundings (separate multiply and add, no fma)
        const float g0 = gate_open ? glo : 0.f, g1 = gate_open ? ghi : 0.f;
        float t0, t1, e94, e95m;
        asm volatile("v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %6\n v_sub_f32 %2, %7, %0\n v_sub_f32 %3, %8, %1\n v_mul_f32 %3, 0x3f22f983, %3\n"
                     : "=&v"(t0), "=&v"(t1), "=&v"(e94), "=&v"(e95m) : "v"(hdt), "v"(g0), "v"(g1), "v"(xhi), "v"(xlo));
        // (pk_mul with neg on src1: -(g) * hdt = -(hdt * g) exactly; pk_add: x + (-(hdt g)) = x - hdt g exactly: the same fp32 values)
        const int nops = wave & 7;
        if (MF == 0) {
            asm volatile("s_mov_b64 vcc, %8\n" WINDOW : "=&v"(r94), "=&v"(r95) : "v"(hdt), "v"(stale), "v"(glo), "v"(ghi), "v"(xlo), "v"(xhi), "s"(m) : CLOB);
        } else if (MF == 1) {
            if (nops & 1) asm volatile("s_nop 3" ::: "memory");
            asm volatile("s_mov_b64 vcc, %8\n" MFMA WINDOW MFMA : "=&v"(r94), "=&v"(r95) : "v"(hdt), "v"(stale), "v"(glo), "v"(ghi), "v"(xlo), "v"(xhi), "s"(m) : CLOB);
        } else {
            if (nops & 1) asm volatile("s_nop 1" ::: "memory");
            if (nops & 2) asm volatile("s_nop 3" ::: "memory");
            if (nops & 4) asm volatile("s_nop 7" ::: "memory");
            asm volatile("s_mov_b64 vcc, %8\n" MFMA MFMA MFMA WINDOW MFMA MFMA MFMA : "=&v"(r94), "=&v"(r95) : "v"(hdt), "v"(stale), "v"(glo), "v"(ghi), "v"(xlo), "v"(xhi), "s"(m) : CLOB);
        }
        ++done;
        if (__float_as_uint(r94) != __float_as_uint(e94) || __float_as_uint(r95) != __float_as_uint(e95m)) ++bad;
    }
    atomicAdd(&stats[0], done);
    if (bad) { atomicAdd(&stats[1], bad); atomicAdd(&stats[2 + (lane >> 4)], bad); }
}

template <int MF>
static void run(const char* name, unsigned long long* d, int waves_per_simd, int iters) {
    // occupancy through dynamic LDS: 160 KB per CU; 256-thread workgroups = one wave per SIMD each
    const int lds = waves_per_simd == 1 ? 100 * 1024 : (waves_per_simd == 2 ? 60 * 1024 : 30 * 1024);
    hipFuncSetAttribute((const void*)k_probe<MF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipMemset(d, 0, 8 * sizeof(unsigned long long));
    hipLaunchKernelGGL(k_probe<MF>, dim3(256 * waves_per_simd * 4), dim3(256), lds, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s %d wave(s)/SIMD: %12llu windows, %llu mismatches (lanes 0-15 %llu | 16-31 %llu | 32-47 %llu | 48-63 %llu)\n", name, waves_per_simd, h[0], h[1], h[2], h[3], h[4], h[5]);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 8 * sizeof(unsigned long long));
    const int iters = 20000;
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("window alone", d, w, iters);
        run<1>("MFMA | window | MFMA", d, w, iters);
        run<6>("3 MFMA | window | 3 MFMA (x6 step)", d, w, iters);
    }
    return 0;
}
