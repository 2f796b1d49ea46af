// Micro-probe, round 5: which register hazards around v_mfma_f32_32x32x16_bf16 does gfx950 NOT interlock?
// Background: the x6 kernels (vel_x6.hip) produced rare wrong columns (points 16..31 of a 32-point tile, 0.1-1 % of the tiles, more with
// two waves per SIMD) that moved with the register allocation.  Every sequence below is written in ONE inline-asm block on fixed
// registers, so the compiler neither reorders nor pads it:
//     B = v[100:103], A = v[104:107], accumulator = v[108:123]
//     init ; CH x v_mfma (a dependent chain on the one accumulator) ; <action> ; <wait> ; read four accumulator registers
// A = 1.0 everywhere, B = b(n) = 1 + (n & 7) for column n = lane & 31: every MFMA adds 16 b(n) to every row, so the expected value is
// CH * 16 * b(n) and anything else is a hazard the hardware let through.
//   action 0: none                                 (is the accumulator read behind <wait> stale?)
//   action 1: VALU overwrites B straight behind the last MFMA's issue
//   action 2: VALU overwrites A
//   action 3: ds_read_b128 into B                  (LDS holds garbage = 64.0)
//   action 4: ds_read_b128 into A
//   action 5: global_load_dwordx4 into B
//   wait   0: s_nop 11   (what the compiler puts between the last MFMA and a VALU read of its result)
//   wait   1: 8 x s_nop 15
//   wait   2: none (the hardware's own interlock, if there is one)
// Each is run with 1, 2 and 4 waves per SIMD (dynamic LDS sets the occupancy), every wave doing the same thing.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_hazard_probe mfma_hazard_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define MFMA "v_mfma_f32_32x32x16_bf16 v[108:123], v[104:107], v[100:103], v[108:123]\n"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
             "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "memory"

#define INIT \
    "v_mov_b32 v100, %4\n v_mov_b32 v101, %4\n v_mov_b32 v102, %4\n v_mov_b32 v103, %4\n" \
    "v_mov_b32 v104, %5\n v_mov_b32 v105, %5\n v_mov_b32 v106, %5\n v_mov_b32 v107, %5\n" \
    "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n" \
    "v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n" \
    "s_nop 7\n"
#define CH1 MFMA
#define CH2 MFMA MFMA
#define CH6 MFMA MFMA MFMA MFMA MFMA MFMA
#define ACT0 ""
#define ACT1 "v_mov_b32 v100, %6\n v_mov_b32 v101, %6\n v_mov_b32 v102, %6\n v_mov_b32 v103, %6\n"
#define ACT2 "v_mov_b32 v104, %6\n v_mov_b32 v105, %6\n v_mov_b32 v106, %6\n v_mov_b32 v107, %6\n"
#define ACT3 "ds_read_b128 v[100:103], %7\n"
#define ACT4 "ds_read_b128 v[104:107], %7\n"
#define ACT5 "global_load_dwordx4 v[100:103], %8, off\n"
#define WAIT0 "s_nop 11\n"
#define WAIT1 "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
#define WAIT2 ""
#define READ \
    "v_mov_b32 %0, v108\n v_mov_b32 %1, v113\n v_mov_b32 %2, v118\n v_mov_b32 %3, v123\n s_waitcnt vmcnt(0) lgkmcnt(0)\n"

#define SEQ(CHS, ACTS, WAITS) \
    asm volatile(INIT CHS ACTS WAITS READ : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(bb), "v"(aa), "v"(gg), "v"(lds_addr), "v"(gptr) : CLOB)

template <int CH, int ACT, int WAIT>
__global__ __launch_bounds__(256) void k_probe(const float4* garbage, unsigned long long* bad, float* first_bad, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, n = lane & 31;
    for (int k = threadIdx.x; k < 256 * 4; k += 256) lds[k] = __uint_as_float(0x42804280u);     // 64.0 | 64.0 in bfloat16
    __syncthreads();
    const float b = 1.f + (float)(n & 7);
    const unsigned bb = (__float_as_uint(b) >> 16) * 0x10001u, aa = 0x3f803f80u, gg = 0x42804280u;
    const unsigned lds_addr = (unsigned)(threadIdx.x * 16);
    const float4* gptr = garbage + threadIdx.x;
    const float expect = (float)CH * 16.f * b;
    unsigned long long nb = 0;
    float fb = 0.f;
    for (int it = 0; it < iters; ++it) {
        float o0, o1, o2, o3;
        if (CH == 1) {
            if (ACT == 0) { if (WAIT == 0) SEQ(CH1, ACT0, WAIT0); else if (WAIT == 1) SEQ(CH1, ACT0, WAIT1); else SEQ(CH1, ACT0, WAIT2); }
            if (ACT == 1) { if (WAIT == 0) SEQ(CH1, ACT1, WAIT0); else if (WAIT == 1) SEQ(CH1, ACT1, WAIT1); else SEQ(CH1, ACT1, WAIT2); }
            if (ACT == 2) { if (WAIT == 0) SEQ(CH1, ACT2, WAIT0); else if (WAIT == 1) SEQ(CH1, ACT2, WAIT1); else SEQ(CH1, ACT2, WAIT2); }
            if (ACT == 3) { if (WAIT == 0) SEQ(CH1, ACT3, WAIT0); else if (WAIT == 1) SEQ(CH1, ACT3, WAIT1); else SEQ(CH1, ACT3, WAIT2); }
            if (ACT == 4) { if (WAIT == 0) SEQ(CH1, ACT4, WAIT0); else if (WAIT == 1) SEQ(CH1, ACT4, WAIT1); else SEQ(CH1, ACT4, WAIT2); }
            if (ACT == 5) { if (WAIT == 0) SEQ(CH1, ACT5, WAIT0); else if (WAIT == 1) SEQ(CH1, ACT5, WAIT1); else SEQ(CH1, ACT5, WAIT2); }
        } else if (CH == 2) {
            if (ACT == 0) { if (WAIT == 0) SEQ(CH2, ACT0, WAIT0); else if (WAIT == 1) SEQ(CH2, ACT0, WAIT1); else SEQ(CH2, ACT0, WAIT2); }
            if (ACT == 1) { if (WAIT == 0) SEQ(CH2, ACT1, WAIT0); else if (WAIT == 1) SEQ(CH2, ACT1, WAIT1); else SEQ(CH2, ACT1, WAIT2); }
            if (ACT == 2) { if (WAIT == 0) SEQ(CH2, ACT2, WAIT0); else if (WAIT == 1) SEQ(CH2, ACT2, WAIT1); else SEQ(CH2, ACT2, WAIT2); }
            if (ACT == 3) { if (WAIT == 0) SEQ(CH2, ACT3, WAIT0); else if (WAIT == 1) SEQ(CH2, ACT3, WAIT1); else SEQ(CH2, ACT3, WAIT2); }
            if (ACT == 4) { if (WAIT == 0) SEQ(CH2, ACT4, WAIT0); else if (WAIT == 1) SEQ(CH2, ACT4, WAIT1); else SEQ(CH2, ACT4, WAIT2); }
            if (ACT == 5) { if (WAIT == 0) SEQ(CH2, ACT5, WAIT0); else if (WAIT == 1) SEQ(CH2, ACT5, WAIT1); else SEQ(CH2, ACT5, WAIT2); }
        } else {
            if (ACT == 0) { if (WAIT == 0) SEQ(CH6, ACT0, WAIT0); else if (WAIT == 1) SEQ(CH6, ACT0, WAIT1); else SEQ(CH6, ACT0, WAIT2); }
            if (ACT == 1) { if (WAIT == 0) SEQ(CH6, ACT1, WAIT0); else if (WAIT == 1) SEQ(CH6, ACT1, WAIT1); else SEQ(CH6, ACT1, WAIT2); }
            if (ACT == 2) { if (WAIT == 0) SEQ(CH6, ACT2, WAIT0); else if (WAIT == 1) SEQ(CH6, ACT2, WAIT1); else SEQ(CH6, ACT2, WAIT2); }
            if (ACT == 3) { if (WAIT == 0) SEQ(CH6, ACT3, WAIT0); else if (WAIT == 1) SEQ(CH6, ACT3, WAIT1); else SEQ(CH6, ACT3, WAIT2); }
            if (ACT == 4) { if (WAIT == 0) SEQ(CH6, ACT4, WAIT0); else if (WAIT == 1) SEQ(CH6, ACT4, WAIT1); else SEQ(CH6, ACT4, WAIT2); }
            if (ACT == 5) { if (WAIT == 0) SEQ(CH6, ACT5, WAIT0); else if (WAIT == 1) SEQ(CH6, ACT5, WAIT1); else SEQ(CH6, ACT5, WAIT2); }
        }
        const float o[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (o[k] != expect) { ++nb; if (fb == 0.f) fb = o[k] - expect + 1e-30f; }
    }
    if (nb) {
        atomicAdd(bad + (n >> 4), nb);                     // [0]: columns 0..15, [1]: columns 16..31
        atomicExch(first_bad, fb);
    }
}


// ---- second experiment: a dependent chain (SrcC = the previous MFMA's vDst, the case the compiler pads with no wait state at all) with
// G idle cycles between consecutive MFMAs; the long wait before the read.  ALT = 1: two accumulators alternate (v[108:123], v[124:139]),
// i.e. every MFMA depends on the one before the previous one
#define MFMB "v_mfma_f32_32x32x16_bf16 v[124:139], v[104:107], v[100:103], v[124:139]\n"
#define INITB \
    "v_mov_b32 v124, 0\n v_mov_b32 v125, 0\n v_mov_b32 v126, 0\n v_mov_b32 v127, 0\n v_mov_b32 v128, 0\n v_mov_b32 v129, 0\n v_mov_b32 v130, 0\n v_mov_b32 v131, 0\n" \
    "v_mov_b32 v132, 0\n v_mov_b32 v133, 0\n v_mov_b32 v134, 0\n v_mov_b32 v135, 0\n v_mov_b32 v136, 0\n v_mov_b32 v137, 0\n v_mov_b32 v138, 0\n v_mov_b32 v139, 0\n"
#define CLOBB "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139"
#define READB "v_mov_b32 %0, v108\n v_mov_b32 %1, v123\n v_mov_b32 %2, v124\n v_mov_b32 %3, v139\n"
#define GAPSEQ(GS) asm volatile(INITB INIT MFMA GS MFMA GS MFMA GS MFMA GS MFMA GS MFMA WAIT1 READB : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(bb), "v"(aa) : CLOB, CLOBB)
#define ALTSEQ(GS) asm volatile(INITB INIT MFMA GS MFMB GS MFMA GS MFMB GS MFMA GS MFMB WAIT1 READB : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(bb), "v"(aa) : CLOB, CLOBB)
#define GAPCASE(N) if constexpr (G == N) { if constexpr (ALT) ALTSEQ("s_nop " #N "\n"); else GAPSEQ("s_nop " #N "\n"); }
template <int G, bool ALT>
__global__ __launch_bounds__(256) void k_gap(unsigned long long* bad, float* first_bad, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, n = lane & 31;
    const float b = 1.f + (float)(n & 7);
    const unsigned bb = (__float_as_uint(b) >> 16) * 0x10001u, aa = 0x3f803f80u;
    const float e0 = (ALT ? 3.f : 6.f) * 16.f * b, e1 = ALT ? 3.f * 16.f * b : 0.f;
    unsigned long long nb = 0;
    float fb = 0.f;
    for (int it = 0; it < iters; ++it) {
        float o0, o1, o2, o3;
        if constexpr (G < 0) { if constexpr (ALT) ALTSEQ(""); else GAPSEQ(""); }
        GAPCASE(0) GAPCASE(1) GAPCASE(2) GAPCASE(3) GAPCASE(4) GAPCASE(5) GAPCASE(6) GAPCASE(7) GAPCASE(8) GAPCASE(9) GAPCASE(10) GAPCASE(11) GAPCASE(12)
        GAPCASE(13) GAPCASE(14) GAPCASE(15)
        if constexpr (G == 31) { if constexpr (ALT) ALTSEQ("s_nop 15\n s_nop 15\n"); else GAPSEQ("s_nop 15\n s_nop 15\n"); }
        if constexpr (G == 47) { if constexpr (ALT) ALTSEQ("s_nop 15\n s_nop 15\n s_nop 15\n"); else GAPSEQ("s_nop 15\n s_nop 15\n s_nop 15\n"); }
        const float o[4] = {o0, o1, o2, o3};
        const float e[4] = {e0, e0, e1, e1};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (o[k] != e[k]) { ++nb; if (fb == 0.f) fb = o[k] - e[k] + 1e-30f; }
    }
    if (nb) { atomicAdd(bad + (n >> 4), nb); atomicExch(first_bad, fb); }
}

template <int G, bool ALT>
static void run_gap(unsigned long long* bad, float* first_bad) {
    static const int occ_lds[3] = {150 * 1024, 72 * 1024, 36 * 1024};
    static const int occ[3] = {1, 2, 4};
    hipFuncSetAttribute((const void*)k_gap<G, ALT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%s chain, gap s_nop %-3d                            ", ALT ? "alternating" : "dependent  ", G);
    for (int o = 0; o < 3; ++o) {
        hipMemset(bad, 0, 16); hipMemset(first_bad, 0, 4);
        const int iters = 4000, wgs = 256 * occ[o] * 2;
        hipLaunchKernelGGL((k_gap<G, ALT>), dim3(wgs), dim3(256), occ_lds[o], 0, bad, first_bad, iters);
        hipDeviceSynchronize();
        unsigned long long hb[2]; float fb;
        hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(&fb, first_bad, 4, hipMemcpyDeviceToHost);
        const double total = (double)wgs * 256 * iters * 4;
        printf("  | %d w/SIMD: bad %.3g (cols 0-15 %.3g, 16-31 %.3g) d=%g", occ[o], (double)(hb[0] + hb[1]) / total, (double)hb[0] / total * 2, (double)hb[1] / total * 2, fb);
    }
    printf("\n");
}

// ---- third experiment: the first one again, but with 16 idle cycles between the MFMAs of the chain, so that the MFMAs of the other waves
// on the SIMD interleave with this wave's (back-to-back chains keep the pipe to themselves and hide any "issued but not started" window)
#define G15 "s_nop 15\n"
#define CSEQ(ACTS, WAITS) \
    asm volatile(INIT MFMA G15 MFMA G15 MFMA G15 MFMA G15 MFMA G15 MFMA ACTS WAITS READ : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(bb), "v"(aa), "v"(gg), "v"(lds_addr), "v"(gptr) : CLOB)
template <int ACT, int WAIT>
__global__ __launch_bounds__(256) void k_contend(const float4* garbage, unsigned long long* bad, float* first_bad, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, n = lane & 31;
    for (int k = threadIdx.x; k < 256 * 4; k += 256) lds[k] = __uint_as_float(0x42804280u);
    __syncthreads();
    const float b = 1.f + (float)(n & 7);
    const unsigned bb = (__float_as_uint(b) >> 16) * 0x10001u, aa = 0x3f803f80u, gg = 0x42804280u;
    const unsigned lds_addr = (unsigned)(threadIdx.x * 16);
    const float4* gptr = garbage + threadIdx.x;
    const float expect = 6.f * 16.f * b;
    unsigned long long nb = 0;
    float fb = 0.f;
    // de-phase the waves: a different number of idle cycles per wave and iteration
    for (int it = 0; it < iters; ++it) {
        float o0, o1, o2, o3;
        const int skew = (it * 7 + (threadIdx.x >> 6) * 3 + blockIdx.x) & 15;
        for (int k = 0; k < skew; ++k) asm volatile("s_nop 3");
        if constexpr (ACT == 0) { if constexpr (WAIT == 0) CSEQ(ACT0, WAIT0); else CSEQ(ACT0, WAIT1); }
        if constexpr (ACT == 1) { if constexpr (WAIT == 0) CSEQ(ACT1, WAIT0); else CSEQ(ACT1, WAIT1); }
        if constexpr (ACT == 2) { if constexpr (WAIT == 0) CSEQ(ACT2, WAIT0); else CSEQ(ACT2, WAIT1); }
        if constexpr (ACT == 3) { if constexpr (WAIT == 0) CSEQ(ACT3, WAIT0); else CSEQ(ACT3, WAIT1); }
        if constexpr (ACT == 4) { if constexpr (WAIT == 0) CSEQ(ACT4, WAIT0); else CSEQ(ACT4, WAIT1); }
        if constexpr (ACT == 5) { if constexpr (WAIT == 0) CSEQ(ACT5, WAIT0); else CSEQ(ACT5, WAIT1); }
        const float o[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (o[k] != expect) { ++nb; if (fb == 0.f) fb = o[k] - expect + 1e-30f; }
    }
    if (nb) { atomicAdd(bad + (n >> 4), nb); atomicExch(first_bad, fb); }
}
template <int ACT, int WAIT>
static void run_contend(const char* what, const float4* garbage, unsigned long long* bad, float* first_bad) {
    static const int occ_lds[3] = {150 * 1024, 72 * 1024, 36 * 1024};
    static const int occ[3] = {1, 2, 4};
    hipFuncSetAttribute((const void*)k_contend<ACT, WAIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%-52s", what);
    for (int o = 0; o < 3; ++o) {
        hipMemset(bad, 0, 16); hipMemset(first_bad, 0, 4);
        const int iters = 4000, wgs = 256 * occ[o] * 2;
        hipLaunchKernelGGL((k_contend<ACT, WAIT>), dim3(wgs), dim3(256), occ_lds[o], 0, garbage, bad, first_bad, iters);
        hipDeviceSynchronize();
        unsigned long long hb[2]; float fb;
        hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(&fb, first_bad, 4, hipMemcpyDeviceToHost);
        const double total = (double)wgs * 256 * iters * 4;
        printf("  | %d w/SIMD: bad %.3g (cols 0-15 %.3g, 16-31 %.3g) d=%g", occ[o], (double)(hb[0] + hb[1]) / total, (double)hb[0] / total * 2, (double)hb[1] / total * 2, fb);
    }
    printf("\n");
}

// ---- fourth experiment: the accumulate pattern of the x6 kernels - three accumulators, six MFMAs per K step, eight steps, the first MFMA
// into each accumulator with the constant 0 as SrcC - with and without idle cycles between the MFMAs:
//   a0 += A1 B1 ; a1 += A1 B2 ; a2 += A2 B2 ; a1 += A2 B1 ; a2 += A1 B3 ; a2 += A3 B1       (A1, A2, A3 = 1, 2, 4;  B1, B2, B3 = b, 8 b, 64 b)
// A: v[100:111], B: v[112:123], a0 = v[124:139], a1 = v[140:155], a2 = v[156:171]
#define XA1 "v[100:103]"
#define XA2 "v[104:107]"
#define XA3 "v[108:111]"
#define XB1 "v[112:115]"
#define XB2 "v[116:119]"
#define XB3 "v[120:123]"
#define XM(ACC, A, B, C) "v_mfma_f32_32x32x16_bf16 " ACC ", " A ", " B ", " C "\n"
#define XC0 "v[124:139]"
#define XC1 "v[140:155]"
#define XC2 "v[156:171]"
#define XSTEP0(G) XM(XC0, XA1, XB1, "0") G XM(XC1, XA1, XB2, "0") G XM(XC2, XA2, XB2, "0") G XM(XC1, XA2, XB1, XC1) G XM(XC2, XA1, XB3, XC2) G XM(XC2, XA3, XB1, XC2) G
#define XSTEP(G) XM(XC0, XA1, XB1, XC0) G XM(XC1, XA1, XB2, XC1) G XM(XC2, XA2, XB2, XC2) G XM(XC1, XA2, XB1, XC1) G XM(XC2, XA1, XB3, XC2) G XM(XC2, XA3, XB1, XC2) G
#define XINIT \
    "v_mov_b32 v100, %6\n v_mov_b32 v101, %6\n v_mov_b32 v102, %6\n v_mov_b32 v103, %6\n v_mov_b32 v104, %7\n v_mov_b32 v105, %7\n v_mov_b32 v106, %7\n v_mov_b32 v107, %7\n" \
    "v_mov_b32 v108, %8\n v_mov_b32 v109, %8\n v_mov_b32 v110, %8\n v_mov_b32 v111, %8\n v_mov_b32 v112, %9\n v_mov_b32 v113, %9\n v_mov_b32 v114, %9\n v_mov_b32 v115, %9\n" \
    "v_mov_b32 v116, %10\n v_mov_b32 v117, %10\n v_mov_b32 v118, %10\n v_mov_b32 v119, %10\n v_mov_b32 v120, %11\n v_mov_b32 v121, %11\n v_mov_b32 v122, %11\n v_mov_b32 v123, %11\n s_nop 7\n"
#define XREAD "v_mov_b32 %0, v124\n v_mov_b32 %1, v139\n v_mov_b32 %2, v140\n v_mov_b32 %3, v155\n v_mov_b32 %4, v156\n v_mov_b32 %5, v171\n"
#define XSEQ(G) asm volatile(XINIT XSTEP0(G) XSTEP(G) XSTEP(G) XSTEP(G) XSTEP(G) XSTEP(G) XSTEP(G) XSTEP(G) WAIT1 XREAD \
    : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]) : "v"(pa1), "v"(pa2), "v"(pa3), "v"(pb1), "v"(pb2), "v"(pb3) \
    : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", \
      "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", \
      "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", \
      "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171")
__device__ __forceinline__ unsigned bf2(float x) { return (__float_as_uint(x) >> 16) * 0x10001u; }
template <int GAP>
__global__ __launch_bounds__(256) void k_x6pat(unsigned long long* bad, float* first_bad, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, n = lane & 31;
    const float b = 1.f + (float)(n & 7);
    const unsigned pa1 = bf2(1.f), pa2 = bf2(2.f), pa3 = bf2(4.f), pb1 = bf2(b), pb2 = bf2(8.f * b), pb3 = bf2(64.f * b);
    const float e[6] = {8 * 16.f * b, 8 * 16.f * b, 8 * 160.f * b, 8 * 160.f * b, 8 * 16.f * 84.f * b, 8 * 16.f * 84.f * b};
    unsigned long long nb = 0;
    float fb = 0.f;
    for (int it = 0; it < iters; ++it) {
        float o[6];
        const int skew = (it * 7 + (threadIdx.x >> 6) * 3 + blockIdx.x) & 15;
        for (int k = 0; k < skew; ++k) asm volatile("s_nop 3");
        if constexpr (GAP == 0) XSEQ("");
        if constexpr (GAP == 1) XSEQ("s_nop 15\n");
        if constexpr (GAP == 2) XSEQ("s_nop 3\n");
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (o[k] != e[k]) { ++nb; if (fb == 0.f) fb = o[k] - e[k] + 1e-30f; }
    }
    if (nb) { atomicAdd(bad + (n >> 4), nb); atomicExch(first_bad, fb); }
}
template <int GAP>
static void run_x6pat(const char* what, unsigned long long* bad, float* first_bad) {
    static const int occ_lds[3] = {150 * 1024, 72 * 1024, 36 * 1024};
    static const int occ[3] = {1, 2, 4};
    hipFuncSetAttribute((const void*)k_x6pat<GAP>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%-52s", what);
    for (int o = 0; o < 3; ++o) {
        hipMemset(bad, 0, 16); hipMemset(first_bad, 0, 4);
        const int iters = 2000, wgs = 256 * occ[o] * 2;
        hipLaunchKernelGGL((k_x6pat<GAP>), dim3(wgs), dim3(256), occ_lds[o], 0, bad, first_bad, iters);
        hipDeviceSynchronize();
        unsigned long long hb[2]; float fb;
        hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(&fb, first_bad, 4, hipMemcpyDeviceToHost);
        const double total = (double)wgs * 256 * iters * 6;
        printf("  | %d w/SIMD: bad %.3g (cols 0-15 %.3g, 16-31 %.3g) d=%g", occ[o], (double)(hb[0] + hb[1]) / total, (double)hb[0] / total * 2, (double)hb[1] / total * 2, fb);
    }
    printf("\n");
}

// ---- fifth experiment: does s_waitcnt release a consumer before the returned data is in ALL lanes of the register?
// B (or SrcC) comes from LDS straight in front of the MFMA; the LDS region alternates between two contents per iteration, so a lane that
// still holds the previous iteration's value shows.  MODE 0: ds_read B ; lgkmcnt(0) ; MFMA     MODE 1: two reads in flight, lgkmcnt(1) ; MFMA on the first
// MODE 2: ds_read x4 into the accumulator ; lgkmcnt(0) ; MFMA with it as SrcC       MODE 3: global_load B ; vmcnt(0) ; MFMA
template <int MODE>
__global__ __launch_bounds__(256) void k_land(const float4* g2, unsigned long long* bad, float* first_bad, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, n = lane & 31;
    // region r (0/1), thread slot: packed bfloat16 pairs of b_r(n) = 1 + r + (n & 7); region 2/3: fp32 accumulator seeds 100 (r + 1)
    for (int r = 0; r < 2; ++r) {
        const float b = 1.f + r + (float)(n & 7);
        const unsigned pk = (__float_as_uint(b) >> 16) * 0x10001u;
        for (int k = 0; k < 4; ++k) lds[r * 1024 + threadIdx.x * 4 + k] = __uint_as_float(pk);
        for (int k = 0; k < 4; ++k) lds[2048 + r * 1024 + threadIdx.x * 4 + k] = 100.f * (r + 1);
    }
    __syncthreads();
    const unsigned aa = 0x3f803f80u;
    unsigned long long nb = 0;
    float fb = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int r = it & 1;
        const float b = 1.f + r + (float)(n & 7);
        const unsigned addr = (unsigned)((r * 1024 + threadIdx.x * 4) * 4), addr2 = (unsigned)(((r ^ 1) * 1024 + threadIdx.x * 4) * 4);
        const unsigned addrc = (unsigned)((2048 + r * 1024 + threadIdx.x * 4) * 4);
        const float4* gp = g2 + r * 256 + threadIdx.x;
        float o0, o1, o2, o3, expect = 16.f * b;
        const int skew = (it * 7 + (threadIdx.x >> 6) * 3 + blockIdx.x) & 15;
        for (int k = 0; k < skew; ++k) asm volatile("s_nop 3");
#define LINIT "v_mov_b32 v104, %5\n v_mov_b32 v105, %5\n v_mov_b32 v106, %5\n v_mov_b32 v107, %5\n" \
    "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n v_mov_b32 v112, 0\n v_mov_b32 v113, 0\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0\n" \
    "v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n s_nop 7\n"
#define LOUT : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(addr), "v"(aa), "v"(addr2), "v"(addrc), "v"(gp) : CLOB, "v124", "v125", "v126", "v127"
        if constexpr (MODE == 0) asm volatile(LINIT "ds_read_b128 v[100:103], %4\n s_waitcnt lgkmcnt(0)\n" MFMA WAIT1 READ LOUT);
        if constexpr (MODE == 1) asm volatile(LINIT "ds_read_b128 v[100:103], %4\n ds_read_b128 v[124:127], %6\n s_waitcnt lgkmcnt(1)\n" MFMA WAIT1 READ LOUT);
        if constexpr (MODE == 2) {
            asm volatile(LINIT "ds_read_b128 v[100:103], %4\n s_waitcnt lgkmcnt(0)\n s_nop 7\n"
                         "ds_read_b128 v[108:111], %7\n ds_read_b128 v[112:115], %7\n ds_read_b128 v[116:119], %7\n ds_read_b128 v[120:123], %7\n s_waitcnt lgkmcnt(0)\n" MFMA WAIT1 READ LOUT);
            expect += 100.f * (r + 1);
        }
        if constexpr (MODE == 3) asm volatile(LINIT "global_load_dwordx4 v[100:103], %8, off\n s_waitcnt vmcnt(0)\n" MFMA WAIT1 READ LOUT);
        const float o[4] = {o0, o1, o2, o3};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (o[k] != expect) { ++nb; if (fb == 0.f) fb = o[k] - expect + 1e-30f; }
    }
    if (nb) { atomicAdd(bad + (n >> 4), nb); atomicExch(first_bad, fb); }
}
template <int MODE>
static void run_land(const char* what, const float4* g2, unsigned long long* bad, float* first_bad) {
    static const int occ_lds[3] = {150 * 1024, 72 * 1024, 36 * 1024};
    static const int occ[3] = {1, 2, 4};
    hipFuncSetAttribute((const void*)k_land<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%-52s", what);
    for (int o = 0; o < 3; ++o) {
        hipMemset(bad, 0, 16); hipMemset(first_bad, 0, 4);
        const int iters = 4000, wgs = 256 * occ[o] * 2;
        hipLaunchKernelGGL((k_land<MODE>), dim3(wgs), dim3(256), occ_lds[o], 0, g2, bad, first_bad, iters);
        hipDeviceSynchronize();
        unsigned long long hb[2]; float fb;
        hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(&fb, first_bad, 4, hipMemcpyDeviceToHost);
        const double total = (double)wgs * 256 * iters * 4;
        printf("  | %d w/SIMD: bad %.3g (cols 0-15 %.3g, 16-31 %.3g) d=%g", occ[o], (double)(hb[0] + hb[1]) / total, (double)hb[0] / total * 2, (double)hb[1] / total * 2, fb);
    }
    printf("\n");
}

// ---- sixth experiment: the software-managed "VALU writes VCC -> VALU reads VCC" distance (2 wait states on gfx940+, the s_nop 1 the
// compiler puts between v_cmp and v_cndmask) while OTHER waves on the SIMD issue MFMAs.  Even workgroups run the VALU role, odd ones
// a stream of 16-bit MFMAs (MROLE 0: none, all workgroups VALU; 1: back to back; 2: with s_nop 3 gaps).  The predicate alternates per lane
// and iteration, so a v_cndmask that sees the previous v_cmp's mask in some lanes shows.  Reported per quarter wave.
template <int NOPS, int MROLE>
__global__ __launch_bounds__(256) void k_vcc(unsigned long long* badq, int iters) {
    const int lane = threadIdx.x & 63;
    if (MROLE != 0 && (blockIdx.x & 1)) {
        const unsigned aa = 0x3f803f80u;
        float o0 = 0.f;
        for (int it = 0; it < iters * 2; ++it) {
            if constexpr (MROLE == 1) asm volatile(MFMA MFMA MFMA MFMA MFMA MFMA MFMA MFMA ::: CLOB);
            else asm volatile(MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" ::: CLOB);
        }
        asm volatile("s_nop 15\n s_nop 15\n v_mov_b32 %0, v108" : "=v"(o0) :: CLOB);
        if (o0 == 12345.f && aa == 0) badq[7] = 1;          // keep the stream alive
        return;
    }
    unsigned long long nb = 0;
    for (int it = 0; it < iters * 8; ++it) {
        const unsigned pred = (unsigned)((it + lane) & 1);
        unsigned r;
        const unsigned a = 0x11110000u + it, b = 0x22220000u + it;
        if constexpr (NOPS == 0) asm volatile("v_cmp_eq_u32 vcc, 0, %1\n s_nop 0\n v_cndmask_b32 %0, %2, %3, vcc" : "=v"(r) : "v"(pred), "v"(a), "v"(b) : "vcc");
        if constexpr (NOPS == 1) asm volatile("v_cmp_eq_u32 vcc, 0, %1\n s_nop 1\n v_cndmask_b32 %0, %2, %3, vcc" : "=v"(r) : "v"(pred), "v"(a), "v"(b) : "vcc");
        if constexpr (NOPS == 2) asm volatile("v_cmp_eq_u32 vcc, 0, %1\n s_nop 3\n v_cndmask_b32 %0, %2, %3, vcc" : "=v"(r) : "v"(pred), "v"(a), "v"(b) : "vcc");
        if constexpr (NOPS == 3) asm volatile("v_cmp_eq_u32 vcc, 0, %1\n v_cndmask_b32 %0, %2, %3, vcc" : "=v"(r) : "v"(pred), "v"(a), "v"(b) : "vcc");
        const unsigned expect = pred == 0 ? b : a;             // vcc ? src1 : src0
        if (r != expect) ++nb;
    }
    if (nb) atomicAdd(badq + (lane >> 4), nb);
}
template <int NOPS, int MROLE>
static void run_vcc(const char* what, unsigned long long* bad) {
    unsigned long long* bq; hipMalloc(&bq, 64);
    static const int occ_lds[3] = {150 * 1024, 72 * 1024, 36 * 1024};
    static const int occ[3] = {1, 2, 4};
    hipFuncSetAttribute((const void*)k_vcc<NOPS, MROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%-52s", what);
    for (int o = 0; o < 3; ++o) {
        hipMemset(bq, 0, 64);
        const int iters = 4000, wgs = 256 * occ[o] * 2;
        hipLaunchKernelGGL((k_vcc<NOPS, MROLE>), dim3(wgs), dim3(256), occ_lds[o], 0, bq, iters);
        hipDeviceSynchronize();
        unsigned long long hb[8]; hipMemcpy(hb, bq, 64, hipMemcpyDeviceToHost);
        const double total = (double)(MROLE ? wgs / 2 : wgs) * 64 * iters * 8;          // per quarter wave
        printf("  | %d w/SIMD: bad by quarter %.3g %.3g %.3g %.3g", occ[o], hb[0] / total, hb[1] / total, hb[2] / total, hb[3] / total);
    }
    printf("\n");
    hipFree(bq);
}

// ---- seventh experiment: VALU instructions executing while global_load_dwordx4 results land in OTHER registers of the same wave
// (the x6 kernels prefetch weights straight into registers while the encoder / activation VALU code runs).  Six loads of a constant
// 64.0 pattern into v[100:123]; 48 VALU instructions that sum x = 1 + (lane & 7) (or, packed, x and 2 x) into an accumulator; a lane that
// ever reads something else than x shows in the sum.  OP 0: v_fma_f32   1: v_pk_fma_f32   2: v_mul_f32 by a literal + v_add_f32
// MROLE as in the sixth experiment (odd workgroups run MFMAs).
#define V8(X) X X X X X X X X
template <int OP, int MROLE>
__global__ __launch_bounds__(256) void k_landvalu(const float4* garbage, unsigned long long* badq, int iters) {
    const int lane = threadIdx.x & 63;
    if (MROLE != 0 && (blockIdx.x & 1)) {
        float o0 = 0.f;
        for (int it = 0; it < iters; ++it) {
            if constexpr (MROLE == 1) asm volatile(MFMA MFMA MFMA MFMA MFMA MFMA MFMA MFMA ::: CLOB);
            else asm volatile(MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" ::: CLOB);
        }
        asm volatile("s_nop 15\n s_nop 15\n v_mov_b32 %0, v108" : "=v"(o0) :: CLOB);
        if (o0 == 12345.f) badq[7] = 1;
        return;
    }
    const float x = 1.f + (float)(lane & 7);
    unsigned long long nb = 0;
    for (int it = 0; it < iters; ++it) {
        const float4* gp = garbage + ((threadIdx.x + it * 17) & 255);
        float r0, r1;
        const int skew = (it * 7 + (threadIdx.x >> 6) * 3 + blockIdx.x) & 15;
        for (int k = 0; k < skew; ++k) asm volatile("s_nop 3");
#define LV_LOADS "global_load_dwordx4 v[100:103], %2, off\n global_load_dwordx4 v[104:107], %2, off offset:1024\n global_load_dwordx4 v[108:111], %2, off offset:2048\n" \
                 "global_load_dwordx4 v[112:115], %2, off offset:3072\n global_load_dwordx4 v[116:119], %2, off offset:512\n global_load_dwordx4 v[120:123], %2, off offset:1536\n"
#define LV_OUT : "=&v"(r0), "=&v"(r1) : "v"(gp), "v"(x), "v"(2.f * x), "v"(1.f) : CLOB, "v124", "v125", "v126", "v127", "v128", "v129"
        if constexpr (OP == 0)
            asm volatile("v_mov_b32 v124, 0\n v_mov_b32 v125, 0\n" LV_LOADS
                         V8("v_fma_f32 v124, %3, %5, v124\n v_fma_f32 v125, %4, %5, v125\n") V8("v_fma_f32 v124, %3, %5, v124\n v_fma_f32 v125, %4, %5, v125\n")
                         V8("v_fma_f32 v124, %3, %5, v124\n v_fma_f32 v125, %4, %5, v125\n")
                         "s_waitcnt vmcnt(0)\n v_mov_b32 %0, v124\n v_mov_b32 %1, v125\n" LV_OUT);
        if constexpr (OP == 1)
            asm volatile("v_mov_b32 v124, 0\n v_mov_b32 v125, 0\n v_mov_b32 v126, %3\n v_mov_b32 v127, %4\n v_mov_b32 v128, %5\n v_mov_b32 v129, %5\n" LV_LOADS
                         V8("v_pk_fma_f32 v[124:125], v[126:127], v[128:129], v[124:125]\n") V8("v_pk_fma_f32 v[124:125], v[126:127], v[128:129], v[124:125]\n")
                         V8("v_pk_fma_f32 v[124:125], v[126:127], v[128:129], v[124:125]\n")
                         "s_waitcnt vmcnt(0)\n v_mov_b32 %0, v124\n v_mov_b32 %1, v125\n" LV_OUT);
        if constexpr (OP == 2)
            asm volatile("v_mov_b32 v124, 0\n v_mov_b32 v125, 0\n" LV_LOADS
                         V8("v_mul_f32 v126, 0x40000000, %3\n v_add_f32 v124, v124, %3\n v_add_f32 v125, v125, v126\n") V8("v_mul_f32 v126, 0x40000000, %3\n v_add_f32 v124, v124, %3\n v_add_f32 v125, v125, v126\n")
                         V8("v_mul_f32 v126, 0x40000000, %3\n v_add_f32 v124, v124, %3\n v_add_f32 v125, v125, v126\n")
                         "s_waitcnt vmcnt(0)\n v_mov_b32 %0, v124\n v_mov_b32 %1, v125\n" LV_OUT);
        if (r0 != 24.f * x) ++nb;
        if (r1 != 48.f * x) ++nb;
    }
    if (nb) atomicAdd(badq + (lane >> 4), nb);
}
template <int OP, int MROLE>
static void run_landvalu(const char* what, const float4* garbage) {
    unsigned long long* bq; hipMalloc(&bq, 64);
    static const int occ_lds[3] = {150 * 1024, 72 * 1024, 36 * 1024};
    static const int occ[3] = {1, 2, 4};
    hipFuncSetAttribute((const void*)k_landvalu<OP, MROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%-52s", what);
    for (int o = 0; o < 3; ++o) {
        hipMemset(bq, 0, 64);
        const int iters = 20000, wgs = 256 * occ[o] * 2;
        hipLaunchKernelGGL((k_landvalu<OP, MROLE>), dim3(wgs), dim3(256), occ_lds[o], 0, garbage, bq, iters);
        hipDeviceSynchronize();
        unsigned long long hb[8]; hipMemcpy(hb, bq, 64, hipMemcpyDeviceToHost);
        const double total = (double)(MROLE ? wgs / 2 : wgs) * 64 * iters * 2;
        printf("  | %d w/SIMD: bad by quarter %.3g %.3g %.3g %.3g", occ[o], hb[0] / total, hb[1] / total, hb[2] / total, hb[3] / total);
    }
    printf("\n");
    hipFree(bq);
}

// ---- eighth experiment: the positional encoder of the velocity net (compiled C++: Cody-Waite reduction + minimax polynomials, what the
// compiler turns into v_rndne / v_cvt_i32 / v_pk_fma with SGPR constants / v_cmp + v_cndmask) evaluated twice on the same input while
// the odd workgroups run 16-bit MFMAs; any difference between the two evaluations is a corrupted VALU result.
__device__ __forceinline__ float p_trig_sel(float a, int want_cos) {
    const float kf = rintf(a * 0.636619772367581f);
    const int k = (int)kf + want_cos;
    float r = __builtin_fmaf(kf, -1.5707963705062866f, a);
    r = __builtin_fmaf(kf, 4.371138828673793e-08f, r);
    r = __builtin_fmaf(kf, 1.7763568394002505e-15f, r);
    const float z = r * r;
    float sp = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611e-1f);
    sp = __builtin_fmaf(sp * z, r, r);
    float cp = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827e-2f);
    cp = __builtin_fmaf(cp * z, z, __builtin_fmaf(z, -0.5f, 1.0f));
    const float v = (k & 1) ? cp : sp;
    return (k & 2) ? -v : v;
}
__device__ __forceinline__ void p_encode(const float4& q, int h, float* x) {
    x[0] = h ? q.y : q.x;
    x[1] = h ? q.w : q.z;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float v = c == 0 ? q.x : (c == 1 ? q.y : (c == 2 ? q.z : q.w));
            x[2 + 4 * k + c] = p_trig_sel(v * (float)(1 << k), h);
        }
}
template <int MROLE>
__global__ __launch_bounds__(256) void k_trig(unsigned long long* badq, unsigned* slotmask, int iters) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    if (MROLE != 0 && (blockIdx.x & 1)) {
        float o0 = 0.f;
        for (int it = 0; it < iters / 2; ++it) {
            if constexpr (MROLE == 1) asm volatile(MFMA MFMA MFMA MFMA MFMA MFMA MFMA MFMA ::: CLOB);
            else asm volatile(MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" MFMA "s_nop 3\n" ::: CLOB);
        }
        asm volatile("s_nop 15\n s_nop 15\n v_mov_b32 %0, v108" : "=v"(o0) :: CLOB);
        if (o0 == 12345.f) badq[7] = 1;
        return;
    }
    unsigned long long nb = 0;
    unsigned sm = 0;
    float4 q = make_float4(0.31f + 0.013f * lane, -0.57f + 0.007f * lane, 0.11f - 0.003f * lane, 0.0125f);
    for (int it = 0; it < iters; ++it) {
        float x1[14], x2[14];
        p_encode(q, h, x1);
        float4 q2 = q;
        asm volatile("" : "+v"(q2.x), "+v"(q2.y), "+v"(q2.z), "+v"(q2.w));
        p_encode(q2, h, x2);
#pragma unroll
        for (int k = 0; k < 14; ++k)
            if (__float_as_uint(x1[k]) != __float_as_uint(x2[k])) { ++nb; sm |= 1u << k; }
        q.x += 0.0013f * x1[2]; q.y -= 0.0017f * x1[3]; q.z += 0.0011f * x1[4];          // the input moves like a point under a small velocity
        if (q.x > 2.f) q.x -= 4.f;
    }
    if (nb) { atomicAdd(badq + (lane >> 4), nb); atomicOr(slotmask, sm); }
}
template <int MROLE>
static void run_trig(const char* what) {
    unsigned long long* bq; hipMalloc(&bq, 64);
    unsigned* sm; hipMalloc(&sm, 4);
    static const int occ_lds[3] = {150 * 1024, 72 * 1024, 36 * 1024};
    static const int occ[3] = {1, 2, 4};
    hipFuncSetAttribute((const void*)k_trig<MROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%-52s", what);
    for (int o = 0; o < 3; ++o) {
        hipMemset(bq, 0, 64); hipMemset(sm, 0, 4);
        const int iters = 20000, wgs = 256 * occ[o] * 2;
        hipLaunchKernelGGL((k_trig<MROLE>), dim3(wgs), dim3(256), occ_lds[o], 0, bq, sm, iters);
        hipDeviceSynchronize();
        unsigned long long hb[8]; hipMemcpy(hb, bq, 64, hipMemcpyDeviceToHost);
        unsigned hs; hipMemcpy(&hs, sm, 4, hipMemcpyDeviceToHost);
        printf("  | %d w/SIMD: wrong slots by quarter %llu %llu %llu %llu (slot mask %x)", occ[o], hb[0], hb[1], hb[2], hb[3], hs);
    }
    printf("\n");
    hipFree(bq); hipFree(sm);
}

// ---- ninth experiment: PACKED fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with an SGPR pair) beside 16-bit MFMAs
// of other waves.  Delta debugging of the x6 kernels points here: with the compiler's SLP vectoriser off (no v_pk_* in the kernel) the
// run-to-run differences at two workgroups per CU are gone.  Y role (odd workgroups): MROLE 1 dependent chain, 3: the x6 pattern at full
// rate (three independent accumulators), 4: the same with s_nop 1 gaps.
template <int MROLE>
__global__ __launch_bounds__(256) void k_pk(unsigned long long* badq, int iters) {
    const int lane = threadIdx.x & 63;
    if (MROLE != 0 && (blockIdx.x & 1)) {
        float o[6];
        const float b = 1.f;
        const unsigned pa1 = bf2(1.f), pa2 = bf2(2.f), pa3 = bf2(4.f), pb1 = bf2(b), pb2 = bf2(8.f * b), pb3 = bf2(64.f * b);
        for (int it = 0; it < iters / 8; ++it) {
            if constexpr (MROLE == 1) { asm volatile(MFMA MFMA MFMA MFMA MFMA MFMA MFMA MFMA ::: CLOB); o[0] = 0.f; }
            if constexpr (MROLE == 3) XSEQ("");
            if constexpr (MROLE == 4) XSEQ("s_nop 1\n");
        }
        if (o[0] == 12345.f) badq[7] = 1;
        return;
    }
    const float x = 1.f + (float)(lane & 7);
    unsigned long long nb = 0;
    for (int it = 0; it < iters; ++it) {
        float r0, r1, r2, r3;
        // p = (x, 2x) ; t = p + p = (2x, 4x) ; u = t * t = (4x^2, 16x^2) ; w = t * (3, 5) + p = (7x, 22x) ; sums over 16 repetitions
        asm volatile("v_mov_b32 v124, %4\n v_mov_b32 v125, %5\n v_mov_b32 v132, 0\n v_mov_b32 v133, 0\n v_mov_b32 v134, 0\n v_mov_b32 v135, 0\n"
                     "s_mov_b32 s20, 0x40400000\n s_mov_b32 s21, 0x40a00000\n"
                     V8("v_pk_add_f32 v[126:127], v[124:125], v[124:125]\n v_pk_mul_f32 v[128:129], v[126:127], v[126:127]\n v_pk_fma_f32 v[130:131], v[126:127], s[20:21], v[124:125]\n"
                        "v_pk_add_f32 v[132:133], v[132:133], v[128:129]\n v_pk_add_f32 v[134:135], v[134:135], v[130:131]\n")
                     V8("v_pk_add_f32 v[126:127], v[124:125], v[124:125]\n v_pk_mul_f32 v[128:129], v[126:127], v[126:127]\n v_pk_fma_f32 v[130:131], v[126:127], s[20:21], v[124:125]\n"
                        "v_pk_add_f32 v[132:133], v[132:133], v[128:129]\n v_pk_add_f32 v[134:135], v[134:135], v[130:131]\n")
                     "v_mov_b32 %0, v132\n v_mov_b32 %1, v133\n v_mov_b32 %2, v134\n v_mov_b32 %3, v135\n"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x), "v"(2.f * x)
                     : "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "s20", "s21");
        if (r0 != 16.f * 4.f * x * x) ++nb;
        if (r1 != 16.f * 16.f * x * x) ++nb;
        if (r2 != 16.f * 7.f * x) ++nb;
        if (r3 != 16.f * 22.f * x) ++nb;
    }
    if (nb) atomicAdd(badq + (lane >> 4), nb);
}
template <int MROLE>
static void run_pk(const char* what) {
    unsigned long long* bq; hipMalloc(&bq, 64);
    static const int occ_lds[3] = {150 * 1024, 72 * 1024, 36 * 1024};
    static const int occ[3] = {1, 2, 4};
    hipFuncSetAttribute((const void*)k_pk<MROLE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%-52s", what);
    for (int o = 0; o < 3; ++o) {
        hipMemset(bq, 0, 64);
        const int iters = 40000, wgs = 256 * occ[o] * 2;
        hipLaunchKernelGGL((k_pk<MROLE>), dim3(wgs), dim3(256), occ_lds[o], 0, bq, iters);
        hipDeviceSynchronize();
        unsigned long long hb[8]; hipMemcpy(hb, bq, 64, hipMemcpyDeviceToHost);
        printf("  | %d w/SIMD: wrong sums by quarter %llu %llu %llu %llu", occ[o], hb[0], hb[1], hb[2], hb[3]);
    }
    printf("\n");
    hipFree(bq);
}

template <int CH, int ACT, int WAIT>
static void run(const char* what, const float4* garbage, unsigned long long* bad, float* first_bad) {
    static const int occ_lds[3] = {150 * 1024, 72 * 1024, 36 * 1024};     // 1, 2, 4 workgroups of 4 waves per CU
    static const int occ[3] = {1, 2, 4};
    hipFuncSetAttribute((const void*)k_probe<CH, ACT, WAIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    printf("%-52s", what);
    for (int o = 0; o < 3; ++o) {
        hipMemset(bad, 0, 16); hipMemset(first_bad, 0, 4);
        const int iters = 4000, wgs = 256 * occ[o] * 2;
        hipLaunchKernelGGL((k_probe<CH, ACT, WAIT>), dim3(wgs), dim3(256), occ_lds[o], 0, garbage, bad, first_bad, iters);
        hipDeviceSynchronize();
        unsigned long long hb[2]; float fb;
        hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(&fb, first_bad, 4, hipMemcpyDeviceToHost);
        const double total = (double)wgs * 256 * iters * 4;
        printf("  | %d w/SIMD: bad %.3g (cols 0-15 %.3g, 16-31 %.3g) d=%g", occ[o], (double)(hb[0] + hb[1]) / total, (double)hb[0] / total * 2, (double)hb[1] / total * 2, fb);
    }
    printf("\n");
}

int main() {
    float4* garbage; unsigned long long* bad; float* fb;
    hipMalloc(&garbage, 256 * sizeof(float4)); hipMalloc(&bad, 16); hipMalloc(&fb, 4);
    unsigned h[256 * 4]; for (int i = 0; i < 1024; ++i) h[i] = 0x42804280u;
    hipMemcpy(garbage, h, sizeof(h), hipMemcpyHostToDevice);
    printf("fraction of checked accumulator values that differ from CH * 16 * b(n); d = one wrong value minus the expected one\n");
#define ROW(CH, ACT, WAIT, txt) run<CH, ACT, WAIT>(txt, garbage, bad, fb)
    ROW(1, 0, 0, "chain 1, no action, s_nop 11");
    ROW(2, 0, 0, "chain 2, no action, s_nop 11");
    ROW(6, 0, 0, "chain 6, no action, s_nop 11");
    ROW(6, 0, 2, "chain 6, no action, no wait (interlock only)");
    ROW(6, 0, 1, "chain 6, no action, 8 x s_nop 15");
    ROW(1, 1, 1, "chain 1, VALU overwrites B, long wait");
    ROW(2, 1, 1, "chain 2, VALU overwrites B, long wait");
    ROW(6, 1, 1, "chain 6, VALU overwrites B, long wait");
    ROW(1, 2, 1, "chain 1, VALU overwrites A, long wait");
    ROW(2, 2, 1, "chain 2, VALU overwrites A, long wait");
    ROW(6, 2, 1, "chain 6, VALU overwrites A, long wait");
    ROW(1, 3, 1, "chain 1, ds_read into B, long wait");
    ROW(2, 3, 1, "chain 2, ds_read into B, long wait");
    ROW(6, 3, 1, "chain 6, ds_read into B, long wait");
    ROW(6, 4, 1, "chain 6, ds_read into A, long wait");
    ROW(2, 5, 1, "chain 2, global_load into B, long wait");
    ROW(6, 5, 1, "chain 6, global_load into B, long wait");
    {
        float4* g2; hipMalloc(&g2, 512 * sizeof(float4));
        unsigned hh[512 * 4];
        for (int r = 0; r < 2; ++r) for (int t = 0; t < 256; ++t) { float b = 1.f + r + (float)((t & 31) & 7); unsigned u; memcpy(&u, &b, 4); u = (u >> 16) * 0x10001u; for (int k = 0; k < 4; ++k) hh[(r * 256 + t) * 4 + k] = u; }
        hipMemcpy(g2, hh, sizeof(hh), hipMemcpyHostToDevice);
        printf("-- operands that land straight in front of the MFMA\n");
        run_land<0>("ds_read B ; lgkmcnt(0) ; MFMA", g2, bad, fb);
        run_land<1>("2 x ds_read ; lgkmcnt(1) ; MFMA on the first", g2, bad, fb);
        run_land<2>("ds_read SrcC ; lgkmcnt(0) ; MFMA", g2, bad, fb);
        run_land<3>("global_load B ; vmcnt(0) ; MFMA", g2, bad, fb);
    }
    {
        float4* g3; hipMalloc(&g3, 1024 * sizeof(float4));
        static float hh3[4096]; for (int i = 0; i < 4096; ++i) hh3[i] = 64.f;
        hipMemcpy(g3, hh3, sizeof(hh3), hipMemcpyHostToDevice);
        printf("-- VALU instructions while global loads land in other registers (fraction of wrong sums per quarter wave)\n");
        run_landvalu<0, 0>("v_fma_f32, VALU waves only", g3);
        run_landvalu<1, 0>("v_pk_fma_f32, VALU waves only", g3);
        run_landvalu<2, 0>("v_mul_f32 literal + v_add_f32, VALU waves only", g3);
        run_landvalu<0, 2>("v_fma_f32, beside gapped MFMAs", g3);
        run_landvalu<1, 2>("v_pk_fma_f32, beside gapped MFMAs", g3);
        run_landvalu<2, 2>("v_mul_f32 literal + v_add_f32, beside gapped MFMAs", g3);
        run_landvalu<1, 1>("v_pk_fma_f32, beside back-to-back MFMAs", g3);
    }
    printf("-- packed fp32 VALU instructions beside 16-bit MFMAs (count of wrong sums per quarter wave)\n");
    run_pk<0>("v_pk_*, VALU waves only");
    run_pk<1>("v_pk_*, beside a dependent MFMA chain");
    run_pk<3>("v_pk_*, beside the x6 pattern at full rate");
    run_pk<4>("v_pk_*, beside the x6 pattern with s_nop 1 gaps");
    printf("-- the positional encoder evaluated twice (count of slots that differ between the two evaluations)\n");
    run_trig<0>("encoder, VALU waves only");
    run_trig<1>("encoder, beside back-to-back MFMAs");
    run_trig<2>("encoder, beside gapped MFMAs");
    printf("-- v_cmp -> v_cndmask through VCC (fraction of wrong selections per quarter wave)\n");
    run_vcc<3, 0>("no wait state, VALU waves only", bad);
    run_vcc<0, 0>("s_nop 0, VALU waves only", bad);
    run_vcc<1, 0>("s_nop 1, VALU waves only", bad);
    run_vcc<3, 1>("no wait state, beside back-to-back MFMAs", bad);
    run_vcc<0, 1>("s_nop 0, beside back-to-back MFMAs", bad);
    run_vcc<1, 1>("s_nop 1, beside back-to-back MFMAs", bad);
    run_vcc<2, 1>("s_nop 3, beside back-to-back MFMAs", bad);
    run_vcc<0, 2>("s_nop 0, beside gapped MFMAs", bad);
    run_vcc<1, 2>("s_nop 1, beside gapped MFMAs", bad);
    run_vcc<2, 2>("s_nop 3, beside gapped MFMAs", bad);
    printf("-- the x6 accumulate pattern (3 accumulators x 8 K steps)\n");
    run_x6pat<0>("x6 pattern, back to back", bad, fb);
    run_x6pat<2>("x6 pattern, s_nop 3 between MFMAs", bad, fb);
    run_x6pat<1>("x6 pattern, s_nop 15 between MFMAs", bad, fb);
    printf("-- chains with 16 idle cycles between the MFMAs and de-phased waves\n");
    run_contend<0, 0>("gapped chain 6, no action, s_nop 11", garbage, bad, fb);
    run_contend<0, 1>("gapped chain 6, no action, long wait", garbage, bad, fb);
    run_contend<1, 1>("gapped chain 6, VALU overwrites B, long wait", garbage, bad, fb);
    run_contend<2, 1>("gapped chain 6, VALU overwrites A, long wait", garbage, bad, fb);
    run_contend<3, 1>("gapped chain 6, ds_read into B, long wait", garbage, bad, fb);
    run_contend<4, 1>("gapped chain 6, ds_read into A, long wait", garbage, bad, fb);
    run_contend<5, 1>("gapped chain 6, global_load into B, long wait", garbage, bad, fb);
#define GROW(G) run_gap<G, false>(bad, fb); run_gap<G, true>(bad, fb)
    GROW(-1); GROW(0); GROW(1); GROW(2); GROW(3); GROW(4); GROW(5); GROW(6); GROW(7); GROW(8); GROW(9); GROW(10); GROW(11); GROW(12); GROW(13); GROW(14); GROW(15); GROW(31); GROW(47);
    return 0;
}
