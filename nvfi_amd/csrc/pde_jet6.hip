// pde_jet6.hip - round 6: the Jacobian program's FORWARD (k_pde_jet_fwd of pde_jet.hip: value + four tangent columns of weight_net per
// kept collocation point; reference: the forward half of functorch's jacrev inside NVFi.get_vel_loss, models/nvfi.py:68-79, through
// VelBasis.weight_net, models/velocity_field.py:58-63) with the hidden layers' fp32 products on the 16-BIT matrix pipe - the x6 scheme of
// vel_x6.hip (x6.h): every operand split into three bfloat16 terms, the six largest term products per K step, two fp32 accumulators by
// magnitude class.
//
// One workgroup of four waves per 32-point tile, wave w owns rows [32 w, 32 w + 32) of every layer - for ALL FIVE columns: an A operand
// (three 16-byte loads per K step from the x6 weight images in L2, two K steps ahead) feeds 5 x 6 MFMAs, so the weight stream is a fifth of the
// per-point kernel's.  The layer inputs of the five columns travel between the waves through LDS already split (120 KB: [column][K step]
// [term][lane]; one workgroup per CU, one wave per SIMD - whose own VALU work the 16-bit MFMAs leave room for, dual_pipe_probe3).  Per layer
// and wave: 240 MFMAs of 32 cycles instead of 320 of 64.  The epilogue is k_pde_jet_fwd's - the pre-activations z / zd_j go to the SAME
// stash rows in the same layout (x4 blocks for layers 0..3), SiLU and SiLU' once per row for the five columns - plus the split of the 5 x 16
// outputs; the 128 -> 6 output layer stays on the fp32 MFMA (K split over the waves, partial sums through LDS) and the acceleration net's
// value column rides in the launch's trailing workgroups, both exactly as in pde_jet.hip.
// Numerics: z / zd_j differ from the fp32 kernel's by the rounding of another summation order (exact products, fp32 accumulation): the PDE
// goldens (kept set, Jacobian rows, loss, gradients) pass unchanged; NVFI_PDE_JET_X6=0 keeps k_pde_jet_fwd.
#include <stdlib.h>
#include <stdio.h>
#include "common.h"
#include "vel.h"
#include "pde.h"
#include "engine16.h"
#include "x6.h"

#define J6_NC 5
#define J6_XCH_H8 (J6_NC * 8 * 3 * 64)                 // [column][K step][term][lane] 16-byte operands
#define J6_LDS_BYTES (J6_XCH_H8 * 16)
static_assert(J6_LDS_BYTES >= ENGINE_LDS_BYTES, "the trailing acceleration-net workgroups stage their fragments in the same LDS");
// -DJ6_TIMING: wave 0 of workgroup 0 accumulates shader-clock intervals: [0] layer 0, [1] epilogue (stash, SiLU), [2] barrier 1, [3] split + LDS writes,
// [4] barrier 2, [5] MFMA loop, [6] output stage, [7] tiles
#ifdef J6_TIMING
__device__ unsigned long long j6_times[8];
#define J6_T(slot) do { if (tm) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tacc[slot] += n_ - t0; t0 = n_; } } while (0)
#else
#define J6_T(slot) do { } while (0)
#endif
typedef const b8_t __attribute__((address_space(1)))* j6_gptr;
__device__ __forceinline__ j6_gptr j6_base(const b8_t* p) { j6_gptr q = (j6_gptr)p; asm("" : "+s"(q)); return q; }

// tangent of the PositionEncoder slots wrt q_j (pde_jet.hip: jet_encode_tangent)
__device__ __forceinline__ void j6_encode_tangent(const float* x0, int h, int j, float* xd) {
#pragma unroll
    for (int s = 0; s < 16; ++s) xd[s] = 0.f;
    if (j == 0 && h == 0) xd[0] = 1.f;
    if (j == 1 && h == 1) xd[0] = 1.f;
    if (j == 2 && h == 0) xd[1] = 1.f;
    if (j == 3 && h == 1) xd[1] = 1.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float mine = x0[2 + 4 * k + c];
            const float other = __shfl_xor(mine, 32);
            const float fr = (float)(1 << k);
            if (c == j) xd[2 + 4 * k + c] = h ? -fr * other : fr * other;
        }
}
// the fp32 output layer of pde_jet.hip (jet_mfma<4, 16>): acc[c] += A (x4 fragment, four K steps per 16-byte load) x x[c]
__device__ __forceinline__ void j6_out_mfma(const float4* __restrict__ a4, int lane, const float (&x)[J6_NC][16], f32x16* acc) {
    float4 cur = a4[lane];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 nxt = cur;
        if (g + 1 < 4) nxt = a4[(g + 1) * 64 + lane];
        const float av[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < J6_NC; ++c) acc[c] = MFMA32(av[k], x[c][4 * g + k], acc[c]);
        cur = nxt;
    }
}

__global__ __launch_bounds__(WG_THREADS, 1) void k_pde_jet6_fwd(PdeJetArgs a, const b8_t* __restrict__ img) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    b8_t* xch = reinterpret_cast<b8_t*>(lds);
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = pde_pass_count_of(a);
    if ((int)blockIdx.x >= a.jet_tiles) {
        // trailing workgroups: value column of the ReLU acceleration net for 4 tiles (pde_jet.hip)
        const int wg = blockIdx.x - a.jet_tiles;
        if (wg * WG_SAMPLES >= count) return;
        const int tile = wg * 4 + w;
        const int i = tile * TILE + (lane & 31);
        const float4 q = i < count ? a.qorig[a.klist[a.first + i]] : zero4();
        float* T = a.stash + (size_t)tile * PDE_TILE_ROWS * REGF;
        float o4[4], aw[6];
        velnet_forward<0, true>(a.Wa, lds, lds + LDS_W_FLOATS, lane, q, T + PDE_ZA * REGF, nullptr, o4);
        gather6(o4, h, aw);
        if (h == 0 && i < a.cap) {
            float* o = a.wout + (size_t)30 * a.cap + i;
#pragma unroll
            for (int k = 0; k < 6; ++k) o[(size_t)k * a.cap] = aw[k];
        }
        return;
    }
    const int tile = blockIdx.x;
    if (tile * TILE >= (count + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES) return;
    const int i = tile * TILE + (lane & 31);
    const bool active = i < count;
    const float4 q = active ? a.qorig[a.klist[a.first + i]] : zero4();
    float* T = a.stash + (size_t)tile * PDE_TILE_ROWS * REGF;
    const b8_t* W1 = img; const b8_t* W2 = img + X6_H8; const b8_t* W3 = img + 2 * X6_H8;
#ifdef J6_TIMING
    const bool tm = blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = __builtin_amdgcn_s_memtime();
#endif
    f32x16 a0[J6_NC], a1[J6_NC];
    // ---- layer 0 (28 -> 128): every wave encodes the point and its four encoder tangents itself; 2 K steps
    {
        b8_t A1[2], A2[2], A3[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int o = X6_L0 + (w * 2 + s) * 64 + lane;
            A1[s] = W1[o]; A2[s] = W2[o]; A3[s] = W3[o];
        }
        float x0[16];
        vel_encode_slots(q, h, x0);
        if (w == 0) stash_store<16>(T + PDE_X0 * REGF, lane, x0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a0[0][r] = a.bv[0][32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
            a1[0][r] = 0.f;
#pragma unroll
            for (int c = 1; c < J6_NC; ++c) { a0[c][r] = 0.f; a1[c][r] = 0.f; }
        }
#pragma unroll
        for (int c = 0; c < J6_NC; ++c) {
            float xin[16];
            if (c == 0) {
#pragma unroll
                for (int s = 0; s < 16; ++s) xin[s] = x0[s];
            } else {
                j6_encode_tangent(x0, h, c - 1, xin);
                if (w == c - 1) stash_store<16>(T + (PDE_X0D + 16 * (c - 1)) * REGF, lane, xin);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                b8_t b1, b2, b3;
                x6_split8(xin + 8 * s, b1, b2, b3);
                x6_mm6(A1[s], A2[s], A3[s], b1, b2, b3, a0[c], a1[c]);
            }
        }
    }
    float act[J6_NC][16];
#ifdef J6_TIMING
    asm volatile("s_nop 0" :: "v"(a1[4][0]));
#endif
    J6_T(0);
#pragma unroll 1
    for (int l = 0; l < 5; ++l) {
        // the next layer's first two K steps start their trip from L2 now; they land behind the epilogue and the exchange
        b8_t A1[2], A2[2], A3[2];
        j6_gptr P1 = j6_base(W1 + X6_LH(l < 4 ? l + 1 : 4) + (w * 8) * 64), P2 = j6_base(W2 + X6_LH(l < 4 ? l + 1 : 4) + (w * 8) * 64), P3 = j6_base(W3 + X6_LH(l < 4 ? l + 1 : 4) + (w * 8) * 64);
        if (l < 4) {
#pragma unroll
            for (int s = 0; s < 2; ++s) { A1[s] = P1[s * 64 + lane]; A2[s] = P2[s * 64 + lane]; A3[s] = P3[s * 64 + lane]; }
        }
        // epilogue of layer l on this wave's 16 rows: stash z / zd_j, activation and its derivative (once for the five columns)
        const int row0 = l * 64 + 16 * w;
        const bool x4 = a.x4 && l < 4;
        {
            f32x16 zc[J6_NC];
#pragma unroll
            for (int c = 0; c < J6_NC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) zc[c][r] = a1[c][r] + a0[c][r];
            if (x4) {
                stash_st16_x4(T + (size_t)(PDE_Z + row0) * REGF, lane, zc[0]);
#pragma unroll
                for (int j = 0; j < 4; ++j) stash_st16_x4(T + (size_t)(PDE_ZD + 320 * j + row0) * REGF, lane, zc[1 + j]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = zc[0][r];
                if (!x4) STASH_ST(T[(size_t)(PDE_Z + row0 + r) * REGF + lane], z);
                const float s = fast_sigmoid(z);
                const float d1 = s * (1.f + z * (1.f - s));
                act[0][r] = z * s;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float zd = zc[1 + j][r];
                    if (!x4) STASH_ST(T[(size_t)(PDE_ZD + 320 * j + row0 + r) * REGF + lane], zd);
                    act[1 + j][r] = d1 * zd;
                }
            }
        }
        J6_T(1);
        if (l == 4) break;
        __syncthreads();                                 // the previous layer's readers of the exchange images are done
        J6_T(2);
#pragma unroll
        for (int c = 0; c < J6_NC; ++c)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                b8_t b1, b2, b3;
                x6_split8(act[c] + 8 * k, b1, b2, b3);
                b8_t* dst = xch + (size_t)((c * 8 + 2 * w + k) * 3) * 64 + lane;
                dst[0] = b1; dst[64] = b2; dst[128] = b3;
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a0[0][r] = a.bv[l + 1][32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
            a1[0][r] = 0.f;
#pragma unroll
            for (int c = 1; c < J6_NC; ++c) { a0[c][r] = 0.f; a1[c][r] = 0.f; }
        }
        J6_T(3);
        __syncthreads();
        J6_T(4);
        // 8 K steps x 5 columns, software-pipelined by hand: the B operands of item (s, c) + 1 leave LDS before the six MFMAs of item (s, c) issue
        // (one wave per SIMD: nobody else hides a 100-200-cycle LDS round trip in front of every item), pinned with sched_barrier
        {
            b8_t Bq[2][3];
            { const b8_t* src = xch + lane; Bq[0][0] = src[0]; Bq[0][1] = src[64]; Bq[0][2] = src[128]; }
#pragma unroll
            for (int it = 0; it < 8 * J6_NC; ++it) {
                const int s = it / J6_NC, c = it % J6_NC;
                if (it + 1 < 8 * J6_NC) {
                    const int s2 = (it + 1) / J6_NC, c2 = (it + 1) % J6_NC;
                    const b8_t* src = xch + (size_t)((c2 * 8 + s2) * 3) * 64 + lane;
                    Bq[(it + 1) & 1][0] = src[0]; Bq[(it + 1) & 1][1] = src[64]; Bq[(it + 1) & 1][2] = src[128];
                }
                x6_mm6(A1[s & 1], A2[s & 1], A3[s & 1], Bq[it & 1][0], Bq[it & 1][1], Bq[it & 1][2], a0[c], a1[c]);
                if (c == J6_NC - 1 && s + 2 < 8) { A1[s & 1] = P1[(s + 2) * 64 + lane]; A2[s & 1] = P2[(s + 2) * 64 + lane]; A3[s & 1] = P3[(s + 2) * 64 + lane]; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef J6_TIMING
        asm volatile("s_nop 0" :: "v"(a1[4][0]));
#endif
        J6_T(5);
    }
    // ---- output layer 128 -> 6 on the fp32 MFMA, split along K: this wave contracts its own 32 rows, partial sums meet in LDS (pde_jet.hip)
    {
        f32x16 acc[J6_NC];
#pragma unroll
        for (int c = 0; c < J6_NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        j6_out_mfma(a.f4[5] + (size_t)(4 * w) * 64, lane, act, acc);
        __syncthreads();      // the exchange images' last readers are done
        float* red = lds;     // [wave][column][4 regs][64 lanes]
#pragma unroll
        for (int c = 0; c < J6_NC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((w * J6_NC + c) * 4 + r) * 64 + lane] = acc[c][r];
        __syncthreads();
        if (w == 0 && h == 0 && i < a.cap) {
#pragma unroll
            for (int c = 0; c < J6_NC; ++c) {
                float o6[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int ln = lane + 32 * (k >> 2), r = k & 3;
                    float s = c == 0 ? a.bv[5][k] : 0.f;
                    s = s + red[((0 * J6_NC + c) * 4 + r) * 64 + ln];
                    s = s + red[((1 * J6_NC + c) * 4 + r) * 64 + ln];
                    s = s + red[((2 * J6_NC + c) * 4 + r) * 64 + ln];
                    s = s + red[((3 * J6_NC + c) * 4 + r) * 64 + ln];
                    o6[k] = s;
                }
                float* o = a.wout + (size_t)(c == 0 ? 0 : 6 * c) * a.cap + i;
#pragma unroll
                for (int k = 0; k < 6; ++k) o[(size_t)k * a.cap] = o6[k];
            }
        }
    }
    J6_T(6);
#ifdef J6_TIMING
    if (tm) { for (int k = 0; k < 7; ++k) j6_times[k] = tacc[k]; j6_times[7] = 1; }
#endif
}

int launch_pde_jet6_fwd(const PdeJetArgs& a0, const void* x6img, unsigned tiles, unsigned anet_wgs, hipStream_t st) {
    static DeviceOnce once;
    if (once.run([] { HIPCK(hipFuncSetAttribute((const void*)k_pde_jet6_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, J6_LDS_BYTES)); return 0; })) return 1;
    PdeJetArgs a = a0; a.jet_tiles = (int)tiles;
    hipLaunchKernelGGL(k_pde_jet6_fwd, dim3(tiles + anet_wgs), dim3(WG_THREADS), J6_LDS_BYTES, st, a, reinterpret_cast<const b8_t*>(x6img));
    LAUNCHCK();
#ifdef J6_TIMING
    static int shots = 0;
    if (++shots % 8 == 0 && shots <= 32) {
        unsigned long long h[8];
        HIPCK(hipStreamSynchronize(st));
        HIPCK(hipMemcpyFromSymbol(h, HIP_SYMBOL(j6_times), sizeof(h)));
        fprintf(stderr, "[jet6 timing] one tile: layer0 %llu | epilogue %llu | bar1 %llu | split+LDS %llu | bar2 %llu | MFMA loop %llu | output %llu\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    }
#endif
    return 0;
}
