// app_fuse.hip - adjoint of the render MLP (MLPRender_PE) WITH its two 128-wide weight gradients formed in the same kernel
// (reference: autograd of models/tensorf_base.py:88-98 through models/tensorf_keyframe.py:738-747).
//
// k_app_bwd (render.hip) is a one-wave-per-tile kernel that stages every transposed layer in LDS between two barriers and writes the layer
// gradients gz2 / gz1 to a stash (128 of the 160 adjoint rows per 32-sample tile) that k_wgrad_ring8 reads back together with the forward's
// h1 / x_in rows.  Here ONE persistent workgroup of twelve waves per CU does both, in the role split of vel_fuse.hip / pde_fuse.hip:
//
//   * waves 0-3 ("adjoint" waves): wave w owns rows [32w, 32w+32) of every layer (x4 transposed fragments straight from L2).  Three phases
//     per tile: 3 -> 128 from the colour seeds (ReLU signs of layer 2 from the forward's bit masks), 128 -> 128 (signs of layer 1 from the
//     h1 rows it loads anyway), 128 -> input slots.  The layer gradients and the layer inputs (h1, x_in: this wave's 16 stash rows) go to
//     LDS in the exchange layout; the epilogue of the last phase is k_app_bwd's: wave 0 stores the feature gradient (edge job of basis_mat),
//     pushes it through basis_mat^T (32 MFMAs from its own registers) and writes the 48 channel gradients of the plane scatter; waves 1 and 2
//     hold the positional-encoding slots of the sample position and form the coordinate gradient (one exchange through LDS);
//   * waves 4-11 ("contraction" waves) hold the 2 x 16 output tiles of the two 128 x 128 gradients (64 accumulator registers) and contract
//     (gz2, h1) and (gz1, x_in) one phase behind;
//   * three barriers per tile, four 16.5 KB images; one slab per layer and workgroup in k_wgrad_ring8's format: k_wgrad_reduce is unchanged.
// Still through the stash and k_wgrad_ring8: the two edge products (3 x 128: the seed rows; basis_mat: the feature gradient rows).
// Not covered (k_app_bwd keeps them): SH shading, the in-kernel plane tail (no sorted-tile scatter), NVFI_DETERMINISTIC.
//
// Numerics: every layer gradient is the number k_app_bwd forms (same operands, same K order); the z component of the coordinate gradient
// adds its two partial sums in another order, and a weight gradient is summed over samples in another order than k_wgrad_ring8's.
#include <stdlib.h>
#include <stdio.h>
#include "common.h"
#include "vel.h"
#include "pde.h"
#include "render.h"
#include "fuse.h"

#define AF_THREADS 768
#define AF_X0 0
#define AF_X1 1
#define AF_Y0 2
#define AF_Y1 3
#define AF_IMAGES 4
#define AF_LDS_BYTES (AF_IMAGES * FUSE_XB * 16 + 64 * 4)       // + one float per lane: wave 2's share of the coordinate gradient

__device__ __forceinline__ gcfp af_base(const float* p) { gcfp q = (gcfp)p; asm volatile("" : "+s"(q)); return q; }
__device__ __forceinline__ gfp af_base(float* p) { gfp q = (gfp)p; asm volatile("" : "+s"(q)); return q; }

struct AfA { float4* S; int w, lane, pos; };

__device__ __forceinline__ float4* af_rows(const AfA& A, int img) { return A.S + img * FUSE_XB + (4 * A.w) * 2 * FUSE_HR + A.pos; }

// dgrad of the tile: acc = T^T fragment (registers) x layer gradient image (LDS)
__device__ __forceinline__ void af_dgrad(const AfA& A, int img, const f32x4v (&wq)[16], f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float4* Xr = A.S + img * FUSE_XB + A.pos;
    float4 b = Xr[0], bn;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (g + 1 < 16) bn = Xr[(g + 1) * 2 * FUSE_HR];
        const float a4[4] = {wq[g].x, wq[g].y, wq[g].z, wq[g].w};
        acc = MFMA32(a4[0], b.x, acc); acc = MFMA32(a4[1], b.y, acc); acc = MFMA32(a4[2], b.z, acc); acc = MFMA32(a4[3], b.w, acc);
        b = bn;
    }
}
__device__ __forceinline__ void af_put(const AfA& A, int img, const float (&v)[16]) {
    float4* Xw = af_rows(A, img);
#pragma unroll
    for (int k = 0; k < 4; ++k) Xw[k * 2 * FUSE_HR] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
}
__device__ __forceinline__ void af_rows16(const float* base16, int lane, float (&v)[16]) {
    gcfp zp = af_base(base16);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = STASH_LD(zp[r * REGF + lane]);
}

__device__ __forceinline__ void af_role_adjoint(const AppFuseArgs& F, float* lds, int w, int lane, int ntiles, int count) {
    const AppArgs& a = F.a;
    AfA A; A.S = reinterpret_cast<float4*>(lds); A.w = w; A.lane = lane;
    const int h = lane >> 5, j = lane & 31;
    A.pos = h * FUSE_HR + j;
    float* exch = lds + AF_IMAGES * FUSE_XB * 4;
    const int G = gridDim.x;
    f32x4v wq[16];
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        const int i = tile * TILE + j;
        const bool active = i < count;
        const int n = active ? a.list[i] : 0;
        const float* stf = a.stash_f + (size_t)tile * (APP_F_ROWS * REGF);
        float* stb = a.stash_b + (size_t)tile * (APP_B_ROWS * REGF);
        // per-tile opaque copies of the launch-invariant fragment bases (pde_fuse.hip: left to loop-invariant code motion they overflow the SGPR file)
        const float4 *t3, *t2, *t1, *tb;
        { gcfp p = af_base(reinterpret_cast<const float*>(F.t3)); t3 = reinterpret_cast<const float4*>((const float*)p); }
        { gcfp p = af_base(reinterpret_cast<const float*>(F.t2)); t2 = reinterpret_cast<const float4*>((const float*)p); }
        { gcfp p = af_base(reinterpret_cast<const float*>(F.t1)); t1 = reinterpret_cast<const float4*>((const float*)p); }
        { gcfp p = af_base(reinterpret_cast<const float*>(F.tb)); tb = reinterpret_cast<const float4*>((const float*)p); }
        float hv[16];
        // ---- phase 0: colour seeds -> 128 (T3), ReLU signs of layer 2 from the forward's bit masks
        {
            // seeds: go_c = w * gr_c * c (1 - c) in rows 0..2 of a D tile (lanes of half 0, registers 0..2), as k_app_bwd forms them
            float go[3] = {0.f, 0.f, 0.f};
            if (active && h == 0 && a.g_rgb) {
                const int r = n / a.S;
                const float4 pre = a.rgb_pre[r];
                const float pv[3] = {pre.x, pre.y, pre.z};
                const float4 c = a.rgbs[i];
                const float cv[3] = {c.x, c.y, c.z};
                const float wgt = a.weight[n];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float gr = (pv[k] >= 0.f && pv[k] <= 1.f) ? a.g_rgb[3 * (size_t)r + k] : 0.f;
                    go[k] = wgt * gr * cv[k] * (1.f - cv[k]);
                }
            }
            if (w == 0) {        // the A operand of the output layer's weight gradient (edge job)
                gfp gr = af_base(stb);
#pragma unroll
                for (int s = 0; s < 16; ++s) STASH_ST(gr[s * REGF + lane], s < 3 ? go[s] : 0.f);
            }
            af_rows16(stf + (size_t)(96 + 16 * w) * REGF, lane, hv);
            const unsigned* mk = a.relu_mask + (size_t)tile * 256 + 128;
            const unsigned mw = w < 2 ? mk[lane] : mk[64 + lane];
            f32x4v w3;
            {
                gcf4p b3 = (gcf4p)(t3 + (size_t)w * 64);
                asm("" : "+s"(b3));
                w3 = b3[lane];
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            acc = MFMA32(w3.x, go[0], acc); acc = MFMA32(w3.y, go[1], acc); acc = MFMA32(w3.z, go[2], acc); acc = MFMA32(w3.w, 0.f, acc);
            float gz[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) gz[r] = ((mw >> ((16 * w + r) & 31)) & 1u) ? acc[r] : 0.f;
            af_put(A, AF_X0, gz);
            af_put(A, AF_Y0, hv);
            __builtin_amdgcn_sched_barrier(0);
            split_load16(t2 + (size_t)w * 16 * 64, lane, wq);
            FUSE_BAR();
        }
        // ---- phase 1: 128 -> 128 (T2), ReLU signs of layer 1 from h1 itself
        float xv[16];
        {
            af_rows16(stf + (size_t)(32 + 16 * w) * REGF, lane, xv);
            f32x16 acc;
            af_dgrad(A, AF_X0, wq, acc);
            asm volatile("" :: "v"(acc[0]));
            split_load16(t1 + (size_t)w * 16 * 64, lane, wq);
            float gz[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) gz[r] = hv[r] > 0.f ? acc[r] : 0.f;
            af_put(A, AF_X1, gz);
            af_put(A, AF_Y1, xv);
            FUSE_BAR();
        }
        // ---- phase 2: 128 -> input slots (T1): this wave's slots 16 w + r (RENDER_IN layout), then k_app_bwd's epilogue
        {
            f32x16 acc;
            af_dgrad(A, AF_X1, wq, acc);
            if (w == 0) {
                // feature gradient (32 rows of the basis tile): edge job of basis_mat, then basis^T -> 48 channel gradients in gather layout
                gfp gr = af_base(stb + (size_t)144 * REGF);
#pragma unroll
                for (int r = 0; r < 16; ++r) STASH_ST(gr[r * REGF + lane], acc[r]);
                f32x16 gg0, gg1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { gg0[r] = 0.f; gg1[r] = 0.f; }
                gcf4p bb = (gcf4p)tb;
                asm("" : "+s"(bb));
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const f32x4v a0 = bb[(0 * 4 + s4) * 64 + lane], a1 = bb[(1 * 4 + s4) * 64 + lane];
                    const float a0v[4] = {a0.x, a0.y, a0.z, a0.w}, a1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        gg0 = MFMA32(a0v[k], acc[4 * s4 + k], gg0);
                        gg1 = MFMA32(a1v[k], acc[4 * s4 + k], gg1);
                    }
                }
                if (active && a.gg) {
#pragma unroll
                    for (int a6 = 0; a6 < 6; ++a6) {
                        const int s0 = 4 * a6;
                        const float4 v = s0 < 16 ? make_float4(gg0[s0 & 15], gg0[(s0 & 15) + 1], gg0[(s0 & 15) + 2], gg0[(s0 & 15) + 3])
                                                 : make_float4(gg1[s0 & 15], gg1[(s0 & 15) + 1], gg1[(s0 & 15) + 2], gg1[(s0 & 15) + 3]);
                        *reinterpret_cast<float4*>(a.gg + (size_t)i * 48 + 4 * (2 * a6 + h)) = v;
                    }
                }
            }
            // coordinate gradient through the raw position (slots 16..18, half 1) and its positional encodings (slots 19..36: sin | cos by half):
            // wave 1 holds slots 16..31, wave 2 slots 32..47 (only 32..36 belong to the position)
            float gp[3] = {0.f, 0.f, 0.f};
            if (w == 1 || w == 2) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float s = (w == 1 && h) ? acc[c] : 0.f;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int sl = 19 + c * 6 + k;
                        if ((sl >> 4) == 1 || (sl >> 4) == 2) {
                            const bool mine_w = (sl >> 4) == w;
                            const float mine = xv[sl & 15];
                            const float other = __shfl_xor(mine, 32);
                            const float fr = (float)(1 << k);
                            const float t = (h ? -fr * other : fr * other) * acc[sl & 15];
                            s += mine_w ? t : 0.f;
                        }
                    }
                    gp[c] = s;
                }
                if (w == 2) exch[lane] = gp[2];
            }
            FUSE_BAR();
            // (no barrier behind this: wave 2 writes its next share two barriers from now, and nothing else of the tile is read here)
            if (w == 1) {
                gp[2] += exch[lane];
#pragma unroll
                for (int c = 0; c < 3; ++c) gp[c] += __shfl_xor(gp[c], 32);
                if (active && h == 0) a.gxw[n] = make_float4(gp[0], gp[1], gp[2], 0.f);
            }
        }
    }
}

// ---------------------------------------------------------------- contraction waves
#define AF_BAR_G() do { __builtin_amdgcn_sched_barrier(0);                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(G0a), "+v"(G0b), "+v"(G1a), "+v"(G1b) :: "memory");   \
        __builtin_amdgcn_sched_barrier(0); } while (0)
#define AF_XBF (FUSE_XB * 4)
#define AF_CONTRACT(L, XIMG, YIMG)                                                                                   \
    do {                                                                                                             \
        const float* xa_ = Sf + (XIMG) * AF_XBF + ob * FUSE_TF + o; const float* yb_ = Sf + (YIMG) * AF_XBF + ib0 * FUSE_TF + o; \
        _Pragma("unroll") for (int st = 0; st < 16; ++st) {                                                          \
            const float av_ = xa_[8 * st], b0_ = yb_[8 * st], b1_ = yb_[8 * st + FUSE_TF];                            \
            G##L##a = MFMA32(av_, b0_, G##L##a); G##L##b = MFMA32(av_, b1_, G##L##b);                               \
            asm("v_add_f32 %0, %0, %1" : "+v"(bs##L) : "v"(av_));                                                    \
        }                                                                                                            \
    } while (0)
#define AF_FLUSH(SLABS, L)                                                                                           \
    do {                                                                                                             \
        float* Sl = (SLABS) + (size_t)blockIdx.x * F.slab_floats;                                                    \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
            gfp Sq = af_base(Sl + (size_t)(32 * ob + 8 * q) * 128 + 32 * ib0);                                       \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                          \
                Sq[c * 128 + flo] = G##L##a[4 * q + c];                                                              \
                Sq[c * 128 + 32 + flo] = G##L##b[4 * q + c];                                                         \
            }                                                                                                        \
        }                                                                                                            \
        if ((v & 1) == 0) {                                                                                          \
            float bsum = bs##L; bsum += __shfl_xor(bsum, 32);                                                        \
            if (kk == 0) Sl[(size_t)128 * 128 + 32 * ob + i] = bsum;                                                 \
        }                                                                                                            \
    } while (0)

__device__ __forceinline__ void af_role_contract(const AppFuseArgs& F, const float* Sf, int v, int lane, int ntiles) {
    const int i = lane & 31, kk = lane >> 5;
    const int ob = v >> 1, ib0 = 2 * (v & 1);
    const int o = ((i >> 3) * 2 + (i & 1)) * (FUSE_HR * 4) + ((i >> 1) & 3) + 4 * kk;
    const int flo = 4 * kk * 128 + i;
    f32x16 G0a, G0b, G1a, G1b;          // 0: layer 1 (gz2 x h1), 1: layer 0 (gz1 x x_in)
    float bs0 = 0.f, bs1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { G0a[r] = 0.f; G0b[r] = 0.f; G1a[r] = 0.f; G1b[r] = 0.f; }
    const int G = gridDim.x;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        AF_BAR_G();                                       // phase 0
        AF_CONTRACT(0, AF_X0, AF_Y0); AF_BAR_G();         // phase 1
        AF_CONTRACT(1, AF_X1, AF_Y1); AF_BAR_G();         // phase 2 (its barrier sits in front of the coordinate gradient's exchange)
    }
    AF_FLUSH(F.slabs_1, 0); AF_FLUSH(F.slabs_0, 1);
}

__global__ __launch_bounds__(AF_THREADS) void k_app_fuse_bwd(AppFuseArgs F) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = __builtin_amdgcn_readfirstlane(*F.a.count);
    const int ntiles = (count + WG_SAMPLES - 1) / WG_SAMPLES * (WG_SAMPLES / TILE);      // whole workgroups of the forward
    if (wave < 4) {
        __builtin_amdgcn_s_setprio(3);
        af_role_adjoint(F, lds, wave, lane, ntiles, count);
    } else {
        af_role_contract(F, lds, wave - 4, lane, ntiles);
    }
}

// x4 copies of the transposed render-MLP fragments into buf (APP_X4_FLOATS); fills F.t3 / t2 / t1 / tb
int pack_app_x4(const RenderFrags& W, float* buf, AppFuseArgs* F, hipStream_t st) {
    X4Jobs xj; xj.n = 0;
    float* p = buf;
    auto add = [&](const float* src, int MT, int NS, const float4** slot) {
        xj.src[xj.n] = src; xj.dst[xj.n] = p; xj.MT[xj.n] = MT; xj.NS[xj.n] = NS; ++xj.n;
        *slot = reinterpret_cast<const float4*>(p);
        p += X4_FLOATS(MT, NS);
    };
    add(W.t3, 4, 4, &F->t3); add(W.t2, 4, 64, &F->t2); add(W.t1, 4, 64, &F->t1); add(W.tb, 2, 16, &F->tb);
    return launch_frag_x4(xj, st);
}

int launch_app_fuse_bwd(const AppFuseArgs& F, int64_t cap_samples, int max_slabs, int* nslab_out, hipStream_t st) {
    *nslab_out = 0;
    const int64_t tiles = (cap_samples + TILE - 1) / TILE;
    if (tiles <= 0) return 0;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t prop;
        ncu = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        HIPCK(hipFuncSetAttribute((const void*)k_app_fuse_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, AF_LDS_BYTES));
    }
    int G = ncu < max_slabs ? ncu : max_slabs;
    if ((int64_t)G > tiles) G = (int)tiles;
    hipLaunchKernelGGL(k_app_fuse_bwd, dim3((unsigned)G), dim3(AF_THREADS), AF_LDS_BYTES, st, F);
    LAUNCHCK();
    *nslab_out = G;
    return 0;
}
