// pde_jet.hip - the PDE regulariser's Jacobian program with ALL FIVE columns of a point tile in one workgroup.
//
// Reference semantics: the functorch vmap(jacrev) Jacobian of the un-gated vel_net and its second-order backward inside
// NVFi.get_vel_loss (models/nvfi.py:68-84).  pde.hip computes it in forward mode: one value column and four tangent columns
// (d/dx, d/dy, d/dz, d/dt) pushed through the 6-layer SiLU net, then the reverse of that program.  There, every column of a tile ran
// in its own workgroup, so the value column's pre-activations travelled through memory to the four tangent workgroups (and the
// sigmoid derivatives were re-formed four times), and the second-derivative corrections act''(z) * zd_j * ghd_j of the four tangent
// adjoints travelled through memory to the value adjoint.  Here a workgroup owns ONE tile of 32 points and all five columns:
//
//   * FEATURE SPLIT: wave w of the workgroup owns output rows [32w, 32w+32) of every layer - one 32x32 MFMA tile - for all five
//     columns.  The weight operand of an MFMA step is shared by the five columns (one 16-byte global load feeds 20 MFMAs), and each
//     wave only ever touches its own quarter of a layer's fragment, so the weights are not staged through LDS at all.
//   * the layer's input (128 features x 5 columns) must be whole in every wave: the four waves swap their 32-row slices through an
//     80 KB LDS buffer between layers (two barriers per layer; the only synchronisation in the kernel).
//   * act'(z), act''(z) are formed once per element; the correction term never leaves registers; the stash carries only what the
//     weight-gradient kernel needs (z, zd_j, gz, gzd_j): 4016 rows per tile instead of 5296.
//   * the 128 -> 6 output layer is split over the waves along K (each wave contracts its own 32 rows) and summed through LDS.
//
// The arithmetic per element is the one of pde.hip (same formulas, same MFMA accumulation order), so both paths agree to rounding of
// the split-K output layer; pde.hip's column kernels remain for the ReLU acceleration net and as NVFI_PDE_JET=0.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "pde.h"

// ---------------------------------------------------------------- x4 fragments: four consecutive K steps of a lane side by side
// dst[((m*NS4 + s4)*64 + lane)*4 + k] = src[(m*NS + 4*s4 + k)*64 + lane]  (zero beyond NS)
__global__ void k_frag_x4(X4Jobs jobs) {
    const int j = blockIdx.x;
    const int MT = jobs.MT[j], NS = jobs.NS[j], NS4 = (NS + 3) >> 2;
    const int total = MT * NS4 * 256;
    const float* __restrict__ src = jobs.src[j];
    float* __restrict__ dst = jobs.dst[j];
    for (int idx = blockIdx.y * blockDim.x + threadIdx.x; idx < total; idx += gridDim.y * blockDim.x) {
        const int k = idx & 3, lane = (idx >> 2) & 63, ms = idx >> 8;
        const int s4 = ms % NS4, m = ms / NS4, s = 4 * s4 + k;
        dst[idx] = s < NS ? src[(m * NS + s) * 64 + lane] : 0.f;
    }
}
int launch_frag_x4(const X4Jobs& jobs, hipStream_t st) {
    if (jobs.n == 0) return 0;
    hipLaunchKernelGGL(k_frag_x4, dim3(jobs.n, 16), dim3(256), 0, st, jobs);
    LAUNCHCK();
    return 0;
}

int pack_vel_x4_fwd(const VelFrags& W, float* buf, const float4** f4, hipStream_t st) {
    X4Jobs xj; xj.n = 0;
    float* p = buf;
    auto add = [&](const float* src, int MT, int NS, const float4** slot) {
        xj.src[xj.n] = src; xj.dst[xj.n] = p; xj.MT[xj.n] = MT; xj.NS[xj.n] = NS; ++xj.n;
        *slot = reinterpret_cast<const float4*>(p);
        p += X4_FLOATS(MT, NS);
    };
    add(W.f[0], 4, 14, &f4[0]);
    for (int l = 1; l <= 4; ++l) add(W.f[l], 4, 64, &f4[l]);
    add(W.f[5], 1, 64, &f4[5]);
    return launch_frag_x4(xj, st);
}

int pack_vel_x4_bwd(const VelFrags& W, float* buf, const float4** t4, hipStream_t st) {
    X4Jobs xj; xj.n = 0;
    float* p = buf;
    auto add = [&](const float* src, int MT, int NS, const float4** slot) {
        xj.src[xj.n] = src; xj.dst[xj.n] = p; xj.MT[xj.n] = MT; xj.NS[xj.n] = NS; ++xj.n;
        *slot = reinterpret_cast<const float4*>(p);
        p += X4_FLOATS(MT, NS);
    };
    add(W.t[0], 1, 64, &t4[0]);
    for (int l = 1; l <= 4; ++l) add(W.t[l], 4, 64, &t4[l]);
    add(W.t[5], 4, 4, &t4[5]);
    return launch_frag_x4(xj, st);
}

#define JET_NC 5
#define JET_XCH_FLOATS (JET_NC * 64 * 64)        // 80 KB: [column][s/4][lane][4]
#define JET_LDS_BYTES (JET_XCH_FLOATS * 4)

// acc[c] += sum over NS4 groups of 4 K-steps:  A = this wave's x4 fragment (one 16-byte load per group, next group in flight),
// B = x[c][s_base + ...] for the five columns
template <int NS4, int XS>
__device__ __forceinline__ void jet_mfma(const float4* __restrict__ a4, int lane, const float (&x)[JET_NC][XS], int s_base, f32x16* acc) {
    float4 cur = a4[lane];
#pragma unroll
    for (int g = 0; g < NS4; ++g) {
        float4 nxt = cur;
        if (g + 1 < NS4) nxt = a4[(g + 1) * 64 + lane];
        const float av[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < JET_NC; ++c) acc[c] = MFMA32(av[k], x[c][s_base + 4 * g + k], acc[c]);
        cur = nxt;
    }
}

// every wave publishes its 16 registers per column, then reads the whole 64-register layer input of all five columns
__device__ __forceinline__ void jet_exchange(float4* xch, int w, int lane, const f32x16* out, float (&x)[JET_NC][64]) {
    __syncthreads();          // the previous layer's readers are done (they read right after the barrier below, a whole layer ago)
#pragma unroll
    for (int c = 0; c < JET_NC; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            xch[(c * 16 + 4 * w + q) * 64 + lane] = make_float4(out[c][4 * q], out[c][4 * q + 1], out[c][4 * q + 2], out[c][4 * q + 3]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < JET_NC; ++c)
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const float4 v = xch[(c * 16 + s4) * 64 + lane];
            x[c][4 * s4] = v.x; x[c][4 * s4 + 1] = v.y; x[c][4 * s4 + 2] = v.z; x[c][4 * s4 + 3] = v.w;
        }
}

// tangent of the PositionEncoder slots wrt q_j (same as pde.hip: encode_tangent)
__device__ __forceinline__ void jet_encode_tangent(const float* x0, int h, int j, float* xd) {
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

// ---------------------------------------------------------------- forward: value + 4 tangents of weight_net
__global__ __launch_bounds__(WG_THREADS, 2) void k_pde_jet_fwd(PdeJetArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* xch = reinterpret_cast<float4*>(lds);
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = pde_pass_count_of(a);
    if ((int)blockIdx.x >= a.jet_tiles) {
        // trailing workgroups: value column of the ReLU acceleration net for 4 tiles (the column kernel of pde.hip, riding in this
        // launch so that its one round of workgroups fills the tail of the jet tiles instead of a launch of its own)
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
    // whole 128-point groups: the weight-gradient kernel walks the tiles of the last, ragged group too (zero seeds, finite z)
    if (tile * TILE >= (count + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES) return;
    const int i = tile * TILE + (lane & 31);
    const bool active = i < count;
    const float4 q = active ? a.qorig[a.klist[a.first + i]] : zero4();
    float* T = a.stash + (size_t)tile * PDE_TILE_ROWS * REGF;
    float x[JET_NC][64];
    f32x16 acc[JET_NC];
    {
        float in0[JET_NC][16];
        vel_encode_slots(q, h, in0[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) jet_encode_tangent(in0[0], h, j, in0[1 + j]);
        // encoder slots / tangents for the first layer's weight gradient: wave w stores tangent w, wave 0 also the value slots
        if (w == 0) stash_store<16>(T + PDE_X0 * REGF, lane, in0[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (w == j) stash_store<16>(T + (PDE_X0D + 16 * j) * REGF, lane, in0[1 + j]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[0][r] = a.bv[0][32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
            for (int c = 1; c < JET_NC; ++c) acc[c][r] = 0.f;
        }
        jet_mfma<4, 16>(a.f4[0] + (size_t)w * 4 * 64, lane, in0, 0, acc);
    }
#pragma unroll 1
    for (int l = 0; l < 5; ++l) {
        // epilogue of layer l on this wave's 16 rows: stash z / zd_j, activation and its derivative (once for the five columns)
        const int row0 = l * 64 + 16 * w;
        const bool x4 = a.x4 && l < 4;            // (layer 4 stays row-major: the output layer's weight gradient reads it through the ring kernel)
        if (x4) {
            stash_st16_x4(T + (size_t)(PDE_Z + row0) * REGF, lane, acc[0]);
#pragma unroll
            for (int j = 0; j < 4; ++j) stash_st16_x4(T + (size_t)(PDE_ZD + 320 * j + row0) * REGF, lane, acc[1 + j]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float z = acc[0][r];
            if (!x4) STASH_ST(T[(size_t)(PDE_Z + row0 + r) * REGF + lane], z);
            const float s = fast_sigmoid(z);
            const float d1 = s * (1.f + z * (1.f - s));
            acc[0][r] = z * s;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float zd = acc[1 + j][r];
                if (!x4) STASH_ST(T[(size_t)(PDE_ZD + 320 * j + row0 + r) * REGF + lane], zd);
                acc[1 + j][r] = d1 * zd;
            }
        }
        if (l == 4) break;
        jet_exchange(xch, w, lane, acc, x);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[0][r] = a.bv[l + 1][32 * w + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
            for (int c = 1; c < JET_NC; ++c) acc[c][r] = 0.f;
        }
        jet_mfma<16, 64>(a.f4[l + 1] + (size_t)w * 16 * 64, lane, x, 0, acc);
    }
    // output layer 128 -> 6, split along K: this wave contracts its own 32 rows (registers of acc), partial sums meet in LDS
    {
        float o[JET_NC][16];
#pragma unroll
        for (int c = 0; c < JET_NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[c][r] = acc[c][r];
#pragma unroll
        for (int c = 0; c < JET_NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        jet_mfma<4, 16>(a.f4[5] + (size_t)(4 * w) * 64, lane, o, 0, acc);
        __syncthreads();      // the exchange buffer's last readers are done
        float* red = lds;     // [wave][column][4 regs][64 lanes]
#pragma unroll
        for (int c = 0; c < JET_NC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((w * JET_NC + c) * 4 + r) * 64 + lane] = acc[c][r];
        __syncthreads();
        if (w == 0 && h == 0 && i < a.cap) {
#pragma unroll
            for (int c = 0; c < JET_NC; ++c) {
                float o6[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    // output k lives in register k&3 of half k>>2 (D layout rows 0..3 | 4..7): lanes l (h=0) and l+32 (h=1)
                    const int ln = lane + 32 * (k >> 2), r = k & 3;
                    float s = c == 0 ? a.bv[5][k] : 0.f;
                    s = s + red[((0 * JET_NC + c) * 4 + r) * 64 + ln];
                    s = s + red[((1 * JET_NC + c) * 4 + r) * 64 + ln];
                    s = s + red[((2 * JET_NC + c) * 4 + r) * 64 + ln];
                    s = s + red[((3 * JET_NC + c) * 4 + r) * 64 + ln];
                    o6[k] = s;
                }
                float* o = a.wout + (size_t)(c == 0 ? 0 : 6 * c) * a.cap + i;
#pragma unroll
                for (int k = 0; k < 6; ++k) o[(size_t)k * a.cap] = o6[k];
            }
        }
    }
}

// ---------------------------------------------------------------- backward: 4 tangent adjoints + value adjoint with the corrections
__global__ __launch_bounds__(WG_THREADS, 2) void k_pde_jet_bwd(PdeJetArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float4* xch = reinterpret_cast<float4*>(lds);
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int count = pde_pass_count_of(a);
    if ((int)blockIdx.x >= a.jet_tiles) {
        // trailing workgroups: adjoint of the acceleration net's value column (4 tiles per workgroup)
        const int wg = blockIdx.x - a.jet_tiles;
        if (wg * WG_SAMPLES >= count) return;
        const int tile = wg * 4 + w;
        const int i = tile * TILE + (lane & 31);
        float* T = a.stash + (size_t)tile * PDE_TILE_ROWS * REGF;
        float r4[4], s6[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) s6[k] = i < (int)a.cap ? a.seeds[(size_t)(30 + k) * a.cap + i] : 0.f;
        scatter6(s6, h, r4);
        velnet_value_backward<0, false>(a.Wa, lds, lds + LDS_W_FLOATS, lane, r4, T + PDE_ZA * REGF, nullptr, T + PDE_GAA * REGF);
        return;
    }
    const int tile = blockIdx.x;
    if (tile * TILE >= (count + WG_SAMPLES - 1) / WG_SAMPLES * WG_SAMPLES) return;
    const int i = tile * TILE + (lane & 31);
    float* T = a.stash + (size_t)tile * PDE_TILE_ROWS * REGF;
    const size_t cs = a.cap;
    const bool ok = i < (int)a.cap;
    float g[JET_NC][64];
    f32x16 acc[JET_NC];
    {
        // adjoint seeds of the 6 outputs, per column, in D-layout registers 0..3 (k_pde_seeds)
        float sd[JET_NC][4];
#pragma unroll
        for (int c = 0; c < JET_NC; ++c) {
            const int sbase = c == 0 ? 0 : 6 * c;
            float s6[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) s6[k] = ok ? a.seeds[(size_t)(sbase + k) * cs + i] : 0.f;
            scatter6(s6, h, sd[c]);
        }
        // rows gw of each column's adjoint image (A operand of the output layer's weight gradient): wave w stores column w (+ 4)
#pragma unroll
        for (int c = 0; c < JET_NC; ++c)
            if (w == (c & 3)) {
                float* gw_rows = T + (size_t)(PDE_GA + 336 * c + 320) * REGF;
#pragma unroll
                for (int s = 0; s < 16; ++s) gw_rows[s * REGF + lane] = s < 4 ? sd[c][s] : 0.f;
            }
#pragma unroll
        for (int c = 0; c < JET_NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        // T5: 6 -> 128, one group of 4 K-steps
        float in5[JET_NC][4];
#pragma unroll
        for (int c = 0; c < JET_NC; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) in5[c][k] = sd[c][k];
        jet_mfma<1, 4>(a.t4[5] + (size_t)w * 64, lane, in5, 0, acc);
    }
#pragma unroll 1
    for (int l = 4; l >= 0; --l) {
        // epilogue at layer l on this wave's 16 rows: acc[c] = adjoint of x_l (value) / xd_l^j (tangents)
        const int row0 = l * 64 + 16 * w;
        float zr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) zr[r] = STASH_LD(T[(size_t)(PDE_Z + row0 + r) * REGF + lane]);
        float zd[4][16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) zd[j][r] = STASH_LD(T[(size_t)(PDE_ZD + 320 * j + row0 + r) * REGF + lane]);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            f32x2 d1, d2;
            act_d12_2<1>((f32x2){zr[r], zr[r + 1]}, d1, d2);
            f32x2 cj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 a2 = {acc[1 + j][r], acc[1 + j][r + 1]};
                cj[j] = d2 * (f32x2){zd[j][r], zd[j][r + 1]} * a2;
                const f32x2 g2 = d1 * a2;
                acc[1 + j][r] = g2.x; acc[1 + j][r + 1] = g2.y;
                STASH_ST(T[(size_t)(PDE_GA + 336 * (1 + j) + row0 + r) * REGF + lane], g2.x);
                STASH_ST(T[(size_t)(PDE_GA + 336 * (1 + j) + row0 + r + 1) * REGF + lane], g2.y);
            }
            f32x2 v = d1 * (f32x2){acc[0][r], acc[0][r + 1]};
            v = v + ((cj[0] + cj[1]) + (cj[2] + cj[3]));
            acc[0][r] = v.x; acc[0][r + 1] = v.y;
            STASH_ST(T[(size_t)(PDE_GA + row0 + r) * REGF + lane], v.x);
            STASH_ST(T[(size_t)(PDE_GA + row0 + r + 1) * REGF + lane], v.y);
        }
        if (l == 0) break;
        jet_exchange(xch, w, lane, acc, g);
#pragma unroll
        for (int c = 0; c < JET_NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        jet_mfma<16, 64>(a.t4[l] + (size_t)w * 16 * 64, lane, g, 0, acc);
    }
}

int ensure_jet_attrs() {
    static bool done = false;
    if (done) return 0;
    HIPCK(hipFuncSetAttribute((const void*)k_pde_jet_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, JET_LDS_BYTES));
    HIPCK(hipFuncSetAttribute((const void*)k_pde_jet_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, JET_LDS_BYTES));
    done = true;
    return 0;
}
int launch_pde_jet_fwd(const PdeJetArgs& a0, unsigned tiles, unsigned anet_wgs, hipStream_t st) {
    if (ensure_jet_attrs()) return 1;
    PdeJetArgs a = a0; a.jet_tiles = (int)tiles;
    hipLaunchKernelGGL(k_pde_jet_fwd, dim3(tiles + anet_wgs), dim3(WG_THREADS), JET_LDS_BYTES, st, a);
    LAUNCHCK();
    return 0;
}
int launch_pde_jet_bwd(const PdeJetArgs& a0, unsigned tiles, unsigned anet_wgs, hipStream_t st) {
    if (ensure_jet_attrs()) return 1;
    PdeJetArgs a = a0; a.jet_tiles = (int)tiles;
    hipLaunchKernelGGL(k_pde_jet_bwd, dim3(tiles + anet_wgs), dim3(WG_THREADS), JET_LDS_BYTES, st, a);
    LAUNCHCK();
    return 0;
}
