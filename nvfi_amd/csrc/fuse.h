// fuse.h - argument block of the fused RK2 adjoint + hidden-layer weight-gradient kernel (vel_fuse.hip)
#pragma once
#include "common.h"
#include "vel.h"

struct FuseBwdArgs {
    Rk2Args r;
    const float4* t4[6];       // x4 transposed fragments (pack_vel_x4_bwd)
    const void* imgT;          // round 6: the transposed x6 images (x6.h: X6PackArgs::imgT) - the dgrad of layers 4..0 on the 16-bit matrix pipe
    float* slabs;              // slab of layer l (1..4) and workgroup g at slabs + l * layer_stride + g * slab_floats
    int64_t layer_stride;      // floats
    int slab_floats;           // 128 * 128 + 128
    unsigned long long* timing; // -DFUSE_TIMING builds only
};
// max_slabs: slabs the buffer holds per layer; *nslab_out: slabs written per hidden layer (= workgroups launched; 0: nothing launched)
int launch_rk2_fuse_bwd(const FuseBwdArgs& a, int64_t cap_samples, int max_slabs, int* nslab_out, hipStream_t st);

// sixteen x4 fragment groups of one wave: four wave-uniform base pointers (opaque to the optimiser, see opaque_u in vel_fuse.hip) with
// immediate offsets - left alone, the compiler hoists sixteen 64-bit per-lane addresses out of the persistent loop and spills them
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4v* gcf4p;
__device__ __forceinline__ void split_load16(const float4* a4, int lane, f32x4v (&wq)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        gcf4p b = (gcf4p)(a4 + q * 4 * 64);
        asm("" : "+s"(b));
#pragma unroll
        for (int k = 0; k < 4; ++k) wq[4 * q + k] = b[k * 64 + lane];
    }
}

// ---------------------------------------------------------------- shared by the fused adjoint + weight-gradient kernels (vel_fuse.hip, pde_fuse.hip)
#define FUSE_HR 33                                    // float4 per half row (32 samples of four consecutive p rows) + one float4 of padding
#define FUSE_XB (16 * 2 * FUSE_HR)                    // float4 per exchange buffer: 16 row groups x 2 halves
#define FUSE_TF (4 * 2 * FUSE_HR * 4)                 // floats per 32-row tile of an exchange buffer
// LDS writes of this wave have landed, then the workgroup barrier; outstanding global loads (the next layer's z rows and weights) stay in
// flight across it (a __syncthreads() would be free to wait for them)
#define FUSE_BAR() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// A wave-uniform row-block pointer the optimiser cannot see through: every stash access of the block becomes SGPR base + lane offset +
// immediate (left to itself, loop strength reduction keeps one 64-bit VGPR induction pointer per stash ROW across the layer loop: 32 registers)
// (the asm sees a GLOBAL pointer: through a generic one the compiler falls back to flat_load / flat_store)
typedef const __attribute__((address_space(1))) float* gcfp; typedef __attribute__((address_space(1))) float* gfp;
__device__ __forceinline__ gcfp opaque_u(const float* p) { gcfp q = (gcfp)p; asm("" : "+s"(q)); return q; }
__device__ __forceinline__ gfp opaque_u(float* p) { gfp q = (gfp)p; asm("" : "+s"(q)); return q; }

// LDS-DMA of one dword per lane: lane L's word at base + voff lands at lds_dst + 4 L, without passing through a VGPR - a prefetch that holds no
// register and that the compiler can neither spill nor wait for (inline asm is outside its vmcnt bookkeeping: the reader waits itself).
// M0 is written and restored inside the statement (wgrad_ring.hip: glds16).
__device__ __forceinline__ void glds4(const float* base, int voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}


// ---------------------------------------------------------------- pde_fuse.hip: adjoint of the PDE Jacobian program + the hidden-layer weight gradients
struct PdeFuseArgs {
    const float4* t4[6];       // x4 transposed fragments of weight_net (t4[1..4]: 4 tiles x 64 steps, t4[5]: 4 tiles x 4 steps)
    const int* kcount; int64_t first, cap;     // device count of kept points; this pass handles [first, first + cap)
    float* stash; const float* seeds;
    float* slabs;              // slab of hidden layer L (0..3 = weight layers 1..4) and workgroup g at slabs + L * layer_stride + g * slab_floats
    int64_t layer_stride;      // floats
    int slab_floats;           // 128 * 128 + 128
    unsigned long long* timing; // -DPF_TIMING builds only
    // second half: the acceleration net's adjoint + its four hidden-layer weight gradients (do_accel)
    int do_accel;
    const float4* ta4[6];      // x4 transposed fragments of a_weight_net
    float* slabs_a;            // same geometry as slabs
    int x4;                    // the forward wrote layers 0..3 of z / zd_j as x4 stash blocks (PdeJetArgs::x4)
    int* queue;                // device word, zero at launch: next acceleration-net tile (NULL: static share, tile = workgroup + k * grid)
};
int launch_pde_fuse_bwd(const PdeFuseArgs& a, int64_t cap_points, int max_slabs, int* nslab_out, hipStream_t st);
