// fuse.h - argument block of the fused RK2 adjoint + hidden-layer weight-gradient kernel (vel_fuse.hip)
#pragma once
#include "common.h"
#include "vel.h"

struct FuseBwdArgs {
    Rk2Args r;
    const float4* t4[6];       // x4 transposed fragments (pack_vel_x4_bwd)
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
