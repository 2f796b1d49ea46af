// x6.h - argument blocks of the x6 kernels (vel_x6.hip): fp32 products of the velocity net's hidden layers formed exactly from three
// bfloat16 terms per operand on the 16-bit matrix pipe
#pragma once
#include "common.h"
#include "engine16.h"

// the 16-bit term type: bfloat16 (8-bit significand, the exponent range of fp32 - no scaling between the terms, no subnormal terms)
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
#define MFMA16B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// per term: layer 0 = 4 row tiles x 2 K steps x 64 lanes (h8 units), layers 1..4 = 4 x 8 x 64 each
#define X6_L0 0
#define X6_LH(l) (512 + ((l) - 1) * 2048)
#define X6_H8 (512 + 4 * 2048)
#define X6_IMAGE_BYTES (3 * X6_H8 * 16)

struct X6PackArgs { const float* W[5]; b8_t* img; };
// the three bfloat16 images of weight_net layers 0..4 (X6_IMAGE_BYTES at img)
int launch_pack_x6(const float* const* W, void* img, hipStream_t st);

struct X6Args {
    nvfi_field_desc f;
    const void* img;                         // X6_IMAGE_BYTES (launch_pack_x6)
    const int* count; int64_t n_direct;      // device count of list entries (NULL -> n_direct)
    const int* list; float4* xw;             // list[i] -> point index (NULL: identity); positions updated in place
    float* xout3;                            // optional (N,3) output instead of the in-place update (nvfi_integrate_pos)
    const float* pt_t; const float* pt_base; int pt_by_list; float dt_max; int max_steps;
};
int launch_rk2_x6(const X6Args& a, int64_t cap_points, hipStream_t st);
