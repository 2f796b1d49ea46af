// x6.h - argument blocks of the x6 kernels (vel_x6.hip): fp32 products of the velocity net's hidden layers formed exactly from three
// binary16 terms per operand on the 16-bit matrix pipe
#pragma once
#include "common.h"
#include "engine16.h"

#define X6_SCALE 2048.f
#define X6_INV1 (1.f / 2048.f)
// per term: layer 0 = 4 row tiles x 2 K steps x 64 lanes (h8 units), layers 1..4 = 4 x 8 x 64 each
#define X6_L0 0
#define X6_LH(l) (512 + ((l) - 1) * 2048)
#define X6_H8 (512 + 4 * 2048)
#define X6_IMAGE_BYTES (3 * X6_H8 * 16)

struct X6PackArgs { const float* W[5]; h8_t* img; int* wmax_bits; };
// the three binary16 images of weight_net layers 0..4 (X6_IMAGE_BYTES at img); wmax_bits (optional, zeroed by the caller): max |w| as int bits
int launch_pack_x6(const float* const* W, void* img, int* wmax_bits, hipStream_t st);

struct X6Args {
    nvfi_field_desc f;
    const void* img;                         // X6_IMAGE_BYTES (launch_pack_x6)
    const int* count; int64_t n_direct;      // device count of list entries (NULL -> n_direct)
    const int* list; float4* xw;             // list[i] -> point index (NULL: identity); positions updated in place
    float* xout3;                            // optional (N,3) output instead of the in-place update (nvfi_integrate_pos)
    const float* pt_t; const float* pt_base; int pt_by_list; float dt_max; int max_steps;
};
int launch_rk2_x6(const X6Args& a, int64_t cap_points, hipStream_t st);
