// x6.h - argument blocks of the x6 kernels (vel_x6.hip): fp32 products of the velocity net's hidden layers formed exactly from three
// bfloat16 terms per operand on the 16-bit matrix pipe
#pragma once
#include "common.h"
#include "engine16.h"
#include "vel.h"

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
#ifdef __HIPCC__
// x = t1 + t2 + t3 exactly: t1 = rn_bf16(x), t2 = rn_bf16(x - t1), t3 = x - t1 - t2 (8 significant bits at most: exact in bfloat16)
__device__ __forceinline__ void split3(float x, __bf16& t1, __bf16& t2, __bf16& t3) {
    t1 = (__bf16)x;
    const float r1 = x - (float)t1;
    t2 = (__bf16)r1;
    t3 = (__bf16)(r1 - (float)t2);
}
// element idx (< X6_H8) of the three images: the A operand of row tile m, K step s, lane (engine16.h's K order: register 8 s + j of lane half h)
__device__ __forceinline__ void x6_pack_body(const X6PackArgs& a, int idx) {
    if (idx >= X6_H8) return;
    int l, local, NS, in, kind;
    if (idx < 512) { l = 0; local = idx; NS = 2; in = 28; kind = SK_VEL_IN; }
    else { l = 1 + (idx - 512) / 2048; local = (idx - 512) % 2048; NS = 8; in = 128; kind = SK_HIDDEN; }
    const int lane = local & 63, ms = local >> 6, s = ms % NS, m = ms / NS;
    const int row = 32 * m + (lane & 31), h = lane >> 5;
    const float* W = a.W[l];
    b8_t v1, v2, v3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int feat = slot_logical(kind, 2 * (8 * s + j) + h);
        float w = 0.f;
        if (feat >= 0 && feat < in) w = W[(size_t)row * in + feat];
        __bf16 t1, t2, t3;
        split3(w, t1, t2, t3);
        v1[j] = t1; v2[j] = t2; v3[j] = t3;
    }
    a.img[idx] = v1; a.img[X6_H8 + idx] = v2; a.img[2 * X6_H8 + idx] = v3;
}
#endif

struct X6Args {
    nvfi_field_desc f;
    const void* img;                         // X6_IMAGE_BYTES (launch_pack_x6)
    const int* count; int64_t n_direct;      // device count of list entries (NULL -> n_direct)
    const int* list; float4* xw;             // list[i] -> point index (NULL: identity); positions updated in place
    float* xout3;                            // optional (N,3) output instead of the in-place update (nvfi_integrate_pos)
    const float* pt_t; const float* pt_base; int pt_by_list; float dt_max; int max_steps;
};
int launch_rk2_x6(const X6Args& a, int64_t cap_points, hipStream_t st);
int launch_rk2_x6w(const X6Args& a, int64_t cap_points, hipStream_t st);      // vel_x6w.hip: one wave per tile, the epilogue in the MFMAs' VALU slots
// the render warp (uniform step schedule, optional training stash) on the x6 evaluation: same arguments as the fp32 kernel + the image
struct X6UniArgs { Rk2Args r; const void* img; };
int launch_rk2_x6_uni(const X6UniArgs& a, int64_t cap_samples, bool stash, hipStream_t st);
int launch_rk2_x6w_uni(const X6UniArgs& a, int64_t cap_samples, bool stash, hipStream_t st);   // vel_x6w.hip
