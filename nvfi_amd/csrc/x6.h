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

struct X6PackArgs { const float* W[5]; b8_t* img; b8_t* imgT; };
// the three bfloat16 images of weight_net layers 0..4 (X6_IMAGE_BYTES at img) and - imgT, optional, round 6 - of their TRANSPOSES, the A
// operands of the dgrad (vel_fuse.hip): layer l >= 1 as 4 row tiles (input features) x 8 K steps (output features in p-space), layer 0 as
// ONE row tile (the 28 encoder inputs in slot order, like fragment T0 of engine.h) x 8 K steps - the same X6_H8 units per term
int launch_pack_x6(const float* const* W, void* img, hipStream_t st, void* imgT = nullptr);
#ifdef __HIPCC__
// x = t1 + t2 + t3 exactly: t1 = rn_bf16(x), t2 = rn_bf16(x - t1), t3 = x - t1 - t2 (8 significant bits at most: exact in bfloat16)
__device__ __forceinline__ void split3(float x, __bf16& t1, __bf16& t2, __bf16& t3) {
    t1 = (__bf16)x;
    const float r1 = x - (float)t1;
    t2 = (__bf16)r1;
    t3 = (__bf16)(r1 - (float)t2);
}
// three-term TRUNCATION split of 8 activations -> the B operands of one K step (vel_x6.hip: split3_8, which documents it): t1 = the upper
// 16 bits of x, r = x - t1 (exact), t2 = the upper 16 bits of r, t3 = r - t2 (<= 8 significant bits); x = t1 + t2 + t3 exactly
__device__ __forceinline__ void x6_split8(const float* v, b8_t& b1, b8_t& b2, b8_t& b3) {
    unsigned p1[4], p2[4], p3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float xa = v[2 * j], xb = v[2 * j + 1];
        const unsigned ua = __float_as_uint(xa), ub = __float_as_uint(xb);
        p1[j] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
        const float ra = xa - __uint_as_float(ua & 0xffff0000u), rb = xb - __uint_as_float(ub & 0xffff0000u);
        const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
        p2[j] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
        const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
        p3[j] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
    }
    typedef unsigned x6_u32x4 __attribute__((ext_vector_type(4)));
    const x6_u32x4 q1 = {p1[0], p1[1], p1[2], p1[3]}, q2 = {p2[0], p2[1], p2[2], p2[3]}, q3 = {p3[0], p3[1], p3[2], p3[3]};
    b1 = __builtin_bit_cast(b8_t, q1); b2 = __builtin_bit_cast(b8_t, q2); b3 = __builtin_bit_cast(b8_t, q3);
}
// the six term products of one K step (vel_x6.hip: x6_step): a0 += A1 B1 | a1 += A1 B2 + A2 B2 + A2 B1 + A1 B3 + A3 B1
__device__ __forceinline__ void x6_mm6(const b8_t& A1, const b8_t& A2, const b8_t& A3, const b8_t& B1, const b8_t& B2, const b8_t& B3,
                                       f32x16& a0, f32x16& a1) {
    a0 = MFMA16B(A1, B1, a0);
    a1 = MFMA16B(A1, B2, a1);
    a1 = MFMA16B(A2, B2, a1);
    a1 = MFMA16B(A2, B1, a1);
    a1 = MFMA16B(A1, B3, a1);
    a1 = MFMA16B(A3, B1, a1);
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
    if (!a.imgT) return;
    // transposed: element j of (row tile mt, K step st, lane) = W[out = slot_logical(SK_HIDDEN, 2 (8 st + j) + h)][in = row of the tile]
    int mt, st;
    if (idx < 512) { mt = 0; st = ms; } else { mt = m; st = s; }
    const int rho = 32 * mt + (lane & 31);
    const int rowL = idx < 512 ? row_logical(RK_VEL_IN, rho) : rho;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int colL = slot_logical(SK_HIDDEN, 2 * (8 * st + j) + h);
        float w = 0.f;
        if (rowL >= 0 && rowL < in) w = W[(size_t)colL * in + rowL];
        __bf16 t1, t2, t3;
        split3(w, t1, t2, t3);
        v1[j] = t1; v2[j] = t2; v3[j] = t3;
    }
    a.imgT[idx] = v1; a.imgT[X6_H8 + idx] = v2; a.imgT[2 * X6_H8 + idx] = v3;
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
