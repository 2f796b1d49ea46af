// frags.h - persistent fragment cache of a field (round 5, ABI v5: nvfi_field_desc.frags)
//
// The MFMA kernels read the Linear weights in fragment order (engine.h).  Rounds 1-4 repacked them inside every call into the call's
// workspace - 3 k_pack + 4 k_frag_x4 launches per training iteration for weights that change once per iteration.  nvfi_pack_frags writes
// EVERY fragment set the render and PDE calls use - render MLP + basis_mat (forward, transposed), both velocity nets (forward, transposed,
// biases) and the x4 copies of the feature-split kernels - into one caller-owned buffer in ONE launch; a call whose descriptor carries the
// buffer (nvfi_field_desc.frags != NULL) skips its pack launches and reads the fragments from it.  The caller repacks after every change of
// the weights (the Python mirror keys the buffer on the parameters' versions).
#pragma once
#include "common.h"
#include "pde.h"
#include "x6.h"

#define A_X4B_FLOATS (4 * X4_FLOATS(4, 64) + X4_FLOATS(4, 4))      // transposed x4 fragments t[1..5] of a_weight_net (pde_fuse.hip)
struct FragCache { float *render, *vel, *anet, *vel_x4f, *vel_x4b, *a_x4b; void* vel_x6; void* vel_x6t; int64_t total; };
static inline void frag_cache_layout(const float* base, FragCache* c) {
    Bump B{(char*)base, 0, 0};
    c->render = B.take<float>(RENDER_FRAG_FLOATS);
    c->vel = B.take<float>(VEL_FRAG_FLOATS);
    c->anet = B.take<float>(VEL_FRAG_FLOATS);
    c->vel_x4f = B.take<float>(VEL_X4F_FLOATS);
    c->vel_x4b = B.take<float>(VEL_X4B_FLOATS);
    c->a_x4b = B.take<float>(A_X4B_FLOATS);
    c->vel_x6 = B.take<float>(X6_IMAGE_BYTES / 4);         // the three bfloat16 images of weight_net's layers 0..4 (vel_x6.hip)
    c->vel_x6t = B.take<float>(X6_IMAGE_BYTES / 4);        // ... and of their transposes (round 6: the dgrad of vel_fuse.hip on x6)
    c->total = align_up(B.off, 256);
}
// pointer tables into the x4 regions (the layouts pack_vel_x4_fwd / pack_vel_x4_bwd write)
static inline void x4f_pointers(const float* buf, const float4** f4) {
    const float* p = buf;
    f4[0] = reinterpret_cast<const float4*>(p); p += X4_FLOATS(4, 14);
    for (int l = 1; l <= 4; ++l) { f4[l] = reinterpret_cast<const float4*>(p); p += X4_FLOATS(4, 64); }
    f4[5] = reinterpret_cast<const float4*>(p);
}
static inline void x4b_pointers(const float* buf, const float4** t4) {      // T0 (1 tile x 64 steps), t1..t4, t5 (4 tiles x 4 steps)
    const float* p = buf;
    t4[0] = reinterpret_cast<const float4*>(p); p += X4_FLOATS(1, 64);
    for (int l = 1; l <= 4; ++l) { t4[l] = reinterpret_cast<const float4*>(p); p += X4_FLOATS(4, 64); }
    t4[5] = reinterpret_cast<const float4*>(p);
}
static inline void a_x4b_pointers(const float* buf, const float4** ta4) {   // t1..t4, t5 of a_weight_net
    const float* p = buf;
    ta4[0] = nullptr;
    for (int l = 1; l <= 4; ++l) { ta4[l] = reinterpret_cast<const float4*>(p); p += X4_FLOATS(4, 64); }
    ta4[5] = reinterpret_cast<const float4*>(p);
}
int launch_pack_all(const PackJobsAll& jobs, const X6PackArgs* x6, hipStream_t st);
