// engine16.h - fp16-input MFMA pieces shared by the fp16 MaskField variant (mask.hip) and the fp16 pre-pass of the PDE occupancy
// prefilter (pre16.hip).  v_mfma_f32_32x32x16_f16: weights and layer inputs rounded to fp16 (RNE), products accumulated in fp32.
// One MFMA covers 16 input features: lane (n, h) supplies k = 8h + j, j = 0..7, which is made to mean "register 8s + j of the
// previous layer's D-layout output" (feature dmap(8s + j, h), engine.h) - so, as in the fp32 engine, activations never leave
// registers and the permutation lives in the packed weight fragments.  32 MFMAs of 32 cycles per 128x128 layer instead of 256 of 64.
#pragma once
#include "engine.h"

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// registers 8s .. 8s+7 of an fp32 activation array -> the B operand of K step s
template <int NS>
__device__ __forceinline__ void to_h8(const float* x, h8_t* B) {
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
        for (int j = 0; j < 8; ++j) B[sidx][j] = (_Float16)x[8 * sidx + j];
}
