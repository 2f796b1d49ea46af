// vel.h - argument blocks of the velocity kernels (vel.hip)
#pragma once
#include "common.h"

#define RK_NF 19                       // per (step, sample) record fields: x3 pmid3 w1[6] w2[6] flags
#define VEL_G_REGS (5 * 64 + 16)       // adjoint stash rows per (eval, tile): gz[5][64] + gw[16]
#define MAX_RK_STEPS 64
static_assert(MAX_RK_STEPS == 64, "SCHED_* offsets in common.h assume 64 RK2 steps");

struct VelEvalArgs {
    nvfi_field_desc f;
    VelFrags Wv, Wa;
    int64_t N;
    const float* xt;   // (N,4)
    float* u6;         // (N,6)
    int gated;
};

struct Rk2Args {
    nvfi_field_desc f;
    VelFrags Wv;
    const int* count;      // device count of compact samples (NULL -> n_direct)
    int64_t n_direct;
    const int* list;       // compact -> dense index (NULL: identity)
    float4* xw;            // dense (x,y,z,zval), updated in place unless xout
    float* xout;           // optional (count,3) output instead of in-place
    // uniform mode (render): per-step dt and start time
    int nsteps;
    float dt[MAX_RK_STEPS];
    float tcur[MAX_RK_STEPS];
    const float* sched;    // optional device-side schedule record (common.h): dt / tcur are read from it instead
    // per-point mode
    const float* pt_t; const float* pt_base; float dt_max; int max_steps;
    int pt_by_list;        // pt_t / pt_base are indexed by the dense index list[i] instead of the compact index i
    // stashes (training)
    float* zst; float* x0st; float* rec; float* gst;
    int64_t cap; int64_t cap_tiles;
    int z_x4;              // the z rows of layers 0..3 are x4 stash blocks (engine.h: stash_st16_x4): written by the x6 warp kernels, read by the fused adjoint only
    // backward
    const float4* gxk;     // (dense, per sample) upstream gradient wrt the warped position
};

int launch_vel_eval(const VelEvalArgs& a, hipStream_t st);
int launch_rk2_fwd(const Rk2Args& a, int64_t cap_samples, bool uniform, hipStream_t st);
