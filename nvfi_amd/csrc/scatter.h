// scatter.h - sorted-tile plane-gradient scatter and the (sample, channel-quad) gather kernels (scatter.hip)
#pragma once
#include "common.h"
#include "render.h"

#define SCATTER_T 8            // tile edge in texels (LDS tile = (T+1)^2 x 24 floats + a 2 x (T+1) x 24 strip of the paired time plane)
#ifndef SCATTER_CHUNK
#define SCATTER_CHUNK 256
#endif
//     // samples per (plane, tile) work item
#define SCATTER_MAX_BINS 6144  // 3 planes x tiles (8x8-texel LDS tiles); larger grids fall back to the atomic scatter
#define SCATTER_T_MFMA 4       // MFMA scatter: 4x4-texel tiles (5x5 with the apron = 25 rows of two 16-row MFMA tiles), 64 samples per item
#ifndef SCATTER_CHUNK_MFMA
#define SCATTER_CHUNK_MFMA 128     // samples per item: a wave walks them in batches of 64
#endif
#define SCATTER_MAX_BINS_MFMA 8192

struct TileGeom { int G[3]; int ntx[3]; int nty[3]; int boff[3]; int nbins; int T; int chunk; int chunk_shift; int colmajor; };   // colmajor: tiles of planes 0 and 2 are numbered
//  column by column, so that consecutive bins (and items) share the texel columns of the paired time plane (plane 1 pairs by rows: row-major)

struct TileWork {              // device buffers inside the caller's workspace
    TileGeom g;
    int* hist; int* cursor; int* nitems; int4* items; float4* sorted; float* og;
    int* start; int* istart;   // [nbins + 1] first sample / first item of every bin (the fill kernel writes the item list from them)
    int64_t cap_items;
};

struct OgArgs {
    nvfi_field_desc f;
    const int* count; const int* list; const float4* xw; float tn;
    const float* sched;
    const float* gxpre;        // density: one upstream gradient per sample (dense)
    const float* gg;           // appearance: (M,48) per-sample channel gradients (compact)
    float* og;                 // [i][6][C]
    const uint8_t* mflag; const float4* gxw; float4* gxk;   // COORD: coordinate gradients (density branch)
    float4* gxw_acc;           // COORD, appearance branch: gxw[n] += plane part of the coordinate gradient
};

struct TileSortArgs {
    TileGeom g;
    const int* count; const int* list; const float4* xw;
    int* hist; int* cursor; int4* items; int* nitems; float4* sorted; int* start; int* istart;
};

struct TileSortArgs2 { TileSortArgs j[2]; };

struct TileScatterArgs {
    nvfi_field_desc f;
    TileGeom geo;
    const int4* items; const int* nitems; const float4* sorted; const int* list; const float4* xw; const float* og;
    float tn; int y0;
    const float* sched;
    nvfi_grads g;
};

int tile_geom(const nvfi_field_desc* f, TileGeom* g);
void plan_tile_scatter(Bump& B, const nvfi_field_desc* f, int64_t N, TileWork* w);
int ensure_scatter_attrs();
int launch_og(const nvfi_field_desc* f, const OgArgs& oa, int C, bool coord, int64_t N, hipStream_t st);
int launch_tile_scatter(const nvfi_field_desc* f, const TileWork& w, const int* count, const int* list, const float4* xw, float tn,
                        const nvfi_grads& g, int C, int64_t N, hipStream_t st, const float* sched = nullptr, bool sorted = false);
int launch_tile_sort(const TileWork* const* w, const int* const* count, const int* const* list, int njobs, const float4* xw, int64_t N, hipStream_t st);
int launch_app_feat(const OgArgs& oa, int64_t N, hipStream_t st);   // oa.og: feat[i][48], the appearance feature of masked sample i (Ca == 48)
int launch_density_q(const DensityArgs& da, int64_t N, hipStream_t st);   // Cd == 24 only
