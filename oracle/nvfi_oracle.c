/*
 * nvfi_oracle.c - CPU ORACLE (test infrastructure, NOT product code).  See nvfi_oracle.h.
 *
 * Each function cites the reference file:line it restates (paths relative to the reference
 * repository root).  Arithmetic is fp32 in the reference's operation order wherever the order
 * is visible from the Python source; ATen kernels (grid_sampler_2d, softplus, cumprod, linear)
 * are restated from their published semantics and pinned by the golden
 * vectors in tests/golden/.
 */
#include "nvfi_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HID 128
#define ENC 28
#define RIN 110
#define MAXC 64

static int g_threads = 1;
static inline int max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
int orc_num_threads(void) { return max_threads(); }
void orc_set_num_threads(int n) {
    g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
    omp_set_num_threads(g_threads);
#endif
}
static inline int tid(void) {
#ifdef _OPENMP
    return omp_get_thread_num();
#else
    return 0;
#endif
}

/* ------------------------------------------------------------------ small math */
static inline float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
static inline float siluf_(float x) { return x / (1.f + expf(-x)); }
static inline float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); } /* F.softplus beta=1 thr=20 */

/* tensorf_base.py:241-242 */
static inline void normalize_coord(const orc_field_t* f, const float* p, float* xn) {
    for (int c = 0; c < 3; ++c) {
        float size = f->aabb[3 + c] - f->aabb[c];
        float inv = 2.0f / size;
        xn[c] = (p[c] - f->aabb[c]) * inv - 1.f;
    }
}
/* tensorf_keyframe.py:501-506 */
static inline float normalize_time(const orc_field_t* f, float t) {
    if (f->K == 1 || f->tmax == 0.f) return t * 0.f;
    return t * 2.f / f->tmax - 1.f;
}
/* tensorf_keyframe.py:646-654 (+ torch.isclose defaults, :683) */
static inline float time_scale(const orc_field_t* f) {
    return f->K > 1 ? (float)((double)f->tmax / (double)(f->K - 1)) : 1.f;
}
static inline float snap_base(const orc_field_t* f, float t) {
    float ts = time_scale(f);
    float q = t / ts;
    float hi = (float)(f->K - 1);
    if (q < 0.f) q = 0.f;
    if (q > hi) q = hi;
    return rintf(q) * ts; /* torch.round = round-half-even */
}
static inline int is_close(float a, float b) { return fabsf(a - b) <= 1e-8f + fabsf(1e-5f * b); }
static inline float dt_max_of(const orc_field_t* f) {
    return f->K > 1 ? (float)(0.5 * (double)f->tmax / (double)(f->K - 1)) : 1.f;
}

/* ------------------------------------------------------------------ bilinear (ATen grid_sampler_2d, align_corners=True, zeros) */
typedef struct {
    int x0, y0;
    float w, e, n, s; /* w = x-floor(x), e = 1-w, n = y-floor(y), s = 1-n */
    int m[4];         /* nw, ne, sw, se in-bounds */
} bl_t;

static inline void bl_setup(float gx, float gy, int W, int H, bl_t* b) {
    float x = (gx + 1.f) * ((float)(W - 1) / 2.f);
    float y = (gy + 1.f) * ((float)(H - 1) / 2.f);
    float xf = floorf(x), yf = floorf(y);
    b->w = x - xf; b->e = 1.f - b->w;
    b->n = y - yf; b->s = 1.f - b->n;
    if (!(xf > -4.f)) xf = -4.f; if (xf > (float)W + 2.f) xf = (float)W + 2.f;
    if (!(yf > -4.f)) yf = -4.f; if (yf > (float)H + 2.f) yf = (float)H + 2.f;
    int x0 = (int)xf, y0 = (int)yf;
    b->x0 = x0; b->y0 = y0;
    int xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
    int yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
    b->m[0] = xin0 && yin0; b->m[1] = xin1 && yin0; b->m[2] = xin0 && yin1; b->m[3] = xin1 && yin1;
}
/* out[c] for c<C; plane (C,H,W). Also returns the 4 tap values if tv != NULL (C x 4). */
static inline void bl_sample(const float* plane, int C, int H, int W, const bl_t* b, float* out, float* tv) {
    float nw = b->e * b->s, ne = b->w * b->s, sw = b->e * b->n, se = b->w * b->n;
    for (int c = 0; c < C; ++c) {
        const float* p = plane + (size_t)c * H * W;
        float v0 = b->m[0] ? p[(size_t)b->y0 * W + b->x0] : 0.f;
        float v1 = b->m[1] ? p[(size_t)b->y0 * W + b->x0 + 1] : 0.f;
        float v2 = b->m[2] ? p[(size_t)(b->y0 + 1) * W + b->x0] : 0.f;
        float v3 = b->m[3] ? p[(size_t)(b->y0 + 1) * W + b->x0 + 1] : 0.f;
        out[c] = v0 * nw + v1 * ne + v2 * sw + v3 * se;
        if (tv) { tv[c * 4 + 0] = v0; tv[c * 4 + 1] = v1; tv[c * 4 + 2] = v2; tv[c * 4 + 3] = v3; }
    }
}
/* backward: gout[c] -> scatter into gplane (atomic), coordinate grads (gx, gy) */
static inline void bl_backward(const float* plane, float* gplane, int C, int H, int W, const bl_t* b,
                               const float* gout, float* ggx, float* ggy) {
    float nw = b->e * b->s, ne = b->w * b->s, sw = b->e * b->n, se = b->w * b->n;
    float gx = 0.f, gy = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* p = plane + (size_t)c * H * W;
        float g = gout[c];
        size_t i0 = (size_t)b->y0 * W + b->x0;
        float v0 = b->m[0] ? p[i0] : 0.f;
        float v1 = b->m[1] ? p[i0 + 1] : 0.f;
        float v2 = b->m[2] ? p[i0 + W] : 0.f;
        float v3 = b->m[3] ? p[i0 + W + 1] : 0.f;
        gx += ((v1 - v0) * b->s + (v3 - v2) * b->n) * g;
        gy += ((v2 - v0) * b->e + (v3 - v1) * b->w) * g;
        if (gplane) {
            float* q = gplane + (size_t)c * H * W;
            if (b->m[0]) {
#pragma omp atomic
                q[i0] += nw * g;
            }
            if (b->m[1]) {
#pragma omp atomic
                q[i0 + 1] += ne * g;
            }
            if (b->m[2]) {
#pragma omp atomic
                q[i0 + W] += sw * g;
            }
            if (b->m[3]) {
#pragma omp atomic
                q[i0 + W + 1] += se * g;
            }
        }
    }
    *ggx = gx * ((float)(W - 1) / 2.f);
    *ggy = gy * ((float)(H - 1) / 2.f);
}

/* plane geometry: matModeSpace = [0,1],[0,2],[1,2]; matModeTime first axis = 2,1,0 (tensorf_keyframe.py:39-40) */
static const int MS_A[3] = {0, 0, 1};
static const int MS_B[3] = {1, 2, 2};
static const int MT_C[3] = {2, 1, 0};

/* compute_densityfeature / compute_appfeature core (tensorf_keyframe.py:233-310): per-channel
 * product of the 6 bilinear lookups.  prod[c] = (ps0*ps1*ps2) * (pt0*pt1*pt2). vals (6 x C) optional. */
static void plane_products(const orc_field_t* f, const float* const* ps, const float* const* pt, int C,
                           const float* xyzt, float* prod, float* vals, bl_t* bls) {
    float buf[6][MAXC];
    bl_t b[6];
    for (int i = 0; i < 3; ++i) {
        int a = MS_A[i], bb = MS_B[i], c = MT_C[i];
        bl_setup(xyzt[a], xyzt[bb], f->G[a], f->G[bb], &b[i]);
        bl_sample(ps[i], C, f->G[bb], f->G[a], &b[i], buf[i], NULL);
        bl_setup(xyzt[c], xyzt[3], f->G[c], f->K, &b[3 + i]);
        bl_sample(pt[i], C, f->K, f->G[c], &b[3 + i], buf[3 + i], NULL);
    }
    for (int c = 0; c < C; ++c) {
        float s = buf[0][c]; s = s * buf[1][c]; s = s * buf[2][c];
        float t = buf[3][c]; t = t * buf[4][c]; t = t * buf[5][c];
        prod[c] = s * t;
    }
    if (vals) for (int i = 0; i < 6; ++i) memcpy(vals + i * C, buf[i], sizeof(float) * C);
    if (bls) memcpy(bls, b, sizeof(b));
}
/* backward of plane_products: gprod[c] -> plane grads + d/d(x,y,z) (time coordinate carries no grad) */
static void plane_products_bwd(const orc_field_t* f, const float* const* ps, const float* const* pt,
                               float* const* gps, float* const* gpt, int C, const float* xyzt,
                               const float* gprod, float* gxyz) {
    float vals[6 * MAXC], prod[MAXC];
    bl_t b[6];
    plane_products(f, ps, pt, C, xyzt, prod, vals, b);
    float g[MAXC];
    for (int i = 0; i < 6; ++i) {
        for (int c = 0; c < C; ++c) {
            float o = 1.f;
            for (int k = 0; k < 6; ++k) if (k != i) o *= vals[k * C + c];
            g[c] = gprod[c] * o;
        }
        float gx, gy;
        if (i < 3) {
            int a = MS_A[i], bb = MS_B[i];
            bl_backward(ps[i], gps ? gps[i] : NULL, C, f->G[bb], f->G[a], &b[i], g, &gx, &gy);
            gxyz[a] += gx; gxyz[bb] += gy;
        } else {
            int c3 = MT_C[i - 3];
            bl_backward(pt[i - 3], gpt ? gpt[i - 3] : NULL, C, f->K, f->G[c3], &b[i], g, &gx, &gy);
            gxyz[c3] += gx;
        }
    }
}

void orc_density_feature(const orc_field_t* f, int64_t N, const float* xyzt, float* feat) {
#pragma omp parallel for
    for (int64_t n = 0; n < N; ++n) {
        float prod[MAXC];
        plane_products(f, f->dps, f->dpt, f->Cd, xyzt + 4 * n, prod, NULL, NULL);
        float s = 0.f;
        for (int c = 0; c < f->Cd; ++c) s += prod[c];
        feat[n] = s;
    }
}
static void app_feature_one(const orc_field_t* f, const float* xyzt, float* g48, float* feat) {
    plane_products(f, f->aps, f->apt, f->Ca, xyzt, g48, NULL, NULL);
    for (int o = 0; o < f->app_dim; ++o) {
        float s = 0.f;
        const float* w = f->basis + (size_t)o * f->Ca;
        for (int c = 0; c < f->Ca; ++c) s += w[c] * g48[c];
        feat[o] = s;
    }
}
void orc_app_feature(const orc_field_t* f, int64_t N, const float* xyzt, float* feat) {
#pragma omp parallel for
    for (int64_t n = 0; n < N; ++n) {
        float g[MAXC];
        app_feature_one(f, xyzt + 4 * n, g, feat + (size_t)n * f->app_dim);
    }
}
/* feature2density (tensorf_keyframe.py:312-321), Density mode + softplus */
void orc_feature2density(const orc_field_t* f, int64_t N, const float* feat, float* sigma) {
    for (int64_t n = 0; n < N; ++n) sigma[n] = softplusf_(feat[n] + f->density_shift);
}

/* ------------------------------------------------------------------ a-3 sample_ray */
static int any_origin_inside(const orc_field_t* f, int64_t R, const float* o) {
    /* tensorf_base.py:294 - elementwise test, then .any() over every coordinate of every ray */
    for (int64_t i = 0; i < R; ++i)
        for (int c = 0; c < 3; ++c)
            if (f->aabb[c] <= o[3 * i + c] && o[3 * i + c] <= f->aabb[3 + c]) return 1;
    return 0;
}
static inline float ray_tmin(const orc_field_t* f, int inside, const float* o, const float* d) {
    if (inside) return f->near_;
    float m = -INFINITY;
    for (int c = 0; c < 3; ++c) {
        float vec = d[c] == 0.f ? 1e-6f : d[c];
        float ra = (f->aabb[3 + c] - o[c]) / vec;
        float rb = (f->aabb[c] - o[c]) / vec;
        float mn = ra < rb ? ra : rb;
        if (mn > m) m = mn;
    }
    if (m < f->near_) m = f->near_;
    if (m > f->far_) m = f->far_;
    return m;
}
static inline float sample_z(const orc_field_t* f, float tmin, int j, float u) {
    float rng = (float)j + u;          /* rng += rand  (tensorf_base.py:302-305) */
    float step = f->step_size * rng;   /* step = stepsize * rng */
    return tmin + step;                /* interpx = t_min + step */
}
static inline int sample_point(const orc_field_t* f, const float* o, const float* d, float z, float* p) {
    int ok = 1;
    for (int c = 0; c < 3; ++c) {
        p[c] = o[c] + d[c] * z;
        if (f->aabb[c] > p[c] || p[c] > f->aabb[3 + c]) ok = 0;
    }
    return ok;
}
void orc_sample_ray(const orc_field_t* f, int64_t R, const float* o, const float* d, const float* u,
                    float* pts, float* z, uint8_t* valid) {
    int inside = any_origin_inside(f, R, o);
    int S = f->n_samples;
    for (int64_t r = 0; r < R; ++r) {
        float tmin = ray_tmin(f, inside, o + 3 * r, d + 3 * r);
        float ur = u ? u[r] : 0.f;
        for (int j = 0; j < S; ++j) {
            float p[3];
            float zz = sample_z(f, tmin, j, ur);
            int ok = sample_point(f, o + 3 * r, d + 3 * r, zz, p);
            z[r * S + j] = zz;
            valid[r * S + j] = (uint8_t)ok;
            if (pts) memcpy(pts + (r * S + j) * 3, p, 12);
        }
    }
}

/* ------------------------------------------------------------------ velocity basis (velocity_field.py:54-98) */
/* PositionEncoder(3) (base_network.py:42-54): [q, sin q, cos q, sin 2q, cos 2q, sin 4q, cos 4q] */
static inline void vel_encode(const float* q, float* e) {
    for (int c = 0; c < 4; ++c) e[c] = q[c];
    float fr = 1.f;
    for (int k = 0; k < 3; ++k) {
        for (int c = 0; c < 4; ++c) {
            float a = q[c] * fr;
            e[4 + 8 * k + c] = sinf(a);
            e[8 + 8 * k + c] = cosf(a);
        }
        fr *= 2.f;
    }
}
static inline void linear(const float* W, const float* b, int out, int in, const float* x, float* y) {
    for (int o = 0; o < out; ++o) {
        float s = b ? b[o] : 0.f;
        const float* w = W + (size_t)o * in;
        for (int k = 0; k < in; ++k) s += w[k] * x[k];
        y[o] = s;
    }
}
/* IEEE binary16 round-to-nearest-even of an fp32 value, returned as fp32 (subnormals kept, overflow -> inf): what v_cvt_f16_f32 does
 * to an MFMA operand of the product's opt-in fp16-input inference mode (nvfi_field_desc.vel_fp16, pre16.hip). */
static inline float f16_round(float x) {
    union { float f; uint32_t u; } v; v.f = x;
    const uint32_t sign = v.u & 0x80000000u;
    const int32_t ex = (int32_t)((v.u >> 23) & 0xff);
    uint32_t man = v.u & 0x7fffffu;
    if (ex == 255) return x;                                  /* inf / nan */
    const int32_t e = ex - 127;                               /* unbiased */
    if (e > 15) { v.u = sign | 0x7f800000u; return v.f; }
    int drop;                                                 /* low mantissa bits that binary16 cannot hold */
    if (e >= -14) drop = 13;
    else {
        if (e < -25) { v.u = sign; return v.f; }              /* below half of the smallest subnormal */
        drop = 13 + (-14 - e);
    }
    uint32_t full = (ex ? 0x800000u : 0u) | man;              /* 24-bit significand */
    const uint32_t half = 1u << (drop - 1), mask = (1u << drop) - 1u;
    const uint32_t rem = full & mask;
    full &= ~mask;
    if (rem > half || (rem == half && (full & (1u << drop)))) full += (1u << drop);
    /* back to fp32: value = full * 2^(e - 23) */
    const float r = ldexpf((float)full, e - 23);
    if (r >= 65520.f) { v.u = sign | 0x7f800000u; return v.f; }
    v.f = r; v.u |= sign;
    return v.f;
}
static int g_vel_fp16 = 0;      /* test switch: the velocity nets' FORWARD in the arithmetic of the fp16-input MFMA mode */
void orc_set_vel_fp16(int on) { g_vel_fp16 = on; }
float orc_f16_round(float x) { return f16_round(x); }
/* a Linear layer whose weights and inputs are rounded to binary16 before the product; fp32 accumulation from the bias */
static inline void linear16(const float* W, const float* b, int out, int in, const float* x, float* y) {
    float xr[HID];
    for (int k = 0; k < in; ++k) xr[k] = f16_round(x[k]);
    for (int o = 0; o < out; ++o) {
        float s = b ? b[o] : 0.f;
        const float* w = W + (size_t)o * in;
        for (int k = 0; k < in; ++k) s += f16_round(w[k]) * xr[k];
        y[o] = s;
    }
}
/* weight_net / a_weight_net forward. zs: 5*HID pre-activations (optional). act: 1 SiLU, 0 ReLU */
static void velnet_fwd(const float* const* W, const float* const* b, int act, const float* q, float* zs, float* out6) {
    float e[ENC], h[HID], z[HID];
    void (*lin)(const float*, const float*, int, int, const float*, float*) = g_vel_fp16 ? linear16 : linear;
    vel_encode(q, e);
    lin(W[0], b[0], HID, ENC, e, z);
    for (int l = 0; l < 5; ++l) {
        if (zs) memcpy(zs + l * HID, z, sizeof(z));
        for (int i = 0; i < HID; ++i) h[i] = act ? siluf_(z[i]) : (z[i] > 0.f ? z[i] : 0.f);
        if (l < 4) lin(W[l + 1], b[l + 1], HID, HID, h, z);
    }
    lin(W[5], b[5], 6, HID, h, out6);
}
static inline float act_f(int act, float z) { return act ? siluf_(z) : (z > 0.f ? z : 0.f); }
static inline float act_d1(int act, float z) {
    if (!act) return z > 0.f ? 1.f : 0.f;
    float s = sigmoidf_(z);
    return s * (1.f + z * (1.f - s));
}
static inline float act_d2(int act, float z) {
    if (!act) return 0.f;
    float s = sigmoidf_(z);
    return s * (1.f - s) * (2.f + z * (1.f - 2.f * s));
}
/* reverse of velnet_fwd: gout6 -> parameter grads (gW/gb, may be NULL) and gq (4, optional, accumulated) */
static void velnet_bwd(const float* const* W, int act, const float* q, const float* zs, const float* gout6,
                       float* const* gW, float* const* gb, float* gq) {
    float gh[HID], gz[HID], hprev[HID], e[ENC];
    /* last layer */
    for (int i = 0; i < HID; ++i) hprev[i] = act_f(act, zs[4 * HID + i]);
    for (int k = 0; k < HID; ++k) gh[k] = 0.f;
    for (int o = 0; o < 6; ++o) {
        float g = gout6[o];
        if (gb) gb[5][o] += g;
        for (int k = 0; k < HID; ++k) {
            if (gW) gW[5][o * HID + k] += g * hprev[k];
            gh[k] += W[5][o * HID + k] * g;
        }
    }
    for (int l = 4; l >= 0; --l) {
        for (int i = 0; i < HID; ++i) gz[i] = gh[i] * act_d1(act, zs[l * HID + i]);
        if (l > 0) {
            for (int i = 0; i < HID; ++i) hprev[i] = act_f(act, zs[(l - 1) * HID + i]);
            for (int k = 0; k < HID; ++k) gh[k] = 0.f;
            for (int o = 0; o < HID; ++o) {
                float g = gz[o];
                if (gb) gb[l][o] += g;
                for (int k = 0; k < HID; ++k) {
                    if (gW) gW[l][o * HID + k] += g * hprev[k];
                    gh[k] += W[l][o * HID + k] * g;
                }
            }
        } else {
            float ge[ENC];
            vel_encode(q, e);
            for (int k = 0; k < ENC; ++k) ge[k] = 0.f;
            for (int o = 0; o < HID; ++o) {
                float g = gz[o];
                if (gb) gb[0][o] += g;
                for (int k = 0; k < ENC; ++k) {
                    if (gW) gW[0][o * ENC + k] += g * e[k];
                    ge[k] += W[0][o * ENC + k] * g;
                }
            }
            if (gq) {
                for (int c = 0; c < 4; ++c) {
                    float s = ge[c], fr = 1.f;
                    for (int k = 0; k < 3; ++k) {
                        /* d sin(fr q) = fr cos ; d cos(fr q) = -fr sin */
                        s += fr * (e[8 + 8 * k + c] * ge[4 + 8 * k + c] - e[4 + 8 * k + c] * ge[8 + 8 * k + c]);
                        fr *= 2.f;
                    }
                    gq[c] += s;
                }
            }
        }
    }
}
/* v = sum_i w_i b_i(x) (velocity_field.py:77-93) */
static inline void vel_from_w(const float* w, const float* x, float* v) {
    v[0] = w[0] - w[4] * x[2] + w[5] * x[1];
    v[1] = w[1] + w[3] * x[2] - w[5] * x[0];
    v[2] = w[2] - w[3] * x[1] + w[4] * x[0];
}
static inline void acc_from_w(const float* aw, const float* x, float* a) {
    a[0] = aw[0] - aw[4] * x[0] - aw[5] * x[0];
    a[1] = aw[1] - aw[3] * x[1] - aw[5] * x[1];
    a[2] = aw[2] - aw[3] * x[2] - aw[4] * x[2];
}
static inline int gated_out(const orc_field_t* f, const float* x) {
    for (int c = 0; c < 3; ++c)
        if (x[c] < f->gate_lo[c] || x[c] > f->gate_hi[c]) return 1;
    return 0;
}
void orc_vel_net(const orc_field_t* f, int64_t N, const float* xt, float* u6) {
#pragma omp parallel for
    for (int64_t n = 0; n < N; ++n) {
        float w[6], aw[6];
        velnet_fwd(f->vW, f->vb, 1, xt + 4 * n, NULL, w);
        velnet_fwd(f->aW, f->ab, 0, xt + 4 * n, NULL, aw);
        vel_from_w(w, xt + 4 * n, u6 + 6 * n);
        acc_from_w(aw, xt + 4 * n, u6 + 6 * n + 3);
    }
}
void orc_get_vel(const orc_field_t* f, int64_t N, const float* xt, float* v3) {
#pragma omp parallel for
    for (int64_t n = 0; n < N; ++n) {
        float w[6];
        velnet_fwd(f->vW, f->vb, 1, xt + 4 * n, NULL, w);
        vel_from_w(w, xt + 4 * n, v3 + 3 * n);
    }
}
/* VelocityAABB / VelocityAABBSur forward (velocity_field.py:28-33, 46-51) */
static inline void vel_gated_one(const orc_field_t* f, const float* q, float* zs, float* w6, float* v, int* gate) {
    if (gated_out(f, q)) { v[0] = v[1] = v[2] = 0.f; *gate = 1; return; }
    *gate = 0;
    velnet_fwd(f->vW, f->vb, 1, q, zs, w6);
    vel_from_w(w6, q, v);
}
void orc_vel_gated(const orc_field_t* f, int64_t N, const float* xt, float* v3) {
#pragma omp parallel for
    for (int64_t n = 0; n < N; ++n) {
        float w[6]; int g;
        vel_gated_one(f, xt + 4 * n, NULL, w, v3 + 3 * n, &g);
    }
}

/* ------------------------------------------------------------------ a-6 integrate_pos (tensorf_keyframe.py:575-611) */
typedef struct {
    float x[3];      /* position at step start */
    float tcur, dt;
    float pmid[3];
    float w1[6], w2[6];
    int gate1, gate2, rejected;
    float z1[5 * HID], z2[5 * HID];
} rk_rec_t;

/* one RK2 midpoint step; rec may be NULL */
static inline void rk2_step(const orc_field_t* f, float* x, float tcur, float dt, rk_rec_t* rec) {
    float q[4] = {x[0], x[1], x[2], tcur}, v1[3], v2[3], w1[6], w2[6];
    int g1, g2;
    vel_gated_one(f, q, rec ? rec->z1 : NULL, w1, v1, &g1);
    float pm[4];
    for (int c = 0; c < 3; ++c) pm[c] = x[c] - 0.5f * dt * v1[c];
    pm[3] = tcur - 0.5f * dt;
    vel_gated_one(f, pm, rec ? rec->z2 : NULL, w2, v2, &g2);
    float xn[3];
    for (int c = 0; c < 3; ++c) xn[c] = x[c] - dt * v2[c];
    int rej = 0;
    if (f->gate_sur) rej = gated_out(f, xn); /* tensorf_keyframe.py:603-605 */
    if (rec) {
        memcpy(rec->x, x, 12); rec->tcur = tcur; rec->dt = dt; memcpy(rec->pmid, pm, 12);
        memcpy(rec->w1, w1, 24); memcpy(rec->w2, w2, 24);
        rec->gate1 = g1; rec->gate2 = g2; rec->rejected = rej;
    }
    if (!rej) memcpy(x, xn, 12);
}
/* returns number of steps; recs (capacity max_rec) optional */
static int integrate_one(const orc_field_t* f, float* x, float t, float base, rk_rec_t* recs, int max_rec) {
    float dtm = dt_max_of(f);
    float off = t - base, tcur = t;
    int n = 0;
    while (fabsf(off) > 0.f) {
        float a = fabsf(off) < dtm ? fabsf(off) : dtm;
        float dt = off > 0.f ? a : -a;
        rk2_step(f, x, tcur, dt, (recs && n < max_rec) ? &recs[n] : NULL);
        off = off - dt;
        tcur = tcur - dt;
        ++n;
        if (n > 4096) break;
    }
    return n;
}
void orc_integrate_pos(const orc_field_t* f, int64_t N, const float* x, const float* t, const float* base, float* xk) {
#pragma omp parallel for
    for (int64_t n = 0; n < N; ++n) {
        float p[3] = {x[3 * n], x[3 * n + 1], x[3 * n + 2]};
        integrate_one(f, p, t[n], base[n], NULL, 0);
        memcpy(xk + 3 * n, p, 12);
    }
}

/* ------------------------------------------------------------------ a-11 raw2alpha */
static inline void ray_weights(int S, const float* sigma, const float* dist, float* alpha, float* T, float* weight) {
    float Tc = 1.f;
    for (int j = 0; j < S; ++j) {
        float a = 1.f - expf(-sigma[j] * dist[j]);
        alpha[j] = a;
        if (T) T[j] = Tc;
        weight[j] = a * Tc;
        Tc = Tc * (1.f - a + 1e-10f);
    }
}
void orc_raw2alpha(int64_t R, int64_t S, const float* sigma, const float* dist, float* alpha, float* weight) {
    for (int64_t r = 0; r < R; ++r) ray_weights((int)S, sigma + r * S, dist + r * S, alpha + r * S, NULL, weight + r * S);
}

/* ------------------------------------------------------------------ a-13 MLPRender_PE (tensorf_base.py:67-98) */
static inline void render_input(const float* feat, const float* view, const float* pts, float* in) {
    memcpy(in, feat, 32 * sizeof(float));
    memcpy(in + 32, view, 12);
    memcpy(in + 35, pts, 12);
    /* positional_encoding (tensorf_model_utils.py:176-183): [sin(p (x) 2^k) (18) | cos (18)] */
    for (int c = 0; c < 3; ++c) {
        float fr = 1.f;
        for (int k = 0; k < 6; ++k) {
            float a = pts[c] * fr, b = view[c] * fr;
            in[38 + c * 6 + k] = sinf(a);
            in[56 + c * 6 + k] = cosf(a);
            in[74 + c * 6 + k] = sinf(b);
            in[92 + c * 6 + k] = cosf(b);
            fr *= 2.f;
        }
    }
}
static inline void render_mlp_fwd(const orc_field_t* f, const float* in, float* z1, float* z2, float* rgb) {
    float h[HID], zz[HID], o3[3];
    float* a = z1 ? z1 : zz;
    linear(f->rW[0], f->rb[0], HID, RIN, in, a);
    for (int i = 0; i < HID; ++i) h[i] = a[i] > 0.f ? a[i] : 0.f;
    float* b = z2 ? z2 : zz;
    linear(f->rW[1], f->rb[1], HID, HID, h, b);
    for (int i = 0; i < HID; ++i) h[i] = b[i] > 0.f ? b[i] : 0.f;
    linear(f->rW[2], f->rb[2], 3, HID, h, o3);
    for (int c = 0; c < 3; ++c) rgb[c] = sigmoidf_(o3[c]);
}
void orc_render_mlp(const orc_field_t* f, int64_t N, const float* pts, const float* view, const float* feat, float* rgb) {
#pragma omp parallel for
    for (int64_t n = 0; n < N; ++n) {
        float in[RIN];
        render_input(feat + 32 * n, view + 3 * n, pts + 3 * n, in);
        render_mlp_fwd(f, in, NULL, NULL, rgb + 3 * n);
    }
}

/* ------------------------------------------------------------------ a-15 AlphaGridMask.sample_alpha (trilinear) */
static inline float sample_alpha_one(const orc_field_t* f, const float* x) {
    int W = f->am_dims[0], H = f->am_dims[1], D = f->am_dims[2];
    float ix = (x[0] + 1.f) * ((float)(W - 1) / 2.f);
    float iy = (x[1] + 1.f) * ((float)(H - 1) / 2.f);
    float iz = (x[2] + 1.f) * ((float)(D - 1) / 2.f);
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    float wx = ix - fx, wy = iy - fy, wz = iz - fz;
    if (!(fx > -4.f)) fx = -4.f; if (fx > W + 2.f) fx = W + 2.f;
    if (!(fy > -4.f)) fy = -4.f; if (fy > H + 2.f) fy = H + 2.f;
    if (!(fz > -4.f)) fz = -4.f; if (fz > D + 2.f) fz = D + 2.f;
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    float s = 0.f;
    for (int dz = 0; dz < 2; ++dz) for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {
        int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
        if (xi < 0 || xi >= W || yi < 0 || yi >= H || zi < 0 || zi >= D) continue;
        float w = (dx ? wx : 1.f - wx) * (dy ? wy : 1.f - wy) * (dz ? wz : 1.f - wz);
        s += f->amask[((size_t)zi * H + yi) * W + xi] * w;
    }
    return s;
}
void orc_sample_alpha(const orc_field_t* f, int64_t N, const float* xyz, float* alpha) {
    for (int64_t n = 0; n < N; ++n) alpha[n] = sample_alpha_one(f, xyz + 3 * n);
}

/* degree-2 real SH bases at (un-normalised) direction d: models/sh.py:87-110 */
static inline void sh_bases9(const float* d, float* b) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    float x = d[0], y = d[1], z = d[2];
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[0] = C0; b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
    b[4] = C2[0] * xy; b[5] = C2[1] * yz; b[6] = C2[2] * (2.0f * zz - xx - yy); b[7] = C2[3] * xz; b[8] = C2[4] * (xx - yy);
}
/* SHRender inside render_pts (shadingMode "SH", tensorf_base.py:196-197; tensorf_model_utils.py:292-296): rgb = relu(sum_k sh_k feat[9c+k] + 0.5) */
static inline void sh_shade(const float* d, const float* feat27, float* rgb3) {
    float b[9];
    sh_bases9(d, b);
    for (int c = 0; c < 3; ++c) {
        float s = 0.f;
        for (int k = 0; k < 9; ++k) s += b[k] * feat27[9 * c + k];
        s += 0.5f;
        rgb3[c] = s > 0.f ? s : 0.f;
    }
}
/* ------------------------------------------------------------------ a-17 SHRender (tensorf_model_utils.py:292-296, sh.py:87-110) */
void orc_sh_render(int64_t N, const float* view, const float* ft, float* rgb) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    for (int64_t n = 0; n < N; ++n) {
        float x = view[3 * n], y = view[3 * n + 1], z = view[3 * n + 2], b[9];
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[0] = C0; b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
        b[4] = C2[0] * xy; b[5] = C2[1] * yz; b[6] = C2[2] * (2.0f * zz - xx - yy); b[7] = C2[3] * xz; b[8] = C2[4] * (xx - yy);
        for (int c = 0; c < 3; ++c) {
            float s = 0.f;
            for (int k = 0; k < 9; ++k) s += b[k] * ft[27 * n + 9 * c + k];
            s += 0.5f;
            rgb[3 * n + c] = s > 0.f ? s : 0.f;
        }
    }
}

/* ------------------------------------------------------------------ render forward / backward */
typedef struct {
    int nvalid, nwarp, nmask;
    int* vidx;        /* sample index j of each valid sample */
    rk_rec_t* rk;     /* nwarp * nsteps records */
    int* widx;        /* valid-slot -> warp-slot or -1 */
    int* midx;        /* sample j of each masked sample */
    float* min_;      /* nmask * RIN  MLP inputs */
    float* mz1; float* mz2; /* nmask * HID */
    float* mg;        /* nmask * Ca  plane products */
} ray_rec_t;

struct orc_ctx {
    orc_field_t f;
    int64_t R; int S; int flags; int nsteps; float t, base, tnorm_eval;
    float *o, *d;
    float *z, *xw, *xpre, *alpha, *T, *weight, *rgbs, *dist;
    uint8_t* valid;
    uint8_t* mask;
    float* rgb_pre; /* R*3 before clamp */
    ray_rec_t* rays;
};

orc_ctx_t* orc_render_fwd(const orc_field_t* f, int64_t R, const float* o, const float* d, const float* u,
                          float t, int flags, float* rgb, float* depth, float* acc, float* weight,
                          int64_t* counters, int keep_ctx) {
    int S = f->n_samples;
    int train = flags & ORC_TRAIN;
    orc_ctx_t* cx = (orc_ctx_t*)calloc(1, sizeof(orc_ctx_t));
    cx->f = *f; cx->R = R; cx->S = S; cx->flags = flags; cx->t = t;
    size_t RS = (size_t)R * S;
    cx->z = (float*)malloc(RS * 4); cx->xw = (float*)malloc(RS * 12); cx->xpre = (float*)malloc(RS * 4);
    cx->alpha = (float*)malloc(RS * 4); cx->T = (float*)malloc(RS * 4); cx->weight = (float*)malloc(RS * 4);
    cx->rgbs = (float*)calloc(RS * 3, 4); cx->dist = (float*)malloc(RS * 4);
    cx->valid = (uint8_t*)malloc(RS); cx->mask = (uint8_t*)malloc(RS);
    cx->rgb_pre = (float*)malloc((size_t)R * 12);
    cx->rays = (ray_rec_t*)calloc((size_t)R, sizeof(ray_rec_t));
    cx->o = (float*)malloc((size_t)R * 12); cx->d = (float*)malloc((size_t)R * 12);
    memcpy(cx->o, o, (size_t)R * 12); memcpy(cx->d, d, (size_t)R * 12);

    /* keyframe snap: one scalar decision per call (tensorf_keyframe.py:646-654, 683) */
    float base = (flags & ORC_TRANSFER) ? 0.f : snap_base(f, t);
    int key = is_close(t, base);
    cx->base = base;
    int warp = f->use_vel && !key;
    float tn_eval = f->use_vel ? normalize_time(f, base) : normalize_time(f, t);
    cx->tnorm_eval = tn_eval;
    /* number of RK2 steps is the same for every sample of the call */
    int nsteps = 0;
    if (warp) {
        float dtm = dt_max_of(f), off = t - base;
        while (fabsf(off) > 0.f && nsteps < 4096) {
            float a = fabsf(off) < dtm ? fabsf(off) : dtm;
            off = off - (off > 0.f ? a : -a);
            ++nsteps;
        }
    }
    cx->nsteps = nsteps;
    int inside = any_origin_inside(f, R, o);
    int64_t c_valid = 0, c_warp = 0, c_mask = 0;

#pragma omp parallel for schedule(dynamic, 4) reduction(+ : c_valid, c_warp, c_mask)
    for (int64_t r = 0; r < R; ++r) {
        ray_rec_t* rr = &cx->rays[r];
        float* z = cx->z + r * S; float* xw = cx->xw + (size_t)r * S * 3; float* xpre = cx->xpre + r * S;
        float* al = cx->alpha + r * S; float* T = cx->T + r * S; float* w = cx->weight + r * S;
        float* dist = cx->dist + r * S; uint8_t* va = cx->valid + r * S; uint8_t* mk = cx->mask + r * S;
        float sig[4096];
        float* sg = S <= 4096 ? sig : (float*)malloc(S * 4);
        float tmin = ray_tmin(f, inside, o + 3 * r, d + 3 * r);
        float ur = (train && u) ? u[r] : 0.f;
        int nv = 0;
        for (int j = 0; j < S; ++j) {
            float p[3];
            z[j] = sample_z(f, tmin, j, ur);
            int ok = sample_point(f, o + 3 * r, d + 3 * r, z[j], p);
            normalize_coord(f, p, xw + 3 * j);
            if (ok && f->has_amask && !train) ok = sample_alpha_one(f, xw + 3 * j) > 0.f; /* tensorf_keyframe.py:656-661 */
            va[j] = (uint8_t)ok; nv += ok;
        }
        for (int j = 0; j < S; ++j) dist[j] = (j + 1 < S ? z[j + 1] - z[j] : 0.f) * f->distance_scale;
        rr->nvalid = nv; rr->nwarp = warp ? nv : 0;
        if (keep_ctx && warp && nv) rr->rk = (rk_rec_t*)malloc(sizeof(rk_rec_t) * (size_t)nv * nsteps);
        int vi = 0;
        for (int j = 0; j < S; ++j) {
            sg[j] = 0.f; xpre[j] = 0.f;
            if (!va[j]) continue;
            if (warp) {
                float x[3] = {xw[3 * j], xw[3 * j + 1], xw[3 * j + 2]};
                integrate_one(f, x, t, base, rr->rk ? rr->rk + (size_t)vi * nsteps : NULL, nsteps);
                memcpy(xw + 3 * j, x, 12);
            }
            float q[4] = {xw[3 * j], xw[3 * j + 1], xw[3 * j + 2], tn_eval}, prod[MAXC];
            plane_products(f, f->dps, f->dpt, f->Cd, q, prod, NULL, NULL);
            float s = 0.f;
            for (int c = 0; c < f->Cd; ++c) s += prod[c];
            xpre[j] = s + f->density_shift;
            sg[j] = softplusf_(xpre[j]);
            ++vi;
        }
        ray_weights(S, sg, dist, al, T, w);
        int nm = 0;
        for (int j = 0; j < S; ++j) { mk[j] = w[j] > f->weight_thres; nm += mk[j]; }
        rr->nmask = nm;
        if (keep_ctx && nm) {
            rr->midx = (int*)malloc(sizeof(int) * nm);
            rr->min_ = (float*)malloc(sizeof(float) * (size_t)nm * RIN);
            rr->mz1 = (float*)malloc(sizeof(float) * (size_t)nm * HID);
            rr->mz2 = (float*)malloc(sizeof(float) * (size_t)nm * HID);
            rr->mg = (float*)malloc(sizeof(float) * (size_t)nm * f->Ca);
        }
        float a = 0.f, c3[3] = {0.f, 0.f, 0.f}, dp = 0.f;
        int mi = 0;
        for (int j = 0; j < S; ++j) {
            a += w[j];
            dp += w[j] * z[j];
            if (!mk[j]) continue;
            float q[4] = {xw[3 * j], xw[3 * j + 1], xw[3 * j + 2], tn_eval};
            float g[MAXC], feat[32], in[RIN], rgb3[3];
            app_feature_one(f, q, g, feat);
            if (f->shading == 1) {
                sh_shade(d + 3 * r, feat, rgb3);
                memset(in, 0, sizeof(in));
            } else {
                render_input(feat, d + 3 * r, xw + 3 * j, in);
                render_mlp_fwd(f, in, rr->mz1 ? rr->mz1 + (size_t)mi * HID : NULL, rr->mz2 ? rr->mz2 + (size_t)mi * HID : NULL, rgb3);
            }
            if (rr->midx) {
                rr->midx[mi] = j;
                memcpy(rr->min_ + (size_t)mi * RIN, in, sizeof(in));
                memcpy(rr->mg + (size_t)mi * f->Ca, g, sizeof(float) * f->Ca);
            }
            for (int c = 0; c < 3; ++c) { cx->rgbs[((size_t)r * S + j) * 3 + c] = rgb3[c]; c3[c] += w[j] * rgb3[c]; }
            ++mi;
        }
        for (int c = 0; c < 3; ++c) {
            float v = c3[c];
            if (flags & ORC_WHITE_BG) v = v + (1.f - a);
            cx->rgb_pre[3 * r + c] = v;
            rgb[3 * r + c] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        }
        acc[r] = a;
        depth[r] = dp + (1.f - a) * f->far_;
        if (weight) memcpy(weight + r * S, w, sizeof(float) * S);
        c_valid += nv; c_warp += rr->nwarp; c_mask += nm;
        if (sg != sig) free(sg);
    }
    if (counters) { counters[0] = c_valid; counters[1] = c_warp; counters[2] = c_mask; counters[3] = c_warp * 2 * nsteps; }
    if (!keep_ctx) { orc_ctx_free(cx); return NULL; }
    return cx;
}

void orc_ctx_free(orc_ctx_t* cx) {
    if (!cx) return;
    for (int64_t r = 0; r < cx->R; ++r) {
        ray_rec_t* rr = &cx->rays[r];
        free(rr->vidx); free(rr->rk); free(rr->widx); free(rr->midx); free(rr->min_); free(rr->mz1); free(rr->mz2); free(rr->mg);
    }
    free(cx->rays); free(cx->z); free(cx->xw); free(cx->xpre); free(cx->alpha); free(cx->T); free(cx->weight);
    free(cx->rgbs); free(cx->dist); free(cx->valid); free(cx->mask); free(cx->rgb_pre); free(cx->o); free(cx->d);
    free(cx);
}

/* thread-local MLP gradient accumulators */
typedef struct {
    float* rW[3]; float* rb[3]; float* vW[6]; float* vb[6]; float* basis;
} tl_grads_t;
static const int V_IN[6] = {ENC, HID, HID, HID, HID, HID};
static const int V_OUT[6] = {HID, HID, HID, HID, HID, 6};
static const int R_IN[3] = {RIN, HID, HID};
static const int R_OUT[3] = {HID, HID, 3};

static void tl_alloc(tl_grads_t* g, const orc_field_t* f) {
    for (int l = 0; l < 3; ++l) { g->rW[l] = (float*)calloc((size_t)R_IN[l] * R_OUT[l], 4); g->rb[l] = (float*)calloc(R_OUT[l], 4); }
    for (int l = 0; l < 6; ++l) { g->vW[l] = (float*)calloc((size_t)V_IN[l] * V_OUT[l], 4); g->vb[l] = (float*)calloc(V_OUT[l], 4); }
    g->basis = (float*)calloc((size_t)f->app_dim * f->Ca, 4);
}
static void tl_reduce_free(tl_grads_t* g, const orc_field_t* f, orc_grads_t* out) {
    for (int l = 0; l < 3; ++l) {
        if (out->rW[l]) for (int i = 0; i < R_IN[l] * R_OUT[l]; ++i) out->rW[l][i] += g->rW[l][i];
        if (out->rb[l]) for (int i = 0; i < R_OUT[l]; ++i) out->rb[l][i] += g->rb[l][i];
        free(g->rW[l]); free(g->rb[l]);
    }
    for (int l = 0; l < 6; ++l) {
        if (out->vW[l]) for (int i = 0; i < V_IN[l] * V_OUT[l]; ++i) out->vW[l][i] += g->vW[l][i];
        if (out->vb[l]) for (int i = 0; i < V_OUT[l]; ++i) out->vb[l][i] += g->vb[l][i];
        free(g->vW[l]); free(g->vb[l]);
    }
    if (out->basis) for (int i = 0; i < f->app_dim * f->Ca; ++i) out->basis[i] += g->basis[i];
    free(g->basis);
}

/* adjoint of v = gate * vel_from_w(w(q), q): gv -> network grads and gq[0..2] */
static inline void vel_eval_bwd(const orc_field_t* f, const float* q, const float* zs, const float* w, int gate,
                                const float* gv, tl_grads_t* tg, float* gq3) {
    if (gate) return;
    float gw[6];
    gw[0] = gv[0]; gw[1] = gv[1]; gw[2] = gv[2];
    gw[3] = q[2] * gv[1] - q[1] * gv[2];
    gw[4] = -q[2] * gv[0] + q[0] * gv[2];
    gw[5] = q[1] * gv[0] - q[0] * gv[1];
    /* explicit dependence of the rigid basis on x */
    gq3[0] += -w[5] * gv[1] + w[4] * gv[2];
    gq3[1] += w[5] * gv[0] - w[3] * gv[2];
    gq3[2] += -w[4] * gv[0] + w[3] * gv[1];
    float gq4[4] = {0.f, 0.f, 0.f, 0.f};
    velnet_bwd(f->vW, 1, q, zs, gw, tg->vW, tg->vb, gq4);
    gq3[0] += gq4[0]; gq3[1] += gq4[1]; gq3[2] += gq4[2];
}

void orc_render_bwd(orc_ctx_t* cx, const float* g_rgb, const float* g_depth, const float* g_acc,
                    const float* g_weight, orc_grads_t* grads) {
    const orc_field_t* f = &cx->f;
    int S = cx->S, nth = max_threads();
    tl_grads_t* tls = (tl_grads_t*)calloc(nth, sizeof(tl_grads_t));
    for (int i = 0; i < nth; ++i) tl_alloc(&tls[i], f);
    float* gps[3] = {grads->dps[0], grads->dps[1], grads->dps[2]};
    float* gpt[3] = {grads->dpt[0], grads->dpt[1], grads->dpt[2]};
    float* gas[3] = {grads->aps[0], grads->aps[1], grads->aps[2]};
    float* gat[3] = {grads->apt[0], grads->apt[1], grads->apt[2]};

#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t r = 0; r < cx->R; ++r) {
        tl_grads_t* tg = &tls[tid()];
        ray_rec_t* rr = &cx->rays[r];
        const float* z = cx->z + r * S; const float* xw = cx->xw + (size_t)r * S * 3;
        const float* al = cx->alpha + r * S; const float* T = cx->T + r * S; const float* w = cx->weight + r * S;
        const float* dist = cx->dist + r * S;
        float gr[3] = {0.f, 0.f, 0.f};
        for (int c = 0; c < 3; ++c) {
            float v = cx->rgb_pre[3 * r + c];
            gr[c] = (g_rgb && v >= 0.f && v <= 1.f) ? g_rgb[3 * r + c] : 0.f; /* clamp backward is inclusive */
        }
        float gd = g_depth ? g_depth[r] : 0.f, ga = g_acc ? g_acc[r] : 0.f;
        float bgsum = (cx->flags & ORC_WHITE_BG) ? (gr[0] + gr[1] + gr[2]) : 0.f;
        float* gw = (float*)malloc(sizeof(float) * S);
        float* gx = (float*)calloc((size_t)S * 3, sizeof(float)); /* grad wrt warped coordinates */
        for (int j = 0; j < S; ++j) {
            const float* c3 = cx->rgbs + ((size_t)r * S + j) * 3;
            gw[j] = gr[0] * c3[0] + gr[1] * c3[1] + gr[2] * c3[2] - bgsum + ga + gd * (z[j] - f->far_)
                    + (g_weight ? g_weight[r * S + j] : 0.f);
        }
        /* appearance branch */
        for (int mi = 0; mi < rr->nmask; ++mi) {
            int j = rr->midx[mi];
            const float* in = rr->min_ + (size_t)mi * RIN;
            const float* z1 = rr->mz1 + (size_t)mi * HID; const float* z2 = rr->mz2 + (size_t)mi * HID;
            const float* c3 = cx->rgbs + ((size_t)r * S + j) * 3;
            float go[3], gh2[HID], gz2[HID], gh1[HID], gz1[HID], gin[RIN], h1[HID], h2[HID];
            if (f->shading == 1) {
                /* SHRender backward: relu' (the stored colour is positive exactly where the pre-activation was), then the SH bases */
                float b9[9];
                sh_bases9(cx->d + 3 * r, b9);
                for (int k = 0; k < RIN; ++k) gin[k] = 0.f;
                for (int c = 0; c < 3; ++c) {
                    float gpre = c3[c] > 0.f ? w[j] * gr[c] : 0.f;
                    for (int k = 0; k < 9; ++k) gin[9 * c + k] = b9[k] * gpre;
                }
                goto basis_part;
            }
            for (int c = 0; c < 3; ++c) go[c] = w[j] * gr[c] * c3[c] * (1.f - c3[c]);
            for (int i = 0; i < HID; ++i) { h1[i] = z1[i] > 0.f ? z1[i] : 0.f; h2[i] = z2[i] > 0.f ? z2[i] : 0.f; gh2[i] = 0.f; gh1[i] = 0.f; }
            for (int o = 0; o < 3; ++o) {
                tg->rb[2][o] += go[o];
                for (int k = 0; k < HID; ++k) { tg->rW[2][o * HID + k] += go[o] * h2[k]; gh2[k] += f->rW[2][o * HID + k] * go[o]; }
            }
            for (int i = 0; i < HID; ++i) gz2[i] = z2[i] > 0.f ? gh2[i] : 0.f;
            for (int o = 0; o < HID; ++o) {
                float g = gz2[o]; if (g == 0.f) continue;
                tg->rb[1][o] += g;
                for (int k = 0; k < HID; ++k) { tg->rW[1][o * HID + k] += g * h1[k]; gh1[k] += f->rW[1][o * HID + k] * g; }
            }
            for (int i = 0; i < HID; ++i) gz1[i] = z1[i] > 0.f ? gh1[i] : 0.f;
            for (int k = 0; k < RIN; ++k) gin[k] = 0.f;
            for (int o = 0; o < HID; ++o) {
                float g = gz1[o]; if (g == 0.f) continue;
                tg->rb[0][o] += g;
                for (int k = 0; k < RIN; ++k) { tg->rW[0][o * RIN + k] += g * in[k]; gin[k] += f->rW[0][o * RIN + k] * g; }
            }
            /* pts (raw + PE) -> coordinate grads; view carries no grad */
            for (int c = 0; c < 3; ++c) {
                float s = gin[35 + c], fr = 1.f;
                for (int k = 0; k < 6; ++k) {
                    s += fr * (in[56 + c * 6 + k] * gin[38 + c * 6 + k] - in[38 + c * 6 + k] * gin[56 + c * 6 + k]);
                    fr *= 2.f;
                }
                gx[3 * j + c] += s;
            }
            /* feat = basis * g48 */
        basis_part:;
            const float* g48 = rr->mg + (size_t)mi * f->Ca;
            float gg[MAXC];
            for (int c = 0; c < f->Ca; ++c) gg[c] = 0.f;
            for (int o = 0; o < f->app_dim; ++o) {
                float g = gin[o];
                for (int c = 0; c < f->Ca; ++c) { tg->basis[o * f->Ca + c] += g * g48[c]; gg[c] += f->basis[o * f->Ca + c] * g; }
            }
            float q[4] = {xw[3 * j], xw[3 * j + 1], xw[3 * j + 2], cx->tnorm_eval};
            plane_products_bwd(f, f->aps, f->apt, gas[0] ? gas : NULL, gat[0] ? gat : NULL, f->Ca, q, gg, gx + 3 * j);
        }
        /* weights -> alpha -> sigma -> density feature */
        float suffix = 0.f; /* sum_{i>j} gw_i * w_i */
        int vi = rr->nvalid;
        for (int j = S - 1; j >= 0; --j) {
            float galpha = gw[j] * T[j] - suffix / (1.f - al[j] + 1e-10f);
            suffix += gw[j] * w[j];
            if (!cx->valid[r * S + j]) continue;
            --vi;
            float gsig = galpha * dist[j] * (1.f - al[j]);
            float xp = cx->xpre[r * S + j];
            float gf = gsig * (xp > 20.f ? 1.f : sigmoidf_(xp));
            float gp[MAXC];
            for (int c = 0; c < f->Cd; ++c) gp[c] = gf;
            float q[4] = {xw[3 * j], xw[3 * j + 1], xw[3 * j + 2], cx->tnorm_eval};
            plane_products_bwd(f, f->dps, f->dpt, gps[0] ? gps : NULL, gpt[0] ? gpt : NULL, f->Cd, q, gp, gx + 3 * j);
            /* RK2 adjoint (only when the sample was warped) */
            if (rr->rk) {
                float g3[3] = {gx[3 * j], gx[3 * j + 1], gx[3 * j + 2]};
                for (int s = cx->nsteps - 1; s >= 0; --s) {
                    rk_rec_t* rc = &rr->rk[(size_t)vi * cx->nsteps + s];
                    if (rc->rejected) continue; /* x_new = x: identity */
                    float gv2[3] = {-rc->dt * g3[0], -rc->dt * g3[1], -rc->dt * g3[2]};
                    float gpm[3] = {0.f, 0.f, 0.f};
                    float qm[4] = {rc->pmid[0], rc->pmid[1], rc->pmid[2], rc->tcur - 0.5f * rc->dt};
                    vel_eval_bwd(f, qm, rc->z2, rc->w2, rc->gate2, gv2, tg, gpm);
                    float gv1[3] = {-0.5f * rc->dt * gpm[0], -0.5f * rc->dt * gpm[1], -0.5f * rc->dt * gpm[2]};
                    float gx1[3] = {0.f, 0.f, 0.f};
                    float q1[4] = {rc->x[0], rc->x[1], rc->x[2], rc->tcur};
                    vel_eval_bwd(f, q1, rc->z1, rc->w1, rc->gate1, gv1, tg, gx1);
                    for (int c = 0; c < 3; ++c) g3[c] = g3[c] + gpm[c] + gx1[c];
                }
            }
        }
        free(gw); free(gx);
    }
    for (int i = 0; i < nth; ++i) tl_reduce_free(&tls[i], f, grads);
    free(tls);
}

/* ------------------------------------------------------------------ a-16 PDE regulariser (nvfi.py:42-84) */
/* forward-mode: value + 4 tangents through weight_net (SiLU). Stores z (5*HID) and zdot (4*5*HID). */
typedef struct {
    float e[ENC], ed[4][ENC];
    float z[5 * HID], zd[4][5 * HID];
    float w[6], wd[4][6];
} jet_t;

static void enc_tangents(const float* q, float* e, float ed[4][ENC]) {
    vel_encode(q, e);
    for (int j = 0; j < 4; ++j) for (int k = 0; k < ENC; ++k) ed[j][k] = 0.f;
    for (int c = 0; c < 4; ++c) {
        ed[c][c] = 1.f;
        float fr = 1.f;
        for (int k = 0; k < 3; ++k) {
            ed[c][4 + 8 * k + c] = fr * e[8 + 8 * k + c];
            ed[c][8 + 8 * k + c] = -fr * e[4 + 8 * k + c];
            fr *= 2.f;
        }
    }
}
static void jet_fwd(const float* const* W, const float* const* b, int act, const float* q, jet_t* J) {
    enc_tangents(q, J->e, J->ed);
    float h[HID], hd[4][HID];
    linear(W[0], b[0], HID, ENC, J->e, J->z);
    for (int j = 0; j < 4; ++j) linear(W[0], NULL, HID, ENC, J->ed[j], J->zd[j]);
    for (int l = 0; l < 5; ++l) {
        for (int i = 0; i < HID; ++i) {
            float zz = J->z[l * HID + i], d1 = act_d1(act, zz);
            h[i] = act_f(act, zz);
            for (int j = 0; j < 4; ++j) hd[j][i] = d1 * J->zd[j][l * HID + i];
        }
        if (l < 4) {
            linear(W[l + 1], b[l + 1], HID, HID, h, J->z + (l + 1) * HID);
            for (int j = 0; j < 4; ++j) linear(W[l + 1], NULL, HID, HID, hd[j], J->zd[j] + (l + 1) * HID);
        }
    }
    linear(W[5], b[5], 6, HID, h, J->w);
    for (int j = 0; j < 4; ++j) linear(W[5], NULL, 6, HID, hd[j], J->wd[j]);
}
/* reverse over the tangent program: adjoints gw (6) and gwd (4x6) -> parameter grads */
static void jet_bwd(const float* const* W, int act, const jet_t* J, const float* gw, float gwd[4][6],
                    float* const* gW, float* const* gb) {
    float gh[HID], ghd[4][HID], h[HID], hd[4][HID], gz[HID], gzd[4][HID];
    /* last layer */
    for (int i = 0; i < HID; ++i) {
        float zz = J->z[4 * HID + i], d1 = act_d1(act, zz);
        h[i] = act_f(act, zz);
        for (int j = 0; j < 4; ++j) hd[j][i] = d1 * J->zd[j][4 * HID + i];
        gh[i] = 0.f;
        for (int j = 0; j < 4; ++j) ghd[j][i] = 0.f;
    }
    for (int o = 0; o < 6; ++o) {
        gb[5][o] += gw[o];
        for (int k = 0; k < HID; ++k) {
            float acc = gw[o] * h[k];
            float wv = W[5][o * HID + k];
            gh[k] += wv * gw[o];
            for (int j = 0; j < 4; ++j) { acc += gwd[j][o] * hd[j][k]; ghd[j][k] += wv * gwd[j][o]; }
            gW[5][o * HID + k] += acc;
        }
    }
    for (int l = 4; l >= 0; --l) {
        for (int i = 0; i < HID; ++i) {
            float zz = J->z[l * HID + i], d1 = act_d1(act, zz), d2 = act_d2(act, zz);
            float g = d1 * gh[i];
            for (int j = 0; j < 4; ++j) {
                gzd[j][i] = d1 * ghd[j][i];
                g += d2 * J->zd[j][l * HID + i] * ghd[j][i];
            }
            gz[i] = g;
        }
        int in = l > 0 ? HID : ENC;
        const float* hp; const float* hdp[4];
        if (l > 0) {
            for (int i = 0; i < HID; ++i) {
                float zz = J->z[(l - 1) * HID + i], d1 = act_d1(act, zz);
                h[i] = act_f(act, zz);
                for (int j = 0; j < 4; ++j) hd[j][i] = d1 * J->zd[j][(l - 1) * HID + i];
                gh[i] = 0.f;
                for (int j = 0; j < 4; ++j) ghd[j][i] = 0.f;
            }
            hp = h; for (int j = 0; j < 4; ++j) hdp[j] = hd[j];
        } else {
            hp = J->e; for (int j = 0; j < 4; ++j) hdp[j] = J->ed[j];
        }
        for (int o = 0; o < HID; ++o) {
            gb[l][o] += gz[o];
            for (int k = 0; k < in; ++k) {
                float acc = gz[o] * hp[k];
                for (int j = 0; j < 4; ++j) acc += gzd[j][o] * hdp[j][k];
                gW[l][o * in + k] += acc;
                if (l > 0) {
                    float wv = W[l][o * in + k];
                    gh[k] += wv * gz[o];
                    for (int j = 0; j < 4; ++j) ghd[j][k] += wv * gzd[j][o];
                }
            }
        }
    }
}

float orc_pde_loss(const orc_field_t* f, int64_t P, const float* points, const float* t,
                   uint8_t* kept_out, int64_t* n_kept_out, int64_t n_jac, float* jac, orc_grads_t* grads,
                   int64_t* rk2_evals_out) {
    float* xn = (float*)malloc((size_t)P * 12);
    uint8_t* kept = (uint8_t*)malloc((size_t)P);
    int64_t nk = 0, evals = 0;
    /* occupancy prefilter (nvfi.py:50-64) */
#pragma omp parallel for reduction(+ : nk, evals)
    for (int64_t n = 0; n < P; ++n) {
        normalize_coord(f, points + 3 * n, xn + 3 * n);
        float base = snap_base(f, t[n]);
        float x[3] = {xn[3 * n], xn[3 * n + 1], xn[3 * n + 2]};
        int ns = integrate_one(f, x, t[n], base, NULL, 0);
        float q[4] = {x[0], x[1], x[2], normalize_time(f, base)}, prod[MAXC];
        plane_products(f, f->dps, f->dpt, f->Cd, q, prod, NULL, NULL);
        float s = 0.f;
        for (int c = 0; c < f->Cd; ++c) s += prod[c];
        float sigma = softplusf_(s + f->density_shift);
        float alpha = 1.f - expf(-sigma * 0.01f * 25.f);
        kept[n] = alpha >= f->alpha_thres;
        nk += kept[n]; evals += 2 * ns;
    }
    if (kept_out) memcpy(kept_out, kept, (size_t)P);
    if (n_kept_out) *n_kept_out = nk;
    if (rk2_evals_out) *rk2_evals_out = evals;
    if (nk == 0) { free(xn); free(kept); return 0.f; }
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * nk);
    { int64_t k = 0; for (int64_t n = 0; n < P; ++n) if (kept[n]) idx[k++] = n; }
    int nth = max_threads();
    float** tW = NULL; /* per-thread grads: [thread][net*6+l] */
    float** tB = NULL;
    if (grads) {
        tW = (float**)calloc((size_t)nth * 12, sizeof(float*));
        tB = (float**)calloc((size_t)nth * 12, sizeof(float*));
        for (int i = 0; i < nth * 12; ++i) {
            int l = i % 6;
            tW[i] = (float*)calloc((size_t)V_IN[l] * V_OUT[l], 4);
            tB[i] = (float*)calloc(V_OUT[l], 4);
        }
    }
    double sum_div = 0.0, sum_tr = 0.0;
    float inv_n = 1.f / (float)nk;
#pragma omp parallel for reduction(+ : sum_div, sum_tr)
    for (int64_t k = 0; k < nk; ++k) {
        int64_t n = idx[k];
        float q[4] = {xn[3 * n], xn[3 * n + 1], xn[3 * n + 2], t[n]};
        jet_t* J = (jet_t*)malloc(sizeof(jet_t));
        jet_fwd(f->vW, f->vb, 1, q, J);
        float aw[6], za[5 * HID], v[3], a[3], Jv[3][4];
        velnet_fwd(f->aW, f->ab, 0, q, za, aw);
        vel_from_w(J->w, q, v);
        acc_from_w(aw, q, a);
        const float x = q[0], y = q[1], z = q[2];
        for (int j = 0; j < 4; ++j) {
            const float* wd = J->wd[j];
            Jv[0][j] = wd[0] - wd[4] * z + wd[5] * y;
            Jv[1][j] = wd[1] + wd[3] * z - wd[5] * x;
            Jv[2][j] = wd[2] - wd[3] * y + wd[4] * x;
        }
        const float* w = J->w;
        Jv[0][2] += -w[4]; Jv[0][1] += w[5];
        Jv[1][2] += w[3];  Jv[1][0] += -w[5];
        Jv[2][1] += -w[3]; Jv[2][0] += w[4];
        if (jac && k < n_jac) {
            /* rows 0-2: velocity; rows 3-5 (acceleration Jacobian) are not used by the loss and left 0 */
            for (int c = 0; c < 3; ++c) for (int j = 0; j < 4; ++j) jac[(k * 6 + c) * 4 + j] = Jv[c][j];
            for (int c = 3; c < 6; ++c) for (int j = 0; j < 4; ++j) jac[(k * 6 + c) * 4 + j] = 0.f;
        }
        float div = Jv[0][0] + Jv[1][1] + Jv[2][2];
        float tr[3];
        for (int c = 0; c < 3; ++c) tr[c] = Jv[c][0] * v[0] + Jv[c][1] * v[1] + Jv[c][2] * v[2] + Jv[c][3] - a[c];
        sum_div += (double)(div * div);
        sum_tr += (double)(tr[0] * tr[0] + tr[1] * tr[1] + tr[2] * tr[2]);
        if (grads) {
            int th = tid();
            float gdiv = 10.f * div * inv_n;
            float gtr[3];
            for (int c = 0; c < 3; ++c) gtr[c] = 0.2f * tr[c] * inv_n / 3.f;
            float gJ[3][4], gv[3] = {0.f, 0.f, 0.f};
            for (int c = 0; c < 3; ++c) {
                for (int i = 0; i < 3; ++i) { gJ[c][i] = gtr[c] * v[i]; gv[i] += gtr[c] * Jv[c][i]; }
                gJ[c][3] = gtr[c];
                gJ[c][c] += gdiv;
            }
            float gw[6], gwd[4][6];
            gw[0] = gv[0]; gw[1] = gv[1]; gw[2] = gv[2];
            gw[3] = z * gv[1] - y * gv[2] + gJ[1][2] - gJ[2][1];
            gw[4] = -z * gv[0] + x * gv[2] - gJ[0][2] + gJ[2][0];
            gw[5] = y * gv[0] - x * gv[1] + gJ[0][1] - gJ[1][0];
            for (int j = 0; j < 4; ++j) {
                gwd[j][0] = gJ[0][j]; gwd[j][1] = gJ[1][j]; gwd[j][2] = gJ[2][j];
                gwd[j][3] = z * gJ[1][j] - y * gJ[2][j];
                gwd[j][4] = -z * gJ[0][j] + x * gJ[2][j];
                gwd[j][5] = y * gJ[0][j] - x * gJ[1][j];
            }
            jet_bwd(f->vW, 1, J, gw, gwd, &tW[th * 12], &tB[th * 12]);
            float ga[3] = {-gtr[0], -gtr[1], -gtr[2]}, gaw[6];
            gaw[0] = ga[0]; gaw[1] = ga[1]; gaw[2] = ga[2];
            gaw[3] = -y * ga[1] - z * ga[2];
            gaw[4] = -x * ga[0] - z * ga[2];
            gaw[5] = -x * ga[0] - y * ga[1];
            velnet_bwd(f->aW, 0, q, za, gaw, &tW[th * 12 + 6], &tB[th * 12 + 6], NULL);
        }
        free(J);
    }
    if (grads) {
        for (int th = 0; th < nth; ++th)
            for (int l = 0; l < 6; ++l) {
                int nW = V_IN[l] * V_OUT[l];
                if (grads->vW[l]) for (int i = 0; i < nW; ++i) grads->vW[l][i] += tW[th * 12 + l][i];
                if (grads->vb[l]) for (int i = 0; i < V_OUT[l]; ++i) grads->vb[l][i] += tB[th * 12 + l][i];
                if (grads->aW[l]) for (int i = 0; i < nW; ++i) grads->aW[l][i] += tW[th * 12 + 6 + l][i];
                if (grads->ab[l]) for (int i = 0; i < V_OUT[l]; ++i) grads->ab[l][i] += tB[th * 12 + 6 + l][i];
            }
        for (int i = 0; i < nth * 12; ++i) { free(tW[i]); free(tB[i]); }
        free(tW); free(tB);
    }
    free(idx); free(xn); free(kept);
    return (float)(5.0 * sum_div / (double)nk + 0.1 * sum_tr / (3.0 * (double)nk));
}

/* ------------------------------------------------------------------ f-1 regularisers */
/* density_L1 (tensorf_keyframe.py:188-203) */
float orc_density_L1(const orc_field_t* f) {
    double tot = 0.0;
    for (int i = 0; i < 3; ++i) {
        size_t ns = (size_t)f->Cd * f->G[MS_B[i]] * f->G[MS_A[i]], nt = (size_t)f->Cd * f->K * f->G[MT_C[i]];
        double a = 0.0, b = 0.0;
        for (size_t k = 0; k < ns; ++k) a += fabsf(f->dps[i][k]);
        for (size_t k = 0; k < nt; ++k) b += fabsf(1.f - f->dpt[i][k]);
        tot += a / (double)ns + b / (double)nt;
    }
    return (float)tot;
}
/* utils.tensorf_utils.TVLoss (utils/tensorf_utils.py:139-158) */
static double tv_plane(const float* p, int C, int H, int W, int tflag) {
    double h = 0.0, w = 0.0;
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float v = p[((size_t)c * H + y) * W + x];
                if (y + 1 < H) { float dd = p[((size_t)c * H + y + 1) * W + x] - v; h += (double)(dd * dd); }
                if (x + 1 < W) { float dd = p[((size_t)c * H + y) * W + x + 1] - v; w += (double)(dd * dd); }
            }
    if (tflag) h *= 3.0;
    double ch = (double)C * (H - 1) * W, cw = (double)C * H * (W - 1);
    return 2.0 * (h / ch + w / cw);
}
float orc_tv_density(const orc_field_t* f) {
    double tot = 0.0;
    for (int i = 0; i < 3; ++i) {
        tot += tv_plane(f->dps[i], f->Cd, f->G[MS_B[i]], f->G[MS_A[i]], 0) * 1e-2;
        if (f->K > 1) tot += tv_plane(f->dpt[i], f->Cd, f->K, f->G[MT_C[i]], 1) * 1e-2;
    }
    return (float)tot;
}
float orc_tv_app(const orc_field_t* f) {
    double tot = 0.0;
    for (int i = 0; i < 3; ++i) tot += tv_plane(f->aps[i], f->Ca, f->G[MS_B[i]], f->G[MS_A[i]], 0) * 1e-2;
    return (float)tot;
}

/* ---------------------------------------------------------------- MaskField forward / backward (config 5)
 * models/mask_field.py:68-83 with skips=[] and point_embed=False: h = relu(point_fc[l](h)) for l = 0..3,
 * mask = softmax(mask_fc(h), dim=1).  Backward: d logits = p * (g - sum_k g_k p_k) (softmax), then Linear / ReLU adjoints. */
void orc_maskfield(const float* const* W, const float* const* b, int mask_dim, int64_t N, const float* xyz, float* mask,
                   const float* g_mask, float* const* gW, float* const* gb) {
    const int H = 128, K = mask_dim;
    for (int64_t n = 0; n < N; ++n) {
        float h[5][128], z5[32], p[32];
        for (int c = 0; c < 3; ++c) h[0][c] = xyz[3 * n + c];
        for (int l = 0; l < 4; ++l) {
            const int in = l == 0 ? 3 : H;
            for (int o = 0; o < H; ++o) {
                float acc = b[l][o];
                for (int k = 0; k < in; ++k) acc += W[l][(size_t)o * in + k] * h[l][k];
                h[l + 1][o] = acc > 0.f ? acc : 0.f;
            }
        }
        float mx = -INFINITY;
        for (int o = 0; o < K; ++o) {
            float acc = b[4][o];
            for (int k = 0; k < H; ++k) acc += W[4][(size_t)o * H + k] * h[4][k];
            z5[o] = acc; if (acc > mx) mx = acc;
        }
        float sum = 0.f;
        for (int o = 0; o < K; ++o) { p[o] = expf(z5[o] - mx); sum += p[o]; }
        for (int o = 0; o < K; ++o) { p[o] = p[o] / sum; mask[(size_t)n * K + o] = p[o]; }
        if (!g_mask) continue;
        float gz[128], gh[128], dot = 0.f;
        for (int o = 0; o < K; ++o) dot += g_mask[(size_t)n * K + o] * p[o];
        for (int o = 0; o < K; ++o) gz[o] = p[o] * (g_mask[(size_t)n * K + o] - dot);
        for (int k = 0; k < H; ++k) gh[k] = 0.f;
        for (int o = 0; o < K; ++o) {
            if (gb && gb[4]) gb[4][o] += gz[o];
            for (int k = 0; k < H; ++k) {
                if (gW && gW[4]) gW[4][(size_t)o * H + k] += gz[o] * h[4][k];
                gh[k] += W[4][(size_t)o * H + k] * gz[o];
            }
        }
        for (int l = 3; l >= 0; --l) {
            const int in = l == 0 ? 3 : H;
            for (int o = 0; o < H; ++o) gz[o] = h[l + 1][o] > 0.f ? gh[o] : 0.f;
            for (int k = 0; k < in; ++k) gh[k] = 0.f;
            for (int o = 0; o < H; ++o) {
                if (gz[o] == 0.f) continue;
                if (gb && gb[l]) gb[l][o] += gz[o];
                for (int k = 0; k < in; ++k) {
                    if (gW && gW[l]) gW[l][(size_t)o * in + k] += gz[o] * h[l][k];
                    gh[k] += W[l][(size_t)o * in + k] * gz[o];
                }
            }
        }
    }
}
