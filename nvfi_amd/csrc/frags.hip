// frags.hip - nvfi_frag_cache_bytes / nvfi_pack_frags: every fragment set of a field in one launch (frags.h).
// Reference counterpart: none - the reference's nn.Linear weights are read by ATen as they are (models/velocity_field.py:60-67,
// models/tensorf_base.py:67-98); this is the MFMA operand layout of those weights.
#include <string.h>
#include "common.h"
#include "frags.h"

extern "C" int nvfi_frag_cache_bytes(const nvfi_field_desc* f, int64_t* bytes) {
    (void)f;
    FragCache c; frag_cache_layout(nullptr, &c);
    *bytes = c.total;
    return 0;
}

extern "C" int nvfi_pack_frags(const nvfi_field_desc* f, void* cache, int64_t cache_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!cache) return nvfi_fail(2, "nvfi_pack_frags: cache is NULL");
    FragCache c; frag_cache_layout((const float*)cache, &c);
    if (c.total > cache_bytes) return nvfi_fail(4, "fragment cache too small: need %lld bytes, got %lld", (long long)c.total, (long long)cache_bytes);
    // the plain fragment sets, through the job builders of the per-call path (same layouts, same kernels' index maps) ...
    PackJobs tmp; tmp.n = 0;
    PackJobsAll all; memset(&all, 0, sizeof(all));
    auto take = [&]() { for (int i = 0; i < tmp.n; ++i) all.j[all.n++] = tmp.j[i]; tmp.n = 0; };
    RenderFrags RW; VelFrags VW, AW;
    memset(&VW, 0, sizeof(VW)); memset(&AW, 0, sizeof(AW));
    if (pack_render_frags(f, c.render, &RW, &tmp)) return 3;
    take();
    if (f->use_vel) {
        if (pack_vel_frags(f->vW, f->vb, c.vel, &VW, &tmp)) return 3;
        take();
        if (pack_vel_frags(f->aW, f->ab, c.anet, &AW, &tmp)) return 3;
        take();
        // ... and the x4 copies straight from the weights: the job of the plain fragment with x4 = 1 and the x4 destination
        auto find = [&](const float* frag) -> const PackJob* { for (int i = 0; i < all.n; ++i) if (all.j[i].frag == frag && !all.j[i].x4) return &all.j[i]; return nullptr; };
        auto add_x4 = [&](const float* frag, const float4* dst) -> int {
            const PackJob* src = find(frag);
            if (!src || all.n >= MAX_PACK_JOBS_ALL) return 1;
            PackJob J = *src;
            J.frag = reinterpret_cast<float*>(const_cast<float4*>(dst)); J.bfrag = nullptr; J.b = nullptr; J.x4 = 1;
            all.j[all.n++] = J;
            return 0;
        };
        const float4* f4[6]; const float4* t4[6]; const float4* ta4[6];
        x4f_pointers(c.vel_x4f, f4); x4b_pointers(c.vel_x4b, t4); a_x4b_pointers(c.a_x4b, ta4);
        int rc = 0;
        for (int l = 0; l < 6; ++l) rc |= add_x4(VW.f[l], f4[l]);
        for (int l = 0; l < 6; ++l) rc |= add_x4(VW.t[l], t4[l]);
        for (int l = 1; l < 6; ++l) rc |= add_x4(AW.t[l], ta4[l]);
        if (rc) return nvfi_fail(3, "nvfi_pack_frags: job table");
    }
    X6PackArgs x6; memset(&x6, 0, sizeof(x6));
    if (f->use_vel) { for (int l = 0; l < 5; ++l) x6.W[l] = f->vW[l]; x6.img = reinterpret_cast<b8_t*>(c.vel_x6); x6.imgT = reinterpret_cast<b8_t*>(c.vel_x6t); }
    return launch_pack_all(all, f->use_vel ? &x6 : nullptr, st);
}
