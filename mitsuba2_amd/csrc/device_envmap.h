// device_envmap.h -- `envmap` emitter, RGB and spectral variants.
//
//   Hierarchical2D<Float, 0>::sample / eval   include/mitsuba/core/distr_2d.h:320-400, :486-517
//   square_to_bilinear / interval_to_linear   include/mitsuba/core/warp.h:367-441
//   EnvironmentMapEmitter                     src/emitters/envmap.cpp:132-208, :270-315
// The MIP hierarchy is built on the host (envmap.cpp) with the reference's memory layout: 2x2 blocks contiguous.
#pragma once
#include "device_math.h"

namespace mtsamd {

constexpr int kEnvMaxLevels = 34;
struct DevEnvmap {
    const float4 *data;                 // RGBA texels (alpha = 1), row-major; spectral variant: (srgb model coefficients, scale)
    const float *warp;                  // every level of the hierarchy, concatenated
    int32_t w, h, n_levels; float scale;
    uint32_t lv_offset[kEnvMaxLevels], lv_width[kEnvMaxLevels];
    float patch_size[2], inv_patch_size[2];
    uint32_t max_patch_index[2];
    float to_world[9], to_local[9];     // linear part of the emitter's to_world and its inverse
};

MTS_DEV float lerp_e(float a, float b, float t) { return fmaf(b, t, fmaf(-a, t, a)); }      // enoki::lerp
MTS_DEV float interval_to_linear(float v0, float v1, float sample) {
    if (fabsf(v0 - v1) > 1e-4f * (v0 + v1)) return (v0 - safe_sqrt(lerp_e(v0 * v0, v1 * v1, sample))) / (v0 - v1);
    return sample;
}
MTS_DEV uint32_t env_level_index(uint32_t width, uint32_t x, uint32_t y) {
    return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * width);
}
MTS_DEV float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

MTS_DEV void hier2d_sample(const DevEnvmap &e, float sx, float sy, float &ox, float &oy, float &pdf) {
    sx = clamp01(sx); sy = clamp01(sy);
    uint32_t offx = 0, offy = 0;
    for (int l = e.n_levels - 2; l > 0; --l) {
        const float *lv = e.warp + e.lv_offset[l];
        offx <<= 1; offy <<= 1;
        const uint32_t oi = env_level_index(e.lv_width[l], offx, offy);
        const float v00 = lv[oi], v10 = lv[oi + 1], v01 = lv[oi + 2], v11 = lv[oi + 3];
        sx = clamp01(sx); sy = clamp01(sy);
        const float r0 = v00 + v10, r1 = v01 + v11;
        sy *= r0 + r1;
        bool mask = sy > r0;
        if (mask) { offy += 1u; sy -= r0; }
        sy /= mask ? r1 : r0;
        const float c0 = mask ? v01 : v00, c1 = mask ? v11 : v10;
        sx *= c0 + c1;
        mask = sx > c0;
        if (mask) sx -= c0;
        sx /= mask ? c1 : c0;
        if (mask) offx += 1u;
    }
    const float *l0 = e.warp + e.lv_offset[0];
    const uint32_t w0 = e.lv_width[0], oi = offx + offy * w0;
    const float v00 = l0[oi], v10 = l0[oi + 1], v01 = l0[oi + w0], v11 = l0[oi + w0 + 1];
    const float r0 = v00 + v10, r1 = v01 + v11;                        // square_to_bilinear (warp.h:398-414)
    sy = interval_to_linear(r0, r1, sy);
    const float c0 = lerp_e(v00, v01, sy), c1 = lerp_e(v10, v11, sy);
    sx = interval_to_linear(c0, c1, sx);
    pdf = lerp_e(c0, c1, sx);
    ox = ((float) (int32_t) offx + sx) * e.patch_size[0];
    oy = ((float) (int32_t) offy + sy) * e.patch_size[1];
}

MTS_DEV float hier2d_eval(const DevEnvmap &e, float px, float py) {
    px = clamp01(px) * e.inv_patch_size[0]; py = clamp01(py) * e.inv_patch_size[1];
    const uint32_t ox = min((uint32_t) (int32_t) px, e.max_patch_index[0]), oy = min((uint32_t) (int32_t) py, e.max_patch_index[1]);
    px -= (float) (int32_t) ox; py -= (float) (int32_t) oy;
    const float *l0 = e.warp + e.lv_offset[0];
    const uint32_t w0 = e.lv_width[0], oi = ox + oy * w0;
    const float v00 = l0[oi], v10 = l0[oi + 1], v01 = l0[oi + w0], v11 = l0[oi + w0 + 1];
    return lerp_e(lerp_e(v00, v10, px), lerp_e(v01, v11, px), py);
}

MTS_DEV f3 mat3_apply(const float *m, f3 v) {
    return mk3(fmaf(m[2], v.z, fmaf(m[1], v.y, m[0] * v.x)), fmaf(m[5], v.z, fmaf(m[4], v.y, m[3] * v.x)),
               fmaf(m[8], v.z, fmaf(m[7], v.y, m[6] * v.x)));
}

// eval_spectrum, RGB branch (envmap.cpp:270-312)
MTS_DEV f3 envmap_lookup(const DevEnvmap &e, float u, float v) {
    u *= (float) (e.w - 1); v *= (float) (e.h - 1);
    const uint32_t px = min((uint32_t) u, (uint32_t) (e.w - 2)), py = min((uint32_t) v, (uint32_t) (e.h - 2));
    const float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const float4 *p = e.data + (size_t) py * e.w + px;
    const float4 v00 = p[0], v10 = p[1], v01 = p[e.w], v11 = p[e.w + 1];
    f3 r;
    { const float a = fmaf(w0x, v00.x, w1x * v10.x), b = fmaf(w0x, v01.x, w1x * v11.x); r.x = fmaf(w0y, a, w1y * b) * e.scale; }
    { const float a = fmaf(w0x, v00.y, w1x * v10.y), b = fmaf(w0x, v01.y, w1x * v11.y); r.y = fmaf(w0y, a, w1y * b) * e.scale; }
    { const float a = fmaf(w0x, v00.z, w1x * v10.z), b = fmaf(w0x, v01.z, w1x * v11.z); r.z = fmaf(w0y, a, w1y * b) * e.scale; }
    return r;
}
MTS_DEV void env_dir_to_uv(f3 v, float &u, float &vv) {            // envmap.cpp:139-142
    const float a = lm_atan2(v.x, -v.z) * (0.5f * kInvPi);
    const float b = lm_acos(fminf(fmaxf(v.y, -1.0f), 1.0f)) * kInvPi;
    u = a - floorf(a); vv = b - floorf(b);
}
// EnvironmentMapEmitter::eval for the world-space direction the ray travels in (si.wi = -d)
MTS_DEV f3 envmap_eval(const DevEnvmap &e, f3 d) {
    float u, v;
    env_dir_to_uv(mat3_apply(e.to_local, d), u, v);
    return envmap_lookup(e, u, v);
}
// sample_direction (envmap.cpp:154-190): world direction, pdf and the texture coordinates the radiance is looked up at
MTS_DEV void envmap_sample(const DevEnvmap &e, f2 sample, f3 &d_out, float &pdf_out, f2 &uv) {
    float u, v, pdf;
    hier2d_sample(e, sample.x, sample.y, u, v, pdf);
    const float theta = v * kPi, phi = u * (2.0f * kPi);
    const float st = lm_sin(theta), ct = lm_cos(theta), sp = lm_sin(phi), cp = lm_cos(phi);
    const f3 sd = mk3(cp * st, sp * st, ct);
    f3 d = mk3(sd.y, sd.z, -sd.x);
    const float inv_sin_theta = 1.0f / sqrtf(fmaxf(d.x * d.x + d.z * d.z, kEpsilon * kEpsilon));
    d = mat3_apply(e.to_world, d);
    pdf_out = pdf > 0.0f ? pdf * inv_sin_theta * (1.0f / (2.0f * (kPi * kPi))) : 0.0f;
    uv.x = u; uv.y = v;
    d_out = d;
}
MTS_DEV float envmap_pdf(const DevEnvmap &e, f3 d_world) {           // envmap.cpp:192-208
    const f3 d = mat3_apply(e.to_local, d_world);
    float u, v;
    env_dir_to_uv(d, u, v);
    const float inv_sin_theta = 1.0f / sqrtf(fmaxf(d.x * d.x + d.z * d.z, kEpsilon * kEpsilon));
    return hier2d_eval(e, u, v) * inv_sin_theta * (1.0f / (2.0f * (kPi * kPi)));
}

} // namespace mtsamd
