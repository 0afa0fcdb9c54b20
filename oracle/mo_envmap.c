/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_math.h header for scope and pinning).
 *
 * `envmap` emitter, RGB variant:
 *   Hierarchical2D<Float, 0>     include/mitsuba/core/distr_2d.h:200-312 (construction), :320-400 (sample), :486-517 (eval)
 *   square_to_bilinear           include/mitsuba/core/warp.h:367-441
 *   EnvironmentMapEmitter        src/emitters/envmap.cpp:66-227, :270-315
 * Pinned by the reference's spot checks of Hierarchical2D0 against Mathematica (src/libcore/tests/test_distr_2d.py:8-60).
 */
#include <stdlib.h>
#include "mo_internal.h"

static inline float lerpf(float a, float b, float t) { return fmaf(b, t, fmaf(-a, t, a)); }      /* enoki::lerp */

/* warp.h:367-373 */
static inline float interval_to_linear(float v0, float v1, float sample) {
    if (fabsf(v0 - v1) > 1e-4f * (v0 + v1))
        return (v0 - mo_safe_sqrt(lerpf(v0 * v0, v1 * v1, sample))) / (v0 - v1);
    return sample;
}
/* warp.h:377-383 */
static inline float linear_to_interval(float v0, float v1, float sample) {
    if (fabsf(v0 - v1) > 1e-4f * (v0 + v1))
        return sample * ((2.0f - sample) * v0 + sample * v1) / (v0 + v1);
    return sample;
}
/* warp.h:398-414 */
static void square_to_bilinear(float v00, float v10, float v01, float v11, float *sx, float *sy, float *pdf) {
    float r0 = v00 + v10, r1 = v01 + v11;
    *sy = interval_to_linear(r0, r1, *sy);
    float c0 = lerpf(v00, v01, *sy), c1 = lerpf(v10, v11, *sy);
    *sx = interval_to_linear(c0, c1, *sx);
    *pdf = lerpf(c0, c1, *sx);
}
/* warp.h:417-433 */
static void bilinear_to_square(float v00, float v10, float v01, float v11, float *sx, float *sy, float *pdf) {
    float r0 = v00 + v10, r1 = v01 + v11, c0 = lerpf(v00, v01, *sy), c1 = lerpf(v10, v11, *sy);
    *pdf = lerpf(c0, c1, *sx);
    *sx = linear_to_interval(c0, c1, *sx);
    *sy = linear_to_interval(r0, r1, *sy);
}

static inline uint32_t level_index(const mo_h2_level *l, uint32_t x, uint32_t y) {     /* distr_2d.h:576-580: 2x2 blocks contiguous */
    return ((x & 1u) | (((x & ~1u) | (y & 1u)) << 1)) + ((y & ~1u) * l->width);
}
static uint32_t log2i_ceil(uint32_t v) { uint32_t r = 0; while ((1u << r) < v) ++r; return r; }

void mo_hier2d_free(mo_hier2d *h) {
    for (int i = 0; i < h->n_levels; ++i) free(h->lv[i].data);
    memset(h, 0, sizeof(*h));
}

/* Hierarchical2D(data, size, normalize = true) (distr_2d.h:200-312), no conditional parameters */
int mo_hier2d_build(mo_hier2d *h, const float *data, uint32_t w, uint32_t hgt, int normalize) {
    memset(h, 0, sizeof(*h));
    if (w < 2 || hgt < 2) return -1;
    uint32_t npx = w - 1, npy = hgt - 1;
    h->patch_size[0] = 1.0f / (float) npx; h->patch_size[1] = 1.0f / (float) npy;
    h->inv_patch_size[0] = (float) npx; h->inv_patch_size[1] = (float) npy;
    h->max_patch_index[0] = npx - 1; h->max_patch_index[1] = npy - 1;
    uint32_t max_level = log2i_ceil(npx > npy ? npx : npy);
    h->n_levels = (int) max_level + 2;
    h->lv[0].width = w; h->lv[0].size = w * hgt;
    h->lv[0].data = (float *) calloc(h->lv[0].size, sizeof(float));
    uint32_t lx = npx, ly = npy;
    for (int level = (int) max_level, k = 1; level >= 0; --level, ++k) {
        lx += lx & 1u; ly += ly & 1u;
        h->lv[k].width = lx; h->lv[k].size = lx * ly;
        h->lv[k].data = (float *) calloc(h->lv[k].size, sizeof(float));
        lx >>= 1; ly >>= 1;
    }
    const float *in = data;
    double sum = 0.0;
    for (uint32_t y = 0; y < npy; ++y) {
        for (uint32_t x = 0; x < npx; ++x) {
            float avg = (in[0] + in[1] + in[w] + in[w + 1]) * 0.25f;
            sum += (double) avg;
            h->lv[1].data[level_index(&h->lv[1], x, y)] = avg;
            ++in;
        }
        ++in;
    }
    float scale = normalize ? (float) ((double) (npx * npy) / sum) : 1.0f;
    for (uint32_t i = 0; i < h->lv[0].size; ++i) h->lv[0].data[i] = data[i] * scale;
    for (uint32_t i = 0; i < h->lv[1].size; ++i) h->lv[1].data[i] *= scale;
    lx = npx; ly = npy;
    for (uint32_t level = 2; level <= max_level + 1; ++level) {
        const mo_h2_level *l0 = &h->lv[level - 1];
        mo_h2_level *l1 = &h->lv[level];
        lx = (lx + 1u) >> 1; ly = (ly + 1u) >> 1;
        for (uint32_t y = 0; y < ly; ++y)
            for (uint32_t x = 0; x < lx; ++x) {
                const float *d0 = l0->data + level_index(l0, x * 2, y * 2);
                l1->data[level_index(l1, x, y)] = d0[0] + d0[1] + d0[2] + d0[3];
            }
    }
    return 0;
}

static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

/* Hierarchical2D::sample (distr_2d.h:320-400) */
void mo_hier2d_sample(const mo_hier2d *h, float sx, float sy, float *ox, float *oy, float *pdf) {
    sx = clamp01(sx); sy = clamp01(sy);
    uint32_t offx = 0, offy = 0;
    for (int l = h->n_levels - 2; l > 0; --l) {
        const mo_h2_level *lv = &h->lv[l];
        offx <<= 1; offy <<= 1;
        uint32_t oi = level_index(lv, offx, offy);
        float v00 = lv->data[oi], v10 = lv->data[oi + 1], v01 = lv->data[oi + 2], v11 = lv->data[oi + 3];
        sx = clamp01(sx); sy = clamp01(sy);
        float r0 = v00 + v10, r1 = v01 + v11;
        sy *= r0 + r1;
        int mask = sy > r0;
        if (mask) { offy += 1u; sy -= r0; }
        sy /= mask ? r1 : r0;
        float c0 = mask ? v01 : v00, c1 = mask ? v11 : v10;
        sx *= c0 + c1;
        mask = sx > c0;
        if (mask) sx -= c0;
        sx /= mask ? c1 : c0;
        if (mask) offx += 1u;
    }
    const mo_h2_level *l0 = &h->lv[0];
    uint32_t oi = offx + offy * l0->width;
    float v00 = l0->data[oi], v10 = l0->data[oi + 1], v01 = l0->data[oi + l0->width], v11 = l0->data[oi + l0->width + 1];
    square_to_bilinear(v00, v10, v01, v11, &sx, &sy, pdf);
    *ox = ((float) (int32_t) offx + sx) * h->patch_size[0];
    *oy = ((float) (int32_t) offy + sy) * h->patch_size[1];
}

/* Hierarchical2D::eval (distr_2d.h:486-517) */
float mo_hier2d_eval(const mo_hier2d *h, float px, float py) {
    px = clamp01(px) * h->inv_patch_size[0]; py = clamp01(py) * h->inv_patch_size[1];
    uint32_t ox = (uint32_t) (int32_t) px, oy = (uint32_t) (int32_t) py;
    if (ox > h->max_patch_index[0]) ox = h->max_patch_index[0];
    if (oy > h->max_patch_index[1]) oy = h->max_patch_index[1];
    px -= (float) (int32_t) ox; py -= (float) (int32_t) oy;
    const mo_h2_level *l0 = &h->lv[0];
    uint32_t oi = ox + oy * l0->width;
    float v00 = l0->data[oi], v10 = l0->data[oi + 1], v01 = l0->data[oi + l0->width], v11 = l0->data[oi + l0->width + 1];
    return lerpf(lerpf(v00, v10, px), lerpf(v01, v11, px), py);           /* square_to_bilinear_pdf (warp.h:435-441) */
}

/* Hierarchical2D::invert (distr_2d.h:403-484) */
void mo_hier2d_invert(const mo_hier2d *h, float sx, float sy, float *ox, float *oy, float *pdf) {
    sx = clamp01(sx) * h->inv_patch_size[0]; sy = clamp01(sy) * h->inv_patch_size[1];
    uint32_t offx = (uint32_t) (int32_t) sx, offy = (uint32_t) (int32_t) sy;
    if (offx > h->max_patch_index[0]) offx = h->max_patch_index[0];
    if (offy > h->max_patch_index[1]) offy = h->max_patch_index[1];
    const mo_h2_level *l0 = &h->lv[0];
    uint32_t oi = offx + offy * l0->width;
    float v00 = l0->data[oi], v10 = l0->data[oi + 1], v01 = l0->data[oi + l0->width], v11 = l0->data[oi + l0->width + 1];
    sx -= (float) (int32_t) offx; sy -= (float) (int32_t) offy;
    bilinear_to_square(v00, v10, v01, v11, &sx, &sy, pdf);
    for (int l = 1; l < h->n_levels - 1; ++l) {
        const mo_h2_level *lv = &h->lv[l];
        oi = level_index(lv, offx & ~1u, offy & ~1u);
        v00 = lv->data[oi]; v10 = lv->data[oi + 1]; v01 = lv->data[oi + 2]; v11 = lv->data[oi + 3];
        int xm = (offx & 1u) != 0, ym = (offy & 1u) != 0;
        float r0 = v00 + v10, r1 = v01 + v11, c0 = ym ? v01 : v00, c1 = ym ? v11 : v10;
        sy *= ym ? r1 : r0;
        if (ym) sy += r0;
        sy /= r0 + r1;
        sx *= xm ? c1 : c0;
        if (xm) sx += c0;
        sx /= c0 + c1;
        sx = clamp01(sx); sy = clamp01(sy);
        offx >>= 1; offy >>= 1;
    }
    *ox = sx; *oy = sy;
}

/* ------------------------------------------------------------------ EnvironmentMapEmitter (RGB variant) */
static inline float luminance(const float *rgb) { return rgb[0] * 0.212671f + rgb[1] * 0.715160f + rgb[2] * 0.072169f; }

/* envmap.cpp:66-125: RGBA texels (alpha = 1) and the luminance * sin(theta) image the warp is built from */
int mo_envmap_init(mo_envmap *e, int w, int h, const float *rgb, float scale, const float *to_world9) {
    memset(e, 0, sizeof(*e));
    if (w < 2 || h < 2) return -1;
    e->w = w; e->h = h; e->scale = scale;
    e->data = (float *) malloc(sizeof(float) * 4 * (size_t) w * h);
    float *lum = (float *) malloc(sizeof(float) * (size_t) w * h);
    for (int y = 0; y < h; ++y) {
        float sin_theta = sinf((float) y / (float) (h - 1) * MO_PI_F);
        for (int x = 0; x < w; ++x) {
            const float *p = rgb + 3 * ((size_t) y * w + x);
            float *o = e->data + 4 * ((size_t) y * w + x);
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 1.0f;
            lum[(size_t) y * w + x] = luminance(p) * sin_theta;
        }
    }
    int rc = mo_hier2d_build(&e->warp, lum, (uint32_t) w, (uint32_t) h, 1);
    free(lum);
    /* world transform: linear part and its inverse (orthonormal or not) */
    static const float ident[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    const float *m = to_world9 ? to_world9 : ident;
    memcpy(e->to_world, m, sizeof(float) * 9);
    double a = m[0], b = m[1], c = m[2], d = m[3], ee = m[4], f = m[5], g = m[6], hh = m[7], i = m[8];
    double det = a * (ee * i - f * hh) - b * (d * i - f * g) + c * (d * hh - ee * g);
    double inv[9] = { (ee * i - f * hh) / det, (c * hh - b * i) / det, (b * f - c * ee) / det,
                      (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                      (d * hh - ee * g) / det, (b * g - a * hh) / det, (a * ee - b * d) / det };
    for (int k = 0; k < 9; ++k) e->to_local[k] = (float) inv[k];
    return rc;
}
void mo_envmap_free(mo_envmap *e) { free(e->data); mo_hier2d_free(&e->warp); memset(e, 0, sizeof(*e)); }

static inline mo_v3 mat3_apply(const float *m, mo_v3 v) {
    return mo_v3_make(fmaf(m[2], v.z, fmaf(m[1], v.y, m[0] * v.x)), fmaf(m[5], v.z, fmaf(m[4], v.y, m[3] * v.x)),
                      fmaf(m[8], v.z, fmaf(m[7], v.y, m[6] * v.x)));
}

/* eval_spectrum, RGB branch (envmap.cpp:270-312) */
static void envmap_lookup(const mo_envmap *e, float u, float v, float out[3]) {
    u *= (float) (e->w - 1); v *= (float) (e->h - 1);
    uint32_t px = (uint32_t) u, py = (uint32_t) v;
    if (px > (uint32_t) (e->w - 2)) px = (uint32_t) (e->w - 2);
    if (py > (uint32_t) (e->h - 2)) py = (uint32_t) (e->h - 2);
    float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const float *v00 = e->data + 4 * ((size_t) py * e->w + px), *v10 = v00 + 4, *v01 = v00 + 4 * (size_t) e->w, *v11 = v01 + 4;
    for (int k = 0; k < 3; ++k) {
        float a = fmaf(w0x, v00[k], w1x * v10[k]), b = fmaf(w0x, v01[k], w1x * v11[k]);
        out[k] = fmaf(w0y, a, w1y * b) * e->scale;
    }
}
static void dir_to_uv(mo_v3 v, float *u, float *vv) {                    /* envmap.cpp:139-142 */
    float a = mo_lm_atan2(v.x, -v.z) * (0.5f * MO_INV_PI);
    float b = mo_lm_acos(fminf(fmaxf(v.y, -1.0f), 1.0f)) * MO_INV_PI;
    *u = a - floorf(a); *vv = b - floorf(b);
}

/* bilinear footprint of envmap_lookup: radiance[k] = sum_i w[i] * texel[idx[i]][k] (weights include `scale`) -- d radiance / d texel */
void mo_envmap_footprint(const mo_envmap *e, float u, float v, uint32_t idx[4], float w[4]) {
    u *= (float) (e->w - 1); v *= (float) (e->h - 1);
    uint32_t px = (uint32_t) u, py = (uint32_t) v;
    if (px > (uint32_t) (e->w - 2)) px = (uint32_t) (e->w - 2);
    if (py > (uint32_t) (e->h - 2)) py = (uint32_t) (e->h - 2);
    float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    idx[0] = py * (uint32_t) e->w + px; idx[1] = idx[0] + 1; idx[2] = idx[0] + (uint32_t) e->w; idx[3] = idx[2] + 1;
    w[0] = (w0y * w0x) * e->scale; w[1] = (w0y * w1x) * e->scale; w[2] = (w1y * w0x) * e->scale; w[3] = (w1y * w1x) * e->scale;
}
void mo_envmap_dir_to_uv(const mo_envmap *e, mo_v3 d_world, float *u, float *v) {
    dir_to_uv(mat3_apply(e->to_local, d_world), u, v);
}
/* new texel values ('data' of envmap.cpp:214-218); rebuild_warp: parameters_changed() (envmap.cpp:220-253) rebuilds the sampling
 * distribution from the new luminances; 0 keeps it, so that a render is exactly linear in the texels (finite-difference tests) */
int mo_envmap_update(mo_envmap *e, const float *rgb, int rebuild_warp) {
    float *lum = rebuild_warp ? (float *) malloc(sizeof(float) * (size_t) e->w * e->h) : NULL;
    for (int y = 0; y < e->h; ++y) {
        float sin_theta = sinf((float) y / (float) (e->h - 1) * MO_PI_F);
        for (int x = 0; x < e->w; ++x) {
            const float *p = rgb + 3 * ((size_t) y * e->w + x);
            float *o = e->data + 4 * ((size_t) y * e->w + x);
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 1.0f;
            if (lum) lum[(size_t) y * e->w + x] = luminance(p) * sin_theta;
        }
    }
    int rc = 0;
    if (lum) {
        mo_hier2d_free(&e->warp);
        rc = mo_hier2d_build(&e->warp, lum, (uint32_t) e->w, (uint32_t) e->h, 1);
        free(lum);
    }
    return rc;
}

/* EnvironmentMapEmitter::eval for the world-space direction `d` the ray travels in (si.wi = -d) */
void mo_envmap_eval(const mo_envmap *e, mo_v3 d, float out[3]) {
    mo_v3 v = mat3_apply(e->to_local, d);
    float u, vv; dir_to_uv(v, &u, &vv);
    envmap_lookup(e, u, vv, out);
}

/* sample_direction (envmap.cpp:154-190): returns the world direction, its pdf and radiance / pdf */
void mo_envmap_sample(const mo_envmap *e, mo_v2 sample, mo_v3 *d_out, float *pdf_out, float spec[3], mo_v2 *uv_out) {
    float u, v, pdf;
    mo_hier2d_sample(&e->warp, sample.x, sample.y, &u, &v, &pdf);
    float theta = v * MO_PI_F, phi = u * (2.0f * MO_PI_F);
    float st = mo_lm_sin(theta), ct = mo_lm_cos(theta), sp = mo_lm_sin(phi), cp = mo_lm_cos(phi);
    mo_v3 sd = mo_v3_make(cp * st, sp * st, ct);                         /* math::sphdir */
    mo_v3 d = mo_v3_make(sd.y, sd.z, -sd.x);
    float inv_sin_theta = 1.0f / sqrtf(fmaxf(d.x * d.x + d.z * d.z, MO_EPSILON * MO_EPSILON));
    d = mat3_apply(e->to_world, d);
    float ds_pdf = pdf > 0.0f ? pdf * inv_sin_theta * (1.0f / (2.0f * (MO_PI_F * MO_PI_F))) : 0.0f;
    float val[3]; envmap_lookup(e, u, v, val);
    float r = mo_rcp(ds_pdf);
    for (int k = 0; k < 3; ++k) spec[k] = val[k] * r;
    *d_out = d; *pdf_out = ds_pdf;
    if (uv_out) { uv_out->x = u; uv_out->y = v; }
}

/* eval_spectrum, spectral branch (envmap.cpp:283-306): texels hold (srgb model coefficients, scale) after
 * mo_scene_set_spectral (envmap.cpp:100-109); whitepoint = Texture::D65(1.f), i.e. the D65 table / 10568 */
void mo_envmap_lookup_spectral(const mo_envmap *e, float u, float v, const float *wav, float *out) {
    u *= (float) (e->w - 1); v *= (float) (e->h - 1);
    uint32_t px = (uint32_t) u, py = (uint32_t) v;
    if (px > (uint32_t) (e->w - 2)) px = (uint32_t) (e->w - 2);
    if (py > (uint32_t) (e->h - 2)) py = (uint32_t) (e->h - 2);
    float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const float *v00 = e->data + 4 * ((size_t) py * e->w + px), *v10 = v00 + 4, *v01 = v00 + 4 * (size_t) e->w, *v11 = v01 + 4;
    float f0 = fmaf(w0x, v00[3], w1x * v10[3]), f1 = fmaf(w0x, v01[3], w1x * v11[3]);
    float f = fmaf(w0y, f0, w1y * f1);
    for (int k = 0; k < MO_WAV; ++k) {
        float s00 = mo_srgb_model_eval(v00, wav[k]), s10 = mo_srgb_model_eval(v10, wav[k]);
        float s01 = mo_srgb_model_eval(v01, wav[k]), s11 = mo_srgb_model_eval(v11, wav[k]);
        float s0 = fmaf(w0x, s00, w1x * s10), s1 = fmaf(w0x, s01, w1x * s11);
        float sp = fmaf(w0y, s0, w1y * s1);
        float wp = mo_d65_eval(1.0f / 10568.0f, wav[k]);
        out[k] = ((sp * wp) * f) * e->scale;
    }
}
void mo_envmap_eval_spectral(const mo_envmap *e, mo_v3 d, const float *wav, float *out) {
    mo_v3 v = mat3_apply(e->to_local, d);
    float u, vv; dir_to_uv(v, &u, &vv);
    mo_envmap_lookup_spectral(e, u, vv, wav, out);
}

/* pdf_direction (envmap.cpp:192-208) */
float mo_envmap_pdf(const mo_envmap *e, mo_v3 d_world) {
    mo_v3 d = mat3_apply(e->to_local, d_world);
    float u, v; dir_to_uv(d, &u, &v);
    float inv_sin_theta = 1.0f / sqrtf(fmaxf(d.x * d.x + d.z * d.z, MO_EPSILON * MO_EPSILON));
    return mo_hier2d_eval(&e->warp, u, v) * inv_sin_theta * (1.0f / (2.0f * (MO_PI_F * MO_PI_F)));
}

/* ------------------------------------------------------------------ known-answer entry points */
/* which: 0 sample, 1 invert, 2 eval (out = (x, y, pdf) or pdf) */
void mo_kat_hier2d(const float *data, uint32_t w, uint32_t h, int normalize, int which, uint64_t n, const float *in2, float *out3) {
    mo_hier2d hd;
    if (mo_hier2d_build(&hd, data, w, h, normalize)) return;
    for (uint64_t i = 0; i < n; ++i) {
        float *o = out3 + 3 * i;
        if (which == 0) mo_hier2d_sample(&hd, in2[2 * i], in2[2 * i + 1], o, o + 1, o + 2);
        else if (which == 1) mo_hier2d_invert(&hd, in2[2 * i], in2[2 * i + 1], o, o + 1, o + 2);
        else { o[0] = in2[2 * i]; o[1] = in2[2 * i + 1]; o[2] = mo_hier2d_eval(&hd, in2[2 * i], in2[2 * i + 1]); }
    }
    mo_hier2d_free(&hd);
}
/* bilinear_to_square (warp.h:417-433) */
void mo_kat_bilinear_to_square(float v00, float v10, float v01, float v11, float x, float y, float *out3) {
    out3[0] = x; out3[1] = y;
    bilinear_to_square(v00, v10, v01, v11, &out3[0], &out3[1], &out3[2]);
}
/* envmap emitter: out per sample = d(3) pdf spec(3) eval_at_d(3) pdf_direction(d) = 11 floats */
void mo_kat_envmap(int w, int h, const float *rgb, float scale, const float *to_world9, uint64_t n, const float *sample2, float *out11) {
    mo_envmap e;
    if (mo_envmap_init(&e, w, h, rgb, scale, to_world9)) return;
    for (uint64_t i = 0; i < n; ++i) {
        float *o = out11 + 11 * i;
        mo_v2 s = { sample2[2 * i], sample2[2 * i + 1] };
        mo_v3 d; mo_envmap_sample(&e, s, &d, &o[3], &o[4], NULL);
        o[0] = d.x; o[1] = d.y; o[2] = d.z;
        mo_envmap_eval(&e, d, &o[7]);
        o[10] = mo_envmap_pdf(&e, d);
    }
    mo_envmap_free(&e);
}
