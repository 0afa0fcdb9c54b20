/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_math.h header for scope and pinning).
 *
 * Driver, sensor, integrator and film of the hot path, restated from:
 *   SamplingIntegrator::render/render_block/render_sample  src/librender/integrator.cpp:52-271
 *   PathIntegrator::sample            src/integrators/path.cpp:100-227
 *   IndependentSampler                src/samplers/independent.cpp:47-95
 *   PerspectiveCamera                 src/sensors/perspective.cpp:106-222
 *   Transform::perspective            include/mitsuba/core/transform.h:203-220
 *   ReconstructionFilter              include/mitsuba/core/rfilter.h:62-65, src/libcore/rfilter.cpp:9-20
 *   GaussianFilter / BoxFilter        src/rfilters/gaussian.cpp:33-47, src/rfilters/box.cpp:30-36
 *   Tent / CatmullRom / Mitchell / Lanczos   src/rfilters/tent.cpp:27-35, catmullrom.cpp:23-43, mitchell.cpp:30-56, lanczos.cpp:33-48
 *   ImageBlock::put                   src/librender/imageblock.cpp:49-172
 *   accumulate_2d                     include/mitsuba/core/bitmap.h:657-716
 *   Spiral                            src/librender/spiral.cpp:8-74
 *   HDRFilm::bitmap                   src/films/hdrfilm.cpp:249-320 (+ struct.cpp:1761-1811)
 *   srgb_to_xyz                       include/mitsuba/core/spectrum.h:220-227
 */
#include "mo_internal.h"
#include <stdlib.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ================================================================== */
/* 4x4 matrices (row-major storage m[r][c]); products follow Enoki's
 * column-wise fmadd order: C(r,j) = fma(A(r,3),B(3,j), fma(A(r,2),B(2,j), fma(A(r,1),B(1,j), A(r,0)*B(0,j)))) */
typedef struct { float m[4][4]; } mat4;

static mat4 mat_identity(void) {
    mat4 r; memset(&r, 0, sizeof(r));
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f;
    return r;
}
static mat4 mat_mul(const mat4 *a, const mat4 *b) {
    mat4 c;
    for (int r = 0; r < 4; ++r)
        for (int j = 0; j < 4; ++j) {
            float acc = a->m[r][0] * b->m[0][j];
            for (int i = 1; i < 4; ++i) acc = fmaf(a->m[r][i], b->m[i][j], acc);
            c.m[r][j] = acc;
        }
    return c;
}
static mat4 mat_transpose(const mat4 *a) {
    mat4 c;
    for (int r = 0; r < 4; ++r) for (int j = 0; j < 4; ++j) c.m[r][j] = a->m[j][r];
    return c;
}
typedef struct { mat4 matrix, inv_t; } xform;
static xform xf_mul(const xform *a, const xform *b) {
    xform r; r.matrix = mat_mul(&a->matrix, &b->matrix); r.inv_t = mat_mul(&a->inv_t, &b->inv_t);
    return r;
}
static xform xf_scale(float x, float y, float z) {
    xform r; r.matrix = mat_identity(); r.inv_t = mat_identity();
    r.matrix.m[0][0] = x; r.matrix.m[1][1] = y; r.matrix.m[2][2] = z;
    r.inv_t.m[0][0] = 1.0f / x; r.inv_t.m[1][1] = 1.0f / y; r.inv_t.m[2][2] = 1.0f / z;
    return r;
}
static xform xf_translate(float x, float y, float z) {
    xform r; r.matrix = mat_identity();
    r.matrix.m[0][3] = x; r.matrix.m[1][3] = y; r.matrix.m[2][3] = z;
    mat4 inv = mat_identity();
    inv.m[0][3] = -x; inv.m[1][3] = -y; inv.m[2][3] = -z;
    r.inv_t = mat_transpose(&inv);
    return r;
}
/* transform.h:203-220 */
static xform xf_perspective(float fov, float near_, float far_) {
    float recip = 1.0f / (far_ - near_);
    float tan_ = tanf((fov * 0.5f) * (MO_PI / 180.0f)), cot = 1.0f / tan_;
    mat4 t; memset(&t, 0, sizeof(t));
    t.m[0][0] = cot; t.m[1][1] = cot; t.m[2][2] = far_ * recip; t.m[3][3] = 0.0f;
    t.m[2][3] = -near_ * far_ * recip; t.m[3][2] = 1.0f;
    mat4 it; memset(&it, 0, sizeof(it));
    it.m[0][0] = tan_; it.m[1][1] = tan_; it.m[2][2] = 0.0f; it.m[3][3] = 1.0f / near_;
    it.m[2][3] = 1.0f; it.m[3][2] = (near_ - far_) / (far_ * near_);
    xform r; r.matrix = t; r.inv_t = mat_transpose(&it);
    return r;
}

typedef struct {
    mat4 sample_to_camera, to_world;
    float near_clip, far_clip;
    float aperture_radius, focus_distance;       /* thin lens (aperture_radius > 0) */
} camera;

/* perspective.cpp:106-131 */
static void camera_init(const mo_render_desc *d, camera *c) {
    float fw = (float) d->film_w, fh = (float) d->film_h;
    float rel_sx = (float) d->crop_w / fw, rel_sy = (float) d->crop_h / fh;
    float rel_ox = (float) d->crop_x / fw, rel_oy = (float) d->crop_y / fh;
    float aspect = fw / fh;
    xform a = xf_scale(1.0f / rel_sx, 1.0f / rel_sy, 1.0f);
    xform b = xf_translate(-rel_ox, -rel_oy, 0.0f);
    xform e = xf_scale(-0.5f, -0.5f * aspect, 1.0f);
    xform f = xf_translate(-1.0f, -1.0f / aspect, 0.0f);
    xform p = xf_perspective(d->fov_x_deg, d->near_clip, d->far_clip);
    xform c2s = xf_mul(&a, &b);
    c2s = xf_mul(&c2s, &e); c2s = xf_mul(&c2s, &f); c2s = xf_mul(&c2s, &p);
    /* Transform::inverse(): matrix = transpose(inverse_transpose) */
    c->sample_to_camera = mat_transpose(&c2s.inv_t);
    memcpy(c->to_world.m, d->to_world, sizeof(float) * 16);
    c->near_clip = d->near_clip; c->far_clip = d->far_clip;
    c->aperture_radius = d->aperture_radius; c->focus_distance = d->focus_distance;
}

/* perspective.cpp:190-222 (ray part; differentials are unused by `path` with constant textures); with an aperture:
 * ThinLensCamera::sample_ray (thinlens.cpp:175-214) */
static void camera_sample_ray(const camera *c, float sx, float sy, float ap_x, float ap_y, mo_ray *ray) {
    const mat4 *m = &c->sample_to_camera;
    float r[4];
    for (int k = 0; k < 4; ++k) {
        float acc = m->m[k][3];
        acc = fmaf(m->m[k][0], sx, acc);
        acc = fmaf(m->m[k][1], sy, acc);
        acc = fmaf(m->m[k][2], 0.0f, acc);
        r[k] = acc;
    }
    float iw = mo_rcp(r[3]);
    mo_v3 near_p = mo_v3_make(r[0] * iw, r[1] * iw, r[2] * iw);
    mo_v3 dl;
    const mat4 *w = &c->to_world;
    if (c->aperture_radius > 0.0f) {
        mo_v2 aps = { ap_x, ap_y };
        mo_v2 t = mo_square_to_uniform_disk_concentric(aps);
        mo_v3 aperture_p = mo_v3_make(c->aperture_radius * t.x, c->aperture_radius * t.y, 0.0f);
        mo_v3 focus_p = mo_scale(near_p, c->focus_distance / near_p.z);
        dl = mo_normalize(mo_sub(focus_p, aperture_p));
        float o[3];
        for (int k = 0; k < 3; ++k) {                   /* transform_affine(aperture_p) */
            float acc = w->m[k][3];
            acc = fmaf(w->m[k][0], aperture_p.x, acc);
            acc = fmaf(w->m[k][1], aperture_p.y, acc);
            acc = fmaf(w->m[k][2], aperture_p.z, acc);
            o[k] = acc;
        }
        ray->o = mo_v3_make(o[0], o[1], o[2]);
    } else {
        dl = mo_normalize(near_p);
        ray->o = mo_v3_make(w->m[0][3], w->m[1][3], w->m[2][3]);
    }
    float inv_z = mo_rcp(dl.z);
    ray->mint = c->near_clip * inv_z;
    ray->maxt = c->far_clip * inv_z;
    float dd[3];
    for (int k = 0; k < 3; ++k) {
        float acc = w->m[k][0] * dl.x;
        acc = fmaf(w->m[k][1], dl.y, acc);
        acc = fmaf(w->m[k][2], dl.z, acc);
        dd[k] = acc;
    }
    ray->d = mo_v3_make(dd[0], dd[1], dd[2]);
}

void mo_camera_rays(const mo_render_desc *d, uint64_t n, const float *sx, const float *sy, const float *ap, float *o3,
                    float *d3, float *mint, float *maxt) {
    camera c; camera_init(d, &c);
    for (uint64_t i = 0; i < n; ++i) {
        mo_ray r; camera_sample_ray(&c, sx[i], sy[i], ap ? ap[2 * i] : 0.5f, ap ? ap[2 * i + 1] : 0.5f, &r);
        o3[3 * i] = r.o.x; o3[3 * i + 1] = r.o.y; o3[3 * i + 2] = r.o.z;
        d3[3 * i] = r.d.x; d3[3 * i + 1] = r.d.y; d3[3 * i + 2] = r.d.z;
        mint[i] = r.mint; maxt[i] = r.maxt;
    }
}

/* ================================================================== */
/* reconstruction filter */
#define MO_FILTER_RES 31
typedef struct {
    int kind; float radius, scale_factor; int border;
    float alpha, bias;          /* gaussian */
    float values[MO_FILTER_RES + 1];
} rfilter;

/* B-spline family of mitchell.cpp:41-56 / catmullrom.cpp:29-43 (B = 0, C = 0.5) */
static float cubic_filter(float x, float B, float C) {
    x = fabsf(x);
    float x2 = x * x, x3 = x2 * x;
    float result = (1.0f / 6.0f) * (x < 1.0f
        ? (12.0f - 9.0f * B - 6.0f * C) * x3 + (-18.0f + 12.0f * B + 6.0f * C) * x2 + (6.0f - 2.0f * B)
        : (-B - 6.0f * C) * x3 + (6.0f * B + 30.0f * C) * x2 + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C));
    return x < 2.0f ? result : 0.0f;
}
static float rfilter_eval(const rfilter *f, float x) {
    switch (f->kind) {
    case 0: return fmaxf(0.0f, mo_lm_exp(f->alpha * (x * x)) - f->bias);
    case 2: return fmaxf(0.0f, 1.0f - fabsf(x * f->alpha));                    /* tent.cpp:33-35, alpha = 1 / radius */
    case 3: return cubic_filter(x, 0.0f, 0.5f);
    case 4: return cubic_filter(x, f->alpha, f->bias);                         /* alpha = B, bias = C */
    case 5: {                                                                  /* lanczos.cpp:38-48 */
        x = fabsf(x);
        float x1 = MO_PI_F * x, x2 = x1 / f->radius, result = (mo_lm_sin(x1) * mo_lm_sin(x2)) / (x1 * x2);
        return x < MO_EPSILON ? 1.0f : (x > f->radius ? 0.0f : result);
    }
    default: return fabsf(x) <= f->radius ? 1.0f : 0.0f;
    }
}
/* param / param2: gaussian stddev | box radius | mitchell B, C | lanczos lobes (the others take none) */
static void rfilter_init(rfilter *f, int kind, float param, float param2) {
    f->kind = kind;
    f->alpha = f->bias = 0;
    if (kind == 0) {
        float stddev = param;
        f->radius = 4 * stddev;
        f->alpha = -1.0f / (2.0f * stddev * stddev);
        f->bias = mo_lm_exp(f->alpha * (f->radius * f->radius));
    } else if (kind == 2) {
        f->radius = 1.0f; f->alpha = 1.0f / f->radius;
    } else if (kind == 3) {
        f->radius = 2.0f;
    } else if (kind == 4) {
        f->radius = 2.0f; f->alpha = param; f->bias = param2;
    } else if (kind == 5) {
        f->radius = (float) (int) param;
    } else {
        f->radius = param + MO_RAY_EPSILON;
    }
    for (int i = 0; i < MO_FILTER_RES; ++i)
        f->values[i] = rfilter_eval(f, (f->radius * (float) i) / (float) MO_FILTER_RES);
    f->values[MO_FILTER_RES] = 0;
    f->scale_factor = (float) MO_FILTER_RES / f->radius;
    f->border = (int) ceilf(f->radius - 0.5f - 2.0f * MO_RAY_EPSILON);
}
static inline float rfilter_eval_discretized(const rfilter *f, float x) {
    int idx = (int) fabsf(x * f->scale_factor);
    if (idx > MO_FILTER_RES) idx = MO_FILTER_RES;
    return f->values[idx];
}
void mo_rfilter_table(int kind, float param, float param2, float *table32, float *radius, int *border) {
    rfilter f; rfilter_init(&f, kind, param, param2);
    memcpy(table32, f.values, sizeof(float) * 32);
    *radius = f.radius; *border = f.border;
}

/* ================================================================== */
/* ImageBlock */
typedef struct {
    int w, h, ox, oy, ch, border;
    const rfilter *filter; int analytic;
    float *data;
} iblock;

static size_t iblock_floats(const iblock *b) {
    return (size_t) b->ch * (size_t) (b->w + 2 * b->border) * (size_t) (b->h + 2 * b->border);
}

/* imageblock.cpp:80-172 (warn_negative = warn_invalid = true, normalize = false) */
static int iblock_put(iblock *b, float px, float py, const float *value) {
    for (int k = 0; k < b->ch; ++k)
        if (!(value[k] >= -1e-5f) || !isfinite(value[k])) return 0;
    const rfilter *f = b->filter;
    float radius = f->radius;
    int sx = b->w + 2 * b->border, sy = b->h + 2 * b->border;
    float posx = px - ((float) (b->ox - b->border) + 0.5f), posy = py - ((float) (b->oy - b->border) + 0.5f);
    if (radius > 1.0f) {
        int lox = (int) ceilf(posx - radius), loy = (int) ceilf(posy - radius);
        int hix = (int) floorf(posx + radius), hiy = (int) floorf(posy + radius);
        if (lox < 0) lox = 0;
        if (loy < 0) loy = 0;
        if (hix > sx - 1) hix = sx - 1;
        if (hiy > sy - 1) hiy = sy - 1;
        uint32_t n = (uint32_t) ceilf((radius - 2.0f * MO_RAY_EPSILON) * 2.0f);
        float wx[64], wy[64];
        float basex = (float) (uint32_t) lox - posx, basey = (float) (uint32_t) loy - posy;
        for (uint32_t i = 0; i < n && i < 64; ++i) {
            float ppx = basex + (float) i, ppy = basey + (float) i;
            wx[i] = b->analytic ? rfilter_eval(f, ppx) : rfilter_eval_discretized(f, ppx);
            wy[i] = b->analytic ? rfilter_eval(f, ppy) : rfilter_eval_discretized(f, ppy);
        }
        for (uint32_t yr = 0; yr < n; ++yr) {
            uint32_t y = (uint32_t) loy + yr;
            if (!(y <= (uint32_t) hiy) || hiy < 0) continue;
            for (uint32_t xr = 0; xr < n; ++xr) {
                uint32_t x = (uint32_t) lox + xr;
                if (!(x <= (uint32_t) hix) || hix < 0) break;
                size_t off = (size_t) b->ch * ((size_t) y * (size_t) sx + x);
                float weight = wy[yr] * wx[xr];
                for (int k = 0; k < b->ch; ++k) b->data[off + k] += value[k] * weight;
            }
        }
    } else {
        int lox = (int) ceilf(posx - 0.5f), loy = (int) ceilf(posy - 0.5f);
        if (lox >= 0 && loy >= 0 && lox < sx && loy < sy) {
            size_t off = (size_t) b->ch * ((size_t) loy * (size_t) sx + (size_t) lox);
            for (int k = 0; k < b->ch; ++k) b->data[off + k] += value[k];
        }
    }
    return 1;
}

int mo_imageblock_put(int w, int h, int ox, int oy, int ch, int kind, float param, float param2, int border,
                      int analytic, uint64_t n, const float *pos, const float *values, float *data) {
    rfilter f; rfilter_init(&f, kind, param, param2);
    iblock b = { w, h, ox, oy, ch, border ? f.border : 0, &f, analytic, data };
    for (uint64_t i = 0; i < n; ++i) iblock_put(&b, pos[2 * i], pos[2 * i + 1], values + (size_t) ch * i);
    return b.border;
}

/* imageblock.cpp:49-77 + bitmap.h:657-716: block (with border) += into target (with its border) */
static void iblock_put_block(iblock *target, const iblock *src) {
    int ssx = src->w + 2 * src->border, ssy = src->h + 2 * src->border;
    int tsx = target->w + 2 * target->border, tsy = target->h + 2 * target->border;
    int sox = 0, soy = 0;
    int tox = (src->ox - src->border) - (target->ox - target->border);
    int toy = (src->oy - src->border) - (target->oy - target->border);
    int szx = ssx, szy = ssy;
    int shx = 0, shy = 0;
    if (-sox > shx) shx = -sox;
    if (-tox > shx) shx = -tox;
    if (-soy > shy) shy = -soy;
    if (-toy > shy) shy = -toy;
    sox += shx; tox += shx; soy += shy; toy += shy;
    { int a = sox + szx - ssx; if (a > 0) szx -= a; a = tox + szx - tsx; if (a > 0) szx -= a; }
    { int a = soy + szy - ssy; if (a > 0) szy -= a; a = toy + szy - tsy; if (a > 0) szy -= a; }
    if (szx <= 0 || szy <= 0) return;
    int ch = src->ch;
    for (int y = 0; y < szy; ++y) {
        const float *sp = src->data + ((size_t) (sox) + (size_t) (soy + y) * ssx) * ch;
        float *tp = target->data + ((size_t) (tox) + (size_t) (toy + y) * tsx) * ch;
        for (int i = 0; i < szx * ch; ++i) tp[i] += sp[i];
    }
}

/* ================================================================== */
/* Spiral (spiral.cpp:8-74) */
typedef struct {
    int block_size, size_x, size_y, off_x, off_y, blocks_x, blocks_y;
    size_t block_count, block_counter, remaining_passes;
    int dir, pos_x, pos_y, steps_left, steps;
} spiral;

static void spiral_reset(spiral *s) {
    s->block_counter = 0; s->dir = 0;
    s->pos_x = s->blocks_x / 2; s->pos_y = s->blocks_y / 2;
    s->steps_left = 1; s->steps = 1;
}
static void spiral_init(spiral *s, int w, int h, int ox, int oy, int block_size, size_t passes) {
    s->block_size = block_size; s->size_x = w; s->size_y = h; s->off_x = ox; s->off_y = oy;
    s->remaining_passes = passes;
    s->blocks_x = (int) ceilf((float) w / (float) block_size);
    s->blocks_y = (int) ceilf((float) h / (float) block_size);
    s->block_count = (size_t) s->blocks_x * (size_t) s->blocks_y;
    spiral_reset(s);
}
/* returns 0 when exhausted */
static int spiral_next(spiral *s, int *ox, int *oy, int *w, int *h, size_t *id) {
    if (s->block_count == s->block_counter) {
        if (s->remaining_passes > 1) { --s->remaining_passes; spiral_reset(s); }
        else return 0;
    }
    *id = s->block_counter + (s->remaining_passes - 1) * s->block_count;
    int offx = s->pos_x * s->block_size, offy = s->pos_y * s->block_size;
    *w = s->size_x - offx < s->block_size ? s->size_x - offx : s->block_size;
    *h = s->size_y - offy < s->block_size ? s->size_y - offy : s->block_size;
    *ox = offx + s->off_x; *oy = offy + s->off_y;
    ++s->block_counter;
    if (s->block_counter != s->block_count) {
        do {
            switch (s->dir) {
                case 0: ++s->pos_x; break;   /* Right */
                case 1: ++s->pos_y; break;   /* Down  */
                case 2: --s->pos_x; break;   /* Left  */
                case 3: --s->pos_y; break;   /* Up    */
            }
            if (--s->steps_left == 0) {
                s->dir = (s->dir + 1) % 4;
                if (s->dir == 2 || s->dir == 0) ++s->steps;
                s->steps_left = s->steps;
            }
        } while (s->pos_x < 0 || s->pos_y < 0 || s->pos_x >= s->blocks_x || s->pos_y >= s->blocks_y);
    }
    return 1;
}

int mo_kat_spiral(int w, int h, int off_x, int off_y, int block_size, int passes, int max_entries,
                  int64_t *out5) {
    spiral s; spiral_init(&s, w, h, off_x, off_y, block_size, (size_t) passes);
    int n = 0, ox, oy, bw, bh; size_t id;
    while (spiral_next(&s, &ox, &oy, &bw, &bh, &id)) {
        if (n < max_entries) {
            out5[5 * n] = ox; out5[5 * n + 1] = oy; out5[5 * n + 2] = bw; out5[5 * n + 3] = bh;
            out5[5 * n + 4] = (int64_t) id;
        }
        ++n;
    }
    return n;
}

/* ================================================================== */
/* PathIntegrator::sample (path.cpp:100-211) */
typedef struct { uint64_t closest, any; } ray_stats;

static inline float mis_weight(float pdf_a, float pdf_b) {
    pdf_a *= pdf_a; pdf_b *= pdf_b;
    return pdf_a > 0.0f ? pdf_a / (pdf_a + pdf_b) : 0.0f;
}

static int scene_intersect(const mo_scene *s, const mo_ray *ray, mo_si *si, ray_stats *st) {
    mo_hit h;
    st->closest++;
    if (mo_intersect(s, ray, 0, 0, &h)) { mo_make_si(s, ray, &h, si); return 1; }
    memset(si, 0, sizeof(*si));
    si->t = INFINITY;
    si->wi = mo_neg(ray->d);
    return 0;
}

/* eg != NULL: besides the radiance, d(loss)/d(envmap texels) = delta * d(radiance)/d(texels) is scattered into eg->grad (h*w*3):
 * the radiance is linear in the texels at its two uses, the emission picked up by an escaped ray and the emitter sample */
typedef struct { const float *delta; float *grad; } env_grad;
static void env_grad_add(const mo_envmap *env, float u, float v, const float coeff[3], const env_grad *eg) {
    uint32_t idx[4]; float w[4];
    mo_envmap_footprint(env, u, v, idx, w);
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 3; ++k) eg->grad[3 * (size_t) idx[i] + k] += (eg->delta[k] * coeff[k]) * w[i];
}
/* pg != NULL: forward-mode derivative of the radiance w.r.t. ONE scalar parameter (kind, comp) of the BSDF of the masked shapes, carried
 * beside the path with DETACHED sampling -- the checker of mtsamd_render_adjoint_param (kernels.hip, bounce_step<PGRAD>): the same terms
 * in the same order; d(value)/d(theta) at fixed directions = central difference of the model code between records perturbed by +-h */
typedef struct { const uint8_t *shape_mask; int kind, comp; float h; float dthr[3], dres[3]; } param_grad;
static void bsdf_perturbed(const mo_bsdf *b, const float refl[3], int kind, int comp, float step, mo_bsdf *out, float refl_out[3]) {
    *out = *b;
    for (int k = 0; k < 3; ++k) refl_out[k] = refl[k];
    switch (kind) {
    case 0: refl_out[comp] = refl[comp] + step; out->d.reflectance[comp] = refl_out[comp]; break;
    case 1: out->d.specular_reflectance[comp] += step; break;
    case 2: out->d.eta[comp] += step; break;
    case 3: out->d.k[comp] += step; break;
    case 4: out->d.alpha_u += step; out->d.alpha_v += step; break;
    default: out->d.specular_transmittance[comp] += step; break;
    }
}
static float param_value(const mo_bsdf *b, const float refl[3], int kind, int comp) {
    switch (kind) {
    case 0: return refl[comp];
    case 1: return b->d.specular_reflectance[comp];
    case 2: return b->d.eta[comp];
    case 3: return b->d.k[comp];
    case 4: return b->d.alpha_u;
    default: return b->d.specular_transmittance[comp];
    }
}
static void path_sample(const mo_scene *s, mo_pcg32 *rng, const mo_ray *ray_in, int max_depth,
                        int rr_depth, float result[3], int *valid_ray, ray_stats *st, const env_grad *eg, param_grad *pg) {
    mo_ray ray = *ray_in;
    float eta = 1.0f, emission_weight = 1.0f;
    float throughput[3] = { 1.0f, 1.0f, 1.0f };
    result[0] = result[1] = result[2] = 0.0f;

    mo_si si;
    int si_valid = scene_intersect(s, &ray, &si, st);
    *valid_ray = si_valid;
    int emitter = si_valid ? s->meshes[si.shape].emitter : s->environment;     /* si.emitter(scene): interaction.h:236-243 */
    int active = 1;

    for (int depth = 1;; ++depth) {
        /* ---------------- Intersection with emitters ---------------- */
        if (emitter >= 0 && active) {
            /* AreaLight::eval (area.cpp:71-79); ConstantBackgroundEmitter::eval (constant.cpp:53-57); envmap.cpp:132-146 */
            if (s->emitters[emitter].type != 0 || si.wi.z > 0.0f) {
                float le_env[3];
                const float *le = s->emitters[emitter].radiance;
                if (s->emitters[emitter].type == 2) { mo_envmap_eval(s->emitters[emitter].env, mo_neg(si.wi), le_env); le = le_env; }
                for (int k = 0; k < 3; ++k) result[k] += (emission_weight * throughput[k]) * le[k];
                if (pg) for (int k = 0; k < 3; ++k) pg->dres[k] = pg->dres[k] + (emission_weight * pg->dthr[k]) * le[k];
                if (eg && s->emitters[emitter].type == 2) {
                    float u, v, coeff[3];
                    mo_envmap_dir_to_uv(s->emitters[emitter].env, mo_neg(si.wi), &u, &v);
                    for (int k = 0; k < 3; ++k) coeff[k] = emission_weight * throughput[k];
                    env_grad_add(s->emitters[emitter].env, u, v, coeff, eg);
                }
            }
        }
        active = active && si_valid;

        /* Russian roulette (path.cpp:137-141) */
        if (depth > rr_depth) {
            float q = fminf(fmaxf(fmaxf(throughput[0], throughput[1]), throughput[2]) * (eta * eta), 0.95f);
            if (active) active = mo_pcg32_next_f32(rng) < q;
            float rq = mo_rcp(q);
            for (int k = 0; k < 3; ++k) throughput[k] *= rq;
            if (pg) for (int k = 0; k < 3; ++k) pg->dthr[k] *= rq;
        }

        if ((uint32_t) depth >= (uint32_t) max_depth || !active) break;

        /* --------------------- Emitter sampling --------------------- */
        const mo_mesh *mesh = &s->meshes[si.shape];
        const mo_bsdf *bsdf = &mesh->bsdf;
        float refl[9];
        mo_surface_reflectance(s, mesh, si.uv, refl);
        const int pg_here = pg && pg->shape_mask[si.shape];
        mo_bsdf b_p, b_m; float refl_p[3], refl_m[3], inv_2h = 0.0f;
        if (pg_here) {
            const float theta = param_value(bsdf, refl, pg->kind, pg->comp);
            bsdf_perturbed(bsdf, refl, pg->kind, pg->comp, (theta + pg->h) - theta, &b_p, refl_p);
            bsdf_perturbed(bsdf, refl, pg->kind, pg->comp, (theta - pg->h) - theta, &b_m, refl_m);
            inv_2h = 1.0f / ((theta + pg->h) - (theta - pg->h));
        }
        if (mo_bsdf_is_smooth(bsdf)) {   /* active_e = active && has_flag(bsdf->flags(), BSDFFlags::Smooth) (path.cpp:154) */
            mo_v2 s2; s2.x = mo_pcg32_next_f32(rng); s2.y = mo_pcg32_next_f32(rng);
            mo_dsample ds; float emitter_val[3];
            mo_sample_emitter_direction(s, si.p, s2, &ds, emitter_val);
            int active_e = ds.pdf != 0.0f, occluded = 0;
            if (active_e && s->n_emitters > 0) {
                /* visibility test (scene.cpp:178-182) */
                mo_ray sr;
                sr.o = si.p; sr.d = ds.d;
                sr.mint = MO_RAY_EPSILON * (1.0f + mo_hmax_abs(si.p));
                sr.maxt = ds.dist * (1.0f - MO_SHADOW_EPSILON);
                st->any++;
                if (mo_intersect(s, &sr, 1, 0, NULL)) { emitter_val[0] = emitter_val[1] = emitter_val[2] = 0.0f; occluded = 1; }
            }
            if (active_e) {
                mo_v3 wo = mo_to_local(&si.sh, ds.d);
                float bsdf_val[3], bsdf_pdf;
                mo_bsdf_eval_pdf(bsdf, refl, si.wi, wo, bsdf_val, &bsdf_pdf);
                float mis = ds.delta ? 1.0f : mis_weight(ds.pdf, bsdf_pdf);      /* path.cpp:170 */
                for (int k = 0; k < 3; ++k)
                    result[k] += ((mis * throughput[k]) * bsdf_val[k]) * emitter_val[k];
                if (pg && !occluded) {
                    float dbv[3] = { 0.0f, 0.0f, 0.0f };
                    if (pg_here) {
                        float vp[3], vm[3], pp, pm;
                        mo_bsdf_eval_pdf(&b_p, refl_p, si.wi, wo, vp, &pp);
                        mo_bsdf_eval_pdf(&b_m, refl_m, si.wi, wo, vm, &pm);
                        for (int k = 0; k < 3; ++k) dbv[k] = (vp[k] - vm[k]) * inv_2h;
                    }
                    for (int k = 0; k < 3; ++k)
                        pg->dres[k] = pg->dres[k] + (mis * fmaf(pg->dthr[k], bsdf_val[k], throughput[k] * dbv[k])) * emitter_val[k];
                }
                if (eg && !occluded && s->emitters[ds.emitter].type == 2) {
                    /* emitter_val = lookup(uv) / pdf_single * emitter count (envmap.cpp:186-189, scene.cpp:160-163) */
                    float g = mo_rcp(ds.pdf_single) * (s->n_emitters > 1 ? (float) s->n_emitters : 1.0f), coeff[3];
                    for (int k = 0; k < 3; ++k) coeff[k] = ((mis * throughput[k]) * bsdf_val[k]) * g;
                    env_grad_add(s->emitters[ds.emitter].env, ds.uv.x, ds.uv.y, coeff, eg);
                }
            }
        }

        /* ----------------------- BSDF sampling ---------------------- */
        float s1 = mo_pcg32_next_f32(rng);
        mo_v2 s2; s2.x = mo_pcg32_next_f32(rng); s2.y = mo_pcg32_next_f32(rng);
        mo_bsample bs; float bsdf_w[3];
        mo_bsdf_sample(bsdf, refl, si.wi, s1, s2, &bs, bsdf_w);
        if (pg) {
            float dw[3] = { 0.0f, 0.0f, 0.0f };
            if (pg_here && !bs.delta && bs.pdf > 0.0f) {
                float vp[3], vm[3], pp, pm;
                mo_bsdf_eval_pdf(&b_p, refl_p, si.wi, bs.wo, vp, &pp);
                mo_bsdf_eval_pdf(&b_m, refl_m, si.wi, bs.wo, vm, &pm);
                const float ip = mo_rcp(bs.pdf);
                for (int k = 0; k < 3; ++k) dw[k] = ((vp[k] - vm[k]) * inv_2h) * ip;
            } else if (pg_here && bs.delta) {
                mo_bsample bp_, bm_; float wp[3], wm[3];
                const int okp = mo_bsdf_sample(&b_p, refl_p, si.wi, s1, s2, &bp_, wp), okm = mo_bsdf_sample(&b_m, refl_m, si.wi, s1, s2, &bm_, wm);
                if (okp && okm && bp_.delta && bm_.delta && bp_.wo.x == bs.wo.x && bp_.wo.y == bs.wo.y && bp_.wo.z == bs.wo.z &&
                    bm_.wo.x == bs.wo.x && bm_.wo.y == bs.wo.y && bm_.wo.z == bs.wo.z)
                    for (int k = 0; k < 3; ++k) dw[k] = (wp[k] - wm[k]) * inv_2h;
            }
            for (int k = 0; k < 3; ++k) pg->dthr[k] = fmaf(pg->dthr[k], bsdf_w[k], throughput[k] * dw[k]);
        }
        for (int k = 0; k < 3; ++k) throughput[k] = throughput[k] * bsdf_w[k];
        active = active && (throughput[0] != 0.0f || throughput[1] != 0.0f || throughput[2] != 0.0f);
        if (!active) break;
        eta *= bs.eta;

        /* spawn_ray (interaction.h:58-61) */
        ray.o = si.p; ray.d = mo_to_world(&si.sh, bs.wo);
        ray.mint = (1.0f + mo_hmax_abs(si.p)) * MO_RAY_EPSILON;
        ray.maxt = INFINITY;
        mo_si si_bsdf;
        int v2 = scene_intersect(s, &ray, &si_bsdf, st);
        emitter = v2 ? s->meshes[si_bsdf.shape].emitter : s->environment;
        if (emitter >= 0) {
            /* DirectionSample(si_bsdf, si) (records.h:168-174); d = -wi for environment emitters */
            mo_v3 d = mo_sub(si_bsdf.p, si.p);
            float dist = mo_norm(d);
            d = mo_div_s(d, dist);
            if (!v2) { d = mo_neg(si_bsdf.wi); dist = 0.0f; si_bsdf.sh.n = d; }
            /* delta lobes cannot be hit by emitter sampling (path.cpp:198-203) */
            float emitter_pdf = bs.delta ? 0.0f : mo_pdf_emitter_direction(s, (uint32_t) emitter, d, si_bsdf.sh.n, dist);
            emission_weight = mis_weight(bs.pdf, emitter_pdf);
        }
        si = si_bsdf; si_valid = v2;
    }
}

/* PathIntegrator::sample for a packet of 8 camera rays (the packet_rgb instantiation of path.cpp:100-211: the loop runs while any
 * lane is active).  The two ray queries of an iteration are traced 8 wide (mo_packet.c: one stack, lane masks, lane voting); the
 * stages between them run lane by lane through the scalar functions above, so every lane computes exactly what path_sample computes
 * for its ray and RNG stream -- tests/test_oracle_packet.py checks that bit for bit. */
static void packet_scene_intersect(const mo_scene *s, const mo_packet_accel *acc, const mo_ray *rays, uint32_t lanes, mo_si *si,
                                   int *valid, ray_stats *st) {
    mo_hit h[8];
    st->closest += (uint64_t) __builtin_popcount(lanes);
    const uint32_t found = mo_packet_intersect(s, acc, rays, lanes, 0, h);
    for (int l = 0; l < 8; ++l) {
        if (!((lanes >> l) & 1u)) continue;
        if ((found >> l) & 1u) { mo_make_si(s, &rays[l], &h[l], &si[l]); valid[l] = 1; }
        else { memset(&si[l], 0, sizeof(si[l])); si[l].t = INFINITY; si[l].wi = mo_neg(rays[l].d); valid[l] = 0; }
    }
}

static void path_sample_packet(const mo_scene *s, const mo_packet_accel *acc, mo_pcg32 *rng, const mo_ray *rays_in, uint32_t lanes,
                               int max_depth, int rr_depth, float (*result)[3], int *valid_ray, ray_stats *st) {
    mo_ray ray[8]; mo_si si[8], si_bsdf[8]; int si_valid[8], v2[8], emitter[8], active[8];
    float eta[8], emission_weight[8], throughput[8][3];
    /* per-lane values that live from the emitter-sampling stage to the BSDF stage of one iteration */
    float refl[8][9], emitter_val[8][3]; mo_dsample ds[8]; int active_e[8], smooth[8]; mo_bsample bs[8];
    for (int l = 0; l < 8; ++l) {
        ray[l] = rays_in[l]; eta[l] = 1.0f; emission_weight[l] = 1.0f; active[l] = 1; active_e[l] = 0; smooth[l] = 0;
        for (int k = 0; k < 3; ++k) { throughput[l][k] = 1.0f; result[l][k] = 0.0f; }
    }
    packet_scene_intersect(s, acc, ray, lanes, si, si_valid, st);
    for (int l = 0; l < 8; ++l) {
        if (!((lanes >> l) & 1u)) continue;
        valid_ray[l] = si_valid[l];
        emitter[l] = si_valid[l] ? s->meshes[si[l].shape].emitter : s->environment;
    }
    uint32_t alive = lanes;                 /* lanes still inside the loop */
    for (int depth = 1; alive; ++depth) {
        mo_ray shadow[8]; uint32_t shadow_lanes = 0;
        for (int l = 0; l < 8; ++l) {
            if (!((alive >> l) & 1u)) continue;
            if (emitter[l] >= 0 && active[l]) {
                const mo_emitter *e = &s->emitters[emitter[l]];
                if (e->type != 0 || si[l].wi.z > 0.0f) {
                    float le_env[3];
                    const float *le = e->radiance;
                    if (e->type == 2) { mo_envmap_eval(e->env, mo_neg(si[l].wi), le_env); le = le_env; }
                    for (int k = 0; k < 3; ++k) result[l][k] += (emission_weight[l] * throughput[l][k]) * le[k];
                }
            }
            active[l] = active[l] && si_valid[l];
            if (depth > rr_depth) {
                float q = fminf(fmaxf(fmaxf(throughput[l][0], throughput[l][1]), throughput[l][2]) * (eta[l] * eta[l]), 0.95f);
                if (active[l]) active[l] = mo_pcg32_next_f32(&rng[l]) < q;
                float rq = mo_rcp(q);
                for (int k = 0; k < 3; ++k) throughput[l][k] *= rq;
            }
            if ((uint32_t) depth >= (uint32_t) max_depth || !active[l]) { alive &= ~(1u << l); continue; }
            const mo_mesh *mesh = &s->meshes[si[l].shape];
            mo_surface_reflectance(s, mesh, si[l].uv, refl[l]);
            smooth[l] = mo_bsdf_is_smooth(&mesh->bsdf);
            active_e[l] = 0;
            if (smooth[l]) {
                mo_v2 s2; s2.x = mo_pcg32_next_f32(&rng[l]); s2.y = mo_pcg32_next_f32(&rng[l]);
                mo_sample_emitter_direction(s, si[l].p, s2, &ds[l], emitter_val[l]);
                active_e[l] = ds[l].pdf != 0.0f;
                if (active_e[l] && s->n_emitters > 0) {
                    shadow[l].o = si[l].p; shadow[l].d = ds[l].d;
                    shadow[l].mint = MO_RAY_EPSILON * (1.0f + mo_hmax_abs(si[l].p));
                    shadow[l].maxt = ds[l].dist * (1.0f - MO_SHADOW_EPSILON);
                    shadow_lanes |= 1u << l;
                }
            }
        }
        if (!alive) break;
        if (shadow_lanes) {
            st->any += (uint64_t) __builtin_popcount(shadow_lanes);
            const uint32_t occluded = mo_packet_intersect(s, acc, shadow, shadow_lanes, 1, NULL);
            for (int l = 0; l < 8; ++l)
                if ((occluded >> l) & 1u) emitter_val[l][0] = emitter_val[l][1] = emitter_val[l][2] = 0.0f;
        }
        uint32_t next_lanes = 0;
        for (int l = 0; l < 8; ++l) {
            if (!((alive >> l) & 1u)) continue;
            const mo_bsdf *bsdf = &s->meshes[si[l].shape].bsdf;
            if (smooth[l] && active_e[l]) {
                mo_v3 wo = mo_to_local(&si[l].sh, ds[l].d);
                float bsdf_val[3], bsdf_pdf;
                mo_bsdf_eval_pdf(bsdf, refl[l], si[l].wi, wo, bsdf_val, &bsdf_pdf);
                float mis = ds[l].delta ? 1.0f : mis_weight(ds[l].pdf, bsdf_pdf);
                for (int k = 0; k < 3; ++k) result[l][k] += ((mis * throughput[l][k]) * bsdf_val[k]) * emitter_val[l][k];
            }
            float s1 = mo_pcg32_next_f32(&rng[l]);
            mo_v2 s2; s2.x = mo_pcg32_next_f32(&rng[l]); s2.y = mo_pcg32_next_f32(&rng[l]);
            float bsdf_w[3];
            mo_bsdf_sample(bsdf, refl[l], si[l].wi, s1, s2, &bs[l], bsdf_w);
            for (int k = 0; k < 3; ++k) throughput[l][k] = throughput[l][k] * bsdf_w[k];
            active[l] = active[l] && (throughput[l][0] != 0.0f || throughput[l][1] != 0.0f || throughput[l][2] != 0.0f);
            if (!active[l]) { alive &= ~(1u << l); continue; }
            eta[l] *= bs[l].eta;
            ray[l].o = si[l].p; ray[l].d = mo_to_world(&si[l].sh, bs[l].wo);
            ray[l].mint = (1.0f + mo_hmax_abs(si[l].p)) * MO_RAY_EPSILON;
            ray[l].maxt = INFINITY;
            next_lanes |= 1u << l;
        }
        if (!next_lanes) break;
        packet_scene_intersect(s, acc, ray, next_lanes, si_bsdf, v2, st);
        for (int l = 0; l < 8; ++l) {
            if (!((next_lanes >> l) & 1u)) continue;
            emitter[l] = v2[l] ? s->meshes[si_bsdf[l].shape].emitter : s->environment;
            if (emitter[l] >= 0) {
                mo_v3 d = mo_sub(si_bsdf[l].p, si[l].p);
                float dist = mo_norm(d);
                d = mo_div_s(d, dist);
                if (!v2[l]) { d = mo_neg(si_bsdf[l].wi); dist = 0.0f; si_bsdf[l].sh.n = d; }
                float emitter_pdf = bs[l].delta ? 0.0f : mo_pdf_emitter_direction(s, (uint32_t) emitter[l], d, si_bsdf[l].sh.n, dist);
                emission_weight[l] = mis_weight(bs[l].pdf, emitter_pdf);
            }
            si[l] = si_bsdf[l]; si_valid[l] = v2[l];
        }
    }
}

/* DirectIntegrator::sample (direct.cpp:105-196) */
static void direct_sample(const mo_scene *s, mo_pcg32 *rng, const mo_ray *ray, int emitter_samples, int bsdf_samples,
                          int hide_emitters, float result[3], int *valid_ray, ray_stats *st) {
    if (emitter_samples == 0 && bsdf_samples == 0) emitter_samples = bsdf_samples = 1;      /* shading_samples = 1 (direct.cpp:88-95) */
    const float sum = (float) (emitter_samples + bsdf_samples);
    const float weight_bsdf = 1.0f / (float) bsdf_samples, weight_lum = 1.0f / (float) emitter_samples;
    const float frac_bsdf = (float) bsdf_samples / sum, frac_lum = (float) emitter_samples / sum;
    result[0] = result[1] = result[2] = 0.0f;
    mo_si si;
    int valid = scene_intersect(s, ray, &si, st);
    *valid_ray = valid;
    if (!valid) {                                                                              /* environment seen directly */
        if (!hide_emitters && s->environment >= 0) {
            float le_env[3];
            const float *le = s->emitters[s->environment].radiance;
            if (s->emitters[s->environment].type == 2) { mo_envmap_eval(s->emitters[s->environment].env, mo_neg(si.wi), le_env); le = le_env; }
            for (int k = 0; k < 3; ++k) result[k] += le[k];
        }
        return;
    }
    const mo_mesh *mesh = &s->meshes[si.shape];
    if (!hide_emitters && mesh->emitter >= 0 && si.wi.z > 0.0f)                                /* visible emitters (direct.cpp:117-121) */
        for (int k = 0; k < 3; ++k) result[k] += s->emitters[mesh->emitter].radiance[k];
    const mo_bsdf *bsdf = &mesh->bsdf;
    float refl[9];
    mo_surface_reflectance(s, mesh, si.uv, refl);
    if (mo_bsdf_is_smooth(bsdf)) {
        for (int i = 0; i < emitter_samples; ++i) {
            mo_v2 s2; s2.x = mo_pcg32_next_f32(rng); s2.y = mo_pcg32_next_f32(rng);
            mo_dsample ds; float emitter_val[3];
            mo_sample_emitter_direction(s, si.p, s2, &ds, emitter_val);
            int active_e = ds.pdf != 0.0f;
            if (active_e && s->n_emitters > 0) {
                mo_ray sr;
                sr.o = si.p; sr.d = ds.d;
                sr.mint = MO_RAY_EPSILON * (1.0f + mo_hmax_abs(si.p));
                sr.maxt = ds.dist * (1.0f - MO_SHADOW_EPSILON);
                st->any++;
                if (mo_intersect(s, &sr, 1, 0, NULL)) emitter_val[0] = emitter_val[1] = emitter_val[2] = 0.0f;
            }
            if (!active_e) continue;
            mo_v3 wo = mo_to_local(&si.sh, ds.d);
            float bsdf_val[3], bsdf_pdf;
            mo_bsdf_eval_pdf(bsdf, refl, si.wi, wo, bsdf_val, &bsdf_pdf);
            float mis = ds.delta ? 1.0f : mis_weight(ds.pdf * frac_lum, bsdf_pdf * frac_bsdf) * weight_lum;      /* direct.cpp:155-156 */
            for (int k = 0; k < 3; ++k) result[k] += (mis * bsdf_val[k]) * emitter_val[k];
        }
    }
    for (int i = 0; i < bsdf_samples; ++i) {
        float s1 = mo_pcg32_next_f32(rng);
        mo_v2 s2; s2.x = mo_pcg32_next_f32(rng); s2.y = mo_pcg32_next_f32(rng);
        mo_bsample bs; float bsdf_val[3];
        mo_bsdf_sample(bsdf, refl, si.wi, s1, s2, &bs, bsdf_val);
        if (!(bsdf_val[0] != 0.0f || bsdf_val[1] != 0.0f || bsdf_val[2] != 0.0f)) continue;
        mo_ray r2;
        r2.o = si.p; r2.d = mo_to_world(&si.sh, bs.wo);
        r2.mint = (1.0f + mo_hmax_abs(si.p)) * MO_RAY_EPSILON; r2.maxt = INFINITY;
        mo_si si_bsdf;
        int v2 = scene_intersect(s, &r2, &si_bsdf, st);
        int emitter = v2 ? s->meshes[si_bsdf.shape].emitter : s->environment;
        if (emitter < 0) continue;
        float emitter_val[3] = { 0.0f, 0.0f, 0.0f };
        if (!v2 && s->emitters[emitter].type == 2) mo_envmap_eval(s->emitters[emitter].env, mo_neg(si_bsdf.wi), emitter_val);
        else if (!v2 || si_bsdf.wi.z > 0.0f) for (int k = 0; k < 3; ++k) emitter_val[k] = s->emitters[emitter].radiance[k];
        mo_v3 d = mo_sub(si_bsdf.p, si.p);
        float dist = mo_norm(d);
        d = mo_div_s(d, dist);
        if (!v2) { d = mo_neg(si_bsdf.wi); dist = 0.0f; si_bsdf.sh.n = d; }
        float emitter_pdf = bs.delta ? 0.0f : mo_pdf_emitter_direction(s, (uint32_t) emitter, d, si_bsdf.sh.n, dist);
        float w = mis_weight(bs.pdf * frac_bsdf, emitter_pdf * frac_lum) * weight_bsdf;
        for (int k = 0; k < 3; ++k) result[k] += (bsdf_val[k] * emitter_val[k]) * w;
    }
}

/* DepthIntegrator::sample (depth.cpp:19-33) */
static void depth_sample(const mo_scene *s, const mo_ray *ray, float result[3], int *valid_ray, ray_stats *st) {
    mo_si si;
    int valid = scene_intersect(s, ray, &si, st);
    *valid_ray = valid;
    result[0] = result[1] = result[2] = valid ? si.t : 0.0f;
}

/* radiance spectrum of emitter `e` at 4 wavelengths: SRGBEmitterSpectrum (srgb_d65.cpp:54-62) for `area` / `constant`,
 * eval_spectrum (envmap.cpp:283-306) for `envmap` -- by direction (`d`) or at known texture coordinates (`uv`) */
static void emitter_spectrum(const mo_emitter *e, const float wav[MO_WAV], const mo_v3 *d, const mo_v2 *uv, float le[MO_WAV]) {
    if (e->type == 2) {
        if (uv) mo_envmap_lookup_spectral(e->env, uv->x, uv->y, wav, le);
        else mo_envmap_eval_spectral(e->env, *d, wav, le);
        return;
    }
    for (int k = 0; k < MO_WAV; ++k) le[k] = mo_d65_eval(e->d65_scale, wav[k]) * mo_srgb_model_eval(e->coeff, wav[k]);
}

/* PathIntegrator::sample for the spectral variant: identical control flow, 4 wavelength channels */
static void path_sample_spectral(const mo_scene *s, mo_pcg32 *rng, const mo_ray *ray_in, const float wav[MO_WAV],
                                 int max_depth, int rr_depth, float result[MO_WAV], int *valid_ray, ray_stats *st) {
    mo_ray ray = *ray_in;
    float eta = 1.0f, emission_weight = 1.0f;
    float throughput[MO_WAV];
    for (int k = 0; k < MO_WAV; ++k) { throughput[k] = 1.0f; result[k] = 0.0f; }
    mo_si si;
    int si_valid = scene_intersect(s, &ray, &si, st);
    *valid_ray = si_valid;
    int emitter = si_valid ? s->meshes[si.shape].emitter : s->environment;
    int active = 1;
    for (int depth = 1;; ++depth) {
        if (emitter >= 0 && active && (s->emitters[emitter].type != 0 || si.wi.z > 0.0f)) {
            float le[MO_WAV];
            mo_v3 d = mo_neg(si.wi);
            emitter_spectrum(&s->emitters[emitter], wav, &d, NULL, le);
            for (int k = 0; k < MO_WAV; ++k) result[k] += (emission_weight * throughput[k]) * le[k];
        }
        active = active && si_valid;
        if (depth > rr_depth) {
            float hm = fmaxf(fmaxf(throughput[0], throughput[1]), fmaxf(throughput[2], throughput[3]));
            float q = fminf(hm * (eta * eta), 0.95f);
            if (active) active = mo_pcg32_next_f32(rng) < q;
            float rq = mo_rcp(q);
            for (int k = 0; k < MO_WAV; ++k) throughput[k] *= rq;
        }
        if ((uint32_t) depth >= (uint32_t) max_depth || !active) break;
        const mo_mesh *mesh = &s->meshes[si.shape];
        const mo_bsdf *bsdf = &mesh->bsdf;
        mo_bsdf_chan chan;
        mo_bsdf_spectral_channels(bsdf, wav, &chan);
        if (mesh->texture >= 0) mo_reflectance_spectral(s, mesh, si.uv, wav, chan.refl);
        if (mo_bsdf_is_smooth(bsdf)) {
            mo_v2 s2; s2.x = mo_pcg32_next_f32(rng); s2.y = mo_pcg32_next_f32(rng);
            mo_dsample ds; float rgb_spec[3];
            mo_sample_emitter_direction(s, si.p, s2, &ds, rgb_spec);
            if (ds.pdf != 0.0f && s->n_emitters > 0) {
                const mo_emitter *e = &s->emitters[ds.emitter];
                /* sample_direction: spec = radiance / pdf, masked for area lights (area.cpp:110-116, constant.cpp:103-106,
                 * envmap.cpp:186-189); Scene: * emitter count (scene.cpp:160-163) */
                float r2 = s->n_emitters > 1 ? mo_rcp(1.0f / (float) s->n_emitters) : 1.0f;
                float pdf_single = s->n_emitters > 1 ? ds.pdf_single : ds.pdf;
                int act = e->type != 0 || (mo_dot(ds.d, ds.n) < 0.0f && pdf_single != 0.0f);
                float r1 = act ? mo_rcp(pdf_single) : 0.0f;
                mo_ray sr;
                sr.o = si.p; sr.d = ds.d;
                sr.mint = MO_RAY_EPSILON * (1.0f + mo_hmax_abs(si.p));
                sr.maxt = ds.dist * (1.0f - MO_SHADOW_EPSILON);
                st->any++;
                int occluded = mo_intersect(s, &sr, 1, 0, NULL);
                mo_v3 wo = mo_to_local(&si.sh, ds.d);
                float bv[MO_WAV], bsdf_pdf, le[MO_WAV];
                mo_bsdf_eval_pdf_spec(bsdf, wav, &chan, si.wi, wo, bv, &bsdf_pdf);
                float mis = ds.delta ? 1.0f : mis_weight(ds.pdf, bsdf_pdf);
                emitter_spectrum(e, wav, &ds.d, &ds.uv, le);
                for (int k = 0; k < MO_WAV; ++k) {
                    float spec = ds.delta ? (le[k] * ds.falloff) * ds.scale : le[k] * r1;
                    if (s->n_emitters > 1) spec *= r2;
                    if (occluded) spec = 0.0f;
                    result[k] += ((mis * throughput[k]) * bv[k]) * spec;
                }
            }
        }
        float s1 = mo_pcg32_next_f32(rng);
        mo_v2 s2; s2.x = mo_pcg32_next_f32(rng); s2.y = mo_pcg32_next_f32(rng);
        mo_bsample bs; float bsdf_w[MO_WAV];
        mo_bsdf_sample_spec(bsdf, wav, &chan, si.wi, s1, s2, &bs, bsdf_w);
        int nz = 0;
        for (int k = 0; k < MO_WAV; ++k) { throughput[k] = throughput[k] * bsdf_w[k]; nz = nz || throughput[k] != 0.0f; }
        active = active && nz;
        if (!active) break;
        eta *= bs.eta;
        ray.o = si.p; ray.d = mo_to_world(&si.sh, bs.wo);
        ray.mint = (1.0f + mo_hmax_abs(si.p)) * MO_RAY_EPSILON;
        ray.maxt = INFINITY;
        mo_si si_bsdf;
        int v2 = scene_intersect(s, &ray, &si_bsdf, st);
        emitter = v2 ? s->meshes[si_bsdf.shape].emitter : s->environment;
        if (emitter >= 0) {
            mo_v3 d = mo_sub(si_bsdf.p, si.p);
            float dist = mo_norm(d);
            d = mo_div_s(d, dist);
            if (!v2) { d = mo_neg(si_bsdf.wi); dist = 0.0f; si_bsdf.sh.n = d; }
            float emitter_pdf = bs.delta ? 0.0f : mo_pdf_emitter_direction(s, (uint32_t) emitter, d, si_bsdf.sh.n, dist);
            emission_weight = mis_weight(bs.pdf, emitter_pdf);
        }
        si = si_bsdf; si_valid = v2;
    }
}

/* The same for the spectral variant: 8 lanes, each with its own 4 wavelengths; ray queries 8 wide (lane voting), shading per lane with
 * the scalar code of path_sample_spectral -- every lane computes exactly what the scalar path computes for its ray and stream. */
static void path_sample_packet_spectral(const mo_scene *s, const mo_packet_accel *acc, mo_pcg32 *rng, const mo_ray *rays_in, uint32_t lanes,
                                        float (*wav)[MO_WAV], int max_depth, int rr_depth, float (*result)[MO_WAV], int *valid_ray,
                                        ray_stats *st) {
    mo_ray ray[8]; mo_si si[8], si_bsdf[8]; int si_valid[8], v2[8], emitter[8], active[8];
    float eta[8], emission_weight[8], throughput[8][MO_WAV];
    mo_bsdf_chan chan[8]; mo_dsample ds[8]; int active_e[8], smooth[8]; mo_bsample bs[8];
    for (int l = 0; l < 8; ++l) {
        ray[l] = rays_in[l]; eta[l] = 1.0f; emission_weight[l] = 1.0f; active[l] = 1; active_e[l] = 0; smooth[l] = 0;
        for (int k = 0; k < MO_WAV; ++k) { throughput[l][k] = 1.0f; result[l][k] = 0.0f; }
    }
    packet_scene_intersect(s, acc, ray, lanes, si, si_valid, st);
    for (int l = 0; l < 8; ++l) {
        if (!((lanes >> l) & 1u)) continue;
        valid_ray[l] = si_valid[l];
        emitter[l] = si_valid[l] ? s->meshes[si[l].shape].emitter : s->environment;
    }
    uint32_t alive = lanes;
    for (int depth = 1; alive; ++depth) {
        mo_ray shadow[8]; uint32_t shadow_lanes = 0;
        for (int l = 0; l < 8; ++l) {
            if (!((alive >> l) & 1u)) continue;
            if (emitter[l] >= 0 && active[l] && (s->emitters[emitter[l]].type != 0 || si[l].wi.z > 0.0f)) {
                float le[MO_WAV];
                mo_v3 d = mo_neg(si[l].wi);
                emitter_spectrum(&s->emitters[emitter[l]], wav[l], &d, NULL, le);
                for (int k = 0; k < MO_WAV; ++k) result[l][k] += (emission_weight[l] * throughput[l][k]) * le[k];
            }
            active[l] = active[l] && si_valid[l];
            if (depth > rr_depth) {
                float hm = fmaxf(fmaxf(throughput[l][0], throughput[l][1]), fmaxf(throughput[l][2], throughput[l][3]));
                float q = fminf(hm * (eta[l] * eta[l]), 0.95f);
                if (active[l]) active[l] = mo_pcg32_next_f32(&rng[l]) < q;
                float rq = mo_rcp(q);
                for (int k = 0; k < MO_WAV; ++k) throughput[l][k] *= rq;
            }
            if ((uint32_t) depth >= (uint32_t) max_depth || !active[l]) { alive &= ~(1u << l); continue; }
            const mo_mesh *mesh = &s->meshes[si[l].shape];
            mo_bsdf_spectral_channels(&mesh->bsdf, wav[l], &chan[l]);
            if (mesh->texture >= 0) mo_reflectance_spectral(s, mesh, si[l].uv, wav[l], chan[l].refl);
            smooth[l] = mo_bsdf_is_smooth(&mesh->bsdf);
            active_e[l] = 0;
            if (smooth[l]) {
                mo_v2 s2; s2.x = mo_pcg32_next_f32(&rng[l]); s2.y = mo_pcg32_next_f32(&rng[l]);
                float rgb_spec[3];
                mo_sample_emitter_direction(s, si[l].p, s2, &ds[l], rgb_spec);
                active_e[l] = ds[l].pdf != 0.0f && s->n_emitters > 0;
                if (active_e[l]) {
                    shadow[l].o = si[l].p; shadow[l].d = ds[l].d;
                    shadow[l].mint = MO_RAY_EPSILON * (1.0f + mo_hmax_abs(si[l].p));
                    shadow[l].maxt = ds[l].dist * (1.0f - MO_SHADOW_EPSILON);
                    shadow_lanes |= 1u << l;
                }
            }
        }
        if (!alive) break;
        uint32_t occluded = 0;
        if (shadow_lanes) {
            st->any += (uint64_t) __builtin_popcount(shadow_lanes);
            occluded = mo_packet_intersect(s, acc, shadow, shadow_lanes, 1, NULL);
        }
        uint32_t next_lanes = 0;
        for (int l = 0; l < 8; ++l) {
            if (!((alive >> l) & 1u)) continue;
            const mo_bsdf *bsdf = &s->meshes[si[l].shape].bsdf;
            if (smooth[l] && active_e[l]) {
                const mo_emitter *e = &s->emitters[ds[l].emitter];
                float r2 = s->n_emitters > 1 ? mo_rcp(1.0f / (float) s->n_emitters) : 1.0f;
                float pdf_single = s->n_emitters > 1 ? ds[l].pdf_single : ds[l].pdf;
                int act = e->type != 0 || (mo_dot(ds[l].d, ds[l].n) < 0.0f && pdf_single != 0.0f);
                float r1 = act ? mo_rcp(pdf_single) : 0.0f;
                mo_v3 wo = mo_to_local(&si[l].sh, ds[l].d);
                float bv[MO_WAV], bsdf_pdf, le[MO_WAV];
                mo_bsdf_eval_pdf_spec(bsdf, wav[l], &chan[l], si[l].wi, wo, bv, &bsdf_pdf);
                float mis = ds[l].delta ? 1.0f : mis_weight(ds[l].pdf, bsdf_pdf);
                emitter_spectrum(e, wav[l], &ds[l].d, &ds[l].uv, le);
                for (int k = 0; k < MO_WAV; ++k) {
                    float spec = ds[l].delta ? (le[k] * ds[l].falloff) * ds[l].scale : le[k] * r1;
                    if (s->n_emitters > 1) spec *= r2;
                    if ((occluded >> l) & 1u) spec = 0.0f;
                    result[l][k] += ((mis * throughput[l][k]) * bv[k]) * spec;
                }
            }
            float s1 = mo_pcg32_next_f32(&rng[l]);
            mo_v2 s2; s2.x = mo_pcg32_next_f32(&rng[l]); s2.y = mo_pcg32_next_f32(&rng[l]);
            float bsdf_w[MO_WAV];
            mo_bsdf_sample_spec(bsdf, wav[l], &chan[l], si[l].wi, s1, s2, &bs[l], bsdf_w);
            int nz = 0;
            for (int k = 0; k < MO_WAV; ++k) { throughput[l][k] = throughput[l][k] * bsdf_w[k]; nz = nz || throughput[l][k] != 0.0f; }
            active[l] = active[l] && nz;
            if (!active[l]) { alive &= ~(1u << l); continue; }
            eta[l] *= bs[l].eta;
            ray[l].o = si[l].p; ray[l].d = mo_to_world(&si[l].sh, bs[l].wo);
            ray[l].mint = (1.0f + mo_hmax_abs(si[l].p)) * MO_RAY_EPSILON;
            ray[l].maxt = INFINITY;
            next_lanes |= 1u << l;
        }
        if (!next_lanes) break;
        packet_scene_intersect(s, acc, ray, next_lanes, si_bsdf, v2, st);
        for (int l = 0; l < 8; ++l) {
            if (!((next_lanes >> l) & 1u)) continue;
            emitter[l] = v2[l] ? s->meshes[si_bsdf[l].shape].emitter : s->environment;
            if (emitter[l] >= 0) {
                mo_v3 d = mo_sub(si_bsdf[l].p, si[l].p);
                float dist = mo_norm(d);
                d = mo_div_s(d, dist);
                if (!v2[l]) { d = mo_neg(si_bsdf[l].wi); dist = 0.0f; si_bsdf[l].sh.n = d; }
                float emitter_pdf = bs[l].delta ? 0.0f : mo_pdf_emitter_direction(s, (uint32_t) emitter[l], d, si_bsdf[l].sh.n, dist);
                emission_weight[l] = mis_weight(bs[l].pdf, emitter_pdf);
            }
            si[l] = si_bsdf[l]; si_valid[l] = v2[l];
        }
    }
}

/* spectrum.h:220-227 (Matrix * Vector: column-wise fmadd) */
static inline void srgb_to_xyz(const float rgb[3], float xyz[3]) {
    static const float M[3][3] = { { 0.412453f, 0.357580f, 0.180423f },
                                   { 0.212671f, 0.715160f, 0.072169f },
                                   { 0.019334f, 0.119193f, 0.950227f } };
    for (int r = 0; r < 3; ++r)
        xyz[r] = fmaf(M[r][2], rgb[2], fmaf(M[r][1], rgb[1], M[r][0] * rgb[0]));
}

/* render_sample (integrator.cpp:224-271); pos = integer pixel position */
static void render_sample(const mo_scene *s, const mo_render_desc *d, const camera *cam, mo_pcg32 *rng,
                          float pos_x, float pos_y, float aovs[5], float pos_sample[2], float rgb_out[3],
                          int *valid_out, ray_stats *st) {
    float jx = mo_pcg32_next_f32(rng), jy = mo_pcg32_next_f32(rng);
    float psx = pos_x + jx, psy = pos_y + jy;
    float apx = 0.5f, apy = 0.5f;                         /* needs_aperture_sample(): integrator.cpp:229-231 */
    if (cam->aperture_radius > 0.0f) { apx = mo_pcg32_next_f32(rng); apy = mo_pcg32_next_f32(rng); }
    float wavelength_sample = mo_pcg32_next_f32(rng);
    float ax = (psx - (float) d->crop_x) / (float) d->crop_w, ay = (psy - (float) d->crop_y) / (float) d->crop_h;
    mo_ray ray; camera_sample_ray(cam, ax, ay, apx, apy, &ray);
    float L[3] = { 0, 0, 0 }; int valid;
    float xyz[3];
    if (s->spectral) {
        /* sample_wavelength (perspective.cpp:196), ray_weight * L, spectrum_to_xyz (integrator.cpp:250-261) */
        float wav[MO_WAV], weight[MO_WAV], Ls[MO_WAV];
        mo_sample_wavelengths(wavelength_sample, wav, weight);
        path_sample_spectral(s, rng, &ray, wav, d->max_depth, d->rr_depth, Ls, &valid, st);
        for (int k = 0; k < MO_WAV; ++k) Ls[k] = weight[k] * Ls[k];
        mo_spectrum_to_xyz(Ls, wav, xyz);
        L[0] = xyz[0]; L[1] = xyz[1]; L[2] = xyz[2];       /* per-sample API: XYZ tristimulus in the spectral variant */
    } else {
    if (d->integrator == 1) direct_sample(s, rng, &ray, d->emitter_samples, d->bsdf_samples, d->hide_emitters, L, &valid, st);
    else if (d->integrator == 2) depth_sample(s, &ray, L, &valid, st);
    else path_sample(s, rng, &ray, d->max_depth, d->rr_depth, L, &valid, st, NULL, NULL);
    /* ray_weight == 1 in RGB mode (spectrum.h:304-309) */
    srgb_to_xyz(L, xyz);
    }
    if (d->film_rgb) { xyz[0] = L[0]; xyz[1] = L[1]; xyz[2] = L[2]; }     /* autodiff.py:53-57: linear RGB channels */
    aovs[0] = xyz[0]; aovs[1] = xyz[1]; aovs[2] = xyz[2]; aovs[3] = valid ? 1.0f : 0.0f; aovs[4] = 1.0f;
    pos_sample[0] = psx; pos_sample[1] = psy;
    if (rgb_out) { rgb_out[0] = L[0]; rgb_out[1] = L[1]; rgb_out[2] = L[2]; }
    if (valid_out) *valid_out = valid;
}

/* IndependentSampler::seed, wavefront flavour (independent.cpp:62-72) */
static inline void seed_wavefront(mo_pcg32 *rng, uint64_t index, uint64_t base_seed) {
    uint64_t seed_value = index + base_seed, idx = index;
    mo_pcg32_seed(rng, mo_tea64_u64(seed_value, idx, 4), mo_tea64_u64(idx, seed_value, 4));
}

static int desc_check(const mo_render_desc *d) {
    if (d->max_depth < 0 && d->max_depth != -1) return -1;   /* integrator.cpp:290-292 */
    if (d->rr_depth <= 0) return -1;
    if (d->crop_w <= 0 || d->crop_h <= 0 || d->spp <= 0) return -1;
    if (d->integrator < 0 || d->integrator > 2 || d->emitter_samples < 0 || d->bsdf_samples < 0) return -1;
    return 0;
}

int mo_sample_radiance(const mo_scene *s, const mo_render_desc *d, uint64_t first, uint64_t count,
                       float *out_rgba, float *out_pos) {
    if (desc_check(d)) return -1;
    camera cam; camera_init(d, &cam);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t k = 0; k < (int64_t) count; ++k) {
        uint64_t i = first + (uint64_t) k;
        mo_pcg32 rng; seed_wavefront(&rng, i, d->base_seed);
        uint64_t pixel = i / (uint64_t) d->spp;
        float px = (float) (uint32_t) (pixel % (uint64_t) d->crop_w), py = (float) (uint32_t) (pixel / (uint64_t) d->crop_w);
        float aovs[5], ps[2], rgb[3]; int valid; ray_stats st = { 0, 0 };
        /* GPU branch positions are crop-relative pixel indices (integrator.cpp:152-161); the
         * sensor sees them through crop_offset, so add it to stay on the same film position. */
        render_sample(s, d, &cam, &rng, px + (float) d->crop_x, py + (float) d->crop_y, aovs, ps, rgb, &valid, &st);
        out_rgba[4 * k] = rgb[0]; out_rgba[4 * k + 1] = rgb[1]; out_rgba[4 * k + 2] = rgb[2];
        out_rgba[4 * k + 3] = valid ? 1.0f : 0.0f;
        if (out_pos) { out_pos[2 * k] = ps[0]; out_pos[2 * k + 1] = ps[1]; }
    }
    return 0;
}

/* col0 / col1: only the pixels of the columns [col0, col1) are sampled (tests of very large films: the film pixels at least a filter
 * radius inside the window still receive every sample that reaches them) */
static int render_wavefront_window(const mo_scene *s, const mo_render_desc *d, int row0, int row1, int col0, int col1,
                                   float *film, uint64_t *stats) {
    camera cam; camera_init(d, &cam);
    rfilter f; rfilter_init(&f, d->rfilter, d->rfilter_param, d->rfilter_param2);
    /* ImageBlock(film_size, 5, filter, border = true) then film->put(block) (integrator.cpp:156-168) */
    iblock blk = { d->crop_w, d->crop_h, d->crop_x, d->crop_y, 5, f.border, &f, d->filter_analytic, NULL };
    blk.data = (float *) calloc(iblock_floats(&blk), sizeof(float));
    uint64_t n0 = (uint64_t) row0 * d->crop_w * d->spp, n1 = (uint64_t) row1 * d->crop_w * d->spp;
    uint64_t chunk = 1u << 16;
    float *buf = (float *) malloc(sizeof(float) * 7 * chunk);
    ray_stats total = { 0, 0 };
    for (uint64_t c0 = n0; c0 < n1; c0 += chunk) {
        uint64_t cn = n1 - c0 < chunk ? n1 - c0 : chunk;
        uint64_t cl = 0, an = 0;
#pragma omp parallel for schedule(dynamic, 512) reduction(+ : cl, an)
        for (int64_t k = 0; k < (int64_t) cn; ++k) {
            uint64_t i = c0 + (uint64_t) k;
            mo_pcg32 rng; seed_wavefront(&rng, i, d->base_seed);
            uint64_t pixel = i / (uint64_t) d->spp;
            const int pxi = (int) (pixel % (uint64_t) d->crop_w);
            if (pxi < col0 || pxi >= col1) { buf[7 * k + 3] = -2.0f; continue; }      /* outside the window: no sample (alpha is never -2) */
            float px = (float) (uint32_t) pxi, py = (float) (uint32_t) (pixel / (uint64_t) d->crop_w);
            ray_stats st = { 0, 0 };
            render_sample(s, d, &cam, &rng, px + (float) d->crop_x, py + (float) d->crop_y, buf + 7 * k, buf + 7 * k + 5, NULL, NULL, &st);
            cl += st.closest; an += st.any;
        }
        total.closest += cl; total.any += an;
        for (uint64_t k = 0; k < cn; ++k)
            if (buf[7 * k + 3] != -2.0f) iblock_put(&blk, buf[7 * k + 5], buf[7 * k + 6], buf + 7 * k);
    }
    free(buf);
    memset(film, 0, sizeof(float) * 5 * (size_t) d->crop_w * d->crop_h);
    iblock storage = { d->crop_w, d->crop_h, d->crop_x, d->crop_y, 5, 0, NULL, 0, film };
    iblock_put_block(&storage, &blk);
    free(blk.data);
    if (stats) { stats[0] = total.closest; stats[1] = total.any; stats[2] = n1 - n0; }
    return 0;
}

static int render_wavefront_rows(const mo_scene *s, const mo_render_desc *d, int row0, int row1, float *film, uint64_t *stats) {
    return render_wavefront_window(s, d, row0, row1, 0, d->crop_w, film, stats);
}
int mo_render_rows(const mo_scene *s, const mo_render_desc *d, int row0, int row1, float *film) {
    if (desc_check(d) || row0 < 0 || row1 > d->crop_h || row0 > row1) return -1;
    return render_wavefront_rows(s, d, row0, row1, film, NULL);
}
int mo_render_window(const mo_scene *s, const mo_render_desc *d, int row0, int row1, int col0, int col1, float *film) {
    if (desc_check(d) || row0 < 0 || row1 > d->crop_h || row0 > row1 || col0 < 0 || col1 > d->crop_w || col0 > col1) return -1;
    return render_wavefront_window(s, d, row0, row1, col0, col1, film, NULL);
}

/* scalar_rgb branch of SamplingIntegrator::render (integrator.cpp:76-143) + render_block (:178-203) */
/* flavour 0: scalar_rgb (one PCG32 stream per block, pixels in Morton order, spp consecutive samples per pixel);
 * flavour 2: packet_rgb (integrator.cpp:204-212): the block's pixel_count * spp sample indices are processed 8 at a time, lane l of
 *            a packet takes index base + l (pixel = morton_decode(index / spp)), and the sampler holds 8 PCG32 streams seeded like
 *            the wavefront flavour with idx = lane (independent.cpp:62-72); ray queries 8 wide (path_sample_packet);
 * flavour 3: the same schedule and streams as 2, every lane traced by the scalar code -- the checker of flavour 2. */
static int render_blocks(const mo_scene *s, const mo_render_desc *d, int n_threads, int block_size,
                         float *film, uint64_t *stats, int flavour) {
    if (flavour == 2 && (d->integrator != 0 || d->aperture_radius > 0.0f)) return -2;      /* `path`, pinhole camera */
    mo_packet_accel *acc = flavour == 2 ? mo_packet_accel_build(s) : NULL;
    camera cam; camera_init(d, &cam);
    rfilter f; rfilter_init(&f, d->rfilter, d->rfilter_param, d->rfilter_param2);
    if (block_size == 0) {
        uint32_t bs = 32;
        while (1) {
            size_t nb = (size_t) ((d->crop_w + bs - 1) / bs) * (size_t) ((d->crop_h + bs - 1) / bs);
            if (bs == 1 || nb >= (size_t) n_threads) break;
            bs /= 2;
        }
        block_size = (int) bs;
    }
    spiral sp; spiral_init(&sp, d->crop_w, d->crop_h, d->crop_x, d->crop_y, block_size, 1);
    size_t nblocks = sp.block_count;
    int *bo = (int *) malloc(sizeof(int) * 4 * nblocks);
    size_t *ids = (size_t *) malloc(sizeof(size_t) * nblocks);
    for (size_t b = 0; b < nblocks; ++b)
        spiral_next(&sp, &bo[4 * b], &bo[4 * b + 1], &bo[4 * b + 2], &bo[4 * b + 3], &ids[b]);
    memset(film, 0, sizeof(float) * 5 * (size_t) d->crop_w * d->crop_h);
    iblock storage = { d->crop_w, d->crop_h, d->crop_x, d->crop_y, 5, 0, NULL, 0, film };
    int bsz = block_size + 2 * f.border;
    size_t per_block = (size_t) 5 * bsz * bsz;
    /* blocks are rendered in parallel into private ImageBlocks and merged in spiral order */
    size_t batch = 256;
    float *bufs = (float *) malloc(sizeof(float) * per_block * batch);
    uint64_t cl = 0, an = 0;
    for (size_t b0 = 0; b0 < nblocks; b0 += batch) {
        size_t bn = nblocks - b0 < batch ? nblocks - b0 : batch;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : cl, an)
        for (int64_t bi = 0; bi < (int64_t) bn; ++bi) {
            size_t b = b0 + (size_t) bi;
            iblock blk = { bo[4 * b + 2], bo[4 * b + 3], bo[4 * b], bo[4 * b + 1], 5, f.border, &f, d->filter_analytic,
                           bufs + per_block * (size_t) bi };
            memset(blk.data, 0, sizeof(float) * iblock_floats(&blk));
            mo_pcg32 rng;  /* sampler->seed(block_id): scalar flavour (independent.cpp:73-75) */
            mo_pcg32_seed(&rng, (uint64_t) ids[b] + d->base_seed, MO_PCG32_DEFAULT_STREAM);
            ray_stats st = { 0, 0 };
            uint32_t pixel_count = (uint32_t) (block_size * block_size);
            if (flavour != 0) {
                mo_pcg32 lane_rng[8];
                for (uint64_t l = 0; l < 8; ++l) {       /* seed(block_id), array flavour: seed_value + idx streams (independent.cpp:66-72) */
                    const uint64_t seed_value = (uint64_t) ids[b] + d->base_seed;
                    mo_pcg32_seed(&lane_rng[l], mo_tea64_u64(seed_value, l, 4), mo_tea64_u64(l, seed_value, 4));
                }
                const uint64_t total = (uint64_t) pixel_count * (uint64_t) d->spp;
                for (uint64_t base = 0; base < total; base += 8) {
                    uint32_t lanes = 0; float px[8], py[8];
                    for (uint32_t l = 0; l < 8; ++l) {
                        const uint64_t index = base + l;
                        if (index >= total) continue;
                        uint32_t mx, my; mo_morton_decode2((uint32_t) (index / (uint64_t) d->spp), &mx, &my);
                        if (mx >= (uint32_t) blk.w || my >= (uint32_t) blk.h) continue;
                        px[l] = (float) (mx + (uint32_t) blk.ox); py[l] = (float) (my + (uint32_t) blk.oy);
                        lanes |= 1u << l;
                    }
                    if (!lanes) continue;
                    float aovs[8][5], ps[8][2];
                    if (flavour == 3) {
                        for (uint32_t l = 0; l < 8; ++l)
                            if ((lanes >> l) & 1u) render_sample(s, d, &cam, &lane_rng[l], px[l], py[l], aovs[l], ps[l], NULL, NULL, &st);
                    } else {
                        /* render_sample (integrator.cpp:224-271) around the 8-wide PathIntegrator::sample */
                        mo_ray rays[8]; float L[8][3]; int valid[8]; float wsample[8];
                        for (uint32_t l = 0; l < 8; ++l) {
                            if (!((lanes >> l) & 1u)) continue;
                            float jx = mo_pcg32_next_f32(&lane_rng[l]), jy = mo_pcg32_next_f32(&lane_rng[l]);
                            ps[l][0] = px[l] + jx; ps[l][1] = py[l] + jy;
                            wsample[l] = mo_pcg32_next_f32(&lane_rng[l]);            /* wavelength sample */
                            float ax = (ps[l][0] - (float) d->crop_x) / (float) d->crop_w, ay = (ps[l][1] - (float) d->crop_y) / (float) d->crop_h;
                            camera_sample_ray(&cam, ax, ay, 0.5f, 0.5f, &rays[l]);
                        }
                        if (s->spectral) {      /* sample_wavelength, ray_weight * L, spectrum_to_xyz as render_sample does per lane */
                            float wav[8][MO_WAV], weight[8][MO_WAV], Ls[8][MO_WAV];
                            for (uint32_t l = 0; l < 8; ++l)
                                if ((lanes >> l) & 1u) mo_sample_wavelengths(wsample[l], wav[l], weight[l]);
                            path_sample_packet_spectral(s, acc, lane_rng, rays, lanes, wav, d->max_depth, d->rr_depth, Ls, valid, &st);
                            for (uint32_t l = 0; l < 8; ++l) {
                                if (!((lanes >> l) & 1u)) continue;
                                for (int k = 0; k < MO_WAV; ++k) Ls[l][k] = weight[l][k] * Ls[l][k];
                                mo_spectrum_to_xyz(Ls[l], wav[l], L[l]);
                            }
                        } else path_sample_packet(s, acc, lane_rng, rays, lanes, d->max_depth, d->rr_depth, L, valid, &st);
                        for (uint32_t l = 0; l < 8; ++l) {
                            if (!((lanes >> l) & 1u)) continue;
                            float xyz[3];
                            if (s->spectral) { xyz[0] = L[l][0]; xyz[1] = L[l][1]; xyz[2] = L[l][2]; }
                            else srgb_to_xyz(L[l], xyz);
                            if (d->film_rgb) { xyz[0] = L[l][0]; xyz[1] = L[l][1]; xyz[2] = L[l][2]; }
                            aovs[l][0] = xyz[0]; aovs[l][1] = xyz[1]; aovs[l][2] = xyz[2]; aovs[l][3] = valid[l] ? 1.0f : 0.0f; aovs[l][4] = 1.0f;
                        }
                    }
                    for (uint32_t l = 0; l < 8; ++l)
                        if ((lanes >> l) & 1u) iblock_put(&blk, ps[l][0], ps[l][1], aovs[l]);
                }
                pixel_count = 0;            /* the scalar loop below is skipped */
            }
            for (uint32_t i = 0; i < pixel_count; ++i) {
                uint32_t mx, my; mo_morton_decode2(i, &mx, &my);
                if (mx >= (uint32_t) blk.w || my >= (uint32_t) blk.h) continue;
                float px = (float) (mx + (uint32_t) blk.ox), py = (float) (my + (uint32_t) blk.oy);
                for (int j = 0; j < d->spp; ++j) {
                    float aovs[5], ps[2];
                    render_sample(s, d, &cam, &rng, px, py, aovs, ps, NULL, NULL, &st);
                    iblock_put(&blk, ps[0], ps[1], aovs);
                }
            }
            cl += st.closest; an += st.any;
        }
        for (size_t bi = 0; bi < bn; ++bi) {
            size_t b = b0 + bi;
            iblock blk = { bo[4 * b + 2], bo[4 * b + 3], bo[4 * b], bo[4 * b + 1], 5, f.border, &f, d->filter_analytic,
                           bufs + per_block * bi };
            iblock_put_block(&storage, &blk);
        }
    }
    free(bufs); free(bo); free(ids);
    mo_packet_accel_free(acc);
    if (stats) { stats[0] = cl; stats[1] = an; stats[2] = (uint64_t) d->crop_w * d->crop_h * d->spp; }
    return 0;
}

int mo_render(const mo_scene *s, const mo_render_desc *d, int mode, int n_threads, int block_size,
              float *film, uint64_t *stats) {
    if (desc_check(d)) return -1;
#ifdef _OPENMP
    int prev = omp_get_max_threads();
    if (n_threads <= 0) n_threads = omp_get_num_procs();
    omp_set_num_threads(n_threads);
#else
    n_threads = 1;
#endif
    int rc = mode == 1 ? render_wavefront_rows(s, d, 0, d->crop_h, film, stats)
                       : render_blocks(s, d, n_threads, block_size, film, stats, mode);
#ifdef _OPENMP
    omp_set_num_threads(prev);
#endif
    return rc;
}

/* ================================================================== */
/* adjoint of the path integrator w.r.t. diffuse reflectances */
typedef struct {
    float E[3], Nc[3], Tp[3], rho[3], invq;
    float T[3]; int rr_channel;      /* throughput before Russian roulette; channel that sets q (-1: none / q clamped) */
    uint32_t texel; float w1[2]; uint32_t shape; int has_bsdf;
    /* radiance-free coefficients for d/d(emitter radiance): E = ew * Le[em_hit], Nc = nk * Le[em_nee] (-1: none) */
    float ew, nk; int em_hit, em_nee;
} vertex_rec;
#define MO_ADJ_MAX_DEPTH 16

/* PathIntegrator::sample (path.cpp:100-211) with per-vertex bookkeeping; returns the number of vertices */
static int path_sample_rec(const mo_scene *s, mo_pcg32 *rng, const mo_ray *ray_in, int max_depth, int rr_depth,
                           vertex_rec *rec) {
    mo_ray ray = *ray_in;
    float eta = 1.0f, emission_weight = 1.0f;
    float throughput[3] = { 1.0f, 1.0f, 1.0f };
    ray_stats st = { 0, 0 };
    mo_si si;
    int si_valid = scene_intersect(s, &ray, &si, &st);
    int emitter = si_valid ? s->meshes[si.shape].emitter : -1;
    int active = 1, n = 0;
    for (int depth = 1; n < MO_ADJ_MAX_DEPTH; ++depth) {
        vertex_rec *r = &rec[n++];
        memset(r, 0, sizeof(*r));
        r->invq = 1.0f; r->texel = 0xffffffffu; r->rr_channel = -1; r->em_hit = r->em_nee = -1;
        for (int k = 0; k < 3; ++k) r->T[k] = throughput[k];
        if (emitter >= 0 && active && si.wi.z > 0.0f) {
            const float *le = s->emitters[emitter].radiance;
            for (int k = 0; k < 3; ++k) r->E[k] = emission_weight * le[k];
            r->ew = emission_weight; r->em_hit = emitter;
        }
        active = active && si_valid;
        if (depth > rr_depth) {
            float hm = fmaxf(fmaxf(throughput[0], throughput[1]), throughput[2]);
            float q = fminf(hm * (eta * eta), 0.95f);
            if (active) active = mo_pcg32_next_f32(rng) < q;
            float rq = mo_rcp(q);
            if (hm * (eta * eta) < 0.95f) r->rr_channel = throughput[0] == hm ? 0 : (throughput[1] == hm ? 1 : 2);
            for (int k = 0; k < 3; ++k) throughput[k] *= rq;
            r->invq = rq;
        }
        if ((uint32_t) depth >= (uint32_t) max_depth || !active) break;
        const mo_mesh *mesh = &s->meshes[si.shape];
        float refl[3];
        mo_reflectance(s, mesh, si.uv, refl, &r->texel, r->w1);
        r->has_bsdf = 1; r->shape = si.shape;
        /* `twosided` around the diffuse BSDF (twosided.cpp:94-175): the back side scatters like the front side, mirrored */
        mo_v3 wi_b = si.wi;
        const int flip = mesh->bsdf.d.twosided && wi_b.z < 0.0f;
        if (flip) wi_b.z = -wi_b.z;
        for (int k = 0; k < 3; ++k) { r->Tp[k] = throughput[k]; r->rho[k] = refl[k]; }
        {
            mo_v2 s2; s2.x = mo_pcg32_next_f32(rng); s2.y = mo_pcg32_next_f32(rng);
            mo_dsample ds; float emitter_val[3];
            mo_sample_emitter_direction(s, si.p, s2, &ds, emitter_val);
            if (ds.pdf != 0.0f && s->n_emitters > 0) {
                mo_ray sr;
                sr.o = si.p; sr.d = ds.d;
                sr.mint = MO_RAY_EPSILON * (1.0f + mo_hmax_abs(si.p));
                sr.maxt = ds.dist * (1.0f - MO_SHADOW_EPSILON);
                const int occluded = mo_intersect(s, &sr, 1, 0, NULL);
                if (occluded) emitter_val[0] = emitter_val[1] = emitter_val[2] = 0.0f;
                mo_v3 wo = mo_to_local(&si.sh, ds.d);
                if (flip) wo.z = -wo.z;
                if (wi_b.z > 0.0f && wo.z > 0.0f) {
                    float bsdf_pdf = mo_square_to_cosine_hemisphere_pdf(wo);
                    float k = (ds.delta ? 1.0f : mis_weight(ds.pdf, bsdf_pdf)) * (MO_INV_PI * wo.z);
                    for (int c = 0; c < 3; ++c) r->Nc[c] = k * emitter_val[c];
                    /* area light: emitter_val = (Le / pdf_single) * n_emitters for a sample on the front side (scene.cpp:141-189) */
                    if (!occluded && s->emitters[ds.emitter].type == 0 && mo_dot(ds.d, ds.n) < 0.0f) {
                        r->nk = k * mo_rcp(ds.pdf_single) * (s->n_emitters > 1 ? (float) s->n_emitters : 1.0f);
                        r->em_nee = (int) ds.emitter;
                    }
                }
            }
        }
        float s1 = mo_pcg32_next_f32(rng); (void) s1;
        mo_v2 s2; s2.x = mo_pcg32_next_f32(rng); s2.y = mo_pcg32_next_f32(rng);
        mo_v3 bs_wo; float bs_pdf, bsdf_w[3];
        mo_diffuse_sample(refl, wi_b, s2, &bs_wo, &bs_pdf, bsdf_w);
        if (flip) bs_wo.z = -bs_wo.z;
        for (int k = 0; k < 3; ++k) throughput[k] = throughput[k] * bsdf_w[k];
        active = active && (throughput[0] != 0.0f || throughput[1] != 0.0f || throughput[2] != 0.0f);
        if (!active) break;
        ray.o = si.p; ray.d = mo_to_world(&si.sh, bs_wo);
        ray.mint = (1.0f + mo_hmax_abs(si.p)) * MO_RAY_EPSILON;
        ray.maxt = INFINITY;
        mo_si si_bsdf;
        int v2 = scene_intersect(s, &ray, &si_bsdf, &st);
        emitter = v2 ? s->meshes[si_bsdf.shape].emitter : -1;
        if (emitter >= 0) {
            mo_v3 d = mo_sub(si_bsdf.p, si.p);
            float dist = mo_norm(d);
            d = mo_div_s(d, dist);
            emission_weight = mis_weight(bs_pdf, mo_pdf_emitter_direction(s, (uint32_t) emitter, d, si_bsdf.sh.n, dist));
        }
        si = si_bsdf; si_valid = v2;
    }
    return n;
}

/* camera sample i of the wavefront order: its RNG stream (advanced past the camera sample), its ray and
 * delta = dLoss/dRadiance of the sample = sum over its filter footprint of w * dLoss/dImage / (W + 1e-8) (autodiff.py:80-91) */
static void adjoint_sample(const mo_render_desc *d, const camera *cam, const rfilter *f, uint32_t taps, uint64_t i, const float *dimage,
                           const float *film, mo_pcg32 *rng, mo_ray *ray, float delta[3]) {
    seed_wavefront(rng, i, d->base_seed);
    uint64_t pixel = i / (uint64_t) d->spp;
    float px0 = (float) (uint32_t) (pixel % (uint64_t) d->crop_w) + (float) d->crop_x, py0 = (float) (uint32_t) (pixel / (uint64_t) d->crop_w) + (float) d->crop_y;
    float jx = mo_pcg32_next_f32(rng), jy = mo_pcg32_next_f32(rng);
    float psx = px0 + jx, psy = py0 + jy;
    float apx = 0.5f, apy = 0.5f;
    if (cam->aperture_radius > 0.0f) { apx = mo_pcg32_next_f32(rng); apy = mo_pcg32_next_f32(rng); }
    (void) mo_pcg32_next_f32(rng);
    float ax = (psx - (float) d->crop_x) / (float) d->crop_w, ay = (psy - (float) d->crop_y) / (float) d->crop_h;
    camera_sample_ray(cam, ax, ay, apx, apy, ray);
    /* delta = dLoss/dRadiance of this sample */
    delta[0] = delta[1] = delta[2] = 0.0f;
    float px = psx - ((float) d->crop_x + 0.5f), py = psy - ((float) d->crop_y + 0.5f);
    if (f->radius > 1.0f) {
        int lox = (int) ceilf(px - f->radius), loy = (int) ceilf(py - f->radius);
        int hix = (int) floorf(px + f->radius), hiy = (int) floorf(py + f->radius);
        if (lox < 0) lox = 0;
        if (loy < 0) loy = 0;
        if (hix > d->crop_w - 1) hix = d->crop_w - 1;
        if (hiy > d->crop_h - 1) hiy = d->crop_h - 1;
        float bx = (float) (uint32_t) lox - px, by = (float) (uint32_t) loy - py;
        for (uint32_t yr = 0; yr < taps && loy + (int) yr <= hiy; ++yr) {
            float wy = d->filter_analytic ? rfilter_eval(f, by + (float) yr) : rfilter_eval_discretized(f, by + (float) yr);
            for (uint32_t xr = 0; xr < taps && lox + (int) xr <= hix; ++xr) {
                float wx = d->filter_analytic ? rfilter_eval(f, bx + (float) xr) : rfilter_eval_discretized(f, bx + (float) xr);
                size_t pix = (size_t) (loy + (int) yr) * d->crop_w + (size_t) (lox + (int) xr);
                float iw = (wy * wx) / (film[5 * pix + 4] + 1e-8f);
                for (int c = 0; c < 3; ++c) delta[c] += iw * dimage[3 * pix + c];
            }
        }
    } else {
        int lox = (int) ceilf(px - 0.5f), loy = (int) ceilf(py - 0.5f);
        if (lox >= 0 && loy >= 0 && lox < d->crop_w && loy < d->crop_h) {
            size_t pix = (size_t) loy * d->crop_w + (size_t) lox;
            float iw = 1.0f / (film[5 * pix + 4] + 1e-8f);
            for (int c = 0; c < 3; ++c) delta[c] = iw * dimage[3 * pix + c];
        }
    }
}

int mo_render_adjoint(const mo_scene *s, const mo_render_desc *d, const float *dimage, const float *film,
                      float *grad_shape, float *grad_tex, float *grad_emitter) {
    if (desc_check(d) || d->max_depth < 0 || d->max_depth > MO_ADJ_MAX_DEPTH) return -1;
    for (uint32_t m = 0; m < s->n_meshes; ++m)          /* the path replay knows `diffuse` (one- or two-sided) only */
        if (s->meshes[m].bsdf_kind != MO_BSDF_DIFFUSE || s->meshes[m].bsdf.nest) return -2;
    camera cam; camera_init(d, &cam);
    rfilter f; rfilter_init(&f, d->rfilter, d->rfilter_param, d->rfilter_param2);
    uint32_t taps = (uint32_t) ceilf((f.radius - 2.0f * MO_RAY_EPSILON) * 2.0f);
    /* texture gradient offsets: concatenated in index order */
    size_t *toff = (size_t *) calloc(s->n_textures + 1, sizeof(size_t));
    for (uint32_t t = 0; t < s->n_textures; ++t) toff[t + 1] = toff[t] + 3 * (size_t) s->textures[t].w * s->textures[t].h;
    uint64_t total = (uint64_t) d->crop_w * d->crop_h * (uint64_t) d->spp;
    for (uint64_t i = 0; i < total; ++i) {
        mo_pcg32 rng; mo_ray ray; float delta[3];
        adjoint_sample(d, &cam, &f, taps, i, dimage, film, &rng, &ray, delta);
        vertex_rec rec[MO_ADJ_MAX_DEPTH];
        int n = path_sample_rec(s, &rng, &ray, d->max_depth, d->rr_depth, rec);
        /* backward sweep; a = dLoss/dT_v (throughput arriving at vertex v).  q = min(hmax(T) eta^2, .95) is
         * differentiated like Enoki does (gradient flows to the maximal channel when q is not clamped); the
         * survival test itself is not differentiable. */
        float a[3] = { 0, 0, 0 };
        for (int v = n - 1; v >= 0; --v) {
            const vertex_rec *r = &rec[v];
            /* radiance is linear in the emitters' radiance: d/dLe = delta * T_v * ew (emitter hit) + delta * T'_v rho_v nk (emitter sampled) */
            if (grad_emitter && r->em_hit >= 0)
                for (int c = 0; c < 3; ++c) grad_emitter[3 * r->em_hit + c] += delta[c] * r->T[c] * r->ew;
            if (grad_emitter && r->em_nee >= 0)
                for (int c = 0; c < 3; ++c) grad_emitter[3 * r->em_nee + c] += delta[c] * (r->Tp[c] * r->rho[c]) * r->nk;
            if (!r->has_bsdf) { for (int c = 0; c < 3; ++c) a[c] = delta[c] * r->E[c]; continue; }
            float Y[3], g[3], b[3];
            for (int c = 0; c < 3; ++c) { Y[c] = delta[c] * r->Nc[c] + a[c]; g[c] = r->Tp[c] * Y[c]; b[c] = r->rho[c] * Y[c]; }
            if (r->texel != 0xffffffffu) {
                if (grad_tex) {
                    int ti = s->meshes[r->shape].texture;
                    const mo_texture *t = &s->textures[ti];
                    float *gt = grad_tex + toff[ti] + 3 * (size_t) r->texel;
                    float w00 = (1.0f - r->w1[1]) * (1.0f - r->w1[0]), w10 = (1.0f - r->w1[1]) * r->w1[0];
                    float w01 = r->w1[1] * (1.0f - r->w1[0]), w11 = r->w1[1] * r->w1[0];
                    for (int c = 0; c < 3; ++c) {
                        gt[c] += g[c] * w00; gt[3 + c] += g[c] * w10;
                        gt[3 * t->w + c] += g[c] * w01; gt[3 * t->w + 3 + c] += g[c] * w11;
                    }
                }
            } else if (grad_shape) {
                for (int c = 0; c < 3; ++c) grad_shape[3 * r->shape + c] += g[c];
            }
            for (int c = 0; c < 3; ++c) a[c] = delta[c] * r->E[c] + r->invq * b[c];
            if (r->rr_channel >= 0)
                a[r->rr_channel] -= (r->invq * r->invq) * (b[0] * r->T[0] + b[1] * r->T[1] + b[2] * r->T[2]);
        }
    }
    free(toff);
    return 0;
}

/* HDRFilm::bitmap: (X,Y,Z,A) * (1/W), RGB = M * XYZ (hdrfilm.cpp:278-299, struct.cpp:1761-1811) */
void mo_film_develop(const float *xyzaw, uint64_t n, float *rgba) {
    for (uint64_t i = 0; i < n; ++i) {
        const float *p = xyzaw + 5 * i;
        float inv_w = 1.0f / p[4];
        float r = 0.0f, g = 0.0f, b = 0.0f;
        r += 3.240479f * p[0]; r += -1.537150f * p[1]; r += -0.498535f * p[2];
        g += -0.969256f * p[0]; g += 1.875991f * p[1]; g += 0.041556f * p[2];
        b += 0.055648f * p[0]; b += -0.204043f * p[1]; b += 1.057311f * p[2];
        rgba[4 * i] = r * inv_w; rgba[4 * i + 1] = g * inv_w; rgba[4 * i + 2] = b * inv_w;
        rgba[4 * i + 3] = p[3] * inv_w;
    }
}

/* ================================================================== */
/* known-answer entry points */
uint32_t mo_kat_tea32(uint32_t v0, uint32_t v1, int rounds) { return mo_tea32(v0, v1, rounds); }
uint64_t mo_kat_tea64_u32(uint32_t v0, uint32_t v1, int rounds) { return mo_tea64_u32(v0, v1, rounds); }
uint64_t mo_kat_tea64_u64(uint64_t v0, uint64_t v1, int rounds) { return mo_tea64_u64(v0, v1, rounds); }
float mo_kat_tea_float32(uint32_t v0, uint32_t v1, int rounds) { return mo_tea_float32(v0, v1, rounds); }
double mo_kat_tea_float64(uint32_t v0, uint32_t v1, int rounds) { return mo_tea_float64(v0, v1, rounds); }

void mo_kat_pcg32(uint64_t initstate, uint64_t initseq, int n, uint32_t *out_u32, float *out_f32) {
    mo_pcg32 r; mo_pcg32_seed(&r, initstate, initseq);
    mo_pcg32 r2 = r;
    for (int i = 0; i < n; ++i) {
        if (out_u32) out_u32[i] = mo_pcg32_next_u32(&r);
        if (out_f32) out_f32[i] = mo_pcg32_next_f32(&r2);
    }
}

void mo_kat_warp(int which, uint64_t n, const float *sx, const float *sy, float *out3) {
    for (uint64_t i = 0; i < n; ++i) {
        mo_v2 s = { sx[i], sy[i] };
        float *o = out3 + 3 * i;
        if (which == 0) { mo_v2 p = mo_square_to_uniform_disk_concentric(s); o[0] = p.x; o[1] = p.y; o[2] = 0; }
        else if (which == 1) { mo_v3 p = mo_square_to_cosine_hemisphere(s); o[0] = p.x; o[1] = p.y; o[2] = p.z; }
        else { mo_v2 p = mo_square_to_uniform_triangle(s); o[0] = p.x; o[1] = p.y; o[2] = 0; }
    }
}

void mo_kat_coordinate_system(const float *n3, float *s3, float *t3) {
    mo_v3 s, t; mo_coordinate_system(mo_v3_make(n3[0], n3[1], n3[2]), &s, &t);
    s3[0] = s.x; s3[1] = s.y; s3[2] = s.z; t3[0] = t.x; t3[1] = t.y; t3[2] = t.z;
}

void mo_kat_morton(uint32_t n, uint32_t *xy) {
    for (uint32_t i = 0; i < n; ++i) mo_morton_decode2(i, &xy[2 * i], &xy[2 * i + 1]);
}

float mo_kat_distr(uint32_t n, const float *pmf, float *cdf, uint32_t nv, const float *values,
                   uint32_t *idx, float *reused) {
    float sum, norm; uint32_t lo, hi;
    if (mo_distr_build(n, pmf, cdf, &sum, &norm, &lo, &hi)) return -1.0f;
    for (uint32_t i = 0; i < nv; ++i) {
        idx[i] = mo_distr_sample(cdf, sum, lo, hi, values[i]);
        if (reused) mo_distr_sample_reuse(pmf, cdf, sum, norm, lo, hi, values[i], &reused[i]);
    }
    return sum;
}

void mo_kat_diffuse(const float *refl, const float *wi3, const float *wo3, const float *sample2,
                    float *eval3, float *pdf, float *s_wo3, float *s_pdf, float *s_weight3) {
    mo_v3 wi = mo_v3_make(wi3[0], wi3[1], wi3[2]), wo = mo_v3_make(wo3[0], wo3[1], wo3[2]);
    mo_diffuse_eval_pdf(refl, wi, wo, eval3, pdf);
    mo_v2 s2 = { sample2[0], sample2[1] };
    mo_v3 swo; mo_diffuse_sample(refl, wi, s2, &swo, s_pdf, s_weight3);
    s_wo3[0] = swo.x; s_wo3[1] = swo.y; s_wo3[2] = swo.z;
}

void mo_kat_sample_emitter(const mo_scene *s, const float *ref_p3, const float *sample2, float *out) {
    mo_dsample ds; float spec[3];
    mo_v2 s2 = { sample2[0], sample2[1] };
    mo_sample_emitter_direction(s, mo_v3_make(ref_p3[0], ref_p3[1], ref_p3[2]), s2, &ds, spec);
    out[0] = ds.d.x; out[1] = ds.d.y; out[2] = ds.d.z; out[3] = ds.dist; out[4] = ds.pdf;
    out[5] = ds.n.x; out[6] = ds.n.y; out[7] = ds.n.z; out[8] = ds.p.x; out[9] = ds.p.y; out[10] = ds.p.z;
    out[11] = spec[0]; out[12] = spec[1]; out[13] = spec[2];
    out[14] = s->n_emitters ? mo_pdf_emitter_direction(s, ds.emitter, ds.d, ds.n, ds.dist) : 0.0f;
}

int mo_render_adjoint_envmap(const mo_scene *s, const mo_render_desc *d, const float *dimage, const float *film, float *grad_env) {
    if (desc_check(d) || !grad_env || s->spectral || s->environment < 0 || s->emitters[s->environment].type != 2) return -1;
    camera cam; camera_init(d, &cam);
    rfilter f; rfilter_init(&f, d->rfilter, d->rfilter_param, d->rfilter_param2);
    uint32_t taps = (uint32_t) ceilf((f.radius - 2.0f * MO_RAY_EPSILON) * 2.0f);
    uint64_t total = (uint64_t) d->crop_w * d->crop_h * (uint64_t) d->spp;
    ray_stats st = { 0, 0 };
    for (uint64_t i = 0; i < total; ++i) {
        mo_pcg32 rng; mo_ray ray; float delta[3], L[3];
        adjoint_sample(d, &cam, &f, taps, i, dimage, film, &rng, &ray, delta);
        env_grad eg = { delta, grad_env };
        int valid;
        path_sample(s, &rng, &ray, d->max_depth, d->rr_depth, L, &valid, &st, &eg, NULL);
    }
    return 0;
}

/* the checker of mtsamd_render_adjoint_param: d(loss)/d(one scalar BSDF parameter), summed over every camera sample (double sum) */
int mo_render_adjoint_param(const mo_scene *s, const mo_render_desc *d, const float *dimage, const float *film, const uint8_t *shape_mask,
                            int kind, int comp, float h, double *grad) {
    if (desc_check(d) || !grad || !shape_mask || s->spectral || !(h > 0.0f) || kind < 0 || kind > 5 || comp < 0 || comp > 2) return -1;
    camera cam; camera_init(d, &cam);
    rfilter f; rfilter_init(&f, d->rfilter, d->rfilter_param, d->rfilter_param2);
    uint32_t taps = (uint32_t) ceilf((f.radius - 2.0f * MO_RAY_EPSILON) * 2.0f);
    uint64_t total = (uint64_t) d->crop_w * d->crop_h * (uint64_t) d->spp;
    double sum = 0.0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : sum)
    for (int64_t i = 0; i < (int64_t) total; ++i) {
        mo_pcg32 rng; mo_ray ray; float delta[3], L[3];
        ray_stats st = { 0, 0 };
        adjoint_sample(d, &cam, &f, taps, (uint64_t) i, dimage, film, &rng, &ray, delta);
        param_grad pg = { shape_mask, kind, comp, h, { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f } };
        int valid;
        path_sample(s, &rng, &ray, d->max_depth, d->rr_depth, L, &valid, &st, NULL, &pg);
        const float g = fmaf(delta[2], pg.dres[2], fmaf(delta[1], pg.dres[1], delta[0] * pg.dres[0]));
        if (isfinite(g)) sum += (double) g;
    }
    *grad = sum;
    return 0;
}

void mo_libm_eval(int fn, uint64_t n, const float *x, const float *y, float *out) {
    for (uint64_t i = 0; i < n; ++i) {
        const float a = x[i];
        switch (fn) {
        case 0: out[i] = mo_lm_sin(a); break;
        case 1: out[i] = mo_lm_cos(a); break;
        case 2: out[i] = mo_lm_tan(a); break;
        case 3: out[i] = mo_lm_exp(a); break;
        case 4: out[i] = mo_lm_log(a); break;
        case 5: out[i] = mo_lm_erf(a); break;
        case 6: out[i] = mo_lm_acos(a); break;
        case 8: out[i] = mo_lm_atanh(a); break;
        case 9: out[i] = mo_lm_cosh(a); break;
        default: out[i] = mo_lm_atan2(a, y[i]); break;
        }
    }
}
