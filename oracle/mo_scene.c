/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_math.h header for scope and pinning).
 *
 * Scene queries of the hot path, restated from:
 *   Mesh::ray_intersect_triangle     include/mitsuba/render/mesh.h:195-221
 *   ShapeKDTree::ray_intersect_naive include/mitsuba/render/kdtree.h:2303-2328
 *   create_surface_interaction       include/mitsuba/render/kdtree.h:2334-2367
 *   Mesh::fill_surface_interaction   src/librender/mesh.cpp:399-462
 *   Mesh::area_distr_build / sample_position   src/librender/mesh.cpp:284-365
 *   DiscreteDistribution             include/mitsuba/core/distr_1d.h:49-203
 *   Shape::sample_direction / pdf_direction    src/librender/shape.cpp:252-283
 *   AreaLight                        src/emitters/area.cpp:71-125
 *   Scene::sample_emitter_direction / pdf_emitter_direction  src/librender/scene.cpp:141-206
 *   SmoothDiffuse                    src/bsdfs/diffuse.cpp:78-135
 *
 * The reference's SAH kd-tree (kdtree.h:676-1881) is NOT restated: only its
 * query result is the contract (closest t / any hit).  The oracle answers
 * queries by brute force (the reference's own test oracle, ray_intersect_naive)
 * or, for speed on large meshes, by a private median-split BVH whose slab tests
 * run in double precision with padded boxes, checked against brute force in
 * tests/.  Tie rule among primitives with exactly equal t: the highest global
 * primitive index wins (= what the brute-force loop order produces).
 */
#include "mo_internal.h"
#include <stdlib.h>
#include <stdio.h>

/* ------------------------------------------------------------------ */
mo_scene *mo_scene_new(void) {
    mo_scene *s = (mo_scene *) calloc(1, sizeof(mo_scene));
    if (s) s->environment = -1;
    return s;
}

void mo_scene_free(mo_scene *s) {
    if (!s) return;
    for (uint32_t i = 0; i < s->n_meshes; ++i) {
        mo_mesh *m = &s->meshes[i];
        free(m->pos); free(m->nrm); free(m->uv); free(m->faces);
        free(m->area_pmf); free(m->area_cdf);
        free(m->bsdf.child[0]); free(m->bsdf.child[1]);
    }
    for (uint32_t i = 0; i < s->n_textures; ++i) free(s->textures[i].data);
    free(s->textures);
    free(s->meshes); free(s->emitters); free(s->prim_shape); free(s->prim_local);
    free(s->bvh_nodes); free(s->bvh_prims);
    free(s);
}

static void *dup_mem(const void *p, size_t bytes) {
    if (!p) return NULL;
    void *q = malloc(bytes ? bytes : 1);
    memcpy(q, p, bytes);
    return q;
}

int mo_scene_add_mesh(mo_scene *s, uint32_t n_verts, const float *positions, const float *normals,
                      const float *texcoords, uint32_t n_faces, const uint32_t *faces, int bsdf_kind,
                      const float *reflectance_rgb, const float *emitter_rgb) {
    if (!s || !positions || !faces || n_faces == 0 || bsdf_kind != 0) return -1;
    for (uint32_t i = 0; i < 3 * n_faces; ++i)
        if (faces[i] >= n_verts) return -2;
    s->meshes = (mo_mesh *) realloc(s->meshes, sizeof(mo_mesh) * (s->n_meshes + 1));
    mo_mesh *m = &s->meshes[s->n_meshes];
    memset(m, 0, sizeof(*m));
    m->n_verts = n_verts; m->n_faces = n_faces;
    m->pos = (float *) dup_mem(positions, sizeof(float) * 3 * n_verts);
    m->nrm = (float *) dup_mem(normals, sizeof(float) * 3 * n_verts);
    m->uv = (float *) dup_mem(texcoords, sizeof(float) * 2 * n_verts);
    m->faces = (uint32_t *) dup_mem(faces, sizeof(uint32_t) * 3 * n_faces);
    m->bsdf_kind = bsdf_kind;
    for (int k = 0; k < 3; ++k) m->refl[k] = reflectance_rgb ? reflectance_rgb[k] : 0.5f;
    m->bsdf.d.type = MO_BSDF_DIFFUSE;
    for (int k = 0; k < 3; ++k) {
        m->bsdf.d.reflectance[k] = m->refl[k];
        m->bsdf.d.specular_reflectance[k] = m->bsdf.d.specular_transmittance[k] = 1.0f;
    }
    mo_bsdf_prepare(&m->bsdf);
    m->emitter = -1; m->texture = -1;
    if (emitter_rgb) {
        s->emitters = (mo_emitter *) realloc(s->emitters, sizeof(mo_emitter) * (s->n_emitters + 1));
        mo_emitter *e = &s->emitters[s->n_emitters];
        memset(e, 0, sizeof(*e));
        e->shape = s->n_meshes;
        for (int k = 0; k < 3; ++k) e->radiance[k] = emitter_rgb[k];
        m->emitter = (int) s->n_emitters++;
    }
    return (int) s->n_meshes++;
}

int mo_scene_add_constant_emitter(mo_scene *s, const float *rgb) {
    if (!s || !rgb || s->environment >= 0) return -1;        /* "Only one environment emitter can be specified per scene." */
    s->emitters = (mo_emitter *) realloc(s->emitters, sizeof(mo_emitter) * (s->n_emitters + 1));
    mo_emitter *e = &s->emitters[s->n_emitters];
    memset(e, 0, sizeof(*e));
    e->type = 1; e->shape = 0xffffffffu;
    for (int k = 0; k < 3; ++k) e->radiance[k] = rgb[k];
    e->radius = 1.0f;                                        /* unit sphere until set_scene (constant.cpp:40-42) */
    s->environment = (int) s->n_emitters;
    return (int) s->n_emitters++;
}
int mo_scene_add_envmap_emitter(mo_scene *s, int w, int h, const float *rgb, float scale, const float *to_world9) {
    if (!s || !rgb || s->environment >= 0) return -1;
    mo_envmap *env = (mo_envmap *) malloc(sizeof(mo_envmap));
    if (mo_envmap_init(env, w, h, rgb, scale, to_world9)) { free(env); return -2; }
    s->emitters = (mo_emitter *) realloc(s->emitters, sizeof(mo_emitter) * (s->n_emitters + 1));
    mo_emitter *e = &s->emitters[s->n_emitters];
    memset(e, 0, sizeof(*e));
    e->type = 2; e->shape = 0xffffffffu; e->env = env; e->radius = 1.0f;
    s->environment = (int) s->n_emitters;
    return (int) s->n_emitters++;
}
int mo_scene_update_envmap(mo_scene *s, const float *rgb, int rebuild_warp) {
    if (!s || !rgb || s->environment < 0 || s->emitters[s->environment].type != 2 || s->spectral) return -1;
    return mo_envmap_update(s->emitters[s->environment].env, rgb, rebuild_warp);
}
int mo_scene_add_delta_emitter(mo_scene *s, int type, const float *rgb, const float *position3, const float *direction3,
                               const float *to_world9, float cutoff_angle_deg, float beam_width_deg) {
    if (!s || !rgb || type < 3 || type > 5) return -1;
    s->emitters = (mo_emitter *) realloc(s->emitters, sizeof(mo_emitter) * (s->n_emitters + 1));
    mo_emitter *e = &s->emitters[s->n_emitters];
    memset(e, 0, sizeof(*e));
    e->type = type; e->shape = 0xffffffffu; e->radius = 1.0f;
    for (int k = 0; k < 3; ++k) e->radiance[k] = rgb[k];
    if (position3) e->pos = mo_v3_make(position3[0], position3[1], position3[2]);
    if (direction3) e->dir = mo_v3_make(direction3[0], direction3[1], direction3[2]);
    if (type == 4) {
        static const float ident[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        const float *m = to_world9 ? to_world9 : ident;
        double a = m[0], b = m[1], c = m[2], d = m[3], ee = m[4], f = m[5], g = m[6], hh = m[7], i = m[8];
        double det = a * (ee * i - f * hh) - b * (d * i - f * g) + c * (d * hh - ee * g);
        double inv[9] = { (ee * i - f * hh) / det, (c * hh - b * i) / det, (b * f - c * ee) / det,
                          (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                          (d * hh - ee * g) / det, (b * g - a * hh) / det, (a * ee - b * d) / det };
        for (int k = 0; k < 9; ++k) e->to_local[k] = (float) inv[k];
        /* spot.cpp:81-90 */
        float cutoff = cutoff_angle_deg * (MO_PI_F / 180.0f), beam = beam_width_deg * (MO_PI_F / 180.0f);
        e->cutoff_angle = cutoff;
        e->inv_transition = 1.0f / (cutoff - beam);
        e->cos_cutoff = cosf(cutoff); e->cos_beam = cosf(beam);
    }
    return (int) s->n_emitters++;
}
int mo_scene_set_emitter_order(mo_scene *s, uint32_t n, const uint32_t *order) {
    if (!s || n != s->n_emitters) return -1;
    mo_emitter *ne = (mo_emitter *) malloc(sizeof(mo_emitter) * (n ? n : 1));
    for (uint32_t i = 0; i < n; ++i) { if (order[i] >= n) { free(ne); return -1; } ne[i] = s->emitters[order[i]]; }
    free(s->emitters); s->emitters = ne;
    s->environment = -1;
    for (uint32_t i = 0; i < n; ++i) {
        if (ne[i].type == 1 || ne[i].type == 2) s->environment = (int) i;
        else if (ne[i].type == 0) s->meshes[ne[i].shape].emitter = (int) i;
    }
    return 0;
}

int mo_scene_set_emitter_radiance(mo_scene *s, uint32_t emitter, const float *rgb) {
    if (!s || emitter >= s->n_emitters || !rgb) return -1;
    for (int k = 0; k < 3; ++k) s->emitters[emitter].radiance[k] = rgb[k];
    return 0;
}

int mo_scene_add_texture(mo_scene *s, int width, int height, const float *rgb) {
    if (!s || width < 2 || height < 2 || !rgb) return -1;
    s->textures = (mo_texture *) realloc(s->textures, sizeof(mo_texture) * (s->n_textures + 1));
    mo_texture *t = &s->textures[s->n_textures];
    memset(t, 0, sizeof(*t));
    t->w = width; t->h = height;
    t->data = (float *) dup_mem(rgb, sizeof(float) * 3 * (size_t) width * height);
    t->uvm[0] = 1.0f; t->uvm[4] = 1.0f;
    mo_texture_update_mean(s, (int) s->n_textures++);
    return (int) s->n_textures - 1;
}
/* Texture::mean() in the RGB variant: mean luminance of a bitmap (bitmap.cpp:124-136), mean of the two colours' channel
 * means for a checkerboard (checkerboard.cpp:88-90, srgb.cpp:52-57); plastic BSDFs that use the texture follow */
void mo_texture_update_mean(mo_scene *s, int texture) {
    mo_texture *t = &s->textures[texture];
    if (t->kind == 1) {
        float m0 = (t->color0[0] + t->color0[1] + t->color0[2]) * (1.0f / 3.0f), m1 = (t->color1[0] + t->color1[1] + t->color1[2]) * (1.0f / 3.0f);
        t->mean = 0.5f * (m0 + m1);
    } else {
        double mean = 0.0;
        for (size_t i = 0; i < (size_t) t->w * t->h; ++i) {
            const float *p = t->data + 3 * i;
            mean += (double) (p[0] * 0.212671f + p[1] * 0.715160f + p[2] * 0.072169f);
        }
        t->mean = (float) (mean / (double) ((size_t) t->w * t->h));
    }
    for (uint32_t i = 0; i < s->n_meshes; ++i) if (s->meshes[i].texture == texture) mo_scene_set_texture(s, i, texture);
}
int mo_scene_set_texture_transform(mo_scene *s, uint32_t texture, const float *uvm6) {
    if (!s || texture >= s->n_textures || !uvm6) return -1;
    memcpy(s->textures[texture].uvm, uvm6, sizeof(float) * 6);
    return 0;
}
int mo_scene_add_checkerboard(mo_scene *s, const float *color0, const float *color1, const float *uvm6) {
    if (!s || !color0 || !color1) return -1;
    s->textures = (mo_texture *) realloc(s->textures, sizeof(mo_texture) * (s->n_textures + 1));
    mo_texture *t = &s->textures[s->n_textures];
    memset(t, 0, sizeof(*t));
    t->kind = 1;
    t->uvm[0] = 1.0f; t->uvm[4] = 1.0f;
    if (uvm6) memcpy(t->uvm, uvm6, sizeof(float) * 6);
    for (int k = 0; k < 3; ++k) { t->color0[k] = color0[k]; t->color1[k] = color1[k]; }
    mo_texture_update_mean(s, (int) s->n_textures++);
    return (int) s->n_textures - 1;
}
int mo_scene_set_texture(mo_scene *s, uint32_t shape, int texture) {
    if (!s || shape >= s->n_meshes || texture >= (int) s->n_textures) return -1;
    mo_mesh *m = &s->meshes[shape];
    m->texture = texture;
    if (m->bsdf.nest) m->bsdf.weight_lum = texture >= 0 && s->textures[texture].kind == 0;      /* eval_1 of a bitmap: luminance */
    if (texture >= 0 && (m->bsdf.d.type == MO_BSDF_PLASTIC || m->bsdf.d.type == MO_BSDF_ROUGHPLASTIC)) {
        /* plastic.cpp:170-175: specular sampling weight from Texture::mean() of both reflectances */
        const float *sr = m->bsdf.d.specular_reflectance;
        float d_mean = s->textures[texture].mean, s_mean = (sr[0] + sr[1] + sr[2]) * (1.0f / 3.0f);
        m->bsdf.spec_weight = s_mean / (d_mean + s_mean);
    }
    return 0;
}
int mo_scene_update_texture(mo_scene *s, uint32_t texture, const float *rgb) {
    if (!s || texture >= s->n_textures) return -1;
    mo_texture *t = &s->textures[texture];
    if (t->kind != 0) return -2;
    if (s->spectral) return -3;                                       /* the texels hold model coefficients */
    memcpy(t->data, rgb, sizeof(float) * 3 * (size_t) t->w * t->h);
    mo_texture_update_mean(s, (int) texture);                         /* bitmap.cpp:308-322 */
    return 0;
}
int mo_scene_set_reflectance(mo_scene *s, uint32_t shape, const float *rgb) {
    if (!s || shape >= s->n_meshes) return -1;
    for (int k = 0; k < 3; ++k) s->meshes[shape].refl[k] = s->meshes[shape].bsdf.d.reflectance[k] = rgb[k];
    mo_bsdf_prepare(&s->meshes[shape].bsdf);
    return 0;
}
int mo_scene_set_bsdf(mo_scene *s, uint32_t shape, const mo_bsdf_desc *desc) {
    if (!s || shape >= s->n_meshes || !desc || desc->type < MO_BSDF_DIFFUSE || desc->type > MO_BSDF_THINDIELECTRIC) return -1;
    if (s->spectral) return -2;                                       /* set the BSDFs before mo_scene_set_spectral */
    mo_mesh *m = &s->meshes[shape];
    m->bsdf.d = *desc;
    m->bsdf_kind = desc->type;
    for (int k = 0; k < 3; ++k) m->refl[k] = desc->reflectance[k];
    mo_bsdf_prepare(&m->bsdf);
    return 0;
}

/* blendbsdf / mask over plain children (see mo_api.h).  The weight takes the place of the shape's reflectance: constant, or the
 * texture attached with mo_scene_set_texture afterwards. */
int mo_scene_set_nested_bsdf(mo_scene *s, uint32_t shape, int kind, float weight, int twosided, const mo_bsdf_desc *child0, const mo_bsdf_desc *child1,
                             int child_tex0, int child_tex1) {
    if (!s || shape >= s->n_meshes || !child0 || (kind != MO_NEST_BLEND && kind != MO_NEST_MASK)) return -1;
    if ((kind == MO_NEST_BLEND) != (child1 != NULL) || (kind == MO_NEST_MASK && twosided)) return -1;
    if (s->spectral) return -2;
    const mo_bsdf_desc *cd[2] = { child0, child1 };
    for (int k = 0; k < 2; ++k)
        if (cd[k] && (cd[k]->type < MO_BSDF_DIFFUSE || cd[k]->type > MO_BSDF_THINDIELECTRIC)) return -1;
    mo_mesh *m = &s->meshes[shape];
    free(m->bsdf.child[0]); free(m->bsdf.child[1]);
    memset(&m->bsdf, 0, sizeof(m->bsdf));
    if (child_tex0 >= (int) s->n_textures || child_tex1 >= (int) s->n_textures) return -1;
    m->bsdf.nest = kind; m->bsdf.d.twosided = twosided; m->bsdf.d.type = -kind; m->bsdf.d.uniform_mask = 1;
    m->bsdf.child_tex[0] = child_tex0 < 0 ? -1 : child_tex0; m->bsdf.child_tex[1] = child_tex1 < 0 ? -1 : child_tex1;
    for (int k = 0; k < 2; ++k) {
        if (!cd[k]) continue;
        mo_bsdf *c = (mo_bsdf *) calloc(1, sizeof(mo_bsdf));
        if (!c) return -1;
        c->d = *cd[k]; mo_bsdf_prepare(c);
        const int ct = k ? child_tex1 : child_tex0;
        if (ct >= 0 && (c->d.type == MO_BSDF_PLASTIC || c->d.type == MO_BSDF_ROUGHPLASTIC)) {      /* plastic.cpp:170-175 with Texture::mean() */
            const float *sr = c->d.specular_reflectance;
            float d_mean = s->textures[ct].mean, s_mean = (sr[0] + sr[1] + sr[2]) * (1.0f / 3.0f);
            c->spec_weight = s_mean / (d_mean + s_mean);
        }
        m->bsdf.child[k] = c;
    }
    m->bsdf_kind = 100 + kind;                          /* not `diffuse`: the adjoint pass rejects the scene */
    for (int k = 0; k < 3; ++k) m->refl[k] = m->bsdf.d.reflectance[k] = weight;
    m->bsdf.weight_lum = m->texture >= 0 && s->textures[m->texture].kind == 0;
    return 0;
}

/* BitmapTextureImpl::interpolate (bitmap.cpp:250-293), identity to_uv */
static void texture_lookup(const mo_scene *s, int texture, const float *constant, mo_v2 uv, float out[3], uint32_t *texel, float w1o[2]);
void mo_reflectance(const mo_scene *s, const mo_mesh *m, mo_v2 uv, float out[3], uint32_t *texel, float w1o[2]) {
    texture_lookup(s, m->texture, m->refl, uv, out, texel, w1o);
}
/* everything BSDF::sample / eval read from textures at a surface point: out9[0..2] = the shape's reflectance (or, for a blendbsdf /
 * mask, what Texture::eval_1 of the weight reads), out9[3..5] / [6..8] = the reflectances of the children of a blendbsdf / mask */
void mo_surface_reflectance(const mo_scene *s, const mo_mesh *m, mo_v2 uv, float out9[9]) {
    texture_lookup(s, m->texture, m->refl, uv, out9, NULL, NULL);
    for (int k = 0; k < 2; ++k) {
        const mo_bsdf *c = m->bsdf.nest ? m->bsdf.child[k] : NULL;
        if (c) texture_lookup(s, m->bsdf.child_tex[k], c->d.reflectance, uv, out9 + 3 + 3 * k, NULL, NULL);
        else out9[3 + 3 * k] = out9[4 + 3 * k] = out9[5 + 3 * k] = 0.0f;
    }
}
static void texture_lookup(const mo_scene *s, int texture, const float *constant, mo_v2 uv, float out[3], uint32_t *texel, float w1o[2]) {
    if (texture < 0) {
        for (int k = 0; k < 3; ++k) out[k] = constant[k];
        if (texel) *texel = 0xffffffffu;
        return;
    }
    const mo_texture *t = &s->textures[texture];
    {   /* m_transform.transform_affine(si.uv) (bitmap.cpp:254, checkerboard.cpp:49) */
        float u2 = fmaf(t->uvm[0], uv.x, fmaf(t->uvm[1], uv.y, t->uvm[2])), v2 = fmaf(t->uvm[3], uv.x, fmaf(t->uvm[4], uv.y, t->uvm[5]));
        uv.x = u2; uv.y = v2;
    }
    if (t->kind == 1) {                                      /* checkerboard.cpp:46-63 */
        int mx = (uv.x - floorf(uv.x)) > 0.5f, my = (uv.y - floorf(uv.y)) > 0.5f;
        const float *c = mx == my ? t->color0 : t->color1;
        for (int k = 0; k < 3; ++k) out[k] = c[k];
        if (texel) *texel = 0xffffffffu;
        return;
    }
    float ux = uv.x - floorf(uv.x), uy = uv.y - floorf(uv.y);
    ux *= (float) (uint32_t) (t->w - 1); uy *= (float) (uint32_t) (t->h - 1);
    uint32_t px = (uint32_t) ux, py = (uint32_t) uy;
    if (px > (uint32_t) (t->w - 2)) px = (uint32_t) (t->w - 2);
    if (py > (uint32_t) (t->h - 2)) py = (uint32_t) (t->h - 2);
    float w1x = ux - (float) px, w1y = uy - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    uint32_t index = px + py * (uint32_t) t->w, width = (uint32_t) t->w;
    const float *v00 = t->data + 3 * (size_t) index, *v10 = v00 + 3, *v01 = t->data + 3 * (size_t) (index + width), *v11 = v01 + 3;
    for (int k = 0; k < 3; ++k) {
        float v0 = fmaf(w0x, v00[k], w1x * v10[k]), v1 = fmaf(w0x, v01[k], w1x * v11[k]);
        out[k] = fmaf(w0y, v0, w1y * v1);
    }
    if (texel) *texel = index;
    if (w1o) { w1o[0] = w1x; w1o[1] = w1y; }
}

/* spectral variant of the lookup: bitmap texels hold srgb model coefficients which are evaluated at the four corners and
 * then interpolated (bitmap.cpp:274-286); checkerboard colours are `srgb` spectra (checkerboard.cpp:46-63) */
void mo_reflectance_spectral(const mo_scene *s, const mo_mesh *m, mo_v2 uv, const float *wav, float *out) {
    const mo_texture *t = &s->textures[m->texture];
    float u2 = fmaf(t->uvm[0], uv.x, fmaf(t->uvm[1], uv.y, t->uvm[2])), v2 = fmaf(t->uvm[3], uv.x, fmaf(t->uvm[4], uv.y, t->uvm[5]));
    uv.x = u2; uv.y = v2;
    if (t->kind == 1) {
        int mx = (uv.x - floorf(uv.x)) > 0.5f, my = (uv.y - floorf(uv.y)) > 0.5f;
        const float *c = mx == my ? t->coeff0 : t->coeff1;
        for (int k = 0; k < MO_WAV; ++k) out[k] = mo_srgb_model_eval(c, wav[k]);
        return;
    }
    float ux = uv.x - floorf(uv.x), uy = uv.y - floorf(uv.y);
    ux *= (float) (uint32_t) (t->w - 1); uy *= (float) (uint32_t) (t->h - 1);
    uint32_t px = (uint32_t) ux, py = (uint32_t) uy;
    if (px > (uint32_t) (t->w - 2)) px = (uint32_t) (t->w - 2);
    if (py > (uint32_t) (t->h - 2)) py = (uint32_t) (t->h - 2);
    float w1x = ux - (float) px, w1y = uy - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    uint32_t index = px + py * (uint32_t) t->w, width = (uint32_t) t->w;
    const float *v00 = t->data + 3 * (size_t) index, *v10 = v00 + 3, *v01 = t->data + 3 * (size_t) (index + width), *v11 = v01 + 3;
    for (int k = 0; k < MO_WAV; ++k) {
        float c00 = mo_srgb_model_eval(v00, wav[k]), c10 = mo_srgb_model_eval(v10, wav[k]);
        float c01 = mo_srgb_model_eval(v01, wav[k]), c11 = mo_srgb_model_eval(v11, wav[k]);
        float c0 = fmaf(w0x, c00, w1x * c10), c1 = fmaf(w0x, c01, w1x * c11);
        out[k] = fmaf(w0y, c0, w1y * c1);
    }
}

static inline mo_v3 vtx(const mo_mesh *m, uint32_t i) {
    return mo_v3_make(m->pos[3 * i], m->pos[3 * i + 1], m->pos[3 * i + 2]);
}
static inline mo_v3 vnrm(const mo_mesh *m, uint32_t i) {
    return mo_v3_make(m->nrm[3 * i], m->nrm[3 * i + 1], m->nrm[3 * i + 2]);
}

/* mesh.h:108-116 */
static float face_area(const mo_mesh *m, uint32_t f) {
    mo_v3 p0 = vtx(m, m->faces[3 * f]), p1 = vtx(m, m->faces[3 * f + 1]), p2 = vtx(m, m->faces[3 * f + 2]);
    return 0.5f * mo_norm(mo_cross(mo_sub(p1, p0), mo_sub(p2, p0)));
}

/* distr_1d.h:49-88 (cdf accumulated in double, stored as float) */
int mo_distr_build(uint32_t n, const float *pmf, float *cdf, float *sum_out, float *norm_out,
                   uint32_t *valid_lo, uint32_t *valid_hi) {
    double sum = 0.0;
    uint32_t lo = 0xffffffffu, hi = 0xffffffffu;
    for (uint32_t i = 0; i < n; ++i) {
        double value = (double) pmf[i];
        sum += value;
        cdf[i] = (float) sum;
        if (value < 0.0) return -1;
        if (value > 0.0) { if (lo == 0xffffffffu) lo = i; hi = i; }
    }
    if (lo == 0xffffffffu) return -2;
    *sum_out = (float) sum;
    *norm_out = (float) (1.0 / sum);
    *valid_lo = lo; *valid_hi = hi;
    return 0;
}

/* distr_1d.h:144-154: enoki::binary_search over [valid.x, valid.y] */
uint32_t mo_distr_sample(const float *cdf, float sum, uint32_t lo, uint32_t hi, float value) {
    value *= sum;
    uint32_t start = lo, end = hi;
    while (start < end) {
        uint32_t middle = (start + end) >> 1;
        if (cdf[middle] < value) { start = middle + 1; if (start > end) start = end; }
        else end = middle;
    }
    return start;
}

/* distr_1d.h:193-203 */
uint32_t mo_distr_sample_reuse(const float *pmf, const float *cdf, float sum, float norm,
                               uint32_t lo, uint32_t hi, float value, float *reused) {
    uint32_t index = mo_distr_sample(cdf, sum, lo, hi, value);
    float p = pmf[index] * norm;
    float c = index > 0 ? cdf[index - 1] * norm : 0.0f;
    *reused = (value - c) / p;
    return index;
}

/* ------------------------------------------------------------------ */
/* private BVH (double-precision conservative slab test) */
typedef struct { double lo[3], hi[3]; } bbox_d;

static void prim_bbox(const mo_scene *s, uint32_t gp, bbox_d *b) {
    const mo_mesh *m = &s->meshes[s->prim_shape[gp]];
    uint32_t f = s->prim_local[gp];
    for (int k = 0; k < 3; ++k) { b->lo[k] = 1e300; b->hi[k] = -1e300; }
    for (int j = 0; j < 3; ++j) {
        uint32_t vi = m->faces[3 * f + j];
        for (int k = 0; k < 3; ++k) {
            double c = m->pos[3 * vi + k];
            if (c < b->lo[k]) b->lo[k] = c;
            if (c > b->hi[k]) b->hi[k] = c;
        }
    }
}

static mo_scene *g_sort_scene; static int g_sort_axis;
static int cmp_centroid(const void *a, const void *b) {
    bbox_d ba, bb;
    prim_bbox(g_sort_scene, *(const uint32_t *) a, &ba);
    prim_bbox(g_sort_scene, *(const uint32_t *) b, &bb);
    double ca = ba.lo[g_sort_axis] + ba.hi[g_sort_axis], cb = bb.lo[g_sort_axis] + bb.hi[g_sort_axis];
    if (ca < cb) return -1;
    if (ca > cb) return 1;
    uint32_t ia = *(const uint32_t *) a, ib = *(const uint32_t *) b;
    return ia < ib ? -1 : (ia > ib ? 1 : 0);
}

static uint32_t bvh_build(mo_scene *s, uint32_t first, uint32_t count) {
    uint32_t idx = s->n_bvh_nodes++;
    mo_bvh_node *n = &s->bvh_nodes[idx];
    bbox_d bb; for (int k = 0; k < 3; ++k) { bb.lo[k] = 1e300; bb.hi[k] = -1e300; }
    for (uint32_t i = 0; i < count; ++i) {
        bbox_d pb; prim_bbox(s, s->bvh_prims[first + i], &pb);
        for (int k = 0; k < 3; ++k) {
            if (pb.lo[k] < bb.lo[k]) bb.lo[k] = pb.lo[k];
            if (pb.hi[k] > bb.hi[k]) bb.hi[k] = pb.hi[k];
        }
    }
    for (int k = 0; k < 3; ++k) {
        double pad = 1e-5 * (fabs(bb.lo[k]) + fabs(bb.hi[k]) + (bb.hi[k] - bb.lo[k])) + 1e-7 * s->scene_extent + 1e-30;
        n->lo[k] = bb.lo[k] - pad; n->hi[k] = bb.hi[k] + pad;
    }
    if (count <= 4) { n->left = 0; n->right = 0; n->first = first; n->count = count; return idx; }
    int axis = 0; double ext = -1;
    for (int k = 0; k < 3; ++k) if (bb.hi[k] - bb.lo[k] > ext) { ext = bb.hi[k] - bb.lo[k]; axis = k; }
    g_sort_scene = s; g_sort_axis = axis;
    qsort(s->bvh_prims + first, count, sizeof(uint32_t), cmp_centroid);
    uint32_t half = count / 2;
    n->count = 0; n->first = 0;
    uint32_t l = bvh_build(s, first, half);
    uint32_t r = bvh_build(s, first + half, count - half);
    n = &s->bvh_nodes[idx];
    n->left = l; n->right = r;
    return idx;
}

int mo_scene_finalize(mo_scene *s) {
    if (!s) return -1;
    if (s->n_meshes == 0) { s->n_prims = 0; return 0; }      /* empty scene: every ray escapes */
    uint32_t total = 0;
    for (uint32_t i = 0; i < s->n_meshes; ++i) { s->meshes[i].prim_offset = total; total += s->meshes[i].n_faces; }
    s->n_prims = total;
    free(s->prim_shape); free(s->prim_local);
    s->prim_shape = (uint32_t *) malloc(sizeof(uint32_t) * total);
    s->prim_local = (uint32_t *) malloc(sizeof(uint32_t) * total);
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    for (uint32_t i = 0; i < s->n_meshes; ++i) {
        mo_mesh *m = &s->meshes[i];
        for (uint32_t f = 0; f < m->n_faces; ++f) {
            s->prim_shape[m->prim_offset + f] = i;
            s->prim_local[m->prim_offset + f] = f;
        }
        for (uint32_t v = 0; v < m->n_verts; ++v)
            for (int k = 0; k < 3; ++k) {
                double c = m->pos[3 * v + k];
                if (c < lo[k]) lo[k] = c;
                if (c > hi[k]) hi[k] = c;
            }
        /* area distribution (mesh.cpp:284-307) -- only needed for emitters, cheap anyway */
        free(m->area_pmf); free(m->area_cdf);
        m->area_pmf = (float *) malloc(sizeof(float) * m->n_faces);
        m->area_cdf = (float *) malloc(sizeof(float) * m->n_faces);
        for (uint32_t f = 0; f < m->n_faces; ++f) m->area_pmf[f] = face_area(m, f);
        int rc = mo_distr_build(m->n_faces, m->area_pmf, m->area_cdf, &m->area_sum, &m->area_norm,
                                &m->valid_lo, &m->valid_hi);
        if (rc != 0 && m->emitter >= 0) return -3;
    }
    s->scene_extent = 0;
    for (int k = 0; k < 3; ++k) {
        double e = fabs(lo[k]) > fabs(hi[k]) ? fabs(lo[k]) : fabs(hi[k]);
        if (e > s->scene_extent) s->scene_extent = e;
        if (hi[k] - lo[k] > s->scene_extent) s->scene_extent = hi[k] - lo[k];
    }
    free(s->bvh_nodes); free(s->bvh_prims);
    s->bvh_nodes = (mo_bvh_node *) malloc(sizeof(mo_bvh_node) * (2 * (size_t) total + 1));
    s->bvh_prims = (uint32_t *) malloc(sizeof(uint32_t) * total);
    for (uint32_t i = 0; i < total; ++i) s->bvh_prims[i] = i;
    s->n_bvh_nodes = 0;
    bvh_build(s, 0, total);
    /* ConstantBackgroundEmitter::set_scene (constant.cpp:47-51): bounding sphere of the scene's bounding box
     * (bbox.h:329-332), radius * (1 + RayEpsilon) */
    for (uint32_t i = 0; i < s->n_emitters; ++i) {
        mo_emitter *e = &s->emitters[i];
        if (e->type == 0 || e->type == 3 || e->type == 4) continue;      /* directional.cpp:65-70 does the same as the environment emitters */
        mo_v3 mn = mo_v3_make((float) lo[0], (float) lo[1], (float) lo[2]), mx = mo_v3_make((float) hi[0], (float) hi[1], (float) hi[2]);
        e->center = mo_scale(mo_add(mx, mn), 0.5f);
        float r = mo_norm(mo_sub(e->center, mx));
        e->radius = fmaxf(MO_RAY_EPSILON, r * (1.0f + MO_RAY_EPSILON));
    }
    return 0;
}

uint32_t mo_scene_prim_count(const mo_scene *s) { return s->n_prims; }
float mo_scene_emitter_area(const mo_scene *s, uint32_t e) {
    return s->meshes[s->emitters[e].shape].area_sum;
}
void mo_scene_set_naive(mo_scene *s, int naive) { s->force_naive = naive; }

/* ------------------------------------------------------------------ */
/* mesh.h:195-221 */
static inline int tri_intersect(const mo_mesh *m, uint32_t f, const mo_ray *ray, float *u_out,
                                float *v_out, float *t_out) {
    mo_v3 p0 = vtx(m, m->faces[3 * f]), p1 = vtx(m, m->faces[3 * f + 1]), p2 = vtx(m, m->faces[3 * f + 2]);
    mo_v3 e1 = mo_sub(p1, p0), e2 = mo_sub(p2, p0);
    mo_v3 pvec = mo_cross(ray->d, e2);
    float inv_det = mo_rcp(mo_dot(e1, pvec));
    mo_v3 tvec = mo_sub(ray->o, p0);
    float u = mo_dot(tvec, pvec) * inv_det;
    int active = (u >= 0.0f) && (u <= 1.0f);
    mo_v3 qvec = mo_cross(tvec, e1);
    float v = mo_dot(ray->d, qvec) * inv_det;
    active = active && (v >= 0.0f) && (u + v <= 1.0f);
    float t = mo_dot(e2, qvec) * inv_det;
    active = active && (t >= ray->mint) && (t <= ray->maxt);
    *u_out = u; *v_out = v; *t_out = t;
    return active;
}

/* kdtree.h:2303-2328 (scalar branch): later primitives with t <= maxt overwrite */
static int intersect_naive(const mo_scene *s, const mo_ray *ray_in, int shadow, mo_hit *hit) {
    mo_ray ray = *ray_in;
    int found = 0;
    for (uint32_t gp = 0; gp < s->n_prims; ++gp) {
        float u, v, t;
        if (tri_intersect(&s->meshes[s->prim_shape[gp]], s->prim_local[gp], &ray, &u, &v, &t)) {
            if (shadow) return 1;
            ray.maxt = t;
            found = 1;
            hit->t = t; hit->prim = gp; hit->u = u; hit->v = v;
        }
    }
    return found;
}

static int intersect_bvh(const mo_scene *s, const mo_ray *ray, int shadow, mo_hit *hit) {
    uint32_t stack[128]; int sp = 0;
    stack[sp++] = 0;
    double o[3] = { ray->o.x, ray->o.y, ray->o.z }, d[3] = { ray->d.x, ray->d.y, ray->d.z };
    double inv[3];
    for (int k = 0; k < 3; ++k) inv[k] = 1.0 / d[k];
    int found = 0; float best_t = ray->maxt; uint32_t best_prim = 0;
    double tmin_r = (double) ray->mint;
    while (sp > 0) {
        const mo_bvh_node *n = &s->bvh_nodes[stack[--sp]];
        double t0 = tmin_r - 1e-5 * (fabs(tmin_r) + 1.0), t1 = (double) best_t;
        if (isfinite(t1)) t1 += 1e-5 * (fabs(t1) + 1.0);
        int miss = 0;
        for (int k = 0; k < 3 && !miss; ++k) {
            if (d[k] == 0.0) { if (o[k] < n->lo[k] || o[k] > n->hi[k]) miss = 1; continue; }
            double a = (n->lo[k] - o[k]) * inv[k], b = (n->hi[k] - o[k]) * inv[k];
            if (a > b) { double c = a; a = b; b = c; }
            if (a > t0) t0 = a;
            if (b < t1) t1 = b;
            if (t0 > t1) miss = 1;
        }
        if (miss) continue;
        if (n->count == 0) { stack[sp++] = n->left; stack[sp++] = n->right; continue; }
        for (uint32_t i = 0; i < n->count; ++i) {
            uint32_t gp = s->bvh_prims[n->first + i];
            float u, v, t;
            if (tri_intersect(&s->meshes[s->prim_shape[gp]], s->prim_local[gp], ray, &u, &v, &t)) {
                if (shadow) return 1;
                if (!found || t < best_t || (t == best_t && gp > best_prim)) {
                    found = 1; best_t = t; best_prim = gp;
                    hit->t = t; hit->prim = gp; hit->u = u; hit->v = v;
                }
            }
        }
    }
    return found;
}

int mo_intersect(const mo_scene *s, const mo_ray *ray, int shadow, int naive, mo_hit *hit) {
    mo_hit tmp;
    if (!hit) hit = &tmp;
    if (s->n_prims == 0) return 0;
    if (naive || s->force_naive) return intersect_naive(s, ray, shadow, hit);
    return intersect_bvh(s, ray, shadow, hit);
}

/* kdtree.h:2334-2367 + mesh.cpp:399-462 */
void mo_make_si(const mo_scene *s, const mo_ray *ray, const mo_hit *hit, mo_si *si) {
    const mo_mesh *m = &s->meshes[s->prim_shape[hit->prim]];
    uint32_t f = s->prim_local[hit->prim];
    si->t = hit->t; si->prim = hit->prim; si->shape = s->prim_shape[hit->prim];
    float b1 = hit->u, b2 = hit->v, b0 = 1.0f - b1 - b2;
    uint32_t i0 = m->faces[3 * f], i1 = m->faces[3 * f + 1], i2 = m->faces[3 * f + 2];
    mo_v3 p0 = vtx(m, i0), p1 = vtx(m, i1), p2 = vtx(m, i2);
    mo_v3 dp0 = mo_sub(p1, p0), dp1 = mo_sub(p2, p0);
    si->p = mo_add(mo_add(mo_scale(p0, b0), mo_scale(p1, b1)), mo_scale(p2, b2));
    mo_v3 n = mo_normalize(mo_cross(dp0, dp1));
    si->n = n;
    mo_v3 dp_du, dp_dv;
    mo_coordinate_system(n, &dp_du, &dp_dv);
    si->uv.x = b1; si->uv.y = b2;
    if (m->uv) {
        mo_v2 uv0 = { m->uv[2 * i0], m->uv[2 * i0 + 1] }, uv1 = { m->uv[2 * i1], m->uv[2 * i1 + 1] },
              uv2 = { m->uv[2 * i2], m->uv[2 * i2 + 1] };
        si->uv.x = (uv0.x * b0 + uv1.x * b1) + uv2.x * b2;
        si->uv.y = (uv0.y * b0 + uv1.y * b1) + uv2.y * b2;
        mo_v2 duv0 = { uv1.x - uv0.x, uv1.y - uv0.y }, duv1 = { uv2.x - uv0.x, uv2.y - uv0.y };
        float det = fmaf(duv0.x, duv1.y, -(duv0.y * duv1.x));
        float inv_det = mo_rcp(det);
        if (det != 0.0f) {
            /* fmsub(duv1.y, dp0, duv0.y*dp1) * inv_det ; fnmadd(duv1.x, dp0, duv0.x*dp1) * inv_det */
            dp_du = mo_v3_make(fmaf(duv1.y, dp0.x, -(duv0.y * dp1.x)) * inv_det,
                               fmaf(duv1.y, dp0.y, -(duv0.y * dp1.y)) * inv_det,
                               fmaf(duv1.y, dp0.z, -(duv0.y * dp1.z)) * inv_det);
            dp_dv = mo_v3_make(fmaf(-duv1.x, dp0.x, duv0.x * dp1.x) * inv_det,
                               fmaf(-duv1.x, dp0.y, duv0.x * dp1.y) * inv_det,
                               fmaf(-duv1.x, dp0.z, duv0.x * dp1.z) * inv_det);
        }
    }
    if (m->nrm) {
        mo_v3 n0 = vnrm(m, i0), n1 = vnrm(m, i1), n2 = vnrm(m, i2);
        n = mo_normalize(mo_add(mo_add(mo_scale(n0, b0), mo_scale(n1, b1)), mo_scale(n2, b2)));
    }
    si->sh.n = n;
    si->dp_du = dp_du; si->dp_dv = dp_dv;
    /* Gram-Schmidt: s = normalize(fnmadd(n, dot(n, dp_du), dp_du)) */
    float dd = mo_dot(si->sh.n, si->dp_du);
    mo_v3 sv = mo_v3_make(fmaf(-si->sh.n.x, dd, si->dp_du.x), fmaf(-si->sh.n.y, dd, si->dp_du.y),
                          fmaf(-si->sh.n.z, dd, si->dp_du.z));
    si->sh.s = mo_normalize(sv);
    si->sh.t = mo_cross(si->sh.n, si->sh.s);
    si->wi = mo_to_local(&si->sh, mo_neg(ray->d));
}

/* ------------------------------------------------------------------ */
/* mesh.cpp:320-365 */
static void mesh_sample_position(const mo_mesh *m, mo_v2 sample, mo_v3 *p, mo_v3 *n, float *pdf) {
    float reused;
    uint32_t f = mo_distr_sample_reuse(m->area_pmf, m->area_cdf, m->area_sum, m->area_norm,
                                       m->valid_lo, m->valid_hi, sample.y, &reused);
    sample.y = reused;
    uint32_t i0 = m->faces[3 * f], i1 = m->faces[3 * f + 1], i2 = m->faces[3 * f + 2];
    mo_v3 p0 = vtx(m, i0), p1 = vtx(m, i1), p2 = vtx(m, i2);
    mo_v3 e0 = mo_sub(p1, p0), e1 = mo_sub(p2, p0);
    mo_v2 b = mo_square_to_uniform_triangle(sample);
    *p = mo_add(mo_add(p0, mo_scale(e0, b.x)), mo_scale(e1, b.y));
    *pdf = m->area_norm;
    if (m->nrm) {
        mo_v3 n0 = vnrm(m, i0), n1 = vnrm(m, i1), n2 = vnrm(m, i2);
        float b0 = 1.0f - b.x - b.y;
        *n = mo_normalize(mo_add(mo_add(mo_scale(n0, b0), mo_scale(n1, b.x)), mo_scale(n2, b.y)));
    } else {
        *n = mo_normalize(mo_cross(e0, e1));
    }
}

/* scene.cpp:141-189 with test_visibility handled by the caller (it needs the ray count);
 * area.cpp:103-117; shape.cpp:252-270 */
void mo_sample_emitter_direction(const mo_scene *s, mo_v3 ref_p, mo_v2 sample, mo_dsample *ds,
                                 float spec[3]) {
    memset(ds, 0, sizeof(*ds));
    spec[0] = spec[1] = spec[2] = 0.0f;
    if (s->n_emitters == 0) return;
    uint32_t index = 0;
    float emitter_pdf = 1.0f;
    if (s->n_emitters > 1) {
        float nf = (float) s->n_emitters;
        emitter_pdf = 1.0f / nf;
        uint32_t idx = (uint32_t) (sample.x * nf);
        index = idx < s->n_emitters - 1 ? idx : s->n_emitters - 1;
        sample.x = (sample.x - (float) index * emitter_pdf) * nf;
    }
    const mo_emitter *e = &s->emitters[index];
    int active;
    if (e->type == 2) {
        /* EnvironmentMapEmitter::sample_direction (envmap.cpp:154-190) */
        mo_v3 d; float pdf, val[3];
        mo_envmap_sample(e->env, sample, &d, &pdf, val, &ds->uv);
        ds->dist = 2.0f * e->radius;
        ds->p = mo_add(ref_p, mo_scale(d, ds->dist));
        ds->n = mo_neg(d); ds->d = d; ds->pdf = pdf; ds->emitter = index; ds->pdf_single = pdf;
        for (int k = 0; k < 3; ++k) spec[k] = val[k];
        if (s->n_emitters > 1) {
            ds->pdf *= emitter_pdf;
            float r = mo_rcp(emitter_pdf);
            for (int k = 0; k < 3; ++k) spec[k] *= r;
        }
        return;
    }
    if (e->type >= 3) {
        /* PointLight / SpotLight / DirectionalEmitter::sample_direction (point.cpp:76-101, spot.cpp:129-151,
         * directional.cpp:104-129): pdf = 1, delta */
        ds->pdf = ds->pdf_single = 1.0f; ds->delta = 1; ds->emitter = index;
        if (e->type == 5) {
            ds->dist = 2.0f * e->radius;
            ds->p = mo_sub(ref_p, mo_scale(e->dir, ds->dist));
            ds->n = e->dir; ds->d = mo_neg(e->dir);
            ds->falloff = 1.0f; ds->scale = 1.0f;
            for (int k = 0; k < 3; ++k) spec[k] = e->radiance[k];
        } else {
            ds->p = e->pos;
            ds->d = mo_sub(ds->p, ref_p);
            ds->dist = mo_norm(ds->d);
            float inv_dist = mo_rcp(ds->dist);
            ds->d = mo_scale(ds->d, inv_dist);
            float falloff = 1.0f;
            if (e->type == 4) {                              /* falloff_curve (spot.cpp:95-113) */
                const float *m = e->to_local;
                mo_v3 nd = mo_neg(ds->d);
                mo_v3 ld = mo_v3_make(fmaf(m[2], nd.z, fmaf(m[1], nd.y, m[0] * nd.x)), fmaf(m[5], nd.z, fmaf(m[4], nd.y, m[3] * nd.x)),
                                      fmaf(m[8], nd.z, fmaf(m[7], nd.y, m[6] * nd.x)));
                float cos_theta = mo_normalize(ld).z;
                if (!(cos_theta >= e->cos_beam)) falloff = (e->cutoff_angle - mo_lm_acos(cos_theta)) * e->inv_transition;
                if (cos_theta <= e->cos_cutoff) falloff = 0.0f;
            }
            ds->falloff = falloff; ds->scale = inv_dist * inv_dist;
            for (int k = 0; k < 3; ++k) spec[k] = (e->radiance[k] * falloff) * ds->scale;
        }
        if (s->n_emitters > 1) {
            ds->pdf *= emitter_pdf;
            float r = mo_rcp(emitter_pdf);
            for (int k = 0; k < 3; ++k) spec[k] *= r;
        }
        return;
    }
    if (e->type == 1) {
        /* ConstantBackgroundEmitter::sample_direction (constant.cpp:82-107); square_to_uniform_sphere (warp.h:262-267) */
        float z = fmaf(-2.0f, sample.y, 1.0f), r = mo_safe_sqrt(fmaf(-z, z, 1.0f));
        float ang = 2.0f * MO_PI_F * sample.x;
        mo_v3 d = mo_v3_make(r * mo_lm_cos(ang), r * mo_lm_sin(ang), z);
        ds->dist = 2.0f * e->radius;
        ds->p = mo_add(ref_p, mo_scale(d, ds->dist));
        ds->n = mo_neg(d);
        ds->pdf = MO_INV_FOUR_PI;
        ds->d = d;
        ds->emitter = index;
        ds->pdf_single = ds->pdf;
        active = 1;
    } else {
    const mo_mesh *m = &s->meshes[e->shape];
    mesh_sample_position(m, sample, &ds->p, &ds->n, &ds->pdf);
    ds->d = mo_sub(ds->p, ref_p);
    float dist_squared = mo_sqnorm(ds->d);
    ds->dist = sqrtf(dist_squared);
    ds->d = mo_div_s(ds->d, ds->dist);
    float dp = fabsf(mo_dot(ds->d, ds->n));
    ds->pdf *= (dp != 0.0f) ? dist_squared / dp : 0.0f;
    ds->emitter = index;
    ds->pdf_single = ds->pdf;
    active = (mo_dot(ds->d, ds->n) < 0.0f) && (ds->pdf != 0.0f);
    }
    if (active) {
        float r = mo_rcp(ds->pdf);
        for (int k = 0; k < 3; ++k) spec[k] = e->radiance[k] * r;
    }
    if (s->n_emitters > 1) {
        ds->pdf *= emitter_pdf;
        float r = mo_rcp(emitter_pdf);
        for (int k = 0; k < 3; ++k) spec[k] *= r;
    }
}

/* scene.cpp:191-206; area.cpp:119-125; shape.cpp:272-283 */
float mo_pdf_emitter_direction(const mo_scene *s, uint32_t emitter, mo_v3 d, mo_v3 n, float dist) {
    if (s->emitters[emitter].type == 2) {                    /* envmap.cpp:192-208 */
        float pdf = mo_envmap_pdf(s->emitters[emitter].env, d);
        if (s->n_emitters > 1) pdf *= 1.0f / (float) s->n_emitters;
        return pdf;
    }
    if (s->emitters[emitter].type == 1) {                    /* constant.cpp:109-114 */
        float pdf = MO_INV_FOUR_PI;
        if (s->n_emitters > 1) pdf *= 1.0f / (float) s->n_emitters;
        return pdf;
    }
    const mo_mesh *m = &s->meshes[s->emitters[emitter].shape];
    float pdf = 0.0f;
    if (mo_dot(d, n) < 0.0f) {
        pdf = m->area_norm;
        float dp = fabsf(mo_dot(d, n));
        pdf *= (dp != 0.0f) ? (dist * dist) / dp : 0.0f;
    }
    if (s->n_emitters > 1) pdf *= 1.0f / (float) s->n_emitters;
    return pdf;
}

/* ------------------------------------------------------------------ */
/* diffuse.cpp:78-135 */
void mo_diffuse_eval_pdf(const float refl[3], mo_v3 wi, mo_v3 wo, float eval[3], float *pdf) {
    float cos_i = wi.z, cos_o = wo.z;
    int active = cos_i > 0.0f && cos_o > 0.0f;
    for (int k = 0; k < 3; ++k) eval[k] = active ? (refl[k] * MO_INV_PI) * cos_o : 0.0f;
    *pdf = active ? mo_square_to_cosine_hemisphere_pdf(wo) : 0.0f;
}

int mo_diffuse_sample(const float refl[3], mo_v3 wi, mo_v2 sample2, mo_v3 *wo, float *pdf,
                      float weight[3]) {
    float cos_i = wi.z;
    *wo = mo_v3_make(0, 0, 0); *pdf = 0.0f;
    weight[0] = weight[1] = weight[2] = 0.0f;
    if (!(cos_i > 0.0f)) return 0;
    *wo = mo_square_to_cosine_hemisphere(sample2);
    *pdf = mo_square_to_cosine_hemisphere_pdf(*wo);
    if (*pdf > 0.0f) for (int k = 0; k < 3; ++k) weight[k] = refl[k];
    return 1;
}

/* ------------------------------------------------------------------ */
/* batch entry points */
void mo_ray_intersect(const mo_scene *s, uint64_t n, const float *ox, const float *oy, const float *oz,
                      const float *dx, const float *dy, const float *dz, const float *mint,
                      const float *maxt, int naive, float *t, uint32_t *prim, uint32_t *shape, float *u,
                      float *v) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t) n; ++i) {
        mo_ray r = { { ox[i], oy[i], oz[i] }, { dx[i], dy[i], dz[i] }, mint[i], maxt[i] };
        mo_hit h;
        if (mo_intersect(s, &r, 0, naive, &h)) {
            t[i] = h.t; prim[i] = h.prim; if (shape) shape[i] = s->prim_shape[h.prim];
            u[i] = h.u; v[i] = h.v;
        } else {
            t[i] = INFINITY; prim[i] = 0xffffffffu; if (shape) shape[i] = 0xffffffffu;
            u[i] = 0; v[i] = 0;
        }
    }
}

void mo_ray_test(const mo_scene *s, uint64_t n, const float *ox, const float *oy, const float *oz,
                 const float *dx, const float *dy, const float *dz, const float *mint,
                 const float *maxt, int naive, uint8_t *hit) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t) n; ++i) {
        mo_ray r = { { ox[i], oy[i], oz[i] }, { dx[i], dy[i], dz[i] }, mint[i], maxt[i] };
        hit[i] = (uint8_t) mo_intersect(s, &r, 1, naive, NULL);
    }
}

void mo_fill_si(const mo_scene *s, uint64_t n, const float *dx, const float *dy, const float *dz,
                const uint32_t *prim, const float *u, const float *v, float *out) {
    for (uint64_t i = 0; i < n; ++i) {
        float *o = out + 26 * i;
        if (prim[i] == 0xffffffffu) { memset(o, 0, sizeof(float) * 26); continue; }
        mo_ray r = { { 0, 0, 0 }, { dx[i], dy[i], dz[i] }, 0, INFINITY };
        mo_hit h = { 0.0f, prim[i], u[i], v[i] };
        mo_si si; mo_make_si(s, &r, &h, &si);
        o[0] = si.p.x; o[1] = si.p.y; o[2] = si.p.z; o[3] = si.n.x; o[4] = si.n.y; o[5] = si.n.z;
        o[6] = si.uv.x; o[7] = si.uv.y;
        o[8] = si.sh.s.x; o[9] = si.sh.s.y; o[10] = si.sh.s.z;
        o[11] = si.sh.t.x; o[12] = si.sh.t.y; o[13] = si.sh.t.z;
        o[14] = si.sh.n.x; o[15] = si.sh.n.y; o[16] = si.sh.n.z;
        o[17] = si.dp_du.x; o[18] = si.dp_du.y; o[19] = si.dp_du.z;
        o[20] = si.dp_dv.x; o[21] = si.dp_dv.y; o[22] = si.dp_dv.z;
        o[23] = si.wi.x; o[24] = si.wi.y; o[25] = si.wi.z;
    }
}
