/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_math.h header for scope and pinning). */
#ifndef MO_INTERNAL_H
#define MO_INTERNAL_H
#include "mo_math.h"
#include "mo_api.h"

typedef struct { mo_v3 o, d; float mint, maxt; } mo_ray;
typedef struct { float t; uint32_t prim; float u, v; } mo_hit;

/* subset of SurfaceInteraction (include/mitsuba/render/interaction.h:102-126) used by `path` */
typedef struct {
    float t; uint32_t prim, shape;
    mo_v3 p, n; mo_v2 uv; mo_frame sh; mo_v3 dp_du, dp_dv, wi;
} mo_si;

/* DirectionSample (include/mitsuba/render/records.h:121-174) */
typedef struct { mo_v3 p, n, d; float dist, pdf; uint32_t emitter; float pdf_single; mo_v2 uv; int delta; float falloff, scale; } mo_dsample;   /* delta emitters: spec = (L * falloff) * scale */   /* pdf_single: before the emitter-selection probability; uv: envmap samples */

/* a BSDF instance: the descriptor plus the constants its constructor derives (plastic.cpp:162-176) */
typedef struct mo_bsdf {
    mo_bsdf_desc d; float eta_rel, inv_eta_2, fdr_int, fdr_ext, spec_weight;
    float refl_coeff[3], spec_coeff[3], trans_coeff[3];      /* spectral variant: srgb_model coefficients */
    float ext_trans[64], internal_reflectance;               /* roughplastic (roughplastic.cpp:380-399) */
    /* nest = MO_NEST_BLEND / MO_NEST_MASK: blendbsdf.cpp / mask.cpp over child[0..1] (mask: child[0]); the weight / opacity
     * arrives where a plain BSDF receives its reflectance (the shape's constant or texture); weight_lum: it is a bitmap
     * texel whose eval_1 is the luminance (bitmap.cpp:215-231) */
    int nest, weight_lum; struct mo_bsdf *child[2]; int child_tex[2];      /* child_tex: reflectance texture of a child or -1 */
} mo_bsdf;
/* per-channel inputs of a BSDF evaluation: 3 colour channels or MO_WAV wavelengths */
typedef struct { float refl[4], spec[4], trans[4], eta[4], k[4]; } mo_bsdf_chan;
/* BSDFSample3 (bsdf.h:193-252): delta = has_flag(sampled_type, BSDFFlags::Delta) */
typedef struct { mo_v3 wo; float pdf, eta; int delta; } mo_bsample;

typedef struct {
    uint32_t n_verts, n_faces;
    float *pos, *nrm, *uv; uint32_t *faces;
    int bsdf_kind; float refl[3]; int emitter; int texture;
    mo_bsdf bsdf;                   /* bsdf.d.type == bsdf_kind; bsdf.d.reflectance mirrors refl */
    float refl_coeff[3];            /* spectral variant: srgb_model coefficients of the reflectance */
    uint32_t prim_offset;
    float *area_pmf, *area_cdf; float area_sum, area_norm; uint32_t valid_lo, valid_hi;
} mo_mesh;

/* Hierarchical2D<Float, 0> (distr_2d.h:180-600) and the `envmap` emitter built on it (mo_envmap.c) */
typedef struct { uint32_t size, width; float *data; } mo_h2_level;
typedef struct { int n_levels; mo_h2_level lv[34]; float patch_size[2], inv_patch_size[2]; uint32_t max_patch_index[2]; } mo_hier2d;
typedef struct { int w, h; float *data; float scale; mo_hier2d warp; float to_world[9], to_local[9]; } mo_envmap;
int mo_hier2d_build(mo_hier2d *h, const float *data, uint32_t w, uint32_t hgt, int normalize);
void mo_hier2d_free(mo_hier2d *h);
void mo_hier2d_sample(const mo_hier2d *h, float sx, float sy, float *ox, float *oy, float *pdf);
float mo_hier2d_eval(const mo_hier2d *h, float px, float py);
int mo_envmap_init(mo_envmap *e, int w, int h, const float *rgb, float scale, const float *to_world9);
void mo_envmap_free(mo_envmap *e);
void mo_envmap_eval(const mo_envmap *e, mo_v3 d, float out[3]);
void mo_envmap_sample(const mo_envmap *e, mo_v2 sample, mo_v3 *d_out, float *pdf_out, float spec[3], mo_v2 *uv_out);
void mo_envmap_lookup_spectral(const mo_envmap *e, float u, float v, const float *wav, float *out);   /* 4 wavelengths */
void mo_envmap_eval_spectral(const mo_envmap *e, mo_v3 d, const float *wav, float *out);
float mo_envmap_pdf(const mo_envmap *e, mo_v3 d_world);
void mo_envmap_footprint(const mo_envmap *e, float u, float v, uint32_t idx[4], float w[4]);
void mo_envmap_dir_to_uv(const mo_envmap *e, mo_v3 d_world, float *u, float *v);
int mo_envmap_update(mo_envmap *e, const float *rgb, int rebuild_warp);

/* type 0: `area` (src/emitters/area.cpp) attached to `shape`; type 1: `constant` environment (src/emitters/constant.cpp)
 * with the scene's bounding sphere (set_scene, constant.cpp:47-51); type 2: `envmap` (src/emitters/envmap.cpp) */
typedef struct { uint32_t shape; float radiance[3]; float coeff[3], d65_scale; int type; mo_v3 center; float radius; mo_envmap *env;
                 /* type 3: `point` (point.cpp), 4: `spot` (spot.cpp), 5: `directional` (directional.cpp): delta emitters */
                 mo_v3 pos, dir; float to_local[9], cutoff_angle, cos_cutoff, cos_beam, inv_transition; } mo_emitter;
/* kind 0: bitmap (src/textures/bitmap.cpp), kind 1: checkerboard (src/textures/checkerboard.cpp); uvm = upper-left 2x3 of the
 * extracted to_uv transform: uv' = (uvm[0] u + uvm[1] v + uvm[2], uvm[3] u + uvm[4] v + uvm[5]) */
typedef struct { int w, h; float *data; int kind; float uvm[6]; float color0[3], color1[3];
                 float mean;        /* Texture::mean(): bitmap.cpp:112-136 / checkerboard.cpp:88-90 (feeds plastic's lobe weights) */
                 float coeff0[3], coeff1[3]; } mo_texture;   /* spectral variant: checkerboard colours upsampled; bitmap `data` holds coefficients */
typedef struct { double lo[3], hi[3]; uint32_t left, right, first, count; } mo_bvh_node;

struct mo_scene {
    mo_mesh *meshes; uint32_t n_meshes;
    mo_emitter *emitters; uint32_t n_emitters;
    mo_texture *textures; uint32_t n_textures;
    uint32_t n_prims; uint32_t *prim_shape, *prim_local;
    mo_bvh_node *bvh_nodes; uint32_t n_bvh_nodes; uint32_t *bvh_prims;
    double scene_extent; int force_naive;
    int environment;                /* index of the environment emitter or -1 (scene.cpp:44-48) */
    int spectral;                   /* 0: RGB variant, 1: spectral variant (4 wavelengths) */
};

int mo_intersect(const mo_scene *s, const mo_ray *ray, int shadow, int naive, mo_hit *hit);
void mo_make_si(const mo_scene *s, const mo_ray *ray, const mo_hit *hit, mo_si *si);
void mo_sample_emitter_direction(const mo_scene *s, mo_v3 ref_p, mo_v2 sample, mo_dsample *ds, float spec[3]);
float mo_pdf_emitter_direction(const mo_scene *s, uint32_t emitter, mo_v3 d, mo_v3 n, float dist);
void mo_diffuse_eval_pdf(const float refl[3], mo_v3 wi, mo_v3 wo, float eval[3], float *pdf);
int mo_diffuse_sample(const float refl[3], mo_v3 wi, mo_v2 sample2, mo_v3 *wo, float *pdf, float weight[3]);
int mo_distr_build(uint32_t n, const float *pmf, float *cdf, float *sum_out, float *norm_out,
                   uint32_t *valid_lo, uint32_t *valid_hi);
uint32_t mo_distr_sample(const float *cdf, float sum, uint32_t lo, uint32_t hi, float value);
uint32_t mo_distr_sample_reuse(const float *pmf, const float *cdf, float sum, float norm, uint32_t lo,
                               uint32_t hi, float value, float *reused);
void mo_scene_set_naive(mo_scene *s, int naive);
/* reflectance at a surface interaction: constant colour or BitmapTexture::interpolate (bitmap.cpp:250-293);
 * footprint (may be NULL): texel index of v00 and the bilinear weights w1.x, w1.y */
void mo_reflectance(const mo_scene *s, const mo_mesh *m, mo_v2 uv, float out[3], uint32_t *texel, float w1[2]);
/* out9 = (reflectance or blend weight lookup, reflectance of child 0, of child 1): what mo_bsdf_sample / mo_bsdf_eval_pdf take as `refl`
 * (plain BSDFs read the first three values only) */
void mo_surface_reflectance(const mo_scene *s, const mo_mesh *m, mo_v2 uv, float out9[9]);


/* 8-wide ray queries of the packet_rgb-equivalent CPU baseline (mo_packet.c) */
typedef struct mo_packet_accel mo_packet_accel;
mo_packet_accel *mo_packet_accel_build(const mo_scene *s);
void mo_packet_accel_free(mo_packet_accel *a);
uint32_t mo_packet_intersect(const mo_scene *s, const mo_packet_accel *a, const mo_ray *rays, uint32_t lanes, int shadow, mo_hit *hits);

/* BSDF models (mo_bsdf.c) */
void mo_fresnel(float cos_theta_i, float eta, float out[4]);
float mo_fresnel_conductor(float cos_theta_i, float eta_r, float eta_i);
float mo_fresnel_diffuse_reflectance(float eta);
void mo_bsdf_prepare(mo_bsdf *b);
int mo_bsdf_is_smooth(const mo_bsdf *b);
int mo_bsdf_sample(const mo_bsdf *b, const float refl[3], mo_v3 wi, float sample1, mo_v2 sample2, mo_bsample *bs, float weight[3]);
void mo_bsdf_eval_pdf(const mo_bsdf *b, const float refl[3], mo_v3 wi, mo_v3 wo, float value[3], float *pdf);
int mo_bsdf_sample_n(const mo_bsdf *b, int n, const mo_bsdf_chan *c, mo_v3 wi, float sample1, mo_v2 sample2, mo_bsample *bs, float *weight);
void mo_bsdf_eval_pdf_n(const mo_bsdf *b, int n, const mo_bsdf_chan *c, mo_v3 wi, mo_v3 wo, float *value, float *pdf);
void mo_bsdf_spectral_channels(const mo_bsdf *b, const float *wav, mo_bsdf_chan *c);
/* spectral variant of mo_bsdf_sample / mo_bsdf_eval_pdf: `c` = mo_bsdf_spectral_channels(b, wav) (+ textured reflectance); nested
 * BSDFs build the channels of their children from `wav` */
int mo_bsdf_sample_spec(const mo_bsdf *b, const float *wav, const mo_bsdf_chan *c, mo_v3 wi, float sample1, mo_v2 sample2, mo_bsample *bs, float *weight);
void mo_bsdf_eval_pdf_spec(const mo_bsdf *b, const float *wav, const mo_bsdf_chan *c, mo_v3 wi, mo_v3 wo, float *value, float *pdf);
float mo_srgb_model_mean(const float coeff[3]);

/* spectral variant (mo_spectral.c) */
#define MO_WAV 4
void mo_reflectance_spectral(const struct mo_scene *s, const mo_mesh *m, mo_v2 uv, const float *wav, float *out);   /* textured reflectance, 4 wavelengths */
void mo_texture_update_mean(struct mo_scene *s, int texture);
void mo_sample_wavelengths(float sample, float wav[MO_WAV], float weight[MO_WAV]);
float mo_srgb_model_eval(const float coeff[3], float lambda);
float mo_d65_eval(float scale, float lambda);
void mo_spectrum_to_xyz(const float value[MO_WAV], const float wav[MO_WAV], float xyz[3]);
#endif
