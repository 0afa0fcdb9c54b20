/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_math.h header for scope and pinning).
 * Flat C API of the CPU restatement, loaded through ctypes by tests/, smoke()
 * and bench.py's cpu_baseline leg.  Never linked or imported by the product. */
#ifndef MO_API_H
#define MO_API_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mo_scene mo_scene;

/* ---- scene construction ------------------------------------------------ */
mo_scene *mo_scene_new(void);
void mo_scene_free(mo_scene *s);
/* Adds one triangle mesh (Mesh buffers as in include/mitsuba/render/mesh.h:80-90).
 * normals / texcoords may be NULL.  bsdf_kind: 0 = diffuse (src/bsdfs/diffuse.cpp).
 * emitter_rgb == NULL: not an emitter, else an `area` emitter (src/emitters/area.cpp).
 * returns the shape index or <0. */
int mo_scene_add_mesh(mo_scene *s, uint32_t n_verts, const float *positions,
                      const float *normals, const float *texcoords, uint32_t n_faces,
                      const uint32_t *faces, int bsdf_kind, const float *reflectance_rgb,
                      const float *emitter_rgb);
/* `constant` environment emitter (src/emitters/constant.cpp); RGB variant only.  Returns the emitter index. */
int mo_scene_add_constant_emitter(mo_scene *s, const float *radiance_rgb);
/* `envmap` emitter (src/emitters/envmap.cpp): linear RGB latitude-longitude image (h * w * 3), `scale`, linear part of
 * to_world (row-major 3x3, NULL = identity); RGB variant only.  Returns the emitter index. */
int mo_scene_add_envmap_emitter(mo_scene *s, int w, int h, const float *rgb, float scale, const float *to_world9);
/* Hierarchical2D0 (distr_2d.h): which = 0 sample / 1 invert / 2 eval for n points -> (x, y, pdf) each */
void mo_kat_hier2d(const float *data, uint32_t w, uint32_t h, int normalize, int which, uint64_t n, const float *in2, float *out3);
void mo_kat_bilinear_to_square(float v00, float v10, float v01, float v11, float x, float y, float *out3);
/* envmap: per sample d(3) pdf spec(3) eval(d)(3) pdf_direction(d) = 11 floats */
void mo_kat_envmap(int w, int h, const float *rgb, float scale, const float *to_world9, uint64_t n, const float *sample2, float *out11);
/* Re-orders the emitter list: new emitter i = old emitter order[i] (Scene::m_emitters follows the order of the scene's
 * children, scene.cpp:31-56). */
/* delta emitters: type 3 `point` (src/emitters/point.cpp: position, intensity), 4 `spot` (spot.cpp: position, intensity,
 * to_world9 = rotation part of to_world, cutoff_angle / beam_width in degrees), 5 `directional` (directional.cpp: unit direction
 * the light travels in = to_world * (0, 0, 1), irradiance).  Returns the emitter index. */
int mo_scene_add_delta_emitter(mo_scene *s, int type, const float *rgb, const float *position3, const float *direction3,
                               const float *to_world9, float cutoff_angle_deg, float beam_width_deg);
int mo_scene_set_emitter_order(mo_scene *s, uint32_t n, const uint32_t *order);
int mo_scene_set_emitter_radiance(mo_scene *s, uint32_t emitter, const float *rgb);
/* Bitmap texture (src/textures/bitmap.cpp, linear RGB data, identity to_uv): returns its index.  A texture is
 * attached to the reflectance of a shape's diffuse BSDF with mo_scene_set_texture (-1 detaches). */
int mo_scene_add_texture(mo_scene *s, int width, int height, const float *rgb);
/* to_uv of a texture: uvm6 = (m00, m01, m02, m10, m11, m12) of the extracted 3x3 transform (bitmap.cpp:62,254) */
int mo_scene_set_texture_transform(mo_scene *s, uint32_t texture, const float *uvm6);
/* Checkerboard texture with constant colours (src/textures/checkerboard.cpp:40-64): returns its index. */
int mo_scene_add_checkerboard(mo_scene *s, const float *color0, const float *color1, const float *uvm6);
int mo_scene_set_texture(mo_scene *s, uint32_t shape, int texture);
int mo_scene_update_texture(mo_scene *s, uint32_t texture, const float *rgb);
int mo_scene_set_reflectance(mo_scene *s, uint32_t shape, const float *rgb);
/* BSDF models beyond `diffuse` (oracle/mo_bsdf.c).  `twosided` wraps the model in the TwoSidedBRDF adapter. */
enum { MO_BSDF_DIFFUSE = 0, MO_BSDF_CONDUCTOR = 1, MO_BSDF_ROUGHCONDUCTOR = 2, MO_BSDF_DIELECTRIC = 3, MO_BSDF_PLASTIC = 4,
       MO_BSDF_ROUGHPLASTIC = 5, MO_BSDF_ROUGHDIELECTRIC = 6, MO_BSDF_THINDIELECTRIC = 7 };
typedef struct {
    int32_t type, twosided;
    float reflectance[3];              /* diffuse.reflectance / plastic.diffuse_reflectance (constant part) */
    float specular_reflectance[3];     /* default 1 */
    float specular_transmittance[3];   /* dielectric, default 1 */
    float eta[3], k[3];                /* conductors: complex index of refraction per colour channel */
    float int_ior, ext_ior;            /* dielectric / plastic */
    float alpha_u, alpha_v;            /* roughconductor */
    int32_t distribution;              /* 0 beckmann, 1 ggx */
    int32_t sample_visible;
    int32_t nonlinear;                 /* plastic */
    int32_t uniform_mask;              /* spectral variant: bit 0 / 1 / 2 set = reflectance / specular_reflectance /
                                          specular_transmittance is a `uniform` spectrum (given as a constant: xml.cpp:1069-1083)
                                          instead of an upsampled RGB colour (`srgb`); conductors need uniform eta and k */
} mo_bsdf_desc;
/* Replaces the BSDF of a shape (keeps an attached reflectance texture). */
int mo_scene_set_bsdf(mo_scene *s, uint32_t shape, const mo_bsdf_desc *desc);
/* blendbsdf (src/bsdfs/blendbsdf.cpp) and mask (src/bsdfs/mask.cpp) over plain child BSDFs with constant parameters.  The blend
 * weight / opacity is `weight`, or -- after mo_scene_set_texture on the shape -- Texture::eval_1 of that texture: the luminance of
 * the interpolated bitmap texel (bitmap.cpp:215-231) or the (constant, first-channel) colour of the checkerboard cell
 * (checkerboard.cpp:67-86).  `twosided` wraps the whole nest in the TwoSidedBRDF adapter (blend only: mask transmits). */
enum { MO_NEST_BLEND = 1, MO_NEST_MASK = 2 };
int mo_scene_set_nested_bsdf(mo_scene *s, uint32_t shape, int kind, float weight, int twosided, const mo_bsdf_desc *child0, const mo_bsdf_desc *child1,
                             int child_tex0, int child_tex1);      /* child_tex: reflectance texture of a child (RGB variant) or -1 */
/* one BSDF query per row as mo_kat_bsdf, on a blend / mask with constant weight */
void mo_kat_nested_bsdf(int kind, float weight, int twosided, const mo_bsdf_desc *child0, const mo_bsdf_desc *child1, uint64_t n,
                        const float *wi3, const float *wo3, const float *sample3, float *out14);
void mo_kat_fresnel(float cos_theta_i, float eta, float *out4);
float mo_kat_fresnel_conductor(float cos_theta_i, float eta_r, float eta_i);
float mo_kat_fresnel_diffuse(float eta);
void mo_kat_microfacet(int ggx, float alpha_u, float alpha_v, int visible, int which, uint64_t n, const float *v3, const float *wi3, float *out);
void mo_kat_microfacet_sample(int ggx, float alpha_u, float alpha_v, int visible, uint64_t n, const float *wi3, const float *sample2, float *m3, float *pdf);
void mo_kat_bsdf(const mo_bsdf_desc *desc, uint64_t n, const float *wi3, const float *wo3, const float *sample3, float *out14);
/* quad::gauss_legendre (src/libcore/quad.cpp:7-66) */
void mo_kat_gauss_legendre(int n, float *nodes, float *weights);
/* roughplastic tables (roughplastic.cpp:380-399): external transmittance at 64 cosines + internal reflectance (out[64]) */
void mo_kat_roughplastic_tables(const mo_bsdf_desc *desc, float *out65);

/* Switches the scene to the spectral variant: every RGB reflectance / radiance is upsampled through the coefficient
 * table at `coeff_path` ("data/srgb.coeff": srgb_model_fetch, src/librender/srgb.cpp:14-40; rgb2spec_fetch,
 * ext/rgb2spec/rgb2spec.c:81-121).  Call after all meshes were added.  Returns 0 on success. */
int mo_scene_set_spectral(mo_scene *s, const char *coeff_path);
/* unit-level entry points of the spectral helpers */
void mo_kat_srgb_model_fetch(const char *coeff_path, const float *rgb3, float *coeff3);
void mo_kat_spectral(float sample, const float *coeff3, float d65_scale, float *out /* 4 wav, 4 weight, 4 refl, 4 d65, xyz(3) of refl */);
/* Builds the oracle's own accelerator + emitter sampling tables. */
int mo_scene_finalize(mo_scene *s);
/* naive != 0: the render entry points answer every query by brute force. */
void mo_scene_set_naive(mo_scene *s, int naive);
uint32_t mo_scene_prim_count(const mo_scene *s);
float mo_scene_emitter_area(const mo_scene *s, uint32_t emitter);

/* ---- scene queries (Scene::ray_intersect / ray_intersect_naive / ray_test) ---- */
/* SoA rays; outputs t (inf on miss), global prim index (0xffffffff on miss),
 * shape index, barycentrics u,v.  naive != 0: brute force (kdtree.h:2303-2328). */
void mo_ray_intersect(const mo_scene *s, uint64_t n, const float *ox, const float *oy,
                      const float *oz, const float *dx, const float *dy, const float *dz,
                      const float *mint, const float *maxt, int naive, float *t,
                      uint32_t *prim, uint32_t *shape, float *u, float *v);
void mo_ray_test(const mo_scene *s, uint64_t n, const float *ox, const float *oy,
                 const float *oz, const float *dx, const float *dy, const float *dz,
                 const float *mint, const float *maxt, int naive, uint8_t *hit);
/* The same two queries answered 8 rays at a time by the packet traversal of the packet_rgb-equivalent baseline (mo_packet.c):
 * closest hit (t, prim, u, v) and any hit.  Results equal mo_ray_intersect / mo_ray_test with naive == 0. */
void mo_packet_ray_intersect(const mo_scene *s, uint64_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                             const float *dy, const float *dz, const float *mint, const float *maxt, float *t, uint32_t *prim,
                             float *u, float *v, uint8_t *any_hit);
/* Full SurfaceInteraction for given hits (kdtree.h:2334-2367 + mesh.cpp:399-462).
 * out: 24 floats per ray: p(3) n(3) uv(2) sh_s(3) sh_t(3) sh_n(3) dp_du(3) dp_dv(3) wi(3) -> 26 */
void mo_fill_si(const mo_scene *s, uint64_t n, const float *dx, const float *dy,
                const float *dz, const uint32_t *prim, const float *u, const float *v,
                float *out26);

/* ---- sensor / film / integrator descriptors ---------------------------- */
typedef struct {
    float to_world[16];         /* row-major camera-to-world matrix */
    float fov_x_deg;            /* horizontal field of view (after parse_fov) */
    float near_clip, far_clip;
    int32_t film_w, film_h;     /* full film size */
    int32_t crop_x, crop_y, crop_w, crop_h;
    int32_t rfilter;            /* 0 gaussian, 1 box, 2 tent, 3 catmullrom, 4 mitchell, 5 lanczos */
    float rfilter_param;        /* gaussian: stddev (0.5), box: radius (0.5), mitchell: B (1/3), lanczos: lobes (3) */
    float rfilter_param2;       /* mitchell: C (1/3) */
    int32_t spp;
    uint64_t base_seed;         /* sampler "seed" property */
    int32_t max_depth, rr_depth;
    int32_t filter_analytic;    /* 0: eval_discretized (scalar_rgb), 1: eval (gpu variants) */
    int32_t film_rgb;           /* 0: film channels X,Y,Z,A,W; 1: R,G,B,A,W (autodiff.py:53-72) */
    int32_t integrator;         /* 0: path (src/integrators/path.cpp), 1: direct (direct.cpp), 2: depth (depth.cpp) */
    int32_t emitter_samples, bsdf_samples;   /* direct: samples per technique (0, 0 = shading_samples default 1, 1) */
    int32_t hide_emitters;      /* direct (integrator.cpp:39, direct.cpp:117-121) */
    float aperture_radius;      /* 0: `perspective` (src/sensors/perspective.cpp); > 0: `thinlens` (src/sensors/thinlens.cpp) */
    float focus_distance;       /* thinlens: distance of the plane in focus (sensor.cpp:104, default far_clip) */
} mo_render_desc;

/* mode 0: scalar_rgb block mode (spiral blocks, Morton order, one PCG32 stream per block);
 * mode 1: wavefront mode (one PCG32 stream per sample, TEA-seeded), single pass;
 * mode 2: packet_rgb block mode (integrator.cpp:204-212: 8 sample indices per packet, 8 PCG32 streams per block, ray queries 8 wide
 *         on AVX2 with lane voting, mo_packet.c) -- RGB `path` with a pinhole camera only, -2 otherwise;
 * mode 3: the schedule and streams of mode 2 traced by the scalar code (the checker of mode 2: identical film).
 * film_xyzaw: crop_h*crop_w*5 floats, overwritten.  n_threads<=0: all cores.
 * block_size 0: reference rule (32 halved until #blocks >= n_threads).
 * stats (may be NULL): [0]=closest-hit queries, [1]=any-hit queries, [2]=samples. */
int mo_render(const mo_scene *s, const mo_render_desc *d, int mode, int n_threads,
              int block_size, float *film_xyzaw, uint64_t *stats);
/* Per-sample radiance in wavefront mode for sample indices [first, first+count):
 * out_rgba[4*i] = (R,G,B, valid_ray), out_pos[2*i] = film position sample. */
int mo_sample_radiance(const mo_scene *s, const mo_render_desc *d, uint64_t first,
                       uint64_t count, float *out_rgba, float *out_pos);
/* Restricted wavefront render: only samples of pixels with crop-relative row in
 * [row0,row1) are traced; splats land in the full crop-sized film (for multi-GPU tests). */
int mo_render_rows(const mo_scene *s, const mo_render_desc *d, int row0, int row1,
                   float *film_xyzaw);
/* the same for the pixels of the columns [col0, col1) of those rows only */
int mo_render_window(const mo_scene *s, const mo_render_desc *d, int row0, int row1, int col0, int col1, float *film);
/* Derivative of the same image w.r.t. the texels of the `envmap` emitter ('data', envmap.cpp:214-218; the inverse-rendering example
 * docs/examples/10_inverse_rendering/invert_bunny.py): radiance is linear in them, the sampling distribution built from their
 * luminances is not differentiated (envmap.cpp:220-253 rebuilds it from plain floats).  Any BSDF, any depth.  grad_env: h*w*3,
 * accumulated. */
int mo_render_adjoint_envmap(const mo_scene *s, const mo_render_desc *d, const float *dimage, const float *film, float *grad_env);
/* new envmap texels (h*w*3); rebuild_warp = 0 keeps the sampling distribution (a render is then exactly linear in the texels) */
int mo_scene_update_envmap(mo_scene *s, const float *rgb, int rebuild_warp);
/* Reverse-mode derivative of Image = RGB / (W + 1e-8) (autodiff.py:80-91) w.r.t. diffuse reflectances, by path replay
 * (restatement of what ek.backward() does through PathIntegrator::sample for these parameters; RR probabilities detached).
 * dimage: crop_h*crop_w*3 = dLoss/dImage; film: the primal R,G,B,A,W film of the same desc (weights);
 * grad_shape: n_shapes*3 (constant reflectance of each shape's BSDF), grad_tex: textures concatenated, grad_emitter:
 * n_emitters*3 (radiance of area lights, 'shape.emitter.radiance.value'); all accumulated, any may be NULL. */
int mo_render_adjoint(const mo_scene *s, const mo_render_desc *d, const float *dimage, const float *film,
                      float *grad_shape, float *grad_tex, float *grad_emitter);
/* HDRFilm::bitmap(): XYZAW -> RGBA float32 (hdrfilm.cpp:249-320, struct.cpp:1761-1811) */
void mo_film_develop(const float *xyzaw, uint64_t n_pixels, float *rgba);

/* ---- camera ------------------------------------------------------------ */
/* PerspectiveCamera::sample_ray (perspective.cpp:106-188) for n film samples in [0,1)^2 */
void mo_camera_rays(const mo_render_desc *d, uint64_t n, const float *sx, const float *sy, const float *aperture2 /* or NULL */,
                    float *o3, float *d3, float *mint, float *maxt);

/* ---- ImageBlock (imageblock.cpp:8-172) ---------------------------------- */
/* Splat n samples (pos 2 floats, value ch floats) into a block of size (w,h) at offset (ox,oy)
 * with optional border; data: (h+2b)*(w+2b)*ch floats, accumulated in place.
 * returns the border size. */
int mo_imageblock_put(int w, int h, int ox, int oy, int ch, int rfilter, float rfilter_param, float rfilter_param2,
                      int border, int analytic, uint64_t n, const float *pos,
                      const float *values, float *data);
/* ReconstructionFilter::eval_discretized table (32 entries), radius and border size */
void mo_rfilter_table(int rfilter, float param, float param2, float *table32, float *radius, int *border);

/* ---- unit-level entry points for known-answer tests --------------------- */
uint32_t mo_kat_tea32(uint32_t v0, uint32_t v1, int rounds);
uint64_t mo_kat_tea64_u32(uint32_t v0, uint32_t v1, int rounds);
uint64_t mo_kat_tea64_u64(uint64_t v0, uint64_t v1, int rounds);
float mo_kat_tea_float32(uint32_t v0, uint32_t v1, int rounds);
double mo_kat_tea_float64(uint32_t v0, uint32_t v1, int rounds);
void mo_kat_pcg32(uint64_t initstate, uint64_t initseq, int n, uint32_t *out_u32, float *out_f32);
void mo_kat_warp(int which, uint64_t n, const float *sx, const float *sy, float *out3);
void mo_kat_coordinate_system(const float *n3, float *s3, float *t3);
/* Spiral(size, offset, block_size, passes): fills up to max entries of (ox,oy,w,h,id) */
int mo_kat_spiral(int w, int h, int off_x, int off_y, int block_size, int passes, int max_entries,
                  int64_t *out5);
void mo_kat_morton(uint32_t n, uint32_t *xy);
/* DiscreteDistribution: returns sum; cdf out; sample / sample_reuse for values */
float mo_kat_distr(uint32_t n, const float *pmf, float *cdf, uint32_t nv, const float *values,
                   uint32_t *idx, float *reused);
/* SmoothDiffuse::sample/eval/pdf for local wi, wo / sample2 */
void mo_kat_diffuse(const float *reflectance, const float *wi3, const float *wo3,
                    const float *sample2, float *eval3, float *pdf, float *s_wo3, float *s_pdf,
                    float *s_weight3);
/* emitter sampling from a reference point: Scene::sample_emitter_direction without the
 * visibility test + Scene::pdf_emitter_direction evaluated on the sampled record.
 * out: d(3) dist pdf n(3) p(3) spec(3) pdf_again = 15 floats */
void mo_kat_sample_emitter(const mo_scene *s, const float *ref_p3, const float *sample2,
                           float *out15);

#ifdef __cplusplus
}
#endif
/* Checker of mtsamd_render_adjoint_param: d(loss)/d(component `comp` of parameter `kind` of the BSDF of the shapes with shape_mask[i] != 0)
 * (kinds: 0 (diffuse_)reflectance, 1 specular_reflectance, 2 eta, 3 k, 4 alpha, 5 specular_transmittance), forward-mode derivative carried
 * beside every replayed path, detached sampling, central difference of the model code with step h at fixed directions. */
int mo_render_adjoint_param(const mo_scene *s, const mo_render_desc *d, const float *dimage, const float *film, const uint8_t *shape_mask,
                            int kind, int comp, float h, double *grad);

/* mo_libm.h on argument arrays (fn: 0 sin, 1 cos, 2 tan, 3 exp, 4 log, 5 erf, 6 acos, 7 atan2(x[i], y[i])); the counterpart of
 * mtsamd_libm_eval, so that a test can compare host and device bit for bit */
void mo_libm_eval(int fn, uint64_t n, const float *x, const float *y, float *out);

#endif
