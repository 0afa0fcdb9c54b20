/* mtsamd.h -- C ABI of the MI355X (gfx950) wavefront path-tracing backend.
 *
 * This is the drop-in boundary for Mitsuba 2's path-tracing hot path.  Every entry
 * point names the reference interface it stands in for (paths relative to the
 * Mitsuba 2 source tree).  Conventions:
 *   - plain C, opaque handles, explicit sizes; no C++/torch types cross the boundary;
 *   - "dev" pointers are device (HIP) addresses owned by the caller; "host" pointers
 *     are ordinary host memory; the library never frees caller memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     asynchronous on that stream unless stated otherwise;
 *   - every function returns 0 on success and a negative mtsamd_status on failure;
 *     mtsamd_last_error() returns a thread-local description (the reference throws
 *     std::runtime_error via Throw(), include/mitsuba/core/logger.h:155-159);
 *   - handles are externally synchronised (one render at a time per handle), except
 *     mtsamd_cancel() which may be called from any thread
 *     (Integrator::cancel, include/mitsuba/render/integrator.h:44-51).
 */
#ifndef MTSAMD_H
#define MTSAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTSAMD_ABI_VERSION 6
/* the library is built with -fvisibility=hidden: only the entry points below are exported */
#define MTSAMD_API __attribute__((visibility("default")))

typedef enum {
    MTSAMD_OK = 0,
    MTSAMD_ERR_INVALID = -1,   /* bad argument (the reference would Throw) */
    MTSAMD_ERR_DEVICE = -2,    /* HIP runtime error */
    MTSAMD_ERR_NOMEM = -3,
    MTSAMD_ERR_CANCELLED = -4, /* render stopped by mtsamd_cancel (render() returns false) */
    MTSAMD_ERR_UNSUPPORTED = -5
} mtsamd_status;

typedef struct mtsamd_scene mtsamd_scene;

/* ---- library ------------------------------------------------------------- */
MTSAMD_API int mtsamd_abi_version(void);
MTSAMD_API const char *mtsamd_last_error(void);
/* Number of HIP devices visible to this process (<0 on error). */
MTSAMD_API int mtsamd_device_count(void);
/* Plugin ABI of the reference (include/mitsuba/core/class.h:205-211, MTS_EXPORT_PLUGIN): PluginManager dlopen()s a
 * plugin .so and reads these two symbols (src/libcore/plugin.cpp:19-31).  The library answers for the `path_amd`
 * integrator shim shown in INTEGRATION.md, so that the shim can be this very shared object. */
MTSAMD_API const char *plugin_name(void);
MTSAMD_API const char *plugin_descr(void);

/* ---- scene description ----------------------------------------------------
 * Mesh buffers exactly as Mesh exposes them (include/mitsuba/render/mesh.h:80-90,
 * 328-332): packed xyz positions, optional packed normals / uv texcoords, u32 faces.
 * Geometry must already be in world space (the OBJ/PLY loaders bake to_world at load
 * time, src/shapes/obj.cpp:94-342). */
typedef struct {
    uint32_t vertex_count;
    uint32_t face_count;
    const float *positions;    /* host, 3 * vertex_count */
    const float *normals;      /* host, 3 * vertex_count, or NULL */
    const float *texcoords;    /* host, 2 * vertex_count, or NULL */
    const uint32_t *faces;     /* host, 3 * face_count */
    int32_t bsdf;              /* index into the bsdf table */
    int32_t emitter;           /* index into the emitter table, or -1 */
} mtsamd_mesh_desc;

/* BSDF plugins (constructor parameters after the host resolved defaults and named IORs):
 *   diffuse src/bsdfs/diffuse.cpp, conductor conductor.cpp, roughconductor roughconductor.cpp + microfacet.h,
 *   dielectric dielectric.cpp, roughdielectric roughdielectric.cpp, plastic plastic.cpp, roughplastic roughplastic.cpp (isotropic alpha_u);
 *   `twosided` = wrapped in the TwoSidedBRDF adapter (twosided.cpp). */
typedef enum { MTSAMD_BSDF_DIFFUSE = 0, MTSAMD_BSDF_CONDUCTOR = 1, MTSAMD_BSDF_ROUGHCONDUCTOR = 2, MTSAMD_BSDF_DIELECTRIC = 3,
               MTSAMD_BSDF_PLASTIC = 4, MTSAMD_BSDF_ROUGHPLASTIC = 5,
               MTSAMD_BSDF_ROUGHDIELECTRIC = 6,
               MTSAMD_BSDF_THINDIELECTRIC = 7 /* src/bsdfs/thindielectric.cpp: delta reflection + null transmission of a thin slab */,
               /* src/bsdfs/blendbsdf.cpp (weight * nested[1] + (1 - weight) * nested[0]) and src/bsdfs/mask.cpp (opacity * nested[0] +
                * a null lobe): `nested` index plain records of the same table (constant parameters; RGB variant: also a textured reflectance); the weight / opacity
                * is reflectance[0], or Texture::eval_1 of `texture` (luminance of a bitmap texel, bitmap.cpp:215-231; first colour
                * channel of a checkerboard cell).  `twosided` wraps a whole blend (mask transmits: twosided.cpp:78-80 refuses it). */
               MTSAMD_BSDF_BLEND = 8, MTSAMD_BSDF_MASK = 9 } mtsamd_bsdf_type;
typedef struct {
    int32_t type;              /* mtsamd_bsdf_type */
    float reflectance[3];      /* diffuse.reflectance / plastic.diffuse_reflectance: constant `srgb` value (src/spectra/srgb.cpp:27-52) */
    int32_t texture;           /* diffuse.reflectance as a bitmap: index into the texture table or -1 (src/textures/bitmap.cpp) */
    int32_t twosided;          /* != 0: both sides scatter like the front side (twosided.cpp:94-175) */
    float specular_reflectance[3];     /* default 1 */
    float specular_transmittance[3];   /* dielectric, default 1 */
    float eta[3], k[3];        /* conductors: complex index of refraction per colour channel (material "none": 0, 1) */
    float int_ior, ext_ior;    /* dielectric / plastic (ior.h) */
    float alpha_u, alpha_v;    /* roughconductor roughness */
    int32_t distribution;      /* 0 = beckmann, 1 = ggx */
    int32_t sample_visible;    /* visible-normal sampling (default true) */
    int32_t nonlinear;         /* plastic.nonlinear */
    int32_t uniform_mask;      /* spectral variant: bit 0 / 1 / 2 = reflectance / specular_reflectance / specular_transmittance is a
                                  constant (`uniform` spectrum, xml.cpp:1069-1083) rather than an RGB colour (`srgb`, upsampled);
                                  conductors need the same eta and k in all three channels (uniform) */
    int32_t nested[2];         /* MTSAMD_BSDF_BLEND: bsdf_0, bsdf_1; MTSAMD_BSDF_MASK: nested_bsdf, -1; otherwise ignored (ABI 5) */
} mtsamd_bsdf_desc;

/* area: src/emitters/area.cpp (attached to a mesh); constant: src/emitters/constant.cpp and envmap: src/emitters/envmap.cpp
 * (environment emitters, at most one per scene; spectral variant: `constant` radiance is upsampled like an area light's,
 * `envmap` texels become (model coefficients, scale) as in envmap.cpp:96-109) */
/* delta emitters: point src/emitters/point.cpp, spot spot.cpp (without projection texture), directional directional.cpp */
typedef enum { MTSAMD_EMITTER_AREA = 0, MTSAMD_EMITTER_CONSTANT = 1, MTSAMD_EMITTER_ENVMAP = 2, MTSAMD_EMITTER_POINT = 3,
               MTSAMD_EMITTER_SPOT = 4, MTSAMD_EMITTER_DIRECTIONAL = 5 } mtsamd_emitter_type;
typedef struct {
    int32_t type;              /* mtsamd_emitter_type; AreaLight = src/emitters/area.cpp */
    float radiance[3];         /* area / constant: radiance; point / spot: intensity; directional: irradiance */
    /* envmap: latitude-longitude image, linear RGB, host pointer (height * width * 3), `scale`, and the emitter's to_world
     * (row-major 4x4; the linear part is used) */
    const float *envmap_data;
    int32_t envmap_width, envmap_height;
    float envmap_scale;
    float to_world[16];        /* envmap; point / spot: position = translation; spot / directional: orientation (the light points
                                  along the local +z axis: spot.cpp:129-151, directional.cpp:104-129) */
    float cutoff_angle, beam_width;   /* spot, degrees (spot.cpp:81-82: defaults 20 and 3/4 of the cutoff angle, resolved by the host) */
} mtsamd_emitter_desc;

typedef struct {
    int32_t width, height;     /* bitmap (src/textures/bitmap.cpp): texels; channels = 3 (RGB) */
    const float *data;         /* host, height*width*3 (NULL for a checkerboard) */
    int32_t kind;              /* 0 = bitmap, 1 = checkerboard (src/textures/checkerboard.cpp) with constant colours */
    float color0[3], color1[3];
    float to_uv[6];            /* (m00, m01, m02, m10, m11, m12) of the extracted to_uv transform; all zero = identity */
} mtsamd_texture_desc;

typedef struct {
    const mtsamd_mesh_desc *meshes;       uint32_t mesh_count;
    const mtsamd_bsdf_desc *bsdfs;        uint32_t bsdf_count;
    const mtsamd_emitter_desc *emitters;  uint32_t emitter_count;
    const mtsamd_texture_desc *textures;  uint32_t texture_count;
    /* variant: 0 = *_rgb, 1 = *_spectral (4 wavelength samples, mitsuba.conf.template:135-138).  In spectral mode
     * RGB reflectances become SRGBReflectanceSpectrum and RGB radiances SRGBEmitterSpectrum (src/libcore/xml.cpp:1045-1146,
     * src/spectra/srgb.cpp, srgb_d65.cpp) through the coefficient table at rgb2spec_path ("data/srgb.coeff" in the
     * reference, src/librender/srgb.cpp:14-40; see mtsamd_rgb2spec_build). */
    int32_t spectral;
    const char *rgb2spec_path;
} mtsamd_scene_desc;

/* Scene::Scene + accel_init (src/librender/scene.cpp:22-98): uploads the geometry to
 * `device`, builds the BVH (replaces ShapeKDTree::build, include/mitsuba/render/kdtree.h:1710)
 * and the emitter sampling tables (Mesh::area_distr_build, src/librender/mesh.cpp:284-307).
 * Synchronous. */
MTSAMD_API int mtsamd_scene_create(const mtsamd_scene_desc *desc, int device, mtsamd_scene **out);
MTSAMD_API void mtsamd_scene_destroy(mtsamd_scene *scene);

/* Scene::bbox (include/mitsuba/render/scene.h): out6 = min xyz, max xyz (host). */
MTSAMD_API int mtsamd_scene_bbox(const mtsamd_scene *scene, float *out6);
/* Scene info: out[0]=primitive count, [1]=BVH node count, [2]=BVH depth, [3]=shape count,
 * [4]=emitter count, [5]=nodes resident in LDS. */
MTSAMD_API int mtsamd_scene_info(const mtsamd_scene *scene, uint32_t *out6);
/* parameters_changed() for constant reflectance / radiance / texture data
 * (src/spectra/srgb.cpp:59-61, src/textures/bitmap.cpp:295-299): host data, synchronous.
 * Spectral variant: the colours are `srgb` / `srgb_d65` spectra -- the setters redo the range check and the
 * coefficient fetch of scene creation (srgb.cpp:31-41, srgb_d65.cpp:31-46); a parameter that was given as a
 * `uniform` spectrum or a texture is not settable this way (MTSAMD_ERR_UNSUPPORTED). */
MTSAMD_API int mtsamd_scene_set_bsdf_reflectance(mtsamd_scene *scene, uint32_t bsdf, const float *rgb);
MTSAMD_API int mtsamd_scene_set_emitter_radiance(mtsamd_scene *scene, uint32_t emitter, const float *rgb);
/* BitmapTexture `data` parameter (bitmap.cpp:295-299): rgb is a host OR device pointer to height*width*3 floats;
 * asynchronous on `stream`. */
MTSAMD_API int mtsamd_scene_update_texture(mtsamd_scene *scene, uint32_t texture, const float *rgb, void *stream);

/* ---- scene queries on SoA ray streams --------------------------------------
 * The stream layout follows the reference's device-stream precedent OptixParams
 * (include/mitsuba/render/optix/common.h:15-35): one float array per component,
 * one entry per ray, optional byte mask; inactive or missed lanes get t = +inf,
 * prim/shape = 0xffffffff (src/librender/optix/optix_rt.cu:35-37).
 * All array arguments are device pointers of `n` elements. */
typedef struct {
    const float *ox, *oy, *oz, *dx, *dy, *dz, *mint, *maxt;
    const uint8_t *active;     /* may be NULL */
} mtsamd_rays;

/* Scene::ray_intersect (include/mitsuba/render/scene.h:36; closest hit,
 * ShapeKDTree::ray_intersect_scalar<false>, kdtree.h:2079-2174): t, global primitive
 * index, shape index, barycentric u,v (the kd-tree "cache", kdtree.h:2432-2452). */
MTSAMD_API int mtsamd_ray_intersect(const mtsamd_scene *scene, uint64_t n, const mtsamd_rays *rays,
                         float *t, uint32_t *prim, uint32_t *shape, float *u, float *v,
                         void *stream);
/* Same query answered by brute force over all triangles
 * (Scene::ray_intersect_naive, scene.h:38-44 / kdtree.h:2303-2328) -- test aid. */
MTSAMD_API int mtsamd_ray_intersect_naive(const mtsamd_scene *scene, uint64_t n, const mtsamd_rays *rays,
                               float *t, uint32_t *prim, uint32_t *shape, float *u, float *v,
                               void *stream);
/* Scene::ray_test (scene.h:62; any hit, ray_intersect_scalar<true>). */
MTSAMD_API int mtsamd_ray_test(const mtsamd_scene *scene, uint64_t n, const mtsamd_rays *rays,
                    uint8_t *hit, void *stream);
/* Full SurfaceInteraction SoA for the closest hit (create_surface_interaction,
 * kdtree.h:2334-2367 + Mesh::fill_surface_interaction, src/librender/mesh.cpp:399-462;
 * GPU twin __closesthit__mesh, src/shapes/optix/mesh.cuh:27-96).  si26 is a device array of
 * 26 planes of n floats each: p(3) n(3) uv(2) sh_frame.s(3) sh_frame.t(3) sh_frame.n(3)
 * dp_du(3) dp_dv(3) wi(3). */
MTSAMD_API int mtsamd_ray_intersect_si(const mtsamd_scene *scene, uint64_t n, const mtsamd_rays *rays,
                            float *t, uint32_t *prim, uint32_t *shape, float *si26,
                            void *stream);

/* ---- sensor / film / sampler / integrator ----------------------------------- */
/* src/rfilters/{gaussian,box,tent,catmullrom,mitchell,lanczos}.cpp */
typedef enum { MTSAMD_RFILTER_GAUSSIAN = 0, MTSAMD_RFILTER_BOX = 1, MTSAMD_RFILTER_TENT = 2, MTSAMD_RFILTER_CATMULLROM = 3,
               MTSAMD_RFILTER_MITCHELL = 4, MTSAMD_RFILTER_LANCZOS = 5 } mtsamd_rfilter_type;

typedef struct {
    /* PerspectiveCamera (src/sensors/perspective.cpp): to_world is row-major 4x4 */
    float to_world[16];
    float fov_x_deg;           /* horizontal fov in degrees (after parse_fov, src/librender/sensor.cpp:119-169) */
    float near_clip, far_clip; /* defaults 1e-2 / 1e4 (sensor.cpp:100-102) */
    /* Film (src/librender/film.cpp:7-64): size and crop window */
    int32_t film_width, film_height;
    int32_t crop_x, crop_y, crop_width, crop_height;
    /* ReconstructionFilter: gaussian stddev (src/rfilters/gaussian.cpp) / box radius (box.cpp) / mitchell B (mitchell.cpp:33) /
     * lanczos lobes (lanczos.cpp:34); param2: mitchell C; tent and catmullrom take no parameter */
    int32_t rfilter;           /* mtsamd_rfilter_type */
    float rfilter_param, rfilter_param2;
    int32_t rfilter_analytic;  /* 0: eval_discretized (scalar/packet variants, imageblock.cpp:131);
                                  1: eval() as the reference's GPU variants do (:132) */
    /* IndependentSampler (src/samplers/independent.cpp): sample_count, seed */
    int32_t sample_count;
    uint64_t seed;
    /* MonteCarloIntegrator (src/librender/integrator.cpp:283-296) */
    int32_t max_depth;         /* -1 = unbounded */
    int32_t rr_depth;          /* default 5 */
    /* work partition (multi-GPU film partition): only pixels of the rows this call owns are sampled;
     * their splats still land in the full crop-sized film, so the per-GPU films simply add up.
     * Either a window of crop-relative rows [row_begin,row_end) (row_end <= 0: all rows), or -- with
     * part_count > 1 -- tiles of part_tile_rows rows (32 = MTS_BLOCK_SIZE, include/mitsuba/render/spiral.h:10)
     * dealt round-robin: this call owns tiles t with t % part_count == part_index. */
    int32_t row_begin, row_end;
    int32_t part_index, part_count, part_tile_rows;
    /* scheduler knobs (0 = library default) */
    int32_t paths_per_wave;    /* in-flight path slots per scheduling wave */
    int32_t pipeline;          /* 0 = automatic: schedule 4 for LDS-resident scenes (<= 64 primitives), split trace / shade /
                                  trace kernels for hierarchy scenes; 1 = force the fused bounce kernel; 2 = force split;
                                  3 = LDS-resident scenes only: closest hit fused with shading, shadow rays queued and
                                  resolved in dense batches by a second kernel; 4 = LDS-resident scenes only: one kernel,
                                  shadow rays collected in a per-wave LDS ring and resolved 64 at a time (full waves in
                                  the any-hit loop).  All produce identical samples. */
    int32_t film_rgb;          /* 0: film channels X,Y,Z,A,W (integrator.cpp:72-74, 254-268);
                                  1: R,G,B,A,W -- linear RGB as mitsuba.python.autodiff._render_helper accumulates (autodiff.py:53-72) */
    int32_t integrator;        /* 0 = path (src/integrators/path.cpp), 1 = direct (direct.cpp), 2 = depth (depth.cpp) */
    int32_t emitter_samples;   /* direct: samples per technique; 0 and 0 = the shading_samples default (1, 1) (direct.cpp:80-103) */
    int32_t bsdf_samples;
    int32_t hide_emitters;     /* direct: do not add directly visible emitters (integrator.cpp:39, direct.cpp:117-121) */
    int32_t moment;            /* != 0: `moment` integrator around the selected one (src/integrators/moment.cpp:56-99): the film
                                  has 11 channels X,Y,Z,A,W, nested.X,nested.Y,nested.Z, m2_nested.X,m2_nested.Y,m2_nested.Z */
    /* ThinLensCamera (src/sensors/thinlens.cpp:110-118): aperture_radius > 0 selects the thin lens model -- two more sampler
     * dimensions per camera sample (integrator.cpp:229-231) -- focused at focus_distance (sensor.cpp:104); 0: pinhole */
    float aperture_radius, focus_distance;
    /* SamplingIntegrator properties (integrator.cpp:27-39) */
    float timeout;             /* seconds; <= 0: none.  Once exceeded no further pass is started and the running one is abandoned
                                  (should_stop(), integrator.h:143-146); the call still returns MTSAMD_OK: only cancel() makes
                                  render() return false (integrator.cpp:175) */
    int32_t samples_per_pass;  /* <= 0: all of sample_count.  sample_count must be a multiple of it (integrator.cpp:59-66); the
                                  image does not depend on it: the RNG streams are seeded per global sample index */
    int32_t profile;           /* != 0: every trace / shade launch of the split pipeline is bracketed by HIP timing events on
                                  the stream it is launched on (stats_host[8..13]) */
    /* scheduler knobs, continued (ABI 6; 0 = library default).  The image does not depend on them (tests/test_gpu_lifecycle.py). */
    int32_t max_pass_log2;     /* a pass holds at most 2^max_pass_log2 camera samples (10..30; default 30) */
    int32_t finish_kernel;     /* end of a pass: 0 = one k_finish launch once few paths are left, 1 = never (launch rounds until the
                                  pool is empty), 2 = as soon as the sample cursors are dry */
} mtsamd_render_desc;

/* SamplingIntegrator::render for the `path` integrator (src/librender/integrator.cpp:52-176,
 * src/integrators/path.cpp:100-211) followed by Film::put: renders the crop window and ADDS
 * the result into film_xyzaw_dev (crop_height*crop_width*5 floats, channels X,Y,Z,A,W as
 * prepared at integrator.cpp:72-74; zero it first for a fresh image).
 * Seeding follows the reference's wavefront branch (one PCG32 stream per sample index,
 * integrator.cpp:144-169, independent.cpp:69-72) with samples_per_pass = sample_count.
 * Synchronous on `stream` (returns when the film is complete).
 * stats_host (may be NULL, 16 entries): [0] closest-hit queries, [1] any-hit queries, [2] camera samples,
 * [3] scheduler iterations (k_bounce launches), [4] path segments shaded, [5] device time of the bounce loop in ns
 * (HIP events on `stream`), [6] device time of the film gather in ns, [7] triangle tests; with desc->profile and the
 * split pipeline: [8] summed duration of the k_trace<closest> launches in ns, [9] their number, [10] / [11] the same for
 * k_trace<any>, [12] / [13] for k_shade; [14] passes; [15] 1 if the render stopped at its timeout. */
MTSAMD_API int mtsamd_render(mtsamd_scene *scene, const mtsamd_render_desc *desc, float *film_xyzaw_dev,
                  uint64_t *stats_host, void *stream);
/* Integrator::cancel (integrator.h:51): thread-safe, makes a running mtsamd_render return
 * MTSAMD_ERR_CANCELLED at the next scheduling step. */
MTSAMD_API int mtsamd_cancel(mtsamd_scene *scene);

/* SamplingIntegrator::sample for whole sample indices (integrator.h:114-119): per-sample radiance
 * of samples [first, first+count) of the render described by desc, without film accumulation.
 * rgba_dev: count*4 floats (R,G,B, valid_ray mask); pos_dev (may be NULL): count*2 floats film
 * position sample.  Synchronous. */
MTSAMD_API int mtsamd_sample_radiance(mtsamd_scene *scene, const mtsamd_render_desc *desc, uint64_t first,
                           uint64_t count, float *rgba_dev, float *pos_dev, void *stream);

/* Reverse-mode derivative of mitsuba.python.autodiff.render (src/python/python/autodiff.py:6-91,121-194) with respect
 * to diffuse reflectances -- what `ek.backward()` propagates into 'bsdf.reflectance.value' (src/spectra/srgb.cpp:59-61)
 * and 'bsdf.reflectance.data' (src/textures/bitmap.cpp:295-299).  The image is values / (weight + 1e-8) of a film
 * rendered with `desc` and desc->film_rgb = 1 (channels R,G,B,A,W); dloss_dimage_dev holds dLoss/dImage
 * (crop_height*crop_width*3), film_dev that primal film.  Paths are replayed with the same per-sample PCG32 streams.
 * grad_bsdf_dev (bsdf_count*3), grad_textures_dev (all textures concatenated in index order, see
 * mtsamd_scene_texture_info) and grad_emitters_dev (emitter_count*3: the radiance of area lights,
 * 'shape.emitter.radiance.value', docs/src/inverse_rendering/diff_render.rst:76) are ACCUMULATED into; each may be NULL.
 * Needs 0 <= max_depth <= 16. */
MTSAMD_API int mtsamd_render_adjoint(mtsamd_scene *scene, const mtsamd_render_desc *desc, const float *dloss_dimage_dev,
                          const float *film_dev, float *grad_bsdf_dev, float *grad_textures_dev, float *grad_emitters_dev,
                          void *stream);
/* The same derivative with respect to the texels of the `envmap` emitter -- 'my_envmap.data' (src/emitters/envmap.cpp:214-218), the
 * parameter docs/examples/10_inverse_rendering/invert_bunny.py optimises.  The radiance is linear in the texels; the sampling
 * distribution built from their luminances is not differentiated (envmap.cpp:220-253 rebuilds it from plain floats).  Any BSDF,
 * any max_depth, RGB variant.  grad_envmap_dev: envmap height * width * 3 floats, ACCUMULATED into. */
MTSAMD_API int mtsamd_render_adjoint_envmap(mtsamd_scene *scene, const mtsamd_render_desc *desc, const float *dloss_dimage_dev,
                                 const float *film_dev, float *grad_envmap_dev, void *stream);
/* New texels for the envmap emitter (host pointer, height * width * 3 linear RGB; parameters_changed of envmap.cpp:220-253).
 * rebuild_distribution = 0 keeps the importance-sampling hierarchy of the previous texels (a render is then exactly linear in the
 * texels: finite-difference tests); the reference always rebuilds.  Synchronises the device.  RGB variant. */
/* Parameters of the BSDF models beyond `diffuse` (what traverse() exposes of e.g. roughconductor.cpp:393-404, plastic.cpp:299-307):
 * REFLECTANCE = diffuse.reflectance / (rough)plastic.diffuse_reflectance, SPECULAR_REFLECTANCE, SPECULAR_TRANSMITTANCE (dielectrics),
 * ETA / K (conductors), ALPHA (isotropic roughness of roughconductor / roughdielectric; one component). */
typedef enum { MTSAMD_PARAM_REFLECTANCE = 0, MTSAMD_PARAM_SPECULAR_REFLECTANCE = 1, MTSAMD_PARAM_ETA = 2, MTSAMD_PARAM_K = 3, MTSAMD_PARAM_ALPHA = 4,
               MTSAMD_PARAM_SPECULAR_TRANSMITTANCE = 5 } mtsamd_bsdf_param;
/* parameters_changed() of a BSDF after one of these parameters was edited (constants): value3 = 3 floats (ALPHA: 1).  Spectral
 * variant: the three colour kinds become srgb spectra (as above), ETA / K must keep three equal components (uniform spectra). */
MTSAMD_API int mtsamd_scene_set_bsdf_param(mtsamd_scene *scene, uint32_t bsdf, int32_t param, const float *value3);
/* d(loss)/d(component of one such parameter) ADDED to *grad1_dev, for the image of mtsamd_render(film_rgb = 1) normalised as
 * mitsuba.python.autodiff.render does (src/python/python/autodiff.py:6-91): every camera sample is replayed with its PCG32 stream through
 * the general path step -- any BSDF, any emitter, any depth -- carrying the derivative forward beside the path.  Sampling is detached:
 * directions, lobe choices, MIS weights and Russian roulette are those of the primal path; the BSDF value at the fixed directions is
 * differentiated by a central difference of the model code (step h; <= 0: 1 % of the value).  The reference differentiates the attached
 * estimator through Enoki's autodiff graph; both are unbiased estimators of the same derivative (up to O(h^2)). */
MTSAMD_API int mtsamd_render_adjoint_param(mtsamd_scene *scene, const mtsamd_render_desc *desc, const float *dloss_dimage_dev,
                                const float *film_dev, uint32_t bsdf, int32_t param, int32_t component, float h,
                                float *grad1_dev, void *stream);
MTSAMD_API int mtsamd_scene_update_envmap(mtsamd_scene *scene, const float *rgb, int32_t rebuild_distribution);
/* Size of a bitmap texture and its float offset inside the concatenated texture-gradient buffer. */
/* RoughPlastic precomputation (roughplastic.cpp:380-399) of BSDF `bsdf`: out65[0..63] = external transmittance at
 * cos(theta) = i / 63, out65[64] = internal diffuse reflectance.  Host pointer. */
MTSAMD_API int mtsamd_scene_roughplastic_tables(const mtsamd_scene *scene, uint32_t bsdf, float *out65);
MTSAMD_API int mtsamd_scene_texture_info(const mtsamd_scene *scene, uint32_t texture, int32_t *width, int32_t *height,
                              uint64_t *grad_offset);

/* Generates the RGB -> spectrum coefficient table (the reference builds it at compile time with
 * ext/rgb2spec/rgb2spec_opt.cpp <resolution = 64> srgb.coeff, ext/rgb2spec/CMakeLists.txt:49-54) and writes it to
 * `path` in the same "SPEC" file format.  Host only; takes about a minute at resolution 64 on 8 threads. */
MTSAMD_API int mtsamd_rgb2spec_build(const char *path, int32_t resolution, int32_t threads);
/* srgb_model_fetch (src/librender/srgb.cpp:14-40): coefficients of the smooth spectrum for a linear sRGB colour. */
MTSAMD_API int mtsamd_srgb_model_fetch(const char *path, const float *rgb3, float *coeff3);

/* PerspectiveCamera::sample_ray (perspective.cpp:153-188) / ThinLensCamera::sample_ray (thinlens.cpp:175-214) for n
 * film-plane samples in [0,1)^2 and, for a thin lens, n aperture samples (NULL: 0.5, as integrator.cpp:229 initialises them)
 * (device SoA in, device SoA out). */
MTSAMD_API int mtsamd_camera_sample_rays(const mtsamd_render_desc *desc, uint64_t n, const float *sx,
                              const float *sy, const float *aperture_x, const float *aperture_y, float *ox, float *oy, float *oz, float *dx,
                              float *dy, float *dz, float *mint, float *maxt, void *stream);

/* ---- ImageBlock / Film ------------------------------------------------------- */
/* ImageBlock::put(pos, value) (src/librender/imageblock.cpp:80-172) for n samples:
 * block of (width,height) at (offset_x,offset_y) with `border` pixels of apron
 * (0, or the filter's border_size); data_dev holds (height+2b)*(width+2b)*channels floats
 * and is accumulated into (float atomics, as the reference's scatter_add).
 * pos_dev: n*2 floats, values_dev: n*channels floats. */
MTSAMD_API int mtsamd_imageblock_put(int32_t width, int32_t height, int32_t offset_x, int32_t offset_y,
                          int32_t channels, int32_t rfilter, float rfilter_param, float rfilter_param2,
                          int32_t rfilter_analytic, int32_t border, uint64_t n,
                          const float *pos_dev, const float *values_dev, float *data_dev,
                          void *stream);
/* ImageBlock::put(const ImageBlock*) (imageblock.cpp:49-77, accumulate_2d bitmap.h:657-716):
 * target += source with clipping; both described by (w,h,offset,border). */
MTSAMD_API int mtsamd_imageblock_put_block(const float *src_dev, int32_t src_w, int32_t src_h, int32_t src_ox,
                                int32_t src_oy, int32_t src_border, float *dst_dev, int32_t dst_w,
                                int32_t dst_h, int32_t dst_ox, int32_t dst_oy, int32_t dst_border,
                                int32_t channels, void *stream);
/* ReconstructionFilter discretisation (src/libcore/rfilter.cpp:9-20): host outputs. */
MTSAMD_API int mtsamd_rfilter_info(int32_t rfilter, float rfilter_param, float rfilter_param2, float *table32_host,
                        float *radius_host, int32_t *border_host);
/* HDRFilm::bitmap (src/films/hdrfilm.cpp:249-320): XYZAW -> RGBA float32, n pixels. */
MTSAMD_API int mtsamd_film_develop(const float *xyzaw_dev, uint64_t n_pixels, float *rgba_dev, void *stream);
/* Elementary functions of the kernels (csrc/device_libm.h: the reference gets them from Enoki's polynomial kernels, e.g.
 * enoki::sincos in include/mitsuba/core/warp.h:54-90, exp / log / erf in include/mitsuba/render/microfacet.h:187-493), evaluated on
 * the device for n arguments.  fn: 0 sin, 1 cos, 2 tan, 3 exp, 4 log, 5 erf, 6 acos, 7 atan2(x = y-coordinate, y = x-coordinate), 8 atanh, 9 cosh;
 * y_dev is read by fn 7 only.  Lets a test prove that host and device produce the same bits. */
MTSAMD_API int mtsamd_libm_eval(int32_t fn, uint64_t n, const float *x_dev, const float *y_dev, float *out_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif
