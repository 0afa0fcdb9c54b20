// kernels.h -- host-visible launch interface of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "device_scene.h"
#include "device_spectral.h"

namespace mtsamd {

// Path-state streams: SoA of 16-byte vectors, one slot per in-flight path (88 B per path).
//   ray_o = (o.xyz, mint)  ray_d = (d.xyz, maxt)  thr = (throughput rgb, bs_pdf)
//   res = (radiance rgb, eta)  rng = PCG32 (state, inc)  misc = (sample ordinal, depth | flags << 16)
struct PoolView {
    float4 *ray_o, *ray_d, *thr, *res;
    uint4 *rng;
    uint2 *misc;
    // spectral variant (100 B per path): thr / res hold 4 spectral samples, aux = (bs_pdf, eta), xi = the wavelength sample the path's
    // four wavelengths are a function of (wavelengths_from_sample; round 2 stored the wavelengths: 112 B)
    float *xi;
    float2 *aux;
    // split pipeline (hierarchy scenes): hit = (t, prim bits, u, v) of the path's ray; per-wave dense queue of
    // pending shadow rays sh_o = (o, mint), sh_d = (d, maxt), the contribution `nee` each guards and the slot it belongs to
    float4 *hit, *sh_o, *sh_d, *nee;
    uint32_t *sh_slot;
};

// Film rows owned by one render call.  count <= 1: the contiguous window [row0, row0 + local_rows);
// count > 1: the film is cut into tiles of tile_rows rows dealt round-robin, this call owns tiles t % count == part.
struct RowMap { int32_t row0, local_rows, tile_rows, part, count; };
MTS_DEV int32_t row_to_global(const RowMap &m, int32_t lr) {
    if (m.count <= 1) return m.row0 + lr;
    int32_t t = lr / m.tile_rows;
    return (t * m.count + m.part) * m.tile_rows + (lr - t * m.tile_rows);
}
MTS_DEV int32_t row_to_local(const RowMap &m, int32_t gr) {     // -1: not owned
    if (m.count <= 1) { int32_t lr = gr - m.row0; return (lr >= 0 && lr < m.local_rows) ? lr : -1; }
    int32_t t = gr / m.tile_rows;
    if (t % m.count != m.part) return -1;
    return (t / m.count) * m.tile_rows + (gr - t * m.tile_rows);
}

struct FilterView {
    float table[32];
    float radius, scale_factor, alpha, bias;
    int32_t kind, analytic, border, taps;   // taps = n in imageblock.cpp:123
};

struct RenderParams {
    SceneView sv;
    CameraView cam;
    PoolView in, out;
    const uint32_t *count_in;   // per scheduling wave
    uint32_t *count_out;
    uint32_t *count_shadow;     // split pipeline: queued shadow rays per scheduling wave (output pool)
    uint64_t *cursor;           // per wave: how many of its samples it has generated (kernels.hip cursor_sample maps them to ordinals)
    const uint64_t *cursor_end;
    uint64_t *wave_stats;       // per wave: closest, any, segments, tri tests
    float4 *out_rgba;           // per sample ordinal: radiance rgb + valid_ray
    float2 *out_pos;            // per sample ordinal: film position sample
    uint32_t chunk;             // samples per chunk dealt to a scheduling wave (kernels.hip, cursor_sample)
    uint32_t n_chains;          // 0 / 1, or the number of launch chains the scheduling waves are cut into (hierarchy scenes): the chunks are dealt
                                // to the chains in turn (chunk_owner), so that every chain sees the same mix of pixels
    uint32_t first_pix, first_rem;       // first_ordinal = first_pix * spp + first_rem (film renders: passes start on a pixel, first_rem == 0)
    uint64_t first_ordinal;     // local sample ordinal of slot 0 of out_rgba / out_pos
    uint64_t base_seed;
    RowMap rows;                // which film rows this render owns (multi-GPU film partition)
    int32_t store_xyz;          // 1: out_rgba holds (X,Y,Z,alpha | -1 if the sample is invalid) for the film, 2: same with
                                // linear RGB instead of XYZ, 0: (R,G,B,alpha) for the per-sample API
    // sample-stream slot of (local pixel lp, sample j): plane_pixels == 0: ordinal - first_ordinal (pixel-major);
    // else j * plane_pixels + (lp - plane_pix0): one plane per sample number, so that a film tile reads contiguous pixels
    uint32_t plane_pix0, plane_pixels;
    uint32_t n_waves, seg_cap, target;
    int32_t spp, crop_x, crop_y, crop_w, crop_h;
    int32_t max_depth, rr_depth;
    int32_t spectral;           // 0: RGB variant, 1: spectral variant (4 wavelengths per sample)
    int32_t integrator, emitter_samples, bsdf_samples, hide_emitters;   // 0 path; 1 direct (direct.cpp); 2 depth (depth.cpp)
    int32_t split;              // 0: fused k_bounce, 1: k_trace<closest> + k_shade + k_trace<any> per iteration, 2: k_shade (flat) + k_trace<any>,
                                // 3: k_shade (flat) with the in-kernel shadow ring
    uint32_t lds_queue_offset;  // split == 3: start of the per-wave shadow rings in LDS, in float4 units
    uint32_t wave_first, wave_last;   // k_shade: scheduling waves covered by this launch (wave_last == 0: all)
    uint32_t gather_w;          // k_shade on LDS-resident scenes: a workgroup gathers the paths of gather_w (4, 16, ... 1024) consecutive scheduling
                                // waves into the first four (pool drain: the sample cursors of the launch are dry); 0 / 4: no gathering
    uint32_t trace_lds_depth;   // k_trace: stack entries per lane kept in LDS; deeper ones go to trace_spill
    uint32_t trace_top_nodes;   // k_trace: BVH4 nodes [0, trace_top_nodes) are staged in LDS by every workgroup
    uint32_t *trace_spill;      // k_trace: [workgroup][entry][thread]
};

// Launch chains of the split pipeline: chain k covers the scheduling waves [chain_first(k), chain_first(k + 1)); the boundaries are
// multiples of kChainAlign (itself a multiple of the k_trace group size: a group never straddles two launches).
constexpr uint32_t kMaxChains = 8u, kChainAlign = 64u;
constexpr uint32_t kTraceChains = 2u;       // chains a render runs
__host__ __device__ inline uint32_t chain_first(uint32_t k, uint32_t n, uint32_t chains) {
    if (k >= chains) return n;
    const uint32_t lo = ((uint32_t) ((uint64_t) n * k / chains) + kChainAlign - 1u) & ~(kChainAlign - 1u);
    return lo < n ? lo : n;
}
// Index of the (first) chunk owned by scheduling wave `wave`: the identity, or -- `chains` launch chains -- chunk i * chains + k for
// the i-th wave of chain k as long as every chain has an i-th wave; the few waves beyond that take the remaining chunks in order.
// A bijection of [0, n).
__host__ __device__ inline uint32_t chunk_owner(uint32_t wave, uint32_t n, uint32_t chains) {
    if (chains <= 1u) return wave;
    uint32_t s_min = n, k = 0u, lo_k = 0u, extra_before = 0u;
    for (uint32_t c = 0; c < chains; ++c) {
        const uint32_t lo = chain_first(c, n, chains), hi = chain_first(c + 1u, n, chains);
        s_min = hi - lo < s_min ? hi - lo : s_min;
        if (wave >= lo && wave < hi) { k = c; lo_k = lo; }
    }
    const uint32_t i = wave - lo_k;
    if (i < s_min) return i * chains + k;
    for (uint32_t c = 0; c < k; ++c) extra_before += chain_first(c + 1u, n, chains) - chain_first(c, n, chains) - s_min;
    return s_min * chains + extra_before + (i - s_min);
}

struct FilmParams {
    const float4 *out_rgba;
    const float2 *out_pos;
    float *film;                // crop_h * crop_w * 5
    FilterView filter;
    uint64_t first_ordinal, n_samples;   // local sample ordinals held by out_rgba / out_pos (pixel-major layout)
    uint32_t plane_pix0, plane_pixels;   // plane layout (see RenderParams): the pass holds whole local rows
    RowMap rows;
    int32_t spp, crop_x, crop_y, crop_w, crop_h;
    int32_t row0, row1;         // target (global) rows [row0,row1)
    // tiled splat: the pass holds the local rows [pass_lr0, pass_lr0 + pass_rows), cut into 16x16 source tiles; every tile
    // accumulates the film pixels it reaches into its scratch tile of `partials`
    int32_t pass_lr0, pass_rows, tiles_x, tiles_y;
    int32_t tile_h;             // local rows per source tile (<= 16; 16 unless a partitioned film forces less)
    int32_t slices;             // the samples of a pixel are cut into this many runs, each accumulated by a workgroup of its own (few tiles, many samples)
    float *partials;
};

struct AdjointParams {
    RenderParams rp;            // scene, sensor, sampler, integrator of the primal render (rows = the whole crop)
    FilterView filter;
    uint64_t n_samples;         // crop_w * crop_h * spp
    const float *dimage;        // dLoss/dImage, crop_h * crop_w * 3
    const float *film;          // primal film (5 channels; only the weight channel is read)
    float *grad_bsdf;           // n_bsdfs * 3, accumulated (may be null)
    float *grad_tex;            // all textures concatenated in index order, accumulated (may be null)
    float *grad_emitter;        // n_emitters * 3 (radiance of area lights), accumulated (may be null)
    float *grad_env;            // k_adjoint_env: envmap height * width * 3, accumulated
    // k_adjoint_param: record `pg_bsdf` of the BSDF table with ONE scalar parameter at +h / -h, 1 / (2 h), and the float the derivative is added to
    int32_t pg_bsdf; DevBsdf pg_plus, pg_minus; float pg_inv_2h; float *grad_param;
};

struct RayStreams {
    const float *ox, *oy, *oz, *dx, *dy, *dz, *mint, *maxt;
    const uint8_t *active;
};

size_t bounce_lds_bytes(const SceneView &sv);
hipError_t launch_bounce(const RenderParams &p, hipStream_t s);
hipError_t launch_split_stage(const RenderParams &p, int stage, hipStream_t s);
uint32_t trace_lds_depth(const SceneView &sv);
uint32_t trace_top_nodes(const SceneView &sv);
uint32_t trace_group();          // scheduling waves per k_trace workgroup (a power of two): launch ranges start on multiples of it
size_t trace_spill_words(const SceneView &sv, uint32_t n_waves);
// `direct` / `depth` integrators: every sample of [first_ordinal, first_ordinal + n) is finished by one thread
hipError_t launch_direct(const RenderParams &p, uint64_t n, hipStream_t s);
hipError_t launch_adjoint(const AdjointParams &a, hipStream_t s);
hipError_t launch_adjoint_env(const AdjointParams &a, hipStream_t s);
hipError_t launch_adjoint_param(const AdjointParams &a, hipStream_t s);
// end of a pass: every path of p.in (counts p.count_in) is run to its end in one launch; needs dry sample cursors (kernels.hip, k_finish)
hipError_t launch_finish(const RenderParams &p, uint64_t alive, hipStream_t s);
hipError_t launch_mega(const RenderParams &p, hipStream_t s);      // small passes: the whole pass in one launch of persistent lanes
// CIE x, y, z and D65 tables (95 floats each) -> device; call once before the first spectral launch
hipError_t upload_spectral_tables(const float *x, const float *y, const float *z, const float *d65);
hipError_t launch_film_gather(const FilmParams &p, hipStream_t s);
// tiled variant for filters with <= 4 taps (gaussian stddev 0.5, box, tent): k_film_accum + k_film_merge
hipError_t launch_film_tiles(const FilmParams &p, hipStream_t s);
bool film_tiles_supported(const FilterView &f);
void film_tile_grid(FilmParams &p);                 // tiles_x / tiles_y from crop_w / pass_rows
size_t film_partial_floats(const FilmParams &p);    // size of `partials`
// mode 0: closest (BVH), 1: closest (brute force)
hipError_t launch_ray_intersect(const SceneView &sv, uint64_t n, const RayStreams &r, int mode, float *t,
                                uint32_t *prim, uint32_t *shape, float *u, float *v, float *si26,
                                hipStream_t s);
hipError_t launch_ray_test(const SceneView &sv, uint64_t n, const RayStreams &r, uint8_t *hit, hipStream_t s);
hipError_t launch_camera_rays(const CameraView &cam, uint64_t n, const float *sx, const float *sy, const float *apx, const float *apy, float *ox,
                              float *oy, float *oz, float *dx, float *dy, float *dz, float *mint, float *maxt,
                              hipStream_t s);
hipError_t launch_imageblock_put(const FilterView &f, int32_t w, int32_t h, int32_t ox, int32_t oy, int32_t ch,
                                 int32_t border, uint64_t n, const float *pos, const float *values, float *data,
                                 hipStream_t s);
hipError_t launch_put_block(const float *src, int32_t sw, int32_t sh, int32_t sox, int32_t soy, int32_t sb,
                            float *dst, int32_t dw, int32_t dh, int32_t dox, int32_t doy, int32_t db, int32_t ch,
                            hipStream_t s);
// moment integrator: squares the (X,Y,Z) of every valid sample of the stream in place; packs the two 5-channel films
// (values, squared values) into the 11-channel film of moment.cpp (accumulating)
hipError_t launch_square_stream(float4 *rgba, uint64_t n, hipStream_t s);
hipError_t launch_moment_pack(const float *values5, const float *squares5, float *film11, uint64_t n_pixels, hipStream_t s);
// roughplastic: fills the 64-entry external transmittance table of bsdfs[index] and its internal diffuse reflectance
// (DevBsdf::eb).  gl = Gauss-Legendre nodes / weights: [nodes_t(128) | weights_t(128) | nodes_r(128) | weights_r(128)]
hipError_t launch_roughplastic_tables(DevBsdf *bsdfs, uint32_t index, float *table, const float *gl, int res_t, int res_r, hipStream_t s);
hipError_t launch_film_develop(const float *xyzaw, uint64_t n, float *rgba, hipStream_t s);
hipError_t launch_libm_eval(int fn, uint64_t n, const float *x, const float *y, float *out, hipStream_t s);

} // namespace mtsamd
