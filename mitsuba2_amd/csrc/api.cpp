// api.cpp -- host side of libmtsamd: scene upload, BVH build, emitter tables, sensor/film setup,
// the wavefront scheduler and the C ABI declared in include/mtsamd.h.
//
// Reference call stack this replaces:
//   Scene::Scene -> accel_init -> ShapeKDTree::build          src/librender/scene.cpp:22-98
//   SamplingIntegrator::render (wavefront branch)              src/librender/integrator.cpp:144-169
//   PerspectiveCamera::update_camera_transforms                src/sensors/perspective.cpp:106-151
//   ReconstructionFilter::init_discretization                  src/libcore/rfilter.cpp:9-20
//   HDRFilm::prepare/put                                       src/films/hdrfilm.cpp:188-209
#include "../../include/mtsamd.h"
#include "bvh.h"
#include "envmap.h"
#include "kernels.h"
#include "spectral_upsampling.h"
#include "cie_data.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

using namespace mtsamd;

namespace {

thread_local std::string g_last_error;

// Experiment switches (scheduler geometry, BVH leaf size, ...) read from the environment exist only in builds made with
// -DMTSAMD_EXPERIMENTS (scripts/ab_build.sh); the release library never looks at the process environment.
#ifdef MTSAMD_EXPERIMENTS
static const char *exp_env(const char *name) { return std::getenv(name); }
#else
static const char *exp_env(const char *) { return nullptr; }
#endif

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? MTSAMD_ERR_NOMEM : MTSAMD_ERR_DEVICE, "%s: %s",    \
                        #expr, hipGetErrorString(e_));                                                 \
    } while (0)

// ---- 4x4 float matrices with Enoki's product order (column-wise fmadd) ---------------------
struct Mat4 {
    float m[4][4];
    static Mat4 identity() { Mat4 r{}; for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f; return r; }
    Mat4 operator*(const Mat4 &b) const {
        Mat4 c{};
        for (int r = 0; r < 4; ++r)
            for (int j = 0; j < 4; ++j) {
                float acc = m[r][0] * b.m[0][j];
                for (int i = 1; i < 4; ++i) acc = std::fma(m[r][i], b.m[i][j], acc);
                c.m[r][j] = acc;
            }
        return c;
    }
    Mat4 transposed() const { Mat4 c{}; for (int r = 0; r < 4; ++r) for (int j = 0; j < 4; ++j) c.m[r][j] = m[j][r]; return c; }
};
struct Xform {   // Transform: matrix + inverse transpose (include/mitsuba/core/transform.h)
    Mat4 matrix, inv_t;
    Xform operator*(const Xform &o) const { return Xform{ matrix * o.matrix, inv_t * o.inv_t }; }
    static Xform scale(float x, float y, float z) {
        Xform r{ Mat4::identity(), Mat4::identity() };
        r.matrix.m[0][0] = x; r.matrix.m[1][1] = y; r.matrix.m[2][2] = z;
        r.inv_t.m[0][0] = 1.0f / x; r.inv_t.m[1][1] = 1.0f / y; r.inv_t.m[2][2] = 1.0f / z;
        return r;
    }
    static Xform translate(float x, float y, float z) {
        Xform r{ Mat4::identity(), Mat4::identity() };
        r.matrix.m[0][3] = x; r.matrix.m[1][3] = y; r.matrix.m[2][3] = z;
        Mat4 inv = Mat4::identity(); inv.m[0][3] = -x; inv.m[1][3] = -y; inv.m[2][3] = -z;
        r.inv_t = inv.transposed();
        return r;
    }
    static Xform perspective(float fov, float near_, float far_) {   // transform.h:203-220
        float recip = 1.0f / (far_ - near_);
        float tan_ = std::tan((fov * 0.5f) * (3.14159265358979323846f / 180.0f)), cot = 1.0f / tan_;
        Mat4 t{}; t.m[0][0] = cot; t.m[1][1] = cot; t.m[2][2] = far_ * recip; t.m[2][3] = -near_ * far_ * recip; t.m[3][2] = 1.0f;
        Mat4 it{}; it.m[0][0] = tan_; it.m[1][1] = tan_; it.m[3][3] = 1.0f / near_; it.m[2][3] = 1.0f;
        it.m[3][2] = (near_ - far_) / (far_ * near_);
        return Xform{ t, it.transposed() };
    }
};

int make_camera(const mtsamd_render_desc &d, CameraView &c) {
    if (!(d.near_clip > 0.0f)) return fail(MTSAMD_ERR_INVALID, "The 'near_clip' parameter must be greater than zero!");
    if (!(d.near_clip < d.far_clip)) return fail(MTSAMD_ERR_INVALID, "The 'near_clip' parameter must be smaller than 'far_clip'.");
    if (!(d.fov_x_deg > 0.0f && d.fov_x_deg < 180.0f))
        return fail(MTSAMD_ERR_INVALID, "The horizontal field of view must be in the range [0, 180]!");
    float fw = (float) d.film_width, fh = (float) d.film_height;
    float rsx = (float) d.crop_width / fw, rsy = (float) d.crop_height / fh;
    float rox = (float) d.crop_x / fw, roy = (float) d.crop_y / fh;
    float aspect = fw / fh;
    Xform c2s = Xform::scale(1.0f / rsx, 1.0f / rsy, 1.0f) * Xform::translate(-rox, -roy, 0.0f) *
                Xform::scale(-0.5f, -0.5f * aspect, 1.0f) * Xform::translate(-1.0f, -1.0f / aspect, 0.0f) *
                Xform::perspective(d.fov_x_deg, d.near_clip, d.far_clip);
    Mat4 s2c = c2s.inv_t.transposed();              // Transform::inverse()
    for (int r = 0; r < 4; ++r) for (int j = 0; j < 4; ++j) c.s2c[4 * r + j] = s2c.m[r][j];
    std::memcpy(c.c2w, d.to_world, sizeof(float) * 16);
    c.near_clip = d.near_clip; c.far_clip = d.far_clip;
    if (d.aperture_radius < 0.0f) return fail(MTSAMD_ERR_INVALID, "The 'aperture_radius' parameter must not be negative");
    c.aperture_radius = d.aperture_radius; c.focus_distance = d.focus_distance;
    if (d.aperture_radius > 0.0f && !(d.focus_distance > 0.0f)) return fail(MTSAMD_ERR_INVALID, "thinlens: 'focus_distance' must be positive");
    return 0;
}

// B-spline family of mitchell.cpp:41-56 / catmullrom.cpp:29-43 (B = 0, C = 0.5)
static float cubic_filter(float x, float B, float C) {
    x = std::fabs(x);
    const float x2 = x * x, x3 = x2 * x;
    const float result = (1.0f / 6.0f) * (x < 1.0f
        ? (12.0f - 9.0f * B - 6.0f * C) * x3 + (-18.0f + 12.0f * B + 6.0f * C) * x2 + (6.0f - 2.0f * B)
        : (-B - 6.0f * C) * x3 + (6.0f * B + 30.0f * C) * x2 + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C));
    return x < 2.0f ? result : 0.0f;
}

// param / param2: gaussian stddev | box radius | mitchell B, C | lanczos lobes (tent and catmullrom take none)
int make_filter(int32_t kind, float param, float param2, int32_t analytic, FilterView &f) {
    std::memset(&f, 0, sizeof(f));
    f.kind = kind; f.analytic = analytic;
    if (kind == MTSAMD_RFILTER_GAUSSIAN) {
        if (!(param > 0.0f)) return fail(MTSAMD_ERR_INVALID, "gaussian rfilter: stddev must be positive");
        f.radius = 4 * param;
        f.alpha = -1.0f / (2.0f * param * param);
        f.bias = lm_exp(f.alpha * (f.radius * f.radius));
    } else if (kind == MTSAMD_RFILTER_BOX) {
        if (!(param > 0.0f)) return fail(MTSAMD_ERR_INVALID, "box rfilter: radius must be positive");
        f.radius = param + kRayEpsilon;
    } else if (kind == MTSAMD_RFILTER_TENT) {
        f.radius = 1.0f; f.alpha = 1.0f / f.radius;                    // alpha = m_inv_radius (tent.cpp:28-30)
    } else if (kind == MTSAMD_RFILTER_CATMULLROM) {
        f.radius = 2.0f;
    } else if (kind == MTSAMD_RFILTER_MITCHELL) {
        f.radius = 2.0f; f.alpha = param; f.bias = param2;            // alpha = B, bias = C
    } else if (kind == MTSAMD_RFILTER_LANCZOS) {
        if (!(param >= 1.0f) || param > 16.0f) return fail(MTSAMD_ERR_INVALID, "lanczos rfilter: 'lobes' must be in [1, 16]");
        f.radius = (float) (int) param;
    } else {
        return fail(MTSAMD_ERR_UNSUPPORTED, "unsupported reconstruction filter %d (gaussian, box, tent, catmullrom, mitchell, lanczos)", kind);
    }
    auto eval = [&](float x) -> float {
        switch (kind) {
        case MTSAMD_RFILTER_GAUSSIAN: return std::max(0.0f, lm_exp(f.alpha * (x * x)) - f.bias);
        case MTSAMD_RFILTER_TENT: return std::max(0.0f, 1.0f - std::fabs(x * f.alpha));
        case MTSAMD_RFILTER_CATMULLROM: return cubic_filter(x, 0.0f, 0.5f);
        case MTSAMD_RFILTER_MITCHELL: return cubic_filter(x, f.alpha, f.bias);
        case MTSAMD_RFILTER_LANCZOS: {
            x = std::fabs(x);
            const float x1 = kPi * x, x2 = x1 / f.radius, result = (lm_sin(x1) * lm_sin(x2)) / (x1 * x2);
            return x < kEpsilon ? 1.0f : (x > f.radius ? 0.0f : result);
        }
        default: return std::fabs(x) <= f.radius ? 1.0f : 0.0f;
        }
    };
    for (int i = 0; i < 31; ++i) f.table[i] = eval((f.radius * (float) i) / 31.0f);
    f.table[31] = 0.0f;
    f.scale_factor = 31.0f / f.radius;
    f.border = (int) std::ceil(f.radius - 0.5f - 2.0f * kRayEpsilon);
    f.taps = (int) std::ceil((f.radius - 2.0f * kRayEpsilon) * 2.0f);
    return 0;
}

template <typename T> int upload(T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void **) dst, bytes));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// Every workspace buffer is zeroed when it is allocated: a kernel that reads a slot before the first write of a render (a count, a
// cursor, the spill area, a pool slot beyond a stale count) then reads zeros on every run, not whatever the allocator handed out
// (round 2's fault came from exactly such a read; tests/conftest.py still dirties device memory before every GPU session).
// Out of device memory is reported as MTSAMD_ERR_NOMEM so that callers can retry with a smaller pass.
static int ws_alloc(void **p, size_t bytes) {
    *p = nullptr;
    bytes = std::max<size_t>(bytes, 4);
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) { (void) hipGetLastError(); *p = nullptr; return fail(MTSAMD_ERR_NOMEM, "out of device memory (%zu bytes of workspace)", bytes); }
    HIP_TRY(e);
    HIP_TRY(hipMemset(*p, 0, bytes));
    return 0;
}

struct Workspace {
    uint32_t n_waves = 0, seg_cap = 0;
    uint64_t pass_cap = 0;
    bool spectral = false, split = false;
    PoolView pool[2] = {};
    uint32_t *count[2] = { nullptr, nullptr };
    uint64_t *cursor = nullptr, *cursor_end = nullptr, *wave_stats = nullptr;
    uint32_t *count_shadow = nullptr;
    float4 *out_rgba = nullptr; float2 *out_pos = nullptr;
    float4 *out_rgba2 = nullptr; float2 *out_pos2 = nullptr; uint64_t pass_cap2 = 0;      // second sample stream: pass k + 1 is traced while pass k is splatted
    hipStream_t film_stream = nullptr; hipEvent_t film_done[2] = {};
    float *moment_film = nullptr; uint64_t moment_pixels = 0;     // moment integrator: two scratch 5-channel films
    float *film_partials = nullptr; size_t film_partial_floats = 0;      // tiled film splat: one scratch tile per 16x16 source tile of a pass
    uint32_t *trace_spill = nullptr; size_t trace_spill_words = 0;      // k_trace: deep stack entries
    uint32_t *h_counts = nullptr;        // pinned, 4 * n_waves
    uint64_t *h_cursor = nullptr;        // pinned, 2 * n_waves
    uint64_t *h_cursor_rb = nullptr;     // pinned, 2 slots x n_waves: cursors read back with the counts (pool-drain gathering)
    hipEvent_t ev[4] = {};               // 0 / 1: count read-back checkpoints, 2: k_shade done, 3: k_trace<any> done
    hipStream_t stream2 = nullptr;       // split pipeline: k_trace<any> of one iteration overlaps k_trace<closest> of the next
    hipStream_t part_stream[3] = {}; hipEvent_t part_ev[3] = {};      // flat scenes: further parts of the scheduling waves
    // hierarchy scenes: launch chain k runs k_trace<closest> + k_shade on chain_main[k] (chain 0: the job's stream) and k_trace<any> on
    // chain_any[k]; chain_ev[2k] = its k_shade is done, chain_ev[2k + 1] = its k_trace<any> is done
    hipStream_t chain_main[kMaxChains] = {}, chain_any[kMaxChains] = {}; hipEvent_t chain_ev[2 * kMaxChains] = {};
    hipEvent_t tev[3] = {};              // timing: bounce loop begin / end, film end
    bool have_events = false;
    std::vector<hipEvent_t> prof_ev;     // desc->profile: begin / end events of the split pipeline's launches, reused pass after pass
    std::vector<hipEvent_t> film_ev;     // begin / end events of the film splats of a render

    void release() {
        for (int k = 0; k < 2; ++k) {
            (void) hipFree(pool[k].ray_o); (void) hipFree(pool[k].ray_d); (void) hipFree(pool[k].thr); (void) hipFree(pool[k].res);
            (void) hipFree(pool[k].rng); (void) hipFree(pool[k].misc); (void) hipFree(count[k]);
            (void) hipFree(pool[k].xi); (void) hipFree(pool[k].aux);
            (void) hipFree(pool[k].hit); (void) hipFree(pool[k].sh_o); (void) hipFree(pool[k].sh_d); (void) hipFree(pool[k].nee); (void) hipFree(pool[k].sh_slot);
            pool[k] = PoolView{}; count[k] = nullptr;
        }
        (void) hipFree(cursor); (void) hipFree(cursor_end); (void) hipFree(wave_stats);
        (void) hipFree(count_shadow); count_shadow = nullptr;
        (void) hipFree(trace_spill); trace_spill = nullptr; trace_spill_words = 0;
        (void) hipFree(moment_film); moment_film = nullptr; moment_pixels = 0;
        (void) hipFree(film_partials); film_partials = nullptr; film_partial_floats = 0;
        (void) hipFree(out_rgba); (void) hipFree(out_pos);
        (void) hipFree(out_rgba2); (void) hipFree(out_pos2); out_rgba2 = nullptr; out_pos2 = nullptr; pass_cap2 = 0;
        if (film_stream) (void) hipStreamDestroy(film_stream);
        film_stream = nullptr;
        for (auto &e : film_done) { if (e) (void) hipEventDestroy(e); e = nullptr; }
        cursor = cursor_end = wave_stats = nullptr; out_rgba = nullptr; out_pos = nullptr;
        if (h_counts) (void) hipHostFree(h_counts);
        if (h_cursor) (void) hipHostFree(h_cursor);
        if (h_cursor_rb) (void) hipHostFree(h_cursor_rb);
        h_cursor_rb = nullptr;
        h_counts = nullptr; h_cursor = nullptr;
        if (stream2) (void) hipStreamDestroy(stream2);
        stream2 = nullptr;
        for (auto &ps : part_stream) { if (ps) (void) hipStreamDestroy(ps); ps = nullptr; }
        for (auto &ps : chain_main) { if (ps) (void) hipStreamDestroy(ps); ps = nullptr; }
        for (auto &ps : chain_any) { if (ps) (void) hipStreamDestroy(ps); ps = nullptr; }
        for (auto &pe : chain_ev) { if (pe) (void) hipEventDestroy(pe); pe = nullptr; }
        for (auto &pe : part_ev) { if (pe) (void) hipEventDestroy(pe); pe = nullptr; }
        if (have_events) for (auto &e : ev) (void) hipEventDestroy(e);
        have_events = false;
        for (auto &e : prof_ev) (void) hipEventDestroy(e);
        prof_ev.clear();
        for (auto &e : film_ev) (void) hipEventDestroy(e);
        film_ev.clear();
        n_waves = seg_cap = 0; pass_cap = 0;
    }
};

} // namespace

// Model parameters of a BSDF -> device record (see DevBsdf in device_bsdf.h), with the constants the reference's
// constructors derive (plastic.cpp:162-176; fresnel_diffuse_reflectance: fresnel.h:331-358).
static float fresnel_diffuse_reflectance(float eta) {
    if (eta < 1.0f) return -1.4399f * (eta * eta) + 0.7099f * eta + 0.6681f + 0.0636f / eta;
    const float i1 = 1.0f / eta, i2 = i1 * i1, i3 = i2 * i1, i4 = i3 * i1, i5 = i4 * i1;
    return 0.919317f - 3.4793f * i1 + 6.75335f * i2 - 7.80989f * i3 + 4.98554f * i4 - 1.36881f * i5;
}
static void fill_bsdf_model(const mtsamd_bsdf_desc &bd, DevBsdf &d) {
    d.flags = (bd.twosided ? kBsdfTwoSided : 0u) | (bd.distribution == 1 ? kBsdfGGX : 0u) | (bd.sample_visible ? kBsdfSampleVisible : 0u) |
              (bd.nonlinear ? kBsdfNonlinear : 0u);
    if (bd.type == MTSAMD_BSDF_DIFFUSE) d.flags &= kBsdfTwoSided;
    d.flags |= ((bd.uniform_mask & 1) ? kBsdfUniformRefl : 0u) | ((bd.uniform_mask & 2) ? kBsdfUniformSpec : 0u) |
               ((bd.uniform_mask & 4) ? kBsdfUniformTrans : 0u);
    d.sr = bd.specular_reflectance[0]; d.sg = bd.specular_reflectance[1]; d.sb = bd.specular_reflectance[2];
    d.alpha_u = bd.alpha_u; d.alpha_v = bd.alpha_v;
    if (bd.type == MTSAMD_BSDF_CONDUCTOR || bd.type == MTSAMD_BSDF_ROUGHCONDUCTOR) {
        d.er = bd.eta[0]; d.eg = bd.eta[1]; d.eb = bd.eta[2];
        d.kr = bd.k[0]; d.kg = bd.k[1]; d.kb = bd.k[2];
    } else if (bd.type == MTSAMD_BSDF_DIELECTRIC || bd.type == MTSAMD_BSDF_ROUGHDIELECTRIC || bd.type == MTSAMD_BSDF_THINDIELECTRIC) {
        d.er = bd.int_ior / bd.ext_ior;
        d.kr = bd.specular_transmittance[0]; d.kg = bd.specular_transmittance[1]; d.kb = bd.specular_transmittance[2];
    } else if (bd.type == MTSAMD_BSDF_PLASTIC || bd.type == MTSAMD_BSDF_ROUGHPLASTIC) {
        const float eta = bd.int_ior / bd.ext_ior;
        d.er = eta; d.eg = 1.0f / (eta * eta);
        d.eb = bd.type == MTSAMD_BSDF_PLASTIC ? fresnel_diffuse_reflectance(1.0f / eta) : 0.0f;     // roughplastic: set by the table kernel
        const float d_mean = (bd.reflectance[0] + bd.reflectance[1] + bd.reflectance[2]) * (1.0f / 3.0f);
        const float s_mean = (bd.specular_reflectance[0] + bd.specular_reflectance[1] + bd.specular_reflectance[2]) * (1.0f / 3.0f);
        d.kr = s_mean / (d_mean + s_mean);
    }
}

// quad::gauss_legendre (src/libcore/quad.cpp:7-66, legendre_pd: math.h:127-154): nodes / weights on [-1, 1]
static void gauss_legendre(int n, float *nodes, float *weights) {
    auto legendre_pd = [](int l, double x, double &lv, double &dv) {
        double l_cur = 0.0, d_cur = 0.0;
        if (l > 1) {
            double l_p_pred = 1.0, l_pred = x, d_p_pred = 0.0, d_pred = 1.0, k0 = 3.0, k1 = 2.0, k2 = 1.0;
            for (int ki = 2; ki <= l; ++ki) {
                l_cur = (k0 * x * l_pred - k2 * l_p_pred) / k1;
                d_cur = d_p_pred + k0 * l_pred;
                l_p_pred = l_pred; l_pred = l_cur; d_p_pred = d_pred; d_pred = d_cur;
                k2 = k1; k0 += 2.0; k1 += 1.0;
            }
        } else if (l == 0) { l_cur = 1.0; d_cur = 0.0; }
        else { l_cur = x; d_cur = 1.0; }
        lv = l_cur; dv = d_cur;
    };
    n--;
    if (n == 0) { nodes[0] = 0.0f; weights[0] = 2.0f; }
    else if (n == 1) { nodes[0] = (float) -std::sqrt(1.0 / 3.0); nodes[1] = -nodes[0]; weights[0] = weights[1] = 1.0f; }
    const int m = (n + 1) / 2;
    for (int i = 0; i < m; ++i) {
        double x = -std::cos((double) (2 * i + 1) / (double) (2 * n + 2) * 3.14159265358979323846);
        double l, d;
        for (int it = 0; it < 20; ++it) {
            legendre_pd(n + 1, x, l, d);
            const double step = l / d;
            x -= step;
            if (std::fabs(step) <= 4 * std::fabs(x) * 2.220446049250313e-16) break;
        }
        legendre_pd(n + 1, x, l, d);
        weights[i] = weights[n - i] = (float) (2 / ((1 - x * x) * (d * d)));
        nodes[i] = (float) x; nodes[n - i] = (float) -x;
    }
    if ((n % 2) == 0) {
        double l, d;
        legendre_pd(n + 1, 0.0, l, d);
        weights[n / 2] = (float) (2.0 / (d * d));
        nodes[n / 2] = 0.0f;
    }
}

struct mtsamd_scene {
    int32_t environment = -1;        // index of the `constant` emitter
    bool general_bsdfs = false;      // any BSDF other than one-sided `diffuse`: the kernels with the BSDF switch are used
    bool nested_bsdfs = false;       // blendbsdf / mask: the fused schedule (k_bounce*, k_direct) is the one that carries the nesting code
    bool non_diffuse_bsdfs = false;  // any BSDF other than `diffuse` (one- or two-sided): what the adjoint path replay cannot differentiate
    bool delta_emitters = false;     // point / spot / directional emitters: handled by the same general kernels
    int device = 0;
    int cu_count = 256;
    BvhOutput bvh;
    uint32_t n_prims = 0, n_shapes = 0;
    std::vector<DevBsdf> bsdfs;
    std::vector<DevEmitter> emitters;
    float4 *d_nodes = nullptr, *d_tris = nullptr;
    uint4 *d_qnodes = nullptr, *d_wnodes = nullptr;
    StackEntry *d_walk_spill = nullptr;
    float *d_tri_pos = nullptr, *d_tri_nrm = nullptr, *d_tri_uv = nullptr;
    uint32_t *d_prim_shape = nullptr;
    DevShape *d_shapes = nullptr; DevBsdf *d_bsdfs = nullptr; DevEmitter *d_emitters = nullptr;
    float *d_area_pmf = nullptr, *d_area_cdf = nullptr;
    float *d_rough_tables = nullptr;     // roughplastic: 64 floats per BSDF that needs them
    float *d_env_texels = nullptr, *d_env_warp = nullptr; DevEnvmap *d_envmap = nullptr;      // envmap emitter
    int32_t env_w = 0, env_h = 0;
    float4 *d_flat = nullptr, *d_pairs = nullptr;
    std::vector<DevTexture> textures;       // device data pointers, owned
    std::vector<float> spec_mean;           // per BSDF: mean of specular_reflectance
    std::vector<float> diff_mean;           // spectral variant, per BSDF: Texture::mean() of a constant reflectance
    Rgb2Spec rgb2spec;                      // spectral variant: the upsampling model, kept for parameter updates
    DevTexture *d_textures = nullptr;
    SceneView view{};
    bool spectral = false;
    Workspace ws;
    std::atomic<int> cancel{ 0 };
};

extern "C" {

int mtsamd_abi_version(void) { return MTSAMD_ABI_VERSION; }
// MTS_EXPORT_PLUGIN (include/mitsuba/core/class.h:205-211): what PluginManager reads after dlopen (src/libcore/plugin.cpp:19-31)
const char *plugin_name(void) { return "path_amd"; }
const char *plugin_descr(void) { return "Wavefront path tracer for AMD MI355X (gfx950)"; }
const char *mtsamd_last_error(void) { return g_last_error.c_str(); }

int mtsamd_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(MTSAMD_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

void mtsamd_scene_destroy(mtsamd_scene *s) {
    if (!s) return;
    (void) hipSetDevice(s->device);
    s->ws.release();
    (void) hipFree(s->d_nodes); (void) hipFree(s->d_qnodes); (void) hipFree(s->d_wnodes); (void) hipFree(s->d_walk_spill); (void) hipFree(s->d_tris); (void) hipFree(s->d_tri_pos); (void) hipFree(s->d_tri_nrm); (void) hipFree(s->d_tri_uv);
    (void) hipFree(s->d_prim_shape); (void) hipFree(s->d_shapes); (void) hipFree(s->d_bsdfs); (void) hipFree(s->d_emitters);
    (void) hipFree(s->d_area_pmf); (void) hipFree(s->d_area_cdf); (void) hipFree(s->d_rough_tables);
    (void) hipFree(s->d_env_texels); (void) hipFree(s->d_env_warp); (void) hipFree(s->d_envmap); (void) hipFree(s->d_flat); (void) hipFree(s->d_pairs);
    for (auto &t : s->textures) (void) hipFree((void *) t.data);
    (void) hipFree(s->d_textures);
    delete s;
}

int mtsamd_scene_create(const mtsamd_scene_desc *desc, int device, mtsamd_scene **out) {
    if (!desc || !out) return fail(MTSAMD_ERR_INVALID, "mtsamd_scene_create: null argument");
    *out = nullptr;
    // an empty scene is valid (it renders to zeros: scenes.py:262-267 of the reference's integrator tests)
    if (desc->mesh_count > 0 && !desc->meshes) return fail(MTSAMD_ERR_INVALID, "scene has no shapes");
    if (desc->mesh_count > 0 && (desc->bsdf_count == 0 || !desc->bsdfs)) return fail(MTSAMD_ERR_INVALID, "scene has no BSDFs");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(MTSAMD_ERR_INVALID, "invalid device index %d (have %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));

    // ---- validate + flatten the meshes into one global primitive list ----------------------
    uint64_t total = 0;
    bool any_nrm = false, any_uv = false;
    std::vector<int32_t> emitter_shape(desc->emitter_count, -1);
    for (uint32_t i = 0; i < desc->mesh_count; ++i) {
        const mtsamd_mesh_desc &m = desc->meshes[i];
        if (!m.positions || !m.faces || m.face_count == 0 || m.vertex_count == 0)
            return fail(MTSAMD_ERR_INVALID, "mesh %u: empty mesh", i);
        if (m.bsdf < 0 || (uint32_t) m.bsdf >= desc->bsdf_count) return fail(MTSAMD_ERR_INVALID, "mesh %u: invalid bsdf index %d", i, m.bsdf);
        if (m.emitter >= (int32_t) desc->emitter_count) return fail(MTSAMD_ERR_INVALID, "mesh %u: invalid emitter index %d", i, m.emitter);
        if (m.emitter >= 0) {
            // "An area emitter can be only be attached to a single shape." (area.cpp:64-66)
            if (emitter_shape[m.emitter] >= 0) return fail(MTSAMD_ERR_INVALID, "An area emitter can be only be attached to a single shape.");
            emitter_shape[m.emitter] = (int32_t) i;
        }
        for (uint64_t k = 0; k < 3ull * m.face_count; ++k)
            if (m.faces[k] >= m.vertex_count) return fail(MTSAMD_ERR_INVALID, "mesh %u: face index out of range", i);
        total += m.face_count;
        any_nrm |= m.normals != nullptr; any_uv |= m.texcoords != nullptr;
    }
    int32_t environment = -1;
    for (uint32_t e = 0; e < desc->emitter_count; ++e) {
        const int32_t et = desc->emitters[e].type;
        if (et == MTSAMD_EMITTER_CONSTANT || et == MTSAMD_EMITTER_ENVMAP) {
            if (emitter_shape[e] >= 0) return fail(MTSAMD_ERR_INVALID, "emitter %u: an environment emitter cannot be attached to a shape", e);
            if (environment >= 0) return fail(MTSAMD_ERR_INVALID, "Only one environment emitter can be specified per scene.");      // scene.cpp:45-46
            if (et == MTSAMD_EMITTER_ENVMAP && (!desc->emitters[e].envmap_data || desc->emitters[e].envmap_width < 2 || desc->emitters[e].envmap_height < 2))
                return fail(MTSAMD_ERR_INVALID, "emitter %u: the environment map must be at least 2x2 pixels in size", e);
            environment = (int32_t) e;
            continue;
        }
        if (et == MTSAMD_EMITTER_POINT || et == MTSAMD_EMITTER_SPOT || et == MTSAMD_EMITTER_DIRECTIONAL) {
            if (emitter_shape[e] >= 0) return fail(MTSAMD_ERR_INVALID, "emitter %u: a point / spot / directional emitter cannot be attached to a shape", e);
            if (et == MTSAMD_EMITTER_SPOT && !(desc->emitters[e].cutoff_angle >= desc->emitters[e].beam_width))
                return fail(MTSAMD_ERR_INVALID, "emitter %u: spot: cutoff_angle must not be smaller than beam_width", e);      // spot.cpp:89
            continue;
        }
        if (et != MTSAMD_EMITTER_AREA) return fail(MTSAMD_ERR_UNSUPPORTED, "emitter %u: unknown emitter type %d", e, et);
        if (emitter_shape[e] < 0) return fail(MTSAMD_ERR_INVALID, "emitter %u is not attached to a shape", e);
    }
    for (uint32_t b = 0; b < desc->bsdf_count; ++b) {
        const mtsamd_bsdf_desc &bd = desc->bsdfs[b];
        if (bd.type < MTSAMD_BSDF_DIFFUSE || bd.type > MTSAMD_BSDF_MASK) return fail(MTSAMD_ERR_UNSUPPORTED, "bsdf %u: unknown BSDF type %d", b, bd.type);
        if (desc->spectral && (bd.type == MTSAMD_BSDF_CONDUCTOR || bd.type == MTSAMD_BSDF_ROUGHCONDUCTOR) &&
            (bd.eta[0] != bd.eta[1] || bd.eta[0] != bd.eta[2] || bd.k[0] != bd.k[1] || bd.k[0] != bd.k[2]))
            return fail(MTSAMD_ERR_UNSUPPORTED, "bsdf %u: the spectral variant needs uniform (constant) eta and k spectra", b);
        if (bd.texture >= 0 && bd.type != MTSAMD_BSDF_DIFFUSE && bd.type != MTSAMD_BSDF_PLASTIC && bd.type != MTSAMD_BSDF_ROUGHPLASTIC &&
            bd.type != MTSAMD_BSDF_BLEND && bd.type != MTSAMD_BSDF_MASK)
            return fail(MTSAMD_ERR_UNSUPPORTED, "bsdf %u: textures are implemented for diffuse.reflectance and (rough)plastic.diffuse_reflectance only", b);
        if (bd.type == MTSAMD_BSDF_ROUGHPLASTIC && (bd.int_ior == bd.ext_ior || bd.alpha_u != bd.alpha_v))
            return fail(MTSAMD_ERR_INVALID, bd.int_ior == bd.ext_ior ? "The interior and exterior indices of refraction must be positive and differ!"
                                                                      : "The 'roughplastic' plugin currently does not support anisotropic microfacet distributions!");
        if (bd.type == MTSAMD_BSDF_ROUGHDIELECTRIC && (bd.int_ior < 0.0f || bd.ext_ior < 0.0f || bd.int_ior == bd.ext_ior))
            return fail(MTSAMD_ERR_INVALID, "The interior and exterior indices of refraction must be positive and differ!");      // roughdielectric.cpp:153-155
        if ((bd.type == MTSAMD_BSDF_DIELECTRIC || bd.type == MTSAMD_BSDF_PLASTIC || bd.type == MTSAMD_BSDF_ROUGHPLASTIC || bd.type == MTSAMD_BSDF_ROUGHDIELECTRIC ||
             bd.type == MTSAMD_BSDF_THINDIELECTRIC) &&
            (bd.int_ior < 0.0f || bd.ext_ior < 0.0f || bd.ext_ior == 0.0f))
            return fail(MTSAMD_ERR_INVALID, "The interior and exterior indices of refraction must be positive!");      // dielectric.cpp:183-185
        if ((bd.type == MTSAMD_BSDF_DIELECTRIC || bd.type == MTSAMD_BSDF_ROUGHDIELECTRIC || bd.type == MTSAMD_BSDF_THINDIELECTRIC) && bd.twosided)
            return fail(MTSAMD_ERR_INVALID, "Only materials without a transmission component can be nested!");          // twosided.cpp:90-91
        if (desc->bsdfs[b].texture >= (int32_t) desc->texture_count) return fail(MTSAMD_ERR_INVALID, "bsdf %u: invalid texture index %d", b, desc->bsdfs[b].texture);
    }
    for (uint32_t t = 0; t < desc->texture_count; ++t) {
        if (!desc->textures) return fail(MTSAMD_ERR_INVALID, "null texture table");
        const mtsamd_texture_desc &td = desc->textures[t];
        if (td.kind != 0 && td.kind != 1) return fail(MTSAMD_ERR_UNSUPPORTED, "texture %u: unknown texture kind %d", t, td.kind);
        if (td.kind == 0 && (!td.data || td.width < 2 || td.height < 2))
            return fail(MTSAMD_ERR_INVALID, "texture %u: image must be at least 2x2 pixels in size", t);      // bitmap.cpp:101-107
    }
    if (total >= (1ull << 27)) return fail(MTSAMD_ERR_UNSUPPORTED, "too many primitives (%llu)", (unsigned long long) total);

    mtsamd_scene *s = new mtsamd_scene();
    s->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) s->cu_count = prop.multiProcessorCount;
    s->n_prims = (uint32_t) total; s->n_shapes = desc->mesh_count;

    std::vector<float> tri_pos(9 * total), tri_nrm(any_nrm ? 9 * total : 0), tri_uv(any_uv ? 6 * total : 0);
    std::vector<uint32_t> prim_shape(total);
    std::vector<DevShape> shapes(desc->mesh_count);
    std::vector<float> area_pmf(total, 0.0f), area_cdf(total, 0.0f);
    s->emitters.resize(desc->emitter_count);
    s->environment = environment;
    uint32_t off = 0;
    for (uint32_t i = 0; i < desc->mesh_count; ++i) {
        const mtsamd_mesh_desc &m = desc->meshes[i];
        shapes[i].bsdf = m.bsdf; shapes[i].emitter = m.emitter; shapes[i].first_prim = off;
        shapes[i].flags = (m.normals ? kShapeHasNormals : 0u) | (m.texcoords ? kShapeHasUV : 0u);
        for (uint32_t f = 0; f < m.face_count; ++f) {
            uint32_t gp = off + f;
            prim_shape[gp] = i;
            for (int j = 0; j < 3; ++j) {
                uint32_t vi = m.faces[3 * f + j];
                for (int k = 0; k < 3; ++k) tri_pos[9 * (size_t) gp + 3 * j + k] = m.positions[3 * (size_t) vi + k];
                if (m.normals) for (int k = 0; k < 3; ++k) tri_nrm[9 * (size_t) gp + 3 * j + k] = m.normals[3 * (size_t) vi + k];
                if (m.texcoords) for (int k = 0; k < 2; ++k) tri_uv[6 * (size_t) gp + 2 * j + k] = m.texcoords[2 * (size_t) vi + k];
            }
        }
        if (m.emitter >= 0) {
            // Mesh::area_distr_build (mesh.cpp:284-307) + DiscreteDistribution::update (distr_1d.h:49-88)
            double sum = 0.0; uint32_t lo = 0xffffffffu, hi = 0xffffffffu;
            for (uint32_t f = 0; f < m.face_count; ++f) {
                const float *tp = &tri_pos[9 * (size_t) (off + f)];
                float e1[3] = { tp[3] - tp[0], tp[4] - tp[1], tp[5] - tp[2] }, e2[3] = { tp[6] - tp[0], tp[7] - tp[1], tp[8] - tp[2] };
                float cx = std::fma(e1[1], e2[2], -(e1[2] * e2[1])), cy = std::fma(e1[2], e2[0], -(e1[0] * e2[2])),
                      cz = std::fma(e1[0], e2[1], -(e1[1] * e2[0]));
                float area = 0.5f * std::sqrt(std::fma(cz, cz, std::fma(cy, cy, cx * cx)));
                area_pmf[off + f] = area;
                sum += (double) area;
                area_cdf[off + f] = (float) sum;
                if (area > 0.0f) { if (lo == 0xffffffffu) lo = f; hi = f; }
            }
            if (lo == 0xffffffffu) { delete s; return fail(MTSAMD_ERR_INVALID, "DiscreteDistribution: no probability mass found!"); }
            DevEmitter &e = s->emitters[m.emitter];
            std::memset(&e, 0, sizeof(e));
            const mtsamd_emitter_desc &ed = desc->emitters[m.emitter];
            e.r = ed.radiance[0]; e.g = ed.radiance[1]; e.b = ed.radiance[2];
            e.shape = i; e.first_prim = off; e.n_prims = m.face_count;
            e.area_sum = (float) sum; e.area_norm = (float) (1.0 / sum);
            e.valid_lo = lo; e.valid_hi = hi;
        }
        off += m.face_count;
    }
    // delta emitters (point.cpp:52-65, spot.cpp:68-91, directional.cpp:43-63)
    for (uint32_t ei = 0; ei < desc->emitter_count; ++ei) {
        const mtsamd_emitter_desc &ed = desc->emitters[ei];
        if (ed.type != MTSAMD_EMITTER_POINT && ed.type != MTSAMD_EMITTER_SPOT && ed.type != MTSAMD_EMITTER_DIRECTIONAL) continue;
        DevEmitter &e = s->emitters[ei];
        std::memset(&e, 0, sizeof(e));
        s->delta_emitters = true;
        e.r = ed.radiance[0]; e.g = ed.radiance[1]; e.b = ed.radiance[2];
        e.shape = 0xffffffffu; e.pad0 = (uint32_t) ed.type;
        const float *m = ed.to_world;
        e.cx = m[3]; e.cy = m[7]; e.cz = m[11];
        if (ed.type == MTSAMD_EMITTER_DIRECTIONAL) {             // d = to_world * (0, 0, 1)
            e.aux[0] = m[2]; e.aux[1] = m[6]; e.aux[2] = m[10];
        } else if (ed.type == MTSAMD_EMITTER_SPOT) {
            const double a = m[0], b = m[1], c = m[2], d2 = m[4], e2 = m[5], f = m[6], g = m[8], h2 = m[9], i2 = m[10];
            const double det = a * (e2 * i2 - f * h2) - b * (d2 * i2 - f * g) + c * (d2 * h2 - e2 * g);
            if (det == 0.0) { delete s; return fail(MTSAMD_ERR_INVALID, "emitter %u: singular to_world transformation", ei); }
            const double inv[9] = { (e2 * i2 - f * h2) / det, (c * h2 - b * i2) / det, (b * f - c * e2) / det,
                                    (f * g - d2 * i2) / det, (a * i2 - c * g) / det, (c * d2 - a * f) / det,
                                    (d2 * h2 - e2 * g) / det, (b * g - a * h2) / det, (a * e2 - b * d2) / det };
            for (int k = 0; k < 9; ++k) e.aux[k] = (float) inv[k];
            const float cutoff = ed.cutoff_angle * (kPi / 180.0f), beam = ed.beam_width * (kPi / 180.0f);
            e.aux[9] = cutoff; e.aux[10] = std::cos(cutoff); e.aux[11] = std::cos(beam); e.aux[12] = 1.0f / (cutoff - beam);
        }
    }
    // spectral variant: RGB -> spectrum coefficients on the host (srgb.cpp:31-41, srgb_d65.cpp:31-46)
    Rgb2Spec &model = s->rgb2spec;
    if (desc->spectral) {
        if (!desc->rgb2spec_path || !rgb2spec_load(desc->rgb2spec_path, model)) {
            delete s;
            return fail(MTSAMD_ERR_INVALID, "Could not load sRGB-to-spectrum upsampling model ('%s'); build it with mtsamd_rgb2spec_build",
                        desc->rgb2spec_path ? desc->rgb2spec_path : "(null)");
        }
        s->spectral = true;
        for (uint32_t e = 0; e < desc->emitter_count; ++e) {
            const float *c = desc->emitters[e].radiance;
            float color[3] = { c[0], c[1], c[2] };
            float scale = std::max(std::max(color[0], color[1]), color[2]) * 2.0f;
            if (scale != 0.0f) { float r = 1.0f / scale; for (float &v : color) v *= r; }
            float coeff[3];
            srgb_model_fetch(model, color, coeff);
            float d65_scale = 1.0f * scale;
            d65_scale *= 1.0f / 10568.0f;                      // d65.cpp:44-50
            DevEmitter &d = s->emitters[e];
            d.c0 = coeff[0]; d.c1 = coeff[1]; d.c2 = coeff[2]; d.d65_scale = d65_scale;
        }
    }
    s->bsdfs.resize(desc->bsdf_count);
    std::vector<float> spec_mean(desc->bsdf_count, 0.0f);          // Texture::mean() of specular_reflectance (plastic lobe weights)
    for (uint32_t b = 0; b < desc->bsdf_count; ++b) {
        DevBsdf &d = s->bsdfs[b];
        spec_mean[b] = (desc->bsdfs[b].specular_reflectance[0] + desc->bsdfs[b].specular_reflectance[1] + desc->bsdfs[b].specular_reflectance[2]) * (1.0f / 3.0f);
        std::memset(&d, 0, sizeof(d));
        d.r = desc->bsdfs[b].reflectance[0]; d.g = desc->bsdfs[b].reflectance[1]; d.b = desc->bsdfs[b].reflectance[2];
        d.type = desc->bsdfs[b].type; d.texture = desc->bsdfs[b].texture < 0 ? -1 : desc->bsdfs[b].texture;
        fill_bsdf_model(desc->bsdfs[b], d);
        if (d.type != kBsdfDiffuse || (d.flags & kBsdfTwoSided)) s->general_bsdfs = true;
        if (d.type != kBsdfDiffuse) s->non_diffuse_bsdfs = true;
        if (d.type == kBsdfBlend || d.type == kBsdfMask) {
            s->nested_bsdfs = true;
            // blendbsdf.cpp:57-79 / mask.cpp:67-91 over plain records of this table (one level of nesting)
            const mtsamd_bsdf_desc &bd = desc->bsdfs[b];
            const int n_child = d.type == kBsdfBlend ? 2 : 1;
            bool smooth = false;
            for (int k = 0; k < n_child; ++k) {
                const int32_t c = bd.nested[k];
                if (c < 0 || (uint32_t) c >= desc->bsdf_count || desc->bsdfs[c].type < MTSAMD_BSDF_DIFFUSE || desc->bsdfs[c].type > MTSAMD_BSDF_THINDIELECTRIC) {
                    delete s;
                    return fail(MTSAMD_ERR_INVALID, "bsdf %u: nested[%d] must index a plain BSDF record", b, k);
                }
                if (desc->spectral && desc->bsdfs[c].texture >= 0) {
                    delete s;
                    return fail(MTSAMD_ERR_UNSUPPORTED, "bsdf %u: textured children of a blendbsdf / mask are implemented for the RGB variant only", b);
                }
                const int ct = desc->bsdfs[c].type;
                smooth = smooth || ct == kBsdfDiffuse || ct == kBsdfRoughConductor || ct == kBsdfPlastic || ct == kBsdfRoughPlastic || ct == kBsdfRoughDielectric;
            }
            if (d.type == kBsdfMask && bd.twosided) { delete s; return fail(MTSAMD_ERR_INVALID, "Only materials without a transmission component can be nested!"); }
            if (desc->spectral && d.texture >= 0) {
                delete s;
                return fail(MTSAMD_ERR_UNSUPPORTED, "eval_1(): a bitmap / checkerboard weight is converted into spectra in the spectral variant (bitmap.cpp:218-222); use a constant");
            }
            d.nested0 = (uint32_t) bd.nested[0]; d.nested1 = (uint32_t) (n_child == 2 ? bd.nested[1] : bd.nested[0]);
            d.flags = (bd.twosided ? kBsdfTwoSided : 0u) | kBsdfUniformRefl | (smooth ? kBsdfNestSmooth : 0u) |
                      ((d.texture >= 0 && desc->textures[d.texture].kind == 0) ? kBsdfWeightLum : 0u);
        }
        if (desc->spectral) {
            // every colour-valued parameter is a `uniform` constant or an `srgb` texture: range check + coefficient fetch
            // (srgb.cpp:31-41); Texture::mean() of either kind feeds the plastic lobe-selection weight (plastic.cpp:170-175)
            const mtsamd_bsdf_desc &bd = desc->bsdfs[b];
            const float *vals[3] = { bd.reflectance, bd.specular_reflectance, bd.specular_transmittance };
            float *coeffs[3] = { &d.c0, &d.sc0, &d.tc0 };
            float means[3] = { 0.0f, 0.0f, 0.0f };
            for (int p = 0; p < 3; ++p) {
                if (d.type == kBsdfBlend || d.type == kBsdfMask) break;   // the weight is a scalar; the children are records of their own
                if (p == 0 && d.texture >= 0) continue;               // textured: coefficients per texel, mean from the texture (below)
                if (bd.uniform_mask & (1 << p)) { means[p] = vals[p][0]; continue; }
                const float *c = vals[p];
                if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] > 1 || c[1] > 1 || c[2] > 1) {
                    delete s;
                    return fail(MTSAMD_ERR_INVALID, "Invalid RGB reflectance value [%g, %g, %g], must be in the range [0, 1]!", c[0], c[1], c[2]);
                }
                float coeff[3];
                srgb_model_fetch(model, c, coeff);
                coeffs[p][0] = coeff[0]; coeffs[p][1] = coeff[1]; coeffs[p][2] = coeff[2];
                means[p] = srgb_model_mean(coeff);
            }
            if (d.type == kBsdfPlastic || d.type == kBsdfRoughPlastic) d.kr = means[1] / (means[0] + means[1]);
            spec_mean[b] = means[1];
            s->diff_mean.resize(desc->bsdf_count, 0.0f);
            s->diff_mean[b] = means[0];
        }
    }

    for (uint32_t t = 0; t < desc->texture_count; ++t) {
        const mtsamd_texture_desc &td = desc->textures[t];
        DevTexture dt{};
        dt.kind = (uint32_t) td.kind;
        dt.w = td.kind == 0 ? td.width : 0; dt.h = td.kind == 0 ? td.height : 0;
        dt.grad_offset = s->textures.empty() ? 0u : s->textures.back().grad_offset + 3u * (uint32_t) s->textures.back().w * (uint32_t) s->textures.back().h;
        bool ident = true;
        for (int k = 0; k < 6; ++k) { dt.uvm[k] = td.to_uv[k]; ident = ident && td.to_uv[k] == 0.0f; }
        if (ident) { dt.uvm[0] = 1.0f; dt.uvm[4] = 1.0f; }
        for (int k = 0; k < 3; ++k) { dt.c0[k] = td.color0[k]; dt.c1[k] = td.color1[k]; }
        if (td.kind == 1) {
            // Texture::mean(): mean of the two colours' means (checkerboard.cpp:88-90, srgb.cpp:52-57); spectral variant: `srgb`
            // spectra with the constructor's range check (srgb.cpp:34-35)
            if (desc->spectral) {
                for (int k = 0; k < 3; ++k)
                    if (td.color0[k] < 0 || td.color0[k] > 1 || td.color1[k] < 0 || td.color1[k] > 1) {
                        mtsamd_scene_destroy(s);
                        return fail(MTSAMD_ERR_INVALID, "Invalid RGB reflectance value in checkerboard texture %u, must be in the range [0, 1]!", t);
                    }
                srgb_model_fetch(model, td.color0, dt.c0);
                srgb_model_fetch(model, td.color1, dt.c1);
                dt.mean = 0.5f * (srgb_model_mean(dt.c0) + srgb_model_mean(dt.c1));
            } else {
                dt.mean = 0.5f * ((td.color0[0] + td.color0[1] + td.color0[2]) * (1.0f / 3.0f) + (td.color1[0] + td.color1[1] + td.color1[2]) * (1.0f / 3.0f));
            }
            s->textures.push_back(dt);
            continue;
        }
        size_t bytes = sizeof(float) * 3 * (size_t) td.width * td.height;
        float *ptr = nullptr;
        // Texture::mean(): mean luminance (bitmap.cpp:124-136); spectral variant: texels become model coefficients, mean of
        // srgb_model_mean (bitmap.cpp:116-123)
        const size_t n_texels = (size_t) td.width * td.height;
        std::vector<float> coeffs;
        const float *src = td.data;
        double mean = 0.0;
        if (desc->spectral) {
            coeffs.resize(3 * n_texels);
            for (size_t i = 0; i < n_texels; ++i) {
                srgb_model_fetch(model, td.data + 3 * i, coeffs.data() + 3 * i);
                mean += (double) srgb_model_mean(coeffs.data() + 3 * i);
            }
            src = coeffs.data();
        } else {
            for (size_t i = 0; i < n_texels; ++i) {
                const float *p = td.data + 3 * i;
                mean += (double) (p[0] * 0.212671f + p[1] * 0.715160f + p[2] * 0.072169f);
            }
        }
        dt.mean = (float) (mean / (double) n_texels);
        if (hipMalloc((void **) &ptr, bytes) != hipSuccess || hipMemcpy(ptr, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
            mtsamd_scene_destroy(s);
            return fail(MTSAMD_ERR_NOMEM, "texture %u: upload failed", t);
        }
        dt.data = ptr;
        s->textures.push_back(dt);
    }
    // plastic.cpp:170-175: specular sampling weight from Texture::mean() of both reflectances
    s->spec_mean = spec_mean;
    for (uint32_t b = 0; b < desc->bsdf_count; ++b) {
        DevBsdf &d = s->bsdfs[b];
        if (d.texture >= 0 && (d.type == kBsdfPlastic || d.type == kBsdfRoughPlastic)) d.kr = spec_mean[b] / (s->textures[d.texture].mean + spec_mean[b]);
    }

    if (desc->spectral) {
        float tx[95], ty[95], tz[95], td[95];
        for (int i = 0; i < 95; ++i) { tx[i] = (float) kCie_x[i]; ty[i] = (float) kCie_y[i]; tz[i] = (float) kCie_z[i]; td[i] = (float) kCie_d65[i]; }
        if (upload_spectral_tables(tx, ty, tz, td) != hipSuccess) { mtsamd_scene_destroy(s); return fail(MTSAMD_ERR_DEVICE, "spectral table upload failed"); }
    }

    // ---- accelerator -------------------------------------------------------------------------
    uint32_t max_leaf = 4;
    if (const char *e = exp_env("MTSAMD_BVH_LEAF")) max_leaf = (uint32_t) std::min(15, std::max(1, atoi(e)));      // experiment switch
    BvhOptions bopt;
    if (const char *e = exp_env("MTSAMD_BVH_BINS")) bopt.bins = atoi(e);                      // experiment switches
    if (const char *e = exp_env("MTSAMD_BVH_ICOST")) bopt.intersect_cost = atof(e);
    if (const char *e = exp_env("MTSAMD_BVH_SWEEP")) bopt.sweep_below = (uint32_t) std::max(0, atoi(e));
    if (const char *e = exp_env("MTSAMD_BVH_ORDER")) bopt.dfs_order = e[0] == 'd';
    if (s->n_prims > 0) build_bvh(tri_pos.data(), s->n_prims, max_leaf, s->bvh, &bopt);
    else { s->bvh = BvhOutput{}; s->bvh.root = s->bvh.wroot = 0x80000000u; s->bvh.wdepth = 1; }       // a leaf with no triangles (BVH2 and BVH4 root: without wroot the walks of the split pipeline started at node 0 of an empty node array)
    if (s->environment >= 0) {       // ConstantBackgroundEmitter::set_scene (constant.cpp:47-51): bounding sphere of Scene::bbox()
        DevEmitter &e = s->emitters[s->environment];
        const float sc[4] = { e.c0, e.c1, e.c2, e.d65_scale };            // spectral variant: filled above
        std::memset(&e, 0, sizeof(e));
        e.c0 = sc[0]; e.c1 = sc[1]; e.c2 = sc[2]; e.d65_scale = sc[3];
        const mtsamd_emitter_desc &ed = desc->emitters[s->environment];
        e.r = ed.radiance[0]; e.g = ed.radiance[1]; e.b = ed.radiance[2];
        e.shape = 0xffffffffu; e.pad0 = ed.type == MTSAMD_EMITTER_ENVMAP ? kEmitterEnvmap : kEmitterConstant;
        const float *bb = s->bvh.bbox;
        e.cx = (bb[3] + bb[0]) * 0.5f; e.cy = (bb[4] + bb[1]) * 0.5f; e.cz = (bb[5] + bb[2]) * 0.5f;
        const float dx = e.cx - bb[3], dy = e.cy - bb[4], dz = e.cz - bb[5];
        const float r = std::sqrt(std::fma(dz, dz, std::fma(dy, dy, dx * dx)));
        e.radius = std::max(kRayEpsilon, r * (1.0f + kRayEpsilon));
    }

    for (DevEmitter &e : s->emitters) {           // DirectionalEmitter::set_scene (directional.cpp:65-70)
        if (e.pad0 != kEmitterDirectional) continue;
        const float *bb = s->bvh.bbox;
        const float cx = (bb[3] + bb[0]) * 0.5f, cy = (bb[4] + bb[1]) * 0.5f, cz = (bb[5] + bb[2]) * 0.5f;
        const float dx = cx - bb[3], dy = cy - bb[4], dz = cz - bb[5];
        const float r = std::sqrt(std::fma(dz, dz, std::fma(dy, dy, dx * dx)));
        e.radius = std::max(kRayEpsilon, r * (1.0f + kRayEpsilon));
    }

    std::vector<float4> nodes(4 * (size_t) s->bvh.n_nodes), tris(3 * (size_t) s->bvh.n_slots);
    std::memcpy(nodes.data(), s->bvh.nodes.data(), s->bvh.nodes.size() * sizeof(float));
    std::vector<uint4> qnodes(2 * (size_t) s->bvh.n_nodes);
    std::memcpy(qnodes.data(), s->bvh.qnodes.data(), s->bvh.qnodes.size() * sizeof(uint32_t));
    std::vector<uint4> wnodes(4 * (size_t) s->bvh.n_wnodes);
    std::memcpy(wnodes.data(), (MTS_NODE_P15 ? s->bvh.wnodes_p : (MTS_NODE_F16 ? s->bvh.wnodes_h : s->bvh.wnodes)).data(), s->bvh.wnodes.size() * sizeof(uint32_t));
    std::memcpy(tris.data(), s->bvh.tris.data(), s->bvh.tris.size() * sizeof(float));
    // flat scenes: 64-byte records in primitive order (device_scene.h)
    uint32_t flat_max = kFlatMaxPrims;
    if (const char *e = exp_env("MTSAMD_FLAT_MAX")) flat_max = std::min<uint32_t>(kFlatMaxPrims, (uint32_t) std::strtoul(e, nullptr, 10));   // experiment switch
    const bool flat = s->n_prims <= flat_max;
    std::vector<float4> flat_recs(flat ? 4 * (size_t) s->n_prims : 0);
    for (uint32_t gp = 0; flat && gp < s->n_prims; ++gp) {
        const float *tp = &tri_pos[9 * (size_t) gp];
        uint32_t sh = prim_shape[gp]; float shf; std::memcpy(&shf, &sh, 4);
        flat_recs[4 * gp + 0] = make_float4(tp[0], tp[1], tp[2], tp[3] - tp[0]);
        flat_recs[4 * gp + 1] = make_float4(tp[4] - tp[1], tp[5] - tp[2], tp[6] - tp[0], tp[7] - tp[1]);
        flat_recs[4 * gp + 2] = make_float4(tp[8] - tp[2], tp[3], tp[4], tp[5]);
        flat_recs[4 * gp + 3] = make_float4(tp[6], tp[7], tp[8], shf);
    }
    const uint32_t n_pairs = flat ? (s->n_prims + 1) / 2 : 0;
    std::vector<float4> pair_recs(5 * (size_t) n_pairs, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    for (uint32_t k = 0; k < n_pairs; ++k) {
        float a[9] = { 0 }, b[9] = { 0 };       // p0, e1, e2 of primitives 2k and 2k+1 (zero = never hit)
        for (int which = 0; which < 2; ++which) {
            uint32_t gp = 2 * k + which;
            if (gp >= s->n_prims) continue;
            const float *tp = &tri_pos[9 * (size_t) gp];
            float *r = which ? b : a;
            r[0] = tp[0]; r[1] = tp[1]; r[2] = tp[2];
            r[3] = tp[3] - tp[0]; r[4] = tp[4] - tp[1]; r[5] = tp[5] - tp[2];
            r[6] = tp[6] - tp[0]; r[7] = tp[7] - tp[1]; r[8] = tp[8] - tp[2];
        }
        pair_recs[5 * k + 0] = make_float4(a[0], b[0], a[1], b[1]);
        pair_recs[5 * k + 1] = make_float4(a[2], b[2], a[3], b[3]);
        pair_recs[5 * k + 2] = make_float4(a[4], b[4], a[5], b[5]);
        pair_recs[5 * k + 3] = make_float4(a[6], b[6], a[7], b[7]);
        pair_recs[5 * k + 4] = make_float4(a[8], b[8], 0.0f, 0.0f);
    }
    // clusters of consecutive pairs that belong to one shape, with their bounding box padded by 1e-4 of the scene's extent (the box
    // test of coherent waves only culls; device_scene.h, traverse_flat_clustered).  Appended to the pair records.
    uint32_t n_clusters = 0;
    if (flat && n_pairs > 0) {
        float ext = 0.0f;
        for (uint32_t gp = 0; gp < s->n_prims; ++gp) for (int q = 0; q < 9; ++q) ext = std::max(ext, std::fabs(tri_pos[9 * (size_t) gp + q]));
        const float pad = 1e-4f * std::max(ext, 1e-3f);
        uint32_t k0 = 0;
        while (k0 < n_pairs) {
            uint32_t k1 = k0 + 1;
            while (k1 < n_pairs && prim_shape[2 * k1] == prim_shape[2 * k0]) ++k1;
            float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
            for (uint32_t gp = 2 * k0; gp < std::min(2 * k1, s->n_prims); ++gp)
                for (int vtx = 0; vtx < 3; ++vtx) for (int a = 0; a < 3; ++a) {
                    lo[a] = std::min(lo[a], tri_pos[9 * (size_t) gp + 3 * vtx + a]); hi[a] = std::max(hi[a], tri_pos[9 * (size_t) gp + 3 * vtx + a]);
                }
            const uint32_t cnt = k1 - k0; float cntf; std::memcpy(&cntf, &cnt, 4);
            pair_recs.push_back(make_float4(lo[0] - pad, lo[1] - pad, lo[2] - pad, cntf));
            pair_recs.push_back(make_float4(hi[0] + pad, hi[1] + pad, hi[2] + pad, 0.0f));
            ++n_clusters; k0 = k1;
        }
    }
    int rc = 0;
    if ((rc = upload(&s->d_textures, s->textures)) || (rc = upload(&s->d_flat, flat_recs)) || (rc = upload(&s->d_pairs, pair_recs)) || (rc = upload(&s->d_nodes, nodes)) || (rc = upload(&s->d_qnodes, qnodes)) || (rc = upload(&s->d_wnodes, wnodes)) || (rc = upload(&s->d_tris, tris)) || (rc = upload(&s->d_tri_pos, tri_pos)) ||
        (rc = upload(&s->d_tri_nrm, tri_nrm)) || (rc = upload(&s->d_tri_uv, tri_uv)) || (rc = upload(&s->d_prim_shape, prim_shape)) ||
        (rc = upload(&s->d_shapes, shapes)) || (rc = upload(&s->d_bsdfs, s->bsdfs)) || (rc = upload(&s->d_emitters, s->emitters)) ||
        (rc = upload(&s->d_area_pmf, area_pmf)) || (rc = upload(&s->d_area_cdf, area_cdf))) {
        mtsamd_scene_destroy(s);
        return rc;
    }
    // envmap emitter: texels + sampling hierarchy (envmap.cpp:66-125, distr_2d.h:200-312)
    if (s->environment >= 0 && desc->emitters[s->environment].type == MTSAMD_EMITTER_ENVMAP) {
        const mtsamd_emitter_desc &ed = desc->emitters[s->environment];
        EnvmapHost eh;
        if (!build_envmap(ed.envmap_data, ed.envmap_width, ed.envmap_height, eh) || eh.lv_offset.size() > (size_t) kEnvMaxLevels) {
            mtsamd_scene_destroy(s);
            return fail(MTSAMD_ERR_INVALID, "envmap: unsupported image size %d x %d", ed.envmap_width, ed.envmap_height);
        }
        if (desc->spectral) {
            // envmap.cpp:96-109: every texel becomes (model coefficients of the colour scaled to a 50% maximum, scale); the
            // sampling hierarchy stays the one built from the RGB luminance
            for (size_t i = 0; i < eh.texels.size() / 4; ++i) {
                float *px = eh.texels.data() + 4 * i;
                const float sc = std::max(std::max(px[0], px[1]), px[2]) * 2.0f, dn = std::max(1e-8f, sc);
                float rgb_norm[3] = { px[0] / dn, px[1] / dn, px[2] / dn }, coeff[3];
                srgb_model_fetch(model, rgb_norm, coeff);             // black: (0, 0, -inf), evaluates to 0 (srgb.cpp:31-33)
                px[0] = coeff[0]; px[1] = coeff[1]; px[2] = coeff[2]; px[3] = sc;
            }
        }
        DevEnvmap de{};
        std::vector<DevEnvmap> one(1);
        if ((rc = upload(&s->d_env_texels, eh.texels)) || (rc = upload(&s->d_env_warp, eh.warp))) { mtsamd_scene_destroy(s); return rc; }
        de.data = reinterpret_cast<const float4 *>(s->d_env_texels); de.warp = s->d_env_warp;
        s->env_w = ed.envmap_width; s->env_h = ed.envmap_height;
        de.w = ed.envmap_width; de.h = ed.envmap_height; de.n_levels = (int32_t) eh.lv_offset.size(); de.scale = ed.envmap_scale;
        for (size_t k = 0; k < eh.lv_offset.size(); ++k) { de.lv_offset[k] = eh.lv_offset[k]; de.lv_width[k] = eh.lv_width[k]; }
        for (int k = 0; k < 2; ++k) { de.patch_size[k] = eh.patch_size[k]; de.inv_patch_size[k] = eh.inv_patch_size[k]; de.max_patch_index[k] = eh.max_patch_index[k]; }
        const float *m = ed.to_world;
        const double a = m[0], b = m[1], c = m[2], d2 = m[4], e2 = m[5], f = m[6], g = m[8], h2 = m[9], i2 = m[10];
        const float lin[9] = { m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10] };
        const double det = a * (e2 * i2 - f * h2) - b * (d2 * i2 - f * g) + c * (d2 * h2 - e2 * g);
        if (det == 0.0) { mtsamd_scene_destroy(s); return fail(MTSAMD_ERR_INVALID, "envmap: singular to_world transformation"); }
        const double inv[9] = { (e2 * i2 - f * h2) / det, (c * h2 - b * i2) / det, (b * f - c * e2) / det,
                                (f * g - d2 * i2) / det, (a * i2 - c * g) / det, (c * d2 - a * f) / det,
                                (d2 * h2 - e2 * g) / det, (b * g - a * h2) / det, (a * e2 - b * d2) / det };
        for (int k = 0; k < 9; ++k) { de.to_world[k] = lin[k]; de.to_local[k] = (float) inv[k]; }
        one[0] = de;
        if ((rc = upload(&s->d_envmap, one))) { mtsamd_scene_destroy(s); return rc; }
    }
    // roughplastic: transmittance tables and internal reflectance are integrated on the device (roughplastic.cpp:380-399)
    {
        std::vector<uint32_t> rough;
        for (uint32_t b = 0; b < (uint32_t) s->bsdfs.size(); ++b) if (s->bsdfs[b].type == kBsdfRoughPlastic) rough.push_back(b);
        if (!rough.empty()) {
            float *d_gl = nullptr;
            if (hipMalloc((void **) &s->d_rough_tables, rough.size() * kRoughTableRes * sizeof(float)) != hipSuccess ||
                hipMalloc((void **) &d_gl, 512 * sizeof(float)) != hipSuccess) {
                mtsamd_scene_destroy(s);
                return fail(MTSAMD_ERR_NOMEM, "roughplastic tables: allocation failed");
            }
            hipError_t err = hipSuccess;
            for (size_t k = 0; k < rough.size() && err == hipSuccess; ++k) {
                const float eta = s->bsdfs[rough[k]].er;
                const int res_t = eta > 1.0f ? 32 : 128, res_r = (1.0f / eta) > 1.0f ? 32 : 128;      // microfacet.h:476-478,520-522
                float gl[512] = {};
                gauss_legendre(res_t, gl, gl + 128);
                gauss_legendre(res_r, gl + 256, gl + 384);
                err = hipMemcpy(d_gl, gl, sizeof(gl), hipMemcpyHostToDevice);
                if (err == hipSuccess) err = launch_roughplastic_tables(s->d_bsdfs, rough[k], s->d_rough_tables + k * kRoughTableRes, d_gl, res_t, res_r, nullptr);
                if (err == hipSuccess) err = hipDeviceSynchronize();
            }
            if (err == hipSuccess) err = hipMemcpy(s->bsdfs.data(), s->d_bsdfs, s->bsdfs.size() * sizeof(DevBsdf), hipMemcpyDeviceToHost);
            (void) hipFree(d_gl);
            if (err != hipSuccess) { mtsamd_scene_destroy(s); return fail(MTSAMD_ERR_DEVICE, "roughplastic tables: %s", hipGetErrorString(err)); }
        }
    }
    SceneView &v = s->view;
    v.nodes = s->d_nodes; v.qnodes = s->d_qnodes; v.wnodes = s->d_wnodes; v.wroot = s->bvh.wroot; v.n_wnodes = s->bvh.n_wnodes; v.tris = s->d_tris; v.root = s->bvh.root;
    for (int k = 0; k < 3; ++k) { v.q_lo[k] = s->bvh.q_lo[k]; v.q_step[k] = s->bvh.q_step[k]; v.q_inv_step[k] = 1.0f / s->bvh.q_step[k]; }
    v.n_nodes = s->bvh.n_nodes; v.n_slots = s->bvh.n_slots; v.n_prims = s->n_prims;
    // LDS residency: flat scenes keep everything in LDS (see flat_recs below).  For hierarchy scenes staging the
    // top of the tree (nodes are stored in BFS order) was measured to LOSE: 384 staged nodes 2.5-3.2 Gray/s vs none
    // 3.8-4.8 Gray/s on a 261 k-triangle mesh -- the 24 KB cost occupancy and the LDS/global select compiles to
    // generic (flat) loads, while the top levels stay L1/L2-resident anyway.  Only the traversal stack lives in LDS.
    v.lds_nodes = 0; v.lds_slots = 0;      // nodes and triangle slots are always read through L1/L2
    // BVH2: one deferred subtree per level; BVH4: up to three
    v.stack_depth = MTS_BVH4 ? 3u * s->bvh.wdepth + 2u : std::max<uint32_t>(s->bvh.depth, 2);
    // standalone ray streams: 8 persistent workgroups per CU, the first 12 (BVH2: 16) stack entries of a lane in LDS
    v.walk_lds_depth = std::min<uint32_t>(v.stack_depth, MTS_BVH4 ? 8u : 16u);      // 16 KB per workgroup: 8 workgroups (k_ray_walk: 8 waves per SIMD) per CU
    v.walk_blocks = 8u * (uint32_t) s->cu_count;
    if (!flat && v.stack_depth > v.walk_lds_depth) {
        const size_t entries = (size_t) v.walk_blocks * (v.stack_depth - v.walk_lds_depth) * 256u;
        if (ws_alloc((void **) &s->d_walk_spill, entries * sizeof(StackEntry))) { mtsamd_scene_destroy(s); return fail(MTSAMD_ERR_NOMEM, "traversal spill area"); }
    }
    v.walk_spill = s->d_walk_spill;
    v.tri_pos = s->d_tri_pos; v.tri_nrm = any_nrm ? s->d_tri_nrm : nullptr; v.tri_uv = any_uv ? s->d_tri_uv : nullptr;
    v.prim_shape = s->d_prim_shape; v.shapes = s->d_shapes; v.bsdfs = s->d_bsdfs;
    v.emitters = s->d_emitters; v.n_emitters = desc->emitter_count;
    v.env_emitter = s->environment; v.envmap = s->d_envmap;
    v.area_pmf = s->d_area_pmf; v.area_cdf = s->d_area_cdf;
    v.n_shapes = desc->mesh_count; v.n_bsdfs = desc->bsdf_count;
    v.textures = s->d_textures; v.n_textures = desc->texture_count;
    v.flat_recs = s->d_flat; v.flat = flat ? 1u : 0u;
    v.general = s->nested_bsdfs ? 2u : (s->general_bsdfs || s->delta_emitters || s->environment >= 0) ? 1u : 0u;       // the diffuse / area-light fast path (kernels.hip) handles none of these
    v.flat_pairs = s->d_pairs; v.n_pairs = n_pairs; v.n_clusters = n_clusters;
    if (bounce_lds_bytes(v) > 150 * 1024) {
        mtsamd_scene_destroy(s);
        return fail(MTSAMD_ERR_UNSUPPORTED, "BVH too deep for the LDS traversal stack (depth %u)", v.stack_depth);
    }
    *out = s;
    return MTSAMD_OK;
}

int mtsamd_scene_bbox(const mtsamd_scene *s, float *out6) {
    if (!s || !out6) return fail(MTSAMD_ERR_INVALID, "null argument");
    std::memcpy(out6, s->bvh.bbox, sizeof(float) * 6);
    return MTSAMD_OK;
}

int mtsamd_scene_info(const mtsamd_scene *s, uint32_t *out6) {
    if (!s || !out6) return fail(MTSAMD_ERR_INVALID, "null argument");
    out6[0] = s->n_prims; out6[1] = s->bvh.n_nodes; out6[2] = s->bvh.depth; out6[3] = s->n_shapes;
    out6[4] = (uint32_t) s->emitters.size(); out6[5] = s->view.lds_nodes;
    return MTSAMD_OK;
}

// Spectral variant: a colour-valued BSDF parameter (p = 0 reflectance, 1 specular_reflectance, 2 specular_transmittance) is an `srgb`
// spectrum -- range check, model coefficients, mean for the plastic lobe weights, exactly as mtsamd_scene_create does it
// (srgb.cpp:31-41, plastic.cpp:170-175).  Parameters given as `uniform` spectra and textured reflectances are not settable this way.
static int spectral_set_colour(mtsamd_scene *s, uint32_t bsdf, int p, const float *rgb) {
    DevBsdf &d = s->bsdfs[bsdf];
    const uint32_t uniform_flag = p == 0 ? kBsdfUniformRefl : (p == 1 ? kBsdfUniformSpec : kBsdfUniformTrans);
    if (d.type == kBsdfBlend || d.type == kBsdfMask || (d.flags & uniform_flag) || (p == 0 && d.texture >= 0))
        return fail(MTSAMD_ERR_UNSUPPORTED, "bsdf %u: this parameter is a uniform spectrum, a texture or a nesting weight; only srgb colours can be set in the spectral variant", bsdf);
    if (rgb[0] < 0 || rgb[1] < 0 || rgb[2] < 0 || rgb[0] > 1 || rgb[1] > 1 || rgb[2] > 1)
        return fail(MTSAMD_ERR_INVALID, "Invalid RGB reflectance value [%g, %g, %g], must be in the range [0, 1]!", rgb[0], rgb[1], rgb[2]);
    float coeff[3];
    srgb_model_fetch(s->rgb2spec, rgb, coeff);
    float *dst = p == 0 ? &d.c0 : (p == 1 ? &d.sc0 : &d.tc0);
    dst[0] = coeff[0]; dst[1] = coeff[1]; dst[2] = coeff[2];
    if (p == 0) { d.r = rgb[0]; d.g = rgb[1]; d.b = rgb[2]; s->diff_mean[bsdf] = srgb_model_mean(coeff); }
    if (p == 1) { d.sr = rgb[0]; d.sg = rgb[1]; d.sb = rgb[2]; s->spec_mean[bsdf] = srgb_model_mean(coeff); }
    if (d.type == kBsdfPlastic || d.type == kBsdfRoughPlastic) {
        const float d_mean = d.texture >= 0 ? s->textures[d.texture].mean : s->diff_mean[bsdf];
        d.kr = s->spec_mean[bsdf] / (d_mean + s->spec_mean[bsdf]);
    }
    HIP_TRY(hipMemcpy(s->d_bsdfs + bsdf, &d, sizeof(DevBsdf), hipMemcpyHostToDevice));
    return MTSAMD_OK;
}

int mtsamd_scene_set_bsdf_reflectance(mtsamd_scene *s, uint32_t bsdf, const float *rgb) {
    if (!s || !rgb || bsdf >= s->bsdfs.size()) return fail(MTSAMD_ERR_INVALID, "invalid bsdf index");
    HIP_TRY(hipSetDevice(s->device));
    if (s->spectral) return spectral_set_colour(s, bsdf, 0, rgb);
    s->bsdfs[bsdf].r = rgb[0]; s->bsdfs[bsdf].g = rgb[1]; s->bsdfs[bsdf].b = rgb[2];
    if (s->bsdfs[bsdf].type == kBsdfPlastic || s->bsdfs[bsdf].type == kBsdfRoughPlastic) {       // parameters_changed(): specular sampling weight (plastic.cpp:170-175)
        DevBsdf &d = s->bsdfs[bsdf];
        const float d_mean = d.texture >= 0 ? s->textures[d.texture].mean : (d.r + d.g + d.b) * (1.0f / 3.0f), s_mean = (d.sr + d.sg + d.sb) * (1.0f / 3.0f);
        d.kr = s_mean / (d_mean + s_mean);
    }
    HIP_TRY(hipMemcpy(s->d_bsdfs + bsdf, &s->bsdfs[bsdf], sizeof(DevBsdf), hipMemcpyHostToDevice));
    return MTSAMD_OK;
}

int mtsamd_scene_roughplastic_tables(const mtsamd_scene *s, uint32_t bsdf, float *out65) {
    if (!s || !out65 || bsdf >= s->bsdfs.size()) return fail(MTSAMD_ERR_INVALID, "invalid bsdf index");
    const DevBsdf &b = s->bsdfs[bsdf];
    if (b.type != kBsdfRoughPlastic || !b.table) return fail(MTSAMD_ERR_INVALID, "bsdf %u is not a roughplastic", bsdf);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpy(out65, b.table, kRoughTableRes * sizeof(float), hipMemcpyDeviceToHost));
    out65[kRoughTableRes] = b.eb;
    return MTSAMD_OK;
}

int mtsamd_scene_update_texture(mtsamd_scene *s, uint32_t texture, const float *rgb, void *stream) {
    if (!s || !rgb || texture >= s->textures.size()) return fail(MTSAMD_ERR_INVALID, "invalid texture index");
    HIP_TRY(hipSetDevice(s->device));
    const DevTexture &t = s->textures[texture];
    if (t.kind != 0) return fail(MTSAMD_ERR_INVALID, "texture %u is not a bitmap", texture);
    if (s->spectral) return fail(MTSAMD_ERR_UNSUPPORTED, "texture updates are implemented for the RGB variant only (the texels hold model coefficients)");
    HIP_TRY(hipMemcpyAsync((void *) t.data, rgb, sizeof(float) * 3 * (size_t) t.w * t.h, hipMemcpyDefault, (hipStream_t) stream));
    // parameters_changed() (bitmap.cpp:308-322): the mean follows the data; only plastic lobe weights read it
    bool used = false;
    for (const DevBsdf &b : s->bsdfs) used = used || (b.texture == (int32_t) texture && (b.type == kBsdfPlastic || b.type == kBsdfRoughPlastic));
    if (used) {
        std::vector<float> host(3 * (size_t) t.w * t.h);
        HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
        HIP_TRY(hipMemcpy(host.data(), t.data, host.size() * sizeof(float), hipMemcpyDeviceToHost));
        double mean = 0.0;
        for (size_t i = 0; i < host.size() / 3; ++i) mean += (double) (host[3 * i] * 0.212671f + host[3 * i + 1] * 0.715160f + host[3 * i + 2] * 0.072169f);
        s->textures[texture].mean = (float) (mean / (double) (host.size() / 3));
        for (size_t b = 0; b < s->bsdfs.size(); ++b) {
            DevBsdf &d = s->bsdfs[b];
            if (d.texture != (int32_t) texture || (d.type != kBsdfPlastic && d.type != kBsdfRoughPlastic)) continue;
            d.kr = s->spec_mean[b] / (s->textures[texture].mean + s->spec_mean[b]);
            HIP_TRY(hipMemcpy(s->d_bsdfs + b, &d, sizeof(DevBsdf), hipMemcpyHostToDevice));
        }
    }
    return MTSAMD_OK;
}

int mtsamd_scene_set_emitter_radiance(mtsamd_scene *s, uint32_t emitter, const float *rgb) {
    if (!s || !rgb || emitter >= s->emitters.size()) return fail(MTSAMD_ERR_INVALID, "invalid emitter index");
    HIP_TRY(hipSetDevice(s->device));
    s->emitters[emitter].r = rgb[0]; s->emitters[emitter].g = rgb[1]; s->emitters[emitter].b = rgb[2];
    if (s->spectral) {       // srgb_d65 spectrum: normalised colour -> coefficients, the scale rides on the D65 curve (srgb_d65.cpp:31-46)
        float color[3] = { rgb[0], rgb[1], rgb[2] }, coeff[3];
        const float scale = std::max(std::max(color[0], color[1]), color[2]) * 2.0f;
        if (scale != 0.0f) { const float r = 1.0f / scale; for (float &v : color) v *= r; }
        srgb_model_fetch(s->rgb2spec, color, coeff);
        float d65_scale = 1.0f * scale;
        d65_scale *= 1.0f / 10568.0f;                      // d65.cpp:44-50
        DevEmitter &d = s->emitters[emitter];
        d.c0 = coeff[0]; d.c1 = coeff[1]; d.c2 = coeff[2]; d.d65_scale = d65_scale;
    }
    HIP_TRY(hipMemcpy(s->d_emitters + emitter, &s->emitters[emitter], sizeof(DevEmitter), hipMemcpyHostToDevice));
    return MTSAMD_OK;
}

// ---- scene queries -------------------------------------------------------------------------
static int check_rays(const mtsamd_scene *s, const mtsamd_rays *r, uint64_t n) {
    if (!s || !r) return fail(MTSAMD_ERR_INVALID, "null argument");
    if (n == 0) return 1;               // empty ray stream: nothing to do (pointers may be null)
    if (!r->ox || !r->oy || !r->oz || !r->dx || !r->dy || !r->dz || !r->mint || !r->maxt)
        return fail(MTSAMD_ERR_INVALID, "ray stream has a null component");
    return 0;
}
static RayStreams to_streams(const mtsamd_rays *r) {
    return RayStreams{ r->ox, r->oy, r->oz, r->dx, r->dy, r->dz, r->mint, r->maxt, r->active };
}

int mtsamd_ray_intersect(const mtsamd_scene *s, uint64_t n, const mtsamd_rays *rays, float *t, uint32_t *prim,
                         uint32_t *shape, float *u, float *v, void *stream) {
    if (int rc = check_rays(s, rays, n)) return rc > 0 ? MTSAMD_OK : rc;
    if (!t || !prim) return fail(MTSAMD_ERR_INVALID, "t and prim outputs are required");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch_ray_intersect(s->view, n, to_streams(rays), 0, t, prim, shape, u, v, nullptr, (hipStream_t) stream));
    return MTSAMD_OK;
}

int mtsamd_ray_intersect_naive(const mtsamd_scene *s, uint64_t n, const mtsamd_rays *rays, float *t, uint32_t *prim,
                               uint32_t *shape, float *u, float *v, void *stream) {
    if (int rc = check_rays(s, rays, n)) return rc > 0 ? MTSAMD_OK : rc;
    if (!t || !prim) return fail(MTSAMD_ERR_INVALID, "t and prim outputs are required");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch_ray_intersect(s->view, n, to_streams(rays), 1, t, prim, shape, u, v, nullptr, (hipStream_t) stream));
    return MTSAMD_OK;
}

int mtsamd_ray_intersect_si(const mtsamd_scene *s, uint64_t n, const mtsamd_rays *rays, float *t, uint32_t *prim,
                            uint32_t *shape, float *si26, void *stream) {
    if (int rc = check_rays(s, rays, n)) return rc > 0 ? MTSAMD_OK : rc;
    if (!t || !prim || !si26) return fail(MTSAMD_ERR_INVALID, "t, prim and si26 outputs are required");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch_ray_intersect(s->view, n, to_streams(rays), 0, t, prim, shape, nullptr, nullptr, si26, (hipStream_t) stream));
    return MTSAMD_OK;
}

int mtsamd_ray_test(const mtsamd_scene *s, uint64_t n, const mtsamd_rays *rays, uint8_t *hit, void *stream) {
    if (int rc = check_rays(s, rays, n)) return rc > 0 ? MTSAMD_OK : rc;
    if (!hit) return fail(MTSAMD_ERR_INVALID, "hit output is required");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(launch_ray_test(s->view, n, to_streams(rays), hit, (hipStream_t) stream));
    return MTSAMD_OK;
}

// ---- render ----------------------------------------------------------------------------------
static int check_desc(const mtsamd_render_desc *d) {
    if (!d) return fail(MTSAMD_ERR_INVALID, "null render descriptor");
    // MonteCarloIntegrator (integrator.cpp:288-295)
    if (d->max_depth < 0 && d->max_depth != -1)
        return fail(MTSAMD_ERR_INVALID, "\"max_depth\" must be set to -1 (infinite) or a value >= 0");
    if (d->rr_depth <= 0) return fail(MTSAMD_ERR_INVALID, "\"rr_depth\" must be set to a value greater than zero!");
    if (d->film_width <= 0 || d->film_height <= 0 || d->crop_width <= 0 || d->crop_height <= 0 || d->crop_x < 0 || d->crop_y < 0 ||
        d->crop_x + d->crop_width > d->film_width || d->crop_y + d->crop_height > d->film_height)
        return fail(MTSAMD_ERR_INVALID, "Invalid crop window specification!");      // film.cpp:24-32
    if (d->sample_count <= 0) return fail(MTSAMD_ERR_INVALID, "sample_count must be positive");
    if (d->samples_per_pass > 0 && d->sample_count % std::min(d->samples_per_pass, d->sample_count) != 0)      // integrator.cpp:59-66
        return fail(MTSAMD_ERR_INVALID, "sample_count (%d) must be a multiple of samples_per_pass (%d).", d->sample_count,
                    std::min(d->samples_per_pass, d->sample_count));
    if (d->integrator < 0 || d->integrator > 2) return fail(MTSAMD_ERR_UNSUPPORTED, "integrator %d is not implemented (0 path, 1 direct, 2 depth)", d->integrator);
    if (d->emitter_samples < 0 || d->bsdf_samples < 0) return fail(MTSAMD_ERR_INVALID, "Must have at least 1 BSDF or emitter sample!");
    if (d->pipeline < 0 || d->pipeline > 4) return fail(MTSAMD_ERR_UNSUPPORTED, "pipeline %d is not available in this build", d->pipeline);
    return 0;
}

static int ensure_workspace(mtsamd_scene *s, uint32_t n_waves, uint32_t seg_cap, uint64_t pass_cap, bool split) {
    Workspace &w = s->ws;
    if (w.n_waves == n_waves && w.seg_cap == seg_cap && w.pass_cap >= pass_cap && w.spectral == s->spectral && (w.split || !split)) return 0;
    w.release();
    struct Guard { Workspace &w; bool ok = false; ~Guard() { if (!ok) w.release(); } } guard{ w };      // nothing half-allocated survives an error
    size_t slots = (size_t) n_waves * seg_cap;
    for (int k = 0; k < 2; ++k) {
        if (int rc = ws_alloc((void **) &w.pool[k].ray_o, slots * sizeof(float4))) return rc;
        if (int rc = ws_alloc((void **) &w.pool[k].ray_d, slots * sizeof(float4))) return rc;
        if (int rc = ws_alloc((void **) &w.pool[k].thr, slots * sizeof(float4))) return rc;
        if (int rc = ws_alloc((void **) &w.pool[k].res, slots * sizeof(float4))) return rc;
        if (int rc = ws_alloc((void **) &w.pool[k].rng, slots * sizeof(uint4))) return rc;
        if (int rc = ws_alloc((void **) &w.pool[k].misc, slots * sizeof(uint2))) return rc;
        if (int rc = ws_alloc((void **) &w.count[k], 2 * (size_t) n_waves * sizeof(uint32_t))) return rc;      // counts + survivor borders (k_shade, flat scenes)
        if (s->spectral) {
            if (int rc = ws_alloc((void **) &w.pool[k].xi, slots * sizeof(float))) return rc;
            if (int rc = ws_alloc((void **) &w.pool[k].aux, slots * sizeof(float2))) return rc;
        }
        if (split) {
            if (int rc = ws_alloc((void **) &w.pool[k].hit, slots * sizeof(float4))) return rc;
            if (int rc = ws_alloc((void **) &w.pool[k].sh_o, slots * sizeof(float4))) return rc;
            if (int rc = ws_alloc((void **) &w.pool[k].sh_d, slots * sizeof(float4))) return rc;
            if (int rc = ws_alloc((void **) &w.pool[k].nee, slots * sizeof(float4))) return rc;
            if (int rc = ws_alloc((void **) &w.pool[k].sh_slot, slots * sizeof(uint32_t))) return rc;
        }
    }
    if (int rc = ws_alloc((void **) &w.cursor, n_waves * sizeof(uint64_t))) return rc;
    if (int rc = ws_alloc((void **) &w.cursor_end, n_waves * sizeof(uint64_t))) return rc;
    if (int rc = ws_alloc((void **) &w.wave_stats, 4 * (size_t) n_waves * sizeof(uint64_t))) return rc;
    if (int rc = ws_alloc((void **) &w.count_shadow, n_waves * sizeof(uint32_t))) return rc;
    if (int rc = ws_alloc((void **) &w.out_rgba, pass_cap * sizeof(float4))) return rc;
    if (int rc = ws_alloc((void **) &w.out_pos, pass_cap * sizeof(float2))) return rc;
    HIP_TRY(hipHostMalloc((void **) &w.h_counts, 4 * (size_t) n_waves * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **) &w.h_cursor, 3 * (size_t) n_waves * sizeof(uint64_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **) &w.h_cursor_rb, 2 * (size_t) n_waves * sizeof(uint64_t), hipHostMallocDefault));
    for (auto &e : w.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!w.stream2) HIP_TRY(hipStreamCreateWithFlags(&w.stream2, hipStreamNonBlocking));
    for (auto &ps : w.part_stream) if (!ps) HIP_TRY(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    for (auto &pe : w.part_ev) if (!pe) HIP_TRY(hipEventCreateWithFlags(&pe, hipEventDisableTiming));
    for (uint32_t k = 0; k < kMaxChains; ++k) {
        if (k > 0 && !w.chain_main[k]) HIP_TRY(hipStreamCreateWithFlags(&w.chain_main[k], hipStreamNonBlocking));
        if (!w.chain_any[k]) HIP_TRY(hipStreamCreateWithFlags(&w.chain_any[k], hipStreamNonBlocking));
    }
    for (auto &pe : w.chain_ev) if (!pe) HIP_TRY(hipEventCreateWithFlags(&pe, hipEventDisableTiming));
    for (auto &e : w.tev) HIP_TRY(hipEventCreate(&e));
    w.have_events = true;
    w.n_waves = n_waves; w.seg_cap = seg_cap; w.pass_cap = pass_cap; w.spectral = s->spectral; w.split = split;
    guard.ok = true;
    return 0;
}

namespace {
struct Job {
    mtsamd_scene *s; const mtsamd_render_desc *d; hipStream_t stream;
    CameraView cam; FilterView filter;
    uint32_t n_waves, target; uint64_t pass_cap;
    uint64_t iterations = 0;
    double bounce_ms = 0.0, film_ms = 0.0;
    RowMap rows{};
    int store_xyz = 1;
    uint32_t plane_pix0 = 0, plane_pixels = 0;
    bool split = false, shadow_queue = false, shadow_ring = false;
    double stage_ms[3] = { 0.0, 0.0, 0.0 }; uint64_t stage_launches[3] = { 0, 0, 0 };      // desc->profile: k_trace<closest>, k_shade, k_trace<any>
    uint64_t passes = 0;
    int buf = 0;                 // sample stream buffer this pass writes
    std::chrono::steady_clock::time_point t_start;       // m_render_timer (integrator.cpp:107)
    bool timed_out = false;
    bool expired() const {                               // should_stop() without m_stop (integrator.h:143-146)
        return d->timeout > 0.0f && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > (double) d->timeout;
    }
};

// Traces the local sample ordinals [first, first+n) of this render's rows to completion; results land in
// ws.out_rgba / out_pos (slot = ordinal - first).
int trace_pass(Job &j, uint64_t first, uint64_t n) {
    Workspace &w = j.s->ws;
    const uint32_t nw = j.n_waves;
    const uint64_t spp = (uint64_t) j.d->sample_count;
    // the pass's samples are dealt to the scheduling waves in chunks, round-robin (kernels.hip, cursor_sample): wave k owns the
    // chunks k, k + nw, ...; its cursor counts the samples of its own it has generated.  64-sample chunks for LDS-resident scenes
    // hierarchy scenes: four chunks of consecutive pixels per scheduling wave (a wave's 256 slots still hold neighbouring pixels, but
    // every wave sees four regions of the film, which evens out when the waves run dry: 1 / 4 / 16 / 64 chunks: 0 / +2.1 / +2.3 / +1.3 %
    // on the 261 k-triangle mesh at 1024 spp; 64-sample chunks as on flat scenes cost 5 %), at least 256 samples each
    uint64_t cpw = 4;
    if (const char *e = exp_env("MTSAMD_CHUNKS_PER_WAVE")) cpw = (uint64_t) std::min(64, std::max(1, atoi(e)));      // experiment switch
    const uint64_t chunk = j.s->view.flat ? 64u : std::max<uint64_t>({ (n + nw * cpw - 1) / (nw * cpw), std::min<uint64_t>(256u, (n + nw - 1) / nw), 1u });
    const uint64_t n_chunks = (n + chunk - 1u) / chunk, last_size = n - (n_chunks - 1u) * chunk;
    // hierarchy scenes run several launch chains over parts of the scheduling waves: their chunks alternate (kernels.h, chunk_owner)
    uint32_t n_chains = 1;
    if (!j.s->view.flat && j.split && nw >= 256u) n_chains = kTraceChains;
    if (const char *e = exp_env("MTSAMD_CHAINS")) n_chains = (uint32_t) std::min<int>(kMaxChains, std::max(1, atoi(e)));      // experiment switch
    if (exp_env("MTSAMD_ONE_CHAIN")) n_chains = 1;
    while (n_chains > 1 && nw / n_chains < 2u * kChainAlign) --n_chains;
    for (uint32_t k = 0; k < nw; ++k) {
        const uint64_t c0 = chunk_owner(k, nw, n_chains);      // this wave owns the chunks c0, c0 + nw, ...
        const uint64_t mine = c0 < n_chunks ? (n_chunks - 1u - c0) / nw + 1u : 0u;
        uint64_t samples = mine * chunk;
        if (mine && (n_chunks - 1u) % nw == c0) samples -= chunk - last_size;       // the last, partial chunk of the pass
        w.h_cursor[k] = 0; w.h_cursor[nw + k] = samples;
    }
    if (n >= (1ull << 31)) return fail(MTSAMD_ERR_INVALID, "a pass holds fewer than 2^31 samples");
    HIP_TRY(hipMemcpyAsync(w.cursor, w.h_cursor, nw * sizeof(uint64_t), hipMemcpyHostToDevice, j.stream));
    HIP_TRY(hipMemcpyAsync(w.cursor_end, w.h_cursor + nw, nw * sizeof(uint64_t), hipMemcpyHostToDevice, j.stream));
    HIP_TRY(hipMemsetAsync(w.count[0], 0, 2 * (size_t) nw * sizeof(uint32_t), j.stream));
    HIP_TRY(hipMemsetAsync(w.count[1], 0, 2 * (size_t) nw * sizeof(uint32_t), j.stream));

    RenderParams p{};
    p.sv = j.s->view; p.cam = j.cam;
    if (p.cam.aperture_radius > 0.0f) p.sv.general = std::max(p.sv.general, 1u);      // thin lens: aperture sampling lives in the general kernels
    p.cursor = w.cursor; p.cursor_end = w.cursor_end; p.wave_stats = w.wave_stats;
    p.out_rgba = j.buf ? w.out_rgba2 : w.out_rgba; p.out_pos = j.buf ? w.out_pos2 : w.out_pos;
    p.count_shadow = w.count_shadow;
    p.first_ordinal = first; p.first_pix = (uint32_t) (first / spp); p.first_rem = (uint32_t) (first % spp);
    p.chunk = (uint32_t) chunk; p.base_seed = j.d->seed;
    p.n_chains = n_chains;
    p.rows = j.rows; p.store_xyz = j.store_xyz;
    p.plane_pix0 = j.plane_pix0; p.plane_pixels = j.plane_pixels;
    p.n_waves = nw; p.seg_cap = w.seg_cap; p.target = j.target;
    p.spp = j.d->sample_count; p.crop_x = j.d->crop_x; p.crop_y = j.d->crop_y; p.crop_w = j.d->crop_width; p.crop_h = j.d->crop_height;
    p.max_depth = j.d->max_depth; p.rr_depth = j.d->rr_depth;
    p.spectral = j.s->spectral ? 1 : 0;
    p.split = j.shadow_ring ? 3 : (j.shadow_queue ? 2 : (j.split ? 1 : 0));
    if (p.split == 1) {       // k_trace: short per-lane stack in LDS, deep entries in a global spill area
        const size_t words = trace_spill_words(p.sv, nw);
        if (words > w.trace_spill_words) {
            (void) hipFree(w.trace_spill); w.trace_spill = nullptr; w.trace_spill_words = 0;
            if (int rc = ws_alloc((void **) &w.trace_spill, std::max<size_t>(words, 1) * sizeof(uint32_t))) return rc;
            w.trace_spill_words = words;
        }
        p.trace_lds_depth = trace_lds_depth(p.sv); p.trace_top_nodes = trace_top_nodes(p.sv); p.trace_spill = w.trace_spill;
    }
    p.integrator = j.d->integrator; p.emitter_samples = j.d->emitter_samples; p.bsdf_samples = j.d->bsdf_samples;
    p.hide_emitters = j.d->hide_emitters;
    if (j.d->integrator != 0) {          // direct / depth: one launch finishes the whole pass
        HIP_TRY(hipEventRecord(w.tev[0], j.stream));
        HIP_TRY(launch_direct(p, n, j.stream));
        HIP_TRY(hipEventRecord(w.tev[1], j.stream));
        HIP_TRY(hipEventSynchronize(w.tev[1]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, w.tev[0], w.tev[1]));
        j.bounce_ms += ms; j.iterations += 1;
        return 0;
    }

    // Small passes of the automatic schedule: one launch of persistent lanes instead of launch rounds (kernels.hip, k_mega).  At most a
    // few samples per lane the launch count, not the kernel, sets the time: differentiable cbox 256^2 @ 1 spp, forward render
    // 0.33 ms of launch rounds.  LDS-resident scenes up to 2^19 samples, hierarchy scenes (where the wavefront kernels win sooner) 2^17.
    const uint64_t small_pass = j.s->view.flat ? (1ull << 19) : (1ull << 17);
    if (((j.d->pipeline == 0 && n <= small_pass) || exp_env("MTSAMD_MEGA")) && (p.split == 1 || p.split == 3) && !j.s->nested_bsdfs) {
        HIP_TRY(hipEventRecord(w.tev[0], j.stream));
        HIP_TRY(launch_mega(p, j.stream));
        HIP_TRY(hipEventRecord(w.tev[1], j.stream));
        HIP_TRY(hipEventSynchronize(w.tev[1]));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, w.tev[0], w.tev[1]));
        j.bounce_ms += ms; j.iterations += 1;
        return 0;
    }
    // the sample cursors cannot run dry before this many launches
    const uint64_t min_iters = (n + (uint64_t) nw * j.target - 1) / ((uint64_t) nw * j.target);
    uint64_t it = 0;
    int cur = 0;
    // Termination test without stalling the device: every `stride` launches the per-wave path counts are copied to
    // pinned memory; the copy issued at the previous checkpoint (long complete) is inspected before issuing a new one.
    // While samples are left to generate a checkpoint every fourth launch round does; once the cursors are dry the pool only shrinks,
    // the rounds get short and every round is checked (against the counts of the round before), so that k_finish takes over as soon
    // as the pool is small enough
    uint64_t stride = 4, next_check = 0;
    int pending = -1, slot = 0;
    HIP_TRY(hipEventRecord(w.tev[0], j.stream));
    uint32_t n_parts = 2;
    if (const char *e = exp_env("MTSAMD_STREAMS")) n_parts = (uint32_t) std::min(4, std::max(1, atoi(e)));      // experiment switch
    if (p.split != 3 || nw < 256u) n_parts = 1;
    uint32_t part_lo[5] = { 0, nw, nw, nw, nw };
    for (uint32_t k = 1; k < n_parts; ++k) part_lo[k] = (uint32_t) (((uint64_t) nw * k / n_parts + 3u) & ~3ull);
    part_lo[n_parts] = nw;
    // Pool drain of LDS-resident scenes: once every cursor is dry, a workgroup gathers the paths of gather_w consecutive scheduling
    // waves at the front of the group (k_shade).  gather_w grows by powers of four as the pool empties -- decided on the counts read
    // back every `stride` launches; they are upper bounds, counts only shrink from then on -- and its groups lie inside one part.
    // pool size below which k_finish ends the pass (0: never).  Hierarchy scenes: measured flat between 2^20 and 2^24 (the fused kernel
    // keeps up with the launch rounds of the split pipeline once they are no longer full): 2^22; LDS-resident scenes, whose drain is
    // already compacted by the gathering below: 2^18
    uint64_t finish_at = p.split == 1 ? (1ull << 22) : (p.split == 3 ? (1ull << 18) : 0ull);
    if (j.d->finish_kernel == 1) finish_at = 0;                      // never (tests: the launch rounds run the pool dry)
    else if (j.d->finish_kernel == 2) finish_at = 1ull << 40;       // as soon as the cursors are dry
    if (j.s->nested_bsdfs) finish_at = 0;                            // blendbsdf / mask: only the fused kernels carry the nesting code
    bool finish = false;
    uint64_t finish_alive = 0;
    uint32_t gather_max = 4u, gather_w = 4u;
    if (p.split == 3 && !exp_env("MTSAMD_NO_GATHER")) {
        gather_max = 1024u;
        for (uint32_t k = 0; k <= n_parts; ++k) while (gather_max > 4u && part_lo[k] % gather_max) gather_max >>= 2;
    }
    const uint32_t split_parts = p.split == 1 ? n_chains : 1u;
    uint32_t split_lo[kMaxChains + 1];
    for (uint32_t k = 0; k <= kMaxChains; ++k) split_lo[k] = chain_first(k, nw, split_parts);      // multiples of the k_trace group size
    auto chain_main = [&](uint32_t k) { return k == 0 ? j.stream : w.chain_main[k]; };
    if (split_parts > 1) {       // the other chains start after the cursors and counts are in place
        HIP_TRY(hipEventRecord(w.part_ev[2], j.stream));
        for (uint32_t k = 1; k < split_parts; ++k) HIP_TRY(hipStreamWaitEvent(chain_main(k), w.part_ev[2], 0));
    }
    auto join_chains = [&]() -> int {        // j.stream waits for what the other chains' main streams hold (their k_shade writes their counts)
        for (uint32_t k = 1; k < split_parts; ++k) {
            HIP_TRY(hipEventRecord(w.chain_ev[2 * k], chain_main(k)));
            HIP_TRY(hipStreamWaitEvent(j.stream, w.chain_ev[2 * k], 0));
        }
        return 0;
    };
    // desc->profile: begin / end timing events around the launches of the split pipeline, each on the stream of its launch
    const bool prof = j.d->profile != 0 && p.split == 1;
    struct ProfRec { int stage; size_t e0; };
    std::vector<ProfRec> prof_recs;
    size_t prof_used = 0;
    hipError_t prof_err = hipSuccess;
    auto prof_mark = [&](hipStream_t st) -> size_t {
        if (prof_used == w.prof_ev.size()) {
            hipEvent_t e = nullptr;
            const hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) { prof_err = rc; return 0; }
            w.prof_ev.push_back(e);
        }
        const hipError_t rc = hipEventRecord(w.prof_ev[prof_used], st);
        if (rc != hipSuccess) prof_err = rc;
        return prof_used++;
    };
    auto sync_all = [&]() {
        (void) hipStreamSynchronize(j.stream);
        if (w.film_stream) (void) hipStreamSynchronize(w.film_stream);
        if (w.stream2) (void) hipStreamSynchronize(w.stream2);
        for (auto &ps : w.part_stream) if (ps) (void) hipStreamSynchronize(ps);
        for (auto &ps : w.chain_main) if (ps) (void) hipStreamSynchronize(ps);
        for (auto &ps : w.chain_any) if (ps) (void) hipStreamSynchronize(ps);
    };
    auto join_parts = [&]() -> int {         // j.stream waits for the other parts' streams
        for (uint32_t k = 1; k < n_parts; ++k) {
            HIP_TRY(hipEventRecord(w.part_ev[k - 1], w.part_stream[k - 1]));
            HIP_TRY(hipStreamWaitEvent(j.stream, w.part_ev[k - 1], 0));
        }
        return 0;
    };
    if (n_parts > 1) {          // the other streams start after the cursors and counts are in place
        HIP_TRY(hipEventRecord(w.ev[2], j.stream));
        for (uint32_t k = 1; k < n_parts; ++k) HIP_TRY(hipStreamWaitEvent(w.part_stream[k - 1], w.ev[2], 0));
    }
    while (true) {
        if (j.s->cancel.load(std::memory_order_relaxed)) {
            sync_all();
            return fail(MTSAMD_ERR_CANCELLED, "render cancelled");
        }
        if (j.expired()) {                   // timeout: this pass is abandoned (a block that was not finished is never put)
            sync_all();
            j.timed_out = true;
            return 1;
        }
        p.in = w.pool[cur]; p.out = w.pool[cur ^ 1];
        p.count_in = w.count[cur]; p.count_out = w.count[cur ^ 1];
        if (p.split == 1) {
            // k_trace<any> of iteration i only adds to the radiance of the pool that iteration i + 1 reads its rays from: it runs on
            // a second stream beside k_trace<closest> of iteration i + 1 (the two fill each other's launch tails); k_shade waits for
            // it.  The scheduling waves are independent, so two such chains (halves of the waves) run side by side.
            for (uint32_t k = 0; k < split_parts; ++k) {
                RenderParams h = p;
                h.wave_first = split_lo[k]; h.wave_last = split_lo[k + 1];
                hipStream_t s_main = chain_main(k), s_any = w.chain_any[k];
                hipEvent_t e_shade = w.chain_ev[2 * k], e_any = w.chain_ev[2 * k + 1];
                size_t pe = 0;
                if (prof) pe = prof_mark(s_main);
                HIP_TRY(launch_split_stage(h, 0, s_main));
                if (prof) { prof_mark(s_main); prof_recs.push_back({ 0, pe }); }
                if (it > 0) HIP_TRY(hipStreamWaitEvent(s_main, e_any, 0));
                if (prof) pe = prof_mark(s_main);
                HIP_TRY(launch_split_stage(h, 1, s_main));
                if (prof) { prof_mark(s_main); prof_recs.push_back({ 1, pe }); }
                HIP_TRY(hipEventRecord(e_shade, s_main));
                HIP_TRY(hipStreamWaitEvent(s_any, e_shade, 0));
                if (prof) pe = prof_mark(s_any);
                HIP_TRY(launch_split_stage(h, 2, s_any));
                if (prof) { prof_mark(s_any); prof_recs.push_back({ 2, pe }); }
                HIP_TRY(hipEventRecord(e_any, s_any));
                HIP_TRY(prof_err);
            }
        } else if (p.split == 3 && n_parts > 1) {
            // the scheduling waves are independent of each other: part-size launches on their own streams advance in their own
            // rhythm and fill each other's launch tails
            RenderParams h = p;
            h.gather_w = gather_w;
            for (uint32_t k = 0; k < n_parts; ++k) {
                h.wave_first = part_lo[k]; h.wave_last = part_lo[k + 1];
                HIP_TRY(launch_bounce(h, k == 0 ? j.stream : w.part_stream[k - 1]));
            }
        } else {
            p.gather_w = gather_w;
            HIP_TRY(launch_bounce(p, j.stream));
        }
        cur ^= 1; ++it;
        if (exp_env("MTSAMD_TRACE_ITERS")) {      // diagnostic: alive paths and wall time of every scheduler iteration (serialises the loop)
            sync_all();
            static thread_local std::vector<uint32_t> hc;
            hc.resize(nw);
            HIP_TRY(hipMemcpy(hc.data(), w.count[cur], nw * sizeof(uint32_t), hipMemcpyDeviceToHost));
            uint64_t alive = 0, busy_waves = 0;
            for (uint32_t k = 0; k < nw; ++k) { alive += hc[k]; busy_waves += hc[k] ? 1 : 0; }
            static thread_local std::chrono::steady_clock::time_point t_prev;
            const auto t_now = std::chrono::steady_clock::now();
            fprintf(stderr, "iter %llu alive %llu waves_with_paths %llu dt_us %.0f gather_w %u\n", (unsigned long long) it, (unsigned long long) alive,
                    (unsigned long long) busy_waves, it > 1 ? std::chrono::duration<double, std::micro>(t_now - t_prev).count() : 0.0, gather_w);
            t_prev = std::chrono::steady_clock::now();
        }
        if (it >= min_iters && it >= next_check) {
            next_check = it + stride;
            if (pending >= 0) {
                HIP_TRY(hipEventSynchronize(w.ev[pending]));
                uint64_t alive = 0;
                const uint32_t *hc = w.h_counts + (size_t) pending * nw;
                for (uint32_t k = 0; k < nw; ++k) alive += hc[k];
                if (alive == 0) break;
                bool dry = false;
                const uint64_t *hcur = w.h_cursor_rb + (size_t) pending * nw;
                if (gather_w < gather_max || finish_at) {
                    dry = true;
                    for (uint32_t k = 0; k < nw && dry; ++k) dry = hcur[k] >= w.h_cursor[nw + k];
                }
                // every sample has been generated and few paths are left (the counts are a few launches old: an upper bound): one
                // k_finish launch instead of the dozens of near-empty launch rounds the deepest paths would still need
                if (dry && alive <= finish_at) { finish = true; finish_alive = alive; break; }
                if (dry && finish_at) { stride = 1; next_check = it + 1; }
                if (gather_w < gather_max) {
                    if (exp_env("MTSAMD_TRACE_ITERS")) {
                        uint32_t wet = 0, first_wet = 0;
                        for (uint32_t k = 0; k < nw; ++k) if (hcur[k] < w.h_cursor[nw + k]) { if (!wet) first_wet = k; ++wet; }
                        fprintf(stderr, "check at it %llu: alive %llu wet %u first_wet %u cur %llu end %llu\n", (unsigned long long) it, (unsigned long long) alive, wet, first_wet,
                                (unsigned long long) hcur[first_wet], (unsigned long long) w.h_cursor[nw + first_wet]);
                    }
                    // a workgroup may take up to eight segments' worth of paths on average (the counts are a few launches old: an
                    // upper bound); a fuller group just takes longer, its survivors spill into the group's next waves
                    while (dry && gather_w < gather_max && alive * (uint64_t) (4u * gather_w) <= 8ull * w.seg_cap * (uint64_t) nw) gather_w *= 4u;
                }
            }
            if (n_parts > 1) { if (int rc = join_parts()) return rc; }
            if (split_parts > 1) { if (int rc = join_chains()) return rc; }      // the counts of the other chains are written by their k_shade
            HIP_TRY(hipMemcpyAsync(w.h_counts + (size_t) slot * nw, w.count[cur], nw * sizeof(uint32_t), hipMemcpyDeviceToHost, j.stream));
            if (gather_w < gather_max || finish_at) HIP_TRY(hipMemcpyAsync(w.h_cursor_rb + (size_t) slot * nw, w.cursor, nw * sizeof(uint64_t), hipMemcpyDeviceToHost, j.stream));
            HIP_TRY(hipEventRecord(w.ev[slot], j.stream));
            pending = slot; slot ^= 1;
        }
        if (it > (1ull << 24)) return fail(MTSAMD_ERR_DEVICE, "wavefront scheduler did not converge");
    }
    if (p.split == 1 && it > 0) {
        for (uint32_t k = 0; k < split_parts; ++k) HIP_TRY(hipStreamWaitEvent(j.stream, w.chain_ev[2 * k + 1], 0));      // the last k_trace<any> of every chain
        if (int rc = join_chains()) return rc;
    }
    if (n_parts > 1) { if (int rc = join_parts()) return rc; }
    if (finish) {          // every stream of the loop has been joined into j.stream
        RenderParams h = p;
        h.in = w.pool[cur]; h.out = w.pool[cur ^ 1];
        h.count_in = w.count[cur]; h.count_out = w.count[cur ^ 1];
        HIP_TRY(launch_finish(h, finish_alive, j.stream));
        ++it;
    }
    HIP_TRY(hipEventRecord(w.tev[1], j.stream));
    HIP_TRY(hipEventSynchronize(w.tev[1]));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, w.tev[0], w.tev[1]));
    j.bounce_ms += ms;
    j.iterations += it;
    for (const ProfRec &r : prof_recs) {        // every stream has been joined into j.stream: all events are complete
        float sm = 0.0f;
        HIP_TRY(hipEventElapsedTime(&sm, w.prof_ev[r.e0], w.prof_ev[r.e0 + 1]));
        j.stage_ms[r.stage] += sm; j.stage_launches[r.stage] += 1;
    }
    return 0;
}

int setup_job(Job &j, mtsamd_scene *s, const mtsamd_render_desc *d, hipStream_t stream, uint64_t max_pass) {
    j.s = s; j.d = d; j.stream = stream;
    if (int rc = make_camera(*d, j.cam)) return rc;
    if (int rc = make_filter(d->rfilter, d->rfilter_param, d->rfilter_param2, d->rfilter_analytic, j.filter)) return rc;
    // One pass holds up to 2^30 camera samples (24 GiB of sample stream; a second buffer of that size lets the film splat of a pass run
    // beside the tracing of the next): every pass ends with a drain phase in which the pool empties, so fewer, larger passes waste
    // less (cbox 1024^2 @ 256 spp: 4 passes of 2^26 -> 1 pass: +7 %).
    uint64_t pass_limit = 1ull << 30;
    if (d->max_pass_log2 > 0) pass_limit = 1ull << std::min(30, std::max(10, d->max_pass_log2));
    // samples_per_pass (integrator.cpp:59-66): a pass holds at most that many samples of every pixel of the crop window -- it bounds the
    // memory of a pass and is where a timeout / cancel can stop; the image does not depend on it (per-sample RNG streams)
    if (d->samples_per_pass > 0)
        pass_limit = std::min<uint64_t>(pass_limit, std::max<uint64_t>((uint64_t) d->crop_width * d->crop_height * (uint64_t) d->samples_per_pass,
                                                                        (uint64_t) d->crop_width * (uint64_t) d->sample_count));
    // pipeline 0: one kernel with the in-kernel shadow ring (4) for LDS-resident (flat) scenes, split kernels (2) for hierarchy
    // scenes; 1 / 2 / 3 / 4 force one schedule
    if (s->spectral && d->integrator != 0) return fail(MTSAMD_ERR_UNSUPPORTED, "the direct and depth integrators are implemented for the RGB variant only");
    j.split = d->integrator == 0 && (d->pipeline == 2 || (d->pipeline == 0 && !s->view.flat));
    j.shadow_queue = d->integrator == 0 && s->view.flat && d->pipeline == 3;
    j.shadow_ring = d->integrator == 0 && s->view.flat && (d->pipeline == 4 || d->pipeline == 0);
    // scenes with a blendbsdf / mask run the fused schedule whatever was asked for: only its kernels carry the nesting code (inside the
    // kernels of the other schedules, capped at 128 VGPRs, it cost every general scene up to 20 %)
    if (s->nested_bsdfs) j.split = j.shadow_queue = j.shadow_ring = false;
    if ((d->pipeline == 3 || d->pipeline == 4) && !s->view.flat) return fail(MTSAMD_ERR_INVALID, "pipelines 3 and 4 (queued shadow rays) apply to LDS-resident scenes only");
    // Paths in flight.  A launch advances every in-flight path by one segment and ends with a tail in which the CUs run
    // dry one by one; the tails (and, for the split pipeline, the gaps between its three launches) only amortise over large
    // launches.  Measured on MI355X -- fused kernel, cbox 1024^2 @ 256 spp, scheduling waves per CU x slots per wave:
    // 16 x 256 -> 1753, 48 x 256 -> 1953, 72 x 512 -> 2366, 104 x 512 -> 2459, 208 x 1024 -> 2507 Msample/s (power-of-two
    // wave counts alias in the memory channels: 64 x 256 is slower than 72 x 256); split pipeline, 261 k-triangle mesh:
    // 16 / 64 / 128 / 208 waves per CU x 256 slots -> 705 / 1162 / 1339 / 1407 Msample/s; shadow-ring kernel (schedule 4), cbox:
    // 72 x 512 -> 2453, 104 x 512 -> 2554, 144 x 512 -> 2576, 208 x 512 -> 2629, 104 x 1024 -> 2598 Msample/s.  Split pipeline
    // after this round's traversal work (two-stream overlap included): 104 / 156 / 208 / 312 / 416 / 624 waves per CU x 256 slots
    // -> 1747 / 1890 / 1946 / 2075 / 2118 / 2142 Msample/s.
    j.target = d->paths_per_wave > 0 ? (uint32_t) d->paths_per_wave : (j.split ? 256u : 512u);
    j.target = std::min<uint32_t>(std::max<uint32_t>(j.target, 64u), 4096u);
    {   // no more scheduling waves than the pass can fill
        const uint64_t want = (std::min<uint64_t>(max_pass, pass_limit) + j.target - 1) / j.target;
        const uint64_t lo = (uint64_t) s->cu_count * 16u, hi = (uint64_t) s->cu_count * (j.split ? 416u : (j.shadow_ring ? 208u : 104u));
        j.n_waves = (uint32_t) std::min<uint64_t>(std::max<uint64_t>(want, lo), hi);
    }
    if (const char *e = exp_env("MTSAMD_WAVES_PER_CU")) j.n_waves = (uint32_t) s->cu_count * (uint32_t) std::max(1, atoi(e));    // experiment switch
    // segments hold a multiple of 64 slots: k_shade deals whole 64-path chunks of a workgroup's list to its waves.
    // The sample stream of a 2^30-sample pass is 24 GiB (twice that with the overlap buffer of multi-pass renders): when the device
    // cannot provide it -- other scenes, the caller's own tensors -- the pass is halved until the workspace fits.
    for (;;) {
        j.pass_cap = std::max<uint64_t>(std::min<uint64_t>(max_pass, pass_limit), 1);
        const int rc = ensure_workspace(s, j.n_waves, (j.target + 63u) & ~63u, j.pass_cap, j.split || j.shadow_queue);
        if (rc == MTSAMD_ERR_NOMEM && pass_limit > (1ull << 22) && max_pass > (1ull << 22)) { pass_limit = std::min(pass_limit, max_pass) >> 1; continue; }
        if (rc) return rc;
        break;
    }
    HIP_TRY(hipMemsetAsync(s->ws.wave_stats, 0, 4 * (size_t) j.n_waves * sizeof(uint64_t), stream));
    s->cancel.store(0);
    j.t_start = std::chrono::steady_clock::now();
    return 0;
}

int collect_stats(Job &j, uint64_t samples, uint64_t *stats_host) {
    if (!stats_host) return 0;
    std::vector<uint64_t> ws(4 * (size_t) j.n_waves);
    HIP_TRY(hipMemcpyAsync(ws.data(), j.s->ws.wave_stats, ws.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, j.stream));
    HIP_TRY(hipStreamSynchronize(j.stream));
    uint64_t tot[4] = { 0, 0, 0, 0 };
    for (uint32_t k = 0; k < j.n_waves; ++k) for (int q = 0; q < 4; ++q) tot[q] += ws[4 * (size_t) k + q];
    stats_host[0] = tot[0]; stats_host[1] = tot[1]; stats_host[2] = samples; stats_host[3] = j.iterations; stats_host[4] = tot[2];
    stats_host[5] = (uint64_t) (j.bounce_ms * 1e6); stats_host[6] = (uint64_t) (j.film_ms * 1e6); stats_host[7] = tot[3];
    stats_host[8] = (uint64_t) (j.stage_ms[0] * 1e6); stats_host[9] = j.stage_launches[0];
    stats_host[10] = (uint64_t) (j.stage_ms[2] * 1e6); stats_host[11] = j.stage_launches[2];
    stats_host[12] = (uint64_t) (j.stage_ms[1] * 1e6); stats_host[13] = j.stage_launches[1];
    stats_host[14] = j.passes; stats_host[15] = j.timed_out ? 1u : 0u;
    return 0;
}
} // namespace

// Film rows owned by this call (include/mtsamd.h: row_begin/row_end window, or interleaved tiles).
static int make_rows(const mtsamd_render_desc *d, RowMap &m) {
    if (d->part_count > 1) {
        if (d->part_index < 0 || d->part_index >= d->part_count || d->part_tile_rows <= 0)
            return fail(MTSAMD_ERR_INVALID, "invalid film partition (index %d of %d, %d rows per tile)", d->part_index, d->part_count, d->part_tile_rows);
        if (d->row_begin != 0 || d->row_end > 0) return fail(MTSAMD_ERR_INVALID, "row window and tile partition are mutually exclusive");
        m.row0 = 0; m.tile_rows = d->part_tile_rows; m.part = d->part_index; m.count = d->part_count;
        int32_t rows = 0;
        for (int32_t t = d->part_index; t * d->part_tile_rows < d->crop_height; t += d->part_count)
            rows += std::min(d->part_tile_rows, d->crop_height - t * d->part_tile_rows);
        m.local_rows = rows;
        return 0;
    }
    int row0 = d->row_begin, row1 = d->row_end <= 0 ? d->crop_height : d->row_end;
    if (row0 < 0 || row1 > d->crop_height || row0 > row1) return fail(MTSAMD_ERR_INVALID, "invalid row range [%d,%d)", row0, row1);
    m.row0 = row0; m.local_rows = row1 - row0; m.tile_rows = std::max(d->crop_height, 1); m.part = 0; m.count = 1;
    return 0;
}

int mtsamd_render(mtsamd_scene *s, const mtsamd_render_desc *d, float *film, uint64_t *stats_host, void *stream_) {
    if (!s || !film) return fail(MTSAMD_ERR_INVALID, "null argument");
    if (int rc = check_desc(d)) return rc;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = (hipStream_t) stream_;
    RowMap rows{};
    if (int rc = make_rows(d, rows)) return rc;
    const uint64_t per_row = (uint64_t) d->crop_width * (uint64_t) d->sample_count;
    const uint64_t total = per_row * (uint64_t) rows.local_rows;       // local sample ordinals [0, total)
    Job j;
    if (int rc = setup_job(j, s, d, stream, total)) return rc;
    j.rows = rows; j.store_xyz = d->film_rgb ? 2 : 1;
    const int R = (int) std::ceil(j.filter.radius);
    // passes hold whole local rows so that the sample stream can be stored as one plane per sample number
    if (per_row > j.pass_cap) return fail(MTSAMD_ERR_UNSUPPORTED, "one film row (%llu samples) exceeds the pass capacity", (unsigned long long) per_row);
    // Passes hold whole local rows.  The film kernel cuts a pass into source tiles of tile_h <= 16 local rows that must be contiguous
    // on the film: with a partitioned film (interleaved row tiles) tile_h divides the partition's tile height and passes start on
    // multiples of tile_h.
    uint64_t rows_per_pass = std::max<uint64_t>(1, j.pass_cap / per_row);
    int32_t film_tile_h = 16;
    if (rows.count > 1) {
        while (rows.tile_rows % film_tile_h) film_tile_h >>= 1;
        if (rows_per_pass >= (uint64_t) film_tile_h) rows_per_pass -= rows_per_pass % (uint64_t) film_tile_h;
        else { while (rows_per_pass & (rows_per_pass - 1)) rows_per_pass &= rows_per_pass - 1; film_tile_h = (int32_t) rows_per_pass; }
    }
    const bool tiled = film_tiles_supported(j.filter);
    // moment integrator: the sample stream is splatted twice (values, then squared values) into two scratch films
    float *film_target = film, *film_sq = nullptr;
    const uint64_t n_pixels = (uint64_t) d->crop_width * (uint64_t) d->crop_height;
    if (d->moment) {
        if (d->film_rgb) return fail(MTSAMD_ERR_UNSUPPORTED, "the moment integrator writes XYZ channels (film_rgb must be 0)");
        Workspace &w = s->ws;
        if (w.moment_pixels < n_pixels) {
            (void) hipFree(w.moment_film); w.moment_film = nullptr; w.moment_pixels = 0;
            if (int rc = ws_alloc((void **) &w.moment_film, 2 * 5 * n_pixels * sizeof(float))) return rc;
            w.moment_pixels = n_pixels;
        }
        HIP_TRY(hipMemsetAsync(w.moment_film, 0, 2 * 5 * n_pixels * sizeof(float), stream));
        film_target = w.moment_film; film_sq = w.moment_film + 5 * n_pixels;
    }
    // Several passes: the film splat of pass k runs on its own stream while pass k + 1 is traced into the other sample stream buffer.
    Workspace &ws = s->ws;
    const uint64_t n_passes = ((uint64_t) rows.local_rows + rows_per_pass - 1) / rows_per_pass;
    bool overlap = n_passes > 1;
    hipStream_t fstream = stream;
    if (overlap) {
        const uint64_t cap2 = std::min<uint64_t>(rows_per_pass * per_row, j.pass_cap);
        if (ws.pass_cap2 < cap2) {
            (void) hipFree(ws.out_rgba2); (void) hipFree(ws.out_pos2); ws.out_rgba2 = nullptr; ws.out_pos2 = nullptr; ws.pass_cap2 = 0;
            const int rc = ws_alloc((void **) &ws.out_rgba2, cap2 * sizeof(float4));
            const int rc2 = rc ? rc : ws_alloc((void **) &ws.out_pos2, cap2 * sizeof(float2));
            if (rc2 == MTSAMD_ERR_NOMEM) {          // no room for the second sample stream: the splat of a pass runs before the next pass
                (void) hipFree(ws.out_rgba2); (void) hipFree(ws.out_pos2); ws.out_rgba2 = nullptr; ws.out_pos2 = nullptr;
                overlap = false;
            } else if (rc2) return rc2;
            else ws.pass_cap2 = cap2;
        }
    }
    if (overlap) {
        if (!ws.film_stream) HIP_TRY(hipStreamCreateWithFlags(&ws.film_stream, hipStreamNonBlocking));
        for (auto &e : ws.film_done) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        fstream = ws.film_stream;
        // the film (and the moment scratch films) may still be written by work queued on `stream` before this call
        HIP_TRY(hipEventRecord(ws.film_done[0], stream));
        HIP_TRY(hipStreamWaitEvent(fstream, ws.film_done[0], 0));
    }
    if (tiled) {          // scratch tiles of the largest pass
        FilmParams f{};
        f.crop_w = d->crop_width; f.pass_rows = (int32_t) std::min<uint64_t>(rows_per_pass, (uint64_t) rows.local_rows); f.tile_h = film_tile_h;
        f.spp = d->sample_count;
        film_tile_grid(f);
        size_t need = film_partial_floats(f);
        const uint64_t last_rows = (uint64_t) rows.local_rows % rows_per_pass;      // a shorter last pass has fewer tiles but more sample runs
        if (last_rows) { f.pass_rows = (int32_t) last_rows; film_tile_grid(f); need = std::max(need, film_partial_floats(f)); }
        if (need > ws.film_partial_floats) {
            (void) hipFree(ws.film_partials); ws.film_partials = nullptr; ws.film_partial_floats = 0;
            if (int rc = ws_alloc((void **) &ws.film_partials, need * sizeof(float))) return rc;
            ws.film_partial_floats = need;
        }
    }
    std::vector<std::pair<hipEvent_t, hipEvent_t>> film_ev;      // timing of the splats on their stream
    int rc_loop = 0;
    uint64_t pass_index = 0;
    for (uint64_t lr0 = 0; lr0 < (uint64_t) rows.local_rows; lr0 += rows_per_pass, ++pass_index) {
        const uint64_t nrows = std::min<uint64_t>(rows_per_pass, (uint64_t) rows.local_rows - lr0);
        const uint64_t a = lr0 * per_row, n = nrows * per_row;
        j.plane_pix0 = (uint32_t) (lr0 * (uint64_t) d->crop_width);
        j.plane_pixels = 0u;          // sample stream: pixel-major
        j.buf = overlap ? (int) (pass_index & 1u) : 0;
        if (j.expired()) { j.timed_out = true; break; }
        // the buffer this pass writes was read by the splat of pass k - 2
        if (overlap && pass_index >= 2) HIP_TRY(hipStreamWaitEvent(stream, ws.film_done[j.buf], 0));
        if (int rc = trace_pass(j, a, n)) {
            if (rc > 0) break;               // timeout inside the pass: its samples are dropped
            rc_loop = rc;
            break;
        }
        j.passes += 1;
        // Film::put: splat this pass into the film rows its samples can reach (trace_pass returns when its samples are complete)
        FilmParams f{};
        f.out_rgba = j.buf ? ws.out_rgba2 : ws.out_rgba; f.out_pos = j.buf ? ws.out_pos2 : ws.out_pos; f.film = film_target; f.filter = j.filter;
        f.first_ordinal = a; f.n_samples = n; f.spp = d->sample_count; f.rows = rows;
        f.plane_pix0 = j.plane_pix0; f.plane_pixels = j.plane_pixels;
        f.crop_x = d->crop_x; f.crop_y = d->crop_y; f.crop_w = d->crop_width; f.crop_h = d->crop_height;
        const int32_t l0 = (int32_t) lr0, l1 = (int32_t) (lr0 + nrows - 1);
        int32_t g0, g1;
        if (rows.count <= 1) { g0 = rows.row0 + l0; g1 = rows.row0 + l1; }
        else {   // global rows are monotone in the local row index
            int32_t t0 = l0 / rows.tile_rows, t1 = l1 / rows.tile_rows;
            g0 = (t0 * rows.count + rows.part) * rows.tile_rows + (l0 - t0 * rows.tile_rows);
            g1 = (t1 * rows.count + rows.part) * rows.tile_rows + (l1 - t1 * rows.tile_rows);
        }
        f.row0 = std::max<int32_t>(0, g0 - R); f.row1 = std::min<int32_t>(d->crop_height, g1 + R + 1);
        if (tiled) {
            f.pass_lr0 = (int32_t) lr0; f.pass_rows = (int32_t) nrows; f.tile_h = film_tile_h;
            film_tile_grid(f);
            f.partials = ws.film_partials;
        }
        if (ws.film_ev.size() < 2 * (film_ev.size() + 1)) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            HIP_TRY(hipEventCreate(&e0)); ws.film_ev.push_back(e0);
            HIP_TRY(hipEventCreate(&e1)); ws.film_ev.push_back(e1);
        }
        film_ev.push_back({ ws.film_ev[2 * film_ev.size()], ws.film_ev[2 * film_ev.size() + 1] });
        HIP_TRY(hipEventRecord(film_ev.back().first, fstream));
        if (tiled) HIP_TRY(launch_film_tiles(f, fstream));
        else HIP_TRY(launch_film_gather(f, fstream));
        if (film_sq) {
            HIP_TRY(launch_square_stream(const_cast<float4 *>(f.out_rgba), n, fstream));
            f.film = film_sq;
            if (tiled) HIP_TRY(launch_film_tiles(f, fstream));
            else HIP_TRY(launch_film_gather(f, fstream));
        }
        HIP_TRY(hipEventRecord(film_ev.back().second, fstream));
        if (overlap) HIP_TRY(hipEventRecord(ws.film_done[j.buf], fstream));
    }
    if (overlap) {          // `stream` continues after the last splat
        HIP_TRY(hipEventRecord(ws.film_done[0], fstream));
        HIP_TRY(hipStreamWaitEvent(stream, ws.film_done[0], 0));
        HIP_TRY(hipStreamSynchronize(fstream));
    } else {
        HIP_TRY(hipStreamSynchronize(stream));
    }
    if (rc_loop) return rc_loop;
    for (auto &e : film_ev) {
        float fms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&fms, e.first, e.second));
        j.film_ms += fms;
    }
    if (film_sq) HIP_TRY(launch_moment_pack(film_target, film_sq, film, n_pixels, stream));
    if (int rc = collect_stats(j, total, stats_host)) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    return MTSAMD_OK;
}

// sensor / sampler / film part of an adjoint launch: the whole crop window of `d`, the primal film's weights, dLoss/dImage
static int fill_adjoint(mtsamd_scene *s, const mtsamd_render_desc *d, const float *dimage, const float *film, AdjointParams &a) {
    if (!s || !dimage || !film) return fail(MTSAMD_ERR_INVALID, "null argument");
    if (int rc = check_desc(d)) return rc;
    if (d->part_count > 1 || d->row_begin != 0 || d->row_end > 0) return fail(MTSAMD_ERR_UNSUPPORTED, "the adjoint pass renders the whole crop window");
    if (s->spectral) return fail(MTSAMD_ERR_UNSUPPORTED, "the adjoint pass is implemented for the RGB variant only");
    HIP_TRY(hipSetDevice(s->device));
    if (int rc = make_camera(*d, a.rp.cam)) return rc;
    if (int rc = make_filter(d->rfilter, d->rfilter_param, d->rfilter_param2, d->rfilter_analytic, a.filter)) return rc;
    if (a.filter.taps > 8) return fail(MTSAMD_ERR_UNSUPPORTED, "reconstruction filter too wide for the adjoint pass");
    a.rp.sv = s->view;
    if (a.rp.cam.aperture_radius > 0.0f) a.rp.sv.general = std::max(a.rp.sv.general, 1u);
    a.rp.base_seed = d->seed; a.rp.spp = d->sample_count;
    a.rp.crop_x = d->crop_x; a.rp.crop_y = d->crop_y; a.rp.crop_w = d->crop_width; a.rp.crop_h = d->crop_height;
    a.rp.max_depth = d->max_depth; a.rp.rr_depth = d->rr_depth;
    a.rp.rows = RowMap{ 0, d->crop_height, std::max(d->crop_height, 1), 0, 1 };
    a.rp.store_xyz = 0; a.rp.out_pos = nullptr; a.rp.out_rgba = nullptr;
    a.n_samples = (uint64_t) d->crop_width * d->crop_height * (uint64_t) d->sample_count;
    a.dimage = dimage; a.film = film;
    return 0;
}

int mtsamd_render_adjoint(mtsamd_scene *s, const mtsamd_render_desc *d, const float *dimage, const float *film, float *grad_bsdf,
                          float *grad_tex, float *grad_emitter, void *stream_) {
    AdjointParams a{};
    if (int rc = fill_adjoint(s, d, dimage, film, a)) return rc;
    if (d->max_depth < 0 || d->max_depth > 16)
        return fail(MTSAMD_ERR_UNSUPPORTED, "the adjoint pass needs a finite max_depth <= 16 (got %d)", d->max_depth);
    if (s->non_diffuse_bsdfs) return fail(MTSAMD_ERR_UNSUPPORTED, "the adjoint pass is implemented for diffuse BSDFs (one- or two-sided) only");
    if (s->environment >= 0 || s->delta_emitters) return fail(MTSAMD_ERR_UNSUPPORTED, "the adjoint pass handles area emitters only");
    if (s->bsdfs.size() > 32 && grad_bsdf) return fail(MTSAMD_ERR_UNSUPPORTED, "at most 32 BSDFs with constant-reflectance gradients");
    if (s->emitters.size() > 32 && grad_emitter) return fail(MTSAMD_ERR_UNSUPPORTED, "at most 32 emitters with radiance gradients");
    a.grad_bsdf = grad_bsdf; a.grad_tex = grad_tex; a.grad_emitter = grad_emitter;
    HIP_TRY(launch_adjoint(a, (hipStream_t) stream_));
    return MTSAMD_OK;
}

// One scalar parameter of a BSDF record: which float(s) of DevBsdf it is.  ok = false: the model has no such (differentiable) parameter.
static bool bsdf_param_fields(const DevBsdf &b, int32_t kind, int32_t comp, float DevBsdf::*&f0, float DevBsdf::*&f1) {
    static float DevBsdf::*const refl[3] = { &DevBsdf::r, &DevBsdf::g, &DevBsdf::b }, DevBsdf::*const spec[3] = { &DevBsdf::sr, &DevBsdf::sg, &DevBsdf::sb },
                 DevBsdf::*const eta[3] = { &DevBsdf::er, &DevBsdf::eg, &DevBsdf::eb }, DevBsdf::*const kk[3] = { &DevBsdf::kr, &DevBsdf::kg, &DevBsdf::kb };
    f1 = nullptr;
    if (comp < 0 || comp > 2 || b.type >= kBsdfBlend) return false;
    const bool conductor = b.type == kBsdfConductor || b.type == kBsdfRoughConductor;
    const bool dielectric = b.type == kBsdfDielectric || b.type == kBsdfRoughDielectric || b.type == kBsdfThinDielectric;
    switch (kind) {
    case MTSAMD_PARAM_REFLECTANCE:        // diffuse.reflectance, (rough)plastic.diffuse_reflectance -- constants only
        if (b.texture >= 0 || !(b.type == kBsdfDiffuse || b.type == kBsdfPlastic || b.type == kBsdfRoughPlastic)) return false;
        f0 = refl[comp]; return true;
    case MTSAMD_PARAM_SPECULAR_REFLECTANCE:
        if (b.type == kBsdfDiffuse) return false;
        f0 = spec[comp]; return true;
    case MTSAMD_PARAM_SPECULAR_TRANSMITTANCE:
        if (!dielectric) return false;
        f0 = kk[comp]; return true;
    case MTSAMD_PARAM_ETA: if (!conductor) return false; f0 = eta[comp]; return true;
    case MTSAMD_PARAM_K: if (!conductor) return false; f0 = kk[comp]; return true;
    case MTSAMD_PARAM_ALPHA:              // isotropic roughness; roughplastic's alpha also shapes its transmittance tables: not offered
        if (!(b.type == kBsdfRoughConductor || b.type == kBsdfRoughDielectric) || b.alpha_u != b.alpha_v || comp != 0) return false;
        f0 = &DevBsdf::alpha_u; f1 = &DevBsdf::alpha_v; return true;
    default: return false;
    }
}

int mtsamd_scene_set_bsdf_param(mtsamd_scene *s, uint32_t bsdf, int32_t kind, const float *value3) {
    if (!s || !value3 || bsdf >= s->bsdfs.size()) return fail(MTSAMD_ERR_INVALID, "invalid bsdf index");
    HIP_TRY(hipSetDevice(s->device));
    if (s->spectral) {       // colours become srgb spectra; eta / k must stay uniform spectra (one value); alpha is a plain number
        if (kind == MTSAMD_PARAM_REFLECTANCE || kind == MTSAMD_PARAM_SPECULAR_REFLECTANCE || kind == MTSAMD_PARAM_SPECULAR_TRANSMITTANCE) {
            float DevBsdf::*f0, DevBsdf::*f1;
            if (!bsdf_param_fields(s->bsdfs[bsdf], kind, 0, f0, f1)) return fail(MTSAMD_ERR_UNSUPPORTED, "bsdf %u (type %d) has no settable parameter of kind %d", bsdf, s->bsdfs[bsdf].type, kind);
            return spectral_set_colour(s, bsdf, kind == MTSAMD_PARAM_REFLECTANCE ? 0 : (kind == MTSAMD_PARAM_SPECULAR_REFLECTANCE ? 1 : 2), value3);
        }
        if ((kind == MTSAMD_PARAM_ETA || kind == MTSAMD_PARAM_K) && !(value3[0] == value3[1] && value3[1] == value3[2]))
            return fail(MTSAMD_ERR_UNSUPPORTED, "bsdf %u: the spectral variant needs uniform (constant) eta and k spectra", bsdf);
    }
    DevBsdf &d = s->bsdfs[bsdf];
    for (int c = 0; c < (kind == MTSAMD_PARAM_ALPHA ? 1 : 3); ++c) {
        float DevBsdf::*f0, DevBsdf::*f1;
        if (!bsdf_param_fields(d, kind, c, f0, f1)) return fail(MTSAMD_ERR_UNSUPPORTED, "bsdf %u (type %d) has no settable parameter of kind %d", bsdf, d.type, kind);
        d.*f0 = value3[c];
        if (f1) d.*f1 = value3[c];
    }
    if (!s->spectral && (d.type == kBsdfPlastic || d.type == kBsdfRoughPlastic)) {       // parameters_changed(): specular sampling weight (plastic.cpp:170-175; spectral: spectral_set_colour)
        const float d_mean = d.texture >= 0 ? s->textures[d.texture].mean : (d.r + d.g + d.b) * (1.0f / 3.0f), s_mean = (d.sr + d.sg + d.sb) * (1.0f / 3.0f);
        d.kr = s_mean / (d_mean + s_mean);
    }
    HIP_TRY(hipMemcpy(s->d_bsdfs + bsdf, &d, sizeof(DevBsdf), hipMemcpyHostToDevice));
    return MTSAMD_OK;
}

int mtsamd_render_adjoint_param(mtsamd_scene *s, const mtsamd_render_desc *d, const float *dimage, const float *film, uint32_t bsdf, int32_t kind,
                                int32_t component, float h, float *grad1, void *stream_) {
    AdjointParams a{};
    if (int rc = fill_adjoint(s, d, dimage, film, a)) return rc;
    if (!grad1 || bsdf >= s->bsdfs.size()) return fail(MTSAMD_ERR_INVALID, "invalid argument");
    if (s->nested_bsdfs) return fail(MTSAMD_ERR_UNSUPPORTED, "the parameter adjoint does not handle blendbsdf / mask materials");
    if (d->integrator != 0) return fail(MTSAMD_ERR_UNSUPPORTED, "the adjoint pass differentiates the path integrator");
    const DevBsdf &b = s->bsdfs[bsdf];
    float DevBsdf::*f0, DevBsdf::*f1;
    if (!bsdf_param_fields(b, kind, component, f0, f1)) return fail(MTSAMD_ERR_UNSUPPORTED, "bsdf %u (type %d) has no differentiable parameter of kind %d", bsdf, b.type, kind);
    const float theta = b.*f0;
    if (!(h > 0.0f)) h = 0.01f * std::max(std::fabs(theta), 0.05f);      // central difference of the model code at fixed directions
    if ((kind == MTSAMD_PARAM_ALPHA || kind == MTSAMD_PARAM_ETA) && theta - h <= 1e-4f) h = 0.5f * (theta - 1e-4f);
    if (!(h > 0.0f)) return fail(MTSAMD_ERR_INVALID, "parameter value %g leaves no room for a central difference", theta);
    a.pg_bsdf = (int32_t) bsdf; a.pg_plus = b; a.pg_minus = b;
    a.pg_plus.*f0 = theta + h; a.pg_minus.*f0 = theta - h;
    if (f1) { a.pg_plus.*f1 = theta + h; a.pg_minus.*f1 = theta - h; }
    a.pg_inv_2h = 1.0f / ((theta + h) - (theta - h));
    a.grad_param = grad1;
    a.rp.sv.general = std::max(a.rp.sv.general, 1u);
    HIP_TRY(launch_adjoint_param(a, (hipStream_t) stream_));
    return MTSAMD_OK;
}

int mtsamd_render_adjoint_envmap(mtsamd_scene *s, const mtsamd_render_desc *d, const float *dimage, const float *film, float *grad_envmap,
                                 void *stream_) {
    AdjointParams a{};
    if (int rc = fill_adjoint(s, d, dimage, film, a)) return rc;
    if (!grad_envmap) return fail(MTSAMD_ERR_INVALID, "null argument");
    if (s->environment < 0 || !s->d_envmap) return fail(MTSAMD_ERR_UNSUPPORTED, "the scene has no envmap emitter");
    if (s->nested_bsdfs) return fail(MTSAMD_ERR_UNSUPPORTED, "the envmap adjoint does not handle blendbsdf / mask materials");
    if (d->integrator != 0) return fail(MTSAMD_ERR_UNSUPPORTED, "the adjoint pass differentiates the path integrator");
    a.grad_env = grad_envmap;
    HIP_TRY(launch_adjoint_env(a, (hipStream_t) stream_));
    return MTSAMD_OK;
}

int mtsamd_scene_update_envmap(mtsamd_scene *s, const float *rgb, int32_t rebuild_distribution) {
    if (!s || !rgb) return fail(MTSAMD_ERR_INVALID, "null argument");
    if (s->environment < 0 || !s->d_envmap) return fail(MTSAMD_ERR_UNSUPPORTED, "the scene has no envmap emitter");
    if (s->spectral) return fail(MTSAMD_ERR_UNSUPPORTED, "envmap updates are implemented for the RGB variant only");
    HIP_TRY(hipSetDevice(s->device));
    EnvmapHost eh;
    if (!build_envmap(rgb, s->env_w, s->env_h, eh)) return fail(MTSAMD_ERR_INVALID, "envmap: unsupported image size");
    HIP_TRY(hipDeviceSynchronize());           // renders in flight read the old texels
    HIP_TRY(hipMemcpy(s->d_env_texels, eh.texels.data(), eh.texels.size() * sizeof(float), hipMemcpyHostToDevice));
    if (rebuild_distribution) HIP_TRY(hipMemcpy(s->d_env_warp, eh.warp.data(), eh.warp.size() * sizeof(float), hipMemcpyHostToDevice));
    return MTSAMD_OK;
}

int mtsamd_scene_texture_info(const mtsamd_scene *s, uint32_t texture, int32_t *width, int32_t *height, uint64_t *grad_offset) {
    if (!s || texture >= s->textures.size()) return fail(MTSAMD_ERR_INVALID, "invalid texture index");
    if (width) *width = s->textures[texture].w;
    if (height) *height = s->textures[texture].h;
    if (grad_offset) *grad_offset = s->textures[texture].grad_offset;
    return MTSAMD_OK;
}

int mtsamd_rgb2spec_build(const char *path, int32_t resolution, int32_t threads) {
    if (!path || resolution < 2 || resolution > 256) return fail(MTSAMD_ERR_INVALID, "invalid rgb2spec arguments");
    Rgb2Spec m;
    rgb2spec_build((uint32_t) resolution, m, threads > 0 ? (unsigned) threads : 1u);
    if (!rgb2spec_save(path, m)) return fail(MTSAMD_ERR_INVALID, "could not write '%s'", path);
    return MTSAMD_OK;
}

int mtsamd_srgb_model_fetch(const char *path, const float *rgb, float *coeff) {
    if (!path || !rgb || !coeff) return fail(MTSAMD_ERR_INVALID, "null argument");
    static thread_local std::string cached_path;
    static thread_local Rgb2Spec cached;
    if (cached_path != path) {
        if (!rgb2spec_load(path, cached)) { cached_path.clear(); return fail(MTSAMD_ERR_INVALID, "Could not load sRGB-to-spectrum upsampling model ('%s')", path); }
        cached_path = path;
    }
    srgb_model_fetch(cached, rgb, coeff);
    return MTSAMD_OK;
}

int mtsamd_cancel(mtsamd_scene *s) {
    if (!s) return fail(MTSAMD_ERR_INVALID, "null argument");
    s->cancel.store(1);
    return MTSAMD_OK;
}

int mtsamd_sample_radiance(mtsamd_scene *s, const mtsamd_render_desc *d, uint64_t first, uint64_t count, float *rgba,
                           float *pos, void *stream_) {
    if (!s || !rgba) return fail(MTSAMD_ERR_INVALID, "null argument");
    if (int rc = check_desc(d)) return rc;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = (hipStream_t) stream_;
    const uint64_t total = (uint64_t) d->crop_width * d->crop_height * (uint64_t) d->sample_count;
    if (first + count > total) return fail(MTSAMD_ERR_INVALID, "sample range exceeds W*H*sample_count");
    if (count == 0) return MTSAMD_OK;
    Job j;
    if (int rc = setup_job(j, s, d, stream, count)) return rc;
    j.rows = RowMap{ 0, d->crop_height, std::max(d->crop_height, 1), 0, 1 };
    j.store_xyz = 0;
    for (uint64_t a = 0; a < count; a += j.pass_cap) {
        uint64_t n = std::min<uint64_t>(j.pass_cap, count - a);
        if (int rc = trace_pass(j, first + a, n)) return rc > 0 ? fail(MTSAMD_ERR_INVALID, "timeout reached before every requested sample was traced") : rc;
        HIP_TRY(hipMemcpyAsync(rgba + 4 * a, s->ws.out_rgba, n * sizeof(float4), hipMemcpyDeviceToDevice, stream));
        if (pos) HIP_TRY(hipMemcpyAsync(pos + 2 * a, s->ws.out_pos, n * sizeof(float2), hipMemcpyDeviceToDevice, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return MTSAMD_OK;
}

int mtsamd_camera_sample_rays(const mtsamd_render_desc *d, uint64_t n, const float *sx, const float *sy, const float *apx, const float *apy, float *ox, float *oy,
                              float *oz, float *dx, float *dy, float *dz, float *mint, float *maxt, void *stream) {
    if (!d || !sx || !sy || !ox || !oy || !oz || !dx || !dy || !dz || !mint || !maxt) return fail(MTSAMD_ERR_INVALID, "null argument");
    CameraView cam;
    if (int rc = make_camera(*d, cam)) return rc;
    HIP_TRY(launch_camera_rays(cam, n, sx, sy, apx, apy, ox, oy, oz, dx, dy, dz, mint, maxt, (hipStream_t) stream));
    return MTSAMD_OK;
}

// ---- ImageBlock / Film ------------------------------------------------------------------------
int mtsamd_imageblock_put(int32_t width, int32_t height, int32_t offset_x, int32_t offset_y, int32_t channels, int32_t rfilter,
                          float rfilter_param, float rfilter_param2, int32_t analytic, int32_t border, uint64_t n, const float *pos, const float *values,
                          float *data, void *stream) {
    if (width <= 0 || height <= 0 || channels <= 0 || channels > 16 || !pos || !values || !data) return fail(MTSAMD_ERR_INVALID, "invalid ImageBlock arguments");
    FilterView f;
    if (int rc = make_filter(rfilter, rfilter_param, rfilter_param2, analytic, f)) return rc;
    if (border != 0 && border != f.border) return fail(MTSAMD_ERR_INVALID, "border must be 0 or the filter's border_size (%d)", f.border);
    HIP_TRY(launch_imageblock_put(f, width, height, offset_x, offset_y, channels, border, n, pos, values, data, (hipStream_t) stream));
    return MTSAMD_OK;
}

int mtsamd_imageblock_put_block(const float *src, int32_t sw, int32_t sh, int32_t sox, int32_t soy, int32_t sb, float *dst, int32_t dw,
                                int32_t dh, int32_t dox, int32_t doy, int32_t db, int32_t channels, void *stream) {
    if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || channels <= 0 || sb < 0 || db < 0)
        return fail(MTSAMD_ERR_INVALID, "invalid ImageBlock arguments");
    HIP_TRY(launch_put_block(src, sw, sh, sox, soy, sb, dst, dw, dh, dox, doy, db, channels, (hipStream_t) stream));
    return MTSAMD_OK;
}

int mtsamd_rfilter_info(int32_t rfilter, float param, float param2, float *table32, float *radius, int32_t *border) {
    FilterView f;
    if (int rc = make_filter(rfilter, param, param2, 0, f)) return rc;
    if (table32) std::memcpy(table32, f.table, sizeof(float) * 32);
    if (radius) *radius = f.radius;
    if (border) *border = f.border;
    return MTSAMD_OK;
}

int mtsamd_film_develop(const float *xyzaw, uint64_t n, float *rgba, void *stream) {
    if (!xyzaw || !rgba) return fail(MTSAMD_ERR_INVALID, "null argument");
    HIP_TRY(launch_film_develop(xyzaw, n, rgba, (hipStream_t) stream));
    return MTSAMD_OK;
}

int mtsamd_libm_eval(int32_t fn, uint64_t n, const float *x, const float *y, float *out, void *stream) {
    if (fn < 0 || fn > 9) return fail(MTSAMD_ERR_INVALID, "libm_eval: unknown function %d", fn);
    if (n && (!x || !out || (fn == 7 && !y))) return fail(MTSAMD_ERR_INVALID, "libm_eval: null buffer");
    if (n >> 40) return fail(MTSAMD_ERR_INVALID, "libm_eval: too many arguments");
    HIP_TRY(launch_libm_eval(fn, n, x, y, out, (hipStream_t) stream));
    return MTSAMD_OK;
}

} // extern "C"
