/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
 *
 * CPU restatement (plain C11) of the arithmetic building blocks of Mitsuba 2's
 * scalar_rgb path-tracing hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this code, and only as the checker.
 *
 * Parity status: the reference cannot be compiled or imported in the build
 * container (Enoki/TBB/pugixml submodules are empty), so this file is pinned by
 * the reference's own in-tree known-answer tests (see tests/test_oracle_kat.py):
 *   - sample_tea_float32 KATs      (src/libcore/tests/test_random.py:6-16)
 *   - spiral block order           (src/librender/tests/test_spiral.py:41-86)
 *   - DiscreteDistribution [1,3,2] (src/libcore/tests/test_distr_1d.py:35-103)
 *   - warp corner cases            (src/libcore/tests/test_warp.py:68-92)
 *   - diffuse eval/pdf closed form (src/bsdfs/tests/test_diffuse.py:16-38)
 *   - stairs mesh hit distances    (src/librender/tests/test_kdtrees.py:26-59)
 *   - ImageBlock splat semantics   (src/librender/tests/test_imageblock.py)
 * PCG32 and the 64-bit flavour of sample_tea_64 live in the absent Enoki
 * submodule / have no in-tree KAT: "parity unpinned" in-tree, pinned to the
 * published PCG32 demo sequence instead.
 *
 * Floating-point conventions (the reference leaves these to Enoki + compiler):
 *   - fp32 everywhere, IEEE division and sqrt, no implicit contraction
 *     (build with -ffp-contract=off); fmaf only where the reference source
 *     writes fmadd/fmsub/fnmadd or where Enoki's generic dot/cross/matrix
 *     products are defined through them;
 *   - dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   - cross(a,b) = fmsub(a.yzx, b.zxy, a.zxy*b.yzx)
 *   - vector / scalar = vector * (1/scalar)   (Enoki's array-by-scalar division)
 *   - normalize(v) = v * (1/sqrt(dot(v,v)))   (scalar rsqrt = exact 1/sqrt)
 */
#ifndef MO_MATH_H
#define MO_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>
#include "mo_libm.h"

typedef struct { float x, y, z; } mo_v3;
typedef struct { float x, y; } mo_v2;

/* include/mitsuba/core/math.h:17-38 */
#define MO_PI          3.14159265358979323846f
#define MO_INV_FOUR_PI 0.07957747154594766788f
#define MO_PI_F 3.14159265358979323846f
#define MO_INV_PI      0.31830988618379067154f
#define MO_EPSILON     (1.1920928955078125e-07f / 2.0f)
#define MO_RAY_EPSILON (MO_EPSILON * 1500.0f)
#define MO_SHADOW_EPSILON (MO_RAY_EPSILON * 10.0f)

static inline mo_v3 mo_v3_make(float x, float y, float z) { mo_v3 r = { x, y, z }; return r; }
static inline mo_v3 mo_add(mo_v3 a, mo_v3 b) { return mo_v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline mo_v3 mo_sub(mo_v3 a, mo_v3 b) { return mo_v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline mo_v3 mo_mul(mo_v3 a, mo_v3 b) { return mo_v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline mo_v3 mo_scale(mo_v3 a, float s) { return mo_v3_make(a.x * s, a.y * s, a.z * s); }
static inline mo_v3 mo_neg(mo_v3 a) { return mo_v3_make(-a.x, -a.y, -a.z); }

static inline float mo_dot(mo_v3 a, mo_v3 b) {
    return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
}
static inline mo_v3 mo_cross(mo_v3 a, mo_v3 b) {
    return mo_v3_make(fmaf(a.y, b.z, -(a.z * b.y)),
                      fmaf(a.z, b.x, -(a.x * b.z)),
                      fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float mo_rcp(float x) { return 1.0f / x; }
static inline mo_v3 mo_div_s(mo_v3 a, float s) { return mo_scale(a, mo_rcp(s)); }
static inline float mo_sqnorm(mo_v3 a) { return mo_dot(a, a); }
static inline float mo_norm(mo_v3 a) { return sqrtf(mo_sqnorm(a)); }
static inline mo_v3 mo_normalize(mo_v3 a) { return mo_scale(a, 1.0f / sqrtf(mo_sqnorm(a))); }
static inline float mo_safe_sqrt(float x) { return sqrtf(fmaxf(x, 0.0f)); }
static inline float mo_hmax_abs(mo_v3 p) { return fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z)); }

static inline float mo_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t mo_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
/* enoki mulsign(a, b): a with its sign flipped by the sign bit of b */
static inline float mo_mulsign(float a, float b) { return mo_u2f(mo_f2u(a) ^ (mo_f2u(b) & 0x80000000u)); }
static inline float mo_mulsign_neg(float a, float b) { return mo_u2f(mo_f2u(a) ^ (~mo_f2u(b) & 0x80000000u)); }

/* ------------------------------------------------------------------ */
/* include/mitsuba/core/random.h:73-138                                */
static inline uint32_t mo_tea32(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v1;
}
/* sample_tea_64 instantiated on 32-bit operands (random.h:104-115) */
static inline uint64_t mo_tea64_u32(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return (uint64_t) v0 + ((uint64_t) v1 << 32);
}
/* sample_tea_64 as IndependentSampler::seed calls it in wavefront mode
 * (src/samplers/independent.cpp:69-72): the operands are UInt64 arrays, so the
 * template's shifts/adds and `sum` are all 64-bit.  No in-tree KAT pins this
 * flavour ("parity unpinned"). */
static inline uint64_t mo_tea64_u64(uint64_t v0, uint64_t v1, int rounds) {
    uint64_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9ull;
        v0 += ((v1 << 4) + 0xa341316cull) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4ull);
        v1 += ((v0 << 4) + 0xad90777dull) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eull);
    }
    return v0 + (v1 << 32);
}
static inline float mo_tea_float32(uint32_t v0, uint32_t v1, int rounds) {
    return mo_u2f((mo_tea32(v0, v1, rounds) >> 9) | 0x3f800000u) - 1.0f;
}
static inline double mo_tea_float64(uint32_t v0, uint32_t v1, int rounds) {
    uint64_t b = (mo_tea64_u32(v0, v1, rounds) >> 12) | 0x3ff0000000000000ull;
    double d; memcpy(&d, &b, 8); return d - 1.0;
}

/* ------------------------------------------------------------------ */
/* PCG32 (enoki/random.h, absent submodule; published algorithm by M. O'Neill) */
#define MO_PCG32_DEFAULT_STATE  0x853c49e6748fea9bull
#define MO_PCG32_DEFAULT_STREAM 0xda3e39cb94b95bdbull
#define MO_PCG32_MULT           0x5851f42d4c957f2dull
typedef struct { uint64_t state, inc; } mo_pcg32;

static inline uint32_t mo_pcg32_next_u32(mo_pcg32 *r) {
    uint64_t old = r->state;
    r->state = old * MO_PCG32_MULT + r->inc;
    uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t) (old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
static inline void mo_pcg32_seed(mo_pcg32 *r, uint64_t initstate, uint64_t initseq) {
    r->state = 0;
    r->inc = (initseq << 1u) | 1u;
    mo_pcg32_next_u32(r);
    r->state += initstate;
    mo_pcg32_next_u32(r);
}
static inline float mo_pcg32_next_f32(mo_pcg32 *r) {
    return mo_u2f((mo_pcg32_next_u32(r) >> 9) | 0x3f800000u) - 1.0f;
}

/* ------------------------------------------------------------------ */
/* include/mitsuba/core/warp.h:54-90 */
static inline mo_v2 mo_square_to_uniform_disk_concentric(mo_v2 s) {
    float x = fmaf(2.0f, s.x, -1.0f), y = fmaf(2.0f, s.y, -1.0f);
    int is_zero = (x == 0.0f) && (y == 0.0f);
    int q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * MO_PI * rp / r;
    if (q13) phi = 0.5f * MO_PI - phi;
    if (is_zero) phi = 0.0f;
    float sn, cs;
    mo_lm_sincos(phi, &sn, &cs);
    mo_v2 o = { r * cs, r * sn };
    return o;
}
/* warp.h:332-341 */
static inline mo_v3 mo_square_to_cosine_hemisphere(mo_v2 s) {
    mo_v2 p = mo_square_to_uniform_disk_concentric(s);
    float sq = fmaf(p.y, p.y, p.x * p.x);      /* squared_norm of a 2-vector */
    float z = mo_safe_sqrt(1.0f - sq);
    return mo_v3_make(p.x, p.y, z);
}
static inline float mo_square_to_cosine_hemisphere_pdf(mo_v3 v) { return MO_INV_PI * v.z; }
/* warp.h:153-156 */
static inline mo_v2 mo_square_to_uniform_triangle(mo_v2 s) {
    float t = mo_safe_sqrt(1.0f - s.x);
    mo_v2 o = { 1.0f - t, t * s.y };
    return o;
}

/* include/mitsuba/core/vector.h:116-136 (Duff et al. branchless ONB) */
static inline void mo_coordinate_system(mo_v3 n, mo_v3 *s, mo_v3 *t) {
    float sign = copysignf(1.0f, n.z);
    float a = -mo_rcp(sign + n.z);
    float b = n.x * n.y * a;
    *s = mo_v3_make(mo_mulsign((n.x * n.x) * a, n.z) + 1.0f, mo_mulsign(b, n.z),
                    mo_mulsign_neg(n.x, n.z));
    *t = mo_v3_make(b, sign + (n.y * n.y) * a, -n.y);
}

/* include/mitsuba/core/frame.h:25-37 */
typedef struct { mo_v3 s, t, n; } mo_frame;
static inline mo_v3 mo_to_local(const mo_frame *f, mo_v3 v) {
    return mo_v3_make(mo_dot(v, f->s), mo_dot(v, f->t), mo_dot(v, f->n));
}
static inline mo_v3 mo_to_world(const mo_frame *f, mo_v3 v) {
    return mo_add(mo_add(mo_scale(f->s, v.x), mo_scale(f->t, v.y)), mo_scale(f->n, v.z));
}

/* enoki::morton_decode<Point2u> (absent submodule): x = even bits, y = odd bits */
static inline uint32_t mo_compact_bits(uint32_t x) {
    x &= 0x55555555u;
    x = (x ^ (x >> 1)) & 0x33333333u;
    x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu;
    x = (x ^ (x >> 8)) & 0x0000ffffu;
    return x;
}
static inline void mo_morton_decode2(uint32_t m, uint32_t *x, uint32_t *y) {
    *x = mo_compact_bits(m);
    *y = mo_compact_bits(m >> 1);
}

#endif
