// device_math.h -- fp32 vector helpers, RNG and warps for the gfx950 kernels.
//
// Arithmetic conventions (shared by every kernel; the translation units are built with
// -ffp-contract=off so nothing is fused unless written as fmaf):
//   dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//   cross(a,b) = fma(a.y,b.z, -(a.z*b.y)), ...
//   v / s      = v * (1/s);  normalize(v) = v * (1/sqrt(dot(v,v)))
// Division and sqrt are IEEE-correct (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
//
// Reference semantics followed (paths relative to the Mitsuba 2 tree):
//   include/mitsuba/core/random.h:73-138 (TEA), enoki PCG32 (published algorithm),
//   include/mitsuba/core/warp.h:54-90,153-156,332-358, core/vector.h:116-136, core/frame.h:25-37.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_libm.h"

namespace mtsamd {

struct f3 { float x, y, z; };
struct f2 { float x, y; };

#define MTS_DEV __device__ __forceinline__

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kEpsilon = 1.1920928955078125e-07f / 2.0f;
constexpr float kRayEpsilon = kEpsilon * 1500.0f;
constexpr float kShadowEpsilon = kRayEpsilon * 10.0f;

MTS_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MTS_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
MTS_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
MTS_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
MTS_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
MTS_DEV float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
MTS_DEV f3 cross(f3 a, f3 b) {
    return mk3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
MTS_DEV float rcp(float x) { return 1.0f / x; }
MTS_DEV f3 div_s(f3 a, float s) { return a * rcp(s); }
MTS_DEV float sqnorm(f3 a) { return dot(a, a); }
MTS_DEV f3 normalize(f3 a) { return a * (1.0f / sqrtf(sqnorm(a))); }
MTS_DEV float safe_sqrt(float x) { return sqrtf(fmaxf(x, 0.0f)); }
MTS_DEV float hmax_abs(f3 p) { return fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z)); }
MTS_DEV float mulsign(float a, float b) {
    return __uint_as_float(__float_as_uint(a) ^ (__float_as_uint(b) & 0x80000000u));
}
MTS_DEV float mulsign_neg(float a, float b) {
    return __uint_as_float(__float_as_uint(a) ^ (~__float_as_uint(b) & 0x80000000u));
}

// ---- TEA, 64-bit operand flavour used by IndependentSampler::seed in wavefront mode
// (src/samplers/independent.cpp:69-72 instantiates random.h:104-115 on UInt64).
MTS_DEV uint64_t tea64(uint64_t v0, uint64_t v1) {
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9ull;
        v0 += ((v1 << 4) + 0xa341316cull) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4ull);
        v1 += ((v0 << 4) + 0xad90777dull) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eull);
    }
    return v0 + (v1 << 32);
}

// ---- PCG32
struct Pcg32 { uint64_t state, inc; };
constexpr uint64_t kPcgMult = 0x5851f42d4c957f2dull;

MTS_DEV uint32_t pcg_next_u32(Pcg32 &r) {
    uint64_t old = r.state;
    r.state = old * kPcgMult + r.inc;
    uint32_t xs = (uint32_t) (((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t) (old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
MTS_DEV void pcg_seed(Pcg32 &r, uint64_t initstate, uint64_t initseq) {
    r.state = 0;
    r.inc = (initseq << 1u) | 1u;
    pcg_next_u32(r);
    r.state += initstate;
    pcg_next_u32(r);
}
MTS_DEV float pcg_next_f32(Pcg32 &r) {
    return __uint_as_float((pcg_next_u32(r) >> 9) | 0x3f800000u) - 1.0f;
}
// one PCG32 stream per global sample index (independent.cpp:62-72)
MTS_DEV void seed_sample(Pcg32 &r, uint64_t index, uint64_t base_seed) {
    uint64_t seed_value = index + base_seed;
    pcg_seed(r, tea64(seed_value, index), tea64(index, seed_value));
}

// ---- warps
MTS_DEV f2 square_to_uniform_disk_concentric(f2 s) {
    float x = fmaf(2.0f, s.x, -1.0f), y = fmaf(2.0f, s.y, -1.0f);
    bool is_zero = (x == 0.0f) && (y == 0.0f);
    bool q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * kPi * rp / r;
    if (q13) phi = 0.5f * kPi - phi;
    if (is_zero) phi = 0.0f;
    float sn, cs;
    lm_sincos(phi, &sn, &cs);
    f2 o; o.x = r * cs; o.y = r * sn;
    return o;
}
MTS_DEV f3 square_to_cosine_hemisphere(f2 s) {
    f2 p = square_to_uniform_disk_concentric(s);
    float sq = fmaf(p.y, p.y, p.x * p.x);
    return mk3(p.x, p.y, safe_sqrt(1.0f - sq));
}
MTS_DEV f2 square_to_uniform_triangle(f2 s) {
    float t = safe_sqrt(1.0f - s.x);
    f2 o; o.x = 1.0f - t; o.y = t * s.y;
    return o;
}
MTS_DEV void coordinate_system(f3 n, f3 &s, f3 &t) {
    float sign = copysignf(1.0f, n.z);
    float a = -rcp(sign + n.z);
    float b = n.x * n.y * a;
    s = mk3(mulsign((n.x * n.x) * a, n.z) + 1.0f, mulsign(b, n.z), mulsign_neg(n.x, n.z));
    t = mk3(b, sign + (n.y * n.y) * a, -n.y);
}

struct Frame { f3 s, t, n; };
MTS_DEV f3 to_local(const Frame &f, f3 v) { return mk3(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }
MTS_DEV f3 to_world(const Frame &f, f3 v) { return (f.s * v.x + f.t * v.y) + f.n * v.z; }

MTS_DEV float mis_weight(float pdf_a, float pdf_b) {
    pdf_a *= pdf_a; pdf_b *= pdf_b;
    return pdf_a > 0.0f ? pdf_a / (pdf_a + pdf_b) : 0.0f;
}

// ---- wave64 helpers
MTS_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` below this lane (prefix rank used for stream compaction)
MTS_DEV uint32_t mask_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

// ---- once-read / once-written streams (path pool, ray / hit / shadow-ray records, sample stream): marked non-temporal on hierarchy
// scenes so that they do not push the BVH nodes and triangle slots -- the only data with reuse -- out of L2 and the Infinity Cache.
// The wave-contiguous 16-byte SoA vectors are full-line accesses, the shape the hint is meant for.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));
typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t nt_u2 __attribute__((ext_vector_type(2)));
template <bool NT> MTS_DEV float4 ld_stream(const float4 *p) {
    if (!NT) return *p;
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <bool NT> MTS_DEV float2 ld_stream(const float2 *p) {
    if (!NT) return *p;
    const nt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f2 *>(p));
    return make_float2(v.x, v.y);
}
template <bool NT> MTS_DEV uint4 ld_stream(const uint4 *p) {
    if (!NT) return *p;
    const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool NT> MTS_DEV uint2 ld_stream(const uint2 *p) {
    if (!NT) return *p;
    const nt_u2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u2 *>(p));
    return make_uint2(v.x, v.y);
}
template <bool NT> MTS_DEV uint32_t ld_stream(const uint32_t *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> MTS_DEV float ld_stream(const float *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> MTS_DEV void st_stream(float4 *p, float4 v) {
    if (!NT) { *p = v; return; }
    const nt_f4 w = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(w, reinterpret_cast<nt_f4 *>(p));
}
template <bool NT> MTS_DEV void st_stream(float2 *p, float2 v) {
    if (!NT) { *p = v; return; }
    const nt_f2 w = { v.x, v.y };
    __builtin_nontemporal_store(w, reinterpret_cast<nt_f2 *>(p));
}
template <bool NT> MTS_DEV void st_stream(uint4 *p, uint4 v) {
    if (!NT) { *p = v; return; }
    const nt_u4 w = { v.x, v.y, v.z, v.w };
    __builtin_nontemporal_store(w, reinterpret_cast<nt_u4 *>(p));
}
template <bool NT> MTS_DEV void st_stream(uint2 *p, uint2 v) {
    if (!NT) { *p = v; return; }
    const nt_u2 w = { v.x, v.y };
    __builtin_nontemporal_store(w, reinterpret_cast<nt_u2 *>(p));
}
template <bool NT> MTS_DEV void st_stream(uint32_t *p, uint32_t v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> MTS_DEV void st_stream(float *p, float v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

} // namespace mtsamd
