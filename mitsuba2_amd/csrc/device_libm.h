// device_libm.h -- single-precision elementary functions of the gfx950 kernels, written out operation by operation.
//
// The reference evaluates sin / cos / exp / log / erf / ... through Enoki's polynomial kernels (Cephes-style, explicit fused
// multiply-adds); OCML on the device and libm on the host each round differently in the last place, which made per-sample radiance
// "99.9 % close" to the CPU oracle instead of equal.  These are the published Cephes single-precision algorithms (S. Moshier:
// sinf.c, tanf.c, expf.c, logf.c, asinf.c, atanf.c, ndtrf.c) with every rounding step explicit -- only +, -, *, IEEE division / sqrt,
// fmaf and integer operations -- so the oracle's C restatement of the same algorithms (oracle/mo_libm.h) produces the same bits on the
// host; tests/test_gpu_libm.py compares the two on 2^22 arguments per function through mtsamd_libm_eval.  Besides parity they are
// cheaper than the OCML entry points (no huge-argument reduction: the path's angles are bounded by a few pi).
//
// Call sites: include/mitsuba/core/warp.h:54-90 (sincos), include/mitsuba/render/microfacet.h:187-493 (exp, log, erf, tan, sincos),
// src/emitters/envmap.cpp:122-188 (atan2, acos, sincos), src/emitters/spot.cpp, src/rfilters/lanczos.cpp (sin).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mtsamd {

// host + device: the host-side filter tables (api.cpp) are built with the same functions
#define MTS_LM __host__ __device__ __forceinline__
MTS_LM float lm_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
MTS_LM uint32_t lm_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// Cody-Waite reduction to [-pi/4, pi/4]: j = even octant index, r = |x| - j * pi/4 (three-part constant, exact products for
// |x| <= 8192).  Larger arguments do not occur on the path (angles are bounded by a few pi).
MTS_LM float lm_reduce(float ax, uint32_t *j_out) {
    uint32_t j = (uint32_t) (ax * 1.27323954473516f);             // 4 / pi, truncation
    j = (j + 1u) & ~1u;
    const float y = (float) j;
    float r = __builtin_fmaf(y, -0.78515625f, ax);
    r = __builtin_fmaf(y, -2.4187564849853515625e-4f, r);
    r = __builtin_fmaf(y, -3.77489497744594108e-8f, r);
    *j_out = j;
    return r;
}
MTS_LM float lm_sin_poly(float r, float z) {            // sin r, |r| <= pi/4
    float p = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = __builtin_fmaf(p, z, -1.6666654611e-1f);
    return __builtin_fmaf(p * z, r, r);
}
MTS_LM float lm_cos_poly(float z) {                     // cos r, z = r^2
    float p = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = __builtin_fmaf(p, z, 4.166664568298827e-2f);
    return __builtin_fmaf(p * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
}
MTS_LM void lm_sincos(float x, float *s, float *c) {
    uint32_t j;
    const float r = lm_reduce(__builtin_fabsf(x), &j), z = r * r;
    const float ps = lm_sin_poly(r, z), pc = lm_cos_poly(z);
    const int swap = (j & 2u) != 0u;
    float sv = swap ? pc : ps, cv = swap ? ps : pc;
    // octant signs: sin negative for j & 4, cos negative for (j + 2) & 4; sin is odd in x
    const uint32_t sbit = ((j & 4u) << 29) ^ (lm_f2u(x) & 0x80000000u), cbit = ((j + 2u) & 4u) << 29;
    *s = lm_u2f(lm_f2u(sv) ^ sbit);
    *c = lm_u2f(lm_f2u(cv) ^ cbit);
}
MTS_LM float lm_sin(float x) { float s, c; lm_sincos(x, &s, &c); return s; }
MTS_LM float lm_cos(float x) { float s, c; lm_sincos(x, &s, &c); return c; }

MTS_LM float lm_tan(float x) {
    uint32_t j;
    const float r = lm_reduce(__builtin_fabsf(x), &j), z = r * r;
    float p = __builtin_fmaf(9.38540185543e-3f, z, 3.11992232697e-3f);
    p = __builtin_fmaf(p, z, 2.44301354525e-2f);
    p = __builtin_fmaf(p, z, 5.34112807005e-2f);
    p = __builtin_fmaf(p, z, 1.33387994085e-1f);
    p = __builtin_fmaf(p, z, 3.33331568548e-1f);
    float y = __builtin_fmaf(p * z, r, r);
    if (j & 2u) y = -1.0f / y;
    return lm_u2f(lm_f2u(y) ^ (lm_f2u(x) & 0x80000000u));
}

// exp: x = n ln2 + g, |g| <= ln2 / 2; results below 2^-126 are flushed to zero (never reached with a visible effect on the path)
MTS_LM float lm_exp(float x) {
    if (!(x <= 88.72283905206835f)) return x != x ? x : __builtin_inff();
    if (x < -87.0f) return 0.0f;
    const float n = __builtin_floorf(__builtin_fmaf(1.44269504088896341f, x, 0.5f));
    float g = __builtin_fmaf(n, -0.693359375f, x);
    g = __builtin_fmaf(n, 2.12194440e-4f, g);
    const float z = g * g;
    float p = __builtin_fmaf(1.9875691500e-4f, g, 1.3981999507e-3f);
    p = __builtin_fmaf(p, g, 8.3334519073e-3f);
    p = __builtin_fmaf(p, g, 4.1665795894e-2f);
    p = __builtin_fmaf(p, g, 1.6666665459e-1f);
    p = __builtin_fmaf(p, g, 5.0000001201e-1f);
    const float y = __builtin_fmaf(p, z, g) + 1.0f;
    // n in [-126, 128]: two exact power-of-two factors
    const int32_t ni = (int32_t) n, n1 = ni / 2, n2 = ni - n1;
    return (y * lm_u2f((uint32_t) (n1 + 127) << 23)) * lm_u2f((uint32_t) (n2 + 127) << 23);
}

// log of a positive normal number (zero -> -inf, negative / NaN -> NaN; subnormal arguments are treated as zero)
MTS_LM float lm_log(float x) {
    const uint32_t u = lm_f2u(x);
    if (u >= 0x7f800000u) return (u == 0x7f800000u) ? x : __builtin_nanf("");     // +inf, NaN, negative
    if (u < 0x00800000u) return -__builtin_inff();
    int32_t e = (int32_t) (u >> 23) - 126;                         // x = m 2^e, m in [0.5, 1)
    float m = lm_u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else m = m - 1.0f;
    const float z = m * m, fe = (float) e;
    float p = __builtin_fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    y = __builtin_fmaf(-2.12194440e-4f, fe, y);
    y = __builtin_fmaf(-0.5f, z, y);
    return __builtin_fmaf(0.693359375f, fe, m + y);
}

// erf / erfc (Cephes ndtrf.c)
MTS_LM float lm_erfc_tail(float ax) {                   // erfc(ax), ax >= 1
    const float z = lm_exp(-(ax * ax)), q = 1.0f / ax, y = q * q;
    float p;
    if (ax < 2.0f) {
        p = __builtin_fmaf(2.326819970068386e-2f, y, -1.387039388740657e-1f);
        p = __builtin_fmaf(p, y, 3.687424674597105e-1f);
        p = __builtin_fmaf(p, y, -5.824733027278666e-1f);
        p = __builtin_fmaf(p, y, 6.210004621745983e-1f);
        p = __builtin_fmaf(p, y, -4.944515323274145e-1f);
        p = __builtin_fmaf(p, y, 3.404879937665872e-1f);
        p = __builtin_fmaf(p, y, -2.741127028184656e-1f);
        p = __builtin_fmaf(p, y, 5.638259427386472e-1f);
    } else {
        p = __builtin_fmaf(-1.047766399936249e+1f, y, 1.297719955372516e+1f);
        p = __builtin_fmaf(p, y, -7.495518717768503e+0f);
        p = __builtin_fmaf(p, y, 2.921019019210786e+0f);
        p = __builtin_fmaf(p, y, -1.015265279202700e+0f);
        p = __builtin_fmaf(p, y, 4.218463358204948e-1f);
        p = __builtin_fmaf(p, y, -2.820767439740514e-1f);
        p = __builtin_fmaf(p, y, 5.641895067754075e-1f);
    }
    return (z * q) * p;
}
MTS_LM float lm_erf(float x) {
    const float ax = __builtin_fabsf(x);
    if (!(ax <= 1.0f)) {
        if (x != x) return x;
        const float r = ax > 10.0f ? 1.0f : 1.0f - lm_erfc_tail(ax);
        return lm_u2f(lm_f2u(r) | (lm_f2u(x) & 0x80000000u));
    }
    const float z = x * x;
    float p = __builtin_fmaf(7.853861353153693e-5f, z, -8.010193625184903e-4f);
    p = __builtin_fmaf(p, z, 5.188327685732524e-3f);
    p = __builtin_fmaf(p, z, -2.685381193529856e-2f);
    p = __builtin_fmaf(p, z, 1.128358514861418e-1f);
    p = __builtin_fmaf(p, z, -3.761262582423300e-1f);
    p = __builtin_fmaf(p, z, 1.128379165726710e+0f);
    return x * p;
}

// asin on [0, 0.5] (series in z = a^2)
MTS_LM float lm_asin_core(float a, float z) {
    float p = __builtin_fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z, 1.6666752422e-1f);
    return __builtin_fmaf(p * z, a, a);
}
MTS_LM float lm_acos(float x) {
    const float ax = __builtin_fabsf(x);
    if (!(ax <= 1.0f)) return __builtin_nanf("");
    if (ax > 0.5f) {
        const float z = 0.5f * (1.0f - ax), s = __builtin_sqrtf(z);
        const float t = 2.0f * lm_asin_core(s, z);              // acos(|x|)
        return x < 0.0f ? 3.14159265358979323846f - t : t;
    }
    return 1.5707963267948966192f - lm_asin_core(x, x * x);
}
MTS_LM float lm_atan_pos(float t) {                     // atan t, t >= 0
    float y0, a;
    if (t > 2.414213562373095f) { y0 = 1.5707963267948966192f; a = -1.0f / t; }
    else if (t > 0.4142135623730950f) { y0 = 0.78539816339744830962f; a = (t - 1.0f) / (t + 1.0f); }
    else { y0 = 0.0f; a = t; }
    const float z = a * a;
    float p = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    return y0 + __builtin_fmaf(p * z, a, a);
}
// atan2 with the usual quadrant conventions (y = x = 0 -> 0 with the sign of y)
MTS_LM float lm_atan2(float y, float x) {
    const float ay = __builtin_fabsf(y), ax = __builtin_fabsf(x);
    float r;
    if (ax == 0.0f && ay == 0.0f) r = (lm_f2u(x) >> 31) ? 3.14159265358979323846f : 0.0f;
    else {
        r = (ay == __builtin_inff() && ax == __builtin_inff()) ? 0.78539816339744830962f : lm_atan_pos(ay / ax);   // ay / 0 = inf -> pi/2
        if (lm_f2u(x) >> 31) r = 3.14159265358979323846f - r;
    }
    return lm_u2f(lm_f2u(r) | (lm_f2u(y) & 0x80000000u));
}

// atanh on (-1, 1) (Cephes atanhf.c) and cosh (coshf.c): sample_rgb_spectrum / its pdf, include/mitsuba/core/spectrum.h:270-314
MTS_LM float lm_atanh(float x) {
    const float ax = __builtin_fabsf(x);
    if (ax < 0.5f) {
        const float z = x * x;
        float p = __builtin_fmaf(1.81740078349e-1f, z, 8.24370301058e-2f);
        p = __builtin_fmaf(p, z, 1.46691431730e-1f);
        p = __builtin_fmaf(p, z, 1.99782164500e-1f);
        p = __builtin_fmaf(p, z, 3.33337300303e-1f);
        return __builtin_fmaf(p * z, x, x);
    }
    if (!(ax < 1.0f)) return ax == 1.0f ? __builtin_copysignf(__builtin_inff(), x) : __builtin_nanf("");
    return 0.5f * lm_log((1.0f + x) / (1.0f - x));
}
MTS_LM float lm_cosh(float x) {
    const float e = lm_exp(__builtin_fabsf(x));
    return __builtin_fmaf(0.5f, e, 0.5f / e);
}

} // namespace mtsamd
