// device_spectral.h -- spectral variant (4 wavelengths per camera sample) of the light-transport stages.
//
// Reference semantics followed (paths relative to the Mitsuba 2 tree):
//   sample_wavelength / sample_rgb_spectrum     include/mitsuba/core/spectrum.h:270-314
//   math::sample_shifted                        include/mitsuba/core/math.h:418-442
//   cie1931_xyz / spectrum_to_xyz               include/mitsuba/core/spectrum.h:127-217
//   srgb_model_eval                             include/mitsuba/render/srgb.h:8-24
//   SRGBReflectanceSpectrum / SRGBEmitterSpectrum  src/spectra/srgb.cpp:27-52, src/spectra/srgb_d65.cpp:27-63
//   D65Spectrum -> RegularSpectrum              src/spectra/d65.cpp:44-66, src/spectra/regular.cpp:68-75
//   ContinuousDistribution::eval_pdf            include/mitsuba/core/distr_1d.h:378-394
#pragma once
#include "device_math.h"

namespace mtsamd {

constexpr int kWav = 4;                         // MTS_WAVELENGTH_SAMPLES of the spectral variants (mitsuba.conf.template:135-138)
constexpr float kCieMin = 360.0f, kCieMax = 830.0f;

// 95-sample tables at 5 nm: CIE 1931 x, y, z and CIE D65 (filled once by upload_spectral_tables())
struct SpectralTables { float x[95], y[95], z[95], d65[95]; };
extern __device__ SpectralTables g_spectral;

struct Spec4 { float v[kWav]; };

// sample_wavelength (spectrum.h:270-314): shifted samples (math.h:418-442) + the "importance spectrum" of Radziszewski et al.  The 1 / pdf
// weights are applied when the path ends (wavelength_weight in store_result_spectral).
// the four wavelengths of a path as a function of its one wavelength sample: this is what the path pool stores (4 B instead of the
// 16 B of the wavelengths themselves); recomputed when a path record is loaded -- the same operations on the same number, so the bits
// are those of sample_wavelengths
MTS_DEV void wavelengths_from_sample(float sample, Spec4 &wav) {
#pragma unroll
    for (int k = 0; k < kWav; ++k) {
        float v = sample + (float) k / (float) kWav;
        if (v > 1.0f) v -= 1.0f;
        wav.v[k] = 538.0f - lm_atanh(0.8569106254698279f - 1.8275019724092267f * v) * 138.88888888888889f;
    }
}
MTS_DEV float wavelength_weight(float l) {
    float t = lm_cosh(0.0072f * (l - 538.0f));
    return 253.82f * t * t;
}

MTS_DEV float srgb_model_eval(float c0, float c1, float c2, float l) {
    float v = fmaf(fmaf(c0, l, c1), l, c2);
    if (isinf(c2)) return fmaf(copysignf(1.0f, c2), 0.5f, 0.5f);
    return fmaxf(0.0f, fmaf(0.5f * v, 1.0f / sqrtf(fmaf(v, v, 1.0f)), 0.5f));
}

// linear interpolation in a 95-entry table over [360, 830] nm, 0 outside (values optionally pre-scaled)
MTS_DEV float table_eval(const float *tbl, float scale, float l) {
    if (!(l >= kCieMin && l <= kCieMax)) return 0.0f;
    float x = (l - kCieMin) * 0.2f;               // m_inv_interval_size = float(1 / (470 / 94))
    uint32_t i = min((uint32_t) x, 93u);
    float y0 = tbl[i] * scale, y1 = tbl[i + 1] * scale;
    float w1 = x - (float) i, w0 = 1.0f - w1;
    return fmaf(w0, y0, w1 * y1);
}

MTS_DEV f3 spectrum_to_xyz(const Spec4 &value, const Spec4 &wav) {
    float X[kWav], Y[kWav], Z[kWav];
#pragma unroll
    for (int k = 0; k < kWav; ++k) {
        float l = wav.v[k];
        float t = (l - kCieMin) * ((95 - 1) / (kCieMax - kCieMin));
        bool active = l >= kCieMin && l <= kCieMax;
        int i0 = min(max((int) t, 0), 93);
        float w1 = t - (float) i0, w0 = 1.0f - w1;
        X[k] = active ? fmaf(w0, g_spectral.x[i0], w1 * g_spectral.x[i0 + 1]) : 0.0f;
        Y[k] = active ? fmaf(w0, g_spectral.y[i0], w1 * g_spectral.y[i0 + 1]) : 0.0f;
        Z[k] = active ? fmaf(w0, g_spectral.z[i0], w1 * g_spectral.z[i0 + 1]) : 0.0f;
    }
    // hmean over the 4 wavelengths
    return mk3((((X[0] * value.v[0]) + (X[1] * value.v[1])) + ((X[2] * value.v[2]) + (X[3] * value.v[3]))) * 0.25f,
               (((Y[0] * value.v[0]) + (Y[1] * value.v[1])) + ((Y[2] * value.v[2]) + (Y[3] * value.v[3]))) * 0.25f,
               (((Z[0] * value.v[0]) + (Z[1] * value.v[1])) + ((Z[2] * value.v[2]) + (Z[3] * value.v[3]))) * 0.25f);
}

} // namespace mtsamd
