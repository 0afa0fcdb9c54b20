/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_math.h header for scope and pinning).
 *
 * Spectral variant, restated from:
 *   sample_wavelength / sample_rgb_spectrum   include/mitsuba/core/spectrum.h:270-314
 *   math::sample_shifted                      include/mitsuba/core/math.h:418-442
 *   cie1931_xyz / spectrum_to_xyz             include/mitsuba/core/spectrum.h:127-217
 *   srgb_model_eval                           include/mitsuba/render/srgb.h:8-24
 *   srgb_model_fetch                          src/librender/srgb.cpp:14-40
 *   rgb2spec_fetch                            ext/rgb2spec/rgb2spec.c:81-121
 *   SRGBReflectanceSpectrum / SRGBEmitterSpectrum   src/spectra/srgb.cpp:27-52, src/spectra/srgb_d65.cpp:27-63
 *   D65Spectrum / RegularSpectrum             src/spectra/d65.cpp:44-66, src/spectra/regular.cpp:68-75
 *   ContinuousDistribution::eval_pdf          include/mitsuba/core/distr_1d.h:378-394
 * The coefficient table is data (generated; "data/srgb.coeff" is a build artefact of the reference and absent from
 * its tree); tests/ compare the generator against the reference's own tool compiled from ext/rgb2spec.
 */
#include "mo_internal.h"
#include "mo_cie_data.h"
#include <stdio.h>
#include <stdlib.h>

/* spectrum.h:287-291 + math.h:418-425 */
void mo_sample_wavelengths(float sample, float wav[MO_WAV], float weight[MO_WAV]) {
    for (int k = 0; k < MO_WAV; ++k) {
        float v = sample + (float) k / (float) MO_WAV;
        if (v > 1.0f) v -= 1.0f;
        float l = 538.0f - mo_lm_atanh(0.8569106254698279f - 1.8275019724092267f * v) * 138.88888888888889f;
        float t = mo_lm_cosh(0.0072f * (l - 538.0f));
        wav[k] = l; weight[k] = 253.82f * t * t;
    }
}

/* srgb.h:8-24 */
float mo_srgb_model_eval(const float c[3], float l) {
    float v = fmaf(fmaf(c[0], l, c[1]), l, c[2]);
    if (isinf(c[2])) return fmaf(copysignf(1.0f, c[2]), 0.5f, 0.5f);
    return fmaxf(0.0f, fmaf(0.5f * v, 1.0f / sqrtf(fmaf(v, v, 1.0f)), 0.5f));
}

/* srgb_model_mean (srgb.h:25-35): mean of the model over 16 equidistant wavelengths in [360, 830] */
float mo_srgb_model_mean(const float c[3]) {
    float sum = 0.0f;
    for (int i = 0; i < 16; ++i) {
        float l = 360.0f + (float) i * ((830.0f - 360.0f) / 15.0f);
        sum += mo_srgb_model_eval(c, l);
    }
    return sum * (1.0f / 16.0f);
}

/* D65 table scaled by `scale` (d65.cpp:58-61) evaluated like RegularSpectrum (distr_1d.h:378-394) */
float mo_d65_eval(float scale, float l) {
    if (!(l >= 360.0f && l <= 830.0f)) return 0.0f;
    float x = (l - 360.0f) * (float) (1.0 / (470.0 / 94.0));
    uint32_t i = (uint32_t) x;
    if (i > 93u) i = 93u;
    float y0 = (float) mo_cie_d65[i] * scale, y1 = (float) mo_cie_d65[i + 1] * scale;
    float w1 = x - (float) i, w0 = 1.0f - w1;
    return fmaf(w0, y0, w1 * y1);
}

/* spectrum.h:145-217 */
void mo_spectrum_to_xyz(const float value[MO_WAV], const float wav[MO_WAV], float xyz[3]) {
    float X[MO_WAV], Y[MO_WAV], Z[MO_WAV];
    for (int k = 0; k < MO_WAV; ++k) {
        float l = wav[k];
        float t = (l - 360.0f) * ((95 - 1) / (830.0f - 360.0f));
        int active = l >= 360.0f && l <= 830.0f;
        int i0 = (int) t;
        if (i0 < 0) i0 = 0;
        if (i0 > 93) i0 = 93;
        float w1 = t - (float) i0, w0 = 1.0f - w1;
        X[k] = active ? fmaf(w0, (float) mo_cie_x[i0], w1 * (float) mo_cie_x[i0 + 1]) : 0.0f;
        Y[k] = active ? fmaf(w0, (float) mo_cie_y[i0], w1 * (float) mo_cie_y[i0 + 1]) : 0.0f;
        Z[k] = active ? fmaf(w0, (float) mo_cie_z[i0], w1 * (float) mo_cie_z[i0 + 1]) : 0.0f;
    }
    xyz[0] = (((X[0] * value[0]) + (X[1] * value[1])) + ((X[2] * value[2]) + (X[3] * value[3]))) * 0.25f;
    xyz[1] = (((Y[0] * value[0]) + (Y[1] * value[1])) + ((Y[2] * value[2]) + (Y[3] * value[3]))) * 0.25f;
    xyz[2] = (((Z[0] * value[0]) + (Z[1] * value[1])) + ((Z[2] * value[2]) + (Z[3] * value[3]))) * 0.25f;
}

/* ---- coefficient table ("SPEC" file: u32 res, res floats scale, 3*res^3*3 floats) ---- */
typedef struct { uint32_t res; float *scale, *data; } coeff_table;

static int table_load(const char *path, coeff_table *t) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    char hdr[4];
    if (fread(hdr, 4, 1, f) != 1 || memcmp(hdr, "SPEC", 4) != 0 || fread(&t->res, 4, 1, f) != 1) { fclose(f); return -1; }
    size_t n = (size_t) 9 * t->res * t->res * t->res;
    t->scale = (float *) malloc(sizeof(float) * t->res);
    t->data = (float *) malloc(sizeof(float) * n);
    int ok = fread(t->scale, sizeof(float) * t->res, 1, f) == 1 && fread(t->data, sizeof(float) * n, 1, f) == 1;
    fclose(f);
    return ok ? 0 : -1;
}

/* rgb2spec.c:81-121 + srgb.cpp:29-39 */
static void model_fetch(const coeff_table *t, const float c[3], float out[3]) {
    if (c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f) { out[0] = out[1] = 0.0f; out[2] = -INFINITY; return; }
    if (c[0] == 1.0f && c[1] == 1.0f && c[2] == 1.0f) { out[0] = out[1] = 0.0f; out[2] = INFINITY; return; }
    int res = (int) t->res;
    float rgb[3];
    for (int j = 0; j < 3; ++j) rgb[j] = fmaxf(fminf(c[j], 1.0f), 0.0f);
    int i = 0;
    for (int j = 1; j < 3; ++j) if (rgb[j] >= rgb[i]) i = j;
    float z = rgb[i], sc = (float) (res - 1) / z, x = rgb[(i + 1) % 3] * sc, y = rgb[(i + 2) % 3] * sc;
    uint32_t xi = (uint32_t) x, yi = (uint32_t) y;
    if (xi > (uint32_t) (res - 2)) xi = (uint32_t) (res - 2);
    if (yi > (uint32_t) (res - 2)) yi = (uint32_t) (res - 2);
    uint32_t zi = 0;                     /* last interval whose left end is <= z */
    for (uint32_t k = 1; k + 1 < (uint32_t) res; ++k) if (t->scale[k] <= z) zi = k;
    size_t off = ((((size_t) i * res + zi) * res + yi) * res + xi) * 3, dx = 3, dy = 3 * (size_t) res, dz = 3 * (size_t) res * res;
    float x1 = x - (float) xi, x0 = 1.0f - x1, y1 = y - (float) yi, y0 = 1.0f - y1;
    float z1 = (z - t->scale[zi]) / (t->scale[zi + 1] - t->scale[zi]), z0 = 1.0f - z1;
    const float *d = t->data;
    for (int j = 0; j < 3; ++j, ++off)
        out[j] = ((d[off] * x0 + d[off + dx] * x1) * y0 + (d[off + dy] * x0 + d[off + dy + dx] * x1) * y1) * z0 +
                 ((d[off + dz] * x0 + d[off + dz + dx] * x1) * y0 + (d[off + dz + dy] * x0 + d[off + dz + dy + dx] * x1) * y1) * z1;
}

int mo_scene_set_spectral(mo_scene *s, const char *coeff_path) {
    coeff_table t = { 0, NULL, NULL };
    if (!s || table_load(coeff_path, &t)) return -1;
    /* textures: bitmap texels become model coefficients (bitmap.cpp:116-123), checkerboard colours are `srgb` spectra with the
     * constructor's range check (srgb.cpp:34-35); Texture::mean() = mean of srgb_model_mean (bitmap.cpp:120, srgb.cpp:54) */
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        mo_texture *tx = &s->textures[i];
        if (tx->kind == 1) {
            for (int k = 0; k < 3; ++k)
                if (tx->color0[k] < 0.0f || tx->color0[k] > 1.0f || tx->color1[k] < 0.0f || tx->color1[k] > 1.0f) { free(t.scale); free(t.data); return -3; }
            model_fetch(&t, tx->color0, tx->coeff0);
            model_fetch(&t, tx->color1, tx->coeff1);
            tx->mean = 0.5f * (mo_srgb_model_mean(tx->coeff0) + mo_srgb_model_mean(tx->coeff1));
        } else {
            double mean = 0.0;
            for (size_t p = 0; p < (size_t) tx->w * tx->h; ++p) {
                float rgb[3] = { tx->data[3 * p], tx->data[3 * p + 1], tx->data[3 * p + 2] };
                model_fetch(&t, rgb, tx->data + 3 * p);
                mean += (double) mo_srgb_model_mean(tx->data + 3 * p);
            }
            tx->mean = (float) (mean / (double) ((size_t) tx->w * tx->h));
        }
    }
    for (uint32_t i = 0; i < s->n_meshes; ++i)
      for (int nest_k = 0; nest_k < 2; ++nest_k) {
        mo_mesh *m = &s->meshes[i];
        mo_bsdf *b = &m->bsdf;
        if (b->nest) {
            /* blendbsdf / mask: the weight is a scalar (a bitmap without raw = true throws in eval_1, bitmap.cpp:218-222);
             * the children upsample their own constant colours */
            if (m->texture >= 0 || b->child_tex[0] >= 0 || b->child_tex[1] >= 0) { free(t.scale); free(t.data); return -5; }      /* textured children: RGB variant only */
            b = b->child[nest_k];
            if (!b) continue;
        } else if (nest_k) continue;
        const int is_child = m->bsdf.nest != 0;
        /* every colour-valued parameter is either `uniform` (a constant) or an `srgb` texture, whose constructor rejects
         * values outside [0, 1] (srgb.cpp:34-35) and fetches the model coefficients */
        const float *vals[3] = { is_child ? b->d.reflectance : m->refl, b->d.specular_reflectance, b->d.specular_transmittance };
        float *coeffs[3] = { b->refl_coeff, b->spec_coeff, b->trans_coeff };
        float means[3] = { 0.0f, 0.0f, 0.0f };
        for (int p = 0; p < 3; ++p) {
            if (p == 0 && !is_child && m->texture >= 0) { means[0] = s->textures[m->texture].mean; continue; }
            if (b->d.uniform_mask & (1 << p)) { means[p] = vals[p][0]; continue; }
            for (int k = 0; k < 3; ++k) if (vals[p][k] < 0.0f || vals[p][k] > 1.0f) { free(t.scale); free(t.data); return -3; }
            model_fetch(&t, vals[p], coeffs[p]);
            means[p] = mo_srgb_model_mean(coeffs[p]);
        }
        if (!is_child) for (int k = 0; k < 3; ++k) m->refl_coeff[k] = b->refl_coeff[k];
        if ((b->d.type == MO_BSDF_CONDUCTOR || b->d.type == MO_BSDF_ROUGHCONDUCTOR) &&
            (b->d.eta[0] != b->d.eta[1] || b->d.eta[0] != b->d.eta[2] || b->d.k[0] != b->d.k[1] || b->d.k[0] != b->d.k[2])) {
            free(t.scale); free(t.data); return -4;          /* RGB eta / k cannot be upsampled (values > 1): uniform spectra only */
        }
        if (b->d.type == MO_BSDF_PLASTIC || b->d.type == MO_BSDF_ROUGHPLASTIC) b->spec_weight = means[1] / (means[0] + means[1]);     /* plastic.cpp:170-175 with Texture::mean() */
    }
    for (uint32_t e = 0; e < s->n_emitters; ++e) {
        mo_emitter *em = &s->emitters[e];
        float color[3] = { em->radiance[0], em->radiance[1], em->radiance[2] };
        float scale = fmaxf(fmaxf(color[0], color[1]), color[2]) * 2.0f;            /* srgb_d65.cpp:36-40 */
        if (scale != 0.0f) { float r = 1.0f / scale; for (int k = 0; k < 3; ++k) color[k] *= r; }
        model_fetch(&t, color, em->coeff);
        float m_scale = 1.0f * scale;
        m_scale *= 1.0f / 10568.0f;                                                  /* d65.cpp:48-49 */
        em->d65_scale = m_scale;
        if (em->type == 2) {
            /* envmap.cpp:96-109: every texel becomes (model coefficients of the colour scaled to a 50% maximum, scale);
             * the sampling hierarchy stays the one built from the RGB luminance in the constructor (envmap.cpp:92-93,111) */
            mo_envmap *env = em->env;
            for (size_t i = 0; i < (size_t) env->w * env->h; ++i) {
                float *px = env->data + 4 * i;
                float sc = fmaxf(fmaxf(px[0], px[1]), px[2]) * 2.0f, dn = fmaxf(1e-8f, sc);
                float rgb_norm[3] = { px[0] / dn, px[1] / dn, px[2] / dn };
                model_fetch(&t, rgb_norm, px);                 /* black: (0, 0, -inf), evaluates to 0 (srgb.cpp:31-33) */
                px[3] = sc;
            }
        }
    }
    s->spectral = 1;
    free(t.scale); free(t.data);
    return 0;
}

void mo_kat_srgb_model_fetch(const char *coeff_path, const float *rgb3, float *coeff3) {
    coeff_table t = { 0, NULL, NULL };
    coeff3[0] = coeff3[1] = coeff3[2] = NAN;
    if (table_load(coeff_path, &t)) return;
    model_fetch(&t, rgb3, coeff3);
    free(t.scale); free(t.data);
}

void mo_kat_spectral(float sample, const float *coeff3, float d65_scale, float *out) {
    float wav[MO_WAV], weight[MO_WAV], refl[MO_WAV];
    mo_sample_wavelengths(sample, wav, weight);
    for (int k = 0; k < MO_WAV; ++k) {
        out[k] = wav[k]; out[4 + k] = weight[k];
        refl[k] = mo_srgb_model_eval(coeff3, wav[k]); out[8 + k] = refl[k];
        out[12 + k] = mo_d65_eval(d65_scale, wav[k]);
    }
    mo_spectrum_to_xyz(refl, wav, out + 16);
}
