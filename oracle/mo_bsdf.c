/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_math.h header for scope and pinning).
 *
 * BSDF models beyond `diffuse`, restated from the reference:
 *   fresnel / fresnel_conductor / fresnel_diffuse_reflectance   include/mitsuba/render/fresnel.h:34-130,331-358
 *   MicrofacetDistribution (Beckmann, GGX; visible-normal sampling) include/mitsuba/render/microfacet.h:187-440
 *   SmoothConductor        src/bsdfs/conductor.cpp:185-275
 *   RoughConductor         src/bsdfs/roughconductor.cpp:196-391
 *   SmoothDielectric       src/bsdfs/dielectric.cpp:201-318
 *   SmoothPlastic          src/bsdfs/plastic.cpp:162-297
 *   TwoSidedBRDF           src/bsdfs/twosided.cpp:94-175 (one nested BSDF used for both sides)
 * Pinned by the reference's own vectors: test_fresnel.py:7-80, test_microfacet.py:18-300 (eval / pdf / smith_g1 / sample
 * tables for both distributions), test_dielectric.py, test_conductor.py, test_twosided.py (tests/test_bsdf_cpu.py).
 * Parity unpinned: erf / erfinv of Beckmann visible-normal sampling come from the absent Enoki submodule; libm erff and
 * Giles' single-precision erfinv polynomial stand in (sampling densities are unaffected, sample positions may differ by ulps).
 */
#include "mo_internal.h"

#define MO_PI 3.14159265358979323846f
#define MO_INV_SQRT_PI 0.56418958354775628695f
#define MO_EPS (1.1920929e-07f * 0.5f)

static inline float sqr(float x) { return x * x; }
static inline float fnmadd(float a, float b, float c) { return fmaf(-a, b, c); }
static inline float fmsub(float a, float b, float c) { return fmaf(a, b, -c); }

/* ------------------------------------------------------------------ Fresnel */
/* fresnel.h:34-77 -> F, cos_theta_t, eta_it, eta_ti */
void mo_fresnel(float cos_theta_i, float eta, float out[4]) {
    int outside = cos_theta_i >= 0.0f;
    float rcp_eta = mo_rcp(eta), eta_it = outside ? eta : rcp_eta, eta_ti = outside ? rcp_eta : eta;
    float cos_theta_t_sqr = fnmadd(fnmadd(cos_theta_i, cos_theta_i, 1.0f), eta_ti * eta_ti, 1.0f);
    float ci = fabsf(cos_theta_i), ct = mo_safe_sqrt(cos_theta_t_sqr);
    int index_matched = eta == 1.0f, special = index_matched || ci == 0.0f;
    float r_sc = index_matched ? 0.0f : 1.0f;
    float a_s = fnmadd(eta_it, ct, ci) / fmaf(eta_it, ct, ci);
    float a_p = fnmadd(eta_it, ci, ct) / fmaf(eta_it, ci, ct);
    float r = 0.5f * (sqr(a_s) + sqr(a_p));
    if (special) r = r_sc;
    out[0] = r; out[1] = mo_mulsign_neg(ct, cos_theta_i); out[2] = eta_it; out[3] = eta_ti;
}

/* fresnel.h:96-122 */
float mo_fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    float c2 = cos_theta_i * cos_theta_i, s2 = 1.0f - c2, s4 = s2 * s2;
    float temp_1 = eta_r * eta_r - eta_i * eta_i - s2;
    float a_2_pb_2 = mo_safe_sqrt(temp_1 * temp_1 + 4.0f * eta_i * eta_i * eta_r * eta_r);
    float a = mo_safe_sqrt(0.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + c2, term_2 = 2.0f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * c2 + s4, term_4 = term_2 * s2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}

/* fresnel.h:331-358 */
float mo_fresnel_diffuse_reflectance(float eta) {
    if (eta < 1.0f)
        return -1.4399f * (eta * eta) + 0.7099f * eta + 0.6681f + 0.0636f / eta;
    float i1 = mo_rcp(eta), i2 = i1 * i1, i3 = i2 * i1, i4 = i3 * i1, i5 = i4 * i1;
    return 0.919317f - 3.4793f * i1 + 6.75335f * i2 - 7.80989f * i3 + 4.98554f * i4 - 1.36881f * i5;
}

static inline mo_v3 reflect_z(mo_v3 wi) { return mo_v3_make(-wi.x, -wi.y, wi.z); }
static inline mo_v3 reflect_m(mo_v3 wi, mo_v3 m) {          /* fresnel.h:286-288: fmsub(m, 2 dot(wi, m), wi) */
    float k = 2.0f * mo_dot(wi, m);
    return mo_v3_make(fmsub(m.x, k, wi.x), fmsub(m.y, k, wi.y), fmsub(m.z, k, wi.z));
}
static inline mo_v3 refract_z(mo_v3 wi, float cos_theta_t, float eta_ti) {
    return mo_v3_make(-eta_ti * wi.x, -eta_ti * wi.y, cos_theta_t);
}

/* ------------------------------------------------------------------ microfacet distribution */
typedef struct { int ggx; float au, av; int visible; } mdf;

static mdf mdf_make(int ggx, float au, float av, int visible) {
    mdf d = { ggx, fmaxf(au, 1e-4f), fmaxf(av, 1e-4f), visible };      /* configure(): microfacet.h:443-446 */
    return d;
}

/* microfacet.h:187-206 */
static float mdf_eval(const mdf *d, mo_v3 m) {
    float alpha_uv = d->au * d->av, cos_theta = m.z, cos_theta_2 = sqr(cos_theta), result;
    if (!d->ggx)
        result = mo_lm_exp(-(sqr(m.x / d->au) + sqr(m.y / d->av)) / cos_theta_2) / (MO_PI * alpha_uv * sqr(cos_theta_2));
    else
        result = mo_rcp(MO_PI * alpha_uv * sqr(sqr(m.x / d->au) + sqr(m.y / d->av) + sqr(m.z)));
    return result * cos_theta > 1e-20f ? result : 0.0f;
}

/* microfacet.h:343-369 */
static float mdf_smith_g1(const mdf *d, mo_v3 v, mo_v3 m) {
    float xy_alpha_2 = sqr(d->au * v.x) + sqr(d->av * v.y), tan_theta_alpha_2 = xy_alpha_2 / sqr(v.z), result;
    if (!d->ggx) {
        float a = 1.0f / sqrtf(tan_theta_alpha_2), a_sqr = sqr(a);
        result = a >= 1.6f ? 1.0f : (3.535f * a + 2.181f * a_sqr) / (1.0f + 2.276f * a + 2.577f * a_sqr);
    } else {
        result = 2.0f / (1.0f + sqrtf(1.0f + tan_theta_alpha_2));
    }
    if (xy_alpha_2 == 0.0f) result = 1.0f;
    if (mo_dot(v, m) * v.z <= 0.0f) result = 0.0f;
    return result;
}
static float mdf_G(const mdf *d, mo_v3 wi, mo_v3 wo, mo_v3 m) { return mdf_smith_g1(d, wi, m) * mdf_smith_g1(d, wo, m); }

/* microfacet.h:219-228 */
static float mdf_pdf(const mdf *d, mo_v3 wi, mo_v3 m) {
    float result = mdf_eval(d, m);
    if (d->visible) result *= mdf_smith_g1(d, wi, m) * fabsf(mo_dot(wi, m)) / wi.z;
    else result *= m.z;
    return result;
}

/* Giles, "Approximating the erfinv function" (single precision branch) */
static float erfinv_f(float x) {
    float w = -mo_lm_log((1.0f - x) * (1.0f + x)), p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f; p = fmaf(p, w, 3.43273939e-07f); p = fmaf(p, w, -3.5233877e-06f);
        p = fmaf(p, w, -4.39150654e-06f); p = fmaf(p, w, 0.00021858087f); p = fmaf(p, w, -0.00125372503f);
        p = fmaf(p, w, -0.00417768164f); p = fmaf(p, w, 0.246640727f); p = fmaf(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f; p = fmaf(p, w, 0.000100950558f); p = fmaf(p, w, 0.00134934322f);
        p = fmaf(p, w, -0.00367342844f); p = fmaf(p, w, 0.00573950773f); p = fmaf(p, w, -0.0076224613f);
        p = fmaf(p, w, 0.00943887047f); p = fmaf(p, w, 1.00167406f); p = fmaf(p, w, 2.83297682f);
    }
    return p * x;
}

/* microfacet.h:372-424: slopes of the visible normals for alpha = 1 */
static mo_v2 mdf_sample_visible_11(const mdf *d, float cos_theta_i, mo_v2 sample) {
    mo_v2 r;
    if (!d->ggx) {
        float tan_theta_i = mo_safe_sqrt(fnmadd(cos_theta_i, cos_theta_i, 1.0f)) / cos_theta_i;
        float cot_theta_i = mo_rcp(tan_theta_i);
        float maxval = mo_lm_erf(cot_theta_i);
        sample.x = fmaxf(fminf(sample.x, 1.0f - 1e-6f), 1e-6f);
        sample.y = fmaxf(fminf(sample.y, 1.0f - 1e-6f), 1e-6f);
        float x = maxval - (maxval + 1.0f) * mo_lm_erf(sqrtf(-mo_lm_log(sample.x)));
        sample.x *= 1.0f + maxval + MO_INV_SQRT_PI * tan_theta_i * mo_lm_exp(-sqr(cot_theta_i));
        for (int i = 0; i < 3; ++i) {
            float slope = erfinv_f(x);
            float value = 1.0f + x + MO_INV_SQRT_PI * tan_theta_i * mo_lm_exp(-sqr(slope)) - sample.x;
            float derivative = 1.0f - slope * tan_theta_i;
            x -= value / derivative;
        }
        r.x = erfinv_f(x); r.y = erfinv_f(fmsub(2.0f, sample.y, 1.0f));
        return r;
    }
    mo_v2 p = mo_square_to_uniform_disk_concentric(sample);
    float s = 0.5f * (1.0f + cos_theta_i);
    float a = mo_safe_sqrt(1.0f - sqr(p.x));
    p.y = fmaf(p.y, s, fnmadd(a, s, a));                     /* enoki::lerp(a, b, t) = fmadd(b, t, fnmadd(a, t, a)) */
    float x = p.x, y = p.y, z = mo_safe_sqrt(1.0f - (sqr(p.x) + sqr(p.y)));
    float sin_theta_i = mo_safe_sqrt(1.0f - sqr(cos_theta_i));
    float norm = mo_rcp(fmaf(sin_theta_i, y, cos_theta_i * z));
    r.x = fmsub(cos_theta_i, y, sin_theta_i * z) * norm; r.y = x * norm;
    return r;
}

/* microfacet.h:239-336 */
static mo_v3 mdf_sample(const mdf *d, mo_v3 wi, mo_v2 sample, float *pdf) {
    if (!d->visible) {
        float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2;
        if (d->au == d->av) {
            float ang = (2.0f * MO_PI) * sample.y;
            sin_phi = mo_lm_sin(ang); cos_phi = mo_lm_cos(ang);
            alpha_2 = d->au * d->au;
        } else {
            float ratio = d->av / d->au, tmp = ratio * mo_lm_tan((2.0f * MO_PI) * sample.y);
            cos_phi = 1.0f / sqrtf(fmaf(tmp, tmp, 1.0f));
            cos_phi = mo_mulsign(cos_phi, fabsf(sample.y - 0.5f) - 0.25f);
            sin_phi = cos_phi * tmp;
            alpha_2 = mo_rcp(sqr(cos_phi / d->au) + sqr(sin_phi / d->av));
        }
        if (!d->ggx) {
            cos_theta = 1.0f / sqrtf(fnmadd(alpha_2, mo_lm_log(1.0f - sample.x), 1.0f));
            cos_theta_2 = sqr(cos_theta);
            float cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            *pdf = (1.0f - sample.x) / (MO_PI * d->au * d->av * cos_theta_3);
        } else {
            float tan_theta_m_2 = alpha_2 * sample.x / (1.0f - sample.x);
            cos_theta = 1.0f / sqrtf(1.0f + tan_theta_m_2);
            cos_theta_2 = sqr(cos_theta);
            float temp = 1.0f + tan_theta_m_2 / alpha_2, cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            *pdf = mo_rcp(MO_PI * d->au * d->av * cos_theta_3 * sqr(temp));
        }
        float sin_theta = sqrtf(1.0f - cos_theta_2);
        return mo_v3_make(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
    }
    mo_v3 wi_p = mo_normalize(mo_v3_make(d->au * wi.x, d->av * wi.y, wi.z));
    float st2 = fmaf(wi_p.x, wi_p.x, sqr(wi_p.y)), inv = 1.0f / sqrtf(st2);      /* Frame::sincos_phi (frame.h:74-86) */
    float cos_phi = 1.0f, sin_phi = 0.0f;
    if (!(fabsf(st2) <= 4.0f * MO_EPS)) {
        cos_phi = fminf(fmaxf(wi_p.x * inv, -1.0f), 1.0f);
        sin_phi = fminf(fmaxf(wi_p.y * inv, -1.0f), 1.0f);
    }
    mo_v2 slope = mdf_sample_visible_11(d, wi_p.z, sample);
    mo_v2 sl;
    sl.x = fmsub(cos_phi, slope.x, sin_phi * slope.y) * d->au;
    sl.y = fmaf(sin_phi, slope.x, cos_phi * slope.y) * d->av;
    mo_v3 m = mo_normalize(mo_v3_make(-sl.x, -sl.y, 1.0f));
    *pdf = mdf_eval(d, m) * mdf_smith_g1(d, wi, m) * fabsf(mo_dot(wi, m)) / wi.z;
    return m;
}

/* ------------------------------------------------------------------ BSDFs */
static void roughplastic_tables(mo_bsdf *b);

/* derived constants (plastic.cpp:162-176, roughplastic.cpp:365-399) */
void mo_bsdf_prepare(mo_bsdf *b) {
    b->eta_rel = 1.0f;
    if (b->d.type == MO_BSDF_DIELECTRIC || b->d.type == MO_BSDF_PLASTIC || b->d.type == MO_BSDF_ROUGHPLASTIC || b->d.type == MO_BSDF_ROUGHDIELECTRIC ||
        b->d.type == MO_BSDF_THINDIELECTRIC) b->eta_rel = b->d.int_ior / b->d.ext_ior;
    if (b->d.type == MO_BSDF_ROUGHPLASTIC) roughplastic_tables(b);
    if (b->d.type == MO_BSDF_PLASTIC || b->d.type == MO_BSDF_ROUGHPLASTIC) {
        b->inv_eta_2 = 1.0f / (b->eta_rel * b->eta_rel);
        b->fdr_int = mo_fresnel_diffuse_reflectance(1.0f / b->eta_rel);
        b->fdr_ext = mo_fresnel_diffuse_reflectance(b->eta_rel);
        float d_mean = (b->d.reflectance[0] + b->d.reflectance[1] + b->d.reflectance[2]) * (1.0f / 3.0f);
        float s_mean = (b->d.specular_reflectance[0] + b->d.specular_reflectance[1] + b->d.specular_reflectance[2]) * (1.0f / 3.0f);
        b->spec_weight = s_mean / (d_mean + s_mean);
    }
}

/* ------------------------------------------------------------------ roughplastic tables */
/* quad::gauss_legendre (src/libcore/quad.cpp:7-66; legendre_pd: math.h:127-154) */
static void legendre_pd(int l, double x, double *lv, double *dv) {
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
    *lv = l_cur; *dv = d_cur;
}
static void gauss_legendre(int n, float *nodes, float *weights) {
    n--;
    if (n == 0) { nodes[0] = 0.0f; weights[0] = 2.0f; }
    else if (n == 1) { nodes[0] = (float) -sqrt(1.0 / 3.0); nodes[1] = -nodes[0]; weights[0] = weights[1] = 1.0f; }
    int m = (n + 1) / 2;
    for (int i = 0; i < m; ++i) {
        double x = -cos((double) (2 * i + 1) / (double) (2 * n + 2) * 3.14159265358979323846);
        for (int it = 0; it < 20; ++it) {
            double l, d; legendre_pd(n + 1, x, &l, &d);
            double step = l / d;
            x -= step;
            if (fabs(step) <= 4 * fabs(x) * 2.220446049250313e-16) break;
        }
        double l, d; legendre_pd(n + 1, x, &l, &d);
        weights[i] = weights[n - i] = (float) (2 / ((1 - x * x) * (d * d)));
        nodes[i] = (float) x; nodes[n - i] = (float) -x;
    }
    if ((n % 2) == 0) {
        double l, d; legendre_pd(n + 1, 0.0, &l, &d);
        weights[n / 2] = (float) (2.0 / (d * d));
        nodes[n / 2] = 0.0f;
    }
}
static inline mo_v3 refract_m(mo_v3 wi, mo_v3 m, float cos_theta_t, float eta_ti) {      /* fresnel.h:318-322 */
    float k = fmaf(mo_dot(wi, m), eta_ti, cos_theta_t);
    return mo_v3_make(fmsub(m.x, k, wi.x * eta_ti), fmsub(m.y, k, wi.y * eta_ti), fmsub(m.z, k, wi.z * eta_ti));
}
/* eval_transmittance / eval_reflectance (microfacet.h:462-553) for one direction, visible-normal sampling */
static float rough_transmittance(const mdf *d, mo_v3 wi, float eta, int res, const float *nodes, const float *weights) {
    float accum = 0.0f;
    for (int a = 0; a < res; ++a)
        for (int b = 0; b < res; ++b) {
            mo_v2 node = { fmaf(nodes[b], 0.5f, 0.5f), fmaf(nodes[a], 0.5f, 0.5f) };
            float pdf; mo_v3 m = mdf_sample(d, wi, node, &pdf);
            float f[4]; mo_fresnel(mo_dot(wi, m), eta, f);
            mo_v3 wo = refract_m(wi, m, f[1], f[3]);
            float smith = mdf_smith_g1(d, wo, m) * (1.0f - f[0]);
            if (wo.z * wi.z >= 0.0f) smith = 0.0f;
            accum += smith * (weights[b] * weights[a]);
        }
    return accum * 0.25f;
}
static float rough_reflectance(const mdf *d, mo_v3 wi, float eta, int res, const float *nodes, const float *weights) {
    float accum = 0.0f;
    for (int a = 0; a < res; ++a)
        for (int b = 0; b < res; ++b) {
            mo_v2 node = { fmaf(nodes[b], 0.5f, 0.5f), fmaf(nodes[a], 0.5f, 0.5f) };
            float pdf; mo_v3 m = mdf_sample(d, wi, node, &pdf);
            mo_v3 wo = reflect_m(wi, m);
            float f[4]; mo_fresnel(mo_dot(wi, m), eta, f);
            float smith = mdf_smith_g1(d, wo, m) * f[0];
            if (wo.z <= 0.0f || wi.z <= 0.0f) smith = 0.0f;
            accum += smith * (weights[b] * weights[a]);
        }
    return accum * 0.25f;
}
/* RoughPlastic::parameters_changed (roughplastic.cpp:380-399) */
static void roughplastic_tables(mo_bsdf *b) {
    mdf d = mdf_make(b->d.distribution, b->d.alpha_u, b->d.alpha_u, 1);
    float nodes_t[128], weights_t[128], nodes_r[128], weights_r[128];
    float eta = b->eta_rel, inv_eta = 1.0f / eta;
    int res_t = eta > 1.0f ? 32 : 128, res_r = inv_eta > 1.0f ? 32 : 128;
    gauss_legendre(res_t, nodes_t, weights_t);
    gauss_legendre(res_r, nodes_r, weights_r);
    float sum = 0.0f;
    for (int i = 0; i < 64; ++i) {
        float mu = fmaxf(1e-6f, (float) i / 63.0f);
        mo_v3 wi = mo_v3_make(sqrtf(1.0f - mu * mu), 0.0f, mu);
        b->ext_trans[i] = rough_transmittance(&d, wi, eta, res_t, nodes_t, weights_t);
        sum += rough_reflectance(&d, wi, inv_eta, res_r, nodes_r, weights_r) * wi.z;
    }
    b->internal_reflectance = (sum * (1.0f / 64.0f)) * 2.0f;
}
static inline float lerp_gather(const float *data, float x, int size) {                 /* roughplastic.cpp:291-302 */
    x *= (float) (size - 1);
    uint32_t index = (uint32_t) x;
    if (index > (uint32_t) (size - 2)) index = (uint32_t) (size - 2);
    float v0 = data[index], v1 = data[index + 1], t = x - (float) index;
    return fmaf(v1, t, fnmadd(v0, t, v0));
}

/* BSDFFlags::Smooth = any diffuse / glossy component (bsdf.h:106-112) */
int mo_bsdf_is_smooth(const mo_bsdf *b) {
    if (b->nest == MO_NEST_BLEND) return mo_bsdf_is_smooth(b->child[0]) || mo_bsdf_is_smooth(b->child[1]);      /* flags = union (blendbsdf.cpp:66-79) */
    if (b->nest == MO_NEST_MASK) return mo_bsdf_is_smooth(b->child[0]);                                         /* nested flags + Null (mask.cpp:69-84) */
    return b->d.type == MO_BSDF_DIFFUSE || b->d.type == MO_BSDF_ROUGHCONDUCTOR || b->d.type == MO_BSDF_PLASTIC || b->d.type == MO_BSDF_ROUGHPLASTIC ||
           b->d.type == MO_BSDF_ROUGHDIELECTRIC;
}

static float plastic_diffuse(const mo_bsdf *b, float refl) {      /* plastic.cpp:233-234,260-261 */
    return refl / (1.0f - (b->d.nonlinear ? (refl * b->fdr_int) : b->fdr_int));
}

static void rgb_channels(const mo_bsdf *b, const float refl[3], mo_bsdf_chan *c) {
    for (int k = 0; k < 3; ++k) {
        c->refl[k] = refl[k]; c->spec[k] = b->d.specular_reflectance[k]; c->trans[k] = b->d.specular_transmittance[k];
        c->eta[k] = b->d.eta[k]; c->k[k] = b->d.k[k];
    }
}

/* spectral variant: `srgb` textures evaluate the upsampled colour at each wavelength (srgb.cpp:45-52), `uniform` ones are
 * constants (uniform.cpp); conductors carry uniform eta / k */
void mo_bsdf_spectral_channels(const mo_bsdf *b, const float *wav, mo_bsdf_chan *c) {
    for (int k = 0; k < MO_WAV; ++k) {
        c->refl[k] = (b->d.uniform_mask & 1) ? b->d.reflectance[0] : mo_srgb_model_eval(b->refl_coeff, wav[k]);
        c->spec[k] = (b->d.uniform_mask & 2) ? b->d.specular_reflectance[0] : mo_srgb_model_eval(b->spec_coeff, wav[k]);
        c->trans[k] = (b->d.uniform_mask & 4) ? b->d.specular_transmittance[0] : mo_srgb_model_eval(b->trans_coeff, wav[k]);
        c->eta[k] = b->d.eta[0]; c->k[k] = b->d.k[0];
    }
}

void mo_bsdf_eval_pdf_n(const mo_bsdf *b, int n, const mo_bsdf_chan *c, mo_v3 wi, mo_v3 wo, float *value, float *pdf);

/* BSDF::sample for n channels -> returns 0 if the sample is invalid (weight 0). */
int mo_bsdf_sample_n(const mo_bsdf *b, int n, const mo_bsdf_chan *c, mo_v3 wi, float sample1, mo_v2 sample2, mo_bsample *bs, float *weight) {
    memset(bs, 0, sizeof(*bs));
    for (int k = 0; k < n; ++k) weight[k] = 0.0f;
    int flip = b->d.twosided && wi.z < 0.0f;                /* twosided.cpp:105-124 */
    if (b->d.twosided && wi.z == 0.0f) return 0;
    if (flip) wi.z = -wi.z;
    int ok = 0;
    switch (b->d.type) {
    case MO_BSDF_DIFFUSE:                                    /* diffuse.cpp:78-106 */
        bs->eta = 1.0f; bs->delta = 0;
        if (wi.z > 0.0f) {
            bs->wo = mo_square_to_cosine_hemisphere(sample2);
            bs->pdf = mo_square_to_cosine_hemisphere_pdf(bs->wo);
            if (bs->pdf > 0.0f) { for (int k = 0; k < n; ++k) weight[k] = c->refl[k]; ok = 1; }
        }
        break;
    case MO_BSDF_CONDUCTOR: {                                /* conductor.cpp:203-252 */
        if (!(wi.z > 0.0f)) break;
        bs->wo = reflect_z(wi); bs->eta = 1.0f; bs->pdf = 1.0f; bs->delta = 1;
        for (int k = 0; k < n; ++k) weight[k] = c->spec[k] * mo_fresnel_conductor(wi.z, c->eta[k], c->k[k]);
        ok = 1;
    } break;
    case MO_BSDF_ROUGHCONDUCTOR: {                           /* roughconductor.cpp:196-272 */
        float cos_theta_i = wi.z;
        if (!(cos_theta_i > 0.0f)) break;
        mdf d = mdf_make(b->d.distribution, b->d.alpha_u, b->d.alpha_v, b->d.sample_visible);
        mo_v3 m = mdf_sample(&d, wi, sample2, &bs->pdf);
        bs->wo = reflect_m(wi, m); bs->eta = 1.0f; bs->delta = 0;
        int active = bs->pdf != 0.0f && bs->wo.z > 0.0f;
        float w;
        if (b->d.sample_visible) w = mdf_smith_g1(&d, bs->wo, m);
        else w = mdf_G(&d, wi, bs->wo, m) * mo_dot(wi, m) / (cos_theta_i * m.z);
        bs->pdf /= 4.0f * mo_dot(bs->wo, m);
        float dwm = mo_dot(wi, m);
        for (int k = 0; k < n; ++k) {
            float F = mo_fresnel_conductor(dwm, c->eta[k], c->k[k]);
            float wk = w * c->spec[k];
            weight[k] = active ? F * wk : 0.0f;
        }
        ok = active;
    } break;
    case MO_BSDF_DIELECTRIC: {                               /* dielectric.cpp:201-318 (both lobes enabled, radiance transport) */
        float f[4];
        mo_fresnel(wi.z, b->eta_rel, f);
        float r_i = f[0], t_i = 1.0f - r_i;
        int selected_r = sample1 <= r_i;
        bs->pdf = selected_r ? r_i : t_i;
        bs->wo = selected_r ? reflect_z(wi) : refract_z(wi, f[1], f[3]);
        bs->eta = selected_r ? 1.0f : f[2];
        bs->delta = 1;
        for (int k = 0; k < n; ++k) {
            float wk = 1.0f;
            wk *= selected_r ? c->spec[k] : c->trans[k];
            if (!selected_r) wk *= sqr(f[3]);
            weight[k] = wk;
        }
        ok = 1;
    } break;
    case MO_BSDF_THINDIELECTRIC: {                           /* thindielectric.cpp:100-148 (both lobes enabled) */
        float f[4];
        mo_fresnel(fabsf(wi.z), b->eta_rel, f);
        float r = f[0];
        r *= 2.0f / (1.0f + r);                              /* internal reflections: r' = r + trt + tr^3t + .. */
        float t = 1.0f - r;
        int selected_r = sample1 <= r;
        bs->pdf = selected_r ? r : t;
        bs->wo = selected_r ? reflect_z(wi) : mo_neg(wi);
        bs->eta = 1.0f;
        bs->delta = 1;                                       /* DeltaReflection or Null: both in BSDFFlags::Delta (bsdf.h:117) */
        for (int k = 0; k < n; ++k) weight[k] = 1.0f * (selected_r ? c->spec[k] : c->trans[k]);
        ok = 1;
    } break;
    case MO_BSDF_PLASTIC: {                                  /* plastic.cpp:178-240 */
        float cos_theta_i = wi.z;
        if (!(cos_theta_i > 0.0f)) break;
        float f[4];
        mo_fresnel(cos_theta_i, b->eta_rel, f);
        float f_i = f[0], prob_specular = f_i * b->spec_weight, prob_diffuse = (1.0f - f_i) * (1.0f - b->spec_weight);
        prob_specular = prob_specular / (prob_specular + prob_diffuse);
        prob_diffuse = 1.0f - prob_specular;
        bs->eta = 1.0f;
        if (sample1 < prob_specular) {
            bs->wo = reflect_z(wi); bs->pdf = prob_specular; bs->delta = 1;
            for (int k = 0; k < n; ++k) weight[k] = (f_i / bs->pdf) * c->spec[k];
        } else {
            bs->wo = mo_square_to_cosine_hemisphere(sample2);
            bs->pdf = prob_diffuse * mo_square_to_cosine_hemisphere_pdf(bs->wo);
            bs->delta = 0;
            mo_fresnel(bs->wo.z, b->eta_rel, f);
            float f_o = f[0];
            for (int k = 0; k < n; ++k) weight[k] = plastic_diffuse(b, c->refl[k]) * (b->inv_eta_2 * (1.0f - f_i) * (1.0f - f_o) / prob_diffuse);
        }
        ok = 1;
    } break;
    case MO_BSDF_ROUGHPLASTIC: {                             /* roughplastic.cpp:180-232 */
        float cos_theta_i = wi.z;
        if (!(cos_theta_i > 0.0f)) break;
        float t_i = lerp_gather(b->ext_trans, cos_theta_i, 64);
        float prob_specular = (1.0f - t_i) * b->spec_weight, prob_diffuse = t_i * (1.0f - b->spec_weight);
        prob_specular = prob_specular / (prob_specular + prob_diffuse);
        bs->eta = 1.0f; bs->delta = 0;
        if (sample1 < prob_specular) {
            mdf d = mdf_make(b->d.distribution, b->d.alpha_u, b->d.alpha_u, b->d.sample_visible);
            float unused; mo_v3 m = mdf_sample(&d, wi, sample2, &unused);
            bs->wo = reflect_m(wi, m);
        } else {
            bs->wo = mo_square_to_cosine_hemisphere(sample2);
        }
        float value[4];
        mo_bsdf twin = *b; twin.d.twosided = 0;              /* wi is already on the front side */
        mo_bsdf_eval_pdf_n(&twin, n, c, wi, bs->wo, value, &bs->pdf);
        if (bs->pdf > 0.0f) { for (int k = 0; k < n; ++k) weight[k] = value[k] / bs->pdf; ok = 1; }
    } break;
    case MO_BSDF_ROUGHDIELECTRIC: {                          /* roughdielectric.cpp:202-300 (both lobes enabled, radiance transport) */
        float cos_theta_i = wi.z;
        int active = cos_theta_i != 0.0f;                    /* perfectly grazing configurations are ignored */
        mdf d = mdf_make(b->d.distribution, b->d.alpha_u, b->d.alpha_v, b->d.sample_visible), sd = d;
        if (!b->d.sample_visible) {                          /* Walter et al.'s trick: widen the sampling distribution */
            float sc = 1.2f - 0.2f * sqrtf(fabsf(cos_theta_i));
            sd.au *= sc; sd.av *= sc;
        }
        mo_v3 wi_up = mo_v3_make(mo_mulsign(wi.x, cos_theta_i), mo_mulsign(wi.y, cos_theta_i), mo_mulsign(wi.z, cos_theta_i));
        mo_v3 m = mdf_sample(&sd, wi_up, sample2, &bs->pdf);
        active = active && bs->pdf != 0.0f;
        float f[4];
        mo_fresnel(mo_dot(wi, m), b->eta_rel, f);            /* F, cos_theta_t, eta_it, eta_ti */
        float F = f[0];
        int selected_r = sample1 <= F && active, selected_t = !selected_r && active;
        bs->pdf *= selected_r ? F : 1.0f - F;
        bs->eta = selected_r ? 1.0f : f[2];
        bs->delta = 0;
        float dwh_dwo = 0.0f;
        if (selected_r) {
            bs->wo = reflect_m(wi, m);
            dwh_dwo = mo_rcp(4.0f * mo_dot(bs->wo, m));
        }
        if (selected_t) {
            bs->wo = refract_m(wi, m, f[1], f[3]);
            dwh_dwo = (sqr(bs->eta) * mo_dot(bs->wo, m)) / sqr(mo_dot(wi, m) + bs->eta * mo_dot(bs->wo, m));
        }
        float g;
        if (b->d.sample_visible) g = mdf_smith_g1(&d, bs->wo, m);
        else g = mdf_G(&d, wi, bs->wo, m) * mo_dot(wi, m) / (cos_theta_i * m.z);
        bs->pdf *= fabsf(dwh_dwo);
        for (int k = 0; k < n; ++k) {
            float wk = 1.0f;
            if (selected_r) wk *= c->spec[k];
            if (selected_t) wk *= sqr(f[3]) * c->trans[k];
            weight[k] = wk * g;
        }
        ok = active;
    } break;
    default: break;
    }
    if (!ok) for (int k = 0; k < n; ++k) weight[k] = 0.0f;
    if (flip) bs->wo.z = -bs->wo.z;
    return ok;
}

/* BSDF::eval and BSDF::pdf for n channels */
void mo_bsdf_eval_pdf_n(const mo_bsdf *b, int n, const mo_bsdf_chan *c, mo_v3 wi, mo_v3 wo, float *value, float *pdf) {
    for (int k = 0; k < n; ++k) value[k] = 0.0f;
    *pdf = 0.0f;
    if (b->d.twosided) {                                     /* twosided.cpp:127-175 */
        if (wi.z == 0.0f) return;
        if (wi.z < 0.0f) { wi.z = -wi.z; wo.z = -wo.z; }
    }
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (b->d.type == MO_BSDF_ROUGHDIELECTRIC) {              /* roughdielectric.cpp:302-375 (eval), :377-447 (pdf) */
        if (cos_theta_i == 0.0f) return;
        int reflect = cos_theta_i * cos_theta_o > 0.0f;
        float m_inv_eta = 1.0f / b->eta_rel;                 /* parameters_changed(): roughdielectric.cpp:198-200 */
        float eta = cos_theta_i > 0.0f ? b->eta_rel : m_inv_eta, inv_eta = cos_theta_i > 0.0f ? m_inv_eta : b->eta_rel;
        mo_v3 m = mo_normalize(mo_add(wi, mo_scale(wo, reflect ? 1.0f : eta)));
        m = mo_v3_make(mo_mulsign(m.x, m.z), mo_mulsign(m.y, m.z), mo_mulsign(m.z, m.z));
        mdf d = mdf_make(b->d.distribution, b->d.alpha_u, b->d.alpha_v, b->d.sample_visible);
        float D = mdf_eval(&d, m);
        float f[4]; mo_fresnel(mo_dot(wi, m), b->eta_rel, f);
        float F = f[0], G = mdf_G(&d, wi, wo, m);
        float dwm = mo_dot(wi, m), dom = mo_dot(wo, m);
        if (reflect) {
            float v = F * D * G / (4.0f * fabsf(cos_theta_i));
            for (int k = 0; k < n; ++k) value[k] = v * c->spec[k];
        } else {
            float scale = sqr(inv_eta);                      /* radiance transport: solid angle compression */
            float v = fabsf((scale * (1.0f - F) * D * G * eta * eta * dwm * dom) / (cos_theta_i * sqr(dwm + eta * dom)));
            for (int k = 0; k < n; ++k) value[k] = v * c->trans[k];
        }
        int active = dwm * cos_theta_i > 0.0f && dom * cos_theta_o > 0.0f;
        float dwh_dwo = reflect ? mo_rcp(4.0f * dom) : (eta * eta * dom) / sqr(dwm + eta * dom);
        mdf sd = d;
        if (!b->d.sample_visible) {
            float sc = 1.2f - 0.2f * sqrtf(fabsf(cos_theta_i));
            sd.au *= sc; sd.av *= sc;
        }
        mo_v3 wi_up = mo_v3_make(mo_mulsign(wi.x, cos_theta_i), mo_mulsign(wi.y, cos_theta_i), mo_mulsign(wi.z, cos_theta_i));
        float prob = mdf_pdf(&sd, wi_up, m);
        prob *= reflect ? F : 1.0f - F;
        *pdf = active ? prob * fabsf(dwh_dwo) : 0.0f;
        return;
    }
    if (!(cos_theta_i > 0.0f && cos_theta_o > 0.0f)) return;
    switch (b->d.type) {
    case MO_BSDF_DIFFUSE:                                    /* diffuse.cpp:108-135 */
        for (int k = 0; k < n; ++k) value[k] = (c->refl[k] * MO_INV_PI) * cos_theta_o;
        *pdf = mo_square_to_cosine_hemisphere_pdf(wo);
        break;
    case MO_BSDF_ROUGHCONDUCTOR: {                           /* roughconductor.cpp:274-391 */
        mo_v3 H = mo_normalize(mo_add(wo, wi));
        mdf d = mdf_make(b->d.distribution, b->d.alpha_u, b->d.alpha_v, b->d.sample_visible);
        float D = mdf_eval(&d, H);
        if (D != 0.0f) {
            float G = mdf_G(&d, wi, wo, H);
            float result = D * G / (4.0f * cos_theta_i);
            float dwh = mo_dot(wi, H);
            for (int k = 0; k < n; ++k) value[k] = mo_fresnel_conductor(dwh, c->eta[k], c->k[k]) * (result * c->spec[k]);
        }
        if (mo_dot(wi, H) > 0.0f && mo_dot(wo, H) > 0.0f) {
            if (b->d.sample_visible) *pdf = mdf_eval(&d, H) * mdf_smith_g1(&d, wi, H) / (4.0f * cos_theta_i);
            else *pdf = mdf_pdf(&d, wi, H) / (4.0f * mo_dot(wo, H));
        }
    } break;
    case MO_BSDF_PLASTIC: {                                  /* plastic.cpp:243-297 */
        float f[4];
        mo_fresnel(cos_theta_i, b->eta_rel, f); float f_i = f[0];
        mo_fresnel(cos_theta_o, b->eta_rel, f); float f_o = f[0];
        float k2 = mo_square_to_cosine_hemisphere_pdf(wo) * b->inv_eta_2 * (1.0f - f_i) * (1.0f - f_o);
        for (int k = 0; k < n; ++k) value[k] = plastic_diffuse(b, c->refl[k]) * k2;
        float prob_specular = f_i * b->spec_weight, prob_diffuse = (1.0f - f_i) * (1.0f - b->spec_weight);
        prob_diffuse = prob_diffuse / (prob_specular + prob_diffuse);
        *pdf = mo_square_to_cosine_hemisphere_pdf(wo) * prob_diffuse;
    } break;
    case MO_BSDF_ROUGHPLASTIC: {                             /* roughplastic.cpp:234-289 (eval), :304-352 (pdf) */
        mdf d = mdf_make(b->d.distribution, b->d.alpha_u, b->d.alpha_u, b->d.sample_visible);
        mo_v3 H = mo_normalize(mo_add(wo, wi));
        float D = mdf_eval(&d, H);
        float f[4]; mo_fresnel(mo_dot(wi, H), b->eta_rel, f);
        float G = mdf_G(&d, wi, wo, H);
        float spec_v = f[0] * D * G / (4.0f * cos_theta_i);
        float t_i = lerp_gather(b->ext_trans, cos_theta_i, 64), t_o = lerp_gather(b->ext_trans, cos_theta_o, 64);
        float kd = MO_INV_PI * b->inv_eta_2 * cos_theta_o * t_i * t_o;
        for (int k = 0; k < n; ++k) {
            float diff = c->refl[k] / (1.0f - (b->d.nonlinear ? (c->refl[k] * b->internal_reflectance) : b->internal_reflectance));
            value[k] = spec_v * c->spec[k] + diff * kd;
        }
        float prob_specular = (1.0f - t_i) * b->spec_weight, prob_diffuse = t_i * (1.0f - b->spec_weight);
        prob_specular = prob_specular / (prob_specular + prob_diffuse);
        prob_diffuse = 1.0f - prob_specular;
        float result;
        if (b->d.sample_visible) result = mdf_eval(&d, H) * mdf_smith_g1(&d, wi, H) / (4.0f * cos_theta_i);
        else result = mdf_pdf(&d, wi, H) / (4.0f * mo_dot(wo, H));
        result *= prob_specular;
        result += prob_diffuse * mo_square_to_cosine_hemisphere_pdf(wo);
        *pdf = result;
    } break;
    default: break;      /* conductor / dielectric: delta lobes only, eval = pdf = 0 */
    }
}

/* ------------------------------------------------------------------ blendbsdf / mask */
/* Texture::eval_1 of the weight / opacity, clamped (blendbsdf.cpp:177-179, mask.cpp:170-172).  `value` is what a plain BSDF would
 * receive as its reflectance: the constant (first channel) or the texture lookup. */
static float nest_weight(const mo_bsdf *b, const float *value) {
    float w = value[0];
    if (b->weight_lum) w = fmaf(0.072169f, value[2], fmaf(0.715160f, value[1], 0.212671f * value[0]));      /* luminance (spectrum.h:239-241) */
    return fminf(fmaxf(w, 0.0f), 1.0f);
}
/* per-channel inputs of child k: its own parameters; `refl` = its reflectance at the surface point (RGB variant: constant or textured) */
static void child_channels(const mo_bsdf *c, int n, const float *wav, const float *refl, mo_bsdf_chan *out) {
    if (n == 3) rgb_channels(c, refl, out);
    else mo_bsdf_spectral_channels(c, wav, out);
}
/* blendbsdf.cpp:82-123 (ctx.component == -1) and mask.cpp:92-131 (both the null and the nested components enabled) */
static int nest_sample(const mo_bsdf *b, int n, const float *wav, const float *refl9, float w, mo_v3 wi, float sample1, mo_v2 sample2, mo_bsample *bs, float *weight) {
    bs->wo = mo_v3_make(0.0f, 0.0f, 0.0f); bs->pdf = 0.0f; bs->eta = 0.0f; bs->delta = 0;
    for (int k = 0; k < n; ++k) weight[k] = 0.0f;
    int flip = 0;
    if (b->d.twosided) {                                    /* twosided.cpp:94-123 around the whole nest */
        if (wi.z == 0.0f) return 0;
        flip = wi.z < 0.0f;
        if (flip) wi.z = -wi.z;
    }
    mo_bsdf_chan c; int ok = 0;
    if (b->nest == MO_NEST_BLEND) {
        if (sample1 > w) {
            child_channels(b->child[0], n, wav, refl9 + 3, &c);
            ok = mo_bsdf_sample_n(b->child[0], n, &c, wi, (sample1 - w) / (1.0f - w), sample2, bs, weight);
        } else if (sample1 <= w) {
            child_channels(b->child[1], n, wav, refl9 + 6, &c);
            ok = mo_bsdf_sample_n(b->child[1], n, &c, wi, sample1 / w, sample2, bs, weight);
        }
    } else {
        bs->wo = mo_neg(wi); bs->eta = 1.0f; bs->pdf = 1.0f - w; bs->delta = 1;      /* BSDFFlags::Null is part of BSDFFlags::Delta (bsdf.h:98) */
        for (int k = 0; k < n; ++k) weight[k] = 1.0f;
        ok = 1;
        if (sample1 < w) {                                  /* the nested sample replaces the record as it is (mask.cpp:123-128) */
            child_channels(b->child[0], n, wav, refl9 + 3, &c);
            ok = mo_bsdf_sample_n(b->child[0], n, &c, wi, sample1 / w, sample2, bs, weight);
        }
    }
    if (flip) bs->wo.z = -bs->wo.z;
    return ok;
}
/* blendbsdf.cpp:125-158, mask.cpp:133-159 */
static void nest_eval_pdf(const mo_bsdf *b, int n, const float *wav, const float *refl9, float w, mo_v3 wi, mo_v3 wo, float *value, float *pdf) {
    for (int k = 0; k < n; ++k) value[k] = 0.0f;
    *pdf = 0.0f;
    if (b->d.twosided) {
        if (wi.z == 0.0f) return;
        if (wi.z < 0.0f) { wi.z = -wi.z; wo.z = -wo.z; }
    }
    mo_bsdf_chan c; float v0[4], p0;
    child_channels(b->child[0], n, wav, refl9 + 3, &c);
    mo_bsdf_eval_pdf_n(b->child[0], n, &c, wi, wo, v0, &p0);
    if (b->nest == MO_NEST_MASK) {
        for (int k = 0; k < n; ++k) value[k] = v0[k] * w;
        *pdf = p0 * w;
        return;
    }
    float v1[4], p1;
    child_channels(b->child[1], n, wav, refl9 + 6, &c);
    mo_bsdf_eval_pdf_n(b->child[1], n, &c, wi, wo, v1, &p1);
    for (int k = 0; k < n; ++k) value[k] = fmaf(v1[k], w, v0[k] * (1.0f - w));
    *pdf = fmaf(p1, w, p0 * (1.0f - w));
}

/* RGB variant */
int mo_bsdf_sample(const mo_bsdf *b, const float refl[3], mo_v3 wi, float sample1, mo_v2 sample2, mo_bsample *bs, float weight[3]) {
    if (b->nest) return nest_sample(b, 3, NULL, refl, nest_weight(b, refl), wi, sample1, sample2, bs, weight);      /* refl: 9 values (mo_surface_reflectance) */
    mo_bsdf_chan c; rgb_channels(b, refl, &c);
    return mo_bsdf_sample_n(b, 3, &c, wi, sample1, sample2, bs, weight);
}
void mo_bsdf_eval_pdf(const mo_bsdf *b, const float refl[3], mo_v3 wi, mo_v3 wo, float value[3], float *pdf) {
    if (b->nest) { nest_eval_pdf(b, 3, NULL, refl, nest_weight(b, refl), wi, wo, value, pdf); return; }
    mo_bsdf_chan c; rgb_channels(b, refl, &c);
    mo_bsdf_eval_pdf_n(b, 3, &c, wi, wo, value, pdf);
}
/* spectral variant */
int mo_bsdf_sample_spec(const mo_bsdf *b, const float *wav, const mo_bsdf_chan *c, mo_v3 wi, float sample1, mo_v2 sample2, mo_bsample *bs, float *weight) {
    if (b->nest) return nest_sample(b, MO_WAV, wav, NULL, nest_weight(b, c->refl), wi, sample1, sample2, bs, weight);
    return mo_bsdf_sample_n(b, MO_WAV, c, wi, sample1, sample2, bs, weight);
}
void mo_bsdf_eval_pdf_spec(const mo_bsdf *b, const float *wav, const mo_bsdf_chan *c, mo_v3 wi, mo_v3 wo, float *value, float *pdf) {
    if (b->nest) { nest_eval_pdf(b, MO_WAV, wav, NULL, nest_weight(b, c->refl), wi, wo, value, pdf); return; }
    mo_bsdf_eval_pdf_n(b, MO_WAV, c, wi, wo, value, pdf);
}

/* ------------------------------------------------------------------ known-answer entry points */
void mo_kat_fresnel(float cos_theta_i, float eta, float *out4) { mo_fresnel(cos_theta_i, eta, out4); }
float mo_kat_fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) { return mo_fresnel_conductor(cos_theta_i, eta_r, eta_i); }
float mo_kat_fresnel_diffuse(float eta) { return mo_fresnel_diffuse_reflectance(eta); }
/* which: 0 eval(m = v), 1 pdf(wi, m = v), 2 smith_g1(v, m = wi) -- the argument order of test_microfacet.py */
void mo_kat_microfacet(int ggx, float alpha_u, float alpha_v, int visible, int which, uint64_t n, const float *v3, const float *wi3, float *out) {
    mdf d = mdf_make(ggx, alpha_u, alpha_v, visible);
    for (uint64_t i = 0; i < n; ++i) {
        mo_v3 v = mo_v3_make(v3[3 * i], v3[3 * i + 1], v3[3 * i + 2]), wi = mo_v3_make(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]);
        out[i] = which == 0 ? mdf_eval(&d, v) : (which == 1 ? mdf_pdf(&d, wi, v) : mdf_smith_g1(&d, v, wi));
    }
}
void mo_kat_microfacet_sample(int ggx, float alpha_u, float alpha_v, int visible, uint64_t n, const float *wi3, const float *sample2, float *m3, float *pdf) {
    mdf d = mdf_make(ggx, alpha_u, alpha_v, visible);
    for (uint64_t i = 0; i < n; ++i) {
        mo_v2 s = { sample2[2 * i], sample2[2 * i + 1] };
        mo_v3 m = mdf_sample(&d, mo_v3_make(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), s, &pdf[i]);
        m3[3 * i] = m.x; m3[3 * i + 1] = m.y; m3[3 * i + 2] = m.z;
    }
}
void mo_kat_gauss_legendre(int n, float *nodes, float *weights) { gauss_legendre(n, nodes, weights); }
void mo_kat_roughplastic_tables(const mo_bsdf_desc *desc, float *out65) {
    mo_bsdf b; memset(&b, 0, sizeof(b)); b.d = *desc; mo_bsdf_prepare(&b);
    memcpy(out65, b.ext_trans, sizeof(float) * 64); out65[64] = b.internal_reflectance;
}
/* generic BSDF evaluation for n (wi, wo, sample1, sample2) tuples.
 * out per tuple: eval(3) pdf | sampled wo(3) pdf eta delta weight(3) valid = 14 floats */
void mo_kat_bsdf(const mo_bsdf_desc *desc, uint64_t n, const float *wi3, const float *wo3, const float *sample3, float *out14) {
    mo_bsdf b; memset(&b, 0, sizeof(b)); b.d = *desc; mo_bsdf_prepare(&b);
    for (uint64_t i = 0; i < n; ++i) {
        mo_v3 wi = mo_v3_make(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), wo = mo_v3_make(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]);
        float *o = out14 + 14 * i;
        mo_bsdf_eval_pdf(&b, desc->reflectance, wi, wo, o, o + 3);
        mo_bsample bs; mo_v2 s2 = { sample3[3 * i + 1], sample3[3 * i + 2] };
        int ok = mo_bsdf_sample(&b, desc->reflectance, wi, sample3[3 * i], s2, &bs, o + 10);
        o[4] = bs.wo.x; o[5] = bs.wo.y; o[6] = bs.wo.z; o[7] = bs.pdf; o[8] = bs.eta; o[9] = (float) bs.delta; o[13] = (float) ok;
    }
}
void mo_kat_nested_bsdf(int kind, float weight, int twosided, const mo_bsdf_desc *child0, const mo_bsdf_desc *child1, uint64_t n,
                        const float *wi3, const float *wo3, const float *sample3, float *out14) {
    mo_bsdf top, c0, c1;
    memset(&top, 0, sizeof(top)); memset(&c0, 0, sizeof(c0)); memset(&c1, 0, sizeof(c1));
    c0.d = *child0; mo_bsdf_prepare(&c0);
    if (child1) { c1.d = *child1; mo_bsdf_prepare(&c1); }
    top.nest = kind; top.d.twosided = twosided; top.child[0] = &c0; top.child[1] = child1 ? &c1 : NULL;
    float wv[9] = { weight, weight, weight, 0, 0, 0, 0, 0, 0 };
    for (int k = 0; k < 3; ++k) { wv[3 + k] = child0->reflectance[k]; if (child1) wv[6 + k] = child1->reflectance[k]; }
    top.child_tex[0] = top.child_tex[1] = -1;
    for (uint64_t i = 0; i < n; ++i) {
        mo_v3 wi = mo_v3_make(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), wo = mo_v3_make(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]);
        float *o = out14 + 14 * i;
        mo_bsdf_eval_pdf(&top, wv, wi, wo, o, o + 3);
        mo_bsample bs; mo_v2 s2 = { sample3[3 * i + 1], sample3[3 * i + 2] };
        int ok = mo_bsdf_sample(&top, wv, wi, sample3[3 * i], s2, &bs, o + 10);
        o[4] = bs.wo.x; o[5] = bs.wo.y; o[6] = bs.wo.z; o[7] = bs.pdf; o[8] = bs.eta; o[9] = (float) bs.delta; o[13] = (float) ok;
    }
}
