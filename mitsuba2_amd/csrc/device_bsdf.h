// device_bsdf.h -- BSDF models beyond `diffuse` for the path kernels.
//
//   fresnel / fresnel_conductor                      include/mitsuba/render/fresnel.h:34-122
//   MicrofacetDistribution (Beckmann, GGX, visible normals)   include/mitsuba/render/microfacet.h:187-440
//   SmoothConductor   src/bsdfs/conductor.cpp:203-262        RoughConductor  src/bsdfs/roughconductor.cpp:196-391
//   SmoothDielectric  src/bsdfs/dielectric.cpp:201-318       SmoothPlastic   src/bsdfs/plastic.cpp:178-297
//   RoughPlastic      src/bsdfs/roughplastic.cpp:180-399 (tables built on the device by k_roughplastic_tables)
//   TwoSidedBRDF      src/bsdfs/twosided.cpp:94-175 (one nested BSDF for both sides)
//
// Arithmetic is written out operation by operation (explicit fmaf where the reference fuses) so that the results agree
// with the test oracle's to the last bit wherever no transcendental function is involved.
#pragma once
#include "device_math.h"

namespace mtsamd {

constexpr int kBsdfDiffuse = 0, kBsdfConductor = 1, kBsdfRoughConductor = 2, kBsdfDielectric = 3, kBsdfPlastic = 4, kBsdfRoughPlastic = 5,
              kBsdfRoughDielectric = 6, kBsdfThinDielectric = 7,
              kBsdfBlend = 8, kBsdfMask = 9;      // blendbsdf.cpp / mask.cpp over plain records of the same table (DevBsdf::nested0/1)
constexpr uint32_t kBsdfTwoSided = 1u, kBsdfGGX = 2u, kBsdfSampleVisible = 4u, kBsdfNonlinear = 8u;
// blend / mask: some child has a smooth component (BSDFFlags::Smooth of the union of the nested flags); the weight / opacity texture
// is a bitmap, whose eval_1 is the luminance of the texel (bitmap.cpp:215-231)
constexpr uint32_t kBsdfNestSmooth = 128u, kBsdfWeightLum = 256u;
// spectral variant: the parameter is a `uniform` spectrum (its constant sits in the first colour channel) instead of `srgb`
constexpr uint32_t kBsdfUniformRefl = 16u, kBsdfUniformSpec = 32u, kBsdfUniformTrans = 64u;
constexpr float kInvSqrtPi = 0.56418958354775628695f, kEps = kEpsilon;

MTS_DEV float sqr(float x) { return x * x; }

struct Fresnel { float r, cos_theta_t, eta_it, eta_ti; };
MTS_DEV Fresnel fresnel(float cos_theta_i, float eta) {
    const bool outside = cos_theta_i >= 0.0f;
    const float rcp_eta = rcp(eta), eta_it = outside ? eta : rcp_eta, eta_ti = outside ? rcp_eta : eta;
    const float cos_theta_t_sqr = fmaf(-fmaf(-cos_theta_i, cos_theta_i, 1.0f), eta_ti * eta_ti, 1.0f);
    const float ci = fabsf(cos_theta_i), ct = safe_sqrt(cos_theta_t_sqr);
    const bool index_matched = eta == 1.0f, special = index_matched || ci == 0.0f;
    const float a_s = fmaf(-eta_it, ct, ci) / fmaf(eta_it, ct, ci);
    const float a_p = fmaf(-eta_it, ci, ct) / fmaf(eta_it, ci, ct);
    float r = 0.5f * (sqr(a_s) + sqr(a_p));
    if (special) r = index_matched ? 0.0f : 1.0f;
    Fresnel f = { r, mulsign_neg(ct, cos_theta_i), eta_it, eta_ti };
    return f;
}

MTS_DEV float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    const float c2 = cos_theta_i * cos_theta_i, s2 = 1.0f - c2, s4 = s2 * s2;
    const float temp_1 = eta_r * eta_r - eta_i * eta_i - s2;
    const float a_2_pb_2 = safe_sqrt(temp_1 * temp_1 + 4.0f * eta_i * eta_i * eta_r * eta_r);
    const float a = safe_sqrt(0.5f * (a_2_pb_2 + temp_1));
    const float term_1 = a_2_pb_2 + c2, term_2 = 2.0f * cos_theta_i * a;
    const float r_s = (term_1 - term_2) / (term_1 + term_2);
    const float term_3 = a_2_pb_2 * c2 + s4, term_4 = term_2 * s2;
    const float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}

MTS_DEV f3 reflect_z(f3 wi) { return mk3(-wi.x, -wi.y, wi.z); }
MTS_DEV f3 reflect_m(f3 wi, f3 m) {
    const float k = 2.0f * dot(wi, m);
    return mk3(fmaf(m.x, k, -wi.x), fmaf(m.y, k, -wi.y), fmaf(m.z, k, -wi.z));
}
MTS_DEV f3 refract_m(f3 wi, f3 m, float cos_theta_t, float eta_ti) {          // fresnel.h:318-322
    const float k = fmaf(dot(wi, m), eta_ti, cos_theta_t);
    return mk3(fmaf(m.x, k, -(wi.x * eta_ti)), fmaf(m.y, k, -(wi.y * eta_ti)), fmaf(m.z, k, -(wi.z * eta_ti)));
}

// ---------------------------------------------------------------------------------------------
struct Mdf { bool ggx; float au, av; bool visible; };
MTS_DEV Mdf mdf_make(bool ggx, float au, float av, bool visible) {
    Mdf d = { ggx, fmaxf(au, 1e-4f), fmaxf(av, 1e-4f), visible };
    return d;
}
MTS_DEV float mdf_eval(const Mdf &d, f3 m) {
    const float alpha_uv = d.au * d.av, cos_theta = m.z, cos_theta_2 = sqr(cos_theta);
    float result;
    if (!d.ggx) result = lm_exp(-(sqr(m.x / d.au) + sqr(m.y / d.av)) / cos_theta_2) / (kPi * alpha_uv * sqr(cos_theta_2));
    else result = rcp(kPi * alpha_uv * sqr(sqr(m.x / d.au) + sqr(m.y / d.av) + sqr(m.z)));
    return result * cos_theta > 1e-20f ? result : 0.0f;
}
MTS_DEV float mdf_smith_g1(const Mdf &d, f3 v, f3 m) {
    const float xy_alpha_2 = sqr(d.au * v.x) + sqr(d.av * v.y), tan_theta_alpha_2 = xy_alpha_2 / sqr(v.z);
    float result;
    if (!d.ggx) {
        const float a = 1.0f / sqrtf(tan_theta_alpha_2), a_sqr = sqr(a);
        result = a >= 1.6f ? 1.0f : (3.535f * a + 2.181f * a_sqr) / (1.0f + 2.276f * a + 2.577f * a_sqr);
    } else {
        result = 2.0f / (1.0f + sqrtf(1.0f + tan_theta_alpha_2));
    }
    if (xy_alpha_2 == 0.0f) result = 1.0f;
    if (dot(v, m) * v.z <= 0.0f) result = 0.0f;
    return result;
}
MTS_DEV float mdf_G(const Mdf &d, f3 wi, f3 wo, f3 m) { return mdf_smith_g1(d, wi, m) * mdf_smith_g1(d, wo, m); }
MTS_DEV float mdf_pdf(const Mdf &d, f3 wi, f3 m) {
    float result = mdf_eval(d, m);
    if (d.visible) result *= mdf_smith_g1(d, wi, m) * fabsf(dot(wi, m)) / wi.z;
    else result *= m.z;
    return result;
}

// Giles, "Approximating the erfinv function" (single precision)
MTS_DEV float erfinv_f(float x) {
    float w = -lm_log((1.0f - x) * (1.0f + x)), p;
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

MTS_DEV f2 mdf_sample_visible_11(const Mdf &d, float cos_theta_i, f2 sample) {
    f2 r;
    if (!d.ggx) {
        const float tan_theta_i = safe_sqrt(fmaf(-cos_theta_i, cos_theta_i, 1.0f)) / cos_theta_i;
        const float cot_theta_i = rcp(tan_theta_i);
        const float maxval = lm_erf(cot_theta_i);
        sample.x = fmaxf(fminf(sample.x, 1.0f - 1e-6f), 1e-6f);
        sample.y = fmaxf(fminf(sample.y, 1.0f - 1e-6f), 1e-6f);
        float x = maxval - (maxval + 1.0f) * lm_erf(sqrtf(-lm_log(sample.x)));
        sample.x *= 1.0f + maxval + kInvSqrtPi * tan_theta_i * lm_exp(-sqr(cot_theta_i));
        for (int i = 0; i < 3; ++i) {
            const float slope = erfinv_f(x);
            const float value = 1.0f + x + kInvSqrtPi * tan_theta_i * lm_exp(-sqr(slope)) - sample.x;
            const float derivative = 1.0f - slope * tan_theta_i;
            x -= value / derivative;
        }
        r.x = erfinv_f(x); r.y = erfinv_f(fmaf(2.0f, sample.y, -1.0f));
        return r;
    }
    f2 p = square_to_uniform_disk_concentric(sample);
    const float s = 0.5f * (1.0f + cos_theta_i);
    const float a = safe_sqrt(1.0f - sqr(p.x));
    p.y = fmaf(p.y, s, fmaf(-a, s, a));          // enoki::lerp(a, b, t) = fmadd(b, t, fnmadd(a, t, a))
    const float x = p.x, y = p.y, z = safe_sqrt(1.0f - (sqr(p.x) + sqr(p.y)));
    const float sin_theta_i = safe_sqrt(1.0f - sqr(cos_theta_i));
    const float norm = rcp(fmaf(sin_theta_i, y, cos_theta_i * z));
    r.x = fmaf(cos_theta_i, y, -(sin_theta_i * z)) * norm; r.y = x * norm;
    return r;
}

MTS_DEV f3 mdf_sample(const Mdf &d, f3 wi, f2 sample, float &pdf) {
    if (!d.visible) {
        float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2;
        if (d.au == d.av) {
            const float ang = (2.0f * kPi) * sample.y;
            sin_phi = lm_sin(ang); cos_phi = lm_cos(ang);
            alpha_2 = d.au * d.au;
        } else {
            const float ratio = d.av / d.au, tmp = ratio * lm_tan((2.0f * kPi) * sample.y);
            cos_phi = 1.0f / sqrtf(fmaf(tmp, tmp, 1.0f));
            cos_phi = mulsign(cos_phi, fabsf(sample.y - 0.5f) - 0.25f);
            sin_phi = cos_phi * tmp;
            alpha_2 = rcp(sqr(cos_phi / d.au) + sqr(sin_phi / d.av));
        }
        if (!d.ggx) {
            cos_theta = 1.0f / sqrtf(fmaf(-alpha_2, lm_log(1.0f - sample.x), 1.0f));
            cos_theta_2 = sqr(cos_theta);
            const float cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            pdf = (1.0f - sample.x) / (kPi * d.au * d.av * cos_theta_3);
        } else {
            const float tan_theta_m_2 = alpha_2 * sample.x / (1.0f - sample.x);
            cos_theta = 1.0f / sqrtf(1.0f + tan_theta_m_2);
            cos_theta_2 = sqr(cos_theta);
            const float temp = 1.0f + tan_theta_m_2 / alpha_2, cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            pdf = rcp(kPi * d.au * d.av * cos_theta_3 * sqr(temp));
        }
        const float sin_theta = sqrtf(1.0f - cos_theta_2);
        return mk3(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
    }
    const f3 wi_p = normalize(mk3(d.au * wi.x, d.av * wi.y, wi.z));
    const float st2 = fmaf(wi_p.x, wi_p.x, sqr(wi_p.y)), inv = 1.0f / sqrtf(st2);
    float cos_phi = 1.0f, sin_phi = 0.0f;
    if (!(fabsf(st2) <= 4.0f * kEps)) {
        cos_phi = fminf(fmaxf(wi_p.x * inv, -1.0f), 1.0f);
        sin_phi = fminf(fmaxf(wi_p.y * inv, -1.0f), 1.0f);
    }
    const f2 slope = mdf_sample_visible_11(d, wi_p.z, sample);
    const float sx = fmaf(cos_phi, slope.x, -(sin_phi * slope.y)) * d.au;
    const float sy = fmaf(sin_phi, slope.x, cos_phi * slope.y) * d.av;
    const f3 m = normalize(mk3(-sx, -sy, 1.0f));
    pdf = mdf_eval(d, m) * mdf_smith_g1(d, wi, m) * fabsf(dot(wi, m)) / wi.z;
    return m;
}

// ---------------------------------------------------------------------------------------------
// Device-side BSDF record (128 B).  Roles of the generic fields per model:
//   conductor / roughconductor   e = eta (rgb), k = extinction (rgb), s = specular_reflectance
//   dielectric, roughdielectric  e.x = eta = int_ior / ext_ior, k = specular_transmittance, s = specular_reflectance
//   plastic                      e.x = eta, e.y = 1 / eta^2, e.z = fdr_int, k.x = specular sampling weight,
//                                (r, g, b) = diffuse_reflectance, s = specular_reflectance
//   roughplastic                 as plastic with e.z = internal diffuse reflectance, alpha_u = roughness and
//                                table = 64 external transmittances over cos(theta) in [0, 1]
struct DevBsdf {
    float r, g, b; int32_t type;
    int32_t texture; float c0, c1, c2;       // c*: srgb_model coefficients of (r, g, b) (spectral variant)
    float sr, sg, sb; uint32_t flags;
    float er, eg, eb, alpha_u;
    float kr, kg, kb, alpha_v;
    float sc0, sc1, sc2, pad0;               // spectral variant: srgb_model coefficients of specular_reflectance
    float tc0, tc1, tc2, pad1;               //                   ... of specular_transmittance
    const float *table; uint32_t nested0, nested1;      // roughplastic: external transmittance table (device memory); blend / mask: child records
};
constexpr int kRoughTableRes = 64;           // MTS_ROUGH_TRANSMITTANCE_RES

MTS_DEV float lerp_gather(const float *data, float x, int size) {             // roughplastic.cpp:291-302
    x *= (float) (size - 1);
    const uint32_t index = min((uint32_t) x, (uint32_t) (size - 2));
    const float v0 = data[index], v1 = data[index + 1], t = x - (float) index;
    return fmaf(v1, t, fmaf(-v0, t, v0));
}

struct BsdfSample { f3 wo; float pdf, eta; bool delta; };

MTS_DEV bool bsdf_is_smooth(const DevBsdf &b) {          // BSDFFlags::Smooth: any diffuse / glossy component
    return b.type == kBsdfDiffuse || b.type == kBsdfRoughConductor || b.type == kBsdfPlastic || b.type == kBsdfRoughPlastic ||
           b.type == kBsdfRoughDielectric || (b.flags & kBsdfNestSmooth) != 0u;
}

// Per-channel inputs of a BSDF evaluation: N = 3 colour channels (RGB variant) or N = 4 wavelengths (spectral variant).
//   refl  diffuse reflectance (diffuse.reflectance / plastic.diffuse_reflectance) at the hit point
//   spec  specular_reflectance, trans  specular_transmittance, eta / k  complex IOR of conductors
template <int N> struct BsdfChannels { float refl[N], spec[N], trans[N], eta[N], k[N]; };

MTS_DEV BsdfChannels<3> rgb_channels(const DevBsdf &b, f3 refl) {
    BsdfChannels<3> c;
    c.refl[0] = refl.x; c.refl[1] = refl.y; c.refl[2] = refl.z;
    c.spec[0] = b.sr; c.spec[1] = b.sg; c.spec[2] = b.sb;
    c.trans[0] = b.kr; c.trans[1] = b.kg; c.trans[2] = b.kb;      // dielectric: k.* holds the transmittance
    c.eta[0] = b.er; c.eta[1] = b.eg; c.eta[2] = b.eb;
    c.k[0] = b.kr; c.k[1] = b.kg; c.k[2] = b.kb;
    return c;
}

MTS_DEV float plastic_diffuse(const DevBsdf &b, float refl) {        // plastic.cpp:233-234,260-261
    return refl / (1.0f - ((b.flags & kBsdfNonlinear) ? (refl * b.eb) : b.eb));
}

template <int N>
MTS_DEV void bsdf_eval_pdf_n(const DevBsdf &b, const BsdfChannels<N> &c, f3 wi, f3 wo, float (&value)[N], float &pdf);

// BSDF::sample.  Returns false (weight 0) for an invalid sample.
template <int N>
MTS_DEV bool bsdf_sample_n(const DevBsdf &b, const BsdfChannels<N> &c, f3 wi, float sample1, f2 sample2, BsdfSample &bs, float (&weight)[N]) {
    bs.wo = mk3(0.0f, 0.0f, 0.0f); bs.pdf = 0.0f; bs.eta = 0.0f; bs.delta = false;
#pragma unroll
    for (int i = 0; i < N; ++i) weight[i] = 0.0f;
    const bool two = (b.flags & kBsdfTwoSided) != 0u;
    if (two && wi.z == 0.0f) return false;
    const bool flip = two && wi.z < 0.0f;
    if (flip) wi.z = -wi.z;
    bool ok = false;
    if (b.type == kBsdfDiffuse) {
        bs.eta = 1.0f;
        if (wi.z > 0.0f) {
            bs.wo = square_to_cosine_hemisphere(sample2);
            bs.pdf = kInvPi * bs.wo.z;
            if (bs.pdf > 0.0f) {
#pragma unroll
                for (int i = 0; i < N; ++i) weight[i] = c.refl[i];
                ok = true;
            }
        }
    } else if (b.type == kBsdfConductor) {
        if (wi.z > 0.0f) {
            bs.wo = reflect_z(wi); bs.eta = 1.0f; bs.pdf = 1.0f; bs.delta = true;
#pragma unroll
            for (int i = 0; i < N; ++i) weight[i] = c.spec[i] * fresnel_conductor(wi.z, c.eta[i], c.k[i]);
            ok = true;
        }
    } else if (b.type == kBsdfRoughConductor) {
        const float cos_theta_i = wi.z;
        if (cos_theta_i > 0.0f) {
            const bool vis = (b.flags & kBsdfSampleVisible) != 0u;
            const Mdf d = mdf_make((b.flags & kBsdfGGX) != 0u, b.alpha_u, b.alpha_v, vis);
            const f3 m = mdf_sample(d, wi, sample2, bs.pdf);
            bs.wo = reflect_m(wi, m); bs.eta = 1.0f;
            const bool active = bs.pdf != 0.0f && bs.wo.z > 0.0f;
            float w;
            if (vis) w = mdf_smith_g1(d, bs.wo, m);
            else w = mdf_G(d, wi, bs.wo, m) * dot(wi, m) / (cos_theta_i * m.z);
            bs.pdf /= 4.0f * dot(bs.wo, m);
            const float dwm = dot(wi, m);
            if (active) {
#pragma unroll
                for (int i = 0; i < N; ++i) weight[i] = fresnel_conductor(dwm, c.eta[i], c.k[i]) * (w * c.spec[i]);
            }
            ok = active;
        }
    } else if (b.type == kBsdfDielectric) {
        const Fresnel f = fresnel(wi.z, b.er);
        const float r_i = f.r, t_i = 1.0f - r_i;
        const bool selected_r = sample1 <= r_i;
        bs.pdf = selected_r ? r_i : t_i;
        bs.wo = selected_r ? reflect_z(wi) : mk3(-f.eta_ti * wi.x, -f.eta_ti * wi.y, f.cos_theta_t);
        bs.eta = selected_r ? 1.0f : f.eta_it;
        bs.delta = true;
        const float q = sqr(f.eta_ti);
#pragma unroll
        for (int i = 0; i < N; ++i) weight[i] = selected_r ? 1.0f * c.spec[i] : (1.0f * c.trans[i]) * q;
        ok = true;
    } else if (b.type == kBsdfThinDielectric) {                // thindielectric.cpp:100-148
        float r = fresnel(fabsf(wi.z), b.er).r;
        r *= 2.0f / (1.0f + r);                                // internal reflections: r' = r + trt + tr^3t + ..
        const float t = 1.0f - r;
        const bool selected_r = sample1 <= r;
        bs.pdf = selected_r ? r : t;
        bs.wo = selected_r ? reflect_z(wi) : mk3(-wi.x, -wi.y, -wi.z);
        bs.eta = 1.0f;
        bs.delta = true;                                       // DeltaReflection or Null: both are in BSDFFlags::Delta (bsdf.h:117)
#pragma unroll
        for (int i = 0; i < N; ++i) weight[i] = selected_r ? 1.0f * c.spec[i] : 1.0f * c.trans[i];
        ok = true;
    } else if (b.type == kBsdfPlastic) {
        const float cos_theta_i = wi.z;
        if (cos_theta_i > 0.0f) {
            const float f_i = fresnel(cos_theta_i, b.er).r;
            float prob_specular = f_i * b.kr, prob_diffuse = (1.0f - f_i) * (1.0f - b.kr);
            prob_specular = prob_specular / (prob_specular + prob_diffuse);
            prob_diffuse = 1.0f - prob_specular;
            bs.eta = 1.0f;
            if (sample1 < prob_specular) {
                bs.wo = reflect_z(wi); bs.pdf = prob_specular; bs.delta = true;
                const float v = f_i / bs.pdf;
#pragma unroll
                for (int i = 0; i < N; ++i) weight[i] = v * c.spec[i];
            } else {
                bs.wo = square_to_cosine_hemisphere(sample2);
                bs.pdf = prob_diffuse * (kInvPi * bs.wo.z);
                const float f_o = fresnel(bs.wo.z, b.er).r;
                const float k = b.eg * (1.0f - f_i) * (1.0f - f_o) / prob_diffuse;
#pragma unroll
                for (int i = 0; i < N; ++i) weight[i] = plastic_diffuse(b, c.refl[i]) * k;
            }
            ok = true;
        }
    } else if (b.type == kBsdfRoughPlastic) {                  // roughplastic.cpp:180-232
        const float cos_theta_i = wi.z;
        if (cos_theta_i > 0.0f) {
            const float t_i = lerp_gather(b.table, cos_theta_i, kRoughTableRes);
            float prob_specular = (1.0f - t_i) * b.kr;
            const float prob_diffuse = t_i * (1.0f - b.kr);
            prob_specular = prob_specular / (prob_specular + prob_diffuse);
            bs.eta = 1.0f;
            if (sample1 < prob_specular) {
                const Mdf d = mdf_make((b.flags & kBsdfGGX) != 0u, b.alpha_u, b.alpha_u, (b.flags & kBsdfSampleVisible) != 0u);
                float unused;
                const f3 m = mdf_sample(d, wi, sample2, unused);
                bs.wo = reflect_m(wi, m);
            } else {
                bs.wo = square_to_cosine_hemisphere(sample2);
            }
            DevBsdf one_sided = b;
            one_sided.flags &= ~kBsdfTwoSided;               // wi is already on the front side
            float value[N];
            bsdf_eval_pdf_n<N>(one_sided, c, wi, bs.wo, value, bs.pdf);
            if (bs.pdf > 0.0f) {
#pragma unroll
                for (int i = 0; i < N; ++i) weight[i] = value[i] / bs.pdf;
                ok = true;
            }
        }
    } else if (b.type == kBsdfRoughDielectric) {               // roughdielectric.cpp:202-300 (both lobes enabled, radiance transport)
        const float cos_theta_i = wi.z;
        bool active = cos_theta_i != 0.0f;                     // perfectly grazing configurations are ignored
        const bool vis = (b.flags & kBsdfSampleVisible) != 0u;
        const Mdf d = mdf_make((b.flags & kBsdfGGX) != 0u, b.alpha_u, b.alpha_v, vis);
        Mdf sd = d;
        if (!vis) {                                            // Walter et al.'s trick: widen the sampling distribution
            const float sc = 1.2f - 0.2f * sqrtf(fabsf(cos_theta_i));
            sd.au *= sc; sd.av *= sc;
        }
        const f3 wi_up = mk3(mulsign(wi.x, cos_theta_i), mulsign(wi.y, cos_theta_i), mulsign(wi.z, cos_theta_i));
        const f3 m = mdf_sample(sd, wi_up, sample2, bs.pdf);
        active = active && bs.pdf != 0.0f;
        const Fresnel f = fresnel(dot(wi, m), b.er);
        const bool selected_r = sample1 <= f.r && active, selected_t = !selected_r && active;
        bs.pdf *= selected_r ? f.r : 1.0f - f.r;
        bs.eta = selected_r ? 1.0f : f.eta_it;
        float dwh_dwo = 0.0f;
        if (selected_r) {
            bs.wo = reflect_m(wi, m);
            dwh_dwo = rcp(4.0f * dot(bs.wo, m));
        }
        if (selected_t) {
            bs.wo = refract_m(wi, m, f.cos_theta_t, f.eta_ti);
            dwh_dwo = (sqr(bs.eta) * dot(bs.wo, m)) / sqr(dot(wi, m) + bs.eta * dot(bs.wo, m));
        }
        float g;
        if (vis) g = mdf_smith_g1(d, bs.wo, m);
        else g = mdf_G(d, wi, bs.wo, m) * dot(wi, m) / (cos_theta_i * m.z);
        bs.pdf *= fabsf(dwh_dwo);
        const float q = sqr(f.eta_ti);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float wk = 1.0f;
            if (selected_r) wk *= c.spec[i];
            if (selected_t) wk *= q * c.trans[i];
            weight[i] = wk * g;
        }
        ok = active;
    }
    if (!ok) {
#pragma unroll
        for (int i = 0; i < N; ++i) weight[i] = 0.0f;
    }
    if (flip) bs.wo.z = -bs.wo.z;
    return ok;
}

// BSDF::eval and BSDF::pdf
template <int N>
MTS_DEV void bsdf_eval_pdf_n(const DevBsdf &b, const BsdfChannels<N> &c, f3 wi, f3 wo, float (&value)[N], float &pdf) {
#pragma unroll
    for (int i = 0; i < N; ++i) value[i] = 0.0f;
    pdf = 0.0f;
    if (b.flags & kBsdfTwoSided) {
        if (wi.z == 0.0f) return;
        if (wi.z < 0.0f) { wi.z = -wi.z; wo.z = -wo.z; }
    }
    const float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (b.type == kBsdfRoughDielectric) {                      // roughdielectric.cpp:302-375 (eval), :377-447 (pdf)
        if (cos_theta_i == 0.0f) return;
        const bool reflect = cos_theta_i * cos_theta_o > 0.0f;
        const float m_inv_eta = 1.0f / b.er;                   // parameters_changed(): roughdielectric.cpp:198-200
        const float eta = cos_theta_i > 0.0f ? b.er : m_inv_eta, inv_eta = cos_theta_i > 0.0f ? m_inv_eta : b.er;
        f3 m = normalize(wi + wo * (reflect ? 1.0f : eta));
        m = mk3(mulsign(m.x, m.z), mulsign(m.y, m.z), mulsign(m.z, m.z));
        const bool vis = (b.flags & kBsdfSampleVisible) != 0u;
        const Mdf d = mdf_make((b.flags & kBsdfGGX) != 0u, b.alpha_u, b.alpha_v, vis);
        const float D = mdf_eval(d, m);
        const float F = fresnel(dot(wi, m), b.er).r, G = mdf_G(d, wi, wo, m);
        const float dwm = dot(wi, m), dom = dot(wo, m);
        if (reflect) {
            const float v = F * D * G / (4.0f * fabsf(cos_theta_i));
#pragma unroll
            for (int i = 0; i < N; ++i) value[i] = v * c.spec[i];
        } else {
            const float scale = sqr(inv_eta);                  // radiance transport: solid angle compression
            const float v = fabsf((scale * (1.0f - F) * D * G * eta * eta * dwm * dom) / (cos_theta_i * sqr(dwm + eta * dom)));
#pragma unroll
            for (int i = 0; i < N; ++i) value[i] = v * c.trans[i];
        }
        const bool active = dwm * cos_theta_i > 0.0f && dom * cos_theta_o > 0.0f;
        const float dwh_dwo = reflect ? rcp(4.0f * dom) : (eta * eta * dom) / sqr(dwm + eta * dom);
        Mdf sd = d;
        if (!vis) {
            const float sc = 1.2f - 0.2f * sqrtf(fabsf(cos_theta_i));
            sd.au *= sc; sd.av *= sc;
        }
        const f3 wi_up = mk3(mulsign(wi.x, cos_theta_i), mulsign(wi.y, cos_theta_i), mulsign(wi.z, cos_theta_i));
        float prob = mdf_pdf(sd, wi_up, m);
        prob *= reflect ? F : 1.0f - F;
        pdf = active ? prob * fabsf(dwh_dwo) : 0.0f;
        return;
    }
    if (!(cos_theta_i > 0.0f && cos_theta_o > 0.0f)) return;      // every reflective model here is one-sided
    if (b.type == kBsdfDiffuse) {
#pragma unroll
        for (int i = 0; i < N; ++i) value[i] = (c.refl[i] * kInvPi) * wo.z;
        pdf = kInvPi * wo.z;
    } else if (b.type == kBsdfRoughConductor) {
        const f3 H = normalize(wo + wi);
        const bool vis = (b.flags & kBsdfSampleVisible) != 0u;
        const Mdf d = mdf_make((b.flags & kBsdfGGX) != 0u, b.alpha_u, b.alpha_v, vis);
        const float D = mdf_eval(d, H);
        if (D != 0.0f) {
            const float G = mdf_G(d, wi, wo, H);
            const float result = D * G / (4.0f * cos_theta_i);
            const float dwh = dot(wi, H);
#pragma unroll
            for (int i = 0; i < N; ++i) value[i] = fresnel_conductor(dwh, c.eta[i], c.k[i]) * (result * c.spec[i]);
        }
        if (dot(wi, H) > 0.0f && dot(wo, H) > 0.0f) {
            if (vis) pdf = mdf_eval(d, H) * mdf_smith_g1(d, wi, H) / (4.0f * cos_theta_i);
            else pdf = mdf_pdf(d, wi, H) / (4.0f * dot(wo, H));
        }
    } else if (b.type == kBsdfPlastic) {
        const float f_i = fresnel(cos_theta_i, b.er).r, f_o = fresnel(cos_theta_o, b.er).r;
        const float k2 = (kInvPi * wo.z) * b.eg * (1.0f - f_i) * (1.0f - f_o);
#pragma unroll
        for (int i = 0; i < N; ++i) value[i] = plastic_diffuse(b, c.refl[i]) * k2;
        const float prob_specular = f_i * b.kr;
        float prob_diffuse = (1.0f - f_i) * (1.0f - b.kr);
        prob_diffuse = prob_diffuse / (prob_specular + prob_diffuse);
        pdf = (kInvPi * wo.z) * prob_diffuse;
    } else if (b.type == kBsdfRoughPlastic) {                  // roughplastic.cpp:234-289 (eval), :304-352 (pdf)
        const bool vis = (b.flags & kBsdfSampleVisible) != 0u;
        const Mdf d = mdf_make((b.flags & kBsdfGGX) != 0u, b.alpha_u, b.alpha_u, vis);
        const f3 H = normalize(wo + wi);
        const float D = mdf_eval(d, H);
        const float F = fresnel(dot(wi, H), b.er).r;
        const float G = mdf_G(d, wi, wo, H);
        const float spec_v = F * D * G / (4.0f * cos_theta_i);
        const float t_i = lerp_gather(b.table, cos_theta_i, kRoughTableRes), t_o = lerp_gather(b.table, cos_theta_o, kRoughTableRes);
        const float kd = kInvPi * b.eg * cos_theta_o * t_i * t_o;
#pragma unroll
        for (int i = 0; i < N; ++i) value[i] = spec_v * c.spec[i] + plastic_diffuse(b, c.refl[i]) * kd;
        float prob_specular = (1.0f - t_i) * b.kr, prob_diffuse = t_i * (1.0f - b.kr);
        prob_specular = prob_specular / (prob_specular + prob_diffuse);
        prob_diffuse = 1.0f - prob_specular;
        float result;
        if (vis) result = mdf_eval(d, H) * mdf_smith_g1(d, wi, H) / (4.0f * cos_theta_i);
        else result = mdf_pdf(d, wi, H) / (4.0f * dot(wo, H));
        result *= prob_specular;
        result += prob_diffuse * (kInvPi * wo.z);
        pdf = result;
    }
}

// Tables of RoughPlastic::parameters_changed (roughplastic.cpp:380-399, microfacet.h:462-553) for one incident cosine:
// Gauss-Legendre quadrature over the visible-normal sampling domain of the transmitted / internally reflected energy.
MTS_DEV float rough_transmittance(const Mdf &d, f3 wi, float eta, int res, const float *nodes, const float *weights) {
    float accum = 0.0f;
    for (int a = 0; a < res; ++a)
        for (int b = 0; b < res; ++b) {
            f2 node; node.x = fmaf(nodes[b], 0.5f, 0.5f); node.y = fmaf(nodes[a], 0.5f, 0.5f);
            float pdf;
            const f3 m = mdf_sample(d, wi, node, pdf);
            const Fresnel f = fresnel(dot(wi, m), eta);
            const f3 wo = refract_m(wi, m, f.cos_theta_t, f.eta_ti);
            float smith = mdf_smith_g1(d, wo, m) * (1.0f - f.r);
            if (wo.z * wi.z >= 0.0f) smith = 0.0f;
            accum += smith * (weights[b] * weights[a]);
        }
    return accum * 0.25f;
}
MTS_DEV float rough_reflectance(const Mdf &d, f3 wi, float eta, int res, const float *nodes, const float *weights) {
    float accum = 0.0f;
    for (int a = 0; a < res; ++a)
        for (int b = 0; b < res; ++b) {
            f2 node; node.x = fmaf(nodes[b], 0.5f, 0.5f); node.y = fmaf(nodes[a], 0.5f, 0.5f);
            float pdf;
            const f3 m = mdf_sample(d, wi, node, pdf);
            const f3 wo = reflect_m(wi, m);
            const Fresnel f = fresnel(dot(wi, m), eta);
            float smith = mdf_smith_g1(d, wo, m) * f.r;
            if (wo.z <= 0.0f || wi.z <= 0.0f) smith = 0.0f;
            accum += smith * (weights[b] * weights[a]);
        }
    return accum * 0.25f;
}

// ---------------------------------------------------------------------------------------------
// blendbsdf (blendbsdf.cpp:82-179) and mask (mask.cpp:92-172) over plain child records, and the plain BSDFs themselves, behind one
// entry point: the record that is actually sampled / evaluated is resolved first (a child, or `b` itself), so the model code above
// is instantiated once.  `table(i)` returns record i of the scene's BSDF table, `chan_of(rec)` the per-channel inputs of a child
// (its own constant parameters); `c` are the inputs of `b` -- for a blend / mask c.refl[0..2] carries what Texture::eval_1 of the
// weight / opacity reads (the constant, or the texture lookup).
// What the entry points below keep of a blend / mask record; kind = 0: a plain BSDF.  The record variable of the caller is working
// storage: for a nest it is overwritten with the child that is evaluated / sampled, so only ONE 128-byte record is ever live (two of
// them -- parent and child -- cost the general kernels 200+ bytes of scratch per lane and 20 % of their speed).  NEST = false compiles
// all of it out: only the kernels of the schedule that scenes with a blend / mask run (k_bounce*, k_direct) carry the nesting code.
struct NestInfo { int kind; uint32_t n0, n1; float w; bool two; };
// `value` = what Texture::eval_1 of the weight / opacity reads: the constant, or the texture lookup (first three channels)
template <bool NEST>
MTS_DEV NestInfo nest_info(const DevBsdf &b, float v0, float v1, float v2) {
    NestInfo ni = { 0, 0u, 0u, 1.0f, false };
    if (NEST && b.type >= kBsdfBlend) {
        float w = v0;
        if (b.flags & kBsdfWeightLum) w = fmaf(0.072169f, v2, fmaf(0.715160f, v1, 0.212671f * v0));      // luminance (spectrum.h:239-241)
        ni.kind = b.type; ni.n0 = b.nested0; ni.n1 = b.nested1; ni.two = (b.flags & kBsdfTwoSided) != 0u;
        ni.w = fminf(fmaxf(w, 0.0f), 1.0f);                  // blendbsdf.cpp:177-179, mask.cpp:170-172
    }
    return ni;
}
// `table(i)` returns record i of the scene's BSDF table; `chan_of(rec, child)` the per-channel inputs of the record that is used: the
// caller's own (child = false) or those of a child record (its constant parameters).
template <bool NEST, int N, typename Table, typename ChanOf>
MTS_DEV bool surface_bsdf_sample(DevBsdf &cur, const NestInfo &ni, const Table &table, const ChanOf &chan_of, f3 wi, float sample1,
                                 f2 sample2, BsdfSample &bs, float (&weight)[N]) {
    float s1 = sample1;
    bool model = true, flip = false;
    if (NEST && ni.kind != 0) {
        bs.wo = mk3(0.0f, 0.0f, 0.0f); bs.pdf = 0.0f; bs.eta = 0.0f; bs.delta = false;
#pragma unroll
        for (int i = 0; i < N; ++i) weight[i] = 0.0f;
        if (ni.two && wi.z == 0.0f) return false;            // twosided.cpp:94-123 around the whole nest
        flip = ni.two && wi.z < 0.0f;
        if (flip) wi.z = -wi.z;
        const float w = ni.w;
        uint32_t child = ni.n0;
        if (ni.kind == kBsdfBlend) {
            if (sample1 > w) s1 = (sample1 - w) / (1.0f - w);
            else if (sample1 <= w) { child = ni.n1; s1 = sample1 / w; }
            else model = false;                              // NaN sample: neither mask of blendbsdf.cpp:108-109
        } else {
            model = sample1 < w;
            s1 = sample1 / w;
        }
        if (model) cur = table(child);
    }
    bool ok = false;
    if (model) ok = bsdf_sample_n<N>(cur, chan_of(cur, NEST && ni.kind != 0), wi, s1, sample2, bs, weight);      // the one call site of the model code
    else if (NEST && ni.kind == kBsdfMask) {                 // the null lobe: straight through (mask.cpp:116-121); Null is part of Delta
        bs.wo = mk3(-wi.x, -wi.y, -wi.z); bs.eta = 1.0f; bs.pdf = 1.0f - ni.w; bs.delta = true;
#pragma unroll
        for (int i = 0; i < N; ++i) weight[i] = 1.0f;
        ok = true;
    }
    if (flip) bs.wo.z = -bs.wo.z;
    return ok;
}
template <bool NEST, int N, typename Table, typename ChanOf>
MTS_DEV void surface_bsdf_eval_pdf(DevBsdf &cur, const NestInfo &ni, const Table &table, const ChanOf &chan_of, f3 wi, f3 wo,
                                   float (&value)[N], float &pdf) {
    const bool nest = NEST && ni.kind != 0, blend = nest && ni.kind == kBsdfBlend;
    const float w = ni.w;
    if (nest) {
#pragma unroll
        for (int i = 0; i < N; ++i) value[i] = 0.0f;
        pdf = 0.0f;
        if (ni.two) {
            if (wi.z == 0.0f) return;
            if (wi.z < 0.0f) { wi.z = -wi.z; wo.z = -wo.z; }
        }
    }
    float v0[N], p0 = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) v0[i] = 0.0f;
#pragma unroll 1
    for (uint32_t k = 0; k < (blend ? 2u : 1u); ++k) {      // one call site of the model code: the record itself, or its children in turn
        if (nest) cur = table(k == 0u ? ni.n0 : ni.n1);
        float v[N], p;
        bsdf_eval_pdf_n<N>(cur, chan_of(cur, nest), wi, wo, v, p);
        if (!nest) {                                         // plain BSDF: unchanged values
#pragma unroll
            for (int i = 0; i < N; ++i) value[i] = v[i];
            pdf = p;
        } else if (!blend) {                                 // mask.cpp:133-159
#pragma unroll
            for (int i = 0; i < N; ++i) value[i] = v[i] * w;
            pdf = p * w;
        } else if (k == 0u) {
#pragma unroll
            for (int i = 0; i < N; ++i) v0[i] = v[i];
            p0 = p;
        } else {                                             // eval0 * (1 - weight) + eval1 * weight (blendbsdf.cpp:141-142,157-158)
#pragma unroll
            for (int i = 0; i < N; ++i) value[i] = fmaf(v[i], w, v0[i] * (1.0f - w));
            pdf = fmaf(p, w, p0 * (1.0f - w));
        }
    }
}

// RGB variant
// `refl_of(rec)`: reflectance of a child record at the surface point (its constant, or its texture: RGB variant)
template <bool NEST, typename Table, typename ReflOf>
MTS_DEV bool surface_bsdf_sample(DevBsdf &cur, const NestInfo &ni, f3 refl, const Table &table, const ReflOf &refl_of, f3 wi, float sample1, f2 sample2,
                                 BsdfSample &bs, f3 &weight) {
    float w[3];
    auto chan_of = [&](const DevBsdf &rec, bool child) { return rgb_channels(rec, child ? refl_of(rec) : refl); };
    const bool ok = surface_bsdf_sample<NEST, 3>(cur, ni, table, chan_of, wi, sample1, sample2, bs, w);
    weight = mk3(w[0], w[1], w[2]);
    return ok;
}
template <bool NEST, typename Table, typename ReflOf>
MTS_DEV void surface_bsdf_eval_pdf(DevBsdf &cur, const NestInfo &ni, f3 refl, const Table &table, const ReflOf &refl_of, f3 wi, f3 wo, f3 &value, float &pdf) {
    float v[3];
    auto chan_of = [&](const DevBsdf &rec, bool child) { return rgb_channels(rec, child ? refl_of(rec) : refl); };
    surface_bsdf_eval_pdf<NEST, 3>(cur, ni, table, chan_of, wi, wo, v, pdf);
    value = mk3(v[0], v[1], v[2]);
}
MTS_DEV bool bsdf_sample(const DevBsdf &b, f3 refl, f3 wi, float sample1, f2 sample2, BsdfSample &bs, f3 &weight) {
    float w[3];
    const bool ok = bsdf_sample_n<3>(b, rgb_channels(b, refl), wi, sample1, sample2, bs, w);
    weight = mk3(w[0], w[1], w[2]);
    return ok;
}
MTS_DEV void bsdf_eval_pdf(const DevBsdf &b, f3 refl, f3 wi, f3 wo, f3 &value, float &pdf) {
    float v[3];
    bsdf_eval_pdf_n<3>(b, rgb_channels(b, refl), wi, wo, v, pdf);
    value = mk3(v[0], v[1], v[2]);
}

} // namespace mtsamd
