// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the wavefront path tracer.
//
//   k_bounce        one path segment per in-flight path: BVH closest hit, surface interaction,
//                   emitter hit + MIS, Russian roulette, emitter sampling + BVH any hit, BSDF
//                   sampling, then ballot/prefix-sum compaction of the surviving paths into the
//                   wave's output segment and regeneration of camera paths into the free slots
//                   (PathIntegrator::sample, src/integrators/path.cpp:100-211; render_sample,
//                   src/librender/integrator.cpp:224-271)
//   k_film_gather   ImageBlock::put as a deterministic per-pixel gather over the per-sample
//                   radiance stream (src/librender/imageblock.cpp:80-172)
//   k_ray_intersect / k_ray_test   Scene::ray_intersect / ray_test on SoA ray streams
//   k_camera_rays, k_imageblock_put, k_put_block, k_film_develop
//
// Scheduling: the in-flight paths live in per-wave segments of the SoA pool.  A scheduling wave
// owns `seg_cap` slots and a private range of sample ordinals; no atomics and no inter-workgroup
// traffic are needed, so the result is independent of dispatch order.
#include "kernels.h"

namespace mtsamd {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
struct PathState {
    f3 o, d; float mint, maxt;
    f3 thr; float bs_pdf;
    f3 res; float eta;
    Pcg32 rng;
    uint32_t ordinal, depth, flags;
};

// NT: the pool is a once-read / once-written stream (hierarchy scenes: keep it out of the caches the BVH lives in)
template <bool NT = false>
MTS_DEV void load_state(const PoolView &p, size_t i, PathState &s) {
    float4 a = ld_stream<NT>(p.ray_o + i), b = ld_stream<NT>(p.ray_d + i), c = ld_stream<NT>(p.thr + i), e = ld_stream<NT>(p.res + i);
    uint4 r = ld_stream<NT>(p.rng + i); uint2 m = ld_stream<NT>(p.misc + i);
    s.o = mk3(a.x, a.y, a.z); s.mint = a.w;
    s.d = mk3(b.x, b.y, b.z); s.maxt = b.w;
    s.thr = mk3(c.x, c.y, c.z); s.bs_pdf = c.w;
    s.res = mk3(e.x, e.y, e.z); s.eta = e.w;
    s.rng.state = (uint64_t) r.x | ((uint64_t) r.y << 32);
    s.rng.inc = (uint64_t) r.z | ((uint64_t) r.w << 32);
    s.ordinal = m.x; s.depth = m.y & 0xffffu; s.flags = m.y >> 16;
}
template <bool NT = false>
MTS_DEV void store_state(const PoolView &p, size_t i, const PathState &s) {
    st_stream<NT>(p.ray_o + i, make_float4(s.o.x, s.o.y, s.o.z, s.mint));
    st_stream<NT>(p.ray_d + i, make_float4(s.d.x, s.d.y, s.d.z, s.maxt));
    st_stream<NT>(p.thr + i, make_float4(s.thr.x, s.thr.y, s.thr.z, s.bs_pdf));
    p.res[i] = make_float4(s.res.x, s.res.y, s.res.z, s.eta);      // read-modify-written by the shadow-ray stage: default policy
    st_stream<NT>(p.rng + i, make_uint4((uint32_t) s.rng.state, (uint32_t) (s.rng.state >> 32), (uint32_t) s.rng.inc,
                                        (uint32_t) (s.rng.inc >> 32)));
    st_stream<NT>(p.misc + i, make_uint2(s.ordinal, (s.depth & 0xffffu) | (s.flags << 16)));
}

struct Counters { uint32_t closest, any, segments, tri_tests; };

// What the adjoint pass remembers about one path vertex k.  With T_k the throughput arriving at the vertex,
//   radiance += T_k * E_k;  T'_k = T_k * invq_k (Russian roulette);  radiance += T'_k * rho_k * Nc_k (next-event
//   estimation);  T_{k+1} = T'_k * rho_k (diffuse BSDF sample weight).
struct VertexRec {
    f3 E, Nc, Tp, rho; float invq;
    f3 T; int32_t rr_channel;       // throughput before Russian roulette; channel that sets q (-1: none / q clamped)
    uint32_t texel; f2 w1; int32_t bsdf; uint32_t has_bsdf;
    // radiance-free coefficients for d/d(emitter radiance): E = ew * Le[em_hit], Nc = nk * Le[em_nee] (-1: none)
    float ew, nk; int32_t em_hit, em_nee;
};

// Split ("wavefront") pipeline: the two ray queries of a segment run in their own kernels.  `hit` / `found` carry the
// closest hit computed by k_trace<false> into the shading step; the shadow ray and the contribution it guards are
// handed back for k_trace<true>, which adds `nee` to the path's radiance if the ray is unoccluded.
struct Deferred {
    Hit hit; bool found;
    bool pending; f3 so, sd; float smint, smaxt; float nee[4];
};

// One iteration of the path.cpp loop, rotated so that it starts with the intersection of the
// ray spawned by the previous iteration (or by the sensor).  Returns true if the path survives.
// GENERAL = false: every BSDF is a one-sided `diffuse` (the code path of the Cornell-box benchmark, unchanged);
// GENERAL = true: switch over the BSDF models of device_bsdf.h (delta lobes, eta, twosided).
constexpr uint32_t kFlagDelta = 4u;       // the ray was spawned by a delta lobe: no emitter-sampling counterpart (path.cpp:198-203)

// DEFER: 0 = both ray queries inline (fused kernel); 1 = closest hit precomputed + shadow ray queued (split pipeline of
// hierarchy scenes); 2 = closest hit inline, shadow ray queued (flat scenes: the any-hit loop then runs on dense batches)
// ENVGRAD (k_adjoint_env): besides the radiance, d(loss)/d(envmap texels) = delta * d(radiance)/d(texels) is scattered into `grad`
// (h * w * 3) -- the radiance is linear in the texels at its two uses, the emission an escaped ray picks up and the emitter sample;
// the sampling distribution built from their luminances is not differentiated (envmap.cpp:220-253 rebuilds it from plain floats)
struct EnvGradCtx { f3 delta; float *grad; };
MTS_DEV void env_grad_add(const DevEnvmap &e, float u, float v, f3 coeff, const EnvGradCtx &eg) {
    u *= (float) (e.w - 1); v *= (float) (e.h - 1);                  // the bilinear footprint of envmap_lookup
    const uint32_t px = min((uint32_t) u, (uint32_t) (e.w - 2)), py = min((uint32_t) v, (uint32_t) (e.h - 2));
    const float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const float wt[4] = { (w0y * w0x) * e.scale, (w0y * w1x) * e.scale, (w1y * w0x) * e.scale, (w1y * w1x) * e.scale };
    const uint32_t idx[4] = { py * (uint32_t) e.w + px, py * (uint32_t) e.w + px + 1u, (py + 1u) * (uint32_t) e.w + px, (py + 1u) * (uint32_t) e.w + px + 1u };
    const f3 c = mk3(eg.delta.x * coeff.x, eg.delta.y * coeff.y, eg.delta.z * coeff.z);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float *g = eg.grad + 3u * (size_t) idx[i];
        atomicAdd(g, c.x * wt[i]); atomicAdd(g + 1, c.y * wt[i]); atomicAdd(g + 2, c.z * wt[i]);
    }
}

// PGRAD (k_adjoint_param): derivative of the path's radiance w.r.t. ONE scalar parameter of ONE BSDF record of any model (roughness,
// complex IOR, reflectances ...), carried forward beside the path as a dual part (dthr = d throughput, dres = d radiance).  Sampling is
// DETACHED: the replay takes the decisions and directions of the primal path (same PCG32 stream, same parameter value), and
// differentiates what depends on the parameter for fixed directions -- the BSDF value f(wi, wo) cos in the emitter-sampling term and in
// the sample weight f cos / pdf (pdf, lobe probabilities, MIS weights and Russian-roulette probabilities are held fixed: any fixed
// partition of unity keeps the estimator unbiased).  d f / d theta at fixed (wi, wo) is a central difference of the model code itself
// between two records perturbed by +-h (bp / bm): a smooth closed form at fixed arguments, O(h^2) truncation, no decision can flip.
// The reference differentiates the attached estimator through Enoki's graph (src/python/python/autodiff.py:6-91); both estimate the same
// derivative of the image.
struct ParamGradCtx { int32_t bsdf; DevBsdf bp, bm; float inv_2h; f3 dthr, dres; };

template <bool FLAT, bool REC = false, int DEFER = 0, bool GENERAL = false, bool ENVGRAD = false, bool NEST = false, bool PGRAD = false>
MTS_DEV bool bounce_step(const RenderParams &P, const LdsView &lds, PathState &s, Counters &c, VertexRec *rec = nullptr,
                         Deferred *df = nullptr, const EnvGradCtx *eg = nullptr, ParamGradCtx *pg = nullptr) {
    static_assert(!PGRAD || (GENERAL && DEFER == 0 && !ENVGRAD && !NEST && !REC), "the parameter gradient rides on the general fused step");
    static_assert(!(REC && GENERAL), "the adjoint replay handles diffuse BSDFs only");
    static_assert(!ENVGRAD || (GENERAL && DEFER == 0), "the envmap gradient rides on the general fused step");
    static_assert(!NEST || (GENERAL && DEFER == 0 && !ENVGRAD), "blendbsdf / mask run the general fused step");
    const SceneView &sv = P.sv;
    if (REC) {
        rec->E = rec->Nc = rec->Tp = rec->rho = mk3(0.0f, 0.0f, 0.0f);
        rec->invq = 1.0f; rec->rr_channel = -1; rec->T = s.thr; rec->texel = kNoPrim; rec->w1.x = rec->w1.y = 0.0f; rec->bsdf = -1; rec->has_bsdf = 0u;
        rec->ew = rec->nk = 0.0f; rec->em_hit = rec->em_nee = -1;
    }
    const Geo<FLAT> geo{ sv, lds };
    Hit hit;
    ++c.closest; ++c.segments;
    bool found;
    if (DEFER) df->pending = false;
    if (DEFER == 1) { hit = df->hit; found = df->found; }
    else found = traverse<FLAT, false>(sv, lds, s.o, s.d, s.mint, s.maxt, hit, c.tri_tests,
                                       FLAT && __ballot(s.depth != 1u) == 0ull);      // all active lanes carry camera rays: cluster culling
    if (s.depth == 1u) s.flags = found ? 1u : 0u;          // valid_ray (path.cpp:121)

    SurfaceInteraction si;
    if (found) {
        fill_si(geo, s.d, hit.prim, hit.u, hit.v, si);
        int32_t emitter = si.shape_rec.emitter;
        if (emitter >= 0) {
            const DevEmitter e = geo.emitter((uint32_t) emitter);
            // emission_weight of the previous iteration (path.cpp:194-205); 1 for camera rays
            float ew = 1.0f;
            if (s.depth > 1u) {
                f3 dd = si.p - s.o;                         // DirectionSample(si_bsdf, si), records.h:168-174
                float dist = sqrtf(sqnorm(dd));
                dd = div_s(dd, dist);
                const float pe = pdf_emitter_direction(sv.n_emitters, e.area_norm, dd, si.sh.n, dist);
                ew = mis_weight(s.bs_pdf, (GENERAL && (s.flags & kFlagDelta)) ? 0.0f : pe);
            }
            if (si.wi.z > 0.0f) {                           // AreaLight::eval (area.cpp:71-79)
                s.res.x += (ew * s.thr.x) * e.r; s.res.y += (ew * s.thr.y) * e.g; s.res.z += (ew * s.thr.z) * e.b;
                if (PGRAD) pg->dres = mk3(pg->dres.x + (ew * pg->dthr.x) * e.r, pg->dres.y + (ew * pg->dthr.y) * e.g, pg->dres.z + (ew * pg->dthr.z) * e.b);
                if (REC) { rec->E = mk3(ew * e.r, ew * e.g, ew * e.b); rec->ew = ew; rec->em_hit = emitter; }
            }
        }
    }
    if (GENERAL && !REC && !found && sv.env_emitter >= 0) {  // si.emitter(scene) of an escaped ray: the environment
        const DevEmitter e = geo.emitter((uint32_t) sv.env_emitter);
        float ew = 1.0f;
        if (s.depth > 1u) ew = mis_weight(s.bs_pdf, (GENERAL && (s.flags & kFlagDelta)) ? 0.0f : pdf_environment(sv, e, s.d));
        const f3 le = environment_radiance(sv, e, s.d);
        s.res.x += (ew * s.thr.x) * le.x; s.res.y += (ew * s.thr.y) * le.y; s.res.z += (ew * s.thr.z) * le.z;
        if (PGRAD) pg->dres = mk3(pg->dres.x + (ew * pg->dthr.x) * le.x, pg->dres.y + (ew * pg->dthr.y) * le.y, pg->dres.z + (ew * pg->dthr.z) * le.z);
        if (ENVGRAD && e.pad0 == kEmitterEnvmap) {
            float u, v;
            env_dir_to_uv(mat3_apply(sv.envmap->to_local, s.d), u, v);
            env_grad_add(*sv.envmap, u, v, mk3(ew * s.thr.x, ew * s.thr.y, ew * s.thr.z), *eg);
        }
    }
    bool active = found;

    // Russian roulette (path.cpp:137-141)
    if ((int32_t) s.depth > P.rr_depth) {
        float hm = fmaxf(fmaxf(s.thr.x, s.thr.y), s.thr.z);
        float q = fminf(hm * (s.eta * s.eta), 0.95f);
        if (active) active = pcg_next_f32(s.rng) < q;
        float rq = rcp(q);
        if (REC) {
            rec->invq = rq;
            if (hm * (s.eta * s.eta) < 0.95f) rec->rr_channel = s.thr.x == hm ? 0 : (s.thr.y == hm ? 1 : 2);
        }
        s.thr = s.thr * rq;
        if (PGRAD) pg->dthr = pg->dthr * rq;
    }
    if (s.depth >= (uint32_t) P.max_depth || !active) return false;

    DevBsdf bsdf = geo.bsdf((uint32_t) si.shape_rec.bsdf);      // blend / mask: overwritten with the child in use (surface_bsdf_*)
    uint32_t texel; f2 tw1;
    const f3 refl = eval_reflectance(sv, bsdf, si.uv, texel, tw1);
    const NestInfo ni = nest_info<NEST>(bsdf, refl.x, refl.y, refl.z);
    auto child_refl = [&](const DevBsdf &rec) { uint32_t t; f2 w; return eval_reflectance(sv, rec, si.uv, t, w); };      // blend / mask children
    const bool smooth = !GENERAL || bsdf_is_smooth(bsdf);
    // PGRAD: is this the record whose parameter is differentiated?  d(value)/d(theta) of the model at fixed directions
    const bool pg_here = PGRAD && si.shape_rec.bsdf == pg->bsdf;
    auto pg_dvalue = [&](f3 wo_l) -> f3 {
        uint32_t tt; f2 tw;
        const f3 rp = eval_reflectance(sv, pg->bp, si.uv, tt, tw), rm = eval_reflectance(sv, pg->bm, si.uv, tt, tw);
        f3 vp, vm; float pp, pm;
        bsdf_eval_pdf(pg->bp, rp, si.wi, wo_l, vp, pp);      // the model code itself (two-sided adapter included), no nesting
        bsdf_eval_pdf(pg->bm, rm, si.wi, wo_l, vm, pm);
        return mk3((vp.x - vm.x) * pg->inv_2h, (vp.y - vm.y) * pg->inv_2h, (vp.z - vm.z) * pg->inv_2h);
    };
    if (REC) { rec->Tp = s.thr; rec->rho = refl; rec->texel = texel; rec->w1 = tw1; rec->bsdf = si.shape_rec.bsdf; rec->has_bsdf = 1u; }
    // adjoint replay of a `twosided` diffuse BSDF (twosided.cpp:94-175; the primal render of such a scene runs the GENERAL kernels, whose
    // diffuse branch does the same arithmetic): the back side scatters like the front side, mirrored
    f3 wi_b = si.wi;
    const bool flip = REC && (bsdf.flags & kBsdfTwoSided) != 0u && wi_b.z < 0.0f;
    if (flip) wi_b.z = -wi_b.z;

    // --------------------- Emitter sampling (path.cpp:153-172) ---------------------
    if (smooth) {                                            // active_e: only BSDFs with a smooth component (path.cpp:154)
        f2 s2; s2.x = pcg_next_f32(s.rng); s2.y = pcg_next_f32(s.rng);
        DirectionSample ds; f3 spec;
        float em_geo = 0.0f;                                 // REC: spec / radiance of an area light
        if (REC) {
            float r1, r2;
            sample_emitter_direction<FLAT, GENERAL>(geo, si.p, s2, ds, r1, r2);
            spec = mk3(0.0f, 0.0f, 0.0f);
            if (sv.n_emitters != 0u) {
                const DevEmitter e = geo.emitter(ds.emitter);
                spec = mk3(e.r * r1, e.g * r1, e.b * r1);
                if (sv.n_emitters > 1u) spec = spec * r2;
                em_geo = r1 * r2;
            }
        } else if (ENVGRAD) {          // the wrapper below, keeping spec / radiance
            float r1, r2;
            sample_emitter_direction<FLAT, GENERAL>(geo, si.p, s2, ds, r1, r2);
            spec = mk3(0.0f, 0.0f, 0.0f);
            if (sv.n_emitters != 0u) {
                const DevEmitter e = geo.emitter(ds.emitter);
                f3 rad = mk3(e.r, e.g, e.b);
                if (e.pad0 == kEmitterEnvmap) { rad = envmap_lookup(*sv.envmap, ds.uv.x, ds.uv.y); em_geo = sv.n_emitters > 1u ? r1 * r2 : r1; }
                if (ds.delta) rad = mk3(rad.x * ds.falloff, rad.y * ds.falloff, rad.z * ds.falloff);
                spec = mk3(rad.x * r1, rad.y * r1, rad.z * r1);
                if (sv.n_emitters > 1u) spec = spec * r2;
            }
        } else sample_emitter_direction<FLAT, GENERAL>(geo, si.p, s2, ds, spec);
        if (ds.pdf != 0.0f) {
            f3 wo = to_local(si.sh, ds.d);
            f3 bv; float bp;
            if (GENERAL) surface_bsdf_eval_pdf<NEST>(bsdf, ni, refl, [&](uint32_t i) { return geo.bsdf(i); }, child_refl, si.wi, wo, bv, bp);
            else diffuse_eval_pdf(refl, wi_b, flip ? mk3(wo.x, wo.y, -wo.z) : wo, bv, bp);
            float mis = (GENERAL && ds.delta) ? 1.0f : mis_weight(ds.pdf, bp);      // path.cpp:170
            f3 contrib = mk3(((mis * s.thr.x) * bv.x) * spec.x, ((mis * s.thr.y) * bv.y) * spec.y,
                             ((mis * s.thr.z) * bv.z) * spec.z);
            // The visibility test only ever zeroes `spec` (scene.cpp:178-182): trace the shadow
            // ray only if an unoccluded sample would contribute.
            if (DEFER) {
                if (contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f) {
                    ++c.any;
                    df->pending = true; df->so = si.p; df->sd = ds.d;
                    df->smint = kRayEpsilon * (1.0f + hmax_abs(si.p)); df->smaxt = ds.dist * (1.0f - kShadowEpsilon);
                    df->nee[0] = contrib.x; df->nee[1] = contrib.y; df->nee[2] = contrib.z; df->nee[3] = 0.0f;
                }
            } else if (contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f || ((REC || ENVGRAD) && em_geo != 0.0f) || pg_here) {
                Hit sh;
                ++c.any;
#if defined(MTS_ABLATE_SHADOW)   // diagnostic build only: wrong image, used to price the any-hit loop in situ
                bool occluded = false;
#else
                bool occluded = traverse<FLAT, true>(sv, lds, si.p, ds.d, kRayEpsilon * (1.0f + hmax_abs(si.p)),
                                               ds.dist * (1.0f - kShadowEpsilon), sh, c.tri_tests);
#endif
                if (!occluded) {
                    s.res = s.res + contrib;
                    if (PGRAD) {          // d(mis thr bv spec) with mis and spec fixed
                        f3 dbv = mk3(0.0f, 0.0f, 0.0f);
                        if (pg_here) dbv = pg_dvalue(wo);
                        pg->dres = mk3(pg->dres.x + (mis * fmaf(pg->dthr.x, bv.x, s.thr.x * dbv.x)) * spec.x,
                                       pg->dres.y + (mis * fmaf(pg->dthr.y, bv.y, s.thr.y * dbv.y)) * spec.y,
                                       pg->dres.z + (mis * fmaf(pg->dthr.z, bv.z, s.thr.z * dbv.z)) * spec.z);
                    }
                    if (ENVGRAD && em_geo != 0.0f)
                        env_grad_add(*sv.envmap, ds.uv.x, ds.uv.y, mk3(((mis * s.thr.x) * bv.x) * em_geo, ((mis * s.thr.y) * bv.y) * em_geo,
                                                                         ((mis * s.thr.z) * bv.z) * em_geo), *eg);
                    const float wo_bz = flip ? -wo.z : wo.z;      // the BSDF's side of the surface (twosided)
                    if (REC && wi_b.z > 0.0f && wo_bz > 0.0f) {     // d(contrib)/d(rho) / T'_k
                        float k = mis * (kInvPi * wo_bz);
                        rec->Nc = mk3(k * spec.x, k * spec.y, k * spec.z);
                        if (em_geo != 0.0f) { rec->nk = k * em_geo; rec->em_nee = (int32_t) ds.emitter; }
                    }
                }
            }
        }
    }

    // ----------------------- BSDF sampling (path.cpp:177-190) ----------------------
    const float s1 = pcg_next_f32(s.rng);                    // sample1 (lobe selection; unused by SmoothDiffuse)
    f2 s2; s2.x = pcg_next_f32(s.rng); s2.y = pcg_next_f32(s.rng);
    f3 wo, weight; float pdf;
    if (GENERAL) {
        BsdfSample bs;
        surface_bsdf_sample<NEST>(bsdf, ni, refl, [&](uint32_t i) { return geo.bsdf(i); }, child_refl, si.wi, s1, s2, bs, weight);
        wo = bs.wo; pdf = bs.pdf;
        s.eta *= bs.eta;                                     // harmless for a failed sample: the path ends below
        s.flags = bs.delta ? (s.flags | kFlagDelta) : (s.flags & ~kFlagDelta);
        if (PGRAD) {          // d(thr weight) = dthr weight + thr dweight;  dweight = d(value)/d(theta) / pdf at the sampled direction
            f3 dw = mk3(0.0f, 0.0f, 0.0f);
            if (pg_here && !bs.delta && bs.pdf > 0.0f) {
                const f3 dv = pg_dvalue(bs.wo);
                const float ip = rcp(bs.pdf);
                dw = mk3(dv.x * ip, dv.y * ip, dv.z * ip);
            } else if (pg_here && bs.delta) {
                // a discrete lobe: its weight is a closed form of the parameters (Fresnel term x specular colour / lobe probability);
                // the same lobe is re-evaluated with the perturbed records and the same random numbers, and counts only if both land on
                // the very direction of the primal sample (a refracted direction moves with the index of refraction: detached -> no term)
                uint32_t tt; f2 tw;
                const f3 rp = eval_reflectance(sv, pg->bp, si.uv, tt, tw), rm = eval_reflectance(sv, pg->bm, si.uv, tt, tw);
                BsdfSample bp_, bm_; f3 wp, wm;
                const bool okp = bsdf_sample(pg->bp, rp, si.wi, s1, s2, bp_, wp);
                const bool okm = bsdf_sample(pg->bm, rm, si.wi, s1, s2, bm_, wm);
                if (okp && okm && bp_.delta && bm_.delta && bp_.wo.x == bs.wo.x && bp_.wo.y == bs.wo.y && bp_.wo.z == bs.wo.z &&
                    bm_.wo.x == bs.wo.x && bm_.wo.y == bs.wo.y && bm_.wo.z == bs.wo.z)
                    dw = mk3((wp.x - wm.x) * pg->inv_2h, (wp.y - wm.y) * pg->inv_2h, (wp.z - wm.z) * pg->inv_2h);
            }
            pg->dthr = mk3(fmaf(pg->dthr.x, weight.x, s.thr.x * dw.x), fmaf(pg->dthr.y, weight.y, s.thr.y * dw.y), fmaf(pg->dthr.z, weight.z, s.thr.z * dw.z));
        }
    } else {
        diffuse_sample(refl, wi_b, s2, wo, pdf, weight);     // eta *= bs.eta (== 1)
        if (flip) wo.z = -wo.z;
    }
    s.thr = mk3(s.thr.x * weight.x, s.thr.y * weight.y, s.thr.z * weight.z);
    if (!(s.thr.x != 0.0f || s.thr.y != 0.0f || s.thr.z != 0.0f)) return false;
    s.o = si.p;                                              // spawn_ray (interaction.h:58-61)
    s.d = to_world(si.sh, wo);
    s.mint = (1.0f + hmax_abs(si.p)) * kRayEpsilon;
    s.maxt = __builtin_inff();
    s.bs_pdf = pdf;
    s.depth += 1u;
    return true;
}

// render_sample up to the camera ray (integrator.cpp:224-246).  `lp` = local pixel index (row-major over the rows
// this render owns), `j` = sample number inside the pixel; the RNG stream is seeded with the GLOBAL sample index
// pixel * spp + j, so the image does not depend on how the film is partitioned.
template <bool GENERAL = true>
MTS_DEV void generate_path(const RenderParams &P, uint64_t ordinal, uint32_t lp, uint32_t j, PathState &s, float2 *pos_out = nullptr) {
    const uint32_t w = (uint32_t) P.crop_w;
    const uint32_t lr = lp / w, px = lp - lr * w;
    const uint32_t py = (uint32_t) row_to_global(P.rows, (int32_t) lr);
    const uint64_t index = ((uint64_t) py * w + px) * (uint64_t) P.spp + j;
    seed_sample(s.rng, index, P.base_seed);
    float jx = pcg_next_f32(s.rng), jy = pcg_next_f32(s.rng);
    float psx = ((float) px + (float) P.crop_x) + jx, psy = ((float) py + (float) P.crop_y) + jy;
    f2 ap; ap.x = ap.y = 0.5f;                               // needs_aperture_sample(): integrator.cpp:229-231
    if (GENERAL && P.cam.aperture_radius > 0.0f) { ap.x = pcg_next_f32(s.rng); ap.y = pcg_next_f32(s.rng); }
    (void) pcg_next_f32(s.rng);                              // wavelength sample (drawn even in RGB mode)
    float ax = (psx - (float) P.crop_x) / (float) P.crop_w, ay = (psy - (float) P.crop_y) / (float) P.crop_h;
    camera_ray<GENERAL>(P.cam, ax, ay, ap, s.o, s.d, s.mint, s.maxt);
    s.thr = mk3(1.0f, 1.0f, 1.0f); s.bs_pdf = 0.0f;
    s.res = mk3(0.0f, 0.0f, 0.0f); s.eta = 1.0f;
    s.ordinal = P.plane_pixels ? j * P.plane_pixels + (lp - P.plane_pix0) : (uint32_t) (ordinal - P.first_ordinal);
    s.depth = 1u; s.flags = 0u;
    if (P.out_pos) P.out_pos[s.ordinal] = make_float2(psx, psy);
    if (pos_out) *pos_out = make_float2(psx, psy);
}

// final radiance of a terminated path -> per-sample stream
MTS_DEV void store_result(const RenderParams &P, const PathState &s) {
    float alpha = (s.flags & 1u) ? 1.0f : 0.0f;
    if (P.store_xyz) {
        f3 xyz = P.store_xyz == 1 ? srgb_to_xyz(s.res) : s.res;   // integrator.cpp:254-262; 2: linear RGB film (autodiff.py:53-57)
        // ImageBlock::put drops invalid samples (imageblock.cpp:85-109); the autodiff film is created with
        // warn_negative = False (autodiff.py:60-67), so there only non-finite values are dropped
        const float lo = P.store_xyz == 1 ? -1e-5f : -__builtin_inff();
        bool valid = (xyz.x >= lo) && (xyz.y >= lo) && (xyz.z >= lo) && isfinite(xyz.x) && isfinite(xyz.y) && isfinite(xyz.z);
        P.out_rgba[s.ordinal] = make_float4(xyz.x, xyz.y, xyz.z, valid ? alpha : -1.0f);
    } else {
        P.out_rgba[s.ordinal] = make_float4(s.res.x, s.res.y, s.res.z, alpha);
    }
}

// Sample ordinals are dealt to the scheduling waves in chunks of P.chunk, round-robin: wave k traces the chunks k, k + n_waves, ... of
// the pass.  LDS-resident scenes use chunks of 64, so that every wave sees the same mix of cheap and expensive pixels and the pool
// stays full until all cursors run dry together (contiguous ranges per wave left waves over easy pixels idle from a third of the pass
// on: 94 -> 82 launches per render); hierarchy scenes keep one chunk per wave -- consecutive pixels -- because the rays of a
// workgroup's scheduling waves should walk the same part of the BVH (64-sample chunks cost them 5 %).  The cursor of a wave counts
// its own samples; this maps the v-th of them to its place in the pass, (local pixel, sample number) included.  Which wave traces a
// sample has no influence on the result: its RNG stream and its slot in the sample stream depend on the ordinal alone.
MTS_DEV void cursor_sample(const RenderParams &P, uint32_t wave, uint64_t v, uint64_t &ordinal, uint32_t &lp, uint32_t &j) {
    const uint32_t t = (uint32_t) v / P.chunk, within = (uint32_t) v - t * P.chunk;
    const uint32_t in_pass = (t * P.n_waves + chunk_owner(wave, P.n_waves, P.n_chains)) * P.chunk + within;      // < 2^31
    ordinal = P.first_ordinal + in_pass;
    const uint32_t r = in_pass + P.first_rem, q = r / (uint32_t) P.spp;
    lp = P.first_pix + q; j = r - q * (uint32_t) P.spp;
}

#ifndef MTS_BOUNCE_WAVES
#define MTS_BOUNCE_WAVES 4
#endif
template <bool FLAT, bool GENERAL, bool NEST = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(MTS_BOUNCE_WAVES, MTS_BOUNCE_WAVES)))
void k_bounce(const RenderParams P) {
    extern __shared__ float4 smem[];
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    {   // a workgroup whose scheduling waves are all idle (pool drain at the end of a pass) leaves before staging the scene
        const bool work = wave < P.n_waves && (P.count_in[wave] > 0u || P.cursor[wave] < P.cursor_end[wave]);
        if (!__syncthreads_or(work ? 1 : 0)) {
            if (wave < P.n_waves && lane_id() == 0u) P.count_out[wave] = 0u;
            return;
        }
    }
    const LdsView lds = lds_stage<FLAT>(P.sv, smem);
    if (wave >= P.n_waves) return;
    const uint32_t lane = lane_id();
    const uint32_t n_in = __builtin_amdgcn_readfirstlane(P.count_in[wave]);
    const size_t base = (size_t) wave * P.seg_cap;
    uint32_t n_out = 0;
    Counters c = { 0u, 0u, 0u, 0u };

    for (uint32_t i0 = 0; i0 < n_in; i0 += 64u) {
        PathState s;
        bool alive = false;
        if (i0 + lane < n_in) {
            load_state(P.in, base + i0 + lane, s);
            alive = bounce_step<FLAT, false, 0, GENERAL, false, NEST>(P, lds, s, c);
            if (!alive) store_result(P, s);
        }
        // wavefront ballot + prefix rank: compact the survivors to the front of the output segment
        const uint64_t m = __ballot(alive);
        if (alive) store_state(P.out, base + n_out + mask_rank(m), s);
        n_out += (uint32_t) __popcll(m);
    }

    // regenerate camera paths into the free slots of this wave's segment.  The cursor is kept decomposed as
    // (local pixel, sample-in-pixel) so that no 64-bit division is needed per generated path.
    uint64_t cursor = P.cursor[wave];
    const uint64_t end = P.cursor_end[wave];
    while (n_out < P.target && cursor < end) {
        uint64_t left = end - cursor;
        uint32_t n_new = min(64u, P.target - n_out);
        if ((uint64_t) n_new > left) n_new = (uint32_t) left;
        if (lane < n_new) {
            PathState s;
            uint64_t ordinal; uint32_t lp, sj;
            cursor_sample(P, wave, cursor + lane, ordinal, lp, sj);
            generate_path<GENERAL>(P, ordinal, lp, sj, s);
            store_state(P.out, base + n_out + lane, s);
        }
        n_out += n_new; cursor += n_new;
    }

    // per-wave bookkeeping (each wave owns its slots: no atomics)
    uint32_t tot[4] = { c.closest, c.any, c.segments, c.tri_tests };
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int off = 32; off > 0; off >>= 1) tot[k] += __shfl_xor(tot[k], off);
    if (lane == 0) {
        P.count_out[wave] = n_out;
        P.cursor[wave] = cursor;
        uint64_t *ws = P.wave_stats + 4u * (size_t) wave;
        ws[0] += tot[0]; ws[1] += tot[1]; ws[2] += tot[2]; ws[3] += tot[3];
    }
}

size_t bounce_lds_bytes(const SceneView &sv) { return lds_bytes(sv, kBlock); }


// ---------------------------------------------------------------------------------------------
// Spectral variant: the same path logic on 4 wavelengths per camera sample (scalar_spectral / gpu_spectral semantics).
// Geometry, sampling decisions and the RNG consumption are wavelength independent; reflectances and emission are
// smooth spectra evaluated per wavelength, and the result is converted to XYZ with the CIE 1931 observer.
__device__ SpectralTables g_spectral;

hipError_t upload_spectral_tables(const float *x, const float *y, const float *z, const float *d65) {
    SpectralTables t;
    for (int i = 0; i < 95; ++i) { t.x[i] = x[i]; t.y[i] = y[i]; t.z[i] = z[i]; t.d65[i] = d65[i]; }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_spectral), &t, sizeof(t));
}

struct PathStateS {
    f3 o, d; float mint, maxt;
    Spec4 thr, res, wav; float bs_pdf, eta, xi;      // wav = wavelengths_from_sample(xi): not stored
    Pcg32 rng;
    uint32_t ordinal, depth, flags;
};

template <bool NT = false>
MTS_DEV void load_state(const PoolView &p, size_t i, PathStateS &s) {
    float4 a = ld_stream<NT>(p.ray_o + i), b = ld_stream<NT>(p.ray_d + i), c = ld_stream<NT>(p.thr + i), e = ld_stream<NT>(p.res + i);
    s.xi = ld_stream<NT>(p.xi + i);
    wavelengths_from_sample(s.xi, s.wav);
    float2 x = ld_stream<NT>(p.aux + i); uint4 r = ld_stream<NT>(p.rng + i); uint2 m = ld_stream<NT>(p.misc + i);
    s.o = mk3(a.x, a.y, a.z); s.mint = a.w;
    s.d = mk3(b.x, b.y, b.z); s.maxt = b.w;
    s.thr.v[0] = c.x; s.thr.v[1] = c.y; s.thr.v[2] = c.z; s.thr.v[3] = c.w;
    s.res.v[0] = e.x; s.res.v[1] = e.y; s.res.v[2] = e.z; s.res.v[3] = e.w;
    s.bs_pdf = x.x; s.eta = x.y;
    s.rng.state = (uint64_t) r.x | ((uint64_t) r.y << 32);
    s.rng.inc = (uint64_t) r.z | ((uint64_t) r.w << 32);
    s.ordinal = m.x; s.depth = m.y & 0xffffu; s.flags = m.y >> 16;
}
template <bool NT = false>
MTS_DEV void store_state(const PoolView &p, size_t i, const PathStateS &s) {
    st_stream<NT>(p.ray_o + i, make_float4(s.o.x, s.o.y, s.o.z, s.mint));
    st_stream<NT>(p.ray_d + i, make_float4(s.d.x, s.d.y, s.d.z, s.maxt));
    st_stream<NT>(p.thr + i, make_float4(s.thr.v[0], s.thr.v[1], s.thr.v[2], s.thr.v[3]));
    p.res[i] = make_float4(s.res.v[0], s.res.v[1], s.res.v[2], s.res.v[3]);      // read-modify-written by the shadow-ray stage
    st_stream<NT>(p.xi + i, s.xi);
    st_stream<NT>(p.aux + i, make_float2(s.bs_pdf, s.eta));
    st_stream<NT>(p.rng + i, make_uint4((uint32_t) s.rng.state, (uint32_t) (s.rng.state >> 32), (uint32_t) s.rng.inc,
                                        (uint32_t) (s.rng.inc >> 32)));
    st_stream<NT>(p.misc + i, make_uint2(s.ordinal, (s.depth & 0xffffu) | (s.flags << 16)));
}

// spectral variant: `srgb` parameters evaluate the upsampled colour at each wavelength (srgb.cpp:45-52), `uniform` ones are
// constants; conductors carry uniform eta / k
MTS_DEV BsdfChannels<kWav> spectral_channels(const DevBsdf &b, const Spec4 &wav) {
    BsdfChannels<kWav> c;
#pragma unroll
    for (int k = 0; k < kWav; ++k) {
        c.refl[k] = (b.flags & kBsdfUniformRefl) ? b.r : srgb_model_eval(b.c0, b.c1, b.c2, wav.v[k]);
        c.spec[k] = (b.flags & kBsdfUniformSpec) ? b.sr : srgb_model_eval(b.sc0, b.sc1, b.sc2, wav.v[k]);
        c.trans[k] = (b.flags & kBsdfUniformTrans) ? b.kr : srgb_model_eval(b.tc0, b.tc1, b.tc2, wav.v[k]);
        c.eta[k] = b.er; c.k[k] = b.kr;
    }
    return c;
}

// radiance spectrum of an emitter at the path's 4 wavelengths: SRGBEmitterSpectrum::eval = d65 * srgb_model_eval
// (srgb_d65.cpp:54-62) for `area` / `constant`; eval_spectrum's spectral branch (envmap.cpp:283-306) at texture
// coordinates (u, v) for `envmap`, whose texels hold (model coefficients, scale) and whose whitepoint is D65 / 10568
MTS_DEV Spec4 envmap_lookup_spectral(const DevEnvmap &e, float u, float v, const Spec4 &wav) {
    u *= (float) (e.w - 1); v *= (float) (e.h - 1);
    const uint32_t px = min((uint32_t) u, (uint32_t) (e.w - 2)), py = min((uint32_t) v, (uint32_t) (e.h - 2));
    const float w1x = u - (float) px, w1y = v - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const float4 *p = e.data + (size_t) py * e.w + px;
    const float4 v00 = p[0], v10 = p[1], v01 = p[e.w], v11 = p[e.w + 1];
    const float f0 = fmaf(w0x, v00.w, w1x * v10.w), f1 = fmaf(w0x, v01.w, w1x * v11.w);
    const float f = fmaf(w0y, f0, w1y * f1);
    Spec4 r;
#pragma unroll
    for (int k = 0; k < kWav; ++k) {
        const float l = wav.v[k];
        const float s00 = srgb_model_eval(v00.x, v00.y, v00.z, l), s10 = srgb_model_eval(v10.x, v10.y, v10.z, l);
        const float s01 = srgb_model_eval(v01.x, v01.y, v01.z, l), s11 = srgb_model_eval(v11.x, v11.y, v11.z, l);
        const float s0 = fmaf(w0x, s00, w1x * s10), s1 = fmaf(w0x, s01, w1x * s11);
        const float sp = fmaf(w0y, s0, w1y * s1);
        const float wp = table_eval(g_spectral.d65, 1.0f / 10568.0f, l);
        r.v[k] = ((sp * wp) * f) * e.scale;
    }
    return r;
}
// textured reflectance in the spectral variant: bitmap texels hold srgb model coefficients, evaluated at the four corners and
// then interpolated (bitmap.cpp:274-286); checkerboard colours are `srgb` spectra (checkerboard.cpp:46-63)
MTS_DEV Spec4 eval_reflectance_spectral(const SceneView &sv, const DevBsdf &b, f2 uv, const Spec4 &wav) {
    const DevTexture t = sv.textures[b.texture];
    {
        const float u2 = fmaf(t.uvm[0], uv.x, fmaf(t.uvm[1], uv.y, t.uvm[2])), v2 = fmaf(t.uvm[3], uv.x, fmaf(t.uvm[4], uv.y, t.uvm[5]));
        uv.x = u2; uv.y = v2;
    }
    Spec4 r;
    if (t.kind == 1u) {
        const bool mx = (uv.x - floorf(uv.x)) > 0.5f, my = (uv.y - floorf(uv.y)) > 0.5f;
        const float *c = mx == my ? t.c0 : t.c1;
        const float a0 = c[0], a1 = c[1], a2 = c[2];
#pragma unroll
        for (int k = 0; k < kWav; ++k) r.v[k] = srgb_model_eval(a0, a1, a2, wav.v[k]);
        return r;
    }
    float ux = uv.x - floorf(uv.x), uy = uv.y - floorf(uv.y);
    ux *= (float) (uint32_t) (t.w - 1); uy *= (float) (uint32_t) (t.h - 1);
    const uint32_t px = min((uint32_t) ux, (uint32_t) (t.w - 2)), py = min((uint32_t) uy, (uint32_t) (t.h - 2));
    const float w1x = ux - (float) px, w1y = uy - (float) py, w0x = 1.0f - w1x, w0y = 1.0f - w1y;
    const float *v00 = t.data + 3u * (size_t) (px + py * (uint32_t) t.w), *v01 = v00 + 3u * (size_t) t.w;
#pragma unroll
    for (int k = 0; k < kWav; ++k) {
        const float l = wav.v[k];
        const float c00 = srgb_model_eval(v00[0], v00[1], v00[2], l), c10 = srgb_model_eval(v00[3], v00[4], v00[5], l);
        const float c01 = srgb_model_eval(v01[0], v01[1], v01[2], l), c11 = srgb_model_eval(v01[3], v01[4], v01[5], l);
        const float c0 = fmaf(w0x, c00, w1x * c10), c1 = fmaf(w0x, c01, w1x * c11);
        r.v[k] = fmaf(w0y, c0, w1y * c1);
    }
    return r;
}
MTS_DEV Spec4 emitter_spectrum(const SceneView &sv, const DevEmitter &e, const Spec4 &wav, f2 uv) {
    if (e.pad0 == kEmitterEnvmap) return envmap_lookup_spectral(*sv.envmap, uv.x, uv.y, wav);
    Spec4 r;
#pragma unroll
    for (int k = 0; k < kWav; ++k) r.v[k] = table_eval(g_spectral.d65, e.d65_scale, wav.v[k]) * srgb_model_eval(e.c0, e.c1, e.c2, wav.v[k]);
    return r;
}

template <bool FLAT, int DEFER = 0, bool GENERAL = false, bool NEST = false>
MTS_DEV bool bounce_step_spectral(const RenderParams &P, const LdsView &lds, PathStateS &s, Counters &c, Deferred *df = nullptr) {
    const SceneView &sv = P.sv;
    const Geo<FLAT> geo{ sv, lds };
    Hit hit;
    ++c.closest; ++c.segments;
    bool found;
    if (DEFER) df->pending = false;
    if (DEFER == 1) { hit = df->hit; found = df->found; }
    else found = traverse<FLAT, false>(sv, lds, s.o, s.d, s.mint, s.maxt, hit, c.tri_tests,
                                       FLAT && __ballot(s.depth != 1u) == 0ull);      // camera rays only: cluster culling (as bounce_step)
    if (s.depth == 1u) s.flags = found ? 1u : 0u;

    SurfaceInteraction si;
    if (found) {
        fill_si(geo, s.d, hit.prim, hit.u, hit.v, si);
        int32_t emitter = si.shape_rec.emitter;
        if (emitter >= 0) {
            const DevEmitter e = geo.emitter((uint32_t) emitter);
            float ew = 1.0f;
            if (s.depth > 1u) {
                f3 dd = si.p - s.o;
                float dist = sqrtf(sqnorm(dd));
                dd = div_s(dd, dist);
                const float pe = pdf_emitter_direction(sv.n_emitters, e.area_norm, dd, si.sh.n, dist);
                ew = mis_weight(s.bs_pdf, (GENERAL && (s.flags & kFlagDelta)) ? 0.0f : pe);
            }
            if (si.wi.z > 0.0f) {
#pragma unroll
                for (int k = 0; k < kWav; ++k) {       // SRGBEmitterSpectrum::eval = d65 * srgb_model_eval (srgb_d65.cpp:54-62)
                    float le = table_eval(g_spectral.d65, e.d65_scale, s.wav.v[k]) * srgb_model_eval(e.c0, e.c1, e.c2, s.wav.v[k]);
                    s.res.v[k] += (ew * s.thr.v[k]) * le;
                }
            }
        }
    }
    if (GENERAL && !found && sv.env_emitter >= 0) {          // si.emitter(scene) of an escaped ray: the environment
        const DevEmitter e = geo.emitter((uint32_t) sv.env_emitter);
        float ew = 1.0f;
        if (s.depth > 1u) ew = mis_weight(s.bs_pdf, (GENERAL && (s.flags & kFlagDelta)) ? 0.0f : pdf_environment(sv, e, s.d));
        f2 uv; uv.x = uv.y = 0.0f;
        if (e.pad0 == kEmitterEnvmap) env_dir_to_uv(mat3_apply(sv.envmap->to_local, s.d), uv.x, uv.y);      // envmap.cpp:135-144
        const Spec4 le = emitter_spectrum(sv, e, s.wav, uv);
#pragma unroll
        for (int k = 0; k < kWav; ++k) s.res.v[k] += (ew * s.thr.v[k]) * le.v[k];
    }
    bool active = found;

    if ((int32_t) s.depth > P.rr_depth) {
        float hm = fmaxf(fmaxf(s.thr.v[0], s.thr.v[1]), fmaxf(s.thr.v[2], s.thr.v[3]));
        float q = fminf(hm * (s.eta * s.eta), 0.95f);
        if (active) active = pcg_next_f32(s.rng) < q;
        float rq = rcp(q);
#pragma unroll
        for (int k = 0; k < kWav; ++k) s.thr.v[k] *= rq;
    }
    if (s.depth >= (uint32_t) P.max_depth || !active) return false;

    DevBsdf bsdf = geo.bsdf((uint32_t) si.shape_rec.bsdf);      // blend / mask: overwritten with the child in use (surface_bsdf_*)
    Spec4 refl;
#pragma unroll
    for (int k = 0; k < kWav; ++k)       // srgb.cpp:45-52 / uniform.cpp
        refl.v[k] = (bsdf.flags & kBsdfUniformRefl) ? bsdf.r : srgb_model_eval(bsdf.c0, bsdf.c1, bsdf.c2, s.wav.v[k]);
    if (bsdf.texture >= 0) refl = eval_reflectance_spectral(sv, bsdf, si.uv, s.wav);
    BsdfChannels<kWav> chan;
    if (GENERAL) {
        chan = spectral_channels(bsdf, s.wav);
#pragma unroll
        for (int k = 0; k < kWav; ++k) chan.refl[k] = refl.v[k];
    }
    // blend / mask: per-wavelength inputs of a child record (its own constant parameters)
    // a scalar weight in the spectral variant is a constant (a textured one is refused at scene creation)
    const NestInfo ni = nest_info<NEST>(bsdf, refl.v[0], refl.v[1], refl.v[2]);
    const bool smooth = !GENERAL || bsdf_is_smooth(bsdf);
    auto chan_of = [&](const DevBsdf &rec, bool child) { return child ? spectral_channels(rec, s.wav) : chan; };

    if (smooth) {
        f2 s2; s2.x = pcg_next_f32(s.rng); s2.y = pcg_next_f32(s.rng);
        DirectionSample ds; float r1, r2;
        sample_emitter_direction<FLAT, GENERAL>(geo, si.p, s2, ds, r1, r2);
        if (ds.pdf != 0.0f) {
            const DevEmitter e = geo.emitter(ds.emitter);
            f3 wo = to_local(si.sh, ds.d);
            bool front = si.wi.z > 0.0f && wo.z > 0.0f;
            float bp = front ? kInvPi * wo.z : 0.0f;
            float bvs[kWav];
            if (GENERAL) surface_bsdf_eval_pdf<NEST, kWav>(bsdf, ni, [&](uint32_t i) { return geo.bsdf(i); }, chan_of, si.wi, wo, bvs, bp);
            float mis = (GENERAL && ds.delta) ? 1.0f : mis_weight(ds.pdf, bp);
            Spec4 contrib; bool nz = false;
            const Spec4 le4 = emitter_spectrum(sv, e, s.wav, ds.uv);
#pragma unroll
            for (int k = 0; k < kWav; ++k) {
                float le = (GENERAL && ds.delta) ? le4.v[k] * ds.falloff : le4.v[k];
                float spec = le * r1;
                if (sv.n_emitters > 1) spec *= r2;
                float bv = GENERAL ? bvs[k] : (front ? (refl.v[k] * kInvPi) * wo.z : 0.0f);
                contrib.v[k] = ((mis * s.thr.v[k]) * bv) * spec;
                nz = nz || contrib.v[k] != 0.0f;
            }
            if (DEFER) {
                if (nz) {
                    ++c.any;
                    df->pending = true; df->so = si.p; df->sd = ds.d;
                    df->smint = kRayEpsilon * (1.0f + hmax_abs(si.p)); df->smaxt = ds.dist * (1.0f - kShadowEpsilon);
#pragma unroll
                    for (int k = 0; k < kWav; ++k) df->nee[k] = contrib.v[k];
                }
            } else if (nz) {
                Hit sh;
                ++c.any;
                bool occluded = traverse<FLAT, true>(sv, lds, si.p, ds.d, kRayEpsilon * (1.0f + hmax_abs(si.p)),
                                               ds.dist * (1.0f - kShadowEpsilon), sh, c.tri_tests);
                if (!occluded) {
#pragma unroll
                    for (int k = 0; k < kWav; ++k) s.res.v[k] += contrib.v[k];
                }
            }
        }
    }

    const float s1 = pcg_next_f32(s.rng);
    f2 s2; s2.x = pcg_next_f32(s.rng); s2.y = pcg_next_f32(s.rng);
    f3 wo = mk3(0.0f, 0.0f, 0.0f); float pdf = 0.0f; bool ok = false;
    bool nz = false;
    if (GENERAL) {
        BsdfSample bs; float w[kWav];
        surface_bsdf_sample<NEST, kWav>(bsdf, ni, [&](uint32_t i) { return geo.bsdf(i); }, chan_of, si.wi, s1, s2, bs, w);
        wo = bs.wo; pdf = bs.pdf;
        s.eta *= bs.eta;
        s.flags = bs.delta ? (s.flags | kFlagDelta) : (s.flags & ~kFlagDelta);
#pragma unroll
        for (int k = 0; k < kWav; ++k) { s.thr.v[k] = s.thr.v[k] * w[k]; nz = nz || s.thr.v[k] != 0.0f; }
    } else {
        if (si.wi.z > 0.0f) {
            wo = square_to_cosine_hemisphere(s2);
            pdf = kInvPi * wo.z;
            ok = pdf > 0.0f;
        }
#pragma unroll
        for (int k = 0; k < kWav; ++k) { s.thr.v[k] = s.thr.v[k] * (ok ? refl.v[k] : 0.0f); nz = nz || s.thr.v[k] != 0.0f; }
    }
    if (!nz) return false;
    s.o = si.p;
    s.d = to_world(si.sh, wo);
    s.mint = (1.0f + hmax_abs(si.p)) * kRayEpsilon;
    s.maxt = __builtin_inff();
    s.bs_pdf = pdf;
    s.depth += 1u;
    return true;
}

MTS_DEV void generate_path_spectral(const RenderParams &P, uint64_t ordinal, uint32_t lp, uint32_t j, PathStateS &s) {
    const uint32_t w = (uint32_t) P.crop_w;
    const uint32_t lr = lp / w, px = lp - lr * w;
    const uint32_t py = (uint32_t) row_to_global(P.rows, (int32_t) lr);
    const uint64_t index = ((uint64_t) py * w + px) * (uint64_t) P.spp + j;
    seed_sample(s.rng, index, P.base_seed);
    float jx = pcg_next_f32(s.rng), jy = pcg_next_f32(s.rng);
    float psx = ((float) px + (float) P.crop_x) + jx, psy = ((float) py + (float) P.crop_y) + jy;
    f2 ap; ap.x = ap.y = 0.5f;
    if (P.cam.aperture_radius > 0.0f) { ap.x = pcg_next_f32(s.rng); ap.y = pcg_next_f32(s.rng); }
    s.xi = pcg_next_f32(s.rng);                                // integrator.cpp:237, perspective.cpp:196: sample_wavelength(sample)
    wavelengths_from_sample(s.xi, s.wav);
    float ax = (psx - (float) P.crop_x) / (float) P.crop_w, ay = (psy - (float) P.crop_y) / (float) P.crop_h;
    camera_ray(P.cam, ax, ay, ap, s.o, s.d, s.mint, s.maxt);
#pragma unroll
    for (int k = 0; k < kWav; ++k) { s.thr.v[k] = 1.0f; s.res.v[k] = 0.0f; }
    s.bs_pdf = 0.0f; s.eta = 1.0f;
    s.ordinal = P.plane_pixels ? j * P.plane_pixels + (lp - P.plane_pix0) : (uint32_t) (ordinal - P.first_ordinal);
    s.depth = 1u; s.flags = 0u;
    if (P.out_pos) P.out_pos[s.ordinal] = make_float2(psx, psy);
}

MTS_DEV void store_result_spectral(const RenderParams &P, const PathStateS &s) {
    float alpha = (s.flags & 1u) ? 1.0f : 0.0f;
    Spec4 v;
#pragma unroll
    for (int k = 0; k < kWav; ++k) v.v[k] = wavelength_weight(s.wav.v[k]) * s.res.v[k];    // ray_weight * L (integrator.cpp:250)
    f3 xyz = spectrum_to_xyz(v, s.wav);                                                       // integrator.cpp:259-261
    bool valid = (xyz.x >= -1e-5f) && (xyz.y >= -1e-5f) && (xyz.z >= -1e-5f) && isfinite(xyz.x) && isfinite(xyz.y) && isfinite(xyz.z);
    P.out_rgba[s.ordinal] = make_float4(xyz.x, xyz.y, xyz.z, (valid || !P.store_xyz) ? alpha : -1.0f);
}

template <bool FLAT, bool GENERAL, bool NEST = false>
__global__ __launch_bounds__(kBlock) void k_bounce_spectral(const RenderParams P) {
    extern __shared__ float4 smem[];
    const LdsView lds = lds_stage<FLAT>(P.sv, smem);
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (wave >= P.n_waves) return;
    const uint32_t lane = lane_id();
    const uint32_t n_in = __builtin_amdgcn_readfirstlane(P.count_in[wave]);
    const size_t base = (size_t) wave * P.seg_cap;
    uint32_t n_out = 0;
    Counters c = { 0u, 0u, 0u, 0u };
    for (uint32_t i0 = 0; i0 < n_in; i0 += 64u) {
        PathStateS s;
        bool alive = false;
        if (i0 + lane < n_in) {
            load_state(P.in, base + i0 + lane, s);
            alive = bounce_step_spectral<FLAT, 0, GENERAL, NEST>(P, lds, s, c);
            if (!alive) store_result_spectral(P, s);
        }
        const uint64_t m = __ballot(alive);
        if (alive) store_state(P.out, base + n_out + mask_rank(m), s);
        n_out += (uint32_t) __popcll(m);
    }
    uint64_t cursor = P.cursor[wave];
    const uint64_t end = P.cursor_end[wave];
    while (n_out < P.target && cursor < end) {
        uint64_t left = end - cursor;
        uint32_t n_new = min(64u, P.target - n_out);
        if ((uint64_t) n_new > left) n_new = (uint32_t) left;
        if (lane < n_new) {
            PathStateS s;
            uint64_t ordinal; uint32_t lp, sj;
            cursor_sample(P, wave, cursor + lane, ordinal, lp, sj);
            generate_path_spectral(P, ordinal, lp, sj, s);
            store_state(P.out, base + n_out + lane, s);
        }
        n_out += n_new; cursor += n_new;
    }
    uint32_t tot[4] = { c.closest, c.any, c.segments, c.tri_tests };
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int off = 32; off > 0; off >>= 1) tot[k] += __shfl_xor(tot[k], off);
    if (lane == 0) {
        P.count_out[wave] = n_out;
        P.cursor[wave] = cursor;
        uint64_t *ws = P.wave_stats + 4u * (size_t) wave;
        ws[0] += tot[0]; ws[1] += tot[1]; ws[2] += tot[2]; ws[3] += tot[3];
    }
}

// ---------------------------------------------------------------------------------------------
// `direct` (src/integrators/direct.cpp:105-196) and `depth` (depth.cpp:19-33): no path state survives a sample, so one
// thread carries a camera sample from the sensor to its result.
template <bool FLAT, bool GENERAL, bool NEST = false>
__global__ __launch_bounds__(kBlock) void k_direct(const RenderParams P, uint64_t n) {
    extern __shared__ float4 smem[];
    const LdsView lds = lds_stage<FLAT>(P.sv, smem);
    const uint64_t gid = (uint64_t) blockIdx.x * kBlock + threadIdx.x;
    Counters c = { 0u, 0u, 0u, 0u };
    if (gid < n) {
        const SceneView &sv = P.sv;
        const Geo<FLAT> geo{ sv, lds };
        const uint64_t ordinal = P.first_ordinal + gid;
        const uint32_t lp = (uint32_t) (ordinal / (uint64_t) P.spp), j = (uint32_t) (ordinal - (uint64_t) lp * (uint64_t) P.spp);
        PathState s;
        generate_path(P, ordinal, lp, j, s);
        Hit hit;
        ++c.closest; ++c.segments;
        const bool found = traverse<FLAT, false>(sv, lds, s.o, s.d, s.mint, s.maxt, hit, c.tri_tests, FLAT);      // camera rays of consecutive samples: cluster culling
        s.flags = found ? 1u : 0u;
        if (P.integrator == 2) {
            const float t = found ? hit.t : 0.0f;
            s.res = mk3(t, t, t);
        } else if (!found) {
            if (GENERAL && !P.hide_emitters && sv.env_emitter >= 0) {
                const DevEmitter e = geo.emitter((uint32_t) sv.env_emitter);
                const f3 le = environment_radiance(sv, e, s.d);
                s.res = mk3(s.res.x + le.x, s.res.y + le.y, s.res.z + le.z);
            }
        } else {
            int32_t ne = P.emitter_samples, nb = P.bsdf_samples;
            if (ne == 0 && nb == 0) ne = nb = 1;
            const float sum = (float) (ne + nb);
            const float weight_bsdf = 1.0f / (float) nb, weight_lum = 1.0f / (float) ne;
            const float frac_bsdf = (float) nb / sum, frac_lum = (float) ne / sum;
            SurfaceInteraction si;
            fill_si(geo, s.d, hit.prim, hit.u, hit.v, si);
            if (!P.hide_emitters && si.shape_rec.emitter >= 0 && si.wi.z > 0.0f) {
                const DevEmitter e = geo.emitter((uint32_t) si.shape_rec.emitter);
                s.res = mk3(s.res.x + e.r, s.res.y + e.g, s.res.z + e.b);
            }
            DevBsdf bsdf = geo.bsdf((uint32_t) si.shape_rec.bsdf);      // blend / mask: overwritten with the child in use (surface_bsdf_*)
            uint32_t texel; f2 tw1;
            const f3 refl = eval_reflectance(sv, bsdf, si.uv, texel, tw1);
            const NestInfo ni = nest_info<NEST>(bsdf, refl.x, refl.y, refl.z);
            auto child_refl = [&](const DevBsdf &rec) { uint32_t t; f2 w; return eval_reflectance(sv, rec, si.uv, t, w); };
            if (!GENERAL || bsdf_is_smooth(bsdf)) {
                for (int32_t i = 0; i < ne; ++i) {
                    f2 s2; s2.x = pcg_next_f32(s.rng); s2.y = pcg_next_f32(s.rng);
                    DirectionSample ds; f3 spec;
                    sample_emitter_direction<FLAT, GENERAL>(geo, si.p, s2, ds, spec);
                    if (ds.pdf == 0.0f) continue;
                    const f3 wo = to_local(si.sh, ds.d);
                    f3 bv; float bp;
                    if (GENERAL) surface_bsdf_eval_pdf<NEST>(bsdf, ni, refl, [&](uint32_t i) { return geo.bsdf(i); }, child_refl, si.wi, wo, bv, bp);
                    else diffuse_eval_pdf(refl, si.wi, wo, bv, bp);
                    const float mis = (GENERAL && ds.delta) ? 1.0f : mis_weight(ds.pdf * frac_lum, bp * frac_bsdf) * weight_lum;      // direct.cpp:155-156
                    const f3 contrib = mk3((mis * bv.x) * spec.x, (mis * bv.y) * spec.y, (mis * bv.z) * spec.z);
                    if (contrib.x != 0.0f || contrib.y != 0.0f || contrib.z != 0.0f) {
                        Hit sh;
                        ++c.any;
                        if (!traverse<FLAT, true>(sv, lds, si.p, ds.d, kRayEpsilon * (1.0f + hmax_abs(si.p)),
                                                  ds.dist * (1.0f - kShadowEpsilon), sh, c.tri_tests))
                            s.res = s.res + contrib;
                    }
                }
            }
            for (int32_t i = 0; i < nb; ++i) {
                const float s1 = pcg_next_f32(s.rng);
                f2 s2; s2.x = pcg_next_f32(s.rng); s2.y = pcg_next_f32(s.rng);
                f3 wo, weight; float pdf; bool delta = false;
                if (GENERAL) {
                    BsdfSample bs;
                    surface_bsdf_sample<NEST>(bsdf, ni, refl, [&](uint32_t i) { return geo.bsdf(i); }, child_refl, si.wi, s1, s2, bs, weight);
                    wo = bs.wo; pdf = bs.pdf; delta = bs.delta;
                } else {
                    diffuse_sample(refl, si.wi, s2, wo, pdf, weight);
                }
                if (!(weight.x != 0.0f || weight.y != 0.0f || weight.z != 0.0f)) continue;
                Hit h2;
                ++c.closest; ++c.segments;
                const f3 d2 = to_world(si.sh, wo);
                f3 le; float pe;
                if (!traverse<FLAT, false>(sv, lds, si.p, d2, (1.0f + hmax_abs(si.p)) * kRayEpsilon, __builtin_inff(), h2, c.tri_tests)) {
                    if (!GENERAL || sv.env_emitter < 0) continue;
                    const DevEmitter e = geo.emitter((uint32_t) sv.env_emitter);
                    le = environment_radiance(sv, e, d2);
                    pe = delta ? 0.0f : pdf_environment(sv, e, d2);
                } else {
                    SurfaceInteraction si2;
                    fill_si(geo, d2, h2.prim, h2.u, h2.v, si2);
                    if (si2.shape_rec.emitter < 0) continue;
                    const DevEmitter e = geo.emitter((uint32_t) si2.shape_rec.emitter);
                    le = si2.wi.z > 0.0f ? mk3(e.r, e.g, e.b) : mk3(0.0f, 0.0f, 0.0f);
                    f3 dd = si2.p - si.p;
                    const float dist = sqrtf(sqnorm(dd));
                    dd = div_s(dd, dist);
                    pe = delta ? 0.0f : pdf_emitter_direction(sv.n_emitters, e.area_norm, dd, si2.sh.n, dist);
                }
                const float w = mis_weight(pdf * frac_bsdf, pe * frac_lum) * weight_bsdf;
                s.res = mk3(s.res.x + (weight.x * le.x) * w, s.res.y + (weight.y * le.y) * w, s.res.z + (weight.z * le.z) * w);
            }
        }
        store_result(P, s);
    }
    uint32_t tot[4] = { c.closest, c.any, c.segments, c.tri_tests };
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int off = 32; off > 0; off >>= 1) tot[k] += __shfl_xor(tot[k], off);
    if (lane_id() == 0) {
        unsigned long long *ws = reinterpret_cast<unsigned long long *>(P.wave_stats + 4u * (size_t) ((gid >> 6) % P.n_waves));
        for (int k = 0; k < 4; ++k) if (tot[k]) atomicAdd(ws + k, (unsigned long long) tot[k]);
    }
}

hipError_t launch_direct(const RenderParams &p, uint64_t n, hipStream_t s) {
    const uint32_t blocks = (uint32_t) ((n + kBlock - 1) / kBlock);
    const size_t lds = bounce_lds_bytes(p.sv);
    if (p.sv.general == 2u) {      // scenes with blendbsdf / mask
        if (p.sv.flat) hipLaunchKernelGGL((k_direct<true, true, true>), dim3(blocks), dim3(kBlock), lds, s, p, n);
        else hipLaunchKernelGGL((k_direct<false, true, true>), dim3(blocks), dim3(kBlock), lds, s, p, n);
    } else if (p.sv.general) {
        if (p.sv.flat) hipLaunchKernelGGL((k_direct<true, true>), dim3(blocks), dim3(kBlock), lds, s, p, n);
        else hipLaunchKernelGGL((k_direct<false, true>), dim3(blocks), dim3(kBlock), lds, s, p, n);
    } else {
        if (p.sv.flat) hipLaunchKernelGGL((k_direct<true, false>), dim3(blocks), dim3(kBlock), lds, s, p, n);
        else hipLaunchKernelGGL((k_direct<false, false>), dim3(blocks), dim3(kBlock), lds, s, p, n);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Split pipeline for hierarchy scenes.  In the fused kernel the BVH walks inherit the shading code's ~100 VGPRs
// (4 waves / SIMD) and are latency-bound at that occupancy: 1.5 Gray/s on a 261 k-triangle mesh against ~5 Gray/s
// for a kernel that only traverses.  So here every iteration runs k_trace<false> (closest hits of the in-flight
// rays -> `hit`), k_shade (the path.cpp iteration on the precomputed hits; shadow rays are written next to the
// state) and k_trace<true> (visibility; adds the guarded contribution to the path's radiance).  The order of the
// floating-point additions into the radiance is the fused kernel's, so both pipelines produce identical samples.
constexpr uint32_t kFlagZombie = 2u;      // path already terminated, kept one iteration for its pending shadow ray

template <bool GENERAL, bool FLAT>
MTS_DEV bool step_deferred(const RenderParams &P, const LdsView &lds, PathState &s, Counters &c, Deferred &df) {
    return bounce_step<FLAT, false, FLAT ? 2 : 1, GENERAL>(P, lds, s, c, nullptr, &df);
}
template <bool GENERAL, bool FLAT>
MTS_DEV bool step_deferred(const RenderParams &P, const LdsView &lds, PathStateS &s, Counters &c, Deferred &df) {
    return bounce_step_spectral<FLAT, FLAT ? 2 : 1, GENERAL>(P, lds, s, c, &df);
}
// radiance += nee of an unoccluded shadow ray: the additions drain_shadow_ring / k_trace<any> make on the stored record
MTS_DEV void add_nee(PathState &s, const float (&nee)[4]) { s.res = mk3(s.res.x + nee[0], s.res.y + nee[1], s.res.z + nee[2]); }
MTS_DEV void add_nee(PathStateS &s, const float (&nee)[4]) {
#pragma unroll
    for (int k = 0; k < kWav; ++k) s.res.v[k] += nee[k];
}
MTS_DEV void finish_path(const RenderParams &P, const PathState &s) { store_result(P, s); }
MTS_DEV void finish_path(const RenderParams &P, const PathStateS &s) { store_result_spectral(P, s); }
MTS_DEV void start_path(const RenderParams &P, uint64_t ordinal, uint32_t lp, uint32_t j, PathState &s) { generate_path(P, ordinal, lp, j, s); }
MTS_DEV void start_path(const RenderParams &P, uint64_t ordinal, uint32_t lp, uint32_t j, PathStateS &s) { generate_path_spectral(P, ordinal, lp, j, s); }

// FLAT = false: hierarchy scene, closest hits precomputed by k_trace<false>; FLAT = true: LDS-resident scene, closest hit
// inline (wave-uniform primitive loop), only the shadow rays are queued
// Register budget of k_shade on hierarchy scenes (waves per SIMD the compiler must reach): the kernel streams the path state through
// HBM and gathers vertex data, so it lives on waves in flight.  5 waves (96 VGPRs) fit the RGB diffuse-only variant without spilling
// (+1.5 %); the others get 4 waves (128 VGPRs; the compiler's own choice was 136-185 VGPRs = 2-3 waves): spectral diffuse +2 %,
// general BSDFs +5 % (RGB) / +10 % (spectral, 128 bytes of scratch per lane) on the 261 k-triangle scenes.
#ifndef MTS_SHADE_WAVES_MIN
#define MTS_SHADE_WAVES_MIN 4
#endif
#ifndef MTS_SHADE_WAVES_MIN_RGB_DIFFUSE
#define MTS_SHADE_WAVES_MIN_RGB_DIFFUSE 5
#endif
template <typename State, bool GENERAL> struct ShadeWaves { static constexpr int kMin = MTS_SHADE_WAVES_MIN; };
template <> struct ShadeWaves<PathState, false> { static constexpr int kMin = MTS_SHADE_WAVES_MIN_RGB_DIFFUSE; };
// INLINE (flat scenes only): the shadow rays of consecutive 64-path chunks are collected in a per-wave LDS ring and resolved
// 64 at a time inside this kernel -- the any-hit loop then always runs on full waves (only about two thirds of the paths cast
// a shadow ray) -- and `nee` is added to the radiance the wave has already stored in its output segment.
constexpr uint32_t kShadowRing = 128u;                       // >= 63 queued + 64 pushed
template <typename State> struct ShadowRing { float4 *o, *d, *nee; uint32_t *slot; };

template <typename State, bool GENERAL>
MTS_DEV void drain_shadow_ring(const RenderParams &P, const LdsView &lds, const ShadowRing<State> &q, uint32_t head, uint32_t count,
                               size_t base, Counters &c) {
    // the wave's earlier stores to its output segment must have reached L2 before other lanes read-modify-write them
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const uint32_t lane = lane_id();
    if (lane < count) {
        const uint32_t k = (head + lane) & (kShadowRing - 1u);
        const float4 o = q.o[k], d = q.d[k];
        Hit h;
        if (!traverse<true, true>(P.sv, lds, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), o.w, d.w, h, c.tri_tests)) {
            const size_t slot = base + q.slot[k];
            float4 r = P.out.res[slot];
            const float4 e = q.nee[k];
            r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w;      // RGB: w = eta + 0
            P.out.res[slot] = r;
        }
    }
}

// Non-temporal pool / ray / hit / shadow-queue streams on hierarchy scenes: bit 0 = k_trace, bit 1 = k_shade
#ifndef MTS_NT_STREAMS
#define MTS_NT_STREAMS 3
#endif
#ifndef MTS_PRIMARY_SHADOW
#define MTS_PRIMARY_SHADOW 1  // 0 (experiment): the shadow rays of camera-path chunks go through the ring like all others
#endif
#ifndef MTS_SURV_ORDER
#define MTS_SURV_ORDER 1      // 0 (experiment): the pooled list of a workgroup in segment order, survivors and camera paths interleaved
#endif
template <typename State, bool GENERAL, bool FLAT, bool INLINE = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(FLAT ? MTS_BOUNCE_WAVES : ShadeWaves<State, GENERAL>::kMin, FLAT ? MTS_BOUNCE_WAVES : 8)))
void k_shade(const RenderParams P) {
    static_assert(FLAT || !INLINE, "the in-kernel shadow queue is for LDS-resident scenes");
    constexpr bool kNT = !FLAT && (MTS_NT_STREAMS & 2) != 0;      // hierarchy scenes: the pool streams bypass the caches the BVH lives in
    extern __shared__ float4 smem[];
    LdsView lds = {};                      // Geo<false> reads the scene tables from global memory
    // a launch covers the scheduling waves [wave_first, wave_last) (all of them, or one half when two launches share the GPU).
    // Pool drain (FLAT, gather_w > 4): once the sample cursors are dry the host lets one workgroup take the paths of gather_w
    // consecutive scheduling waves and leave the survivors at the front of the group -- hardware wave h fills the segments of the
    // waves h, h + 4, h + 8, ... of the group one after the other: the pool is compacted as it is advanced, and the number of
    // workgroups that stage the scene for a handful of paths shrinks with it.
    const uint32_t gw = (FLAT && P.gather_w > 4u) ? P.gather_w : 4u;
    const uint32_t hw = threadIdx.x >> 6;
    const uint32_t wave = gw > 4u ? P.wave_first + blockIdx.x * gw + hw : P.wave_first + ((blockIdx.x * kBlock + threadIdx.x) >> 6);
    const uint32_t wave_last = P.wave_last ? P.wave_last : P.n_waves;
    __shared__ uint32_t s_cnt[kBlock / 64u], s_surv[kBlock / 64u];
    __shared__ uint32_t s_pre[FLAT ? 1025 : 1];      // gather: s_pre[k] = paths in the group's waves before the k-th
    if (FLAT && gw > 4u) {
        const uint32_t g0 = wave - hw;
        for (uint32_t k = threadIdx.x; k < gw; k += kBlock) s_pre[k + 1u] = (g0 + k < wave_last) ? P.count_in[g0 + k] : 0u;
        if (threadIdx.x == 0u) s_pre[0] = 0u;
        __syncthreads();
        if (hw == 0u) {                      // inclusive prefix sums by one wave: a run of `per` entries per lane + a wave scan
            const uint32_t per = (gw + 63u) / 64u, ln = lane_id();
            uint32_t local = 0u;
            for (uint32_t i = 0; i < per; ++i) { const uint32_t idx = ln * per + i; if (idx < gw) local += s_pre[idx + 1u]; }
            uint32_t incl = local;
            for (uint32_t off = 1u; off < 64u; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if (ln >= off) incl += v; }
            uint32_t run = incl - local;
            for (uint32_t i = 0; i < per; ++i) { const uint32_t idx = ln * per + i; if (idx < gw) { run += s_pre[idx + 1u]; s_pre[idx + 1u] = run; } }
        }
        __syncthreads();
        if (s_pre[gw] == 0u) {               // nothing left in the whole group
            for (uint32_t k = threadIdx.x; k < gw; k += kBlock) if (g0 + k < wave_last) P.count_out[g0 + k] = 0u;
            return;
        }
        lds = lds_stage<true>(P.sv, smem);
    } else if (FLAT) {
        // While the pool drains at the end of a pass most scheduling waves have nothing left to do: a workgroup whose four
        // waves are all idle leaves before staging the scene into LDS (its output counts still have to be reset).
        const uint32_t n_own = wave < wave_last ? P.count_in[wave] : 0u;
        // the segment of a wave is [survivors of the last launch | camera paths generated by it]; the second half of the count array says
        // where the border is (a hint: any value <= the count gives a valid order, so stale entries of other schedules are harmless)
        if (lane_id() == 0u) { s_cnt[threadIdx.x >> 6] = n_own; s_surv[threadIdx.x >> 6] = (wave < wave_last && MTS_SURV_ORDER) ? min(P.count_in[P.n_waves + wave], n_own) : 0u; }
        const bool work = wave < wave_last && (n_own > 0u || P.cursor[wave] < P.cursor_end[wave]);
        if (!__syncthreads_or(work ? 1 : 0)) {
            if (wave < wave_last && lane_id() == 0u) {
                P.count_out[wave] = 0u;
                if (!INLINE) P.count_shadow[wave] = 0u;
            }
            return;
        }
        lds = lds_stage<true>(P.sv, smem);
    }
    ShadowRing<State> ring = {};
    if (INLINE) {
        float4 *qb = smem + P.lds_queue_offset + (size_t) (threadIdx.x >> 6) * (kShadowRing * 13u / 4u);
        ring.o = qb; ring.d = qb + kShadowRing; ring.nee = qb + 2u * kShadowRing;
        ring.slot = reinterpret_cast<uint32_t *>(qb + 3u * kShadowRing);
    }
    if (wave >= wave_last) return;
    const uint32_t lane = lane_id();
    const size_t base = (size_t) wave * P.seg_cap;
    uint32_t n_out = 0;
    Counters c = { 0u, 0u, 0u, 0u };
    uint32_t n_sh = 0;
    uint32_t q_head = 0, q_count = 0;
    // FLAT: the input segments of the workgroup's scheduling waves form one list whose 64-path chunks are dealt round-robin to
    // the waves -- at most one partial chunk per workgroup instead of one per wave while the pool drains.  Survivors go to the
    // output segment of the wave that processed them (seg_cap is a multiple of 64, so a wave never gets more than it can hold).
    const uint32_t wg_wave0 = wave - (threadIdx.x >> 6);
    const uint32_t n_valid = FLAT ? min((uint32_t) (kBlock / 64u), wave_last - wg_wave0) : 1u;
    // FLAT: the pooled list is ordered [survivors of wave 0 .. 3 | new camera paths of wave 0 .. 3] (reg[0 .. 7]): the chunks of the second
    // part hold camera rays only -- the samples of one or two pixels -- and take the cluster-culling closest-hit loop (traverse())
    uint32_t reg[2u * (kBlock / 64u)], surv4[kBlock / 64u], n_in = 0;
    if (FLAT && gw > 4u) {
        n_in = s_pre[gw];
    } else if (FLAT) {
#pragma unroll
        for (uint32_t g = 0; g < kBlock / 64u; ++g) {
            const uint32_t cg = g < n_valid ? s_cnt[g] : 0u;
            surv4[g] = g < n_valid ? s_surv[g] : 0u;
            reg[g] = surv4[g]; reg[kBlock / 64u + g] = cg - surv4[g];
            n_in += cg;
        }
    } else {
        n_in = __builtin_amdgcn_readfirstlane(P.count_in[wave]);
    }

    for (uint32_t i0 = FLAT ? 64u * (threadIdx.x >> 6) : 0u; i0 < n_in; i0 += 64u * n_valid) {
        State s;
        Deferred df;
        df.pending = false;
        bool alive = false, zombie_now = false;
        uint32_t depth0 = 0u;                                // depth of the path before this step (0: no path, or a zombie)
        if (i0 + lane < n_in) {
            size_t i = base + i0 + lane;
            if (FLAT && gw > 4u) {          // s_pre[k] <= index < s_pre[k + 1]: the k-th wave of the group holds the path
                const uint32_t idx = i0 + lane;
                uint32_t k = 0u;
                for (uint32_t step = gw >> 1; step; step >>= 1) if (s_pre[k + step] <= idx) k += step;
                i = (size_t) (wg_wave0 + k) * P.seg_cap + (idx - s_pre[k]);
            } else if (FLAT) {
                uint32_t r = 0u, j = i0 + lane;
#pragma unroll
                for (uint32_t g = 0; g + 1 < 2u * (kBlock / 64u); ++g)
                    if (r == g && j >= reg[g]) { j -= reg[g]; ++r; }
                constexpr uint32_t kW = kBlock / 64u;
                const uint32_t g = r >= kW ? r - kW : r;
                uint32_t off = 0u;
#pragma unroll
                for (uint32_t q = 0; q < kW; ++q) off = (r >= kW && g == q) ? surv4[q] : off;
                i = (size_t) (wg_wave0 + g) * P.seg_cap + off + j;
            }
            load_state<kNT>(P.in, i, s);
            if (s.flags & kFlagZombie) {
                finish_path(P, s);
            } else {
                depth0 = s.depth;
                if (!FLAT) {
                    const float4 h = ld_stream<kNT>(P.in.hit + i);
                    df.hit.t = h.x; df.hit.prim = __float_as_uint(h.y); df.hit.u = h.z; df.hit.v = h.w;
                    df.found = df.hit.prim != kNoPrim;
                }
                alive = step_deferred<GENERAL, FLAT>(P, lds, s, c, df);
                if (!alive) {
                    if (df.pending) { s.flags |= kFlagZombie; alive = true; zombie_now = true; }
                    else finish_path(P, s);
                }
            }
        }
        // A chunk of camera paths only (the samples of one or two pixels): nearly every lane casts a shadow ray and the rays leave a
        // pixel-sized patch towards one emitter, so they are resolved here, on the spot, by the cluster-culling any-hit loop instead of
        // going through the ring (same additions to the radiance in the same order: res + nee; a path that died after casting its ray
        // is finished at once instead of waiting one launch as a zombie)
        if (INLINE && MTS_FLAT_CULL && MTS_PRIMARY_SHADOW && P.sv.n_clusters > 1u && __ballot(depth0 != 1u && (i0 + lane < n_in)) == 0ull) {
            if (df.pending) {
                if (!traverse_flat_clustered_any(P.sv, lds, df.so, df.sd, df.smint, df.smaxt, c.tri_tests)) add_nee(s, df.nee);
                df.pending = false;
                if (zombie_now) { s.flags &= ~kFlagZombie; finish_path(P, s); alive = false; }
            }
        }
        // survivors are compacted to the front of the output segment, their shadow rays into a dense queue of their own
        const uint64_t m = __ballot(alive), ms = __ballot(alive && df.pending);
        if (alive) {
            const uint32_t slot = n_out + mask_rank(m);
            // gathering: the segment of wave + 4 q takes the slots [q seg_cap, (q + 1) seg_cap) of this hardware wave
            const size_t oidx = (FLAT && gw > 4u) ? ((size_t) wave + 4u * (slot / P.seg_cap)) * P.seg_cap + slot % P.seg_cap : base + slot;
            store_state<kNT>(P.out, oidx, s);
            if (df.pending) {
                if (INLINE) {
                    const uint32_t k = (q_head + q_count + mask_rank(ms)) & (kShadowRing - 1u);
                    ring.o[k] = make_float4(df.so.x, df.so.y, df.so.z, df.smint);
                    ring.d[k] = make_float4(df.sd.x, df.sd.y, df.sd.z, df.smaxt);
                    ring.nee[k] = make_float4(df.nee[0], df.nee[1], df.nee[2], df.nee[3]);
                    ring.slot[k] = (uint32_t) (oidx - base);      // may exceed the segment (gathering): an offset from `base` all the same
                } else {
                    const size_t q = base + n_sh + mask_rank(ms);
                    st_stream<kNT>(P.out.sh_o + q, make_float4(df.so.x, df.so.y, df.so.z, df.smint));
                    st_stream<kNT>(P.out.sh_d + q, make_float4(df.sd.x, df.sd.y, df.sd.z, df.smaxt));
                    st_stream<kNT>(P.out.nee + q, make_float4(df.nee[0], df.nee[1], df.nee[2], df.nee[3]));
                    st_stream<kNT>(P.out.sh_slot + q, slot);
                }
            }
        }
        n_out += (uint32_t) __popcll(m);
        n_sh += (uint32_t) __popcll(ms);
        if (INLINE) {
            q_count += (uint32_t) __popcll(ms);
            if (q_count >= 64u) {
                drain_shadow_ring<State, GENERAL>(P, lds, ring, q_head, 64u, base, c);
                q_head = (q_head + 64u) & (kShadowRing - 1u); q_count -= 64u;
            }
        }
    }
    if (INLINE && q_count > 0u) drain_shadow_ring<State, GENERAL>(P, lds, ring, q_head, q_count, base, c);

    const uint32_t n_surv = n_out;                           // border between the survivors and the camera paths generated below
    uint64_t cursor = P.cursor[wave];
    const uint64_t end = P.cursor_end[wave];
    while (n_out < P.target && cursor < end) {
        uint64_t left = end - cursor;
        uint32_t n_new = min(64u, P.target - n_out);
        if ((uint64_t) n_new > left) n_new = (uint32_t) left;
        if (lane < n_new) {
            State s;
            uint64_t ordinal; uint32_t lp, sj;
            cursor_sample(P, wave, cursor + lane, ordinal, lp, sj);
            start_path(P, ordinal, lp, sj, s);
            store_state<kNT>(P.out, base + n_out + lane, s);
        }
        n_out += n_new; cursor += n_new;
    }

    uint32_t tot[4] = { c.closest, c.any, c.segments, c.tri_tests };
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int off = 32; off > 0; off >>= 1) tot[k] += __shfl_xor(tot[k], off);
    if (FLAT && gw > 4u) {               // the segments this hardware wave filled: full ones, one partial, empty ones
        for (uint32_t q = lane; q < gw / 4u; q += 64u) {
            const uint32_t w2 = wave + 4u * q, lo = q * P.seg_cap;
            if (w2 < wave_last) P.count_out[w2] = n_out > lo ? min(n_out - lo, P.seg_cap) : 0u;
        }
    }
    if (lane == 0) {
        if (!(FLAT && gw > 4u)) P.count_out[wave] = n_out;
        if (FLAT && !(gw > 4u)) P.count_out[P.n_waves + wave] = n_surv;
        if (!INLINE) P.count_shadow[wave] = n_sh;
        P.cursor[wave] = cursor;
        uint64_t *ws = P.wave_stats + 4u * (size_t) wave;
        ws[0] += tot[0]; ws[1] += tot[1]; ws[2] += tot[2];
        if (FLAT) ws[3] += tot[3];
    }
}

// ---------------------------------------------------------------------------------------------
// End of a pass.  Once the sample cursors are dry the pool only shrinks, and a launch round (two or three dependent kernels per
// chain, every one spanning all scheduling waves) costs its fixed ~0.1 ms however few paths are left; the deepest paths need dozens
// of such rounds.  k_finish takes the pool as it stands and runs every remaining path to its end inside ONE launch: a workgroup of
// one hardware wave gathers the paths of `per` consecutive scheduling waves (prefix sums in LDS), a lane that finishes a path takes
// the next one of the workgroup's list at once, every loop trip is one path.cpp iteration (both ray queries inline, as in k_bounce:
// the floating-point operations on a sample and their order are those of the other schedules, the film is unchanged).
constexpr uint32_t kFinishMaxPer = 1024u, kFinishLdsDepth = 8u;      // the spill area is sized for k_trace AND for this depth (trace_spill_words)
template <bool GENERAL, bool FLAT>
MTS_DEV bool step_fused(const RenderParams &P, const LdsView &lds, PathState &s, Counters &c) { return bounce_step<FLAT, false, 0, GENERAL>(P, lds, s, c); }
template <bool GENERAL, bool FLAT>
MTS_DEV bool step_fused(const RenderParams &P, const LdsView &lds, PathStateS &s, Counters &c) { return bounce_step_spectral<FLAT, false, GENERAL>(P, lds, s, c); }

template <typename State, bool GENERAL, bool FLAT>
__global__ __launch_bounds__(64) void k_finish(const RenderParams P, uint32_t per) {
    extern __shared__ float4 smem[];
    __shared__ uint32_t s_pre[kFinishMaxPer + 1];
    const uint32_t lane = threadIdx.x, w0 = blockIdx.x * per;
    const uint32_t n_w = min(per, P.n_waves - min(P.n_waves, w0));
    {   // s_pre[k] = paths in the workgroup's scheduling waves before the k-th (a run of entries per lane + a wave scan)
        const uint32_t run = (n_w + 63u) / 64u;
        uint32_t local = 0u;
        for (uint32_t i = 0; i < run; ++i) { const uint32_t k = lane * run + i; if (k < n_w) local += P.count_in[w0 + k]; }
        uint32_t incl = local;
        for (uint32_t off = 1u; off < 64u; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        uint32_t sum = incl - local;
        if (lane == 0u) s_pre[0] = 0u;
        for (uint32_t i = 0; i < run; ++i) { const uint32_t k = lane * run + i; if (k < n_w) { sum += P.count_in[w0 + k]; s_pre[k + 1u] = sum; } }
    }
    __syncthreads();
    const uint32_t total = s_pre[n_w];
    if (total == 0u) return;
    for (uint32_t k = lane; k < n_w; k += 64u) P.count_out[w0 + k] = 0u;
    LdsView lds = {};
    if (FLAT) lds = lds_stage<true>(P.sv, smem);
    else {          // the BVH walks keep the first entries of their stack in LDS, the rest in this workgroup's slice of the k_trace spill area
        lds.stride = 64u; lds.stack = reinterpret_cast<uint32_t *>(smem);
        lds.stack_lds_depth = min(P.sv.stack_depth, kFinishLdsDepth);
        lds.spill = reinterpret_cast<StackEntry *>(P.trace_spill) + (size_t) blockIdx.x * (P.sv.stack_depth - lds.stack_lds_depth) * 64u + lane;
        lds.spill_stride = 64u;
    }
    Counters c = { 0u, 0u, 0u, 0u };
    State s;
    bool busy = false;
    uint32_t next = 0u;
    while (true) {
        const uint64_t m = __ballot(!busy);
        if (m != 0ull && next < total) {
            const uint32_t idx = next + mask_rank(m);
            if (!busy && idx < total) {
                uint32_t k = 0u;          // s_pre[k] <= idx < s_pre[k + 1]
                for (uint32_t step = kFinishMaxPer >> 1; step; step >>= 1) if (k + step < n_w && s_pre[k + step] <= idx) k += step;
                load_state(P.in, (size_t) (w0 + k) * P.seg_cap + (idx - s_pre[k]), s);
                if (s.flags & kFlagZombie) finish_path(P, s);      // only its shadow ray was outstanding
                else busy = true;
            }
            next += (uint32_t) __popcll(m);
        }
        if (__ballot(busy) == 0ull) {
            if (next >= total) break;
            continue;
        }
        if (busy && !step_fused<GENERAL, FLAT>(P, lds, s, c)) { finish_path(P, s); busy = false; }
    }
    uint32_t tot[4] = { c.closest, c.any, c.segments, c.tri_tests };
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int off = 32; off > 0; off >>= 1) tot[k] += __shfl_xor(tot[k], off);
    if (lane == 0u) {                  // the workgroup owns its scheduling waves: no atomics
        uint64_t *ws = P.wave_stats + 4u * (size_t) w0;
        ws[0] += tot[0]; ws[1] += tot[1]; ws[2] += tot[2]; ws[3] += tot[3];
    }
}

// Small passes (and, in experiment builds, MTSAMD_MEGA=1 for any pass): the whole pass as ONE launch of persistent lanes -- no pool, no
// launch rounds, no host polling.  A workgroup (one hardware wave) owns the samples of one scheduling wave; a lane whose path ends starts
// the next sample at once.  In steady state the wavefront schedule is faster (1.9 x on the 261 k-triangle mesh, 2 % on
// the Cornell box); a pass of a few samples per lane -- one iteration of an inverse-rendering loop at 1 spp -- is bound by the number of
// launches instead: a dozen launch rounds against one launch.  Same floating-point operations per sample in the same order.
template <typename State, bool GENERAL, bool FLAT>
__global__ __launch_bounds__(64) void k_mega(const RenderParams P) {
    extern __shared__ float4 smem[];
    const uint32_t lane = threadIdx.x, wave = blockIdx.x;
    uint64_t cursor = P.cursor[wave];
    const uint64_t end = P.cursor_end[wave];
    if (cursor >= end) return;
    LdsView lds = {};
    if (FLAT) lds = lds_stage<true>(P.sv, smem);
    else {
        lds.stride = 64u; lds.stack = reinterpret_cast<uint32_t *>(smem);
        lds.stack_lds_depth = min(P.sv.stack_depth, kFinishLdsDepth);
        lds.spill = reinterpret_cast<StackEntry *>(P.trace_spill) + (size_t) blockIdx.x * (P.sv.stack_depth - lds.stack_lds_depth) * 64u + lane;
        lds.spill_stride = 64u;
    }
    Counters c = { 0u, 0u, 0u, 0u };
    State s;
    bool busy = false;
    while (true) {
        const uint64_t m = __ballot(!busy);
        if (m != 0ull && cursor < end) {
            const uint64_t v = cursor + mask_rank(m);
            if (!busy && v < end) {
                uint64_t ordinal; uint32_t lp, sj;
                cursor_sample(P, wave, v, ordinal, lp, sj);
                start_path(P, ordinal, lp, sj, s);
                busy = true;
            }
            cursor += (uint64_t) __popcll(m);
        }
        if (__ballot(busy) == 0ull) {
            if (cursor >= end) break;
            continue;
        }
        if (busy && !step_fused<GENERAL, FLAT>(P, lds, s, c)) { finish_path(P, s); busy = false; }
    }
    uint32_t tot[4] = { c.closest, c.any, c.segments, c.tri_tests };
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int off = 32; off > 0; off >>= 1) tot[k] += __shfl_xor(tot[k], off);
    if (lane == 0u) {
        P.cursor[wave] = end;
        uint64_t *ws = P.wave_stats + 4u * (size_t) wave;
        ws[0] += tot[0]; ws[1] += tot[1]; ws[2] += tot[2]; ws[3] += tot[3];
    }
}

hipError_t launch_mega(const RenderParams &p, hipStream_t s) {
    const size_t lds = p.sv.flat ? lds_bytes(p.sv, 64u) : sizeof(StackEntry) * (std::min(p.sv.stack_depth, kFinishLdsDepth) + 1u) * 64u;
    const uint32_t blocks = p.n_waves;
    if (p.sv.flat) {
        if (p.spectral && p.sv.general) hipLaunchKernelGGL((k_mega<PathStateS, true, true>), dim3(blocks), dim3(64), lds, s, p);
        else if (p.spectral) hipLaunchKernelGGL((k_mega<PathStateS, false, true>), dim3(blocks), dim3(64), lds, s, p);
        else if (p.sv.general) hipLaunchKernelGGL((k_mega<PathState, true, true>), dim3(blocks), dim3(64), lds, s, p);
        else hipLaunchKernelGGL((k_mega<PathState, false, true>), dim3(blocks), dim3(64), lds, s, p);
    } else {
        if (p.spectral && p.sv.general) hipLaunchKernelGGL((k_mega<PathStateS, true, false>), dim3(blocks), dim3(64), lds, s, p);
        else if (p.spectral) hipLaunchKernelGGL((k_mega<PathStateS, false, false>), dim3(blocks), dim3(64), lds, s, p);
        else if (p.sv.general) hipLaunchKernelGGL((k_mega<PathState, true, false>), dim3(blocks), dim3(64), lds, s, p);
        else hipLaunchKernelGGL((k_mega<PathState, false, false>), dim3(blocks), dim3(64), lds, s, p);
    }
    return hipGetLastError();
}

hipError_t launch_finish(const RenderParams &p, uint64_t alive, hipStream_t s) {
    // about four paths per lane (alive: upper bound of the paths left), at least 2048 workgroups if there are that many scheduling waves
    const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(alive / 256u, 2048u), p.n_waves);
    const uint32_t per = std::min((uint32_t) ((p.n_waves + want - 1u) / want), kFinishMaxPer);
    const uint32_t blocks = (p.n_waves + per - 1u) / per;
    const size_t lds = p.sv.flat ? lds_bytes(p.sv, 64u) : sizeof(StackEntry) * (std::min(p.sv.stack_depth, kFinishLdsDepth) + 1u) * 64u;
    if (p.sv.flat) {
        if (p.spectral && p.sv.general) hipLaunchKernelGGL((k_finish<PathStateS, true, true>), dim3(blocks), dim3(64), lds, s, p, per);
        else if (p.spectral) hipLaunchKernelGGL((k_finish<PathStateS, false, true>), dim3(blocks), dim3(64), lds, s, p, per);
        else if (p.sv.general) hipLaunchKernelGGL((k_finish<PathState, true, true>), dim3(blocks), dim3(64), lds, s, p, per);
        else hipLaunchKernelGGL((k_finish<PathState, false, true>), dim3(blocks), dim3(64), lds, s, p, per);
    } else {
        if (p.spectral && p.sv.general) hipLaunchKernelGGL((k_finish<PathStateS, true, false>), dim3(blocks), dim3(64), lds, s, p, per);
        else if (p.spectral) hipLaunchKernelGGL((k_finish<PathStateS, false, false>), dim3(blocks), dim3(64), lds, s, p, per);
        else if (p.sv.general) hipLaunchKernelGGL((k_finish<PathState, true, false>), dim3(blocks), dim3(64), lds, s, p, per);
        else hipLaunchKernelGGL((k_finish<PathState, false, false>), dim3(blocks), dim3(64), lds, s, p, per);
    }
    return hipGetLastError();
}

// k_trace<false>: closest hit of every path's ray (input pool).
// k_trace<true>: visibility of the queued shadow rays (output pool of k_shade), radiance[slot] += nee if unoccluded.
// A workgroup drains the rays of a GROUP of consecutive scheduling waves back to back (dynamic fetch, see the kernel body): kTraceGroup
// waves on hierarchy scenes, kFlatGroup on LDS-resident scenes (shadow queues only: about a third of the paths queue a shadow ray;
// 64-thread workgroups sized to the queues were measured slower).
// kTraceBlock: threads per workgroup on hierarchy scenes.  256; MTS_TRACE_BLOCK=1024 (two workgroups per CU, each with a copy of the
// top of the BVH in LDS next to its stack rows) was built and measured slower.
#ifndef MTS_TRACE_BLOCK
#define MTS_TRACE_BLOCK 256
#endif
// Scheduling waves per workgroup: the longer a workgroup's ray list, the smaller the share of its rounds that run in the tail of the
// list (a few lanes finishing the longest walks, profiles/r03_trace_phases.txt), the fewer workgroups a launch has to balance:
// 4 / 8 / 16 / 32 / 64 waves per 256 threads -> mesh render 55.1 / 51.5 / 49.9 / 51.0 / 58.8 ms (profiles/r03_ab_trace_group.txt).
#ifndef MTS_TRACE_GROUP
#define MTS_TRACE_GROUP (MTS_TRACE_BLOCK / 16)
#endif
constexpr uint32_t kTraceBlock = MTS_TRACE_BLOCK;      // threads per k_trace workgroup (hierarchy scenes)
constexpr uint32_t kTraceGroup = MTS_TRACE_GROUP;      // hierarchy scenes: scheduling waves per workgroup (a power of two)
constexpr uint32_t kFlatGroup = 8u;                    // LDS-resident scenes (shadow queues only)
static_assert((kTraceGroup & (kTraceGroup - 1u)) == 0u, "locate() searches a power-of-two table");
static_assert(kChainAlign % kTraceGroup == 0u, "a k_trace group must not straddle two launch chains");
// LDS part of k_trace's per-lane stack: entries of 8 bytes (BVH4: reference + entry distance); the full 41-entry stack of the
// 261 k-triangle mesh would cap the CU at a fraction of a workgroup.  Deeper entries go to a global spill area (rare).
#ifndef MTS_TRACE_LDS_DEPTH
#define MTS_TRACE_LDS_DEPTH (MTS_BVH4 ? (MTS_TRACE_BLOCK >= 1024 ? 6 : 8) : 16)
#endif
constexpr uint32_t kTraceLdsDepth = MTS_TRACE_LDS_DEPTH;
// Top of the BVH4 staged in LDS by every k_trace workgroup: the first kTraceTopNodes nodes in BFS order (64 B each).
#ifndef MTS_TRACE_TOP_NODES
#define MTS_TRACE_TOP_NODES 0
#endif
constexpr uint32_t kTraceTopNodes = MTS_TRACE_TOP_NODES;

// Register budget of k_trace: 64 VGPRs = 8 waves per SIMD.  The walks are bound by the latency of their node / triangle fetches
// (L2 and beyond for a 261 k-triangle scene), which only more waves in flight hide: 5 waves (84 VGPRs, the compiler's own choice)
// -> 8 waves: +7.5 % on the whole render (RGB and spectral).
#ifndef MTS_TRACE_WAVES
#define MTS_TRACE_WAVES 8
#endif
template <bool ANY, bool FLAT = false>
__global__ __launch_bounds__(FLAT ? kBlock : kTraceBlock)
#if MTS_TRACE_WAVES > 0
__attribute__((amdgpu_waves_per_eu(MTS_TRACE_WAVES, MTS_TRACE_WAVES)))
#endif
void k_trace(const RenderParams P) {
    constexpr uint32_t kShadowGroup = FLAT ? kFlatGroup : kTraceGroup;
    extern __shared__ float4 smem[];
    __shared__ uint32_t s_pre[kShadowGroup + 1u];            // s_pre[g] = work items of the group's scheduling waves before the g-th
    LdsView lds = {};
    const uint32_t n_top = (FLAT || kTraceTopNodes == 0u) ? 0u : P.trace_top_nodes;
    if (FLAT) {
        lds = lds_stage<true>(P.sv, smem);
    } else {
        // LDS: [top of the BVH4: n_top nodes of 4 x 16 B][traversal stack rows]
        uint4 *top = reinterpret_cast<uint4 *>(smem);
        for (uint32_t i = threadIdx.x; i < 4u * n_top; i += blockDim.x) top[i] = P.sv.wnodes[i];
        lds.stride = blockDim.x;
        lds.stack = reinterpret_cast<uint32_t *>(smem + 4u * n_top);
    }
    uint32_t tri_tests = 0;
    // work list of this workgroup: the shadow queues (ANY) or the path slots (closest hit) of kShadowGroup scheduling waves
    const PoolView &pool = ANY ? P.out : P.in;
    // a launch covers the scheduling waves [wave_first, wave_last) (wave_first is a multiple of kShadowGroup)
    const uint32_t wave_last = P.wave_last ? P.wave_last : P.n_waves;
    const uint32_t group = P.wave_first / kShadowGroup + blockIdx.x;      // global group index
    const uint32_t w0 = group * kShadowGroup;
    const uint32_t *counts = ANY ? P.count_shadow : P.count_in;
    // small groups: the counts are wave-uniform and stay in scalar registers; large groups (1024-thread workgroups): prefix sums in LDS
    constexpr bool kCountsInLds = kShadowGroup > 8u;
    uint32_t cnt[kCountsInLds ? 1u : kShadowGroup], total = 0;
    if (kCountsInLds) {
        if (threadIdx.x < 64u) {                             // prefix sums of the group's counts by the first wave
            const uint32_t ln = threadIdx.x;
            uint32_t incl = (ln < kShadowGroup && w0 + ln < wave_last) ? counts[w0 + ln] : 0u;
            for (uint32_t off = 1u; off < kShadowGroup; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if (ln >= off) incl += v; }
            if (ln < kShadowGroup) s_pre[ln + 1u] = incl;
            if (ln == 0u) s_pre[0] = 0u;
        }
    } else {
#pragma unroll
        for (uint32_t g = 0; g < kShadowGroup; ++g) { cnt[g] = (w0 + g < wave_last) ? counts[w0 + g] : 0u; total += cnt[g]; }
    }
    __shared__ uint32_t s_next;
    if (threadIdx.x == 0) s_next = 0u;
    __syncthreads();
    if (kCountsInLds) total = s_pre[kShadowGroup];
    auto locate = [&](uint32_t idx) -> size_t {              // work item -> pool index
        if (kCountsInLds) {                                  // s_pre[g] <= idx < s_pre[g + 1]
            uint32_t g = 0u;
#pragma unroll
            for (uint32_t step = kShadowGroup >> 1; step; step >>= 1) if (s_pre[g + step] <= idx) g += step;
            return (size_t) (w0 + g) * P.seg_cap + (idx - s_pre[g]);
        }
        uint32_t wave = w0, i = idx;
#pragma unroll
        for (uint32_t g = 0; g + 1 < kShadowGroup; ++g)
            if (wave == w0 + g && i >= cnt[g]) { i -= cnt[g]; ++wave; }
        return (size_t) wave * P.seg_cap + i;
    };
    constexpr bool kNT = !FLAT && (MTS_NT_STREAMS & 1) != 0;
    auto retire_any = [&](size_t k) {                        // unoccluded shadow ray: radiance[slot] += nee
        const size_t slot = (k / P.seg_cap) * P.seg_cap + ld_stream<kNT>(pool.sh_slot + k);
        float4 r = pool.res[slot];
        const float4 e = ld_stream<kNT>(pool.nee + k);
        r.x += e.x; r.y += e.y; r.z += e.z; r.w += e.w;      // RGB: w = eta + 0
        pool.res[slot] = r;
    };
    if (FLAT) {
        Hit h;
        for (uint32_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
            const size_t k = locate(idx);
            const float4 o = pool.sh_o[k], d = pool.sh_d[k];
            if (!traverse<true, true>(P.sv, lds, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), o.w, d.w, h, tri_tests)) retire_any(k);
        }
    } else {
        // Dynamic ray fetch: a lane that has finished its walk takes the next item of the workgroup's list at once instead of
        // idling until the slowest lane of its wave is done (the walks of incoherent rays differ several-fold in length: with
        // one ray per lane and loop trip only 22-29 % of the VALU lane-cycles were useful).
        const uint32_t lane = lane_id();
        // LDS holds the first P.trace_lds_depth stack entries of every lane; deeper entries spill to this workgroup's slice of
        // P.trace_spill ([entry][thread])
        const uint32_t spill_depth = P.sv.stack_depth > P.trace_lds_depth ? P.sv.stack_depth - P.trace_lds_depth : 0u;
        const WalkStack st = { reinterpret_cast<StackEntry *>(lds.stack) + threadIdx.x, log2_stride(lds.stride), P.trace_lds_depth,
                               reinterpret_cast<StackEntry *>(P.trace_spill) + ((size_t) (ANY ? (P.n_waves + kShadowGroup - 1u) / kShadowGroup : 0u) + group) * spill_depth * kTraceBlock + threadIdx.x, blockDim.x,
                               reinterpret_cast<const uint4 *>(smem), n_top };
        BvhWalk w;
        w.cur = kNoNode; w.sp = 0u; w.found = false;
        constexpr uint32_t kNoItem = 0xffffffffu;
        uint32_t k = kNoItem;                                // pool index of the work item the lane holds (the pools have fewer than 2^32 slots)
        bool exhausted = false;
        while (true) {
            const bool need = w.cur == kNoNode;
            MTS_PROF(ANY, 12);                               // iterations of the work loop
            if (need && k != kNoItem) {                      // retire the finished item
                MTS_PROF(ANY, 14);
                if (ANY) { if (!w.found) retire_any(k); }
                else st_stream<kNT>(pool.hit + k, w.found ? make_float4(w.best, __uint_as_float(w.best_prim), w.hit.u, w.hit.v)
                                                          : make_float4(__builtin_inff(), __uint_as_float(kNoPrim), 0.0f, 0.0f));
                k = kNoItem;
            }
            const uint64_t m = __ballot(need);
            if (m && !exhausted) {
                const uint32_t first = (uint32_t) __ffsll((long long) m) - 1u, want = (uint32_t) __popcll(m);
                uint32_t base = 0u;
                if (lane == first) base = atomicAdd(&s_next, want);
                base = __shfl(base, (int) first);
                exhausted = base + want >= total;
                if (need) {
                    MTS_PROF(ANY, 10);                       // fetches
                    const uint32_t idx = base + mask_rank(m);
                    if (idx < total) {
                        const uint32_t item = (uint32_t) locate(idx);
                        if (ANY) {
                            const float4 o = ld_stream<kNT>(pool.sh_o + item), d = ld_stream<kNT>(pool.sh_d + item);
                            walk_begin(w, P.sv, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), o.w, d.w);
                            k = item;
                        } else if (!((pool.misc[item].y >> 16) & kFlagZombie)) {      // zombies only wait for their shadow ray
                            const float4 o = ld_stream<kNT>(pool.ray_o + item), d = ld_stream<kNT>(pool.ray_d + item);
                            walk_begin(w, P.sv, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), o.w, d.w);
                            k = item;
                        }
                    }
                }
            }
            if (__ballot(w.cur != kNoNode) == 0ull) {
                if (exhausted) break;
                continue;                                    // only zombies were fetched: try again
            }
            MTS_PROF_MASK(ANY, exhausted ? 22 : 20, __ballot(w.cur != kNoNode));      // rounds before / after the list ran out: lanes with a ray
            if (__ballot(w.cur != kNoNode && w.far) != 0ull) {      // wave-uniform choice of the slab-test form
                if (w.cur != kNoNode) walk_round<ANY, true, kTraceTopNodes != 0u>(w, P.sv, st, tri_tests);
            } else {
                if (w.cur != kNoNode) walk_round<ANY, false, kTraceTopNodes != 0u>(w, P.sv, st, tri_tests);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) tri_tests += __shfl_xor(tri_tests, off);
    if (lane_id() == 0 && tri_tests)
        atomicAdd(reinterpret_cast<unsigned long long *>(P.wave_stats + 4u * (size_t) w0 + 3u),
                  (unsigned long long) tri_tests);
}

#if MTS_TRACE_PROF
// experiment builds: the phase counters of device_scene.h (scripts/debug/trace_prof.py)
extern "C" __attribute__((visibility("default"))) int mtsamd_debug_trace_prof(unsigned long long *out, int reset) {
    static unsigned long long host[64 * 64];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace_prof), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 64; ++i) { out[i] = 0; for (int c = 0; c < 64; ++c) out[i] += host[64 * c + i]; }
    if (reset) { for (auto &h : host) h = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_prof), host, sizeof(host)) != hipSuccess) return -1; }
    return 0;
}
#endif

uint32_t trace_top_nodes(const SceneView &sv) { return std::min(sv.n_wnodes, kTraceTopNodes); }
uint32_t trace_group() { return kTraceGroup; }
size_t trace_lds_bytes(const SceneView &sv) {      // top of the tree + the stack rows (+ the scratch row of stack_row())
    return (size_t) 64 * trace_top_nodes(sv) + sizeof(StackEntry) * (std::min(sv.stack_depth, kTraceLdsDepth) + 1u) * kTraceBlock;
}
uint32_t trace_lds_depth(const SceneView &sv) { return std::min(sv.stack_depth, kTraceLdsDepth); }
// spill area shared by k_trace (closest-hit and any-hit launches may overlap: two slices per group) and k_finish (its workgroups are
// single waves that keep kFinishLdsDepth entries in LDS)
size_t trace_spill_words(const SceneView &sv, uint32_t n_waves) {
    const uint32_t spill = sv.stack_depth > kTraceLdsDepth ? sv.stack_depth - kTraceLdsDepth : 0u;
    const uint32_t spill_f = sv.stack_depth > kFinishLdsDepth ? sv.stack_depth - kFinishLdsDepth : 0u;
    const size_t trace = (size_t) 2 * ((n_waves + kTraceGroup - 1) / kTraceGroup) * spill * kTraceBlock;
    const size_t finish = (size_t) n_waves * spill_f * 64u;      // at most one k_finish workgroup per scheduling wave
    return (sizeof(StackEntry) / 4) * std::max(trace, finish);
}
static hipError_t allow_lds(const void *fn, size_t bytes) {
    return bytes > 48u * 1024u ? hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes) : hipSuccess;
}

// split pipeline of hierarchy scenes, one stage at a time: 0 = k_trace<closest>, 1 = k_shade, 2 = k_trace<any>.  Stage 2 of one
// iteration and stage 0 of the next touch disjoint arrays, so the host runs them on two streams (api.cpp).
hipError_t launch_split_stage(const RenderParams &p, int stage, hipStream_t s) {
    const uint32_t n_launch = (p.wave_last ? p.wave_last : p.n_waves) - p.wave_first;
    const uint32_t shade_blocks = (n_launch * 64u + kBlock - 1) / kBlock, trace_blocks = (n_launch + kTraceGroup - 1) / kTraceGroup;
    if (stage == 0) {
        if (hipError_t e = allow_lds(reinterpret_cast<const void *>(&k_trace<false, false>), trace_lds_bytes(p.sv))) return e;
        hipLaunchKernelGGL((k_trace<false, false>), dim3(trace_blocks), dim3(kTraceBlock), trace_lds_bytes(p.sv), s, p);
    } else if (stage == 1) {
        if (p.spectral && p.sv.general) hipLaunchKernelGGL((k_shade<PathStateS, true, false>), dim3(shade_blocks), dim3(kBlock), 0, s, p);
        else if (p.spectral) hipLaunchKernelGGL((k_shade<PathStateS, false, false>), dim3(shade_blocks), dim3(kBlock), 0, s, p);
        else if (p.sv.general) hipLaunchKernelGGL((k_shade<PathState, true, false>), dim3(shade_blocks), dim3(kBlock), 0, s, p);
        else hipLaunchKernelGGL((k_shade<PathState, false, false>), dim3(shade_blocks), dim3(kBlock), 0, s, p);
    } else {
        if (hipError_t e = allow_lds(reinterpret_cast<const void *>(&k_trace<true, false>), trace_lds_bytes(p.sv))) return e;
        hipLaunchKernelGGL((k_trace<true, false>), dim3(trace_blocks), dim3(kTraceBlock), trace_lds_bytes(p.sv), s, p);
    }
    return hipGetLastError();
}

hipError_t launch_bounce(const RenderParams &p_, hipStream_t s) {
    RenderParams p = p_;
    if (p.split == 3) {       // LDS-resident scene, one kernel: shadow rays collected in a per-wave LDS ring and resolved 64 at a time
        const uint32_t n_launch = (p.wave_last ? p.wave_last : p.n_waves) - p.wave_first;
        const uint32_t shade_blocks = p.gather_w > 4u ? (n_launch + p.gather_w - 1u) / p.gather_w : (n_launch * 64u + kBlock - 1) / kBlock;
        const size_t scene_bytes = (bounce_lds_bytes(p.sv) + 15u) & ~(size_t) 15u;
        p.lds_queue_offset = (uint32_t) (scene_bytes / 16u);
        const size_t lds = scene_bytes + (size_t) (kBlock / 64u) * kShadowRing * 52u;
        if (p.spectral && p.sv.general) hipLaunchKernelGGL((k_shade<PathStateS, true, true, true>), dim3(shade_blocks), dim3(kBlock), lds, s, p);
        else if (p.spectral) hipLaunchKernelGGL((k_shade<PathStateS, false, true, true>), dim3(shade_blocks), dim3(kBlock), lds, s, p);
        else if (p.sv.general) hipLaunchKernelGGL((k_shade<PathState, true, true, true>), dim3(shade_blocks), dim3(kBlock), lds, s, p);
        else hipLaunchKernelGGL((k_shade<PathState, false, true, true>), dim3(shade_blocks), dim3(kBlock), lds, s, p);
        return hipGetLastError();
    }
    if (p.split == 2) {       // LDS-resident scene: closest hit + shading fused, shadow rays queued and resolved in dense batches
        const uint32_t shade_blocks = (p.n_waves * 64u + kBlock - 1) / kBlock;
        const size_t lds = bounce_lds_bytes(p.sv);
        if (p.spectral && p.sv.general) hipLaunchKernelGGL((k_shade<PathStateS, true, true>), dim3(shade_blocks), dim3(kBlock), lds, s, p);
        else if (p.spectral) hipLaunchKernelGGL((k_shade<PathStateS, false, true>), dim3(shade_blocks), dim3(kBlock), lds, s, p);
        else if (p.sv.general) hipLaunchKernelGGL((k_shade<PathState, true, true>), dim3(shade_blocks), dim3(kBlock), lds, s, p);
        else hipLaunchKernelGGL((k_shade<PathState, false, true>), dim3(shade_blocks), dim3(kBlock), lds, s, p);
        hipLaunchKernelGGL((k_trace<true, true>), dim3((p.n_waves + kFlatGroup - 1) / kFlatGroup), dim3(kBlock), lds, s, p);
        return hipGetLastError();
    }
    if (p.split) {
        hipError_t e = launch_split_stage(p, 0, s);
        if (e == hipSuccess) e = launch_split_stage(p, 1, s);
        if (e == hipSuccess) e = launch_split_stage(p, 2, s);
        return e;
    }
    uint32_t blocks = (p.n_waves * 64u + kBlock - 1) / kBlock;
    if (p.spectral) {
        if (p.sv.general == 2u) {
            if (p.sv.flat) hipLaunchKernelGGL((k_bounce_spectral<true, true, true>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
            else hipLaunchKernelGGL((k_bounce_spectral<false, true, true>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
        } else if (p.sv.general) {
            if (p.sv.flat) hipLaunchKernelGGL((k_bounce_spectral<true, true>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
            else hipLaunchKernelGGL((k_bounce_spectral<false, true>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
        } else {
            if (p.sv.flat) hipLaunchKernelGGL((k_bounce_spectral<true, false>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
            else hipLaunchKernelGGL((k_bounce_spectral<false, false>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
        }
        return hipGetLastError();
    }
    if (p.sv.general == 2u) {
        if (p.sv.flat) hipLaunchKernelGGL((k_bounce<true, true, true>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
        else hipLaunchKernelGGL((k_bounce<false, true, true>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
    } else if (p.sv.general) {
        if (p.sv.flat) hipLaunchKernelGGL((k_bounce<true, true>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
        else hipLaunchKernelGGL((k_bounce<false, true>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
    } else {
        if (p.sv.flat) hipLaunchKernelGGL((k_bounce<true, false>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
        else hipLaunchKernelGGL((k_bounce<false, false>), dim3(blocks), dim3(kBlock), bounce_lds_bytes(p.sv), s, p);
    }
    return hipGetLastError();
}

// reconstruction filter (rfilter.h:62-65, gaussian.cpp:45-47, box.cpp:34-36, tent.cpp:33-35, catmullrom.cpp:29-43,
// mitchell.cpp:41-56, lanczos.cpp:38-48); FilterView::alpha / bias carry tent's 1 / radius and mitchell's B / C
MTS_DEV float cubic_filter(float x, float B, float C) {
    x = fabsf(x);
    const float x2 = x * x, x3 = x2 * x;
    const float result = (1.0f / 6.0f) * (x < 1.0f
        ? (12.0f - 9.0f * B - 6.0f * C) * x3 + (-18.0f + 12.0f * B + 6.0f * C) * x2 + (6.0f - 2.0f * B)
        : (-B - 6.0f * C) * x3 + (6.0f * B + 30.0f * C) * x2 + (-12.0f * B - 48.0f * C) * x + (8.0f * B + 24.0f * C));
    return x < 2.0f ? result : 0.0f;
}
MTS_DEV float filter_eval(const FilterView &f, float x) {
    switch (f.kind) {
    case 0: return fmaxf(0.0f, lm_exp(f.alpha * (x * x)) - f.bias);
    case 2: return fmaxf(0.0f, 1.0f - fabsf(x * f.alpha));
    case 3: return cubic_filter(x, 0.0f, 0.5f);
    case 4: return cubic_filter(x, f.alpha, f.bias);
    case 5: {
        x = fabsf(x);
        const float x1 = kPi * x, x2 = x1 / f.radius, result = (lm_sin(x1) * lm_sin(x2)) / (x1 * x2);
        return x < kEpsilon ? 1.0f : (x > f.radius ? 0.0f : result);
    }
    default: return fabsf(x) <= f.radius ? 1.0f : 0.0f;
    }
}
MTS_DEV float filter_weight(const FilterView &f, float x) {
    if (f.analytic) return filter_eval(f, x);
    int idx = min((int) fabsf(x * f.scale_factor), 31);
    return f.table[idx];
}


// ---------------------------------------------------------------------------------------------
// Reverse-mode derivative of the rendered image with respect to diffuse reflectances (constant colours and
// bitmap texels), the role Enoki's autodiff plays for mitsuba.python.autodiff.render (autodiff.py:6-91,121-194).
// One thread replays one camera sample with the same PCG32 stream as the primal pass, records its vertices,
// and sweeps them backwards:   Y_k = Nc_k + X_{k+1},  dL/drho_k = delta * T'_k * Y_k,  X_k = E_k + invq_k rho_k Y_k,
// where delta = dLoss/dRadiance of this sample = sum over its filter footprint of w * dLoss/dImage / (W + 1e-8)
// (Image = values / (weight + 1e-8), autodiff.py:80-91); the sweep below also carries the derivative of the
// Russian-roulette factor 1/q(T) (path.cpp:137-141), as Enoki's autodiff does.
constexpr int kAdjointMaxDepth = 16;

// dLoss/dRadiance of the camera sample at film position `pos`: the adjoint of ImageBlock::put (imageblock.cpp:117-169, block = the
// crop window, no border) followed by Image = values / (weight + 1e-8) (autodiff.py:80-91)
MTS_DEV f3 adjoint_delta(const AdjointParams &A, float2 pos) {
    const RenderParams &P = A.rp;
    const FilterView &f = A.filter;
    f3 delta = mk3(0.0f, 0.0f, 0.0f);
    const float px = pos.x - ((float) P.crop_x + 0.5f), py = pos.y - ((float) P.crop_y + 0.5f);
    if (f.radius > 1.0f) {
        const int lox = max((int) ceilf(px - f.radius), 0), loy = max((int) ceilf(py - f.radius), 0);
        const int hix = min((int) floorf(px + f.radius), P.crop_w - 1), hiy = min((int) floorf(py + f.radius), P.crop_h - 1);
        const float bx = (float) (uint32_t) lox - px, by = (float) (uint32_t) loy - py;
        for (int yr = 0; yr < f.taps && loy + yr <= hiy; ++yr) {
            const float wy = filter_weight(f, by + (float) yr);
            for (int xr = 0; xr < f.taps && lox + xr <= hix; ++xr) {
                const float w = wy * filter_weight(f, bx + (float) xr);
                const size_t pix = (size_t) (loy + yr) * P.crop_w + (size_t) (lox + xr);
                const float iw = w / (A.film[5 * pix + 4] + 1e-8f);
                delta.x += iw * A.dimage[3 * pix]; delta.y += iw * A.dimage[3 * pix + 1]; delta.z += iw * A.dimage[3 * pix + 2];
            }
        }
    } else {
        const int lox = (int) ceilf(px - 0.5f), loy = (int) ceilf(py - 0.5f);
        if (lox >= 0 && loy >= 0 && lox < P.crop_w && loy < P.crop_h) {
            const size_t pix = (size_t) loy * P.crop_w + (size_t) lox;
            const float iw = 1.0f / (A.film[5 * pix + 4] + 1e-8f);
            delta = mk3(iw * A.dimage[3 * pix], iw * A.dimage[3 * pix + 1], iw * A.dimage[3 * pix + 2]);
        }
    }
    return delta;
}

template <bool FLAT>
__global__ __launch_bounds__(kBlock) void k_adjoint(const AdjointParams A) {
    extern __shared__ float4 smem[];
    const RenderParams &P = A.rp;
    const LdsView lds = lds_stage<FLAT>(P.sv, smem);
    __shared__ float s_grad[3 * 32];                       // per-workgroup sums for constant reflectances
    __shared__ float s_grad_em[3 * 32];                    // ... and for the radiance of area lights
    for (uint32_t i = threadIdx.x; i < 3u * 32u; i += kBlock) s_grad[i] = s_grad_em[i] = 0.0f;
    __syncthreads();
    const uint32_t spp = (uint32_t) P.spp;
    for (uint64_t k = (uint64_t) blockIdx.x * kBlock + threadIdx.x; k < A.n_samples; k += (uint64_t) gridDim.x * kBlock) {
        PathState s; float2 pos;
        generate_path(P, k, (uint32_t) (k / spp), (uint32_t) (k % spp), s, &pos);
        const f3 delta = adjoint_delta(A, pos);
        // ---- replay the path, remembering its vertices
        VertexRec rec[kAdjointMaxDepth];
        int n = 0;
        Counters c = { 0u, 0u, 0u, 0u };
        bool alive = true;
        while (alive && n < kAdjointMaxDepth) { alive = bounce_step<FLAT, true>(P, lds, s, c, &rec[n]); ++n; }
        // ---- backward sweep; a = dLoss/dT_v.  q = min(hmax(T) eta^2, .95) is differentiated like Enoki does (the
        // gradient flows to the maximal channel when q is not clamped); the survival test is not differentiable.
        f3 a = mk3(0.0f, 0.0f, 0.0f);
        for (int v = n - 1; v >= 0; --v) {
            const VertexRec &r = rec[v];
            if (A.grad_emitter) {     // radiance is linear in Le: delta * T_v * ew (emitter hit) + delta * T'_v rho_v nk (emitter sampled)
                if (r.em_hit >= 0 && r.em_hit < 32) {
                    atomicAdd(&s_grad_em[3 * r.em_hit], delta.x * r.T.x * r.ew); atomicAdd(&s_grad_em[3 * r.em_hit + 1], delta.y * r.T.y * r.ew);
                    atomicAdd(&s_grad_em[3 * r.em_hit + 2], delta.z * r.T.z * r.ew);
                }
                if (r.em_nee >= 0 && r.em_nee < 32) {
                    atomicAdd(&s_grad_em[3 * r.em_nee], delta.x * (r.Tp.x * r.rho.x) * r.nk); atomicAdd(&s_grad_em[3 * r.em_nee + 1], delta.y * (r.Tp.y * r.rho.y) * r.nk);
                    atomicAdd(&s_grad_em[3 * r.em_nee + 2], delta.z * (r.Tp.z * r.rho.z) * r.nk);
                }
            }
            if (!r.has_bsdf) { a = mk3(delta.x * r.E.x, delta.y * r.E.y, delta.z * r.E.z); continue; }
            const f3 Y = mk3(delta.x * r.Nc.x + a.x, delta.y * r.Nc.y + a.y, delta.z * r.Nc.z + a.z);
            const f3 g = mk3(r.Tp.x * Y.x, r.Tp.y * Y.y, r.Tp.z * Y.z);
            const f3 b = mk3(r.rho.x * Y.x, r.rho.y * Y.y, r.rho.z * Y.z);
            if (r.texel != kNoPrim) {
                const DevTexture t = P.sv.textures[P.sv.bsdfs[r.bsdf].texture];
                if (A.grad_tex) {
                    float *gt = A.grad_tex + t.grad_offset + 3u * (size_t) r.texel;
                    const float w00 = (1.0f - r.w1.y) * (1.0f - r.w1.x), w10 = (1.0f - r.w1.y) * r.w1.x,
                                w01 = r.w1.y * (1.0f - r.w1.x), w11 = r.w1.y * r.w1.x;
                    const float gg[3] = { g.x, g.y, g.z };
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        atomicAdd(gt + ch, gg[ch] * w00); atomicAdd(gt + 3 + ch, gg[ch] * w10);
                        atomicAdd(gt + 3 * t.w + ch, gg[ch] * w01); atomicAdd(gt + 3 * t.w + 3 + ch, gg[ch] * w11);
                    }
                }
            } else if (A.grad_bsdf && r.bsdf >= 0 && r.bsdf < 32) {
                atomicAdd(&s_grad[3 * r.bsdf], g.x); atomicAdd(&s_grad[3 * r.bsdf + 1], g.y); atomicAdd(&s_grad[3 * r.bsdf + 2], g.z);
            }
            a = mk3(delta.x * r.E.x + r.invq * b.x, delta.y * r.E.y + r.invq * b.y, delta.z * r.E.z + r.invq * b.z);
            if (r.rr_channel >= 0) {
                const float corr = (r.invq * r.invq) * (b.x * r.T.x + b.y * r.T.y + b.z * r.T.z);
                if (r.rr_channel == 0) a.x -= corr; else if (r.rr_channel == 1) a.y -= corr; else a.z -= corr;
            }
        }
    }
    __syncthreads();
    if (A.grad_bsdf)
        for (uint32_t i = threadIdx.x; i < 3u * min(P.sv.n_bsdfs, 32u); i += kBlock)
            if (s_grad[i] != 0.0f) atomicAdd(A.grad_bsdf + i, s_grad[i]);
    if (A.grad_emitter)
        for (uint32_t i = threadIdx.x; i < 3u * min(P.sv.n_emitters, 32u); i += kBlock)
            if (s_grad_em[i] != 0.0f) atomicAdd(A.grad_emitter + i, s_grad_em[i]);
}

hipError_t launch_adjoint(const AdjointParams &a, hipStream_t s) {
    if (a.n_samples == 0) return hipSuccess;
    uint64_t blocks = (a.n_samples + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    if (a.rp.sv.flat) hipLaunchKernelGGL(k_adjoint<true>, dim3((uint32_t) blocks), dim3(kBlock), bounce_lds_bytes(a.rp.sv), s, a);
    else hipLaunchKernelGGL(k_adjoint<false>, dim3((uint32_t) blocks), dim3(kBlock), bounce_lds_bytes(a.rp.sv), s, a);
    return hipGetLastError();
}

// Derivative w.r.t. the texels of the `envmap` emitter ('data', envmap.cpp:214-218; docs/examples/10_inverse_rendering/invert_bunny.py):
// one thread replays one camera sample with the PCG32 stream of the primal pass through the general fused step -- any BSDF, any depth
// -- and scatters delta * d(radiance)/d(texel) at every use of the map (bounce_step, ENVGRAD).
template <bool FLAT>
__global__ __launch_bounds__(kBlock) void k_adjoint_env(const AdjointParams A) {
    extern __shared__ float4 smem[];
    const RenderParams &P = A.rp;
    const LdsView lds = lds_stage<FLAT>(P.sv, smem);
    const uint32_t spp = (uint32_t) P.spp;
    for (uint64_t k = (uint64_t) blockIdx.x * kBlock + threadIdx.x; k < A.n_samples; k += (uint64_t) gridDim.x * kBlock) {
        PathState s; float2 pos;
        generate_path(P, k, (uint32_t) (k / spp), (uint32_t) (k % spp), s, &pos);
        const EnvGradCtx eg = { adjoint_delta(A, pos), A.grad_env };
        Counters c = { 0u, 0u, 0u, 0u };
        while (bounce_step<FLAT, false, 0, true, true>(P, lds, s, c, nullptr, nullptr, &eg)) { }
    }
}

hipError_t launch_adjoint_env(const AdjointParams &a, hipStream_t s) {
    if (a.n_samples == 0) return hipSuccess;
    uint64_t blocks = (a.n_samples + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    if (a.rp.sv.flat) hipLaunchKernelGGL(k_adjoint_env<true>, dim3((uint32_t) blocks), dim3(kBlock), bounce_lds_bytes(a.rp.sv), s, a);
    else hipLaunchKernelGGL(k_adjoint_env<false>, dim3((uint32_t) blocks), dim3(kBlock), bounce_lds_bytes(a.rp.sv), s, a);
    return hipGetLastError();
}

// d(loss)/d(one scalar BSDF parameter): every camera sample is replayed through the general fused step with the dual part of
// bounce_step<PGRAD>; its contribution delta . d(radiance)/d(theta) is summed over the workgroup and added to A.grad_param[0].
template <bool FLAT>
__global__ __launch_bounds__(kBlock) void k_adjoint_param(const AdjointParams A) {
    extern __shared__ float4 smem[];
    __shared__ float s_sum[kBlock / 64];
    const RenderParams &P = A.rp;
    const LdsView lds = lds_stage<FLAT>(P.sv, smem);
    const uint32_t spp = (uint32_t) P.spp;
    float acc = 0.0f;
    for (uint64_t k = (uint64_t) blockIdx.x * kBlock + threadIdx.x; k < A.n_samples; k += (uint64_t) gridDim.x * kBlock) {
        PathState s; float2 pos;
        generate_path(P, k, (uint32_t) (k / spp), (uint32_t) (k % spp), s, &pos);
        const f3 delta = adjoint_delta(A, pos);
        ParamGradCtx pg = { A.pg_bsdf, A.pg_plus, A.pg_minus, A.pg_inv_2h, mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 0.0f) };
        Counters c = { 0u, 0u, 0u, 0u };
        while (bounce_step<FLAT, false, 0, true, false, false, true>(P, lds, s, c, nullptr, nullptr, nullptr, &pg)) { }
        const float g = fmaf(delta.z, pg.dres.z, fmaf(delta.y, pg.dres.y, delta.x * pg.dres.x));
        if (isfinite(g)) acc += g;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane_id() == 0u) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.0f;
        for (uint32_t w = 0; w < kBlock / 64; ++w) tot += s_sum[w];
        if (tot != 0.0f) atomicAdd(A.grad_param, tot);
    }
}

hipError_t launch_adjoint_param(const AdjointParams &a, hipStream_t s) {
    if (a.n_samples == 0) return hipSuccess;
    uint64_t blocks = (a.n_samples + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    if (a.rp.sv.flat) hipLaunchKernelGGL(k_adjoint_param<true>, dim3((uint32_t) blocks), dim3(kBlock), bounce_lds_bytes(a.rp.sv), s, a);
    else hipLaunchKernelGGL(k_adjoint_param<false>, dim3((uint32_t) blocks), dim3(kBlock), bounce_lds_bytes(a.rp.sv), s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// ImageBlock::put as a gather: one wave per film pixel, lanes stride over the samples of the
// (2R+1)^2 neighbouring pixels, fixed-order butterfly reduction -> bitwise reproducible film.
// weight of the sample at block-relative position `pos` for block pixel `t` along one axis
// (imageblock.cpp:117-161); `size` = block extent incl. border
MTS_DEV float axis_weight(const FilterView &f, float pos, int t, int size) {
    if (f.radius > 1.0f) {
        int lo = max((int) ceilf(pos - f.radius), 0);
        int hi = min((int) floorf(pos + f.radius), size - 1);
        int i = t - lo;
        if (i < 0 || i >= f.taps || t > hi) return 0.0f;
        float base = (float) (uint32_t) lo - pos;
        return filter_weight(f, base + (float) i);
    } else {
        int lo = (int) ceilf(pos - 0.5f);
        return (lo == t && lo >= 0 && lo < size) ? 1.0f : 0.0f;
    }
}

__global__ __launch_bounds__(kBlock) void k_film_gather(const FilmParams F) {
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t lane = lane_id();
    const uint32_t n_rows = (uint32_t) (F.row1 - F.row0);
    if (wave >= n_rows * (uint32_t) F.crop_w) return;
    const int x = (int) (wave % (uint32_t) F.crop_w), y = F.row0 + (int) (wave / (uint32_t) F.crop_w);
    const FilterView &f = F.filter;
    const int b = f.border;
    const int R = (int) ceilf(f.radius);
    const int sx = F.crop_w + 2 * b, sy = F.crop_h + 2 * b;
    // block offset = crop offset, with border: pos = pos_ - (offset - border + 0.5)
    const float offx = (float) (F.crop_x - b) + 0.5f, offy = (float) (F.crop_y - b) + 0.5f;
    const uint64_t i0 = F.first_ordinal, i1 = F.first_ordinal + F.n_samples;
    float acc[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    for (int qy = y - R; qy <= y + R; ++qy) {
        if (qy < 0 || qy >= F.crop_h) continue;
        const int lr = row_to_local(F.rows, qy);
        if (lr < 0) continue;
        for (int qx = x - R; qx <= x + R; ++qx) {
            if (qx < 0 || qx >= F.crop_w) continue;
            uint64_t q = (uint64_t) lr * (uint64_t) F.crop_w + (uint64_t) qx;
            uint64_t s_lo = q * (uint64_t) F.spp, s_hi = s_lo + (uint64_t) F.spp;
            if (s_lo < i0) s_lo = i0;
            if (s_hi > i1) s_hi = i1;
            for (uint64_t sidx = s_lo + lane; sidx < s_hi; sidx += 64u) {
                const float4 val = F.out_rgba[sidx - i0];    // (X, Y, Z, alpha), alpha < 0: invalid sample
                if (val.w < 0.0f) continue;
                const float2 pp = F.out_pos[sidx - i0];
                float wx = axis_weight(f, pp.x - offx, x + b, sx);
                if (wx == 0.0f) continue;                 // outside the footprint (or a zero tap: adds nothing)
                float wy = axis_weight(f, pp.y - offy, y + b, sy);
                float w = wy * wx;                        // box filter: exactly 1 or 0
                if (w == 0.0f) continue;
                acc[0] += val.x * w; acc[1] += val.y * w; acc[2] += val.z * w; acc[3] += val.w * w; acc[4] += 1.0f * w;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off);
    if (lane == 0) {
        float *dst = F.film + 5u * ((size_t) y * (size_t) F.crop_w + (size_t) x);
#pragma unroll
        for (int k = 0; k < 5; ++k) dst[k] += acc[k];
    }
}

// ---------------------------------------------------------------------------------------------
// ImageBlock::put, tiled by SOURCE pixels.  A workgroup owns a 16x16 block of source pixels (one thread each) of the pass and
// reads every one of their samples exactly once: 8 samples per pixel at a time, i.e. 128 contiguous bytes of radiance and 64 of
// film position -- whole sectors.  Two samples at a time it turns each sample into a record in LDS -- value (X,Y,Z,A) and the
// <= 4 filter taps per axis, exactly as imageblock.cpp:117-147 computes them, stored source-aligned (tap k <-> film pixel
// q - R + k) -- and the threads gather, branch-free and in a fixed order, the (16 + 2R)^2 film pixels the block's samples reach
// (imageblock.cpp:148-161 as a gather).  The (16 + 2R)^2 x 5 sums go to a scratch tile; k_film_merge adds the <= 4 overlapping
// tiles of every film pixel in a fixed order.  No atomics: the film is bitwise reproducible.  HBM traffic: 24 B per sample, once.
constexpr int kFilmTile = 16, kFilmTaps = 5;      // taps per axis seen from the source pixel: 2R+1 <= 5
constexpr int kFilmDest = kFilmTile + 4;          // scratch tiles are laid out for the widest supported apron (R = 2)
constexpr int kFilmPartial = kFilmDest * kFilmDest * 5;      // floats per scratch tile

// Filter weights of one sample along one axis for the 2R+1 film pixels q-R .. q+R around its source pixel q
// (block coordinates t = q + border + k - R), exactly as imageblock.cpp:117-147 evaluates them: 0 outside [lo, hi]
// and beyond the n = `taps` entries starting at lo.
MTS_DEV void axis_taps(const FilterView &f, const float *table, float pos, int size, int t0, int R, float w[kFilmTaps]) {
#pragma unroll
    for (int k = 0; k < kFilmTaps; ++k) w[k] = 0.0f;
    if (f.radius > 1.0f) {
        const int lo = max((int) ceilf(pos - f.radius), 0);
        const int hi = min((int) floorf(pos + f.radius), size - 1);
        const float base = (float) (uint32_t) lo - pos;
#pragma unroll
        for (int k = 0; k < kFilmTaps; ++k) {
            const int t = t0 + k, i = t - lo;
            if (k <= 2 * R && i >= 0 && i < f.taps && t <= hi) {
                const float xx = base + (float) i;
                w[k] = f.analytic ? filter_eval(f, xx) : table[min((int) fabsf(xx * f.scale_factor), 31)];
            }
        }
    } else {
        const int lo = (int) ceilf(pos - 0.5f);
        const int k = lo - t0;
        if (lo >= 0 && lo < size && k >= 0 && k <= 2 * R) w[k] = 1.0f;
    }
}

// The same weights for the common case -- a tabulated filter of radius exactly 2 (gaussian, mitchell, catmullrom) and a source pixel
// whose five taps t0 .. t0 + 4 all lie on the film -- without the per-tap window tests: a tap outside [lo, hi] is more than the radius
// away from the sample, (int) (|x| * 31 / 2) is then >= 31 and the table (extended with zeros to 64 entries) returns the 0 the window
// test would have set; x itself is formed exactly as above, (lo - pos) + i with i = k - (lo - t0), so the weights inside the window
// are the same bits.  (|x| < 4 and radius = 2 exactly: the product with the scale factor cannot round across 31.)  Per tap: three
// cheap VALU ops, one convert, one LDS read -- the window tests, selects and clamps above were 45 % of k_film_accum's VALU time.
MTS_DEV void axis_taps_r2(const float *table64, float scale_factor, float pos, float t0f, float w[kFilmTaps]) {
    const float lo = ceilf(pos - 2.0f), base = lo - pos, dk = lo - t0f;      // lo - t0 is 0 or 1
#pragma unroll
    for (int k = 0; k < kFilmTaps; ++k) {
        const float xx = base + ((float) k - dk);
        w[k] = table64[(int) fabsf(xx * scale_factor)];
    }
}

// One thread per source pixel.  The 5 x 5 x 5 partial sums of its own samples -- for every tap (kx, ky) of the pixel's
// neighbourhood the sum over the samples of w_y[ky] w_x[kx] (X, Y, Z, A, 1) -- stay in registers for the whole pass: the main loop
// touches neither LDS (beyond the 32-entry filter table) nor a barrier.  At the end the block exchanges the sums through LDS, one tap
// row at a time, and every film pixel of the (16 + 2R)^2 region adds up the <= 25 source pixels that reach it.
#ifndef MTS_FILM_PREFETCH
#define MTS_FILM_PREFETCH 2
#endif
#ifndef MTS_FILM_NT
#define MTS_FILM_NT 0
#endif
typedef float film_v4 __attribute__((ext_vector_type(4)));
typedef float film_v2 __attribute__((ext_vector_type(2)));
MTS_DEV float4 film_load(const float4 *p) {
#if MTS_FILM_NT
    const film_v4 v = __builtin_nontemporal_load(reinterpret_cast<const film_v4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
MTS_DEV float2 film_load(const float2 *p) {
#if MTS_FILM_NT
    const film_v2 v = __builtin_nontemporal_load(reinterpret_cast<const film_v2 *>(p));
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}
#define MTS_FILM_LOAD(p) film_load(p)
constexpr int kFilmPrefetch = MTS_FILM_PREFETCH;     // samples in flight per thread (x 24 B)
// DMA (round 3): the stream is staged through LDS by wave-cooperative loads.  Per-lane loads of the pixel-major stream touch one 128-byte
// line per lane and use 16 bytes of it at a time; the rest has to survive in a 32 KB L1 that 512 lanes share until the lane comes back
// for it, and the bytes a thread can keep in flight are bounded by its registers (125 accumulators leave room for two samples).  Now
// eight adjacent lanes fetch the 8 x 16 bytes of one pixel's line (rgba: 8 samples; positions: 4 lanes x 16 bytes) with
// global_load_lds_dwordx4 -- the data lands in LDS without passing through registers, every line is requested once, by one instruction,
// and used whole -- and after a barrier every thread consumes the 8 samples of its own pixel from LDS (rotated start: conflict-free
// ds_read_b128).  Taken when the runs are whole groups of 8 samples; the per-lane loop stays for the rest.
#ifndef MTS_FILM_DMA
#define MTS_FILM_DMA 1
#endif
constexpr int kFilmRound = 8;                        // samples of a pixel per staging round: one 128-byte line of rgba
typedef __attribute__((address_space(3))) void film_lds_void;
typedef __attribute__((address_space(1))) const void film_glb_void;
template <bool DMA>
__global__ __launch_bounds__(kBlock) void k_film_accum(const FilmParams F) {
    __shared__ float table[64];                          // the filter's 32 entries + zeros (axis_taps_r2)
    // staging: rgba [pixel][8] float4 (32 KB) + positions [pixel][8] float2 (16 KB); the exchange buffer E of the epilogue aliases it
    __shared__ float4 stage[DMA ? kBlock * kFilmRound * 3 / 2 : (kFilmTile * kFilmTile * (kFilmTaps * 5 + 1) + 3) / 4];
    float *E = reinterpret_cast<float *>(stage);         // one tap row of every source pixel: [pixel][kx][channel], padded
    static_assert(kBlock * kFilmRound * 3 / 2 * 16 >= kFilmTile * kFilmTile * (kFilmTaps * 5 + 1) * 4, "the exchange buffer fits the staging area");
    const FilterView &f = F.filter;
    const int b = f.border, R = (int) ceilf(f.radius), DW = kFilmTile + 2 * R;
    if (threadIdx.x < 64) table[threadIdx.x] = threadIdx.x < 32 ? f.table[threadIdx.x] : 0.0f;
    const int tcx = (int) (blockIdx.x % (uint32_t) F.tiles_x), tcy = (int) (blockIdx.x / (uint32_t) F.tiles_x);
    const int sp = (int) threadIdx.x, lx = sp % kFilmTile, ly = sp / kFilmTile;
    // this thread's source pixel: column qx, local row lr (the pass holds whole local rows; a tile's rows are contiguous on the film)
    const int qx = tcx * kFilmTile + lx, lr = F.pass_lr0 + tcy * F.tile_h + ly;
    const bool have = qx < F.crop_w && ly < F.tile_h && lr < F.pass_lr0 + F.pass_rows;
    const int qy = have ? row_to_global(F.rows, lr) : 0;
    const int sx = F.crop_w + 2 * b, sy = F.crop_h + 2 * b;
    const float offx = (float) (F.crop_x - b) + 0.5f, offy = (float) (F.crop_y - b) + 0.5f;
    const int tap_x0 = qx + b - R, tap_y0 = qy + b - R;
    // radius-2 table filter, every tap of this pixel on the film: the short form of the weights (axis_taps_r2)
    const bool fast_taps = !f.analytic && f.radius == 2.0f && f.table[31] == 0.0f && tap_x0 >= 0 && tap_x0 + 4 <= sx - 1 && tap_y0 >= 0 && tap_y0 + 4 <= sy - 1;
    const float tap_x0f = (float) tap_x0, tap_y0f = (float) tap_y0;
    const size_t slot0 = have ? (size_t) (((uint64_t) lr * (uint64_t) F.crop_w + (uint64_t) qx) * (uint64_t) F.spp - F.first_ordinal) : 0;
    __syncthreads();
    float acc[kFilmTaps][kFilmTaps][5];
#pragma unroll
    for (int ky = 0; ky < kFilmTaps; ++ky)
#pragma unroll
        for (int kx = 0; kx < kFilmTaps; ++kx)
#pragma unroll
            for (int c = 0; c < 5; ++c) acc[ky][kx][c] = 0.0f;
    float4 pv[kFilmPrefetch]; float2 pq[kFilmPrefetch];
    // blockIdx.y: which run of the pixel's samples (runs start on multiples of 8: whole 128-byte lines)
    const int run = ((F.spp + F.slices - 1) / F.slices + 7) & ~7;
    const int s_begin = min((int) blockIdx.y * run, F.spp), n_spp = have ? min(s_begin + run, F.spp) : s_begin;
    auto accumulate = [&](const float4 rec, const float2 rp) {
        if (!(rec.w >= 0.0f)) return;                        // (X, Y, Z, alpha); alpha < 0: invalid or absent sample
        float wxs[kFilmTaps], wys[kFilmTaps];
        if (fast_taps) {
            axis_taps_r2(table, f.scale_factor, rp.x - offx, tap_x0f, wxs);
            axis_taps_r2(table, f.scale_factor, rp.y - offy, tap_y0f, wys);
        } else {
            axis_taps(f, table, rp.x - offx, sx, tap_x0, R, wxs);
            axis_taps(f, table, rp.y - offy, sy, tap_y0, R, wys);
        }
        // a zero weight adds (signed) zeros, which leaves the sums as they are: the film equals the one of a loop over the non-zero
        // taps only (imageblock.cpp:148-161)
#pragma unroll
        for (int ky = 0; ky < kFilmTaps; ++ky)
#pragma unroll
            for (int kx = 0; kx < kFilmTaps; ++kx) {
                const float w = wys[ky] * wxs[kx];
                acc[ky][kx][0] += rec.x * w; acc[ky][kx][1] += rec.y * w; acc[ky][kx][2] += rec.z * w; acc[ky][kx][3] += rec.w * w;
                acc[ky][kx][4] += 1.0f * w;
            }
    };
    if (DMA) {
        // (host: DMA only if F.spp % 8 == 0, so every run is a whole number of rounds and the stream offsets are 16-byte aligned)
        float4 *s_rgba = stage;
        float2 *s_pos = reinterpret_cast<float2 *>(stage + kBlock * kFilmRound);
        const int wv = sp >> 6, ln = sp & 63;
        const int n_end = min(s_begin + run, F.spp);         // workgroup-uniform
        const int last_lr = F.pass_lr0 + F.pass_rows;
        auto slot_of = [&](int q, bool &ok) -> size_t {      // first stream slot of tile pixel q (any lane computes any pixel's)
            const int qlx = q % kFilmTile, qly = q / kFilmTile;
            const int qqx = tcx * kFilmTile + qlx, qlr = F.pass_lr0 + tcy * F.tile_h + qly;
            ok = qqx < F.crop_w && qly < F.tile_h && qlr < last_lr;
            return ok ? (size_t) (((uint64_t) qlr * (uint64_t) F.crop_w + (uint64_t) qqx) * (uint64_t) F.spp - F.first_ordinal) : 0;
        };
        for (int s0 = s_begin; s0 < n_end; s0 += kFilmRound) {
            // rgba: 8 instructions per wave, lanes 8 g .. 8 g + 7 fetch the line of pixel 64 wv + 8 i + g
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int q = 64 * wv + 8 * i + (ln >> 3);
                bool ok; const size_t q0 = slot_of(q, ok);
                if (ok) __builtin_amdgcn_global_load_lds((film_glb_void *) (F.out_rgba + q0 + (size_t) (s0 + (ln & 7))),
                                                         (film_lds_void *) (s_rgba + (64 * wv + 8 * i) * kFilmRound), 16, 0, 0);
            }
            // positions: 4 instructions per wave, lanes 4 g .. 4 g + 3 fetch the 64 bytes (8 x float2) of pixel 64 wv + 16 i + g
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 64 * wv + 16 * i + (ln >> 2);
                bool ok; const size_t q0 = slot_of(q, ok);
                if (ok) __builtin_amdgcn_global_load_lds((film_glb_void *) (F.out_pos + q0 + (size_t) (s0 + 2 * (ln & 3))),
                                                         (film_lds_void *) (s_pos + (64 * wv + 16 * i) * kFilmRound), 16, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0);                   // vmcnt(0): the wave's DMA has landed
            __syncthreads();
            if (have) {
#pragma unroll 2
                for (int kk = 0; kk < kFilmRound; ++kk) {
                    const int j = (kk + (sp >> 1)) & (kFilmRound - 1);      // rotated start: the 16 lanes of a ds_read_b128 group hit 16 different banks
                    accumulate(s_rgba[sp * kFilmRound + j], s_pos[sp * kFilmRound + j]);
                }
            }
            __syncthreads();
        }
    } else {
#pragma unroll
    for (int k = 0; k < kFilmPrefetch; ++k) {
        pv[k] = make_float4(0.0f, 0.0f, 0.0f, -1.0f); pq[k] = make_float2(0.0f, 0.0f);
        if (s_begin + k < n_spp) { pv[k] = MTS_FILM_LOAD(F.out_rgba + slot0 + (size_t) (s_begin + k)); pq[k] = MTS_FILM_LOAD(F.out_pos + slot0 + (size_t) (s_begin + k)); }
    }
    for (int s0 = s_begin; s0 < n_spp; s0 += kFilmPrefetch) {
#pragma unroll
        for (int k = 0; k < kFilmPrefetch; ++k) {
            const float4 rec = pv[k]; const float2 rp = pq[k];
            {   // the register is free: fetch the sample that will be processed kFilmPrefetch samples from now
                const int j = s0 + kFilmPrefetch + k;
                pv[k] = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
                if (j < n_spp) { pv[k] = MTS_FILM_LOAD(F.out_rgba + slot0 + (size_t) j); pq[k] = MTS_FILM_LOAD(F.out_pos + slot0 + (size_t) j); }
            }
            accumulate(rec, rp);
        }
    }
    }
    // ---- exchange: film pixel (dx, dy) of the region <- tap (dx - sxl, dy - syl) of the source pixel at block position (sxl, syl),
    // in ascending (ky, kx) order
    constexpr int kStride = kFilmTaps * 5 + 1;
    const int n_dest = DW * DW;
    float out0[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f }, out1[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    const int d0 = (int) threadIdx.x, d1 = (int) threadIdx.x + kBlock;
    auto collect = [&](int d, int ky, float (&out)[5]) {
        const int dy = d / DW, dx = d - dy * DW, syl = dy - ky;
        if (syl < 0 || syl >= kFilmTile) return;
#pragma unroll
        for (int kx = 0; kx < kFilmTaps; ++kx) {
            const int sxl = dx - kx;
            if (kx > 2 * R || sxl < 0 || sxl >= kFilmTile) continue;
            const float *e = E + (syl * kFilmTile + sxl) * kStride + kx * 5;
#pragma unroll
            for (int c = 0; c < 5; ++c) out[c] += e[c];
        }
    };
#pragma unroll
    for (int ky = 0; ky < kFilmTaps; ++ky) {
        if (ky > 2 * R) break;
        __syncthreads();
#pragma unroll
        for (int kx = 0; kx < kFilmTaps; ++kx)
#pragma unroll
            for (int c = 0; c < 5; ++c) E[sp * kStride + kx * 5 + c] = acc[ky][kx][c];
        __syncthreads();
        collect(d0, ky, out0);
        if (d1 < n_dest) collect(d1, ky, out1);
    }
    float *dst = F.partials + ((size_t) blockIdx.y * gridDim.x + blockIdx.x) * kFilmPartial;
    if (d0 < n_dest)
#pragma unroll
        for (int k = 0; k < 5; ++k) dst[5 * d0 + k] = out0[k];
    if (d1 < n_dest)
#pragma unroll
        for (int k = 0; k < 5; ++k) dst[5 * d1 + k] = out1[k];
}

// film[y][x] += the scratch tiles that reach (x, y), in ascending (tile row, tile column) order
__global__ __launch_bounds__(kBlock) void k_film_merge(const FilmParams F) {
    const int R = (int) ceilf(F.filter.radius), DW = kFilmTile + 2 * R;
    const uint32_t n_rows = (uint32_t) (F.row1 - F.row0);
    const uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x;
    if (i >= (uint64_t) n_rows * (uint64_t) F.crop_w) return;
    const int x = (int) (i % (uint64_t) F.crop_w), y = F.row0 + (int) (i / (uint64_t) F.crop_w);
    const int tc0 = max(x - R, 0) / kFilmTile, tc1 = min(x + R, F.crop_w - 1) / kFilmTile;
    float acc[5] = { 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    int prev = -1;
    for (int gy = max(y - R, 0); gy <= min(y + R, F.crop_h - 1); ++gy) {
        const int lr = row_to_local(F.rows, gy);
        if (lr < F.pass_lr0 || lr >= F.pass_lr0 + F.pass_rows) continue;
        const int tr = (lr - F.pass_lr0) / F.tile_h;
        if (tr == prev) continue;
        prev = tr;
        const int g0 = row_to_global(F.rows, F.pass_lr0 + tr * F.tile_h);       // film row of the tile's first source row
        const int dy = y - (g0 - R);
        for (int tc = tc0; tc <= tc1; ++tc) {
            const int dx = x - (tc * kFilmTile - R);
            if (dx < 0 || dx >= DW || dy < 0 || dy >= DW) continue;
            for (int sl = 0; sl < F.slices; ++sl) {
                const float *p = F.partials + ((size_t) sl * (size_t) (F.tiles_x * F.tiles_y) + (size_t) (tr * F.tiles_x + tc)) * kFilmPartial + 5 * (dy * DW + dx);
#pragma unroll
                for (int k = 0; k < 5; ++k) acc[k] += p[k];
            }
        }
    }
    if (prev < 0) return;                  // a film row between two of this rank's row tiles: nothing of this pass reaches it
    float *dst = F.film + 5u * ((size_t) y * (size_t) F.crop_w + (size_t) x);
#pragma unroll
    for (int k = 0; k < 5; ++k) dst[k] += acc[k];
}

bool film_tiles_supported(const FilterView &f) { return f.taps <= 4 && (int) ceilf(f.radius) <= 2; }
size_t film_partial_floats(const FilmParams &p) { return (size_t) p.slices * (size_t) p.tiles_x * (size_t) p.tiles_y * kFilmPartial; }
void film_tile_grid(FilmParams &p) {
    p.tiles_x = (p.crop_w + kFilmTile - 1) / kFilmTile;
    p.tiles_y = (p.pass_rows + p.tile_h - 1) / p.tile_h;
    // few tiles with many samples each (a rank's share of a partitioned film, a pass of a large film): cut the samples of a pixel
    // into runs of >= 64 so that about 2048 workgroups share the stream
    const int tiles = std::max(p.tiles_x * p.tiles_y, 1);
    p.slices = std::max(1, std::min({ 2048 / tiles, p.spp / 64, 64 }));
}

hipError_t launch_film_tiles(const FilmParams &p, hipStream_t s) {
    if (p.row1 <= p.row0 || p.pass_rows <= 0) return hipSuccess;
    // LDS-staged loads when every run of a pixel's samples is a whole number of 8-sample rounds (and the stream offsets 16-byte aligned)
    if (MTS_FILM_DMA && p.spp % kFilmRound == 0) hipLaunchKernelGGL(k_film_accum<true>, dim3((uint32_t) (p.tiles_x * p.tiles_y), (uint32_t) p.slices), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL(k_film_accum<false>, dim3((uint32_t) (p.tiles_x * p.tiles_y), (uint32_t) p.slices), dim3(kBlock), 0, s, p);
    const uint64_t n = (uint64_t) (p.row1 - p.row0) * (uint64_t) p.crop_w;
    hipLaunchKernelGGL(k_film_merge, dim3((uint32_t) ((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_film_gather(const FilmParams &p, hipStream_t s) {
    uint64_t waves = (uint64_t) (p.row1 - p.row0) * (uint64_t) p.crop_w;
    if (waves == 0) return hipSuccess;
    uint64_t blocks = (waves * 64u + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_film_gather, dim3((uint32_t) blocks), dim3(kBlock), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Scene::ray_intersect / ray_intersect_naive / ray_test on SoA ray streams
template <int MODE, bool FLAT>
__global__ __launch_bounds__(kBlock) void k_ray_intersect(const SceneView sv, uint64_t n, const RayStreams r,
                                                          float *t, uint32_t *prim, uint32_t *shape, float *u,
                                                          float *v, float *si26) {
    extern __shared__ float4 smem[];
    const LdsView lds = lds_stage<FLAT>(sv, smem);
    const Geo<FLAT> geo{ sv, lds };
    for (uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t) gridDim.x * kBlock) {
        bool active = r.active ? r.active[i] != 0 : true;
        Hit hit; bool found = false;
        f3 o = mk3(r.ox[i], r.oy[i], r.oz[i]), d = mk3(r.dx[i], r.dy[i], r.dz[i]);
        if (active) {
            uint32_t tt = 0;
            if (MODE == 0) found = traverse<FLAT, false>(sv, lds, o, d, r.mint[i], r.maxt[i], hit, tt);
            else found = traverse_naive<false>(sv, o, d, r.mint[i], r.maxt[i], hit);
        }
        t[i] = found ? hit.t : __builtin_inff();
        prim[i] = found ? hit.prim : kNoPrim;
        if (shape) shape[i] = found ? geo.prim_shape(hit.prim) : kNoPrim;
        if (u) u[i] = found ? hit.u : 0.0f;
        if (v) v[i] = found ? hit.v : 0.0f;
        if (si26) {
            float o26[26];
#pragma unroll
            for (int k = 0; k < 26; ++k) o26[k] = 0.0f;
            if (found) {
                SurfaceInteraction si;
                fill_si(geo, d, hit.prim, hit.u, hit.v, si);
                o26[0] = si.p.x; o26[1] = si.p.y; o26[2] = si.p.z; o26[3] = si.n.x; o26[4] = si.n.y; o26[5] = si.n.z;
                o26[6] = si.uv.x; o26[7] = si.uv.y;
                o26[8] = si.sh.s.x; o26[9] = si.sh.s.y; o26[10] = si.sh.s.z;
                o26[11] = si.sh.t.x; o26[12] = si.sh.t.y; o26[13] = si.sh.t.z;
                o26[14] = si.sh.n.x; o26[15] = si.sh.n.y; o26[16] = si.sh.n.z;
                o26[17] = si.dp_du.x; o26[18] = si.dp_du.y; o26[19] = si.dp_du.z;
                o26[20] = si.dp_dv.x; o26[21] = si.dp_dv.y; o26[22] = si.dp_dv.z;
                o26[23] = si.wi.x; o26[24] = si.wi.y; o26[25] = si.wi.z;
            } else {
                o26[23] = -d.x; o26[24] = -d.y; o26[25] = -d.z;   // scene_native.inl:28-31
            }
#pragma unroll
            for (int k = 0; k < 26; ++k) si26[(size_t) k * n + i] = o26[k];
        }
    }
}

template <bool FLAT>
__global__ __launch_bounds__(kBlock) void k_ray_test(const SceneView sv, uint64_t n, const RayStreams r, uint8_t *hit_out) {
    extern __shared__ float4 smem[];
    const LdsView lds = lds_stage<FLAT>(sv, smem);
    for (uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t) gridDim.x * kBlock) {
        bool active = r.active ? r.active[i] != 0 : true;
        bool found = false;
        if (active) {
            Hit hit; uint32_t tt = 0;
            found = traverse<FLAT, true>(sv, lds, mk3(r.ox[i], r.oy[i], r.oz[i]), mk3(r.dx[i], r.dy[i], r.dz[i]), r.mint[i],
                                   r.maxt[i], hit, tt);
        }
        hit_out[i] = found ? 1 : 0;
    }
}

static uint32_t stream_grid(uint64_t n) {
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    return (uint32_t) (blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks));
}

// Hierarchy scenes, plain hit records (no full SurfaceInteraction): every workgroup owns a contiguous chunk of the stream and its
// lanes fetch the next ray of the chunk as soon as their walk ends (dynamic ray fetch, as k_trace does for the path pool).
constexpr uint32_t kRayChunk = 16u * kBlock;
template <bool ANY>
__global__ __launch_bounds__(kBlock)
#if MTS_TRACE_WAVES > 0
__attribute__((amdgpu_waves_per_eu(MTS_TRACE_WAVES, MTS_TRACE_WAVES)))
#endif
void k_ray_walk(const SceneView sv, uint64_t n, const RayStreams r, float *t, uint32_t *prim,
                                                     uint32_t *shape, float *u, float *v, uint8_t *hit) {
    extern __shared__ float4 smem[];
    __shared__ uint32_t s_next;
    const uint32_t lane = lane_id();
    // the first sv.walk_lds_depth stack entries of a lane live in LDS, deeper ones in this workgroup's slice of sv.walk_spill
    const uint32_t spill_depth = sv.stack_depth > sv.walk_lds_depth ? sv.stack_depth - sv.walk_lds_depth : 0u;
    const WalkStack st = { reinterpret_cast<StackEntry *>(smem) + threadIdx.x, log2_stride(kBlock), sv.walk_lds_depth,
                           sv.walk_spill + (size_t) blockIdx.x * spill_depth * kBlock + threadIdx.x, kBlock };
    const LdsView lds = {};
    const Geo<false> geo{ sv, lds };
    uint32_t tri_tests = 0;
    const uint64_t n_chunks = (n + kRayChunk - 1) / kRayChunk;
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {      // persistent workgroups: a bounded spill area
    __syncthreads();
    if (threadIdx.x == 0) s_next = 0u;
    __syncthreads();
    const uint64_t base = chunk * kRayChunk;
    const uint32_t total = (uint32_t) min((uint64_t) kRayChunk, n - base);
    BvhWalk w;
    w.cur = kNoNode; w.sp = 0u; w.found = false;
    bool busy = false, exhausted = false;
    uint64_t i = 0;
    auto retire = [&](uint64_t k, bool found) {
        if (ANY) { hit[k] = found ? 1 : 0; return; }
        t[k] = found ? w.best : __builtin_inff();
        prim[k] = found ? w.best_prim : kNoPrim;
        if (shape) shape[k] = found ? geo.prim_shape(w.best_prim) : kNoPrim;
        if (u) u[k] = found ? w.hit.u : 0.0f;
        if (v) v[k] = found ? w.hit.v : 0.0f;
    };
    while (true) {
        const bool need = w.cur == kNoNode;
        if (need && busy) { retire(i, w.found); busy = false; }
        const uint64_t m = __ballot(need);
        if (m && !exhausted) {
            const uint32_t first = (uint32_t) __ffsll((long long) m) - 1u, want = (uint32_t) __popcll(m);
            uint32_t b0 = 0u;
            if (lane == first) b0 = atomicAdd(&s_next, want);
            b0 = __shfl(b0, (int) first);
            exhausted = b0 + want >= total;
            if (need) {
                const uint32_t idx = b0 + mask_rank(m);
                if (idx < total) {
                    i = base + idx;
                    if (r.active && r.active[i] == 0) {
                        w.found = false; retire(i, false);            // inactive lanes: t = inf, no shape (optix_rt.cu:35-37)
                    } else {
                        walk_begin(w, sv, mk3(r.ox[i], r.oy[i], r.oz[i]), mk3(r.dx[i], r.dy[i], r.dz[i]), r.mint[i], r.maxt[i]);
                        busy = true;
                    }
                }
            }
        }
        if (__ballot(w.cur != kNoNode) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (__ballot(w.cur != kNoNode && w.far) != 0ull) {
            if (w.cur != kNoNode) walk_round<ANY, true>(w, sv, st, tri_tests);
        } else {
            if (w.cur != kNoNode) walk_round<ANY, false>(w, sv, st, tri_tests);
        }
    }
    }
}
static uint32_t walk_grid(const SceneView &sv, uint64_t n) {
    return (uint32_t) std::min<uint64_t>((n + kRayChunk - 1) / kRayChunk, sv.walk_blocks);
}
static size_t walk_lds(const SceneView &sv) { return sizeof(StackEntry) * (std::min(sv.stack_depth, sv.walk_lds_depth) + 1u) * kBlock; }

hipError_t launch_ray_intersect(const SceneView &sv, uint64_t n, const RayStreams &r, int mode, float *t,
                                uint32_t *prim, uint32_t *shape, float *u, float *v, float *si26, hipStream_t s) {
    if (n == 0) return hipSuccess;
    size_t lds = bounce_lds_bytes(sv);
    if (mode == 0 && !sv.flat && !si26) {
        hipLaunchKernelGGL((k_ray_walk<false>), dim3(walk_grid(sv, n)), dim3(kBlock), walk_lds(sv), s,
                           sv, n, r, t, prim, shape, u, v, (uint8_t *) nullptr);
        return hipGetLastError();
    }
    if (mode == 0 && sv.flat)
        hipLaunchKernelGGL((k_ray_intersect<0, true>), dim3(stream_grid(n)), dim3(kBlock), lds, s, sv, n, r, t, prim, shape, u, v, si26);
    else if (mode == 0)
        hipLaunchKernelGGL((k_ray_intersect<0, false>), dim3(stream_grid(n)), dim3(kBlock), lds, s, sv, n, r, t, prim, shape, u, v, si26);
    else if (sv.flat)
        hipLaunchKernelGGL((k_ray_intersect<1, true>), dim3(stream_grid(n)), dim3(kBlock), lds, s, sv, n, r, t, prim, shape, u, v, si26);
    else
        hipLaunchKernelGGL((k_ray_intersect<1, false>), dim3(stream_grid(n)), dim3(kBlock), lds, s, sv, n, r, t, prim, shape, u, v, si26);
    return hipGetLastError();
}

hipError_t launch_ray_test(const SceneView &sv, uint64_t n, const RayStreams &r, uint8_t *hit, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (sv.flat) hipLaunchKernelGGL(k_ray_test<true>, dim3(stream_grid(n)), dim3(kBlock), bounce_lds_bytes(sv), s, sv, n, r, hit);
    else hipLaunchKernelGGL((k_ray_walk<true>), dim3(walk_grid(sv, n)), dim3(kBlock), walk_lds(sv), s,
                            sv, n, r, (float *) nullptr, (uint32_t *) nullptr, (uint32_t *) nullptr, (float *) nullptr, (float *) nullptr, hit);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_camera_rays(const CameraView cam, uint64_t n, const float *sx, const float *sy,
                                                        const float *apx, const float *apy, float *ox, float *oy, float *oz, float *dx, float *dy, float *dz,
                                                        float *mint, float *maxt) {
    for (uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t) gridDim.x * kBlock) {
        f3 o, d; float t0, t1;
        f2 ap; ap.x = apx ? apx[i] : 0.5f; ap.y = apy ? apy[i] : 0.5f;
        camera_ray(cam, sx[i], sy[i], ap, o, d, t0, t1);
        ox[i] = o.x; oy[i] = o.y; oz[i] = o.z; dx[i] = d.x; dy[i] = d.y; dz[i] = d.z; mint[i] = t0; maxt[i] = t1;
    }
}
hipError_t launch_camera_rays(const CameraView &cam, uint64_t n, const float *sx, const float *sy, const float *apx, const float *apy, float *ox, float *oy,
                              float *oz, float *dx, float *dy, float *dz, float *mint, float *maxt, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_camera_rays, dim3(stream_grid(n)), dim3(kBlock), 0, s, cam, n, sx, sy, apx, apy, ox, oy, oz, dx, dy, dz, mint, maxt);
    return hipGetLastError();
}

// ImageBlock::put(pos, value) as a scatter with float atomics (the reference's scatter_add)
__global__ __launch_bounds__(kBlock) void k_imageblock_put(const FilterView f, int32_t w, int32_t h, int32_t ox, int32_t oy,
                                                           int32_t ch, int32_t border, uint64_t n, const float *pos,
                                                           const float *values, float *data) {
    const int sx = w + 2 * border, sy = h + 2 * border;
    for (uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t) gridDim.x * kBlock) {
        const float *val = values + (size_t) ch * i;
        bool valid = true;
        for (int k = 0; k < ch; ++k) valid = valid && (val[k] >= -1e-5f) && isfinite(val[k]);
        if (!valid) continue;
        float px = pos[2 * i] - ((float) (ox - border) + 0.5f), py = pos[2 * i + 1] - ((float) (oy - border) + 0.5f);
        if (f.radius > 1.0f) {
            int lox = max((int) ceilf(px - f.radius), 0), loy = max((int) ceilf(py - f.radius), 0);
            int hix = min((int) floorf(px + f.radius), sx - 1), hiy = min((int) floorf(py + f.radius), sy - 1);
            float bx = (float) (uint32_t) lox - px, by = (float) (uint32_t) loy - py;
            for (int yr = 0; yr < f.taps; ++yr) {
                int y = loy + yr;
                if (y > hiy) break;
                float wy = filter_weight(f, by + (float) yr);
                for (int xr = 0; xr < f.taps; ++xr) {
                    int x = lox + xr;
                    if (x > hix) break;
                    float wgt = wy * filter_weight(f, bx + (float) xr);
                    float *dst = data + (size_t) ch * ((size_t) y * sx + x);
                    for (int k = 0; k < ch; ++k) atomicAdd(dst + k, val[k] * wgt);
                }
            }
        } else {
            int lox = (int) ceilf(px - 0.5f), loy = (int) ceilf(py - 0.5f);
            if (lox >= 0 && loy >= 0 && lox < sx && loy < sy) {
                float *dst = data + (size_t) ch * ((size_t) loy * sx + lox);
                for (int k = 0; k < ch; ++k) atomicAdd(dst + k, val[k]);
            }
        }
    }
}
hipError_t launch_imageblock_put(const FilterView &f, int32_t w, int32_t h, int32_t ox, int32_t oy, int32_t ch,
                                 int32_t border, uint64_t n, const float *pos, const float *values, float *data,
                                 hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_imageblock_put, dim3(stream_grid(n)), dim3(kBlock), 0, s, f, w, h, ox, oy, ch, border, n, pos, values, data);
    return hipGetLastError();
}

// ImageBlock::put(block): clipped rectangular += (accumulate_2d, bitmap.h:657-716)
__global__ __launch_bounds__(kBlock) void k_put_block(const float *src, int ssx, int sox, int soy, float *dst, int tsx, int tox,
                                                      int toy, int szx, int szy, int ch) {
    uint64_t total = (uint64_t) szx * szy * ch;
    for (uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t) gridDim.x * kBlock) {
        int row_elems = szx * ch;
        int y = (int) (i / row_elems), c = (int) (i % row_elems);
        dst[((size_t) (toy + y) * tsx + tox) * ch + c] += src[((size_t) (soy + y) * ssx + sox) * ch + c];
    }
}
hipError_t launch_put_block(const float *src, int32_t sw, int32_t sh, int32_t sox_, int32_t soy_, int32_t sb, float *dst,
                            int32_t dw, int32_t dh, int32_t dox, int32_t doy, int32_t db, int32_t ch, hipStream_t s) {
    int ssx = sw + 2 * sb, ssy = sh + 2 * sb, tsx = dw + 2 * db, tsy = dh + 2 * db;
    int sox = 0, soy = 0, tox = (sox_ - sb) - (dox - db), toy = (soy_ - sb) - (doy - db);
    int szx = ssx, szy = ssy;
    int shx = std::max(0, std::max(-sox, -tox)), shy = std::max(0, std::max(-soy, -toy));
    sox += shx; tox += shx; soy += shy; toy += shy;
    szx -= std::max(sox + szx - ssx, 0); szx -= std::max(tox + szx - tsx, 0);
    szy -= std::max(soy + szy - ssy, 0); szy -= std::max(toy + szy - tsy, 0);
    if (szx <= 0 || szy <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_put_block, dim3(stream_grid((uint64_t) szx * szy * ch)), dim3(kBlock), 0, s, src, ssx, sox, soy, dst,
                       tsx, tox, toy, szx, szy, ch);
    return hipGetLastError();
}

// HDRFilm::bitmap: (X,Y,Z,A) / W, RGB = M * XYZ (hdrfilm.cpp:278-299, struct.cpp:1761-1811)
// RoughPlastic::parameters_changed (roughplastic.cpp:380-399): one thread per table entry
__global__ __launch_bounds__(kRoughTableRes) void k_roughplastic_tables(DevBsdf *bsdfs, uint32_t index, float *table, const float *gl,
                                                                      int res_t, int res_r) {
    __shared__ float part[kRoughTableRes];
    const DevBsdf b = bsdfs[index];
    const Mdf d = mdf_make((b.flags & kBsdfGGX) != 0u, b.alpha_u, b.alpha_u, true);
    const int i = (int) threadIdx.x;
    const float mu = fmaxf(1e-6f, (float) i / (float) (kRoughTableRes - 1));
    const f3 wi = mk3(sqrtf(1.0f - mu * mu), 0.0f, mu);
    const float eta = b.er, inv_eta = 1.0f / eta;
    table[i] = rough_transmittance(d, wi, eta, res_t, gl, gl + 128);
    part[i] = rough_reflectance(d, wi, inv_eta, res_r, gl + 256, gl + 384) * wi.z;
    __syncthreads();
    if (i == 0) {
        float sum = 0.0f;
        for (int k = 0; k < kRoughTableRes; ++k) sum += part[k];
        bsdfs[index].eb = (sum * (1.0f / (float) kRoughTableRes)) * 2.0f;
        bsdfs[index].table = table;
    }
}
hipError_t launch_roughplastic_tables(DevBsdf *bsdfs, uint32_t index, float *table, const float *gl, int res_t, int res_r, hipStream_t s) {
    hipLaunchKernelGGL(k_roughplastic_tables, dim3(1), dim3(kRoughTableRes), 0, s, bsdfs, index, table, gl, res_t, res_r);
    return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void k_square_stream(float4 *rgba, uint64_t n) {
    const uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float4 v = rgba[i];
    v.x *= v.x; v.y *= v.y; v.z *= v.z;            // m2 AOVs: sqr of the nested integrator's XYZ (moment.cpp:91-93)
    rgba[i] = v;
}
__global__ __launch_bounds__(kBlock) void k_moment_pack(const float *a, const float *b, float *out, uint64_t n) {
    const uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *pa = a + 5u * i, *pb = b + 5u * i;
    float *o = out + 11u * i;
    o[0] += pa[0]; o[1] += pa[1]; o[2] += pa[2]; o[3] += pa[3]; o[4] += pa[4];
    o[5] += pa[0]; o[6] += pa[1]; o[7] += pa[2];
    o[8] += pb[0]; o[9] += pb[1]; o[10] += pb[2];
}
hipError_t launch_square_stream(float4 *rgba, uint64_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_square_stream, dim3((uint32_t) ((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, rgba, n);
    return hipGetLastError();
}
hipError_t launch_moment_pack(const float *values5, const float *squares5, float *film11, uint64_t n_pixels, hipStream_t s) {
    if (n_pixels) hipLaunchKernelGGL(k_moment_pack, dim3((uint32_t) ((n_pixels + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, values5, squares5, film11, n_pixels);
    return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void k_film_develop(const float *xyzaw, uint64_t n, float *rgba) {
    for (uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t) gridDim.x * kBlock) {
        const float *p = xyzaw + 5 * i;
        float inv_w = 1.0f / p[4];
        float r = 0.0f, g = 0.0f, b = 0.0f;
        r += 3.240479f * p[0]; r += -1.537150f * p[1]; r += -0.498535f * p[2];
        g += -0.969256f * p[0]; g += 1.875991f * p[1]; g += 0.041556f * p[2];
        b += 0.055648f * p[0]; b += -0.204043f * p[1]; b += 1.057311f * p[2];
        rgba[4 * i] = r * inv_w; rgba[4 * i + 1] = g * inv_w; rgba[4 * i + 2] = b * inv_w; rgba[4 * i + 3] = p[3] * inv_w;
    }
}
hipError_t launch_film_develop(const float *xyzaw, uint64_t n, float *rgba, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_film_develop, dim3(stream_grid(n)), dim3(kBlock), 0, s, xyzaw, n, rgba);
    return hipGetLastError();
}

// device_libm.h on argument streams (mtsamd_libm_eval: host / device bit-equality tests)
__global__ __launch_bounds__(kBlock) void k_libm_eval(int fn, uint64_t n, const float *x, const float *y, float *out) {
    const uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float a = x[i];
    float r;
    switch (fn) {
    case 0: r = lm_sin(a); break;
    case 1: r = lm_cos(a); break;
    case 2: r = lm_tan(a); break;
    case 3: r = lm_exp(a); break;
    case 4: r = lm_log(a); break;
    case 5: r = lm_erf(a); break;
    case 6: r = lm_acos(a); break;
    case 8: r = lm_atanh(a); break;
    case 9: r = lm_cosh(a); break;
    default: r = lm_atan2(a, y[i]); break;
    }
    out[i] = r;
}
hipError_t launch_libm_eval(int fn, uint64_t n, const float *x, const float *y, float *out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_libm_eval, dim3((uint32_t) ((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, fn, n, x, y, out);
    return hipGetLastError();
}

} // namespace mtsamd
