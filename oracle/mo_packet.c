/* ORACLE -- TEST INFRASTRUCTURE ONLY (see mo_math.h header for scope and pinning).
 *
 * packet_rgb-equivalent ray queries: 8 rays at a time on AVX2, the CPU BASELINE leg bench.py times beside the scalar
 * restatement.  This is a restatement of the *method* of the reference's packet traversal
 *   ShapeKDTree::ray_intersect_packet   include/mitsuba/render/kdtree.h:2176-2300
 * -- one traversal stack for the whole packet, a lane mask per stack entry, the packet descends into a child when ANY
 * active lane needs it, and the visiting order of two children is decided by a vote of the active lanes
 * (`left_votes >= right_votes`, kdtree.h:2228-2233) -- over the oracle's own bounding-volume hierarchy instead of a
 * kd-tree (only the query result is the contract: closest t / primitive / barycentrics, any-hit boolean).
 * The triangle test is Mesh::ray_intersect_triangle (include/mitsuba/render/mesh.h:195-221) on 8 lanes with the very
 * operation order of the scalar oracle (mo_scene.c tri_intersect), so hits are bit-identical to the scalar queries;
 * the box tests only cull and are widened so that they never reject what the scalar walk keeps.
 * Parity: pinned against the scalar oracle by tests/test_oracle_packet.py (identical films for identical schedules).
 */
#include "mo_internal.h"
#include <immintrin.h>
#include <stdlib.h>

typedef struct {
    float lo[2][3], hi[2][3];       /* boxes of the two children, rounded outward from the oracle's padded double boxes */
    uint32_t child[2];              /* inner node index, or 0x80000000 | leaf index */
} pk_node;
typedef struct { uint32_t first, count; } pk_leaf;

struct mo_packet_accel {
    pk_node *nodes; uint32_t n_nodes;
    pk_leaf *leaves; uint32_t n_leaves;
    /* triangles in leaf order, SoA: p0, e1 = p1 - p0, e2 = p2 - p0 (the differences the scalar test forms per call) */
    float *tri[9]; uint32_t *prim; uint32_t n_tris;
    uint32_t root;                  /* child reference of the root */
    float root_lo[3], root_hi[3];
};

static float round_down(double x) { float f = (float) x; return (double) f > x ? nextafterf(f, -INFINITY) : f; }
static float round_up(double x) { float f = (float) x; return (double) f < x ? nextafterf(f, INFINITY) : f; }

static uint32_t pk_convert(const mo_scene *s, mo_packet_accel *a, uint32_t idx) {
    const mo_bvh_node *n = &s->bvh_nodes[idx];
    if (n->count > 0) {
        uint32_t li = a->n_leaves++;
        a->leaves[li].first = a->n_tris; a->leaves[li].count = n->count;
        for (uint32_t i = 0; i < n->count; ++i) {
            uint32_t gp = s->bvh_prims[n->first + i];
            const mo_mesh *m = &s->meshes[s->prim_shape[gp]];
            uint32_t f = s->prim_local[gp];
            const float *p0 = m->pos + 3 * m->faces[3 * f], *p1 = m->pos + 3 * m->faces[3 * f + 1], *p2 = m->pos + 3 * m->faces[3 * f + 2];
            uint32_t k = a->n_tris++;
            for (int c = 0; c < 3; ++c) { a->tri[c][k] = p0[c]; a->tri[3 + c][k] = p1[c] - p0[c]; a->tri[6 + c][k] = p2[c] - p0[c]; }
            a->prim[k] = gp;
        }
        return 0x80000000u | li;
    }
    uint32_t ni = a->n_nodes++;
    const uint32_t ch[2] = { n->left, n->right };
    for (int c = 0; c < 2; ++c) {
        const mo_bvh_node *cn = &s->bvh_nodes[ch[c]];
        for (int k = 0; k < 3; ++k) { a->nodes[ni].lo[c][k] = round_down(cn->lo[k]); a->nodes[ni].hi[c][k] = round_up(cn->hi[k]); }
    }
    uint32_t l = pk_convert(s, a, n->left), r = pk_convert(s, a, n->right);
    a->nodes[ni].child[0] = l; a->nodes[ni].child[1] = r;
    return ni;
}

mo_packet_accel *mo_packet_accel_build(const mo_scene *s) {
    mo_packet_accel *a = (mo_packet_accel *) calloc(1, sizeof(*a));
    if (s->n_prims == 0) return a;
    a->nodes = (pk_node *) malloc(sizeof(pk_node) * (s->n_bvh_nodes + 1));
    a->leaves = (pk_leaf *) malloc(sizeof(pk_leaf) * (s->n_bvh_nodes + 1));
    for (int c = 0; c < 9; ++c) a->tri[c] = (float *) malloc(sizeof(float) * (s->n_prims + 8));
    a->prim = (uint32_t *) malloc(sizeof(uint32_t) * (s->n_prims + 8));
    a->root = pk_convert(s, a, 0);
    for (int k = 0; k < 3; ++k) { a->root_lo[k] = round_down(s->bvh_nodes[0].lo[k]); a->root_hi[k] = round_up(s->bvh_nodes[0].hi[k]); }
    return a;
}

void mo_packet_accel_free(mo_packet_accel *a) {
    if (!a) return;
    free(a->nodes); free(a->leaves); free(a->prim);
    for (int c = 0; c < 9; ++c) free(a->tri[c]);
    free(a);
}

/* a direction component too small to divide by: the slab is either missed or spans all t; 1e30 keeps (lo - o) * inv finite */
static inline __m256 safe_inv(__m256 d) {
    const __m256 sign = _mm256_and_ps(d, _mm256_set1_ps(-0.0f));
    const __m256 absd = _mm256_andnot_ps(_mm256_set1_ps(-0.0f), d);
    const __m256 small = _mm256_cmp_ps(absd, _mm256_set1_ps(1e-30f), _CMP_LT_OQ);
    const __m256 inv = _mm256_div_ps(_mm256_set1_ps(1.0f), d);
    return _mm256_blendv_ps(inv, _mm256_or_ps(_mm256_set1_ps(1e30f), sign), small);
}

typedef struct { __m256 ox, oy, oz, ix, iy, iz; } pk_rays;

/* slab test of 8 rays against one box, interval [t0, t1] widened by a few ulps on both ends; returns the lane mask and tnear */
static inline __m256 box_test(const pk_rays *r, const float lo[3], const float hi[3], __m256 t0, __m256 t1, __m256 *tnear) {
    const __m256 ax = _mm256_mul_ps(_mm256_sub_ps(_mm256_set1_ps(lo[0]), r->ox), r->ix), bx = _mm256_mul_ps(_mm256_sub_ps(_mm256_set1_ps(hi[0]), r->ox), r->ix);
    const __m256 ay = _mm256_mul_ps(_mm256_sub_ps(_mm256_set1_ps(lo[1]), r->oy), r->iy), by = _mm256_mul_ps(_mm256_sub_ps(_mm256_set1_ps(hi[1]), r->oy), r->iy);
    const __m256 az = _mm256_mul_ps(_mm256_sub_ps(_mm256_set1_ps(lo[2]), r->oz), r->iz), bz = _mm256_mul_ps(_mm256_sub_ps(_mm256_set1_ps(hi[2]), r->oz), r->iz);
    __m256 tn = _mm256_max_ps(_mm256_max_ps(_mm256_min_ps(ax, bx), _mm256_min_ps(ay, by)), _mm256_min_ps(az, bz));
    __m256 tf = _mm256_min_ps(_mm256_min_ps(_mm256_max_ps(ax, bx), _mm256_max_ps(ay, by)), _mm256_max_ps(az, bz));
    const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff));
    const __m256 slack = _mm256_set1_ps(4e-6f);
    tn = _mm256_sub_ps(tn, _mm256_mul_ps(_mm256_and_ps(tn, absmask), slack));
    tf = _mm256_add_ps(tf, _mm256_mul_ps(_mm256_and_ps(tf, absmask), slack));
    tn = _mm256_max_ps(tn, t0);
    tf = _mm256_min_ps(tf, t1);
    *tnear = tn;
    return _mm256_cmp_ps(tn, tf, _CMP_LE_OQ);
}

#define PK_STACK 256
typedef struct { uint32_t ref; uint32_t mask; float tnear[8]; } pk_entry;

/* lanes: bit mask of the rays to trace.  shadow != 0: returns the mask of occluded lanes (hits untouched).
 * shadow == 0: returns the mask of lanes with a hit and fills hits[lane]; among equal t the highest global primitive index
 * wins, the tie rule of the scalar oracle (mo_scene.c intersect_bvh). */
uint32_t mo_packet_intersect(const mo_scene *s, const mo_packet_accel *a, const mo_ray *rays, uint32_t lanes, int shadow, mo_hit *hits) {
    if (s->n_prims == 0 || lanes == 0) return 0;
    if (s->force_naive) {           /* brute-force scenes (tests): lane by lane through the scalar query */
        uint32_t out = 0;
        for (int l = 0; l < 8; ++l)
            if ((lanes >> l) & 1u) { mo_hit h; if (mo_intersect(s, &rays[l], shadow, 1, &h)) { out |= 1u << l; if (!shadow) hits[l] = h; } }
        return out;
    }
    float tmp[8][8];                /* ox oy oz dx dy dz mint maxt */
    for (int l = 0; l < 8; ++l) {
        const mo_ray *r = &rays[((lanes >> l) & 1u) ? l : __builtin_ctz(lanes)];      /* inactive lanes copy an active ray: no NaNs */
        tmp[0][l] = r->o.x; tmp[1][l] = r->o.y; tmp[2][l] = r->o.z; tmp[3][l] = r->d.x; tmp[4][l] = r->d.y; tmp[5][l] = r->d.z;
        tmp[6][l] = r->mint; tmp[7][l] = r->maxt;
    }
    const __m256 ox = _mm256_loadu_ps(tmp[0]), oy = _mm256_loadu_ps(tmp[1]), oz = _mm256_loadu_ps(tmp[2]);
    const __m256 dx = _mm256_loadu_ps(tmp[3]), dy = _mm256_loadu_ps(tmp[4]), dz = _mm256_loadu_ps(tmp[5]);
    const __m256 mint = _mm256_loadu_ps(tmp[6]), maxt = _mm256_loadu_ps(tmp[7]);
    pk_rays R = { ox, oy, oz, safe_inv(dx), safe_inv(dy), safe_inv(dz) };
    const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff));
    /* the scalar walk widens [mint, best] by 1e-5 relative + 1e-5 absolute; so does this one */
    const __m256 t0 = _mm256_sub_ps(mint, _mm256_mul_ps(_mm256_add_ps(_mm256_and_ps(mint, absmask), _mm256_set1_ps(1.0f)), _mm256_set1_ps(1e-5f)));
    __m256 best = maxt;             /* closest hit so far (ray.maxt shrinks on every hit, kdtree.h:2273) */
    __m256i best_prim = _mm256_set1_epi32(0);
    __m256 best_u = _mm256_setzero_ps(), best_v = _mm256_setzero_ps();
    uint32_t found = 0, active = lanes;
    static const uint32_t lane_bits[8] = { 1, 2, 4, 8, 16, 32, 64, 128 };
    const __m256i bits = _mm256_loadu_si256((const __m256i *) lane_bits);
#define MASK_TO_VEC(m) _mm256_castsi256_ps(_mm256_cmpeq_epi32(_mm256_and_si256(_mm256_set1_epi32((int) (m)), bits), bits))
    pk_entry stack[PK_STACK]; int sp = 0;
    uint32_t cur = a->root;
    {   /* scene bounding box first (kdtree.h:2199-2202) */
        __m256 tn;
        const __m256 t1 = _mm256_add_ps(best, _mm256_mul_ps(_mm256_add_ps(_mm256_and_ps(best, absmask), _mm256_set1_ps(1.0f)), _mm256_set1_ps(1e-5f)));
        active &= (uint32_t) _mm256_movemask_ps(box_test(&R, a->root_lo, a->root_hi, t0, t1, &tn));
        if (!active) return 0;
    }
    while (1) {
        if (!(cur & 0x80000000u)) {
            const pk_node *n = &a->nodes[cur];
            const __m256 t1 = _mm256_add_ps(best, _mm256_mul_ps(_mm256_add_ps(_mm256_and_ps(best, absmask), _mm256_set1_ps(1.0f)), _mm256_set1_ps(1e-5f)));
            __m256 tnl, tnr;
            const uint32_t hl = (uint32_t) _mm256_movemask_ps(box_test(&R, n->lo[0], n->hi[0], t0, t1, &tnl)) & active;
            const uint32_t hr = (uint32_t) _mm256_movemask_ps(box_test(&R, n->lo[1], n->hi[1], t0, t1, &tnr)) & active;
            if (hl && hr) {
                /* lane vote on the visiting order (kdtree.h:2228-2233): lanes that enter the left box first vs the others */
                const uint32_t left_first = (uint32_t) _mm256_movemask_ps(_mm256_cmp_ps(tnl, tnr, _CMP_LE_OQ));
                const uint32_t voters = hl | hr;
                const int left_votes = __builtin_popcount(left_first & voters), right_votes = __builtin_popcount(~left_first & voters);
                const int go_left = left_votes >= right_votes;
                pk_entry *e = &stack[sp++];
                e->ref = n->child[go_left ? 1 : 0]; e->mask = go_left ? hr : hl;
                _mm256_storeu_ps(e->tnear, go_left ? tnr : tnl);
                cur = n->child[go_left ? 0 : 1]; active = go_left ? hl : hr;
                continue;
            } else if (hl) { cur = n->child[0]; active = hl; continue; }
            else if (hr) { cur = n->child[1]; active = hr; continue; }
        } else {
            const pk_leaf *lf = &a->leaves[cur & 0x7fffffffu];
            const __m256 am = MASK_TO_VEC(active);
            for (uint32_t i = lf->first; i < lf->first + lf->count; ++i) {
                /* Mesh::ray_intersect_triangle (mesh.h:195-221), operation order of mo_scene.c tri_intersect */
                const __m256 p0x = _mm256_set1_ps(a->tri[0][i]), p0y = _mm256_set1_ps(a->tri[1][i]), p0z = _mm256_set1_ps(a->tri[2][i]);
                const __m256 e1x = _mm256_set1_ps(a->tri[3][i]), e1y = _mm256_set1_ps(a->tri[4][i]), e1z = _mm256_set1_ps(a->tri[5][i]);
                const __m256 e2x = _mm256_set1_ps(a->tri[6][i]), e2y = _mm256_set1_ps(a->tri[7][i]), e2z = _mm256_set1_ps(a->tri[8][i]);
                /* pvec = cross(d, e2) */
                const __m256 pvx = _mm256_fmsub_ps(dy, e2z, _mm256_mul_ps(dz, e2y));
                const __m256 pvy = _mm256_fmsub_ps(dz, e2x, _mm256_mul_ps(dx, e2z));
                const __m256 pvz = _mm256_fmsub_ps(dx, e2y, _mm256_mul_ps(dy, e2x));
                const __m256 det = _mm256_fmadd_ps(e1z, pvz, _mm256_fmadd_ps(e1y, pvy, _mm256_mul_ps(e1x, pvx)));
                const __m256 inv_det = _mm256_div_ps(_mm256_set1_ps(1.0f), det);
                const __m256 tx = _mm256_sub_ps(ox, p0x), ty = _mm256_sub_ps(oy, p0y), tz = _mm256_sub_ps(oz, p0z);
                const __m256 u = _mm256_mul_ps(_mm256_fmadd_ps(tz, pvz, _mm256_fmadd_ps(ty, pvy, _mm256_mul_ps(tx, pvx))), inv_det);
                /* qvec = cross(tvec, e1) */
                const __m256 qx = _mm256_fmsub_ps(ty, e1z, _mm256_mul_ps(tz, e1y));
                const __m256 qy = _mm256_fmsub_ps(tz, e1x, _mm256_mul_ps(tx, e1z));
                const __m256 qz = _mm256_fmsub_ps(tx, e1y, _mm256_mul_ps(ty, e1x));
                const __m256 v = _mm256_mul_ps(_mm256_fmadd_ps(dz, qz, _mm256_fmadd_ps(dy, qy, _mm256_mul_ps(dx, qx))), inv_det);
                const __m256 t = _mm256_mul_ps(_mm256_fmadd_ps(e2z, qz, _mm256_fmadd_ps(e2y, qy, _mm256_mul_ps(e2x, qx))), inv_det);
                __m256 ok = _mm256_and_ps(_mm256_cmp_ps(u, _mm256_setzero_ps(), _CMP_GE_OQ), _mm256_cmp_ps(u, _mm256_set1_ps(1.0f), _CMP_LE_OQ));
                ok = _mm256_and_ps(ok, _mm256_and_ps(_mm256_cmp_ps(v, _mm256_setzero_ps(), _CMP_GE_OQ),
                                                     _mm256_cmp_ps(_mm256_add_ps(u, v), _mm256_set1_ps(1.0f), _CMP_LE_OQ)));
                ok = _mm256_and_ps(ok, _mm256_and_ps(_mm256_cmp_ps(t, mint, _CMP_GE_OQ), _mm256_cmp_ps(t, maxt, _CMP_LE_OQ)));
                ok = _mm256_and_ps(ok, am);
                const uint32_t okm = (uint32_t) _mm256_movemask_ps(ok);
                if (!okm) continue;
                if (shadow) { found |= okm; continue; }
                /* t < best, or t == best and a higher primitive index (or the first hit of the lane) */
                const __m256i gp = _mm256_set1_epi32((int) a->prim[i]);
                const __m256 fm = MASK_TO_VEC(found);
                const __m256 lt = _mm256_cmp_ps(t, best, _CMP_LT_OQ);
                const __m256 eq_hi = _mm256_and_ps(_mm256_cmp_ps(t, best, _CMP_EQ_OQ),
                                                   _mm256_castsi256_ps(_mm256_cmpgt_epi32(_mm256_xor_si256(gp, _mm256_set1_epi32((int) 0x80000000u)),
                                                                                          _mm256_xor_si256(best_prim, _mm256_set1_epi32((int) 0x80000000u)))));
                const __m256 take = _mm256_and_ps(ok, _mm256_or_ps(_mm256_andnot_ps(fm, _mm256_castsi256_ps(_mm256_set1_epi32(-1))), _mm256_or_ps(lt, eq_hi)));
                best = _mm256_blendv_ps(best, t, take);
                best_u = _mm256_blendv_ps(best_u, u, take);
                best_v = _mm256_blendv_ps(best_v, v, take);
                best_prim = _mm256_castps_si256(_mm256_blendv_ps(_mm256_castsi256_ps(best_prim), _mm256_castsi256_ps(gp), take));
                found |= (uint32_t) _mm256_movemask_ps(take);
            }
        }
        /* pop: lanes that are done (shadow rays with a hit) or whose closest hit is nearer than the entry point drop out */
        while (1) {
            if (sp == 0) goto done;
            const pk_entry *e = &stack[--sp];
            uint32_t m = e->mask;
            if (shadow) m &= ~found;
            else {
                const __m256 t1 = _mm256_add_ps(best, _mm256_mul_ps(_mm256_add_ps(_mm256_and_ps(best, absmask), _mm256_set1_ps(1.0f)), _mm256_set1_ps(1e-5f)));
                m &= (uint32_t) _mm256_movemask_ps(_mm256_cmp_ps(_mm256_loadu_ps(e->tnear), t1, _CMP_LE_OQ));
            }
            if (m) { cur = e->ref; active = m; break; }
        }
        if (shadow && (found & lanes) == lanes) break;
    }
done:
    if (!shadow && found) {
        float bt[8], bu[8], bv[8]; uint32_t bp[8];
        _mm256_storeu_ps(bt, best); _mm256_storeu_ps(bu, best_u); _mm256_storeu_ps(bv, best_v);
        _mm256_storeu_si256((__m256i *) bp, best_prim);
        for (int l = 0; l < 8; ++l)
            if ((found >> l) & 1u) { hits[l].t = bt[l]; hits[l].prim = bp[l]; hits[l].u = bu[l]; hits[l].v = bv[l]; }
    }
    return found & lanes;
}

/* SoA entry point for the tests: closest hit (t, prim, u, v; inf / 0xffffffff on a miss) and any hit of n rays, 8 at a time */
void mo_packet_ray_intersect(const mo_scene *s, uint64_t n, const float *ox, const float *oy, const float *oz, const float *dx,
                             const float *dy, const float *dz, const float *mint, const float *maxt, float *t, uint32_t *prim,
                             float *u, float *v, uint8_t *any_hit) {
    mo_packet_accel *a = mo_packet_accel_build(s);
    for (uint64_t b = 0; b < n; b += 8) {
        mo_ray r[8]; mo_hit h[8]; uint32_t lanes = 0;
        for (uint32_t l = 0; l < 8 && b + l < n; ++l) {
            const uint64_t i = b + l;
            r[l].o = mo_v3_make(ox[i], oy[i], oz[i]); r[l].d = mo_v3_make(dx[i], dy[i], dz[i]); r[l].mint = mint[i]; r[l].maxt = maxt[i];
            lanes |= 1u << l;
        }
        const uint32_t found = mo_packet_intersect(s, a, r, lanes, 0, h), occl = mo_packet_intersect(s, a, r, lanes, 1, NULL);
        for (uint32_t l = 0; l < 8 && b + l < n; ++l) {
            const uint64_t i = b + l; const int f = (found >> l) & 1u;
            t[i] = f ? h[l].t : INFINITY; prim[i] = f ? h[l].prim : 0xffffffffu; u[i] = f ? h[l].u : 0.0f; v[i] = f ? h[l].v : 0.0f;
            any_hit[i] = (uint8_t) ((occl >> l) & 1u);
        }
    }
    mo_packet_accel_free(a);
}
