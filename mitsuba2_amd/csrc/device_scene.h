// device_scene.h -- device-side scene view, BVH traversal, surface interaction and the
// light-transport stages of the `path` integrator, as inline device functions so that the
// fused and the split wavefront kernels are built from the same code.
//
// Reference semantics followed (paths relative to the Mitsuba 2 tree):
//   Mesh::ray_intersect_triangle           include/mitsuba/render/mesh.h:195-221
//   query contract of the kd-tree walk     include/mitsuba/render/kdtree.h:2079-2174
//   create_surface_interaction             include/mitsuba/render/kdtree.h:2334-2367
//   Mesh::fill_surface_interaction         src/librender/mesh.cpp:399-462
//   Mesh::sample_position                  src/librender/mesh.cpp:320-365
//   DiscreteDistribution::sample_reuse     include/mitsuba/core/distr_1d.h:144-203
//   Shape::sample_direction/pdf_direction  src/librender/shape.cpp:252-283
//   AreaLight                              src/emitters/area.cpp:71-125
//   Scene::sample_emitter_direction        src/librender/scene.cpp:141-206
//   SmoothDiffuse                          src/bsdfs/diffuse.cpp:78-135
#pragma once
#include "device_math.h"
#include "device_bsdf.h"
#include "device_envmap.h"

namespace mtsamd {

constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kLeafCountShift = 27;
constexpr uint32_t kLeafStartMask = (1u << kLeafCountShift) - 1u;
constexpr uint32_t kNoPrim = 0xffffffffu;

#ifndef MTS_BVH4
#define MTS_BVH4 1
#endif
#if MTS_BVH4
// BVH4 walk: an entry is (child reference, distance at which the ray enters the child's box): a popped subtree whose entry distance
// lies beyond the closest hit found meanwhile is dropped without touching its node.
typedef uint2 StackEntry;
typedef uint64_t StackRaw;      // the same entry as one LDS word
#else
typedef uint32_t StackEntry;
typedef uint32_t StackRaw;
#endif

// BVH2 node, 64 B: both child boxes + child references.
//   q0 = (l.min.x, l.min.y, l.min.z, l.max.x)   q1 = (l.max.y, l.max.z, r.min.x, r.min.y)
//   q2 = (r.min.z, r.max.x, r.max.y, r.max.z)   q3 = (left ref, right ref, -, -) as bits
// child ref: inner node index, or kLeafFlag | count << 27 | first triangle slot.
// Triangle slot, 48 B: t0 = (p0.xyz, e1.x) t1 = (e1.y, e1.z, e2.x, e2.y) t2 = (e2.z, prim id bits, -, -)

struct DevShape { int32_t bsdf; int32_t emitter; uint32_t flags; uint32_t first_prim; };
constexpr uint32_t kShapeHasNormals = 1u, kShapeHasUV = 2u;
struct DevTexture {
    const float *data; int32_t w, h; uint32_t grad_offset, kind;        // kind 0: bitmap, 1: checkerboard (checkerboard.cpp)
    float uvm[6];                                                       // to_uv: uv' = (m0 u + m1 v + m2, m3 u + m4 v + m5)
    float c0[3], c1[3];                                                 // checkerboard colours (spectral variant: model coefficients)
    float mean;                                                         // Texture::mean() (plastic lobe weights)
};   // grad_offset: float offset in the concatenated gradient buffer       // linear RGB bitmap (src/textures/bitmap.cpp), identity to_uv
struct DevEmitter {
    float r, g, b; uint32_t shape;
    uint32_t first_prim, n_prims; float area_sum, area_norm;
    uint32_t valid_lo, valid_hi; uint32_t pad0, pad1;
    float c0, c1, c2, d65_scale;       // spectral variant: SRGBEmitterSpectrum = D65 * d65_scale * srgb_model(c) (srgb_d65.cpp:27-63)
    float cx, cy, cz, radius;          // constant emitter (pad0 == 1): the scene's bounding sphere (constant.cpp:47-51);
                                       // point / spot: position; directional: radius of the bounding sphere
    // delta emitters -- spot: aux[0..8] = world-to-local rotation, aux[9] = cutoff angle, aux[10] = cos(cutoff), aux[11] = cos(beam width),
    // aux[12] = 1 / (cutoff - beam width) (spot.cpp:81-90); directional: aux[0..2] = direction the light travels in
    float aux[16];
};
constexpr uint32_t kEmitterConstant = 1u, kEmitterEnvmap = 2u, kEmitterPoint = 3u, kEmitterSpot = 4u, kEmitterDirectional = 5u;     // DevEmitter::pad0
constexpr float kInvFourPi = 0.07957747154594766788f;

struct SceneView {
    const float4 *nodes;       // 4 per node
    const uint4 *qnodes;       // 2 per node: child boxes on a 16-bit grid (origin q_lo, cell q_step), child references
    const uint4 *wnodes;       // BVH4 collapsed from the BVH2, 4 per node: per child (x, y, z) = lo | hi << 16 on the same grid, w = child
                               // reference (kNoNode: no child, its box is inverted); wroot = reference of the root
    uint32_t wroot, n_wnodes;
    float q_lo[3], q_step[3], q_inv_step[3];      // q_inv_step = 1 / q_step
    const float4 *tris;        // 3 per slot (leaf order)
    uint32_t root;             // child ref of the root
    uint32_t n_nodes, n_slots, n_prims;
    uint32_t lds_nodes;        // nodes [0, lds_nodes) are staged in LDS
    uint32_t lds_slots;        // triangle slots [0, lds_slots) are staged in LDS
    uint32_t stack_depth;      // entries per lane
    // standalone ray queries (k_ray_walk): LDS part of the per-lane stack, global spill area [workgroup][entry][thread] for the rest,
    // number of (persistent) workgroups the spill area is sized for
    uint32_t walk_lds_depth, walk_blocks;
    StackEntry *walk_spill;
    const float *tri_pos;      // 9 per prim (p0,p1,p2)
    const float *tri_nrm;      // 9 per prim or nullptr
    const float *tri_uv;       // 6 per prim or nullptr
    const uint32_t *prim_shape;
    const DevShape *shapes;
    const DevBsdf *bsdfs;
    const DevEmitter *emitters; uint32_t n_emitters;
    uint32_t n_shapes, n_bsdfs;
    const DevTexture *textures; uint32_t n_textures;
    const float *area_pmf, *area_cdf;   // per prim (global index), valid inside emitter ranges
    // "flat" scenes (n_prims <= kFlatMaxPrims): no hierarchy pays off; the whole scene is kept in LDS
    // as 64-byte records in primitive order and every query is a wave-uniform loop over them.
    //   r0 = (p0.xyz, e1.x) r1 = (e1.y, e1.z, e2.x, e2.y) r2 = (e2.z, p1.xyz) r3 = (p2.xyz, shape bits)
    // plus 80-byte intersection records for primitive pairs (A = 2k, B = 2k+1), laid out for packed fp32 math:
    //   (p0x.ab, p0y.ab) (p0z.ab, e1x.ab) (e1y.ab, e1z.ab) (e2x.ab, e2y.ab) (e2z.ab, -, -)
    const float4 *flat_recs;
    const float4 *flat_pairs;  // 5 float4 per primitive pair, followed by 2 float4 per pair cluster (padded box; .w of the first = pair count)
    uint32_t flat, n_pairs, n_clusters;
    int32_t env_emitter;       // index of the environment emitter or -1 (scene.cpp:44-48)
    const DevEnvmap *envmap;   // its image + sampling hierarchy if it is an `envmap`
    uint32_t general;          // some BSDF is not a one-sided `diffuse`: kernels instantiated with the BSDF switch are used
};
constexpr uint32_t kFlatMaxPrims = 64;

// LDS carve-up of one workgroup.
//   hierarchy scenes: [BVH nodes (top of the tree)][triangle slots][traversal stack]
//   flat scenes:      [flat records][shapes][bsdfs][emitters][area pmf][area cdf]
struct LdsView {
    const float4 *nodes;
    const float4 *tris;
    uint32_t *stack;           // [depth][blockDim]
    uint32_t stride;           // blockDim.x
    StackEntry *spill;         // null: the whole traversal stack lives in LDS; else this thread's column of a global spill area ...
    uint32_t spill_stride, stack_lds_depth;      // ... that takes the entries beyond the first stack_lds_depth
    const float4 *flat;        // 4 per prim (shading records)
    const float4 *pairs;       // 5 per primitive pair (intersection records), + 1 all-zero pair
    const float4 *clusters;    // 2 per cluster of consecutive pairs (one shape): (lo.xyz, pair count), (hi.xyz, -)
    const DevShape *shapes;
    const DevBsdf *bsdfs;
    const DevEmitter *emitters;
    const float *pmf, *cdf;
};

inline uint32_t flat_shape_count(const SceneView &sv) { return sv.n_shapes; }

template <bool FLAT>
MTS_DEV LdsView lds_stage(const SceneView &sv, float4 *smem) {
    LdsView l = {};
    l.stride = blockDim.x;
    if (FLAT) {
        float4 *f = smem;
        for (uint32_t i = threadIdx.x; i < 4u * sv.n_prims; i += blockDim.x) f[i] = sv.flat_recs[i];
        l.flat = f;
        float4 *pr = f + 4u * sv.n_prims;
        for (uint32_t i = threadIdx.x; i < 5u * sv.n_pairs; i += blockDim.x) pr[i] = sv.flat_pairs[i];
        if (threadIdx.x < 5u) pr[5u * sv.n_pairs + threadIdx.x] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        l.pairs = pr;
        float4 *cl = pr + 5u * sv.n_pairs + 5u;
        for (uint32_t i = threadIdx.x; i < 2u * sv.n_clusters; i += blockDim.x) cl[i] = sv.flat_pairs[5u * sv.n_pairs + i];
        l.clusters = cl;
        uint32_t *w = reinterpret_cast<uint32_t *>(cl + 2u * sv.n_clusters);
        const uint32_t n_sh = (sizeof(DevShape) / 4u) * sv.n_shapes, n_bs = (sizeof(DevBsdf) / 4u) * sv.n_bsdfs,
                       n_em = (sizeof(DevEmitter) / 4u) * sv.n_emitters;
        const uint32_t *g_sh = reinterpret_cast<const uint32_t *>(sv.shapes), *g_bs = reinterpret_cast<const uint32_t *>(sv.bsdfs),
                       *g_em = reinterpret_cast<const uint32_t *>(sv.emitters);
        for (uint32_t i = threadIdx.x; i < n_sh; i += blockDim.x) w[i] = g_sh[i];
        l.shapes = reinterpret_cast<const DevShape *>(w); w += n_sh;
        for (uint32_t i = threadIdx.x; i < n_bs; i += blockDim.x) w[i] = g_bs[i];
        l.bsdfs = reinterpret_cast<const DevBsdf *>(w); w += n_bs;
        for (uint32_t i = threadIdx.x; i < n_em; i += blockDim.x) w[i] = g_em[i];
        l.emitters = reinterpret_cast<const DevEmitter *>(w); w += n_em;
        float *fw = reinterpret_cast<float *>(w);
        for (uint32_t i = threadIdx.x; i < sv.n_prims; i += blockDim.x) { fw[i] = sv.area_pmf[i]; fw[sv.n_prims + i] = sv.area_cdf[i]; }
        l.pmf = fw; l.cdf = fw + sv.n_prims;
    } else {
        float4 *n = smem;
        float4 *t = n + 4u * sv.lds_nodes;
        for (uint32_t i = threadIdx.x; i < 4u * sv.lds_nodes; i += blockDim.x) n[i] = sv.nodes[i];
        for (uint32_t i = threadIdx.x; i < 3u * sv.lds_slots; i += blockDim.x) t[i] = sv.tris[i];
        l.nodes = n; l.tris = t;
        l.stack = reinterpret_cast<uint32_t *>(t + 3u * sv.lds_slots);
    }
    __syncthreads();
    return l;
}
inline size_t lds_bytes(const SceneView &sv, uint32_t block) {
    if (sv.flat)
        return (size_t) 64 * sv.n_prims + (size_t) 80 * (sv.n_pairs + 1) + (size_t) 32 * sv.n_clusters + sizeof(DevShape) * sv.n_shapes + sizeof(DevBsdf) * sv.n_bsdfs +
               sizeof(DevEmitter) * sv.n_emitters + (size_t) 8 * sv.n_prims;
    return (size_t) 64 * sv.lds_nodes + (size_t) 48 * sv.lds_slots + sizeof(StackEntry) * sv.stack_depth * block;
}

#ifndef MTS_FLAT_UNROLL
#define MTS_FLAT_UNROLL 1
#endif
#ifndef MTS_FLAT_PREFETCH
#define MTS_FLAT_PREFETCH 1
#endif

struct Hit { float t; uint32_t prim; float u, v; };

// Moeller-Trumbore exactly as mesh.h:195-221 (no culling, closed intervals).
MTS_DEV bool tri_test(f3 p0, f3 e1, f3 e2, f3 o, f3 d, float mint, float maxt, float &u, float &v, float &t) {
    f3 pvec = cross(d, e2);
    float inv_det = rcp(dot(e1, pvec));
    f3 tvec = o - p0;
    u = dot(tvec, pvec) * inv_det;
    bool ok = (u >= 0.0f) && (u <= 1.0f);
    f3 qvec = cross(tvec, e1);
    v = dot(d, qvec) * inv_det;
    ok = ok && (v >= 0.0f) && (u + v <= 1.0f);
    t = dot(e2, qvec) * inv_det;
    return ok && (t >= mint) && (t <= maxt);
}

MTS_DEV float clamp_mag(float r) { return fabsf(r) <= 3.0e38f ? r : copysignf(3.0e38f, r); }
MTS_DEV float clamp_mag33(float r) { return fabsf(r) <= 1.0e33f ? r : copysignf(1.0e33f, r); }
MTS_DEV float clamp_inv(float d) {
    float r = 1.0f / d;
    // zero / denormal components: keep the slab test NaN-free (0 * huge = 0, never inf * 0)
    return fabsf(r) <= 3.0e38f ? r : copysignf(3.0e38f, d);
}

// Hierarchy scenes.  Closest-hit (ANY=false) or any-hit (ANY=true) query with a per-lane stack in LDS.
// Among hits with exactly equal t the highest primitive index wins (what the brute-force loop of
// ray_intersect_naive produces).  The slab test only culls; boxes are padded on the host so that it never
// rejects a triangle the fp32 Moeller-Trumbore test would accept.
constexpr uint32_t kNoNode = 0x7fffffffu;       // "nothing left": not a leaf, never a valid inner index

// State of one BVH walk.  `cur` == kNoNode: finished (or never started).
#ifndef MTS_QNODES
#define MTS_QNODES 1
#endif
// BVH4 planes as fp16 values of (grid coordinate - 32768), fed straight into v_fma_mix_f32 (fp16 operand x fp32 + fp32 in one
// instruction): 24 v_cvt_f32_u32 fewer per step; the boxes are coarser (11 significant bits instead of a 16-bit grid)
#ifndef MTS_NODE_F16
#define MTS_NODE_F16 0
#endif
#ifndef MTS_WALK_T
#define MTS_WALK_T 24
#endif
#ifndef MTS_NODE_P15
#define MTS_NODE_P15 0      // 1: BVH4 planes as 15-bit values the walk turns into floats with v_perm_b32 (measured equal); 0: 16-bit planes + v_cvt
#endif
#ifndef MTS_WALK_T_ANY
#define MTS_WALK_T_ANY MTS_WALK_T
#endif
struct BvhWalk {
    f3 o, d, inv; float mint, maxt, best;      // MTS_QNODES: o / inv of the slab test are in grid units (o_q, inv_q)
    f3 noi;                                    // noi = -(o_q * inv): t = fma(q, inv, noi); o_q itself is not kept (walk_origin_q)
    bool far;                                  // origin too far from the scene box for the fma form (cancellation)
    uint32_t sel[3];                           // v_perm_b32 selectors: (lo | hi << 16) -> (near | far << 16) per axis
    uint32_t sp, cur, best_prim; bool found;
    Hit hit;
};
// Ray origin in grid units (plus the offset of the node format).  Only the FAR form of the slab test needs it per step, and walks
// from that far away are rare: it is recomputed there instead of occupying three registers of every walk.
MTS_DEV f3 walk_origin_q(const SceneView &sv, f3 o) {
#if MTS_QNODES
    f3 o_q = mk3((o.x - sv.q_lo[0]) * sv.q_inv_step[0], (o.y - sv.q_lo[1]) * sv.q_inv_step[1], (o.z - sv.q_lo[2]) * sv.q_inv_step[2]);
#if MTS_NODE_F16 && MTS_BVH4
    o_q = mk3(o_q.x - 32768.0f, o_q.y - 32768.0f, o_q.z - 32768.0f);      // the fp16 planes are centred on the middle of the grid
#elif MTS_NODE_P15 && MTS_BVH4
    o_q = mk3(o_q.x + 65536.0f, o_q.y + 65536.0f, o_q.z + 65536.0f);      // a plane is decoded as 65536 + grid coordinate
#endif
    return o_q;
#else
    return o;
#endif
}
MTS_DEV void walk_begin(BvhWalk &w, const SceneView &sv, f3 o, f3 d, float mint, float maxt) {
    w.o = o; w.d = d;
#if MTS_QNODES
    // box coordinate x = q_lo + q * q_step  =>  t = (q - o_q) * inv_q with o_q = (o - q_lo) / q_step, inv_q = q_step / d
    // |inv| <= 1e33 keeps q * inv finite for q <= 65535 (a direction component that small is parallel to the slab either way).
    // The slab test only culls, against boxes padded by 1/8 cell or more, so the walk's own frame is set up with the cheap forms:
    // v_rcp_f32 (1 ulp) instead of a division, a multiplication by 1 / q_step -- a walk starts every ~8 steps and the exact
    // divisions were a tenth of k_trace's VALU time.  Error budget: see `far` below.
    w.inv = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    const f3 o_q = walk_origin_q(sv, o);
    w.inv = mk3(clamp_mag33(w.inv.x * sv.q_step[0]), clamp_mag33(w.inv.y * sv.q_step[1]), clamp_mag33(w.inv.z * sv.q_step[2]));
    w.noi = mk3(-(o_q.x * w.inv.x), -(o_q.y * w.inv.y), -(o_q.z * w.inv.z));
    // fma(q, inv, noi) is off by at most eps * |o_q| grid cells, o_q itself by another half of that (the multiplication by 1 / q_step)
    // and inv by 1 ulp, which moves t * |d_q| by up to eps * (|o_q| + 65535) cells: 0.015 + 0.008 + 0.04 cells at |o_q| = 2.5e5 (an origin
    // ~4 scene extents away); the boxes are padded by 1/8 cell or more.  Farther origins use (q - o_q) * inv with an explicit pad.
    w.far = !(hmax_abs(o_q) <= 2.5e5f);
#if MTS_NODE_P15 && MTS_BVH4
    // selector of the NEAR plane: (0x47 from the constant, the plane's two bytes, 0x00) -- low half of the word if the ray runs up
    // the axis, high half otherwise; the far plane's selector is this one ^ kP15Flip
    w.sel[0] = w.inv.x >= 0.0f ? 0x0305040cu : 0x0307060cu;
    w.sel[1] = w.inv.y >= 0.0f ? 0x0305040cu : 0x0307060cu;
    w.sel[2] = w.inv.z >= 0.0f ? 0x0305040cu : 0x0307060cu;
#else
    w.sel[0] = w.inv.x >= 0.0f ? 0x03020100u : 0x01000302u;
    w.sel[1] = w.inv.y >= 0.0f ? 0x03020100u : 0x01000302u;
    w.sel[2] = w.inv.z >= 0.0f ? 0x03020100u : 0x01000302u;
#endif
#else
    w.inv = mk3(clamp_inv(d.x), clamp_inv(d.y), clamp_inv(d.z));
    w.noi = o; w.far = true; w.sel[0] = w.sel[1] = w.sel[2] = 0u;
#endif
    w.mint = mint; w.maxt = maxt; w.best = maxt;
    w.sp = 0; w.cur = MTS_BVH4 ? sv.wroot : sv.root; w.best_prim = kNoPrim; w.found = false;
}
// One round of the "while-while" traversal: the lane descends until it holds a leaf (or nothing), then the wave tests leaves
// together -- the two phases are not interleaved lane by lane, which keeps more lanes busy in each of them.  ANY: the walk
// ends (cur = kNoNode, found = true) at the first hit.
// Traversal stack of one lane: the first `lds_depth` entries live in LDS ([depth][lane], conflict-free), deeper ones -- rare:
// a walk seldom defers more than a dozen subtrees -- in a global spill area, so that the LDS footprint does not cap occupancy.
// top_nodes / n_top (k_trace): the first n_top BVH4 nodes -- the top of the tree, BFS order -- staged in LDS by the workgroup (0: none)
struct WalkStack { StackEntry *lds; uint32_t shift, lds_depth; StackEntry *spill; uint32_t spill_stride; const uint4 *top_nodes; uint32_t n_top; };      // shift: log2 of the row stride (threads per workgroup, a power of two)
typedef uint32_t node_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const node_u4 LdsNodeWord;
// The LDS part is addressed through an explicit LDS pointer (ds_read / ds_write of a whole entry; a pointer select between LDS and
// the spill area compiles to flat accesses of half entries).  Row min(sp, lds_depth) is touched unconditionally: with a spill area
// the row after the lds_depth real ones is a scratch row, so the deep (rare) case only adds the spill access behind a branch that
// is almost never taken; lds_depth = 0xffffffff: the whole stack is in LDS.
typedef __attribute__((address_space(3))) StackRaw LdsStackRaw;
MTS_DEV LdsStackRaw *stack_row(const WalkStack &st, uint32_t sp) {
    return reinterpret_cast<LdsStackRaw *>((uint32_t) reinterpret_cast<uintptr_t>(st.lds)) + (min(sp, st.lds_depth) << st.shift);
}
MTS_DEV uint32_t log2_stride(uint32_t stride) { return 31u - (uint32_t) __builtin_clz(stride); }
#if MTS_BVH4
MTS_DEV StackRaw stack_raw(StackEntry v) { return (uint64_t) v.x | ((uint64_t) v.y << 32); }
MTS_DEV StackEntry stack_entry(StackRaw r) { return make_uint2((uint32_t) r, (uint32_t) (r >> 32)); }
#else
MTS_DEV StackRaw stack_raw(StackEntry v) { return v; }
MTS_DEV StackEntry stack_entry(StackRaw r) { return r; }
#endif
MTS_DEV void stack_push(const WalkStack &st, uint32_t sp, StackEntry v) {
    *stack_row(st, sp) = stack_raw(v);
    if (sp >= st.lds_depth) st.spill[(size_t) (sp - st.lds_depth) * st.spill_stride] = v;
}
// Branch-free push: the entry is written above the top in any case and only counted when `take` (the next push overwrites it).
MTS_DEV void stack_push_if(const WalkStack &st, uint32_t &sp, bool take, StackEntry v) {
    *stack_row(st, sp) = stack_raw(v);
    if (take && sp >= st.lds_depth) st.spill[(size_t) (sp - st.lds_depth) * st.spill_stride] = v;
    sp += take ? 1u : 0u;
}
MTS_DEV StackEntry stack_pop(const WalkStack &st, uint32_t sp) {
    // volatile: keeps the LDS read a ds_read of the whole entry (otherwise it is merged with the spill read into flat loads)
    StackEntry e = stack_entry(*const_cast<const volatile LdsStackRaw *>(stack_row(st, sp)));
    if (sp >= st.lds_depth) e = st.spill[(size_t) (sp - st.lds_depth) * st.spill_stride];
    return e;
}

// Experiment builds (-DMTS_TRACE_PROF): where the lanes of the k_trace waves go.  Wave-level event counts and lane sums per phase,
// added by the first active lane to one of 64 copies of the counter row (scripts/debug/trace_prof.py reads them).
#ifndef MTS_TRACE_PROF
#define MTS_TRACE_PROF 0
#endif
#if MTS_TRACE_PROF
static __device__ unsigned long long g_trace_prof[64 * 64];
template <bool ANY> MTS_DEV void prof_add(int i) {      // placed inside a (divergent) region: +1 event, +active lanes
    const uint64_t exec = __ballot(true);
    if (lane_id() == (uint32_t) __ffsll((long long) exec) - 1u) {
        unsigned long long *row = g_trace_prof + 64u * (blockIdx.x & 63u) + (ANY ? 32 : 0);
        atomicAdd(row + i, 1ull);
        atomicAdd(row + i + 1, (unsigned long long) __popcll(exec));
    }
}
template <bool ANY> MTS_DEV void prof_mask(int i, uint64_t mask) {      // +1 event, +lanes of a (wave-uniform) mask
    const uint64_t exec = __ballot(true);
    if (lane_id() == (uint32_t) __ffsll((long long) exec) - 1u) {
        unsigned long long *row = g_trace_prof + 64u * (blockIdx.x & 63u) + (ANY ? 32 : 0);
        atomicAdd(row + i, 1ull);
        atomicAdd(row + i + 1, (unsigned long long) __popcll(mask));
    }
}
#define MTS_PROF(ANY, i) prof_add<ANY>(i)
#define MTS_PROF_MASK(ANY, i, mask) prof_mask<ANY>(i, mask)
#else
#define MTS_PROF(ANY, i) ((void) 0)
#define MTS_PROF_MASK(ANY, i, mask) ((void) 0)
#endif

#if MTS_BVH4
// Triangle tests of the leaf a lane holds (all lanes of the wave together), then the next subtree from the stack.
#ifndef MTS_LEAF_STEP
#define MTS_LEAF_STEP 0
#endif
#ifndef MTS_CLOSEST_SORT
#define MTS_CLOSEST_SORT 5  // compare-exchanges of the closest-hit walk's child sort: 5 = full order, 4 / 3 = nearest first, the rest partly ordered
#endif
#ifndef MTS_ANY_SORT
#define MTS_ANY_SORT 1      // 2: full sort for any-hit walks as well, 1: nearest first only, 0: none
#endif
template <bool ANY>
MTS_DEV void walk_leaf(BvhWalk &w, const SceneView &sv, uint32_t &cur, uint32_t &tri_tests) {
    const uint32_t start = cur & kLeafStartMask, total = (cur >> kLeafCountShift) & 0xfu;
    auto test_one = [&](uint32_t slot) -> bool {
        const float4 *p = sv.tris + 3u * slot;
        const float4 t0 = p[0], t1 = p[1], t2 = p[2];
        float u, v, t;
        ++tri_tests;
        // closest hit: the interval ends at the closest hit so far (w.best <= maxt, equal until something is found), which is what the
        // update rule below would enforce anyway; the hit lives in (best, best_prim, hit.u, hit.v) -- no second copy of t and prim
        if (tri_test(mk3(t0.x, t0.y, t0.z), mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, t2.x), w.o, w.d, w.mint, ANY ? w.maxt : w.best, u, v, t)) {
            if (ANY) { w.found = true; return true; }
            const uint32_t prim = __float_as_uint(t2.y);
            if (!w.found || t < w.best || (t == w.best && prim > w.best_prim)) {
                w.found = true; w.best = t; w.best_prim = prim;
                w.hit.u = u; w.hit.v = v;
            }
        }
        return false;
    };
#if MTS_LEAF_STEP == 1
    // one triangle per leaf phase; the rest of the leaf stays in `cur` for the next round: the wave's leaf phase has no trip-count
    // divergence and a lane with a long leaf does not hold the others
    cur = total > 1u ? cur - (1u << kLeafCountShift) + 1u : kNoNode;
    test_one(start);
#else
    cur = kNoNode;
    MTS_PROF(ANY, 4);                                   // leaf phases
    for (uint32_t i = 0; i < total; ++i) {
        MTS_PROF(ANY, 6);                               // triangle-test iterations
        if (test_one(start + i)) return;
    }
#endif
}

// One round of the "while-while" traversal over the BVH4: the lane descends until it holds a leaf (or nothing), then the wave tests
// leaves together.  A step loads one 64-byte node, tests the four child boxes (per child: 3 v_perm_b32, 6 cvt, 6 fma, min3 / max3),
// sorts the hit children by entry distance (5 compare-exchanges), continues with the nearest and pushes the others, farthest first,
// together with their entry distances.
// TOP: nodes below st.n_top are read from the workgroup's LDS copy (ds_read_b128 through an explicit LDS pointer, as the stack): the
// per-lane 16-byte loads of a divergent walk cost the vector-memory pipe ~39 CU-cycles per wave-instruction even when every line hits
// L1, a random ds_read_b128 ~7 (scripts/ubench/gather_rate.hip); the upper levels are where every ray passes.
template <bool ANY, bool FAR = true, bool TOP = false>
MTS_DEV void walk_round(BvhWalk &w, const SceneView &sv, const WalkStack &st, uint32_t &tri_tests) {
    uint32_t cur = w.cur, sp = w.sp;
    const f3 inv = w.inv;
    constexpr float kInf = __builtin_inff();
    // next subtree from the stack whose entry distance is not beyond the closest hit
    auto pop_next = [&]() {
        cur = kNoNode;
        while (sp) {
            MTS_PROF(ANY, 8);                           // stack pops
            --sp;
            const StackEntry e = stack_pop(st, sp);
            if (__uint_as_float(e.y) <= w.best) { cur = e.x; break; }
        }
    };
    MTS_PROF(ANY, 0);                                   // rounds
    while (true) {
        const bool inner = (int32_t) cur >= 0 && cur != kNoNode;
        const uint64_t mi = __ballot(inner);
#if MTS_TRACE_PROF
        const uint64_t ml = __ballot((cur & kLeafFlag) && cur != kNoNode), mn = __ballot(cur == kNoNode);
#endif
        if (mi == 0ull) break;
        constexpr int kT = ANY ? MTS_WALK_T_ANY : MTS_WALK_T;
        if (kT > 0 && __popcll(mi) < kT && __ballot((int32_t) cur < 0) != 0ull) break;
        if (!inner) continue;
        MTS_PROF(ANY, 2);                               // node steps
        MTS_PROF_MASK(ANY, 16, ml);        // ... lanes waiting at a leaf meanwhile
        MTS_PROF_MASK(ANY, 18, mn);                     // ... lanes without a ray meanwhile
        const f3 oq = FAR ? walk_origin_q(sv, w.o) : w.noi;      // (unused in the near form)
        const f3 noi = FAR ? mk3(fminf(4.8e-7f * fabsf(oq.x * inv.x), 1.0e30f), fminf(4.8e-7f * fabsf(oq.y * inv.y), 1.0e30f),
                                 fminf(4.8e-7f * fabsf(oq.z * inv.z), 1.0e30f)) : w.noi;
        uint4 c0, c1, c2, c3;
        if (TOP && cur < st.n_top) {
            const LdsNodeWord *node = reinterpret_cast<LdsNodeWord *>((uint32_t) reinterpret_cast<uintptr_t>(st.top_nodes)) + 4u * cur;
            const node_u4 n0 = node[0], n1 = node[1], n2 = node[2], n3 = node[3];
            c0 = make_uint4(n0.x, n0.y, n0.z, n0.w); c1 = make_uint4(n1.x, n1.y, n1.z, n1.w);
            c2 = make_uint4(n2.x, n2.y, n2.z, n2.w); c3 = make_uint4(n3.x, n3.y, n3.z, n3.w);
        } else {
            const uint4 *node = sv.wnodes + 4u * cur;
            c0 = node[0]; c1 = node[1]; c2 = node[2]; c3 = node[3];
        }
        auto slab_n = [&](uint32_t q, float o1, float i1, float n1) -> float {
            return FAR ? fmaf((float) q - o1, i1, -n1) : fmaf((float) q, i1, n1);
        };
        auto slab_f = [&](uint32_t q, float o1, float i1, float n1) -> float {
            return FAR ? fmaf((float) q - o1, i1, n1) : fmaf((float) q, i1, n1);
        };
        // entry distance of the ray into a child box, +inf if it misses (an absent child has an inverted box)
#if MTS_NODE_F16
        // planes are fp16: near = low half, far = high half after the permute; t = fma(plane, inv, noi) in ONE v_fma_mix_f32 each
        // (FAR: plane - o_q by v_fma_mix_f32(plane, 1, -o_q), then the padded fma)
        auto mix_lo = [](uint32_t p, float a, float b) -> float { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(a), "v"(b)); return r; };
        auto mix_hi = [](uint32_t p, float a, float b) -> float { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(a), "v"(b)); return r; };
        auto child = [&](const uint4 &c) -> float {
            const uint32_t px = __builtin_amdgcn_perm(c.x, c.x, w.sel[0]), py = __builtin_amdgcn_perm(c.y, c.y, w.sel[1]),
                           pz = __builtin_amdgcn_perm(c.z, c.z, w.sel[2]);
            float nx, ny, nz, fx, fy, fz;
            if (FAR) {
                nx = fmaf(mix_lo(px, 1.0f, -oq.x), inv.x, -noi.x); ny = fmaf(mix_lo(py, 1.0f, -oq.y), inv.y, -noi.y); nz = fmaf(mix_lo(pz, 1.0f, -oq.z), inv.z, -noi.z);
                fx = fmaf(mix_hi(px, 1.0f, -oq.x), inv.x, noi.x); fy = fmaf(mix_hi(py, 1.0f, -oq.y), inv.y, noi.y); fz = fmaf(mix_hi(pz, 1.0f, -oq.z), inv.z, noi.z);
            } else {
                nx = mix_lo(px, inv.x, noi.x); ny = mix_lo(py, inv.y, noi.y); nz = mix_lo(pz, inv.z, noi.z);
                fx = mix_hi(px, inv.x, noi.x); fy = mix_hi(py, inv.y, noi.y); fz = mix_hi(pz, inv.z, noi.z);
            }
            const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, w.mint));
            const float tf = fminf(fminf(fx, fy), fminf(fz, w.best));
            return tn <= tf ? tn : kInf;
        };
#elif MTS_NODE_P15
        // planes are 0x8000 | q15: v_perm_b32 puts the two bytes under the constant's 0x47 and over a zero byte, which IS the float
        // 65536 + 2 q15 (exponent 2^16, the stored top bit is the exponent's lowest bit): one permute per plane orders near / far
        // AND converts; t = fma(plane, inv, noi) with the 65536 folded into o_q (walk_begin)
        constexpr uint32_t kP15Const = 0x47000000u, kP15Flip = 0x00020200u;
        const uint32_t fsx = w.sel[0] ^ kP15Flip, fsy = w.sel[1] ^ kP15Flip, fsz = w.sel[2] ^ kP15Flip;
        auto child = [&](const uint4 &c) -> float {
            const float nx = __uint_as_float(__builtin_amdgcn_perm(c.x, kP15Const, w.sel[0])), ny = __uint_as_float(__builtin_amdgcn_perm(c.y, kP15Const, w.sel[1])),
                        nz = __uint_as_float(__builtin_amdgcn_perm(c.z, kP15Const, w.sel[2]));
            const float fx = __uint_as_float(__builtin_amdgcn_perm(c.x, kP15Const, fsx)), fy = __uint_as_float(__builtin_amdgcn_perm(c.y, kP15Const, fsy)),
                        fz = __uint_as_float(__builtin_amdgcn_perm(c.z, kP15Const, fsz));
            const float tn = fmaxf(fmaxf(FAR ? fmaf(nx - oq.x, inv.x, -noi.x) : fmaf(nx, inv.x, noi.x), FAR ? fmaf(ny - oq.y, inv.y, -noi.y) : fmaf(ny, inv.y, noi.y)),
                                   fmaxf(FAR ? fmaf(nz - oq.z, inv.z, -noi.z) : fmaf(nz, inv.z, noi.z), w.mint));
            const float tf = fminf(fminf(FAR ? fmaf(fx - oq.x, inv.x, noi.x) : fmaf(fx, inv.x, noi.x), FAR ? fmaf(fy - oq.y, inv.y, noi.y) : fmaf(fy, inv.y, noi.y)),
                                   fminf(FAR ? fmaf(fz - oq.z, inv.z, noi.z) : fmaf(fz, inv.z, noi.z), w.best));
            return tn <= tf ? tn : kInf;
        };
#else
        auto child = [&](const uint4 &c) -> float {
            const uint32_t px = __builtin_amdgcn_perm(c.x, c.x, w.sel[0]), py = __builtin_amdgcn_perm(c.y, c.y, w.sel[1]),
                           pz = __builtin_amdgcn_perm(c.z, c.z, w.sel[2]);
            const float tn = fmaxf(fmaxf(slab_n(px & 0xffffu, oq.x, inv.x, noi.x), slab_n(py & 0xffffu, oq.y, inv.y, noi.y)),
                                   fmaxf(slab_n(pz & 0xffffu, oq.z, inv.z, noi.z), w.mint));
            const float tf = fminf(fminf(slab_f(px >> 16, oq.x, inv.x, noi.x), slab_f(py >> 16, oq.y, inv.y, noi.y)),
                                   fminf(slab_f(pz >> 16, oq.z, inv.z, noi.z), w.best));
            return tn <= tf ? tn : kInf;
        };
#endif
        float t0 = child(c0), t1 = child(c1), t2 = child(c2), t3 = child(c3);
        uint32_t r0 = c0.w, r1 = c1.w, r2 = c2.w, r3 = c3.w;
        // sorting network (0,1)(2,3)(0,2)(1,3)(1,2): ascending entry distance, misses (+inf) last
        auto cswap = [](float &ta, uint32_t &ra, float &tb, uint32_t &rb) {
            const bool s = tb < ta;
            const float tlo = s ? tb : ta, thi = s ? ta : tb;
            const uint32_t rlo = s ? rb : ra, rhi = s ? ra : rb;
            ta = tlo; tb = thi; ra = rlo; rb = rhi;
        };
#if MTS_ANY_SORT == 0
        // any-hit walks need no front-to-back order: the hit children are taken as they come (a quarter of the step's cost is the sort)
        if (!ANY)
#elif MTS_ANY_SORT == 1
        if (ANY) { cswap(t0, r0, t1, r1); cswap(t2, r2, t3, r3); cswap(t0, r0, t2, r2); }      // nearest first, the rest as they come
        else
#endif
#if MTS_CLOSEST_SORT == 3
        { cswap(t0, r0, t1, r1); cswap(t2, r2, t3, r3); cswap(t0, r0, t2, r2); }
#elif MTS_CLOSEST_SORT == 4
        { cswap(t0, r0, t1, r1); cswap(t2, r2, t3, r3); cswap(t0, r0, t2, r2); cswap(t1, r1, t3, r3); }
#else
        { cswap(t0, r0, t1, r1); cswap(t2, r2, t3, r3); cswap(t0, r0, t2, r2); cswap(t1, r1, t3, r3); cswap(t1, r1, t2, r2); }
#endif
        stack_push_if(st, sp, t3 < kInf, make_uint2(r3, __float_as_uint(t3)));
        stack_push_if(st, sp, t2 < kInf, make_uint2(r2, __float_as_uint(t2)));
        stack_push_if(st, sp, t1 < kInf, make_uint2(r1, __float_as_uint(t1)));
        if (t0 < kInf) cur = r0;
        else pop_next();
    }
    if (cur & kLeafFlag) {
        walk_leaf<ANY>(w, sv, cur, tri_tests);
        if (ANY && w.found) { w.cur = kNoNode; w.sp = 0; return; }
        if (cur == kNoNode) pop_next();
    }
    w.cur = cur; w.sp = sp;
}
#else
template <bool ANY, bool FAR = true, bool TOP = false>       // TOP: BVH4 only (ignored here)
MTS_DEV void walk_round(BvhWalk &w, const SceneView &sv, const WalkStack &st, uint32_t &tri_tests) {
    uint32_t cur = w.cur, sp = w.sp;
    const f3 o = w.o, inv = w.inv;
    // The descent loop ends for the whole wave as soon as fewer than MTS_WALK_T lanes are still inside the tree while others
    // wait at a leaf: a few long walks no longer hold the wave in a sparsely populated loop (0 = every lane reaches its leaf).
    while (true) {
        const bool inner = (int32_t) cur >= 0 && cur != kNoNode;
        const uint64_t mi = __ballot(inner);
        if (mi == 0ull) break;
        constexpr int kT = ANY ? MTS_WALK_T_ANY : MTS_WALK_T;
        if (kT > 0 && __popcll(mi) < kT && __ballot((int32_t) cur < 0) != 0ull) break;
        if (!inner) continue;
#if MTS_QNODES
        // FAR: (q - o_q) is only good to eps * |o_q| cells out there, so that form widens every slab interval by that much (in t)
        // instead -- an axis the ray is parallel to then stops culling, nothing is ever culled wrongly.  The pad is recomputed
        // per step (a wave walks in one form, chosen by its farthest origin, so every lane needs it and a register is dearer).
        const f3 oq = FAR ? walk_origin_q(sv, w.o) : w.noi;      // (unused in the near form)
        const f3 noi = FAR ? mk3(fminf(4.8e-7f * fabsf(oq.x * inv.x), 1.0e30f), fminf(4.8e-7f * fabsf(oq.y * inv.y), 1.0e30f),
                                 fminf(4.8e-7f * fabsf(oq.z * inv.z), 1.0e30f)) : w.noi;
        const uint4 a = sv.qnodes[2u * cur], bq = sv.qnodes[2u * cur + 1u];
        // near plane (low half after the permute) / far plane (high half).  FAR: noi holds the error pad e >= 0 of the axis
        auto slab_n = [&](uint32_t q, float o1, float i1, float n1) -> float {
            return FAR ? fmaf((float) q - o1, i1, -n1) : fmaf((float) q, i1, n1);
        };
        auto slab_f = [&](uint32_t q, float o1, float i1, float n1) -> float {
            return FAR ? fmaf((float) q - o1, i1, n1) : fmaf((float) q, i1, n1);
        };
        // per axis: one v_perm_b32 orders the two planes by the sign of the direction, so near = t(low half), far = t(high half)
        const uint32_t lx = __builtin_amdgcn_perm(a.x, a.x, w.sel[0]), ly = __builtin_amdgcn_perm(a.y, a.y, w.sel[1]),
                       lz = __builtin_amdgcn_perm(a.z, a.z, w.sel[2]), rx = __builtin_amdgcn_perm(a.w, a.w, w.sel[0]),
                       ry = __builtin_amdgcn_perm(bq.x, bq.x, w.sel[1]), rz = __builtin_amdgcn_perm(bq.y, bq.y, w.sel[2]);
        const float nearL = fmaxf(fmaxf(slab_n(lx & 0xffffu, oq.x, inv.x, noi.x), slab_n(ly & 0xffffu, oq.y, inv.y, noi.y)),
                                  fmaxf(slab_n(lz & 0xffffu, oq.z, inv.z, noi.z), w.mint));
        const float farL = fminf(fminf(slab_f(lx >> 16, oq.x, inv.x, noi.x), slab_f(ly >> 16, oq.y, inv.y, noi.y)),
                                 fminf(slab_f(lz >> 16, oq.z, inv.z, noi.z), w.best));
        const float nearR = fmaxf(fmaxf(slab_n(rx & 0xffffu, oq.x, inv.x, noi.x), slab_n(ry & 0xffffu, oq.y, inv.y, noi.y)),
                                  fmaxf(slab_n(rz & 0xffffu, oq.z, inv.z, noi.z), w.mint));
        const float farR = fminf(fminf(slab_f(rx >> 16, oq.x, inv.x, noi.x), slab_f(ry >> 16, oq.y, inv.y, noi.y)),
                                 fminf(slab_f(rz >> 16, oq.z, inv.z, noi.z), w.best));
        const bool hl = nearL <= farL, hr = nearR <= farR;
        const uint32_t cl = bq.z, cr = bq.w;
#else
        const float4 *p = sv.nodes + 4u * cur;
        const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        float ax = (q0.x - o.x) * inv.x, bx = (q0.w - o.x) * inv.x;
        float ay = (q0.y - o.y) * inv.y, by = (q1.x - o.y) * inv.y;
        float az = (q0.z - o.z) * inv.z, bz = (q1.y - o.z) * inv.z;
        float nearL = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), w.mint));
        float farL = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), w.best));
        ax = (q1.z - o.x) * inv.x; bx = (q2.y - o.x) * inv.x;
        ay = (q1.w - o.y) * inv.y; by = (q2.z - o.y) * inv.y;
        az = (q2.x - o.z) * inv.z; bz = (q2.w - o.z) * inv.z;
        float nearR = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), w.mint));
        float farR = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), w.best));
        const bool hl = nearL <= farL, hr = nearR <= farR;
        const uint32_t cl = __float_as_uint(q3.x), cr = __float_as_uint(q3.y);
#endif
        if (hl && hr) {
            const bool lf = nearL <= nearR;
            stack_push(st, sp, lf ? cr : cl);
            ++sp;
            cur = lf ? cl : cr;
        } else if (hl) {
            cur = cl;
        } else if (hr) {
            cur = cr;
        } else if (sp) {
            --sp;
            cur = stack_pop(st, sp);
        } else {
            cur = kNoNode;
        }
    }
    if (cur & kLeafFlag) {
        const uint32_t start = cur & kLeafStartMask, count = (cur >> kLeafCountShift) & 0xfu;
        for (uint32_t i = 0; i < count; ++i) {
            const float4 *p = sv.tris + 3u * (start + i);
            const float4 t0 = p[0], t1 = p[1], t2 = p[2];
            float u, v, t;
            ++tri_tests;
            if (tri_test(mk3(t0.x, t0.y, t0.z), mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, t2.x), o, w.d, w.mint, w.maxt, u, v, t)) {
                if (ANY) { w.found = true; w.cur = kNoNode; w.sp = 0; return; }
                const uint32_t prim = __float_as_uint(t2.y);
                if (!w.found || t < w.best || (t == w.best && prim > w.best_prim)) {
                    w.found = true; w.best = t; w.best_prim = prim;
                    w.hit.t = t; w.hit.prim = prim; w.hit.u = u; w.hit.v = v;
                }
            }
        }
        if (sp) {
            --sp;
            cur = stack_pop(st, sp);
        } else {
            cur = kNoNode;
        }
    }
    w.cur = cur; w.sp = sp;
}

#endif

template <bool ANY>
MTS_DEV bool traverse_bvh(const SceneView &sv, const LdsView &lds, f3 o, f3 d, float mint, float maxt,
                          Hit &hit, uint32_t &tri_tests) {
    BvhWalk w;
    walk_begin(w, sv, o, d, mint, maxt);
    const WalkStack st = { reinterpret_cast<StackEntry *>(lds.stack) + threadIdx.x, log2_stride(lds.stride), lds.spill ? lds.stack_lds_depth : 0xffffffffu,
                           lds.spill, lds.spill_stride, nullptr, 0u };      // k_bounce: whole stack in LDS; k_finish: short LDS part + spill
    while (w.cur != kNoNode) {
        if (w.far) walk_round<ANY, true>(w, sv, st, tri_tests);
        else walk_round<ANY, false>(w, sv, st, tri_tests);
    }
    if (!ANY && w.found) { hit.t = w.best; hit.prim = w.best_prim; hit.u = w.hit.u; hit.v = w.hit.v; }
    return w.found;
}

// Flat scenes: a wave-uniform loop over every primitive record in LDS (broadcast reads, no stack,
// no divergence).  Primitive order + "t <= best" reproduces the brute-force loop of
// ray_intersect_naive (kdtree.h:2303-2328) literally: later primitives win ties.
typedef float v2f __attribute__((ext_vector_type(2)));
MTS_DEV v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
MTS_DEV v2f splat(float x) { v2f r = { x, x }; return r; }

// Correctly rounded 1/x for normal-range x: v_rcp_f32 (<= 1 ulp) followed by two Newton-Raphson steps with
// exact fma residuals.  x == 0 gives NaN instead of inf; every comparison of the triangle test still fails.
MTS_DEV v2f rcp_nr2(v2f x) {
    v2f r = { __builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y) };
    const v2f one = { 1.0f, 1.0f };
    v2f e = vfma(-x, r, one);
    r = vfma(e, r, r);
    e = vfma(-x, r, one);
    return vfma(e, r, r);
}

// Barycentrics of primitive `prim` for the ray (o, d), by the very operations of the packed loops below on that element (the packed
// instructions are element-wise IEEE operations, so the bits are the same).  With MTS_FLAT_LATE_UV the closest-hit loops carry only
// (t, primitive) through their 36 tests -- two conditional moves per triangle less -- and the winner's (u, v) are formed once here.
MTS_DEV void flat_hit_uv(const LdsView &lds, f3 o, f3 d, uint32_t prim, float &u, float &v) {
    const float4 *rec = lds.pairs + 5u * (prim >> 1);
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4];
    const bool hi = (prim & 1u) != 0u;
    const f3 p0 = hi ? mk3(q0.y, q0.w, q1.y) : mk3(q0.x, q0.z, q1.x);
    const f3 e1 = hi ? mk3(q1.w, q2.y, q2.w) : mk3(q1.z, q2.x, q2.z);
    const f3 e2 = hi ? mk3(q3.y, q3.w, q4.y) : mk3(q3.x, q3.z, q4.x);
    const float pvx = fmaf(d.y, e2.z, -(d.z * e2.y)), pvy = fmaf(d.z, e2.x, -(d.x * e2.z)), pvz = fmaf(d.x, e2.y, -(d.y * e2.x));
    const float det = fmaf(e1.z, pvz, fmaf(e1.y, pvy, e1.x * pvx));
    float r = __builtin_amdgcn_rcpf(det);
    float e = fmaf(-det, r, 1.0f); r = fmaf(e, r, r);
    e = fmaf(-det, r, 1.0f); r = fmaf(e, r, r);
    const float tx = o.x - p0.x, ty = o.y - p0.y, tz = o.z - p0.z;
    u = fmaf(tz, pvz, fmaf(ty, pvy, tx * pvx)) * r;
    const float qx = fmaf(ty, e1.z, -(tz * e1.y)), qy = fmaf(tz, e1.x, -(tx * e1.z)), qz = fmaf(tx, e1.y, -(ty * e1.x));
    v = fmaf(d.z, qz, fmaf(d.y, qy, d.x * qx)) * r;
}
#ifndef MTS_FLAT_LATE_UV
#define MTS_FLAT_LATE_UV 1
#endif

// Flat scenes: a wave-uniform loop over every primitive, two per iteration on the packed-fp32 pipe
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), operands broadcast from LDS (no stack, no divergence).
// Primitive order + "t <= best" reproduces the brute-force loop of ray_intersect_naive
// (kdtree.h:2303-2328) literally: later primitives win ties.  Per element the arithmetic is exactly
// Mesh::ray_intersect_triangle (mesh.h:195-221).
template <bool ANY>
MTS_DEV bool traverse_flat(const SceneView &sv, const LdsView &lds, f3 o, f3 d, float mint, float maxt,
                           Hit &hit, uint32_t &tri_tests) {
    float best = maxt, bu = 0.0f, bv = 0.0f;
    uint32_t best_prim = kNoPrim;
    bool any = false;
    const uint32_t np = sv.n_pairs;
    tri_tests += sv.n_prims;
    const v2f ox = splat(o.x), oy = splat(o.y), oz = splat(o.z), dx = splat(d.x), dy = splat(d.y), dz = splat(d.z);
    // software pipeline: the next pair is fetched while the current one is tested (the staged array ends
    // with an all-zero pair, which can never be hit: det == 0)
    const float4 *rec = lds.pairs;
#if MTS_FLAT_PREFETCH
    float4 n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3], n4 = rec[4];
#endif
#pragma unroll MTS_FLAT_UNROLL
    for (uint32_t k = 0; k < np; ++k) {
#if MTS_FLAT_PREFETCH
        const float4 q0 = n0, q1 = n1, q2 = n2, q3 = n3, q4 = n4;
        rec += 5;
        n0 = rec[0]; n1 = rec[1]; n2 = rec[2]; n3 = rec[3]; n4 = rec[4];
#else
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4];
        rec += 5;
#endif
        const v2f p0x = { q0.x, q0.y }, p0y = { q0.z, q0.w }, p0z = { q1.x, q1.y };
        const v2f e1x = { q1.z, q1.w }, e1y = { q2.x, q2.y }, e1z = { q2.z, q2.w };
        const v2f e2x = { q3.x, q3.y }, e2y = { q3.z, q3.w }, e2z = { q4.x, q4.y };
        // pvec = cross(d, e2)
        const v2f pvx = vfma(dy, e2z, -(dz * e2y)), pvy = vfma(dz, e2x, -(dx * e2z)), pvz = vfma(dx, e2y, -(dy * e2x));
        const v2f det = vfma(e1z, pvz, vfma(e1y, pvy, e1x * pvx));
        const v2f inv = rcp_nr2(det);
        const v2f tx = ox - p0x, ty = oy - p0y, tz = oz - p0z;
        const v2f u = vfma(tz, pvz, vfma(ty, pvy, tx * pvx)) * inv;
        // qvec = cross(tvec, e1)
        const v2f qx = vfma(ty, e1z, -(tz * e1y)), qy = vfma(tz, e1x, -(tx * e1z)), qz = vfma(tx, e1y, -(ty * e1x));
        const v2f v = vfma(dz, qz, vfma(dy, qy, dx * qx)) * inv;
        const v2f t = vfma(e2z, qz, vfma(e2y, qy, e2x * qx)) * inv;
        const v2f uv = u + v;
        // u <= 1 is implied by v >= 0 && u + v <= 1 (rounded addition is monotone); NaNs fail u >= 0 or u + v <= 1
        if (ANY) {
            bool ok_a = (u.x >= 0.0f) && (v.x >= 0.0f) && (uv.x <= 1.0f) && (t.x >= mint) && (t.x <= maxt);
            bool ok_b = (u.y >= 0.0f) && (v.y >= 0.0f) && (uv.y <= 1.0f) && (t.y >= mint) && (t.y <= maxt);
            any = any || ok_a || ok_b;
        } else {
            bool ok_a = (u.x >= 0.0f) && (v.x >= 0.0f) && (uv.x <= 1.0f) && (t.x >= mint) && (t.x <= best);
            best = ok_a ? t.x : best; best_prim = ok_a ? 2u * k : best_prim;
            if (!MTS_FLAT_LATE_UV) { bu = ok_a ? u.x : bu; bv = ok_a ? v.x : bv; }
            bool ok_b = (u.y >= 0.0f) && (v.y >= 0.0f) && (uv.y <= 1.0f) && (t.y >= mint) && (t.y <= best);
            best = ok_b ? t.y : best; best_prim = ok_b ? 2u * k + 1u : best_prim;
            if (!MTS_FLAT_LATE_UV) { bu = ok_b ? u.y : bu; bv = ok_b ? v.y : bv; }
        }
    }
    if (ANY) return any;
    if (MTS_FLAT_LATE_UV && best_prim != kNoPrim) flat_hit_uv(lds, o, d, best_prim, bu, bv);
    hit.t = best; hit.prim = best_prim; hit.u = bu; hit.v = bv;
    return best_prim != kNoPrim;
}

// Closest hit of a COHERENT wave on a flat scene (the camera rays of one or two pixels: all lanes at depth 1): the pairs are walked
// cluster by cluster -- a cluster = the consecutive pairs of one shape with their padded bounding box -- and a cluster that no lane's
// ray reaches before its closest hit so far is skipped by the whole wave (a wave-uniform branch: no divergence, nothing to compact).
// The boxes only cull: primitive order, arithmetic and tie rule of the triangle tests are those of traverse_flat, so the hit is
// bit-identical.  A wall pixel of the Cornell box tests 1 pair instead of 18.  Incoherent waves (any lane at depth > 1) keep the plain
// loop: there some lane reaches nearly every box and the 8 box tests (~20 VALU each) would only be added work.
#ifndef MTS_FLAT_CULL
#define MTS_FLAT_CULL 1
#endif
MTS_DEV bool traverse_flat_clustered(const SceneView &sv, const LdsView &lds, f3 o, f3 d, float mint, float maxt, Hit &hit, uint32_t &tri_tests) {
    float best = maxt, bu = 0.0f, bv = 0.0f;
    uint32_t best_prim = kNoPrim;
#ifndef MTS_CULL_STATS
#define MTS_CULL_STATS 0                                     // 1 (diagnostic builds): count the triangles really tested
#endif
    if (!MTS_CULL_STATS) tri_tests += sv.n_prims;            // nominal count, as the plain loop (the statistics do not depend on the schedule)
    const v2f ox = splat(o.x), oy = splat(o.y), oz = splat(o.z), dx = splat(d.x), dy = splat(d.y), dz = splat(d.z);
    const f3 inv = mk3(clamp_inv(d.x), clamp_inv(d.y), clamp_inv(d.z));
    const float4 *rec = lds.pairs;
    uint32_t k = 0u;
    for (uint32_t c = 0; c < sv.n_clusters; ++c) {
        const float4 lo = lds.clusters[2u * c], hi = lds.clusters[2u * c + 1u];
        const uint32_t n = __float_as_uint(lo.w);
        // slab test against the padded box; the interval is [mint, best]
        const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
        const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
        const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
        const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), mint));
        const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), best));
        if (__ballot(tn <= tf) == 0ull) { rec += 5u * n; k += n; continue; }
        if (MTS_CULL_STATS) tri_tests += 2u * n;
        for (uint32_t i = 0; i < n; ++i, ++k) {
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4];
            rec += 5;
            const v2f p0x = { q0.x, q0.y }, p0y = { q0.z, q0.w }, p0z = { q1.x, q1.y };
            const v2f e1x = { q1.z, q1.w }, e1y = { q2.x, q2.y }, e1z = { q2.z, q2.w };
            const v2f e2x = { q3.x, q3.y }, e2y = { q3.z, q3.w }, e2z = { q4.x, q4.y };
            const v2f pvx = vfma(dy, e2z, -(dz * e2y)), pvy = vfma(dz, e2x, -(dx * e2z)), pvz = vfma(dx, e2y, -(dy * e2x));
            const v2f det = vfma(e1z, pvz, vfma(e1y, pvy, e1x * pvx));
            const v2f ivd = rcp_nr2(det);
            const v2f tx = ox - p0x, ty = oy - p0y, tz = oz - p0z;
            const v2f u = vfma(tz, pvz, vfma(ty, pvy, tx * pvx)) * ivd;
            const v2f qx = vfma(ty, e1z, -(tz * e1y)), qy = vfma(tz, e1x, -(tx * e1z)), qz = vfma(tx, e1y, -(ty * e1x));
            const v2f v = vfma(dz, qz, vfma(dy, qy, dx * qx)) * ivd;
            const v2f t = vfma(e2z, qz, vfma(e2y, qy, e2x * qx)) * ivd;
            const v2f uv = u + v;
            bool ok_a = (u.x >= 0.0f) && (v.x >= 0.0f) && (uv.x <= 1.0f) && (t.x >= mint) && (t.x <= best);
            best = ok_a ? t.x : best; best_prim = ok_a ? 2u * k : best_prim;
            if (!MTS_FLAT_LATE_UV) { bu = ok_a ? u.x : bu; bv = ok_a ? v.x : bv; }
            bool ok_b = (u.y >= 0.0f) && (v.y >= 0.0f) && (uv.y <= 1.0f) && (t.y >= mint) && (t.y <= best);
            best = ok_b ? t.y : best; best_prim = ok_b ? 2u * k + 1u : best_prim;
            if (!MTS_FLAT_LATE_UV) { bu = ok_b ? u.y : bu; bv = ok_b ? v.y : bv; }
        }
    }
    if (MTS_FLAT_LATE_UV && best_prim != kNoPrim) flat_hit_uv(lds, o, d, best_prim, bu, bv);
    hit.t = best; hit.prim = best_prim; hit.u = bu; hit.v = bv;
    return best_prim != kNoPrim;
}

// Any-hit counterpart for the shadow rays of such a wave (they leave a pixel-sized patch of a surface towards one emitter): clusters no
// active lane's segment reaches are skipped; a lane that has found an occluder no longer votes.
MTS_DEV bool traverse_flat_clustered_any(const SceneView &sv, const LdsView &lds, f3 o, f3 d, float mint, float maxt, uint32_t &tri_tests) {
    bool any = false;
    tri_tests += sv.n_prims;                                 // nominal count, as the plain loop
    const v2f ox = splat(o.x), oy = splat(o.y), oz = splat(o.z), dx = splat(d.x), dy = splat(d.y), dz = splat(d.z);
    const f3 inv = mk3(clamp_inv(d.x), clamp_inv(d.y), clamp_inv(d.z));
    const float4 *rec = lds.pairs;
    for (uint32_t c = 0; c < sv.n_clusters; ++c) {
        const float4 lo = lds.clusters[2u * c], hi = lds.clusters[2u * c + 1u];
        const uint32_t n = __float_as_uint(lo.w);
        const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
        const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
        const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
        const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), mint));
        const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), maxt));
        if (__ballot(!any && tn <= tf) == 0ull) { rec += 5u * n; continue; }
        for (uint32_t i = 0; i < n; ++i) {
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4];
            rec += 5;
            const v2f p0x = { q0.x, q0.y }, p0y = { q0.z, q0.w }, p0z = { q1.x, q1.y };
            const v2f e1x = { q1.z, q1.w }, e1y = { q2.x, q2.y }, e1z = { q2.z, q2.w };
            const v2f e2x = { q3.x, q3.y }, e2y = { q3.z, q3.w }, e2z = { q4.x, q4.y };
            const v2f pvx = vfma(dy, e2z, -(dz * e2y)), pvy = vfma(dz, e2x, -(dx * e2z)), pvz = vfma(dx, e2y, -(dy * e2x));
            const v2f det = vfma(e1z, pvz, vfma(e1y, pvy, e1x * pvx));
            const v2f ivd = rcp_nr2(det);
            const v2f tx = ox - p0x, ty = oy - p0y, tz = oz - p0z;
            const v2f u = vfma(tz, pvz, vfma(ty, pvy, tx * pvx)) * ivd;
            const v2f qx = vfma(ty, e1z, -(tz * e1y)), qy = vfma(tz, e1x, -(tx * e1z)), qz = vfma(tx, e1y, -(ty * e1x));
            const v2f v = vfma(dz, qz, vfma(dy, qy, dx * qx)) * ivd;
            const v2f t = vfma(e2z, qz, vfma(e2y, qy, e2x * qx)) * ivd;
            const v2f uv = u + v;
            const bool ok_a = (u.x >= 0.0f) && (v.x >= 0.0f) && (uv.x <= 1.0f) && (t.x >= mint) && (t.x <= maxt);
            const bool ok_b = (u.y >= 0.0f) && (v.y >= 0.0f) && (uv.y <= 1.0f) && (t.y >= mint) && (t.y <= maxt);
            any = any || ok_a || ok_b;
        }
    }
    return any;
}

// `coherent` (wave-uniform): the active lanes carry the camera rays of one or two pixels -> cluster culling (flat scenes, closest hit)
template <bool FLAT, bool ANY>
MTS_DEV bool traverse(const SceneView &sv, const LdsView &lds, f3 o, f3 d, float mint, float maxt, Hit &hit, uint32_t &tri_tests,
                      bool coherent = false) {
    if (FLAT && !ANY && MTS_FLAT_CULL && coherent && sv.n_clusters > 1u) return traverse_flat_clustered(sv, lds, o, d, mint, maxt, hit, tri_tests);
    if (FLAT) return traverse_flat<ANY>(sv, lds, o, d, mint, maxt, hit, tri_tests);
    return traverse_bvh<ANY>(sv, lds, o, d, mint, maxt, hit, tri_tests);
}

// Brute force over every triangle slot from global memory (ray_intersect_naive, kdtree.h:2303-2328).
template <bool ANY>
MTS_DEV bool traverse_naive(const SceneView &sv, f3 o, f3 d, float mint, float maxt, Hit &hit) {
    bool found = false; float best = maxt; uint32_t best_prim = kNoPrim;
    for (uint32_t s = 0; s < sv.n_slots; ++s) {
        const float4 *p = sv.tris + 3u * s;
        const float4 t0 = p[0], t1 = p[1], t2 = p[2];
        float u, v, t;
        if (tri_test(mk3(t0.x, t0.y, t0.z), mk3(t0.w, t1.x, t1.y), mk3(t1.z, t1.w, t2.x), o, d, mint, maxt, u, v, t)) {
            if (ANY) return true;
            uint32_t prim = __float_as_uint(t2.y);
            if (!found || t < best || (t == best && prim > best_prim)) {
                found = true; best = t; best_prim = prim;
                hit.t = t; hit.prim = prim; hit.u = u; hit.v = v;
            }
        }
    }
    return found;
}

// ---------------------------------------------------------------------------
// Scene data access: LDS for flat scenes, global memory (L2) otherwise.
template <bool FLAT> struct Geo {
    const SceneView &sv; const LdsView &lds;
    MTS_DEV void tri_positions(uint32_t prim, f3 &p0, f3 &p1, f3 &p2) const {
        if (FLAT) {
            const float4 r0 = lds.flat[4u * prim], r2 = lds.flat[4u * prim + 2u], r3 = lds.flat[4u * prim + 3u];
            p0 = mk3(r0.x, r0.y, r0.z); p1 = mk3(r2.y, r2.z, r2.w); p2 = mk3(r3.x, r3.y, r3.z);
        } else {
            const float *tp = sv.tri_pos + 9u * prim;
            p0 = mk3(tp[0], tp[1], tp[2]); p1 = mk3(tp[3], tp[4], tp[5]); p2 = mk3(tp[6], tp[7], tp[8]);
        }
    }
    MTS_DEV uint32_t prim_shape(uint32_t prim) const {
        return FLAT ? __float_as_uint(lds.flat[4u * prim + 3u].w) : sv.prim_shape[prim];
    }
    MTS_DEV DevShape shape(uint32_t i) const { return FLAT ? lds.shapes[i] : sv.shapes[i]; }
    MTS_DEV DevBsdf bsdf(uint32_t i) const { return FLAT ? lds.bsdfs[i] : sv.bsdfs[i]; }
    MTS_DEV DevEmitter emitter(uint32_t i) const { return FLAT ? lds.emitters[i] : sv.emitters[i]; }
    MTS_DEV float pmf(uint32_t i) const { return FLAT ? lds.pmf[i] : sv.area_pmf[i]; }
    MTS_DEV float cdf(uint32_t i) const { return FLAT ? lds.cdf[i] : sv.area_cdf[i]; }
};

struct SurfaceInteraction {
    f3 p, n; f2 uv; Frame sh; f3 dp_du, dp_dv, wi; uint32_t shape; DevShape shape_rec;
};

template <bool FLAT>
MTS_DEV void fill_si(const Geo<FLAT> &g, f3 ray_d, uint32_t prim, float b1, float b2, SurfaceInteraction &si) {
    const SceneView &sv = g.sv;
    f3 p0, p1, p2;
    g.tri_positions(prim, p0, p1, p2);
    float b0 = 1.0f - b1 - b2;
    f3 dp0 = p1 - p0, dp1 = p2 - p0;
    si.p = (p0 * b0 + p1 * b1) + p2 * b2;
    f3 n = normalize(cross(dp0, dp1));
    si.n = n;
    uint32_t shape = g.prim_shape(prim);
    si.shape = shape;
    si.shape_rec = g.shape(shape);
    uint32_t flags = si.shape_rec.flags;
    f3 dp_du, dp_dv;
    coordinate_system(n, dp_du, dp_dv);
    si.uv.x = b1; si.uv.y = b2;
    if (flags & kShapeHasUV) {
        const float *tu = sv.tri_uv + 6u * prim;
        f2 uv0 = { tu[0], tu[1] }, uv1 = { tu[2], tu[3] }, uv2 = { tu[4], tu[5] };
        si.uv.x = (uv0.x * b0 + uv1.x * b1) + uv2.x * b2;
        si.uv.y = (uv0.y * b0 + uv1.y * b1) + uv2.y * b2;
        f2 duv0 = { uv1.x - uv0.x, uv1.y - uv0.y }, duv1 = { uv2.x - uv0.x, uv2.y - uv0.y };
        float det = fmaf(duv0.x, duv1.y, -(duv0.y * duv1.x));
        float inv_det = rcp(det);
        if (det != 0.0f) {
            dp_du = mk3(fmaf(duv1.y, dp0.x, -(duv0.y * dp1.x)) * inv_det,
                        fmaf(duv1.y, dp0.y, -(duv0.y * dp1.y)) * inv_det,
                        fmaf(duv1.y, dp0.z, -(duv0.y * dp1.z)) * inv_det);
            dp_dv = mk3(fmaf(-duv1.x, dp0.x, duv0.x * dp1.x) * inv_det,
                        fmaf(-duv1.x, dp0.y, duv0.x * dp1.y) * inv_det,
                        fmaf(-duv1.x, dp0.z, duv0.x * dp1.z) * inv_det);
        }
    }
    if (flags & kShapeHasNormals) {
        const float *tn = sv.tri_nrm + 9u * prim;
        f3 n0 = mk3(tn[0], tn[1], tn[2]), n1 = mk3(tn[3], tn[4], tn[5]), n2 = mk3(tn[6], tn[7], tn[8]);
        n = normalize((n0 * b0 + n1 * b1) + n2 * b2);
    }
    si.sh.n = n;
    si.dp_du = dp_du; si.dp_dv = dp_dv;
    float dd = dot(n, dp_du);
    si.sh.s = normalize(mk3(fmaf(-n.x, dd, dp_du.x), fmaf(-n.y, dd, dp_du.y), fmaf(-n.z, dd, dp_du.z)));
    si.sh.t = cross(n, si.sh.s);
    si.wi = to_local(si.sh, -ray_d);
}

// ---------------------------------------------------------------------------
struct DirectionSample { f3 p, n, d; float dist, pdf; uint32_t emitter; f2 uv; bool delta; float falloff; };      // uv: texture coordinates of an envmap sample; delta emitters: spec = (L * falloff) * r1

// Scene::sample_emitter_direction without the visibility test.  The emitted spectrum is returned in factored
// form: spec = (radiance * r1) * r2 with r1 = 1/pdf (0 when the sample is masked) and r2 = emitter count.
// DELTA = false compiles everything but area lights out (the diffuse / area-light fast path): environment and delta emitters.
template <bool FLAT, bool DELTA = true>
MTS_DEV void sample_emitter_direction(const Geo<FLAT> &g, f3 ref_p, f2 sample, DirectionSample &ds, float &r1, float &r2) {
    const SceneView &sv = g.sv;
    ds.pdf = 0.0f; ds.dist = 0.0f; ds.emitter = 0;
    ds.p = ds.n = ds.d = mk3(0, 0, 0); ds.uv.x = ds.uv.y = 0.0f; ds.delta = false; ds.falloff = 1.0f;
    r1 = 0.0f; r2 = 1.0f;
    if (sv.n_emitters == 0) return;
    uint32_t index = 0;
    float emitter_pdf = 1.0f;
    if (sv.n_emitters > 1) {
        float nf = (float) sv.n_emitters;
        emitter_pdf = 1.0f / nf;
        uint32_t idx = (uint32_t) (sample.x * nf);
        index = min(idx, sv.n_emitters - 1u);
        sample.x = (sample.x - (float) index * emitter_pdf) * nf;
    }
    const DevEmitter e = g.emitter(index);
    if (DELTA && e.pad0 == kEmitterEnvmap) {
        // EnvironmentMapEmitter::sample_direction (envmap.cpp:154-190); r1 = 1 / pdf, the radiance is looked up by the caller
        f3 d; f2 uv; float pdf;
        envmap_sample(*sv.envmap, sample, d, pdf, uv);
        ds.dist = 2.0f * e.radius;
        ds.p = ref_p + d * ds.dist;
        ds.n = -d; ds.d = d; ds.pdf = pdf; ds.emitter = index; ds.uv = uv;
        r1 = rcp(pdf);
        if (sv.n_emitters > 1) { ds.pdf *= emitter_pdf; r2 = rcp(emitter_pdf); }
        return;
    }
    if (DELTA && e.pad0 >= kEmitterPoint) {
        // PointLight / SpotLight / DirectionalEmitter::sample_direction (point.cpp:76-101, spot.cpp:129-151,
        // directional.cpp:104-129): pdf = 1, delta; spec = (L * falloff) * r1 with r1 = 1 / dist^2 (1 for `directional`)
        ds.pdf = 1.0f; ds.delta = true; ds.emitter = index;
        if (e.pad0 == kEmitterDirectional) {
            const f3 dir = mk3(e.aux[0], e.aux[1], e.aux[2]);
            ds.dist = 2.0f * e.radius;
            ds.p = ref_p - dir * ds.dist;
            ds.n = dir; ds.d = -dir;
            r1 = 1.0f;
        } else {
            ds.p = mk3(e.cx, e.cy, e.cz);
            ds.d = ds.p - ref_p;
            ds.dist = sqrtf(sqnorm(ds.d));
            const float inv_dist = rcp(ds.dist);
            ds.d = ds.d * inv_dist;
            if (e.pad0 == kEmitterSpot) {                      // falloff_curve (spot.cpp:95-113)
                const float cos_theta = normalize(mat3_apply(e.aux, -ds.d)).z;
                if (!(cos_theta >= e.aux[11])) ds.falloff = (e.aux[9] - lm_acos(cos_theta)) * e.aux[12];
                if (cos_theta <= e.aux[10]) ds.falloff = 0.0f;
            }
            r1 = inv_dist * inv_dist;
        }
        if (sv.n_emitters > 1) { ds.pdf *= emitter_pdf; r2 = rcp(emitter_pdf); }
        return;
    }
    if (DELTA && e.pad0 == kEmitterConstant) {
        // ConstantBackgroundEmitter::sample_direction (constant.cpp:82-107), square_to_uniform_sphere (warp.h:262-267)
        const float z = fmaf(-2.0f, sample.y, 1.0f), r = safe_sqrt(fmaf(-z, z, 1.0f));
        const float ang = 2.0f * kPi * sample.x;
        const f3 d = mk3(r * lm_cos(ang), r * lm_sin(ang), z);
        ds.dist = 2.0f * e.radius;
        ds.p = ref_p + d * ds.dist;
        ds.n = -d; ds.d = d;
        ds.pdf = kInvFourPi;
        ds.emitter = index;
        r1 = rcp(ds.pdf);
        if (sv.n_emitters > 1) { ds.pdf *= emitter_pdf; r2 = rcp(emitter_pdf); }
        return;
    }
    // Mesh::sample_position: DiscreteDistribution::sample_reuse (distr_1d.h:144-203)
    uint32_t f;
    {
        float scaled = sample.y * e.area_sum;
        uint32_t start = e.valid_lo, end = e.valid_hi;
        while (start < end) {
            uint32_t middle = (start + end) >> 1;
            if (g.cdf(e.first_prim + middle) < scaled) { start = middle + 1; if (start > end) start = end; }
            else end = middle;
        }
        float p = g.pmf(e.first_prim + start) * e.area_norm;
        float c = start > 0 ? g.cdf(e.first_prim + start - 1) * e.area_norm : 0.0f;
        sample.y = (sample.y - c) / p;
        f = start;
    }
    uint32_t prim = e.first_prim + f;
    f3 p0, p1, p2;
    g.tri_positions(prim, p0, p1, p2);
    f3 e0 = p1 - p0, e1 = p2 - p0;
    f2 b = square_to_uniform_triangle(sample);
    ds.p = (p0 + e0 * b.x) + e1 * b.y;
    ds.pdf = e.area_norm;
    if (g.shape(e.shape).flags & kShapeHasNormals) {
        const float *tn = sv.tri_nrm + 9u * prim;
        f3 n0 = mk3(tn[0], tn[1], tn[2]), n1 = mk3(tn[3], tn[4], tn[5]), n2 = mk3(tn[6], tn[7], tn[8]);
        float b0 = 1.0f - b.x - b.y;
        ds.n = normalize((n0 * b0 + n1 * b.x) + n2 * b.y);
    } else {
        ds.n = normalize(cross(e0, e1));
    }
    // Shape::sample_direction
    ds.d = ds.p - ref_p;
    float dist_squared = sqnorm(ds.d);
    ds.dist = sqrtf(dist_squared);
    ds.d = div_s(ds.d, ds.dist);
    float dp = fabsf(dot(ds.d, ds.n));
    ds.pdf *= (dp != 0.0f) ? dist_squared / dp : 0.0f;
    ds.emitter = index;
    // AreaLight::sample_direction
    bool active = (dot(ds.d, ds.n) < 0.0f) && (ds.pdf != 0.0f);
    if (active) r1 = rcp(ds.pdf);
    if (sv.n_emitters > 1) {
        ds.pdf *= emitter_pdf;
        r2 = rcp(emitter_pdf);
    }
}
// RGB form: spec = radiance / pdf (masked)
template <bool FLAT, bool DELTA = true>
MTS_DEV void sample_emitter_direction(const Geo<FLAT> &g, f3 ref_p, f2 sample, DirectionSample &ds, f3 &spec) {
    float r1, r2;
    sample_emitter_direction<FLAT, DELTA>(g, ref_p, sample, ds, r1, r2);
    spec = mk3(0, 0, 0);
    if (g.sv.n_emitters == 0) return;
    const DevEmitter e = g.emitter(ds.emitter);
    f3 rad = mk3(e.r, e.g, e.b);
    if (DELTA && e.pad0 == kEmitterEnvmap) rad = envmap_lookup(*g.sv.envmap, ds.uv.x, ds.uv.y);      // eval_spectrum at the sampled (u, v)
    if (DELTA && ds.delta) rad = mk3(rad.x * ds.falloff, rad.y * ds.falloff, rad.z * ds.falloff);
    spec = mk3(rad.x * r1, rad.y * r1, rad.z * r1);
    if (g.sv.n_emitters > 1) spec = spec * r2;
}

// pdf_direction of the environment emitter for world direction d (constant.cpp:109-114, envmap.cpp:192-208) + Scene (scene.cpp:191-206)
MTS_DEV float pdf_environment(const SceneView &sv, const DevEmitter &e, f3 d) {
    float pdf = e.pad0 == kEmitterEnvmap ? envmap_pdf(*sv.envmap, d) : kInvFourPi;
    if (sv.n_emitters > 1) pdf *= 1.0f / (float) sv.n_emitters;
    return pdf;
}
// radiance an escaped ray travelling along d picks up from the environment emitter (constant.cpp:53-57, envmap.cpp:132-146)
MTS_DEV f3 environment_radiance(const SceneView &sv, const DevEmitter &e, f3 d) {
    return e.pad0 == kEmitterEnvmap ? envmap_eval(*sv.envmap, d) : mk3(e.r, e.g, e.b);
}
MTS_DEV float pdf_emitter_direction(uint32_t n_emitters, float area_norm, f3 d, f3 n, float dist) {
    float pdf = 0.0f;
    if (dot(d, n) < 0.0f) {
        pdf = area_norm;
        float dp = fabsf(dot(d, n));
        pdf *= (dp != 0.0f) ? (dist * dist) / dp : 0.0f;
    }
    if (n_emitters > 1) pdf *= 1.0f / (float) n_emitters;
    return pdf;
}

// Diffuse reflectance at a surface interaction: the constant `srgb` colour (src/spectra/srgb.cpp:27-52) or
// BitmapTextureImpl::interpolate (src/textures/bitmap.cpp:250-293, identity to_uv).  `texel` / `w1` return the bilinear
// footprint (index of v00, weights towards +x / +y) for the adjoint; texel = kNoPrim for constant reflectance.
MTS_DEV f3 eval_reflectance(const SceneView &sv, const DevBsdf &b, f2 uv, uint32_t &texel, f2 &w1) {
    texel = kNoPrim; w1.x = w1.y = 0.0f;
    if (b.texture < 0) return mk3(b.r, b.g, b.b);
    const DevTexture t = sv.textures[b.texture];
    {   // m_transform.transform_affine(si.uv) (bitmap.cpp:254, checkerboard.cpp:49)
        const float u2 = fmaf(t.uvm[0], uv.x, fmaf(t.uvm[1], uv.y, t.uvm[2])), v2 = fmaf(t.uvm[3], uv.x, fmaf(t.uvm[4], uv.y, t.uvm[5]));
        uv.x = u2; uv.y = v2;
    }
    if (t.kind == 1u) {                                      // checkerboard.cpp:46-63
        const bool mx = (uv.x - floorf(uv.x)) > 0.5f, my = (uv.y - floorf(uv.y)) > 0.5f;
        return mx == my ? mk3(t.c0[0], t.c0[1], t.c0[2]) : mk3(t.c1[0], t.c1[1], t.c1[2]);
    }
    float ux = uv.x - floorf(uv.x), uy = uv.y - floorf(uv.y);
    ux *= (float) (uint32_t) (t.w - 1); uy *= (float) (uint32_t) (t.h - 1);
    uint32_t px = min((uint32_t) ux, (uint32_t) (t.w - 2)), py = min((uint32_t) uy, (uint32_t) (t.h - 2));
    w1.x = ux - (float) px; w1.y = uy - (float) py;
    const float w0x = 1.0f - w1.x, w0y = 1.0f - w1.y;
    texel = px + py * (uint32_t) t.w;
    const float *v00 = t.data + 3u * (size_t) texel, *v01 = v00 + 3u * (size_t) t.w;
    f3 r;
    { float v0 = fmaf(w0x, v00[0], w1.x * v00[3]), v1 = fmaf(w0x, v01[0], w1.x * v01[3]); r.x = fmaf(w0y, v0, w1.y * v1); }
    { float v0 = fmaf(w0x, v00[1], w1.x * v00[4]), v1 = fmaf(w0x, v01[1], w1.x * v01[4]); r.y = fmaf(w0y, v0, w1.y * v1); }
    { float v0 = fmaf(w0x, v00[2], w1.x * v00[5]), v1 = fmaf(w0x, v01[2], w1.x * v01[5]); r.z = fmaf(w0y, v0, w1.y * v1); }
    return r;
}

// SmoothDiffuse
MTS_DEV void diffuse_eval_pdf(f3 refl, f3 wi, f3 wo, f3 &value, float &pdf) {
    bool active = wi.z > 0.0f && wo.z > 0.0f;
    value = active ? mk3((refl.x * kInvPi) * wo.z, (refl.y * kInvPi) * wo.z, (refl.z * kInvPi) * wo.z)
                   : mk3(0, 0, 0);
    pdf = active ? kInvPi * wo.z : 0.0f;
}
MTS_DEV void diffuse_sample(f3 refl, f3 wi, f2 sample2, f3 &wo, float &pdf, f3 &weight) {
    wo = mk3(0, 0, 0); pdf = 0.0f; weight = mk3(0, 0, 0);
    if (!(wi.z > 0.0f)) return;
    wo = square_to_cosine_hemisphere(sample2);
    pdf = kInvPi * wo.z;
    if (pdf > 0.0f) weight = refl;
}

// srgb_to_xyz (include/mitsuba/core/spectrum.h:220-227)
MTS_DEV f3 srgb_to_xyz(f3 c) {
    return mk3(fmaf(0.180423f, c.z, fmaf(0.357580f, c.y, 0.412453f * c.x)),
               fmaf(0.072169f, c.z, fmaf(0.715160f, c.y, 0.212671f * c.x)),
               fmaf(0.950227f, c.z, fmaf(0.119193f, c.y, 0.019334f * c.x)));
}

// ---------------------------------------------------------------------------
// PerspectiveCamera::sample_ray_differential (perspective.cpp:190-222), ray part; with an aperture
// (aperture_radius > 0): ThinLensCamera::sample_ray (thinlens.cpp:175-214).
struct CameraView {
    float s2c[16];      // sample_to_camera, row-major
    float c2w[16];      // to_world, row-major
    float near_clip, far_clip;
    float aperture_radius, focus_distance;
};
template <bool LENS = true>
MTS_DEV void camera_ray(const CameraView &c, float sx, float sy, f2 aperture_sample, f3 &o, f3 &d, float &mint, float &maxt) {
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float acc = c.s2c[4 * k + 3];
        acc = fmaf(c.s2c[4 * k + 0], sx, acc);
        acc = fmaf(c.s2c[4 * k + 1], sy, acc);
        acc = fmaf(c.s2c[4 * k + 2], 0.0f, acc);
        r[k] = acc;
    }
    float iw = rcp(r[3]);
    const f3 near_p = mk3(r[0] * iw, r[1] * iw, r[2] * iw);
    f3 dl;
    if (LENS && c.aperture_radius > 0.0f) {
        const f2 t = square_to_uniform_disk_concentric(aperture_sample);
        const f3 aperture_p = mk3(c.aperture_radius * t.x, c.aperture_radius * t.y, 0.0f);
        const f3 focus_p = near_p * (c.focus_distance / near_p.z);
        dl = normalize(focus_p - aperture_p);
        float oo[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {                        // transform_affine(aperture_p)
            float acc = c.c2w[4 * k + 3];
            acc = fmaf(c.c2w[4 * k + 0], aperture_p.x, acc);
            acc = fmaf(c.c2w[4 * k + 1], aperture_p.y, acc);
            acc = fmaf(c.c2w[4 * k + 2], aperture_p.z, acc);
            oo[k] = acc;
        }
        o = mk3(oo[0], oo[1], oo[2]);
    } else {
        dl = normalize(near_p);
        o = mk3(c.c2w[3], c.c2w[7], c.c2w[11]);
    }
    float inv_z = rcp(dl.z);
    mint = c.near_clip * inv_z;
    maxt = c.far_clip * inv_z;
    float dd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float acc = c.c2w[4 * k + 0] * dl.x;
        acc = fmaf(c.c2w[4 * k + 1], dl.y, acc);
        acc = fmaf(c.c2w[4 * k + 2], dl.z, acc);
        dd[k] = acc;
    }
    d = mk3(dd[0], dd[1], dd[2]);
}

} // namespace mtsamd
