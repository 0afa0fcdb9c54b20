// bvh.cpp -- binned-SAH BVH2 builder (host).  See bvh.h.
#include "bvh.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <queue>

namespace mtsamd {
namespace {

constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kLeafCountShift = 27;
constexpr int kMaxBins = 64;
constexpr double kTraversalCost = 1.0;

struct Box {
    double lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; ++k) { lo[k] = DBL_MAX; hi[k] = -DBL_MAX; } }
    void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    void grow(const double *p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    double area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0) return 0.0;
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    }
};

struct TmpNode {
    Box box;
    int32_t left = -1, right = -1;   // children (TmpNode indices) or -1
    uint32_t first = 0, count = 0;   // leaf range in the permuted primitive list
    uint32_t depth = 0;
};

struct Builder {
    const float *tri_pos;
    uint32_t n_prims, max_leaf;
    std::vector<Box> prim_box;
    std::vector<double> centroid;    // 3 per prim
    std::vector<uint32_t> perm;
    std::vector<TmpNode> nodes;
    uint32_t max_depth = 0;
    int kBins = 16;                   // BvhOptions
    double kIntersectCost = 1.5;
    uint32_t sweep_below = 0;
    std::vector<uint32_t> scratch;
    std::vector<double> suffix_area;

    int32_t build(uint32_t first, uint32_t count, uint32_t depth) {
        int32_t idx = (int32_t) nodes.size();
        nodes.emplace_back();
        TmpNode n;
        n.depth = depth;
        n.box.reset();
        Box cb; cb.reset();
        for (uint32_t i = 0; i < count; ++i) {
            uint32_t p = perm[first + i];
            n.box.grow(prim_box[p]);
            cb.grow(&centroid[3 * p]);
        }
        max_depth = std::max(max_depth, depth);
        auto make_leaf = [&]() { n.first = first; n.count = count; nodes[idx] = n; return idx; };
        if (count <= 1) return make_leaf();

        // binned SAH over the three axes
        double best_cost = DBL_MAX; int best_axis = -1, best_bin = -1;
        double parent_area = std::max(n.box.area(), 1e-300);
        for (int axis = 0; axis < 3; ++axis) {
            double lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0.0)) continue;
            Box bin_box[kMaxBins]; uint32_t bin_cnt[kMaxBins];
            for (int b = 0; b < kBins; ++b) { bin_box[b].reset(); bin_cnt[b] = 0; }
            double scale = kBins / ext;
            for (uint32_t i = 0; i < count; ++i) {
                uint32_t p = perm[first + i];
                int b = std::min(kBins - 1, std::max(0, (int) ((centroid[3 * p + axis] - lo) * scale)));
                bin_box[b].grow(prim_box[p]); bin_cnt[b]++;
            }
            double right_area[kMaxBins]; uint32_t right_cnt[kMaxBins];
            Box acc; acc.reset(); uint32_t cnt = 0;
            for (int b = kBins - 1; b > 0; --b) {
                if (bin_cnt[b]) acc.grow(bin_box[b]);
                cnt += bin_cnt[b];
                right_area[b] = cnt ? acc.area() : 0.0; right_cnt[b] = cnt;
            }
            acc.reset(); cnt = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                if (bin_cnt[b]) acc.grow(bin_box[b]);
                cnt += bin_cnt[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                double cost = kTraversalCost +
                              kIntersectCost * (acc.area() * cnt + right_area[b + 1] * right_cnt[b + 1]) / parent_area;
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        // small nodes: the exact SAH sweep over every split position of every axis (centroid order) instead of the bins
        int sweep_axis = -1; uint32_t sweep_left = 0;
        if (count <= sweep_below) {
            best_cost = DBL_MAX;
            scratch.resize(count); suffix_area.resize(count + 1);
            for (int axis = 0; axis < 3; ++axis) {
                for (uint32_t i = 0; i < count; ++i) scratch[i] = perm[first + i];
                std::sort(scratch.begin(), scratch.end(), [&](uint32_t a, uint32_t b) {
                    const double ca = centroid[3 * a + axis], cbb = centroid[3 * b + axis];
                    return ca < cbb || (ca == cbb && a < b);
                });
                Box acc; acc.reset();
                for (uint32_t i = count; i-- > 0;) { acc.grow(prim_box[scratch[i]]); suffix_area[i] = acc.area(); }
                acc.reset();
                for (uint32_t i = 0; i + 1 < count; ++i) {
                    acc.grow(prim_box[scratch[i]]);
                    const double cost = kTraversalCost + kIntersectCost * (acc.area() * (i + 1) + suffix_area[i + 1] * (count - i - 1)) / parent_area;
                    if (cost < best_cost) { best_cost = cost; sweep_axis = axis; sweep_left = i + 1; }
                }
            }
            if (sweep_axis >= 0) best_axis = 3;                 // marks "sweep split chosen"
        }
        double leaf_cost = kIntersectCost * count;
        if (count <= max_leaf && (best_axis < 0 || leaf_cost <= best_cost)) return make_leaf();

        // Depth guard: the traversal kernels keep a per-lane stack of `depth` entries in LDS (<= 64 KB per 256-thread workgroup).
        // Past depth 40 the split is an object median along the widest centroid axis, which bounds the total depth by
        // 40 + log2(n) whatever the SAH would have done on a pathological primitive distribution.
        if (depth >= 40 && best_axis >= 0) {
            int axis = 0;
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
            std::nth_element(perm.begin() + first, perm.begin() + first + count / 2, perm.begin() + first + count,
                             [&](uint32_t a, uint32_t b) { return centroid[3 * a + axis] < centroid[3 * b + axis]; });
            best_axis = -2;
        }
        uint32_t mid;
        if (best_axis == -2) {
            mid = first + count / 2;
        } else if (best_axis == 3) {
            std::sort(perm.begin() + first, perm.begin() + first + count, [&](uint32_t a, uint32_t b) {
                const double ca = centroid[3 * a + sweep_axis], cbb = centroid[3 * b + sweep_axis];
                return ca < cbb || (ca == cbb && a < b);
            });
            mid = first + sweep_left;
        } else if (best_axis >= 0) {
            double lo = cb.lo[best_axis], ext = cb.hi[best_axis] - cb.lo[best_axis];
            double scale = kBins / ext;
            auto it = std::partition(perm.begin() + first, perm.begin() + first + count, [&](uint32_t p) {
                int b = std::min(kBins - 1, std::max(0, (int) ((centroid[3 * p + best_axis] - lo) * scale)));
                return b <= best_bin;
            });
            mid = (uint32_t) (it - perm.begin());
        } else {
            // all centroids coincide: split by index
            std::sort(perm.begin() + first, perm.begin() + first + count);
            mid = first + count / 2;
        }
        if (mid == first || mid == first + count) mid = first + count / 2;
        nodes[idx] = n;
        int32_t l = build(first, mid - first, depth + 1);
        int32_t r = build(mid, first + count - mid, depth + 1);
        nodes[idx].left = l; nodes[idx].right = r;
        return idx;
    }
};

// outward-rounded, padded float bounds: the slab test must never cull a triangle that the fp32
// Moeller-Trumbore test (mesh.h:195-221) would accept.
inline void padded(const Box &b, double scene_extent, float lo[3], float hi[3]) {
    for (int k = 0; k < 3; ++k) {
        double pad = 1e-5 * std::max({ std::fabs(b.lo[k]), std::fabs(b.hi[k]), b.hi[k] - b.lo[k] }) +
                     1e-7 * scene_extent + 1e-30;
        lo[k] = std::nextafter((float) (b.lo[k] - pad), -INFINITY);
        hi[k] = std::nextafter((float) (b.hi[k] + pad), INFINITY);
    }
}

inline float bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

} // namespace

void build_bvh(const float *tri_pos, uint32_t n_prims, uint32_t max_leaf, BvhOutput &out, const BvhOptions *opt) {
    Builder b;
    b.tri_pos = tri_pos; b.n_prims = n_prims; b.max_leaf = std::min<uint32_t>(std::max<uint32_t>(max_leaf, 1), 15);
    const BvhOptions defaults;
    if (!opt) opt = &defaults;
    b.kBins = std::min(std::max(opt->bins, 2), kMaxBins); b.kIntersectCost = opt->intersect_cost; b.sweep_below = opt->sweep_below;
    b.prim_box.resize(n_prims); b.centroid.resize(3 * (size_t) n_prims); b.perm.resize(n_prims);
    Box scene; scene.reset();
    for (uint32_t p = 0; p < n_prims; ++p) {
        Box bx; bx.reset();
        for (int j = 0; j < 3; ++j) {
            double v[3] = { tri_pos[9 * (size_t) p + 3 * j], tri_pos[9 * (size_t) p + 3 * j + 1], tri_pos[9 * (size_t) p + 3 * j + 2] };
            bx.grow(v);
        }
        b.prim_box[p] = bx;
        for (int k = 0; k < 3; ++k) b.centroid[3 * (size_t) p + k] = 0.5 * (bx.lo[k] + bx.hi[k]);
        b.perm[p] = p;
        scene.grow(bx);
    }
    double extent = 0.0;
    for (int k = 0; k < 3; ++k) {
        extent = std::max({ extent, std::fabs(scene.lo[k]), std::fabs(scene.hi[k]), scene.hi[k] - scene.lo[k] });
        out.bbox[k] = (float) scene.lo[k]; out.bbox[3 + k] = (float) scene.hi[k];
    }
    b.nodes.reserve(2 * (size_t) n_prims);
    int32_t root = b.build(0, n_prims, 0);

    // emit: inner nodes in BFS order (the top of the tree comes first, so a prefix of the node
    // array is what gets staged in LDS), triangle slots in the order their leaves are reached.
    std::vector<int32_t> inner_index(b.nodes.size(), -1);
    std::vector<int32_t> bfs;
    if (opt->dfs_order) {                                          // experiment: pre-order layout
        std::vector<int32_t> st;
        if (b.nodes[root].left >= 0) st.push_back(root);
        while (!st.empty()) {
            int32_t t = st.back(); st.pop_back();
            inner_index[t] = (int32_t) bfs.size(); bfs.push_back(t);
            const TmpNode &n = b.nodes[t];
            if (b.nodes[n.right].left >= 0) st.push_back(n.right);
            if (b.nodes[n.left].left >= 0) st.push_back(n.left);
        }
    } else {
        std::queue<int32_t> q;
        if (b.nodes[root].left >= 0) q.push(root);
        while (!q.empty()) {
            int32_t t = q.front(); q.pop();
            inner_index[t] = (int32_t) bfs.size(); bfs.push_back(t);
            const TmpNode &n = b.nodes[t];
            if (b.nodes[n.left].left >= 0) q.push(n.left);
            if (b.nodes[n.right].left >= 0) q.push(n.right);
        }
    }
    out.nodes.assign(16 * bfs.size(), 0.0f);
    out.tris.clear(); out.tris.reserve(12 * (size_t) n_prims);
    uint32_t slot = 0;
    auto emit_leaf = [&](const TmpNode &n) -> uint32_t {
        uint32_t start = slot;
        for (uint32_t i = 0; i < n.count; ++i) {
            uint32_t p = b.perm[n.first + i];
            const float *tp = tri_pos + 9 * (size_t) p;
            float p0[3] = { tp[0], tp[1], tp[2] };
            float e1[3] = { tp[3] - tp[0], tp[4] - tp[1], tp[5] - tp[2] };
            float e2[3] = { tp[6] - tp[0], tp[7] - tp[1], tp[8] - tp[2] };
            float rec[12] = { p0[0], p0[1], p0[2], e1[0], e1[1], e1[2], e2[0], e2[1], e2[2], bits(p), 0.0f, 0.0f };
            out.tris.insert(out.tris.end(), rec, rec + 12);
            ++slot;
        }
        return kLeafFlag | (n.count << kLeafCountShift) | start;
    };
    std::vector<uint32_t> leaf_ref(b.nodes.size(), 0u);
    auto child_ref = [&](int32_t t) -> uint32_t {
        const TmpNode &n = b.nodes[t];
        if (n.left >= 0) return (uint32_t) inner_index[t];
        return leaf_ref[t] = emit_leaf(n);
    };
    for (size_t i = 0; i < bfs.size(); ++i) {
        const TmpNode &n = b.nodes[bfs[i]];
        float llo[3], lhi[3], rlo[3], rhi[3];
        padded(b.nodes[n.left].box, extent, llo, lhi);
        padded(b.nodes[n.right].box, extent, rlo, rhi);
        uint32_t cl = child_ref(n.left), cr = child_ref(n.right);
        float *q = out.nodes.data() + 16 * i;
        q[0] = llo[0]; q[1] = llo[1]; q[2] = llo[2]; q[3] = lhi[0];
        q[4] = lhi[1]; q[5] = lhi[2]; q[6] = rlo[0]; q[7] = rlo[1];
        q[8] = rlo[2]; q[9] = rhi[0]; q[10] = rhi[1]; q[11] = rhi[2];
        q[12] = bits(cl); q[13] = bits(cr); q[14] = 0.0f; q[15] = 0.0f;
    }
    // 32-byte nodes: child boxes snapped outward (by at least 1/8 cell: the device decodes them with an error below 0.07 cells)
    // to a 65536^3 grid over the (padded) scene box.  One traversal step then costs two
    // 16-byte loads per lane instead of four; the walk only culls with the boxes, so hits are unchanged.
    {
        double glo[3], gstep[3];
        for (int k = 0; k < 3; ++k) {
            const double pad = (scene.hi[k] - scene.lo[k] + extent + 1.0) * 1e-5;
            glo[k] = scene.lo[k] - pad;
            // the padded scene box ends at cell 65532: every snapped plane stays below 65534, so that the 15-bit variant of the BVH4
            // nodes (even cells only) can round a far plane up without leaving the grid
            gstep[k] = std::max((scene.hi[k] + pad - glo[k]) / 65532.0, 1e-30);
            out.q_lo[k] = (float) glo[k]; out.q_step[k] = (float) gstep[k];
            // the device decodes with the float values: make them the reference for the snapping below
            glo[k] = (double) out.q_lo[k]; gstep[k] = (double) out.q_step[k];
        }
        auto qlo = [&](float v, int k) -> uint32_t {
            double q = std::floor(((double) v - glo[k]) / gstep[k] - 0.125);
            return (uint32_t) std::min(65535.0, std::max(0.0, q));
        };
        auto qhi = [&](float v, int k) -> uint32_t {
            double q = std::ceil(((double) v - glo[k]) / gstep[k] + 0.125);
            return (uint32_t) std::min(65535.0, std::max(0.0, q));
        };
        out.qnodes.assign(8 * bfs.size(), 0u);
        for (size_t i = 0; i < bfs.size(); ++i) {
            const float *q = out.nodes.data() + 16 * i;
            const float llo[3] = { q[0], q[1], q[2] }, lhi[3] = { q[3], q[4], q[5] }, rlo[3] = { q[6], q[7], q[8] }, rhi[3] = { q[9], q[10], q[11] };
            uint32_t *w = out.qnodes.data() + 8 * i;
            // one word per child and axis: lo | hi << 16 (the walk swaps the halves with one v_perm_b32 according to the sign
            // of the ray direction and so gets (near plane, far plane) without min / max)
            for (int k = 0; k < 3; ++k) {
                w[k] = qlo(llo[k], k) | (qhi(lhi[k], k) << 16);
                w[3 + k] = qlo(rlo[k], k) | (qhi(rhi[k], k) << 16);
            }
            std::memcpy(&w[6], &q[12], 4); std::memcpy(&w[7], &q[13], 4);
        }
    }
    if (bfs.empty()) out.root = emit_leaf(b.nodes[root]);   // whole scene is one leaf
    else out.root = 0;
    // ---- BVH4: collapse (open the child of largest surface area until four children), BFS order, same grid
    out.wnodes.clear(); out.wnodes_h.clear(); out.wnodes_p.clear(); out.wroot = out.root; out.n_wnodes = 0; out.wdepth = 1;
    if (!bfs.empty()) {
        double glo[3], gstep[3];
        for (int k = 0; k < 3; ++k) { glo[k] = (double) out.q_lo[k]; gstep[k] = (double) out.q_step[k]; }
        auto qlo = [&](float v, int k) -> uint32_t { return (uint32_t) std::min(65535.0, std::max(0.0, std::floor(((double) v - glo[k]) / gstep[k] - 0.125))); };
        auto qhi = [&](float v, int k) -> uint32_t { return (uint32_t) std::min(65535.0, std::max(0.0, std::ceil(((double) v - glo[k]) / gstep[k] + 0.125))); };
        struct Wide { int32_t child[4]; int n; uint32_t depth; };
        std::vector<Wide> wide;
        std::vector<int32_t> wide_of(b.nodes.size(), -1);      // BVH2 inner node -> wide node that replaces it
        std::queue<std::pair<int32_t, uint32_t>> q;
        q.push({ root, 1u });
        wide_of[root] = 0; wide.push_back(Wide{});
        while (!q.empty()) {
            const int32_t t = q.front().first; const uint32_t depth = q.front().second; q.pop();
            Wide wn{}; wn.depth = depth;
            wn.child[0] = b.nodes[t].left; wn.child[1] = b.nodes[t].right; wn.n = 2;
            while (wn.n < 4) {
                int best = -1; double best_area = -1.0;
                for (int c = 0; c < wn.n; ++c)
                    if (b.nodes[wn.child[c]].left >= 0 && b.nodes[wn.child[c]].box.area() > best_area) { best = c; best_area = b.nodes[wn.child[c]].box.area(); }
                if (best < 0) break;
                const int32_t open = wn.child[best];
                wn.child[best] = b.nodes[open].left; wn.child[wn.n++] = b.nodes[open].right;
            }
            for (int c = 0; c < wn.n; ++c)
                if (b.nodes[wn.child[c]].left >= 0) { wide_of[wn.child[c]] = (int32_t) wide.size(); wide.push_back(Wide{}); q.push({ wn.child[c], depth + 1 }); }
            wide[wide_of[t]] = wn;
            out.wdepth = std::max(out.wdepth, depth);
        }
        // fp16 with directed rounding: the largest half <= v / the smallest half >= v (v an integer in [-32768, 32768])
        auto half_of = [](float v, bool up) -> uint32_t {
            uint32_t sign = v < 0.0f ? 0x8000u : 0u;
            float a = std::fabs(v);
            if (a == 0.0f) return sign;
            int e; float m = std::frexp(a, &e);                       // a = m 2^e, m in [0.5, 1)
            // 11 significant bits: a = k 2^(e - 11), k in [1024, 2048)
            const float scaled = std::ldexp(m, 11);
            const bool away = (v >= 0.0f) == up;                       // towards +inf of a positive / -inf of a negative value: magnitude up
            float k = away ? std::ceil(scaled) : std::floor(scaled);
            if (k >= 2048.0f) { k = 1024.0f; ++e; }
            const int be = e - 1 + 15;                                // biased exponent of k 2^(e - 11) = (k / 1024) 2^(e - 1)
            return sign | ((uint32_t) be << 10) | ((uint32_t) k - 1024u);
        };
        out.n_wnodes = (uint32_t) wide.size();
        out.wnodes_h.assign(16 * wide.size(), 0u);
        for (size_t i = 0; i < wide.size(); ++i) {
            uint32_t *w = out.wnodes_h.data() + 16 * i;
            for (int c = 0; c < 4; ++c) {
                if (c >= wide[i].n) { w[4 * c] = w[4 * c + 1] = w[4 * c + 2] = 0x7800u | (0xf800u << 16); w[4 * c + 3] = 0x7fffffffu; continue; }      // lo = +32768, hi = -32768
                const int32_t t = wide[i].child[c];
                float lo[3], hi[3];
                padded(b.nodes[t].box, extent, lo, hi);
                for (int k = 0; k < 3; ++k) {
                    const float vlo = (float) qlo(lo[k], k) - 32768.0f, vhi = (float) qhi(hi[k], k) - 32768.0f;
                    w[4 * c + k] = half_of(vlo, false) | (half_of(vhi, true) << 16);
                }
                w[4 * c + 3] = b.nodes[t].left >= 0 ? (uint32_t) wide_of[t] : leaf_ref[t];
            }
        }
        // 15-bit planes on the even cells of the same grid, stored as 0x8000 | q15: the two bytes are the upper mantissa (and the lowest
        // exponent bit) of the float 65536 + 2 q15, which the walk assembles with one v_perm_b32 per plane -- no integer-to-float convert
        out.wnodes_p.assign(16 * wide.size(), 0u);
        for (size_t i = 0; i < wide.size(); ++i) {
            uint32_t *w = out.wnodes_p.data() + 16 * i;
            for (int c = 0; c < 4; ++c) {
                if (c >= wide[i].n) { w[4 * c] = w[4 * c + 1] = w[4 * c + 2] = 0xffffu | (0x8000u << 16); w[4 * c + 3] = 0x7fffffffu; continue; }      // lo = 65534, hi = 0
                const int32_t t = wide[i].child[c];
                float lo[3], hi[3];
                padded(b.nodes[t].box, extent, lo, hi);
                for (int k = 0; k < 3; ++k) {
                    const uint32_t l15 = qlo(lo[k], k) >> 1, h15 = std::min(32767u, (qhi(hi[k], k) + 1u) >> 1);      // qhi <= 65534 (grid above)
                    w[4 * c + k] = (0x8000u | l15) | ((0x8000u | h15) << 16);
                }
                w[4 * c + 3] = b.nodes[t].left >= 0 ? (uint32_t) wide_of[t] : leaf_ref[t];
            }
        }
        out.wnodes.assign(16 * wide.size(), 0u);
        for (size_t i = 0; i < wide.size(); ++i) {
            uint32_t *w = out.wnodes.data() + 16 * i;
            for (int c = 0; c < 4; ++c) {
                if (c >= wide[i].n) { w[4 * c] = w[4 * c + 1] = w[4 * c + 2] = 65535u; w[4 * c + 3] = 0x7fffffffu; continue; }      // lo = 65535, hi = 0
                const int32_t t = wide[i].child[c];
                float lo[3], hi[3];
                padded(b.nodes[t].box, extent, lo, hi);
                for (int k = 0; k < 3; ++k) w[4 * c + k] = qlo(lo[k], k) | (qhi(hi[k], k) << 16);
                w[4 * c + 3] = b.nodes[t].left >= 0 ? (uint32_t) wide_of[t] : leaf_ref[t];
            }
        }
        out.wroot = 0;
    }
    out.n_nodes = (uint32_t) bfs.size();
    out.n_slots = slot;
    out.depth = b.max_depth + 1;
}

} // namespace mtsamd
