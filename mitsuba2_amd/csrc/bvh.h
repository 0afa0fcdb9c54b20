// bvh.h -- host-side binned-SAH BVH2 builder for the gfx950 traversal kernels.
//
// Replaces the reference's SAH kd-tree builder (include/mitsuba/render/kdtree.h:676-1881):
// only the *query result* of the accelerator is part of the contract (closest t / any hit,
// kdtree.h:2079-2174), so the structure is chosen for the GPU: 64-byte two-child nodes that a
// lane fetches with four 16-byte loads (LDS or L2), leaves of at most 4 triangles stored as
// 48-byte pre-subtracted (p0, e1, e2) records in leaf order.
#pragma once
#include <cstdint>
#include <vector>

namespace mtsamd {

struct BvhOutput {
    std::vector<float> nodes;      // 16 floats per node (see device_scene.h for the layout)
    std::vector<uint32_t> qnodes;  // 8 words per node: the same child boxes on a 16-bit grid over the scene box, rounded outward
    float q_lo[3] = { 0, 0, 0 }, q_step[3] = { 1, 1, 1 };      // grid origin and cell size per axis
    // BVH4 collapsed from the BVH2 (the child of largest surface area is opened until a node has four children): 16 words per node,
    // per child (lo | hi << 16) x, y, z on the grid + child reference; an absent child has reference 0x7fffffff and an inverted box
    std::vector<uint32_t> wnodes;
    // the same BVH4 with the planes as fp16 values of (grid coordinate - 32768), rounded outward (lo down, hi up): the walk reads them
    // straight into v_fma_mix_f32 (no integer -> float conversion); an absent child has lo = +32768, hi = -32768
    std::vector<uint32_t> wnodes_h;
    std::vector<uint32_t> wnodes_p;      // same layout, 15-bit planes stored as 0x8000 | q15 (even cells of the grid): see bvh.cpp
    uint32_t wroot = 0, n_wnodes = 0, wdepth = 0;
    std::vector<float> tris;       // 12 floats per triangle slot
    uint32_t root = 0;             // child reference of the root
    uint32_t n_nodes = 0, n_slots = 0, depth = 0;
    float bbox[6] = { 0, 0, 0, 0, 0, 0 };
};

// builder knobs (defaults = the shipped configuration; experiment builds read others from the environment, api.cpp)
struct BvhOptions {
    int bins = 16;                 // SAH bins per axis
    double intersect_cost = 1.5;   // cost of a triangle test relative to a traversal step
    uint32_t sweep_below = 0;      // nodes of at most this many primitives: exact SAH sweep instead of the bins (0: never)
    bool dfs_order = false;        // pre-order instead of BFS node layout
};

// positions: 9 floats per primitive (p0, p1, p2), n_prims >= 1
void build_bvh(const float *tri_pos, uint32_t n_prims, uint32_t max_leaf, BvhOutput &out, const BvhOptions *opt = nullptr);

} // namespace mtsamd
