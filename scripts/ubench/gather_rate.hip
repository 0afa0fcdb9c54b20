// Micro-benchmark: how fast does a CU serve the node fetches of a divergent BVH walk?  Every lane chases pointers through a table of
// 64-byte "nodes" (the next index comes out of the node just read: a dependent chain per lane, parallelism only across lanes and
// waves, as in k_trace).  Variants of the fetch:
//   0  per lane 4 x global_load_dwordx4 of its own node (what the BVH4 walk does today)
//   1  per lane 3 x dwordx4 (48-byte node)          2  per lane 2 x dwordx4 (32-byte node)        3  per lane 1 x dwordx4
//   4  quad-cooperative: instruction i loads, with the four lanes of a quad, the four 16-byte chunks of the node of quad lane i
//      (one 64-byte line per quad and instruction), no redistribution (prices the memory pipe alone)
//   5  as 4 + the 4 x 4 transpose inside the quad with v_cndmask_b32_dpp (what a walk would have to do)
//   6  nodes in LDS (16 KB image per workgroup), per lane 4 x ds_read_b128
//   7  as 0 with every second lane idle (exec mask), 8 as 0 with a random half of the lanes idle
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CTRL> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t) __builtin_amdgcn_mov_dpp((int) v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ uint32_t dpp_sel(uint32_t mine, uint32_t other, bool take_other) {
    // take_other ? other[quad-permuted lane] : mine
    const uint32_t o = (uint32_t) __builtin_amdgcn_mov_dpp((int) other, CTRL, 0xf, 0xf, true);
    return take_other ? o : mine;
}

__device__ __forceinline__ uint32_t fold(uint4 u) { return (u.x + u.w) ^ (u.y + (u.z << 1)); }

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k(const uint4 *table, uint32_t n_nodes, uint32_t steps, uint32_t *out, uint32_t seed) {
    extern __shared__ uint4 lds[];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed;
    idx = __umulhi(idx, n_nodes);
    uint32_t acc = 0u;
    if (V == 6) {
        for (uint32_t i = threadIdx.x; i < 1024u; i += 256u) lds[i] = table[i];      // 256 nodes = 16 KB (8 workgroups per CU)
        __syncthreads();
        idx &= 255u;
    }
    if (V == 7 && (lane & 1u)) { out[blockIdx.x * 256u + threadIdx.x] = 0u; return; }
    if (V == 8 && ((idx * 2246822519u) >> 31)) { out[blockIdx.x * 256u + threadIdx.x] = 0u; return; }
    for (uint32_t s = 0; s < steps; ++s) {
        if (V == 0 || V == 7 || V == 8) {
            const uint4 *p = table + 4u * (size_t) idx;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += fold(a) ^ fold(b);
            idx = __umulhi((fold(d) + fold(a) + fold(b) + fold(c)) * 2654435761u, n_nodes);
        } else if (V == 1) {
            const uint4 *p = table + 3u * (size_t) idx;
            const uint4 a = p[0], b = p[1], c = p[2];
            acc += fold(a) ^ fold(b);
            idx = __umulhi((fold(c) + fold(a) + fold(b)) * 2654435761u, n_nodes);
        } else if (V == 2) {
            const uint4 *p = table + 2u * (size_t) idx;
            const uint4 a = p[0], b = p[1];
            acc += fold(a);
            idx = __umulhi((fold(b) + fold(a)) * 2654435761u, n_nodes);
        } else if (V == 3) {
            const uint4 a = table[idx];
            acc += a.x;
            idx = __umulhi(fold(a) * 2654435761u, n_nodes);
        } else if (V == 4 || V == 5) {
            const uint32_t c = lane & 3u;
            const uint32_t i0 = quad_bcast<0x00>(idx), i1 = quad_bcast<0x55>(idx), i2 = quad_bcast<0xaa>(idx), i3 = quad_bcast<0xff>(idx);
            uint4 r0 = table[4u * (size_t) i0 + c], r1 = table[4u * (size_t) i1 + c], r2 = table[4u * (size_t) i2 + c], r3 = table[4u * (size_t) i3 + c];
            if (V == 4) {
                // chunk c of the nodes of quad lanes 0..3; pick one dependent word per lane so the chain stays per lane
                const uint4 mine = c == 0u ? r0 : (c == 1u ? r1 : (c == 2u ? r2 : r3));
                acc += fold(r0) ^ fold(r1) ^ fold(r2) ^ fold(r3);
                idx = __umulhi(fold(mine) * 2654435761u, n_nodes);
            } else {
                // 4 x 4 transpose of 16-byte elements inside the quad: lane j ends with chunks 0..3 of ITS node in r0..r3.
                // stage 1 (lane bit 0 <-> register bit 0): quad_perm [1,0,3,2] = 0xb1
                const bool odd = (c & 1u) != 0u, hi = (c & 2u) != 0u;
#define XCH(A, B, CTRL, SEL) { const uint32_t na = dpp_sel<CTRL>(A, B, SEL), nb = dpp_sel<CTRL>(B, A, !(SEL)); A = na; B = nb; }
                // lane even keeps r0 (its own chunk row), takes partner's r0 into r1; lane odd keeps r1, takes partner's r1 into r0
                XCH(r0.x, r1.x, 0xb1, odd) XCH(r0.y, r1.y, 0xb1, odd) XCH(r0.z, r1.z, 0xb1, odd) XCH(r0.w, r1.w, 0xb1, odd)
                XCH(r2.x, r3.x, 0xb1, odd) XCH(r2.y, r3.y, 0xb1, odd) XCH(r2.z, r3.z, 0xb1, odd) XCH(r2.w, r3.w, 0xb1, odd)
                // stage 2 (lane bit 1 <-> register bit 1): quad_perm [2,3,0,1] = 0x4e
                XCH(r0.x, r2.x, 0x4e, hi) XCH(r0.y, r2.y, 0x4e, hi) XCH(r0.z, r2.z, 0x4e, hi) XCH(r0.w, r2.w, 0x4e, hi)
                XCH(r1.x, r3.x, 0x4e, hi) XCH(r1.y, r3.y, 0x4e, hi) XCH(r1.z, r3.z, 0x4e, hi) XCH(r1.w, r3.w, 0x4e, hi)
#undef XCH
                acc += fold(r0) ^ fold(r1);
                idx = __umulhi((fold(r3) + fold(r0) + fold(r1) + fold(r2)) * 2654435761u, n_nodes);
            }
        } else if (V == 6) {
            const uint4 *p = lds + 4u * idx;
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += fold(a) ^ fold(b);
            idx = (fold(d) + fold(a) + fold(b) + fold(c)) & 255u;
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + idx;
}

int main(int argc, char **argv) {
    const uint32_t steps = 512u;
    const int blocks = 256 * 8 * 4;                        // 4 rounds of full occupancy
    uint32_t *out; hipMalloc(&out, (size_t) blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[] = { 16u << 10, 1u << 20, 5u << 20, 18u << 20, 64u << 20 };
    for (size_t bytes : sizes) {
        const uint32_t n_nodes = (uint32_t) (bytes / 64);
        std::vector<uint32_t> h(bytes / 4);
        uint32_t x = 12345u;
        for (auto &v : h) { x = x * 1664525u + 1013904223u; v = x >> 3; }
        uint4 *table; hipMalloc(&table, bytes + 4096);
        hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice);
        for (int v = 0; v <= 8; ++v) {
            if (v == 6 && bytes != (16u << 10)) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                const uint32_t nn = v == 1 ? (uint32_t) (bytes / 48) : (v == 2 ? (uint32_t) (bytes / 32) : (v == 3 ? (uint32_t) (bytes / 16) : n_nodes));
                switch (v) {
#define L(V) case V: hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), V == 6 ? 16384 : 0, 0, table, nn, steps, out, (uint32_t) rep); break;
                    L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8)
#undef L
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double lanes = (double) blocks * 256 * ((v == 7 || v == 8) ? 0.5 : 1.0);
            const double lookups = lanes * steps;
            printf("table %6.2f MB  variant %d: %8.3f ms  %7.2f G node-fetches/s  (%.1f CU-cycles @2.4GHz per wave-step)\n", bytes / 1048576.0, v, best,
                   lookups / (best * 1e-3) / 1e9, best * 1e-3 * 2.4e9 * 256.0 / ((double) blocks * 4 * steps));
        }
        hipFree(table);
    }
    return 0;
}
