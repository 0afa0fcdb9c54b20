// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (wave64) on gfx950, N waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_t p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float b = 1.0001f, c = 0.5f;
    const float2_t b2 = {b, b}, c2 = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (PK == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(b2), "v"(c2));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
int main() {
    float *out; hipMalloc(&out, 256 * 8192 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pk = 0; pk < 2; ++pk)
        for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
            int blocks = 256 * blocks_per_cu, iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instr = (double) blocks * 4 * iters * 64;   // wave-instructions
            double cyc_per_instr = (ms * 1e-3 * 2.4e9) / (instr / (256.0 * 4));   // SIMD cycles @2.4GHz per wave-instr
            printf("pk=%d waves/SIMD=%d: %.3f ms, %.2f SIMD-cycles(@2.4GHz)/wave-instr, %.1f TFLOP/s\n", pk, blocks_per_cu, ms, cyc_per_instr,
                   instr * 64 * 2 * (pk ? 2 : 1) / (ms * 1e-3) / 1e12);
        }
    return 0;
}
