// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (wave64) on gfx950, N waves per SIMD.  Lane 0 of every workgroup also stamps
// s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop, so the cycles per instruction are reported both against the
// nominal 2.4 GHz and against the clock the chip really held under this load (MI355X_MICROARCH.md, "DVFS give-back" item 6).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_t __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *stamps) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_t p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float b = 1.0001f, c = 0.5f;
    const float2_t b2 = {b, b}, c2 = {c, c};
    // PK == 2: v_fma_f32 with only the lower 32 lanes enabled -- does a SIMD-32 skip the pass of an all-inactive half?
    if (PK == 2 && (threadIdx.x & 32)) { out[blockIdx.x * 256 + threadIdx.x] = 0.0f; return; }
    for (int i = 0; i < iters; ++i) {
        if (PK == 0 || PK == 2) {
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
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
int main() {
    float *out; hipMalloc(&out, 256 * 8192 * 4);
    unsigned long long *stamps; hipMalloc(&stamps, 2 * 8192 * 8);
    static unsigned long long h[2 * 8192];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pk = 0; pk < 3; ++pk)
        for (int blocks_per_cu = 1; blocks_per_cu <= 8; blocks_per_cu *= 2) {
            int blocks = 256 * blocks_per_cu, iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, stamps);
                else if (pk == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, stamps);
                else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, stamps);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instr = (double) blocks * 4 * iters * 64;   // wave-instructions
            double cyc_per_instr = (ms * 1e-3 * 2.4e9) / (instr / (256.0 * 4));   // SIMD cycles @2.4GHz per wave-instr
            hipMemcpy(h, stamps, 2 * blocks * 8, hipMemcpyDeviceToHost);
            double clk = 0.0, in_kernel = 0.0;      // median-free: mean over workgroups of shader cycles / 100 MHz ticks
            for (int b = 0; b < blocks; ++b) { clk += (double) h[2 * b] / (double) h[2 * b + 1] * 100e6; in_kernel += (double) h[2 * b]; }
            clk /= blocks; in_kernel /= blocks;
            // a workgroup's 4 waves sit on 4 SIMDs: its waves issue 4 * iters * 64 instructions per SIMD ... per wave: iters * 64
            double real_cyc_per_instr = in_kernel / ((double) iters * 64.0) / blocks_per_cu;
            printf("pk=%d waves/SIMD=%d: %.3f ms, %.2f SIMD-cycles(@2.4GHz)/wave-instr, %.1f TFLOP/s, in-kernel clock %.2f GHz, %.2f real SIMD-cycles/wave-instr\n", pk, blocks_per_cu, ms, cyc_per_instr,
                   instr * 64 * 2 * (pk == 1 ? 2 : 1) / (ms * 1e-3) / 1e12, clk * 1e-9, real_cyc_per_instr);
        }
    return 0;
}
