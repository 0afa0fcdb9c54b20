// Micro-benchmark: issue cost of the instruction classes the BVH4 step is made of (wave64, gfx950), 8 waves per SIMD, eight independent
// dependency chains per wave.  Reports SIMD-cycles per wave-instruction = kernel time (HIP events) x the shader clock the chip really
// held (s_memtime / s_memrealtime) / wave-instructions per SIMD.  (The workgroups' own s_memtime spans are shorter than the kernel --
// they do not all run at once -- so they must not be used as the time base: that was the 1.36-vs-2.26 discrepancy of
// profiles/r02_valu_rate.txt.)
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define I1(n, a, b, c) a "%" #n b "%" #n c "\n"
#define I8(a, b, c) I1(0, a, b, c) I1(1, a, b, c) I1(2, a, b, c) I1(3, a, b, c) I1(4, a, b, c) I1(5, a, b, c) I1(6, a, b, c) I1(7, a, b, c)
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s20", "s21"
#define REGS2 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(b2), "v"(c2) : "vcc", "s20", "s21"
typedef float float2_t __attribute__((ext_vector_type(2)));
static const char *kNames[] = {
    "v_fma_f32", "v_pk_fma_f32", "v_perm_b32", "v_cvt_f32_u32", "v_cvt_f32_u32 sdwa WORD_1", "v_cvt_f32_ubyte1", "v_max3_f32", "v_min_f32",
    "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32 -> sgpr pair", "v_cndmask_b32 (vcc)", "v_cndmask_b32 (sgpr pair)", "v_and_b32", "v_lshrrev_b32",
    "v_add_u32", "v_mov_b32 dpp quad_perm", "v_fma_mix_f32", "v_med3_f32", "v_bfe_u32", "v_lshl_add_u32", "v_mul_f32", "v_add_f32",
    "v_rcp_f32", "v_cvt_f32_f16", "v_mad_u32_u24", "v_mov_b32", "v_pk_mul_f32", "v_cvt_f32_u32 sdwa WORD_0", "v_and_or_b32", "v_min3_f32",
    "v_cmp + v_cndmask pair (vcc)", "v_max_f32", "v_cvt_f32_ubyte0 + ubyte1 + ubyte2 + ubyte3 (4 instr)", "v_cmp_le_f32 e64 + v_cndmask e64 (sgpr)" };
constexpr int kOps = sizeof(kNames) / sizeof(kNames[0]);
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *stamps) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_t p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float b = 1.0001f, c = 0.5f;
    const float2_t b2 = {b, b}, c2 = {c, c};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) asm volatile(I8("v_fma_f32 ", ", ", ", %8, %9") REGS);
            if (OP == 1) asm volatile(I8("v_pk_fma_f32 ", ", ", ", %8, %9") REGS2);
            if (OP == 2) asm volatile(I8("v_perm_b32 ", ", ", ", %8, %9") REGS);
            if (OP == 3) asm volatile(I8("v_cvt_f32_u32_e32 ", ", ", "") REGS);
            if (OP == 4) asm volatile(I8("v_cvt_f32_u32_sdwa ", ", ", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1") REGS);
            if (OP == 5) asm volatile(I8("v_cvt_f32_ubyte1_e32 ", ", ", "") REGS);
            if (OP == 6) asm volatile(I8("v_max3_f32 ", ", ", ", %8, %9") REGS);
            if (OP == 7) asm volatile(I8("v_min_f32_e32 ", ", %8, ", "") REGS);
            if (OP == 8) asm volatile(I8("v_cmp_lt_f32_e32 vcc, ", ", %8 ; ", "") REGS);
            if (OP == 9) asm volatile(I8("v_cmp_lt_f32_e64 s[20:21], ", ", %8 ; ", "") REGS);
            if (OP == 10) asm volatile(I8("v_cndmask_b32_e32 ", ", %8, ", ", vcc") REGS);
            if (OP == 11) asm volatile(I8("v_cndmask_b32_e64 ", ", %8, ", ", s[20:21]") REGS);
            if (OP == 12) asm volatile(I8("v_and_b32_e32 ", ", %8, ", "") REGS);
            if (OP == 13) asm volatile(I8("v_lshrrev_b32_e32 ", ", 1, ", "") REGS);
            if (OP == 14) asm volatile(I8("v_add_u32_e32 ", ", %8, ", "") REGS);
            if (OP == 15) asm volatile(I8("v_mov_b32_dpp ", ", ", " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") REGS);
            if (OP == 16) asm volatile(I8("v_fma_mix_f32 ", ", ", ", %8, %9 op_sel_hi:[1,0,0]") REGS);
            if (OP == 17) asm volatile(I8("v_med3_f32 ", ", ", ", %8, %9") REGS);
            if (OP == 18) asm volatile(I8("v_bfe_u32 ", ", ", ", 16, 16") REGS);
            if (OP == 19) asm volatile(I8("v_lshl_add_u32 ", ", ", ", 3, %9") REGS);
            if (OP == 20) asm volatile(I8("v_mul_f32_e32 ", ", %8, ", "") REGS);
            if (OP == 21) asm volatile(I8("v_add_f32_e32 ", ", %8, ", "") REGS);
            if (OP == 22) asm volatile(I8("v_rcp_f32_e32 ", ", ", "") REGS);
            if (OP == 23) asm volatile(I8("v_cvt_f32_f16_e32 ", ", ", "") REGS);
            if (OP == 24) asm volatile(I8("v_mad_u32_u24 ", ", ", ", %8, %9") REGS);
            if (OP == 25) asm volatile(I8("v_mov_b32_e32 ", ", %8 ; ", "") REGS);
            if (OP == 26) asm volatile(I8("v_pk_mul_f32 ", ", ", ", %8") REGS2);
            if (OP == 27) asm volatile(I8("v_cvt_f32_u32_sdwa ", ", ", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0") REGS);
            if (OP == 28) asm volatile(I8("v_and_or_b32 ", ", ", ", %8, %9") REGS);
            if (OP == 29) asm volatile(I8("v_min3_f32 ", ", ", ", %8, %9") REGS);
            if (OP == 30) asm volatile(I8("v_cmp_lt_f32_e32 vcc, %8, ", "\n v_cndmask_b32_e32 ", ", %9, %8, vcc") REGS);      // counted as 2 instructions below
            if (OP == 31) asm volatile(I8("v_max_f32_e32 ", ", %8, ", "") REGS);
            if (OP == 32) asm volatile("v_cvt_f32_ubyte0_e32 %0, %8\n v_cvt_f32_ubyte1_e32 %1, %8\n v_cvt_f32_ubyte2_e32 %2, %8\n v_cvt_f32_ubyte3_e32 %3, %8\n"
                                       "v_cvt_f32_ubyte0_e32 %4, %9\n v_cvt_f32_ubyte1_e32 %5, %9\n v_cvt_f32_ubyte2_e32 %6, %9\n v_cvt_f32_ubyte3_e32 %7, %9\n" REGS);
            if (OP == 33) asm volatile(I8("v_cmp_le_f32_e64 s[20:21], %8, ", "\n v_cndmask_b32_e64 ", ", %9, %8, s[20:21]") REGS);  // 2 instructions
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int OP> void run(float *out, unsigned long long *stamps, unsigned long long *h) {
    const int blocks_per_cu = 8, blocks = 256 * blocks_per_cu, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, stamps);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, stamps, 2 * blocks * 8, hipMemcpyDeviceToHost);
    double clk = 0.0, in_kernel = 0.0;
    for (int b = 0; b < blocks; ++b) { clk += (double) h[2 * b] / (double) h[2 * b + 1] * 100e6; in_kernel += (double) h[2 * b]; }
    clk /= blocks; in_kernel /= blocks;
    const double per_iter = (OP == 30 || OP == 33) ? 128.0 : 64.0;
    const double per_simd = (double) blocks * 4.0 * iters * per_iter / 1024.0;      // wave-instructions per SIMD (256 CUs x 4)
    (void) in_kernel;
    printf("%-58s %6.2f SIMD-cycles per wave-instruction (clock %.2f GHz, %.2f ms, %.2f ns)\n", kNames[OP], ms * 1e-3 * clk / per_simd, clk * 1e-9, ms, ms * 1e6 / per_simd);
    if constexpr (OP + 1 < kOps) run<OP + 1>(out, stamps, h);
}
int main() {
    float *out; hipMalloc(&out, 256 * 8192 * 4);
    unsigned long long *stamps; hipMalloc(&stamps, 2 * 8192 * 8);
    static unsigned long long h[2 * 8192];
    run<0>(out, stamps, h);
    return 0;
}
