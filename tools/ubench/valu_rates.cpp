// VALU issue rates on gfx950, measured: N dependent-free instructions per lane in a loop, 4 waves per SIMD resident.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rates.cpp -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if (OP == 1) { asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)); }
        if (OP == 2) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if (OP == 3) { asm volatile("v_mul_lo_u32 %0, %0, %0\n v_mul_lo_u32 %1, %1, %1\n v_mul_lo_u32 %2, %2, %2\n v_mul_lo_u32 %3, %3, %3\n v_mul_lo_u32 %4, %4, %4\n v_mul_lo_u32 %5, %5, %5\n v_mul_lo_u32 %6, %6, %6\n v_mul_lo_u32 %7, %7, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)); }
        if (OP == 4) { asm volatile("v_mul_u32_u24 %0, %0, %0\n v_mul_u32_u24 %1, %1, %1\n v_mul_u32_u24 %2, %2, %2\n v_mul_u32_u24 %3, %3, %3\n v_mul_u32_u24 %4, %4, %4\n v_mul_u32_u24 %5, %5, %5\n v_mul_u32_u24 %6, %6, %6\n v_mul_u32_u24 %7, %7, %7" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)); }
        if (OP == 5) { asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if (OP == 6) { asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if (OP == 7) { asm volatile("v_pk_sub_i16 %0, %0, %1\n v_pk_sub_i16 %1, %1, %2\n v_pk_sub_i16 %2, %2, %3\n v_pk_sub_i16 %3, %3, %4\n v_pk_sub_i16 %4, %4, %5\n v_pk_sub_i16 %5, %5, %6\n v_pk_sub_i16 %6, %6, %7\n v_pk_sub_i16 %7, %7, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)); }
        if (OP == 8) { asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n v_pk_mul_f32 %4, %4, %4\n v_pk_add_f32 %5, %5, %5\n v_pk_mul_f32 %6, %6, %6\n v_pk_add_f32 %7, %7, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)); }
        if (OP == 9) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %4\n v_mad_u32_u24 %3, %3, %4, %5\n v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %5, %5, %6, %7\n v_mad_u32_u24 %6, %6, %7, %0\n v_mad_u32_u24 %7, %7, %0, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)); }
        if (OP == 10) { asm volatile("v_xor_b32 %0, %0, %1\n v_lshrrev_b32 %1, 7, %1\n v_xor_b32 %2, %2, %3\n v_lshlrev_b32 %3, 9, %3\n v_xor_b32 %4, %4, %5\n v_lshrrev_b32 %5, 3, %5\n v_and_b32 %6, %6, %7\n v_or_b32 %7, %7, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.x + p6.x + p7.x + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
}
template <int OP> void run(const char* name, float* d) {
    const int iters = 4096, blocks = 256 * 4;       // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)iters * 8 * blocks * 4;            // per-wave instructions in total
    const double per_simd = wave_instr / (256.0 * 4);                    // per SIMD
    printf("%-22s %8.3f ms  -> %.2f clk per wave-instruction per SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("v_fma_f32", d); run<1>("v_pk_fma_f32", d); run<8>("v_pk_mul/add_f32", d); run<2>("v_exp_f32", d); run<3>("v_mul_lo_u32", d);
    run<4>("v_mul_u32_u24", d); run<9>("v_mad_u32_u24", d); run<5>("v_cvt_pk_bf16_f32", d); run<6>("v_max3_f32", d); run<7>("v_pk_sub_i16", d);
    run<10>("xor/shift/and/or", d);
    return 0;
}
