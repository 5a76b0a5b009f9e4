// VALU issue rates on gfx950: fma vs packed fma vs v_exp_f32 vs v_rcp_f32 vs v_mul_lo_u32 (ops per clock per CU), 8 waves per SIMD resident
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2;} } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f + 0.5f, a1 = a0 + 0.1f, a2 = a0 + 0.2f, a3 = a0 + 0.3f, a4 = a0 + .4f, a5 = a0 + .5f, a6 = a0 + .6f, a7 = a0 + .7f;
    unsigned u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { a0 = a0 * 1.0001f + 0.1f; a1 = a1 * 1.0001f + 0.1f; a2 = a2 * 1.0001f + 0.1f; a3 = a3 * 1.0001f + 0.1f; a4 = a4 * 1.0001f + 0.1f; a5 = a5 * 1.0001f + 0.1f; a6 = a6 * 1.0001f + 0.1f; a7 = a7 * 1.0001f + 0.1f; }
        if (MODE == 1) { p0 = p0 * 1.0001f + 0.1f; p1 = p1 * 1.0001f + 0.1f; p2 = p2 * 1.0001f + 0.1f; p3 = p3 * 1.0001f + 0.1f; }
        if (MODE == 2) { a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3); a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7); }
        if (MODE == 3) { a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3); a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7); }
        if (MODE == 4) { u0 = u0 * 0x7feb352du; u1 = u1 * 0x7feb352du; u2 = u2 * 0x7feb352du; u3 = u3 * 0x7feb352du; }
        if (MODE == 5) { u0 = __umul24(u0, 0x352du) + 1; u1 = __umul24(u1, 0x352du) + 1; u2 = __umul24(u2, 0x352du) + 1; u3 = __umul24(u3, 0x352du) + 1; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(u0 ^ u1 ^ u2 ^ u3);
}
template <int MODE> int run(const char* name, int ops_per_iter) {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, out, 100);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_ops = (double)256 * 8 * 256 * iters * ops_per_iter;
    printf("%-14s %8.2f ms  %7.1f lane-ops/ns/chip = %5.1f lane-ops per clock per CU at 2.1 GHz\n", name, ms, lane_ops / (ms * 1e6), lane_ops / (ms * 1e6) / 256 / 2.1);
    hipFree(out);
    return 0;
}
int main() {
    run<0>("v_fma_f32", 8); run<1>("v_pk_fma_f32", 8); run<2>("v_exp_f32", 8); run<3>("v_rcp_f32", 8); run<4>("v_mul_lo_u32", 4); run<5>("v_mul_u32_u24", 4);
    return 0;
}
