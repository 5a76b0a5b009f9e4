// Stand-alone prototype: NT GEMM with TWO 4-wave workgroups per CU so that one workgroup's epilogue (GELU math, stores) runs under the
// other's main loop.  C[M,N] = A[M,K] . B[N,K]^T (+bias, +GELU with the pre-activation as second output), bf16 in/out, fp32 accumulate.
//   * 128 x 256 tile per 256-thread workgroup (4 column waves, wave tile 128 x 64 = 8 x 4 fragments of v_mfma_f32_16x16x32_bf16,
//     operands swapped: a lane owns one output row), BK = 32, 3-stage ring of 24 KiB (A 128 x 32, B 256 x 32) = 72 KiB -> 2 WG/CU;
//   * in-wave software pipeline (as the TN kernel): the fragments of K step kt+1 are read between the MFMAs of step kt, A fragments
//     reloaded right after their last use, B fragments double buffered; DMA pieces of step kt+3 interleaved; one barrier per K step;
//   * LDS rows are 64 B (32 k): chunk position g ^ p((row >> 2) & 3), p(q) = (-q) & 3, is conflict free for ds_read_b128 fragments;
//   * the B fragment rows are PERMUTED (fragment pair (2j, 2j+1) holds columns g*8 + 0..3 | g*8 + 4..7): a lane then owns 8 consecutive
//     output columns of its row -> 16-byte global stores straight from registers, no LDS staging in the epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include <cmath>
#include "../../spokennlp_amd/csrc/common.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

struct Args { const bf16_t* A; const bf16_t* B; bf16_t* C; bf16_t* C2; const float* bias; int M, N, K, lda, ldb, ldc, tiles_m, tiles_n; };

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
#define HW_STG 24576
#define HW_LDS (3 * HW_STG)
__device__ __forceinline__ int hw_p(int q) { return (4 - q) & 3; }

template <int EPI>   // 0: none, 1: bias, 2: bias + GELU (C2 = pre-activation)
__global__ __launch_bounds__(256, 2) void gemm_hw_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = l >> 4, i16 = l & 15;
    const int nwg = a.tiles_m * a.tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    constexpr int GROUP_M = 16;
    const int gsz_full = GROUP_M * a.tiles_n;
    const int grp = t / gsz_full, first_m = grp * GROUP_M;
    const int gmn = min(a.tiles_m - first_m, GROUP_M);
    const int rem = t - grp * gsz_full;
    const int tm = first_m + rem % gmn, tn = rem / gmn;
    const int m0 = tm * 128, n0 = tn * 256;
    const bf16_t* pA = a.A + (size_t)m0 * a.lda;
    const bf16_t* pB = a.B + (size_t)n0 * a.ldb;
    // DMA duty: A pieces 2w, 2w+1 (16 rows each), B pieces 4w .. 4w+3; lane -> row (l >> 2), LDS chunk position (l & 3)
    uint32_t offA[2], offB[4];
    {
        const int gch = (l & 3) ^ hw_p((l >> 4) & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) offA[j] = (uint32_t)((((2 * w + j) * 16 + (l >> 2)) * a.lda + gch * 8) * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) offB[j] = (uint32_t)((((4 * w + j) * 16 + (l >> 2)) * a.ldb + gch * 8) * 2);
    }
#define HW_PIECE(s, kt, j) do { if ((j) < 2) amdseg_glds16_saddr(pA + (size_t)(kt) * 32, offA[j], smem + (s) * HW_STG + (2 * w + (j)) * 1024); \
        else amdseg_glds16_saddr(pB + (size_t)(kt) * 32, offB[(j) - 2], smem + (s) * HW_STG + 8192 + (4 * w + (j) - 2) * 1024); } while (0)
#define HW_DMA(s, kt) do { _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) HW_PIECE(s, kt, j_); } while (0)
    // fragment lane addresses inside a stage
    const int laA = i16 * 64 + ((g ^ hw_p((i16 >> 2) & 3)) << 4);
    int laB[2];
#pragma unroll
    for (int e1 = 0; e1 < 2; ++e1) {
        const int row = w * 64 + (i16 >> 2) * 8 + e1 * 4 + (i16 & 3);
        laB[e1] = 8192 + row * 64 + ((g ^ hw_p((row >> 2) & 3)) << 4);
    }
#define HW_FA(s, mf) (*reinterpret_cast<const bf16x8*>(smem + (s) * HW_STG + laA + (mf) * 1024))
#define HW_FB(s, e) (*reinterpret_cast<const bf16x8*>(smem + (s) * HW_STG + laB[(e) & 1] + ((e) >> 1) * 2048))
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = a.K / 32;
    HW_DMA(0, 0);
    HW_DMA(1, min(1, nk - 1));
    HW_DMA(2, min(2, nk - 1));
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 fa[8], fb0[4], fb1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) fb0[e] = HW_FB(0, e);
#pragma unroll
    for (int mf = 0; mf < 8; ++mf) fa[mf] = HW_FA(0, mf);
    int sc = 0, sn = 1;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define HW_BODY(FC, FN) do { \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
        SB(); __builtin_amdgcn_s_barrier(); SB(); \
        const int ktd_ = min(kt + 3, nk - 1); \
        __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int mf = 0; mf < 8; ++mf) { \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) acc[mf][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FC[e], fa[mf], acc[mf][e], 0, 0, 0); \
            SB(); \
            if (mf < 4) FN[mf] = HW_FB(sn, mf); \
            fa[mf] = HW_FA(sn, mf); \
            SB(); \
            if (mf >= 1 && mf <= 6) { HW_PIECE(sc, ktd_, mf - 1); SB(); } } \
        __builtin_amdgcn_s_setprio(0); \
        sc = sc == 2 ? 0 : sc + 1; sn = sn == 2 ? 0 : sn + 1; } while (0)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        HW_BODY(fb0, fb1);
        ++kt; HW_BODY(fb1, fb0); --kt;
    }
    if (kt < nk) HW_BODY(fb0, fb1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // ---- epilogue: lane owns row m0 + mf*16 + i16 and columns n0 + w*64 + ep*32 + g*8 .. +8 (ep = fragment pair)
    float bv[2][8];
    if (EPI >= 1) {
#pragma unroll
        for (int ep = 0; ep < 2; ++ep) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.bias + n0 + w * 64 + ep * 32 + g * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(a.bias + n0 + w * 64 + ep * 32 + g * 8 + 4);
            bv[ep][0] = b0.x; bv[ep][1] = b0.y; bv[ep][2] = b0.z; bv[ep][3] = b0.w; bv[ep][4] = b1.x; bv[ep][5] = b1.y; bv[ep][6] = b1.z; bv[ep][7] = b1.w;
        }
    }
#pragma unroll
    for (int mf = 0; mf < 8; ++mf) {
        const size_t row = (size_t)(m0 + mf * 16 + i16);
#pragma unroll
        for (int ep = 0; ep < 2; ++ep) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[mf][2 * ep][r]; v[4 + r] = acc[mf][2 * ep + 1][r]; }
            if (EPI >= 1) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += bv[ep][r];
            }
            const size_t col = (size_t)(n0 + w * 64 + ep * 32 + g * 8);
            if (EPI == 2) {
                uint4 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); pk.z = pack2bf(v[4], v[5]); pk.w = pack2bf(v[6], v[7]);
                *reinterpret_cast<uint4*>(a.C2 + row * a.ldc + col) = pk;
                gelu_act4(v, 0); gelu_act4(v + 4, 0);
            }
            uint4 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); pk.z = pack2bf(v[4], v[5]); pk.w = pack2bf(v[6], v[7]);
            *reinterpret_cast<uint4*>(a.C + row * a.ldc + col) = pk;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

static float bf2f(bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }
static double gelu_ref(double x) { return 0.5 * x * (1.0 + erf(x / sqrt(2.0))); }

template <int EPI> static void run(int M, int N, int K) {
    std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
    std::vector<float> hb(N);
    srand(1);
    for (auto& v : hA) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    for (auto& v : hB) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
    for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f);
    bf16_t *A, *B, *C, *C2; float* bias;
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 2)); CK(hipMalloc(&C2, (size_t)M * N * 2));
    CK(hipMalloc(&bias, N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0, (size_t)M * N * 2));
    Args a{A, B, C, C2, bias, M, N, K, K, K, N, M / 128, N / 256};
    auto kern = gemm_hw_kernel<EPI>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, HW_LDS));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), HW_LDS, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), HW_LDS, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<bf16_t> hC((size_t)M * N), hC2((size_t)M * N);
    CK(hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hC2.data(), C2, hC2.size() * 2, hipMemcpyDeviceToHost));
    double maxerr = 0; int bad = 0;
    for (int s = 0; s < 600; ++s) {
        const int m = rand() % M, n = rand() % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
        if (EPI >= 1) ref += hb[n];
        double want = EPI == 2 ? gelu_ref(ref) : ref;
        double err = fabs(bf2f(hC[(size_t)m * N + n]) - want);
        if (EPI == 2) err = fmax(err, fabs(bf2f(hC2[(size_t)m * N + n]) - ref));
        if (err > 0.02 * fabs(ref) + 0.05) ++bad;
        if (err > maxerr) maxerr = err;
    }
    printf("EPI=%d M=%d N=%d K=%d: %.1f us  %.0f TF  (tiles %d)  maxerr %.4f bad %d/600\n", EPI, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12,
           a.tiles_m * a.tiles_n, maxerr, bad);
    hipFree(A); hipFree(B); hipFree(C); hipFree(C2); hipFree(bias);
}

int main() {
    run<1>(16384, 2304, 768);
    run<2>(16384, 3072, 768);
    run<0>(16384, 3072, 768);
    run<1>(16384, 768, 768);
    run<1>(16384, 768, 3072);
    run<0>(8192, 7680, 8192);
    return 0;
}
