// Stand-alone prototype of the deep-pipeline TN GEMM (weight gradients): C[N,K'] (+)= sum_m A[m,N] . B[m,K'], bf16 in, fp32 out.
// 256(N) x 128(K') tile, 8 waves = 2 N-groups x 4 K'-waves (wave tile 128 x 32), v_mfma_f32_16x16x32_bf16 with BOTH operands
// gathered k(=m)-strided from [64 m][64] LDS tile images by ds_read_b64_tr_b16, 3-stage 48-KiB ring filled by DMA two K tiles
// ahead, one phase per K tile (2 barriers), groups staggered by one barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include "../../spokennlp_amd/csrc/common.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

struct Args { const bf16_t* A; const bf16_t* B; float* C; int M, N, Kp, lda, ldb, ldc, tiles_k, accumulate; };

// swizzle under which the transposed b64 gathers are conflict free (b128 fragment reads are not: see tile64.h)
__device__ __forceinline__ int swz_tr(int r) { return (((r >> 1) & 3) << 1) | ((r >> 3) & 1); }
__device__ __forceinline__ void glds16(const void* g, void* lds) { __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(void, lds), 16, 0, 0); }
// lane (i16 = l&15, g = l>>4) receives tile[r0 + j][col0 + i16] (j < 4) and tile[r0 + 16 + j - 4][col0 + i16] (j >= 4): with
// r0 = kk*32 + g*4 a 32-lane group of one ds_read_b64_tr_b16 touches 8 CONSECUTIVE rows x 32 B, which swz_tr spreads over all
// 64 banks (rows r0 = g*8 .. as k-slots gave 2-way conflicts: rows r and r+8 share a 32-B span).  Any k-slot assignment is
// legal as long as both MFMA operands use the same one.
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int r0, int col0, int l) {
    const int i16 = l & 15;
    const int c = (col0 >> 3) + ((i16 & 3) >> 1);
    bf16x8 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = r0 + h * 16 + (i16 >> 2);
        bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4, tile + row * 128 + ((c ^ swz_tr(row)) << 4) + (i16 & 1) * 8));
        f[h * 4 + 0] = v[0]; f[h * 4 + 1] = v[1]; f[h * 4 + 2] = v[2]; f[h * 4 + 3] = v[3];
    }
    return f;
}

#define STG 49152
__global__ __launch_bounds__(512, 1) void gemm_tn_dp_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3, wq = w & 3;
    const int g = l >> 4, i16 = l & 15;
    const int tn = blockIdx.x / a.tiles_k, tk = blockIdx.x % a.tiles_k;
    const int n0 = tn * 256, k0 = tk * 128;
#define TILE_A(s, i) (smem + (s) * STG + (i) * 8192)
#define TILE_B(s, j) (smem + (s) * STG + 32768 + (j) * 8192)
    int offA[4], offB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wq * 2 + q) * 8 + (l >> 3), c = (l & 7) ^ swz_tr(r);
        offA[q] = r * a.lda + (wr * 2) * 64 + c * 8;
        offA[2 + q] = r * a.lda + (wr * 2 + 1) * 64 + c * 8;
        offB[q] = r * a.ldb + wr * 64 + c * 8;
    }
    const bf16_t* pA = a.A + n0;
    const bf16_t* pB = a.B + k0;
#define DMA(s, kt) do { const bf16_t* ba = pA + (size_t)(kt) * 64 * a.lda; const bf16_t* bb = pB + (size_t)(kt) * 64 * a.ldb; \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q) \
            glds16(ba + offA[i * 2 + q], TILE_A(s, wr * 2 + i) + (wq * 2 + q) * 1024); \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) glds16(bb + offB[q], TILE_B(s, wr) + (wq * 2 + q) * 1024); } while (0)
    f32x4 acc[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = a.M / 64;
    DMA(0, 0);
    if (nk > 1) { DMA(1, 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    int s = 0, s2 = 2;                                      // stage of K tile kt, stage of K tile kt + 2
    for (int kt = 0; kt < nk; ++kt) {
        bf16x8 fa[8][2], fb[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int e = 0; e < 2; ++e) fb[e][kk] = frag_tr(TILE_B(s, wc >> 1), kk * 32 + g * 4, (wc & 1) * 32 + e * 16, l);
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) fa[nf][kk] = frag_tr(TILE_A(s, wr * 2 + (nf >> 2)), kk * 32 + g * 4, (nf & 3) * 16, l);
        }
        if (kt + 2 < nk) DMA(s2, kt + 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                for (int e = 0; e < 2; ++e) acc[nf][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[e][kk], fa[nf][kk], acc[nf][e], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        s = s == 2 ? 0 : s + 1; s2 = s2 == 2 ? 0 : s2 + 1;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    // epilogue: lane owns row n = nf*16 + i16, columns e*16 + g*4 .. +4 of the wave tile
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) {
        float* crow = a.C + (size_t)(n0 + wr * 128 + nf * 16 + i16) * a.ldc + k0 + wc * 32 + g * 4;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float4 v = make_float4(acc[nf][e][0], acc[nf][e][1], acc[nf][e][2], acc[nf][e][3]);
            float4* p = reinterpret_cast<float4*>(crow + e * 16);
            if (a.accumulate) { const float4 o = *p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *p = v;
        }
    }
}

static float bf2f(bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

int main() {
    const int M = 16384;
    int shapes[][2] = {{768, 3072}, {3072, 768}, {2304, 768}, {768, 768}};
    for (auto& sh : shapes) {
        const int N = sh[0], Kp = sh[1];
        std::vector<bf16_t> hA((size_t)M * N), hB((size_t)M * Kp);
        srand(2);
        for (auto& v : hA) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
        for (auto& v : hB) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 2.f);
        bf16_t *A, *B; float* C;
        CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C, (size_t)N * Kp * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        Args a{A, B, C, M, N, Kp, N, Kp, Kp, Kp / 128, 0};
        const int lds = 3 * STG, tiles = (N / 256) * (Kp / 128);
        CK(hipFuncSetAttribute((const void*)gemm_tn_dp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_tn_dp_kernel, dim3(tiles), dim3(512), lds, 0, a);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_tn_dp_kernel, dim3(tiles), dim3(512), lds, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        std::vector<float> hC((size_t)N * Kp);
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0; int bad = 0;
        for (int t = 0; t < 300; ++t) {
            const int n = rand() % N, k = rand() % Kp;
            double ref = 0;
            for (int m = 0; m < M; ++m) ref += (double)bf2f(hA[(size_t)m * N + n]) * bf2f(hB[(size_t)m * Kp + k]);
            const double err = fabs(hC[(size_t)n * Kp + k] - ref);
            if (err > 1e-3 * fabs(ref) + 2e-2) ++bad;
            if (err > maxerr) maxerr = err;
        }
        printf("TN M=%d N=%d K'=%d: %.1f us  %.0f TF  (tiles %d of 256 CUs)  maxerr %.4f bad %d/300\n", M, N, Kp, ms * 1e3,
               2.0 * M * N * Kp / (ms * 1e-3) / 1e12, tiles, maxerr, bad);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
