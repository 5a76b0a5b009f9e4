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
#include <cstdint>
#include "../../spokennlp_amd/csrc/common.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

typedef int i32x2 __attribute__((ext_vector_type(2)));
// wave-uniform pointer the loop optimiser cannot turn into per-lane 64-bit induction variables (keeps the DMA addresses as
// SGPR base + 32-bit lane offset: 6 VGPRs instead of 24)
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const void*)(((uint64_t)hi << 32) | lo);
}
struct Args { const bf16_t* A; const bf16_t* B; float* C; int M, N, Kp, lda, ldb, ldc, tiles_k, accumulate, remap, ntiles; };
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

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
#ifdef NO_READ
#define RD(x) ((bf16x8){(short)kt, (short)l, 1, 2, 3, 4, 5, 6})
#else
#define RD(x) (x)
#endif
#ifdef NO_MFMA
#define MF(c, x, y) do { c[0] += (float)x[0] * (float)y[0]; } while (0)
#else
#define MF(c, x, y) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0)
#endif
#ifdef NO_DMA
#define DMA_ON false
#else
#define DMA_ON true
#endif
#ifdef HOT
#define HOTMASK(k) ((k) & 3)
#else
#define HOTMASK(k) (k)
#endif
// 8 waves = 2 K-halves (kh: rows kh*32..+32 of every 64-token K tile; also the stagger group) x 2 N-halves x 2 K'-halves, wave
// tile 128 x 64 over HALF the K tile: 24 transposed gathers per 32 MFMAs (the 128 x 32 full-K wave tile needed 40 and was LDS
// bound: 208 KiB of LDS traffic per K tile = 1664 clk vs 1024 clk of MFMA); the two K-halves are summed through LDS once at the end.
__global__ __launch_bounds__(512, 1) void gemm_tn_dp_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dr = w >> 2, wq = w & 3;                       // DMA duty: images 2dr, 2dr+1 of A and dr of B, 8-row chunks 2wq, 2wq+1
    const int kh = w >> 2, wr = (w >> 1) & 1, wc = w & 1;    // compute role
    const int g = l >> 4, i16 = l & 15;
    const int bt = a.remap ? xcd_remap(blockIdx.x, a.ntiles) : blockIdx.x;
    const int tn = bt / a.tiles_k, tk = bt % a.tiles_k;
    const int n0 = tn * 256, k0 = tk * 128;
#define TILE_A(s, i) (smem + (s) * STG + (i) * 8192)
#define TILE_B(s, j) (smem + (s) * STG + 32768 + (j) * 8192)
    int offA[4], offB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wq * 2 + q) * 8 + (l >> 3), c = (l & 7) ^ swz_tr(r);
        offA[q] = r * a.lda + (dr * 2) * 64 + c * 8;
        offA[2 + q] = r * a.lda + (dr * 2 + 1) * 64 + c * 8;
        offB[q] = r * a.ldb + dr * 64 + c * 8;
    }
    const bf16_t* pA = a.A + n0;
    const bf16_t* pB = a.B + k0;
#define DMA(s, kt_) do { const int kq_ = HOTMASK(kt_); const bf16_t* ba = (const bf16_t*)uniform_ptr(pA + (size_t)(kq_) * 64 * a.lda); const bf16_t* bb = (const bf16_t*)uniform_ptr(pB + (size_t)(kq_) * 64 * a.ldb); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q) \
            glds16(ba + offA[i * 2 + q], TILE_A(s, dr * 2 + i) + (wq * 2 + q) * 1024); \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) glds16(bb + offB[q], TILE_B(s, dr) + (wq * 2 + q) * 1024); } while (0)
#define DMA_PIECE(s, kt_, j) do { if ((j) < 4) glds16((const bf16_t*)uniform_ptr(pA + (size_t)(HOTMASK(kt_)) * 64 * a.lda) + offA[j], TILE_A(s, dr * 2 + ((j) >> 1)) + (wq * 2 + ((j) & 1)) * 1024); \
        else glds16((const bf16_t*)uniform_ptr(pB + (size_t)(HOTMASK(kt_)) * 64 * a.ldb) + offB[(j) - 4], TILE_B(s, dr) + (wq * 2 + ((j) - 4)) * 1024); } while (0)
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = a.M / 64;
    // software pipeline inside every wave: the fragments of K tile kt+1 are gathered BETWEEN the MFMAs of K tile kt (a wave can
    // only issue one ds_read_b64_tr per ~16 clk, so a separate load phase of 48 gathers costs ~770 clk; beside MFMAs they are free)
    DMA(0, 0);
    if (nk > 1) DMA(1, 1);
    if (nk > 2) DMA(2, 2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // The gathers are issued as inline asm: the compiler puts an s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 BUILTIN that
    // follows an LDS-DMA (it cannot prove they do not alias), which drains the whole DMA ring once per K tile.  The waits for
    // these asm reads are therefore explicit: lgkmcnt(0) at the top of every K tile.
    uint32_t laA[4], laB[4];
    {
        const int row = kh * 32 + g * 4 + (i16 >> 2);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int c = c4 * 2 + ((i16 & 3) >> 1);
            const uint32_t o = (uint32_t)(uintptr_t)LDS_PTR(char, smem) + row * 128 + ((c ^ swz_tr(row)) << 4) + (i16 & 1) * 8;
            laA[c4] = o + wr * 2 * 8192;
            laB[c4] = o + 32768 + wc * 8192;
        }
    }
    i32x2 fa[8][2], fb0[4][2], fb1[4][2];
#define RD_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LDA(nf) do { RD_TR(fa[nf][0], aA[(nf) & 3], ((nf) >> 2) * 8192); RD_TR(fa[nf][1], aA[(nf) & 3], ((nf) >> 2) * 8192 + 2048); } while (0)
#define LDB(e, FB) do { RD_TR(FB[e][0], aB[e], 0); RD_TR(FB[e][1], aB[e], 2048); } while (0)
#define CAT(x) __builtin_bit_cast(bf16x8, __builtin_shufflevector(x[0], x[1], 0, 1, 2, 3))
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF1(nf, e, FB) acc[nf][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(CAT(FB[e]), CAT(fa[nf]), acc[nf][e], 0, 0, 0)
    uint32_t aA[4], aB[4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) { aA[c4] = laA[c4]; aB[c4] = laB[c4]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) LDB(e, fb0);
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) LDA(nf);
    int sc = 0, sn = 1, sp = 2;                             // stages of K tiles kt, kt+1, kt+2
// every fragment register is tied through the wait, so nothing the compiler derives from a gathered value (packing, copies) can be
// placed before the data has arrived
#define WAIT_FRAGS(FB) asm volatile("s_waitcnt lgkmcnt(0)" \
        : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]), \
          "+v"(fa[4][0]), "+v"(fa[4][1]), "+v"(fa[5][0]), "+v"(fa[5][1]), "+v"(fa[6][0]), "+v"(fa[6][1]), "+v"(fa[7][0]), "+v"(fa[7][1]), \
          "+v"(FB[0][0]), "+v"(FB[0][1]), "+v"(FB[1][0]), "+v"(FB[1][1]), "+v"(FB[2][0]), "+v"(FB[2][1]), "+v"(FB[3][0]), "+v"(FB[3][1]) :: "memory")
#define BODY(FC, FN) do { \
        WAIT_FRAGS(FC); \
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        SB(); __builtin_amdgcn_s_barrier(); SB(); \
        const bool dma_ = DMA_ON && kt + 3 < nk; \
        _Pragma("unroll") for (int c4 = 0; c4 < 4; ++c4) { aA[c4] = laA[c4] + sn * STG; aB[c4] = laB[c4] + sn * STG; } \
        __builtin_amdgcn_s_setprio(1); \
        _Pragma("unroll") for (int nf = 0; nf < 4; ++nf) { \
            MF1(nf, 0, FC); SB(); MF1(nf, 1, FC); SB(); LDB(nf, FN); SB(); MF1(nf, 2, FC); SB(); MF1(nf, 3, FC); SB(); LDA(nf); SB(); \
            if (nf >= 1 && dma_) { DMA_PIECE(sc, kt + 3, nf - 1); SB(); } } \
        _Pragma("unroll") for (int nf = 4; nf < 8; ++nf) { \
            MF1(nf, 0, FC); MF1(nf, 1, FC); MF1(nf, 2, FC); MF1(nf, 3, FC); SB(); LDA(nf); SB(); \
            if (nf <= 6 && dma_) { DMA_PIECE(sc, kt + 3, nf - 1); SB(); } } \
        __builtin_amdgcn_s_setprio(0); \
        sc = sc == 2 ? 0 : sc + 1; sn = sn == 2 ? 0 : sn + 1; } while (0)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        BODY(fb0, fb1);
        ++kt; BODY(fb1, fb0); --kt;
    }
    if (kt < nk) BODY(fb0, fb1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // K-half exchange through LDS: wave (kh, wr, wc) keeps row fragments nf = kh*4 .. +4 and hands the other four to its partner
    {
        f32x4* xch = reinterpret_cast<f32x4*>(smem) + (size_t)w * 16 * 64;
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) xch[(i * 4 + e) * 64 + l] = acc[4 + i][e];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) { xch[(i * 4 + e) * 64 + l] = acc[i][e]; acc[i][e] = acc[4 + i][e]; }
        }
        __syncthreads();
        const f32x4* got = reinterpret_cast<const f32x4*>(smem) + (size_t)(w ^ 4) * 16 * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* crow = a.C + (size_t)(n0 + wr * 128 + (kh * 4 + i) * 16 + i16) * a.ldc + k0 + wc * 64 + g * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 o = got[(i * 4 + e) * 64 + l];
                const f32x4 m = acc[i][e];
                float4 v = make_float4(m[0] + o[0], m[1] + o[1], m[2] + o[2], m[3] + o[3]);
                float4* p = reinterpret_cast<float4*>(crow + e * 16);
                if (a.accumulate) { const float4 c = *p; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
                *p = v;
            }
        }
    }
}

static float bf2f(bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

int main() {
    // {N, K', remap, accumulate-run, M}: the last two rows emulate a stream-K split of the encoder layer's 216 tiles over 256 CUs
    // (216 tiles x 256 K tiles  vs  256 tiles x 216 K tiles: what an ideal split could reach, no fix-up cost)
    int shapes[][5] = {{768, 3072, 1, 0, 16384}, {2304, 1536, 1, 0, 16384}, {3072, 2048, 1, 0, 16384}, {4096, 2048, 1, 0, 16384},
                       {2304, 3072, 1, 0, 16384}, {4096, 2048, 1, 0, 13824}};
    for (auto& sh : shapes) {
        const int N = sh[0], Kp = sh[1], M = sh[4];
        std::vector<bf16_t> hA((size_t)M * N), hB((size_t)M * Kp);
        srand(2);
        for (auto& v : hA) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
        for (auto& v : hB) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 2.f);
        bf16_t *A, *B; float* C;
        CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C, (size_t)N * Kp * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        const int lds = 3 * STG, tiles = (N / 256) * (Kp / 128);
        Args a{A, B, C, M, N, Kp, N, Kp, Kp, Kp / 128, 0, sh[2], tiles};
        CK(hipFuncSetAttribute((const void*)gemm_tn_dp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_tn_dp_kernel, dim3(tiles), dim3(512), lds, 0, a);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_tn_dp_kernel, dim3(tiles), dim3(512), lds, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        if (sh[3]) { a.accumulate = 1; CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_tn_dp_kernel, dim3(tiles), dim3(512), lds, 0, a);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float m2; CK(hipEventElapsedTime(&m2, e0, e1)); printf("  accumulate: %.1f us\n", m2 / reps * 1e3); a.accumulate = 0;
            hipLaunchKernelGGL(gemm_tn_dp_kernel, dim3(tiles), dim3(512), lds, 0, a); CK(hipDeviceSynchronize()); }
        std::vector<float> hC((size_t)N * Kp);
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0; int bad = 0; int hist[16] = {0}, histk[8] = {0};
        for (int t = 0; t < 300; ++t) {
            const int n = rand() % N, k = rand() % Kp;
            double ref = 0;
            for (int m = 0; m < M; ++m) ref += (double)bf2f(hA[(size_t)m * N + n]) * bf2f(hB[(size_t)m * Kp + k]);
            const double err = fabs(hC[(size_t)n * Kp + k] - ref);
            if (err > 1e-3 * fabs(ref) + 2e-2) { ++bad; hist[(n % 256) / 16]++; histk[(k % 128) / 16]++; }
            if (err > maxerr) maxerr = err;
        }
        printf("TN M=%d N=%d K'=%d remap %d: %.1f us  %.0f TF  (tiles %d of 256 CUs)  maxerr %.4f bad %d/300\n", M, N, Kp, sh[2], ms * 1e3,
               2.0 * M * N * Kp / (ms * 1e-3) / 1e12, tiles, maxerr, bad);
        if (bad) { printf("  bad by n16:"); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("  by k16:"); for (int i = 0; i < 8; ++i) printf(" %d", histk[i]); printf("\n"); }
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
