// Stand-alone prototype of a 4-wave NT GEMM for gfx950 (bf16 in, fp32 accumulate, bf16 out):
//   C[M,N] = A[M,K] . B[N,K]^T,  256 x 256 x 32 tile, ONE wave per SIMD (4 waves = 2 x 2, wave tile 128 x 128 = 4 x 4 fragments of
//   v_mfma_f32_32x32x16_bf16, 256 fp32 accumulators per lane), four 32-KiB LDS stages filled by global->LDS DMA three K tiles ahead,
//   fragments of the next k-step gathered between the MFMAs of the current one (software pipeline inside the wave), one barrier per K tile.
// Why: the 8-wave kernel (gemm8p.cpp -> csrc/gemm_dp.hip) reads 192 KiB of fragments from LDS per 256 x 256 x 64 tile product and keeps the
// MFMA pipe ~50 % busy; 128 x 128 wave tiles read 128 KiB, and 32-cycle MFMAs leave 8 issue slots per MFMA for the reads and DMA.
// Built and run by hand:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/gemm4w.cpp -o gpurun_out/gemm4w && gpurun_out/gemm4w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
#include "../../spokennlp_amd/csrc/common.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

struct Args { const bf16_t* A; const bf16_t* B; bf16_t* C; int M, N, K, lda, ldb, ldc, tiles_m, tiles_n; };
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
#define SB() __builtin_amdgcn_sched_barrier(0)
#define W4_STAGE 32768
#define W4_LDS (4 * W4_STAGE)

// LDS stage: A image [256 rows][64 B] at +0, B image at +16384; the 16-B chunk c of row r sits at chunk c ^ ((r >> 2) & 3): conflict-free
// for ds_read_b128 of a 32 x 32 x 16 fragment (lane = row l & 31, chunk = kstep * 2 + (l >> 5)), natural and permuted rows alike.
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int x = l & 31, h = l >> 5;
    const int nwg = a.tiles_m * a.tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    constexpr int GROUP_M = 8;
    const int gsz_full = GROUP_M * a.tiles_n;
    const int grp = t / gsz_full, first_m = grp * GROUP_M;
    const int gmn = min(a.tiles_m - first_m, GROUP_M);
    const int rem = t - grp * gsz_full;
    const int tm = first_m + rem % gmn, tn = rem / gmn;
    const int m0 = tm * 256, n0 = tn * 256;
    // ---- DMA duty of wave w: rows w*64 .. +64 of the A and of the B image, 4 pieces of 16 rows each; lane i of a piece: row i >> 2, slot i & 3
    const int prow = l >> 2, pslot = l & 3, pf = (l >> 4) & 3;                 // (row >> 2) & 3 of the piece row = (l >> 4) & 3
    const uint32_t offA = (uint32_t)(((w * 64 + prow) * a.lda + ((pslot ^ pf) * 8)) * 2);
    const uint32_t offB = (uint32_t)(((w * 64 + prow) * a.ldb + ((pslot ^ pf) * 8)) * 2);
    const bf16_t* gA = a.A + (size_t)m0 * a.lda;
    const bf16_t* gB = a.B + (size_t)n0 * a.ldb;
    const int nk = a.K / 32;
#define W4_DMA_PIECE(j, stage, q) do { \
        if ((q) < 4) amdseg_glds16_saddr(gA + (size_t)(j) * 32 + (size_t)(q) * 16 * a.lda, offA, smem + (stage) * W4_STAGE + (w * 64 + (q) * 16) * 64); \
        else amdseg_glds16_saddr(gB + (size_t)(j) * 32 + (size_t)((q) - 4) * 16 * a.ldb, offB, smem + (stage) * W4_STAGE + 16384 + (w * 64 + ((q) - 4) * 16) * 64); } while (0)
    // ---- fragment addresses inside a stage (byte offsets), one per 16-B chunk value c = 0..3
    // A: row wr*128 + fm*32 + x; B: row wc*128 + fn*32 + perm(x), perm(x = 8q + 4h' + r) = 16 (q >> 1) + 8 h' + 4 (q & 1) + r, so that the
    // accumulator registers i = 0..15 of lane half h' are the columns 16 (i >> 3) + 8 h' + (i & 7) of the fragment: two 32-B runs per row
    const int px = 16 * (x >> 4) + 8 * ((x >> 2) & 1) + 4 * ((x >> 3) & 1) + (x & 3);
    uint32_t aA[4], aB[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        aA[c] = (uint32_t)(uintptr_t)LDS_PTR(char, smem) + (wr * 128 + x) * 64 + ((c ^ ((x >> 2) & 3)) << 4);
        aB[c] = (uint32_t)(uintptr_t)LDS_PTR(char, smem) + 16384 + (wc * 128 + px) * 64 + ((c ^ ((px >> 2) & 3)) << 4);
    }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[2][4], fb[2][4];
#define W4_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define W4_LDA(buf, f, stage_off, ks) W4_RD(fa[buf][f], aA[(ks) * 2 + h] + (stage_off), (f) * 2048)
#define W4_LDB(buf, f, stage_off, ks) W4_RD(fb[buf][f], aB[(ks) * 2 + h] + (stage_off), (f) * 2048)
#define W4_MF(buf, fm, fn) acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[buf][fn], fa[buf][fm], acc[fm][fn], 0, 0, 0)
#define W4_WAIT_FRAGS(buf) asm volatile("s_waitcnt lgkmcnt(0)" \
        : "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fa[buf][2]), "+v"(fa[buf][3]), "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fb[buf][2]), "+v"(fb[buf][3]) :: "memory")
    // ---- prologue: tiles 0 .. 3 in flight (clamped: the vmcnt arithmetic below is uniform), fragments of tile 0 k-step 0
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int q = 0; q < 8; ++q) W4_DMA_PIECE(min(st, nk - 1), st, q);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < 4; ++f) { W4_LDA(0, f, 0, 0); W4_LDB(0, f, 0, 0); }
    int s_cur = 0;                                          // stage of tile j
    for (int j = 0; j < nk; ++j) {
        const uint32_t so = s_cur * W4_STAGE, so_n = ((s_cur + 1) & 3) * W4_STAGE;
        // ---- k-step 0: MFMAs on buffer 0; the fragments of k-step 1 are requested under the first four
        W4_WAIT_FRAGS(0);
        SB();
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) { W4_MF(0, fm, fn); SB(); if (fm == 0) { W4_LDA(1, fn, so, 1); W4_LDB(1, fn, so, 1); SB(); } }
        }
        // every fragment of tile j is in registers (after the wait); this wave's pieces of tile j + 1 have landed
        W4_WAIT_FRAGS(1);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // outstanding: tiles j+1, j+2, j+3 (8 pieces each) -> tile j+1 done
        SB(); __builtin_amdgcn_s_barrier(); SB();
        // ---- k-step 1: MFMAs on buffer 1; fragments of tile j+1 k-step 0 (complete since the barrier); DMA of tile j+4 into the stage of
        //      tile j (every wave has read it)
        const int jd = min(j + 4, nk - 1);
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) {
#pragma unroll
            for (int fn = 0; fn < 4; ++fn) {
                W4_MF(1, fm, fn); SB();
                if (fm == 0) { W4_LDA(0, fn, so_n, 0); W4_LDB(0, fn, so_n, 0); SB(); }
                if (fm == 1 || fm == 2) { W4_DMA_PIECE(jd, s_cur, (fm - 1) * 4 + fn); SB(); }
            }
        }
        s_cur = (s_cur + 1) & 3;
    }
    W4_WAIT_FRAGS(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- epilogue: lane (x, h) of fragment (fm, fn) owns row wr*128 + fm*32 + x and the columns wc*128 + fn*32 + 16 (i >> 3) + 8 h + (i & 7)
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
        bf16_t* crow = a.C + (size_t)(m0 + wr * 128 + fm * 32 + x) * a.ldc + n0 + wc * 128 + 8 * h;
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x16 v = acc[fm][fn];
                uint4 pk;
                pk.x = pack2bf(v[half * 8 + 0], v[half * 8 + 1]); pk.y = pack2bf(v[half * 8 + 2], v[half * 8 + 3]);
                pk.z = pack2bf(v[half * 8 + 4], v[half * 8 + 5]); pk.w = pack2bf(v[half * 8 + 6], v[half * 8 + 7]);
                *reinterpret_cast<uint4*>(crow + fn * 32 + half * 16) = pk;
            }
    }
}

static float bf2f_h(bf16_t h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf_h(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int shapes[][3] = {{16384, 768, 3072}, {16384, 2304, 768}, {16384, 3072, 768}, {16384, 768, 768}, {16384, 768, 2304}, {8192, 7680, 8192}};
    const int reps = argc > 1 ? atoi(argv[1]) : 30;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS));
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
        srand(1);
        for (auto& v : hA) v = f2bf_h((rand() % 2001 - 1000) / 1000.0f);
        for (auto& v : hB) v = f2bf_h((rand() % 2001 - 1000) / 1000.0f);
        bf16_t *A, *B, *C;
        CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
        CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(C, 0, (size_t)M * N * 2));
        Args a{A, B, C, M, N, K, K, K, N, M / 256, N / 256};
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm4w_kernel, dim3(a.tiles_m * a.tiles_n), dim3(256), W4_LDS, 0, a);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm4w_kernel, dim3(a.tiles_m * a.tiles_n), dim3(256), W4_LDS, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        std::vector<bf16_t> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0; int bad = 0;
        for (int s = 0; s < 600; ++s) {
            const int m = rand() % M, n = rand() % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf2f_h(hA[(size_t)m * K + k]) * bf2f_h(hB[(size_t)n * K + k]);
            const double got = bf2f_h(hC[(size_t)m * N + n]);
            const double err = fabs(got - ref);
            if (err > 0.02 * fabs(ref) + 0.08) ++bad;
            if (err > maxerr) maxerr = err;
        }
        printf("gemm4w M=%d N=%d K=%d: %.1f us  %.0f TF  (tiles %d)  maxerr %.4f bad %d/600\n", M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12,
               a.tiles_m * a.tiles_n, maxerr, bad);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
