// Stand-alone prototype of a deep-pipelined NT GEMM for gfx950 (bf16 in, fp32 accumulate, bf16 out):
//   (variant of gemm8p.cpp on v_mfma_f32_32x32x16_bf16: wave tile 128 x 64 = 4 x 2 fragments of 32 x 32, BN = 256 only)
//   C[M,N] = A[M,K] . B[N,K]^T,  256 x BN x 64 tile, 8 waves (2 row groups x 4 column waves),
//   two 64-KiB LDS stages, 4 phases per K tile, the two row groups staggered by one barrier so that one group's
//   fragment reads run under the other's MFMAs, global->LDS DMA issued 1.4 K tiles ahead with counted vmcnt.
// Built and run by hand:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/gemm8p.cpp -o gpurun_out/gemm8p && gpurun_out/gemm8p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
#include "../../spokennlp_amd/csrc/common.h"
#include "../../spokennlp_amd/csrc/tile64.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct Args { const bf16_t* A; const bf16_t* B; bf16_t* C; int M, N, K, lda, ldb, ldc, tiles_m, tiles_n; unsigned long long* dbg; };

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// swizzle for [64][64] bf16 tiles read as 16x16x32 MFMA fragments with ds_read_b128 (lane = row l&15, 16-B chunk kk*4 + (l>>4)):
// the instruction is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... = 8 rows at chunk c plus the other 8
// rows at chunk c^1; chunk ^ f((row>>1)&7) with f(p) = p ^ (p in {2,3,4,5}) makes the 16 lanes of a group hit 16 distinct 16-B slots
__device__ __forceinline__ int swz2(int r) { const int p = (r >> 1) & 7; return p ^ (((p + 2) >> 2) & 1); }
__device__ __forceinline__ bf16x8 g_frag(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ swz2(r)) << 4));
}
#define BARRIER() __builtin_amdgcn_s_barrier()
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

template <int BN>
__global__ __launch_bounds__(512, 1) void gemm8p_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NBT = BN / 64;                       // B tile64s per stage
    constexpr int STAGE = (4 + NBT) * 8192;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3, wq = w & 3;     // wq: wave index inside its group (DMA duty)
    const int g = l >> 4, i16 = l & 15; (void)g; (void)i16;
    const int x = l & 31, h = l >> 5;
    const int px = 16 * (x >> 4) + 8 * ((x >> 2) & 1) + 4 * ((x >> 3) & 1) + (x & 3);     // B row read by operand lane x: accumulator i <-> column 16 (i >> 3) + 8 h + (i & 7)
    const int nwg = a.tiles_m * a.tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    constexpr int GROUP_M = 8;
    const int gsz_full = GROUP_M * a.tiles_n;
    const int grp = t / gsz_full, first_m = grp * GROUP_M;
    const int gmn = min(a.tiles_m - first_m, GROUP_M);
    const int rem = t - grp * gsz_full;
    const int tm = first_m + rem % gmn, tn = rem / gmn;
    const int m0 = tm * 256, n0 = tn * BN;
#define TILE_A(s, i) (smem + (s) * STAGE + (i) * 8192)
#define TILE_B(s, i) (smem + (s) * STAGE + (4 + (i)) * 8192)
    const bf16_t* pA = a.A + (size_t)(m0 + wr * 128) * a.lda;       // this group's A half
    const bf16_t* pB = a.B + (size_t)n0 * a.ldb;
    // DMA duty of a group: its own A half (tile64s 2wr, 2wr+1) and B tile64s {2wr, 2wr+1} (BN = 256) / {0,1} | {2} (BN = 192).
    // Lane offsets are computed once; a K tile only moves the wave-uniform base (saddr + 32-bit voffset loads).
    int offA[4], offB[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wq * 2 + q) * 8 + (l >> 3), c = (l & 7) ^ swz2(r);
            offA[i * 2 + q] = (i * 64 + r) * a.lda + c * 8;
            offB[i * 2 + q] = (i * 64 + r) * a.ldb + c * 8;
        }
    const bf16_t* pBg = pB + (size_t)((NBT == 4 || wr == 0) ? wr * 128 : 128) * a.ldb;     // this group's B rows
    const int bt0 = (NBT == 4 || wr == 0) ? wr * 2 : 2;
    auto dma = [&](int s, int kt) {
        const bf16_t* ba = pA + kt * 64;
        const bf16_t* bb = pBg + kt * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) at_glds16(ba + offA[i * 2 + q], TILE_A(s, wr * 2 + i) + (wq * 2 + q) * 1024);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (NBT == 3 && wr == 1 && i == 1) break;
#pragma unroll
            for (int q = 0; q < 2; ++q) at_glds16(bb + offB[i * 2 + q], TILE_B(s, bt0 + i) + (wq * 2 + q) * 1024);
        }
    };
    constexpr int NF = BN / 64;                        // 16-column fragments per wave: 4 (BN = 256) or 3 (BN = 192)
    static_assert(BN == 256, "32x32 variant: BN = 256 only");
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nk = a.K / 64;
    dma(0, 0);
    if (nk > 1) dma(1, 1);
    if (nk > 1) { if (NBT == 4 || wr == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BARRIER();
    if (wr == 1) BARRIER();                            // stagger: group 1 runs one barrier behind group 0
    bf16x8 fa[2][4], fb[2][4];
#define LOAD_A(s, hph) _Pragma("unroll") for (int f = 0; f < 2; ++f) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) \
        fa[f][ks] = g_frag(TILE_A(s, wr * 2 + (hph)), f * 32 + x, ks * 2 + h);
#define LOAD_B(s) _Pragma("unroll") for (int e = 0; e < 2; ++e) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) \
        fb[e][ks] = g_frag(TILE_B(s, wc), e * 32 + px, ks * 2 + h);
#define MFMA_H(ah) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) _Pragma("unroll") for (int f = 0; f < 2; ++f) \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) \
        acc[(ah) * 2 + f][e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[e][ks], fa[f][ks], acc[(ah) * 2 + f][e], 0, 0, 0);
#ifdef TIMERS
    unsigned long long tL = 0, tB1 = 0, tM = 0, tB2 = 0, tLg = 0, tVm = 0, t0 = __builtin_readcyclecounter(), t1;
#define TICK(acc) do { t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1; } while (0)
#else
#define TICK(acc) do {} while (0)
#endif
#define PHASE_MID() do { SCHED_FENCE(); TICK(tL); BARRIER(); TICK(tB1); SCHED_FENCE(); __builtin_amdgcn_s_setprio(1); } while (0)
#define PHASE_END() do { __builtin_amdgcn_s_setprio(0); SCHED_FENCE(); TICK(tM); BARRIER(); TICK(tB2); SCHED_FENCE(); } while (0)
#define WAIT_DMA() do { if (NBT == 4 || wr == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } while (0)
    // two phases per K tile (4 barriers): P1 = a0 x (b0 | b1), P2 = a1 x (b0 | b1), 32 MFMAs each.
    //   load half of P1: 8 A + 8 B fragment reads of stage s; of P2: 8 A reads + the DMA of K tile kt+2's a0 tile and B tiles into
    //   stage s (their last readers -- this group's P1 and the other group's P1, one slot later -- retired their reads
    //   with an lgkmcnt(0) before the barrier that ends their load half); the a1 tile of kt+2 follows in the next P1.
    for (int kt = 0; kt < nk; ++kt) {
        const int s = kt & 1;
        // ---- phase 1
        LOAD_B(s) LOAD_A(s, 0)
        if (kt >= 1 && kt + 1 < nk) {                 // a1 tile of K tile kt+1 into the other stage (read last in P2 of kt-1)
#pragma unroll
            for (int q = 0; q < 2; ++q) at_glds16(pA + (kt + 1) * 64 + offA[2 + q], TILE_A(s ^ 1, wr * 2 + 1) + (wq * 2 + q) * 1024);
        }
        TICK(tL);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TICK(tLg);
        if (kt >= 1 && kt + 1 < nk) WAIT_DMA(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TICK(tVm);
        PHASE_MID();
        MFMA_H(0)
        PHASE_END();
        // ---- phase 2
        LOAD_A(s, 1)
        if (kt + 2 < nk) {                            // a0 tile + B tiles of K tile kt+2 into this stage
            const bf16_t* ba = pA + (kt + 2) * 64;
            const bf16_t* bb = pBg + (kt + 2) * 64;
#pragma unroll
            for (int q = 0; q < 2; ++q) at_glds16(ba + offA[q], TILE_A(s, wr * 2) + (wq * 2 + q) * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (NBT == 3 && wr == 1 && i == 1) break;
#pragma unroll
                for (int q = 0; q < 2; ++q) at_glds16(bb + offB[i * 2 + q], TILE_B(s, bt0 + i) + (wq * 2 + q) * 1024);
            }
        }
        TICK(tL);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TICK(tLg);
        if (kt + 2 < nk) WAIT_DMA(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TICK(tVm);
        PHASE_MID();
        MFMA_H(1)
        PHASE_END();
    }
    if (wr == 0) BARRIER();                            // group 0 pays back the stagger barrier
#ifdef TIMERS
    if (a.dbg && l == 0) { unsigned long long* d = a.dbg + ((size_t)blockIdx.x * 8 + w) * 4; d[0] = tL + ((tLg) << 32); d[1] = tB1; d[2] = tM; d[3] = tB2 + (tVm << 32); }
#endif
    // ---- epilogue: lane (x, h) of fragment (fm, fn) owns row wr*128 + fm*32 + x and the columns wc*64 + fn*32 + 16 (i >> 3) + 8 h + (i & 7)
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
        bf16_t* crow = a.C + (size_t)(m0 + wr * 128 + fm * 32 + x) * a.ldc + n0 + wc * 64 + 8 * h;
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
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

static float bf2f(bf16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

int main(int argc, char** argv) {
    int shapes[][3] = {{16384, 3072, 3072}, {8192, 7680, 8192}, {16384, 768, 3072}, {16384, 3072, 768}, {4096, 3840, 4096}, {16384, 2304, 768}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        for (int bn = 256; bn >= 256; bn -= 64) {
        if (N % bn) continue;
        std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
        srand(1);
        for (auto& v : hA) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 2.f);
        for (auto& v : hB) v = f2b((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
        bf16_t *A, *B, *C;
        CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
        CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(C, 0, (size_t)M * N * 2));
        unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)(M / 256) * (N / 192 + 1) * 8 * 4 * 8)); CK(hipMemset(dbg, 0, (size_t)(M / 256) * (N / 192 + 1) * 8 * 4 * 8));
        Args a{A, B, C, M, N, K, K, K, N, M / 256, N / bn, dbg};
        const int lds = 2 * (4 + bn / 64) * 8192;
        auto kern = gemm8p_kernel<256>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, 0, a);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        std::vector<bf16_t> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0; int bad = 0;
        for (int s = 0; s < 400; ++s) {
            const int m = rand() % M, n = rand() % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hB[(size_t)n * K + k]);
            const double got = bf2f(hC[(size_t)m * N + n]);
            const double err = fabs(got - ref);
            if (err > 0.02 * fabs(ref) + 0.05) ++bad;
            if (err > maxerr) maxerr = err;
        }
#ifdef TIMERS
        {
            const int nb = a.tiles_m * a.tiles_n;
            std::vector<unsigned long long> h((size_t)nb * 32);
            CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            for (int g = 0; g < 2; ++g) {
                double sacc[6] = {0, 0, 0, 0, 0, 0};
                for (int b = 0; b < nb; ++b) for (int w = g * 4; w < g * 4 + 4; ++w) {
                    const unsigned long long* d = &h[((size_t)b * 8 + w) * 4];
                    sacc[0] += d[0] & 0xffffffffull; sacc[4] += d[0] >> 32; sacc[1] += d[1]; sacc[2] += d[2]; sacc[3] += d[3] & 0xffffffffull; sacc[5] += d[3] >> 32;
                }
                const double nphase = 2.0 * (K / 64) * nb * 4;
                printf("   group %d per phase (clk): issue reads+dma %.0f  lgkmcnt %.0f  vmcnt %.0f  barrier1 %.0f  mfma-half %.0f  barrier2 %.0f\n", g,
                       sacc[0] / nphase, sacc[4] / nphase, sacc[5] / nphase, sacc[1] / nphase, sacc[2] / nphase, sacc[3] / nphase);
            }
        }
#endif
        printf("32x32 BN=%d M=%d N=%d K=%d: %.1f us  %.0f TF  (tiles %d)  maxerr %.4f bad %d/400\n", bn, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12,
               a.tiles_m * a.tiles_n, maxerr, bad);
        hipFree(A); hipFree(B); hipFree(C); hipFree(dbg);
        }
    }
    return 0;
}
