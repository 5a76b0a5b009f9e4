// bf16 MFMA GEMMs for the BERT encoder projections on gfx950 (MI355X).
//
//   gemm_nt : C[M,N] = A[M,K] . B[N,K]^T  (+ fused epilogue)      forward projections and dgrad (with W^T shadows)
//             replaces the torch.nn.Linear calls of [hf] models/bert/modeling_bert.py:175-177 (q,k,v), :282-293
//             (attention output), :325-337 (intermediate + GELU), :340-351 (output) and their autograd backward.
//   gemm_tn : C[N,K'] (+)= sum_m A[m,N] . B[m,K']   grouped launch   weight gradients (dW = dY^T X)
//
// Design (MI355X-first, not a port): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per
// wave, v_mfma_f32_32x32x16_bf16), BK = 64, operands staged HBM -> LDS by direct DMA (global_load_lds, 16 B/lane)
// into a double-buffered 64 KiB LDS image, XOR-swizzled on the SOURCE address so the ds_read_b128 / tr_b16
// fragment reads are bank-conflict free; the k-strided operands of gemm_tn are read with the LDS transpose read
// ds_read_b64_tr_b16 so no transposed activation copies ever touch HBM.  The fp32 accumulators go back through
// LDS so every global store (and every epilogue operand load) is a coalesced 16 B/lane access.  Workgroup ids
// are remapped so that each XCD (private 4 MiB L2) walks a contiguous run of tiles sharing operand panels.
#include "common.h"
#include "amdseg_internal.h"
#include <stdlib.h>

#define BM 128
#define BN 128
#define BK 64
#include "gemm_epi.h"

#define PT_DECL
#define PT_A
#define PT_B
#define PT_C
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(void, lds_wave_base), 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------ gemm_nt
// LDS image of an operand tile: [128 rows][64 k] bf16, 128 B per row, 16-B chunk c of row r stored at slot
// c ^ ((r >> 1) & 7): a ds_read_b128 lane group (16 distinct rows mod 16, one k-chunk) covers all 16 slots of the
// 256-B bank row -> conflict free.
// per-lane element offsets of the 4 DMA pieces a wave contributes to one operand tile (constant over the K loop);
// the K-step only moves the wave-uniform base pointer, so the loads use the saddr + 32-bit voffset form
struct NtLane { int off[4]; };
__device__ __forceinline__ NtLane nt_lane_offsets(int ld, int w, int l) {
    NtLane o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = w * 32 + q * 8 + (l >> 3), s = l & 7;
        o.off[q] = r * ld + ((s ^ ((r >> 1) & 7)) << 3);
    }
    return o;
}
__device__ __forceinline__ void nt_stage(const bf16_t* __restrict__ base, const NtLane& o, char* lds_tile, int w) {
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(base + o.off[q], lds_tile + (w * 32 + q * 8) * 128);
}
__device__ __forceinline__ bf16x8 nt_frag(const char* lds_tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}

// NST = LDS stages of [A 16 KiB | B 16 KiB].  2 (64 KiB): two workgroups per CU hide each other's waits -- the form for grids of more tiles than CUs.
// 4 (128 KiB, round 6): the SMALL-M form.  A grid of at most one tile per CU (bert-base at 8 x 512 tokens: 192 tiles of the N = 768 GEMMs) left every CU with
// ONE 4-wave workgroup whose only prefetch was the next K tile, requested 0.8 us before it was needed while its A panel comes from HBM: 1,790 cycles per K
// tile against 512 of MFMA work (rocprofv3, profiles/r06_smallm_bert8_kernel_stats.md: 38.9 us for N = 768, K = 3072 at M = 4096, 0.2 of the peak).  The ring of
// four keeps three K tiles in flight (LDS-DMA as inline asm + counted vmcnt, as in gemm_dp.hip: the compiler then knows of no vector memory operation it would
// have to drain in front of the fragment reads).  Same MFMA sequence per accumulator: the same bits.
template <int EPIX, typename OutT, int NST>
__global__ __launch_bounds__(256, NST == 2 ? 2 : 1) void gemm_nt_kernel(GemmNTArgs a) {
    constexpr int EPI = EPI_BASE(EPIX); constexpr int ACT = EPI_ACT(EPIX); (void)ACT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave id as a scalar: LDS bases / M0 stay in SGPRs
    const int wr = w >> 1, wc = w & 1;
    const int nwg = a.tiles_m * a.tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    // grouped tile order inside each XCD's contiguous range: the ~64 tiles resident on one XCD (32 CUs x 2) form a
    // GROUP_M x 8 patch, so each A/B panel fetched into the XCD's 4 MiB L2 is shared by 8 tiles and the resident
    // working set (8 + 8 panels) stays below the L2 size
    const int gsz_full = GROUP_M * a.tiles_n;
    const int grp = t / gsz_full, first_m = grp * GROUP_M;
    const int gm = min(a.tiles_m - first_m, GROUP_M);
    const int rem = t - grp * gsz_full;
    const int tm = first_m + rem % gm, tn = rem / gm;
    const int m0 = tm * BM, n0 = tn * BN;
#define bufA(i) (smem + (i) * 32768)
#define bufB(i) (smem + 16384 + (i) * 32768)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = a.K / BK;
    PT_DECL
    const NtLane offA = nt_lane_offsets(a.lda, w, l), offB = nt_lane_offsets(a.ldb, w, l);
    const bf16_t* pA = a.A + (size_t)m0 * a.lda;
    const bf16_t* pB = a.B + (size_t)n0 * a.ldb;
    // NST == 4: the 8 pieces of a K tile (4 A + 4 B per wave) as saddr + lane-offset LDS-DMA with the destination in m0 (common.h)
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem) + (uint32_t)(w * 32) * 128;
#define NT_DMA_TILE(kt_, st_) do { const bf16_t* ga_ = a.A + (size_t)m0 * a.lda + (size_t)(kt_) * BK; const bf16_t* gb_ = a.B + (size_t)n0 * a.ldb + (size_t)(kt_) * BK; \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) amdseg_glds16_saddr_lds(ga_, (uint32_t)offA.off[q] * 2u, lds0 + (st_) * 32768 + q * 1024); \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) amdseg_glds16_saddr_lds(gb_, (uint32_t)offB.off[q] * 2u, lds0 + (st_) * 32768 + 16384 + q * 1024); } while (0)
#define NT_READ(FA, FB, st_) do { const char* tA_ = smem + (st_) * 32768; const char* tB_ = tA_ + 16384; \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { const int c_ = kk * 2 + (l >> 5); \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) FA[kk][i] = nt_frag(tA_, wr * 64 + i * 32 + (l & 31), c_); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) FB[kk][j] = nt_frag(tB_, wc * 64 + j * 32 + (l & 31), c_); } } while (0)
    // operands swapped (B fragment first): D[row = n][col = m], so a lane owns ONE output row m = lane&31 and
    // 4 consecutive n per register quad -> the epilogue stores straight from registers, no LDS round trip
#define NT_MFMA(FA, FB) do { _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[kk][j], FA[kk][i], acc[i][j], 0, 0, 0); } while (0)
    if constexpr (NST == 4) {
        // ---- small-M form: ring of four stages, and the fragments of K tile kt + 1 are READ UNDER the MFMAs of K tile kt (two register sets, the
        // loop unrolled by two): one workgroup per CU = one wave per SIMD has nobody else to cover its LDS latency or its barrier.
        // Iteration kt: [tile kt + 1 landed: counted vmcnt | every fragment read of tile kt retired: lgkmcnt(0) | barrier] -> the stage of tile kt is
        // free (its fragments are in registers in every wave): DMA of tile kt + 4 into it -> 16 reads of tile kt + 1 interleaved with the 16 MFMAs of tile kt.
        bf16x8 fa0[4][2], fb0[4][2], fa1[4][2], fb1[4][2];
        NT_DMA_TILE(0, 0);
        if (nk > 1) NT_DMA_TILE(1, 1);
        if (nk > 2) NT_DMA_TILE(2, 2);
        if (nk > 3) NT_DMA_TILE(3, 3);
        if (nk > 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (nk > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        NT_READ(fa0, fb0, 0);
#define NT_ITER(kt_, CA, CB, NA, NB) do { \
            if ((kt_) + 1 < nk) {                                /* tile kt + 1 has landed; tiles kt + 2, kt + 3 (8 pieces per wave each) stay in flight */ \
                if ((kt_) + 3 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); \
                else if ((kt_) + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
            } \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
            __syncthreads(); \
            if ((kt_) + 4 < nk) NT_DMA_TILE((kt_) + 4, (kt_) & 3); \
            if ((kt_) + 1 < nk) NT_READ(NA, NB, ((kt_) + 1) & 3); \
            NT_MFMA(CA, CB); \
            _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); } \
            __builtin_amdgcn_sched_barrier(0); \
        } while (0)
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) { NT_ITER(kt, fa0, fb0, fa1, fb1); NT_ITER(kt + 1, fa1, fb1, fa0, fb0); }
        if (kt < nk) NT_ITER(kt, fa0, fb0, fa1, fb1);
    } else {
    nt_stage(pA, offA, bufA(0), w);
    nt_stage(pB, offB, bufB(0), w);
    for (int kt = 0; kt < nk; ++kt) {
        PT_A
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            pA += BK; pB += BK;
            nt_stage(pA, offA, bufA(cur ^ 1), w);
            nt_stage(pB, offB, bufB(cur ^ 1), w);
        }
        PT_B
        // all 16 fragment reads of the K-step are issued back to back, the 16 MFMAs then retire behind counted
        // lgkmcnt waits: one exposed LDS latency per K-step instead of four (phase timers: 1510 -> see profiles/)
        bf16x8 fa[4][2], fb[4][2];
        NT_READ(fa, fb, cur);
        NT_MFMA(fa, fb);
        // schedule: 8 reads up front, then one read behind each of the first 8 MFMAs, then the last 8 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        PT_C
    }
    }
    // ---- epilogue.  The accumulator layout gives a lane one row and 4 consecutive columns (8 B of bf16): stored directly,
    // one instruction touches 32 rows x 16 B -- 32 partial cache lines -- and the CU's address path (shared with the
    // other resident workgroup's global->LDS staging) becomes the bottleneck (store ablation: +23 us on the QKV shape,
    // +59 us on the dual-output FFN shape).  bf16 outputs therefore go through a wave-private LDS transpose: 8-B
    // ds_writes in the accumulator layout, 16-B ds_reads in a row-contiguous layout (8 lanes = one 128-B row segment),
    // so every global store instruction writes 8 complete 128-B lines.  16-B chunk c of row r lives at chunk
    // c ^ ((r ^ (r >> 3)) & 7): conflict-free for both access shapes (two 64 x 128-B images per wave = all 64 KiB).
    const int hi = l >> 5;
    float4 bv[2][4];
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[j][q] = *reinterpret_cast<const float4*>(a.bias + n0 + wc * 64 + j * 32 + q * 8 + hi * 4);
    }
    constexpr bool STAGED = sizeof(OutT) == 2;
    constexpr int SROW = 128;
    char* stg = smem + w * (64 * SROW);                           // 8 KiB per wave; a second image for C2 follows
    char* stg2 = smem + 4 * (64 * SROW) + w * (64 * SROW);
#define STG_OFF(r, c16) ((r) * SROW + ((((c16) ^ ((r) ^ ((r) >> 3))) & 7) << 4))
    // (staging the R operand through LDS the same way was measured slower: +13 us on the GELU-backward shape)
    constexpr bool RSTAGED = false;
    uint4 rload[8];
    if (RSTAGED) {                                                // the R tile of this wave, row-contiguous 16-B loads
        const bf16_t* rbase = a.R + (size_t)(m0 + wr * 64) * a.ldr + n0 + wc * 64 + (l & 7) * 8;
#pragma unroll
        for (int p = 0; p < 8; ++p) rload[p] = *reinterpret_cast<const uint4*>(rbase + (size_t)(p * 8 + (l >> 3)) * a.ldr);
    }
    if (STAGED) __syncthreads();                                  // every wave is done reading the K-loop tiles
    if (RSTAGED) {
#pragma unroll
        for (int p = 0; p < 8; ++p) *reinterpret_cast<uint4*>(stg2 + STG_OFF(p * 8 + (l >> 3), l & 7)) = rload[p];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const size_t gm = (size_t)(m0 + wr * 64 + i * 32 + (l & 31));
        uint2 rr[2][4];
        if (EPI == EPI_ADD_RES || EPI == EPI_GELU_BWD) {        // all operand loads of this row block first, then the math
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (RSTAGED) rr[j][q] = *reinterpret_cast<const uint2*>(stg2 + STG_OFF(i * 32 + (l & 31), j * 4 + q) + hi * 8);
                    else rr[j][q] = *reinterpret_cast<const uint2*>(a.R + gm * a.ldr + n0 + wc * 64 + j * 32 + q * 8 + hi * 4);
                }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = n0 + wc * 64 + j * 32 + q * 8 + hi * 4;
                const int soff = STG_OFF(i * 32 + (l & 31), j * 4 + q) + hi * 8;
                float v[4] = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
                    v[0] += bv[j][q].x; v[1] += bv[j][q].y; v[2] += bv[j][q].z; v[3] += bv[j][q].w;
                }
                if (EPI == EPI_BIAS_GELU) {
                    uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
                    if (a.C2) *reinterpret_cast<uint2*>(stg2 + soff) = pk;          // pre-activation u, kept for backward
                    gelu_act4(v, ACT);
                } else if (EPI == EPI_ADD_RES || EPI == EPI_GELU_BWD) {
                    const float r0 = __uint_as_float(rr[j][q].x << 16), r1 = __uint_as_float(rr[j][q].x & 0xffff0000u);
                    const float r2 = __uint_as_float(rr[j][q].y << 16), r3 = __uint_as_float(rr[j][q].y & 0xffff0000u);
                    if (EPI == EPI_ADD_RES) { v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3; }
                    else gelu_grad_mul4(v, r0, r1, r2, r3, ACT);
                }
                if (STAGED) {
                    uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>(stg + soff) = pk;
                } else {
                    OutT* dst = reinterpret_cast<OutT*>(a.C) + gm * a.ldc + gn;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
    }
    if (STAGED) {
        __builtin_amdgcn_wave_barrier();                              // wave-private image: only this wave's ds ops matter
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int rr0 = l >> 3, cc = l & 7;                           // lane -> (row within an 8-row group, 16-B chunk)
        bf16_t* cbase = reinterpret_cast<bf16_t*>(a.C) + (size_t)(m0 + wr * 64) * a.ldc + n0 + wc * 64 + cc * 8;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int r = p * 8 + rr0;
            const uint4 val = *reinterpret_cast<const uint4*>(stg + STG_OFF(r, cc));
            *reinterpret_cast<uint4*>(cbase + (size_t)r * a.ldc) = val;
        }
        if (EPI == EPI_BIAS_GELU && a.C2) {                 // inference passes C2 = NULL: single-output GELU
            bf16_t* c2base = a.C2 + (size_t)(m0 + wr * 64) * a.ldc2 + n0 + wc * 64 + cc * 8;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int r = p * 8 + rr0;
                const uint4 val = *reinterpret_cast<const uint4*>(stg2 + STG_OFF(r, cc));
                *reinterpret_cast<uint4*>(c2base + (size_t)r * a.ldc2) = val;
            }
        }
    }
}

#define g_force_small_tile amdseg_force_small_tile()       // test hook of the call's context (amdseg_ctx_force_small_tile): the 128x128 kernels on big shapes

// ------------------------------------------------------------------------------------------------ gemm_nt, ping-pong 256x192
// Large-shape kernel (M % 256 == 0, N % 192 == 0): one 512-thread workgroup per CU computes a 256 x 192 tile.
// Why this shape: every projection of the encoder has N in {768, 2304, 3072} = {4, 12, 16} x 192 and M = 64 x 256 at the
// bench size, so the tile count is an exact multiple of 256 CUs (no tail), and it moves 0.0091 B/FLOP from L2 instead of
// the 0.0156 of a 128 x 128 tile.
// Why ping-pong: the 8 waves form two groups of 4 (hardware puts wave w and w+4 on the same SIMD).  Group g owns output
// columns [96 g, 96 g + 96); wave q of a group owns rows [64 q, 64 q + 64): 2 x 3 fragments of v_mfma_f32_32x32x16_bf16.
// Every half K-step is a phase ended by s_barrier; in each phase one group only moves data (10 ds_read_b128 fragment
// loads, and once per K-step its share of the global->LDS staging) while the other group only issues MFMAs (12 per wave)
// from registers, then they swap, so each SIMD's matrix pipe is fed by one wave while the other hides memory latency:
//   phase 4t   : G0 stage + mem(t,0)      | G1 mfma(t-1,1)
//   phase 4t+1 : G0 mfma(t,0)             | G1 stage + mem(t,0)
//   phase 4t+2 : G0 mem(t,1)              | G1 mfma(t,0)
//   phase 4t+3 : G0 mfma(t,1)             | G1 mem(t,1)
// Staging: global_load_lds DMA (16 B/lane) into a 2-stage LDS ring (2 x 56 KiB), the stage for K-step t+1 is issued in
// the first data phase of K-step t and awaited (vmcnt(0)) before the last barrier of K-step t.  Measured alternatives
// (profiles/r01_gemm_experiments.md): register staging two K-steps deeper and an L2 "touch" prefetch were both slower.
// LDS image = 128-B rows, XOR-swizzled via the per-lane SOURCE address: SQ_LDS_BANK_CONFLICT = 0.
#define PP_BM 256
#define PP_BN 192
#define PP_A_SLOT 32768                 // A stage: 256 rows x 128 B
#define PP_B_SLOT 24576                 // B stage: 192 rows x 128 B
#define PP_B_BASE (3 * PP_A_SLOT)        // LDS: 3-slot A ring (96 KiB) | 2-slot B ring (48 KiB) = 144 KiB
#define PP_LDS (3 * PP_A_SLOT + 2 * PP_B_SLOT)
#ifndef PP_ABL_NO_DMA
#define PP_ABL_NO_DMA 0
#endif
#ifndef PP_ABL_NO_MFMA
#define PP_ABL_NO_MFMA 0
#endif
#ifndef PP_ABL_NO_READ
#define PP_ABL_NO_READ 0
#endif

struct PpLane { int a[4]; int b[3]; };

// tile t (in XCD-contiguous, GROUP_M-grouped order) -> (m0, n0)
__device__ __forceinline__ void pp_tile_coords(const GemmNTArgs& a, int t_, int& m0, int& n0) {
    const int gsz_full = GROUP_M * a.tiles_n;
    const int gidx = t_ / gsz_full, first_m = gidx * GROUP_M;
    const int gm_ = min(a.tiles_m - first_m, GROUP_M);
    const int rem = t_ - gidx * gsz_full;
    m0 = (first_m + rem % gm_) * PP_BM;
    n0 = (rem / gm_) * PP_BN;
}
// persistent schedule: 256 workgroups (one per CU); workgroup b lives on XCD b & 7 and walks that XCD's contiguous tile
// range 32 tiles per round, so the 32 CUs of an XCD always work on neighbouring tiles (shared A/B panels in its L2)
__device__ __forceinline__ int pp_tile_of(int b, int r, int T) {
    const int x = b & 7, i = b >> 3;
    const int q = T >> 3, rm = T & 7;
    const int start = x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q;
    const int len = q + (x < rm ? 1 : 0);
    const int idx = r * 32 + i;
    return idx < len ? start + idx : -1;
}

// epilogue operands (bias quad or residual / pre-activation tile) are loaded BEFORE the next stage's DMA is queued and the
// stores are issued AFTER it, so (in-order vmcnt) the K-step can wait for its DMA with vmcnt(#stores) and leave the stores
// in flight -- waiting for ~100 KB of stores per tile with vmcnt(0) cost 5.7 us per tile (profiles/r01_gemm_experiments.md)
struct PpEpiRegs { float4 bv[3][4]; uint2 rr[2][3][4]; };
template <int EPIX>
__device__ __forceinline__ void pp_epi_load(const GemmNTArgs& a, PpEpiRegs& e, int m0, int n0, int grp, int wq, int l) {
    constexpr int EPI = EPI_BASE(EPIX); constexpr int ACT = EPI_ACT(EPIX); (void)ACT;
    const int hi = l >> 5;
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) e.bv[j][q] = *reinterpret_cast<const float4*>(a.bias + n0 + grp * 96 + j * 32 + q * 8 + hi * 4);
    }
    if (EPI == EPI_ADD_RES || EPI == EPI_GELU_BWD) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t gm = (size_t)(m0 + wq * 64 + i * 32 + (l & 31));
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    e.rr[i][j][q] = *reinterpret_cast<const uint2*>(a.R + gm * a.ldr + n0 + grp * 96 + j * 32 + q * 8 + hi * 4);
        }
    }
}
template <int EPIX, typename OutT>
__device__ __forceinline__ void pp_epi_store(const GemmNTArgs& a, const PpEpiRegs& e, f32x16 (&acc)[2][3], int m0, int n0, int grp,
                                             int wq, int l) {
    constexpr int EPI = EPI_BASE(EPIX); constexpr int ACT = EPI_ACT(EPIX); (void)ACT;
    const int hi = l >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const size_t gm = (size_t)(m0 + wq * 64 + i * 32 + (l & 31));
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = n0 + grp * 96 + j * 32 + q * 8 + hi * 4;
                float v[4] = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
                    v[0] += e.bv[j][q].x; v[1] += e.bv[j][q].y; v[2] += e.bv[j][q].z; v[3] += e.bv[j][q].w;
                }
                if (EPI == EPI_BIAS_GELU) {
                    uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>(a.C2 + gm * a.ldc2 + gn) = pk;
                    gelu_act4(v, ACT);
                } else if (EPI == EPI_ADD_RES || EPI == EPI_GELU_BWD) {
                    const uint2 r = e.rr[i][j][q];
                    const float r0 = __uint_as_float(r.x << 16), r1 = __uint_as_float(r.x & 0xffff0000u);
                    const float r2 = __uint_as_float(r.y << 16), r3 = __uint_as_float(r.y & 0xffff0000u);
                    if (EPI == EPI_ADD_RES) { v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3; }
                    else gelu_grad_mul4(v, r0, r1, r2, r3, ACT);
                }
                OutT* dst = reinterpret_cast<OutT*>(a.C) + gm * a.ldc + gn;
                if (sizeof(OutT) == 2) {
                    uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>(dst) = pk;
                } else {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}
// number of store instructions pp_epi_store issues per wave
template <int EPIX> struct PpStores { static constexpr int n = (EPI_BASE(EPIX) == EPI_BIAS_GELU) ? 48 : 24; };

template <int EPIX, typename OutT>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(GemmNTArgs a) {
    constexpr int EPI = EPI_BASE(EPIX); constexpr int ACT = EPI_ACT(EPIX); (void)ACT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = w >> 2, wq = w & 3;
    const int T = a.tiles_m * a.tiles_n;
    const int nk = a.K / BK;
    // tiles of this workgroup (persistent: the K-loops of consecutive tiles form ONE software pipeline, so the DMA of the
    // next tile's first stage flies during the last K-step of the current tile and the epilogue stores overlap the
    // next tile's first phases -- no per-tile prologue / re-dispatch bubble)
    int ntl = 0;
    while (pp_tile_of(blockIdx.x, ntl, T) >= 0) ++ntl;
    if (ntl == 0) return;
    const int total = ntl * nk;

    PpLane off;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int r = w * 32 + q * 8 + (l >> 3), sl = l & 7; off.a[q] = r * a.lda + ((sl ^ ((r >> 1) & 7)) << 3); }
#pragma unroll
    for (int q = 0; q < 3; ++q) { const int r = w * 24 + q * 8 + (l >> 3), sl = l & 7; off.b[q] = r * a.ldb + ((sl ^ ((r >> 1) & 7)) << 3); }
    // DMA cursors: (tile round, k) of the NEXT A stage / B stage to fetch.  A (the activation panel, 57 % of the bytes, shared
    // by only tiles_n tiles of an XCD -> 1/4 of its lines miss L2 and pay HBM latency) runs TWO K-steps ahead in a 3-slot
    // ring; B (weights, shared by 8 tiles, L2-resident) one K-step ahead in a 2-slot ring.  144 KiB of the 160 KiB LDS.
    int arA = 0, akA = 0, arB = 0, akB = 0, tm0, tn0;
    pp_tile_coords(a, pp_tile_of(blockIdx.x, 0, T), tm0, tn0);
    const bf16_t* pA = a.A + (size_t)tm0 * a.lda;
    const bf16_t* pB = a.B + (size_t)tn0 * a.ldb;
#define PP_DMA_A(slot)                                                                                       \
    do {                                                                                                     \
        if (!PP_ABL_NO_DMA) {                                                                                \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) glds16(pA + off.a[q], smem + (slot) * PP_A_SLOT + (w * 32 + q * 8) * 128); \
        }                                                                                                    \
        if (++akA == nk) {                                                                                   \
            akA = 0; ++arA;                                                                                  \
            if (arA < ntl) { pp_tile_coords(a, pp_tile_of(blockIdx.x, arA, T), tm0, tn0); pA = a.A + (size_t)tm0 * a.lda; } \
        } else pA += BK;                                                                                     \
    } while (0)
#define PP_DMA_B(slot)                                                                                       \
    do {                                                                                                     \
        if (!PP_ABL_NO_DMA) {                                                                                \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) glds16(pB + off.b[q], smem + PP_B_BASE + (slot) * PP_B_SLOT + (w * 24 + q * 8) * 128); \
        }                                                                                                    \
        if (++akB == nk) {                                                                                   \
            akB = 0; ++arB;                                                                                  \
            if (arB < ntl) { pp_tile_coords(a, pp_tile_of(blockIdx.x, arB, T), tm0, tn0); pB = a.B + (size_t)tn0 * a.ldb; } \
        } else pB += BK;                                                                                     \
    } while (0)

    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[2][2] = {}, fb[2][3] = {};          // fragments of one half K-step: [kk][frag]

    const int rowA = wq * 64 + (l & 31);            // + i*32
    const int rowB = grp * 96 + (l & 31);           // + j*32
    const int hi = l >> 5;
#define PP_MEM(baseA, baseB, h)                                                                              \
    do {                                                                                                     \
        if (PP_ABL_NO_READ) break;                                                                           \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                                   \
            const int c = ((h) * 2 + kk) * 2 + hi;                                                           \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) fa[kk][i] = nt_frag((baseA), rowA + i * 32, c);    \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) fb[kk][j] = nt_frag((baseB), rowB + j * 32, c);    \
        }                                                                                                    \
    } while (0)
#define PP_MFMA()                                                                                            \
    do {                                                                                                     \
        if (PP_ABL_NO_MFMA) { asm volatile("" :: "v"(fa[0][0]), "v"(fa[1][1]), "v"(fb[0][0]), "v"(fb[1][2]), "v"(fb[0][1]), "v"(fb[1][0]), "v"(fa[0][1]), "v"(fa[1][0]), "v"(fb[0][2]), "v"(fb[1][1])); break; } \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                     \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                    \
                _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
    } while (0)
#define PP_SYNC_MEM() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#define PP_SYNC() __builtin_amdgcn_s_barrier()
// end of K-step s: A(s+1) and B(s+1) must have landed; the younger A(s+2) pieces (4) and this K-step's epilogue stores may fly
#define PP_WAIT_STAGE(a2, epi)                                                                               \
    do {                                                                                                     \
        const int S_ = PpStores<EPIX>::n;                                                                     \
        if (epi) {                                                                                           \
            if (a2) { if (S_ == 48) asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); } \
            else { if (S_ == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); } \
        } else {                                                                                             \
            if (a2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        }                                                                                                    \
    } while (0)
    PpEpiRegs er;

    // prologue: A(0), B(0) must land, A(1) stays in flight
    PP_DMA_A(0);
    PP_DMA_B(0);
    if (total > 1) { PP_DMA_A(1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int sa = 0;                          // s % 3

    // compute cursor: tile round cr, k-step ck
    int cr = 0, ck = 0, cm0, cn0;
    pp_tile_coords(a, pp_tile_of(blockIdx.x, 0, T), cm0, cn0);
    int pm0 = cm0, pn0 = cn0;          // tile whose accumulators are complete and await their epilogue
    bool pending = false;

    if (grp == 0) {
        for (int s = 0; s < total; ++s) {
            const char* curA = smem + sa * PP_A_SLOT;
            const char* curB = smem + PP_B_BASE + (s & 1) * PP_B_SLOT;
            // phase 4s: [epilogue of the previous tile] + DMA B(s+1), A(s+2) + mem(s,0)       (G1: mfma(s-1,1))
            const bool epi = pending, a2 = s + 2 < total;
            if (epi) pp_epi_load<EPIX>(a, er, pm0, pn0, grp, wq, l);
            if (s + 1 < total) PP_DMA_B((s + 1) & 1);
            if (a2) PP_DMA_A(sa == 0 ? 2 : sa - 1);            // (s + 2) % 3
            if (epi) { pp_epi_store<EPIX, OutT>(a, er, acc, pm0, pn0, grp, wq, l); pending = false; }
            PP_MEM(curA, curB, 0);
            PP_SYNC_MEM();
            // phase 4s+1: mfma(s,0)
            PP_MFMA();
            PP_SYNC();
            // phase 4s+2: mem(s,1)
            PP_MEM(curA, curB, 1);
            PP_SYNC_MEM();
            // phase 4s+3: mfma(s,1); then make stage s+1 visible
            PP_MFMA();
            PP_WAIT_STAGE(a2, epi);
            PP_SYNC();
            sa = sa == 2 ? 0 : sa + 1;
            if (++ck == nk) {
                ck = 0; pm0 = cm0; pn0 = cn0; pending = true;
                if (++cr < ntl) pp_tile_coords(a, pp_tile_of(blockIdx.x, cr, T), cm0, cn0);
            }
        }
        pp_epi_load<EPIX>(a, er, pm0, pn0, grp, wq, l);
        pp_epi_store<EPIX, OutT>(a, er, acc, pm0, pn0, grp, wq, l);
    } else {
        for (int s = 0; s < total; ++s) {
            const char* curA = smem + sa * PP_A_SLOT;
            const char* curB = smem + PP_B_BASE + (s & 1) * PP_B_SLOT;
            // phase 4s: mfma(s-1,1)
            if (s > 0) PP_MFMA();
            PP_SYNC();
            // phase 4s+1: [epilogue of the previous tile] + DMA B(s+1), A(s+2) + mem(s,0)     (G0: mfma(s,0))
            const bool epi = pending, a2 = s + 2 < total;
            if (epi) pp_epi_load<EPIX>(a, er, pm0, pn0, grp, wq, l);
            if (s + 1 < total) PP_DMA_B((s + 1) & 1);
            if (a2) PP_DMA_A(sa == 0 ? 2 : sa - 1);
            if (epi) { pp_epi_store<EPIX, OutT>(a, er, acc, pm0, pn0, grp, wq, l); pending = false; }
            PP_MEM(curA, curB, 0);
            PP_SYNC_MEM();
            // phase 4s+2: mfma(s,0)
            PP_MFMA();
            PP_SYNC();
            // phase 4s+3: mem(s,1); then make stage s+1 visible
            PP_MEM(curA, curB, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_WAIT_STAGE(a2, epi);
            PP_SYNC();
            sa = sa == 2 ? 0 : sa + 1;
            if (++ck == nk) {
                ck = 0; pm0 = cm0; pn0 = cn0; pending = true;
                if (++cr < ntl) pp_tile_coords(a, pp_tile_of(blockIdx.x, cr, T), cm0, cn0);
            }
        }
        PP_MFMA();                       // trailing mfma(total-1,1)
        pp_epi_load<EPIX>(a, er, pm0, pn0, grp, wq, l);
        pp_epi_store<EPIX, OutT>(a, er, acc, pm0, pn0, grp, wq, l);
    }
}

// the 128 x 128 kernel: two LDS stages when the grid gives the CUs more than one tile each (two co-resident workgroups hide each other's waits), the
// ring of four when it does not (round 6, small M: one workgroup per CU has to cover its own HBM latency)
template <int EPIX, typename OutT>
static int launch_nt_small(const GemmNTArgs& a, hipStream_t s) {
    const int tiles = a.tiles_m * a.tiles_n;
    if (tiles <= amdseg_num_cus()) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<EPIX, OutT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        hipLaunchKernelGGL((gemm_nt_kernel<EPIX, OutT, 4>), dim3(tiles), dim3(256), 4 * 32768, s, a);
    } else
        hipLaunchKernelGGL((gemm_nt_kernel<EPIX, OutT, 2>), dim3(tiles), dim3(256), 2 * 32768, s, a);
    return amdseg_launch_status();
}

template <int EPIX, typename OutT>
static int launch_nt(const GemmNTArgs& a_in, hipStream_t s) {
    constexpr int EPI = EPI_BASE(EPIX); constexpr int ACT = EPI_ACT(EPIX); (void)ACT;
    // tile choice (measured at M = 16384, tools/bench_kernels.py): the 256x192 ping-pong kernel wins for long K (its one
    // workgroup per CU pays an exposed prologue + epilogue per tile), the 128x128 kernel (2 workgroups per CU overlap each
    // other's prologue/epilogue) for K <= 768; shapes the small kernel cannot tile always take the ping-pong kernel
    const bool small_ok = (a_in.M % BM) == 0 && (a_in.N % BN) == 0;
    constexpr int dp_min_k = 768;
    // M % 256 == 0 and N a multiple of 256 (or of 192), K >= 768: the deep-pipeline kernel (gemm_dp.hip).  Measured at M = 16384 against
    // the kernels below: K = 3072 / 2304: 73 / 56 us vs 90 / 68 (ping-pong); K = 768 (since the LDS-DMA is issued as inline asm and
    // the GELU epilogues were slimmed): QKV 65 vs 67, dual-output FFN 95 vs 131, GELU-bwd 92 vs 127, N = 768: 24 vs 27; train step
    // 16.4 vs 16.9 ms.
    // (a persistent 128 x 256 variant that runs the GELU epilogue of tile i under the main loop of tile i + 1 was built and measured: the
    // slices cost their full time there too -- the two waves of a SIMD share its VALU issue and matrix pipe -- and the half-height tile's
    // main loop is 40 % slower: tools/ubench/gemm_hp_experiment.hip, profiles/r03_gemm_epilogue_overlap.md)
    if ((a_in.M % 256) == 0 && ((a_in.N % 256) == 0 || (a_in.N % 192) == 0) && a_in.K >= dp_min_k && !g_force_small_tile) {
        // small M (the per-GPU batches run_finetune.sh ships with: 4 x 2048 or 8 x 512 tokens): the 256-row tiles no longer fill 256 CUs
        // (M = 8192, N = 768: 128 workgroups) and the 128 x 128 kernel's four times as many workgroups, two per CU, win although it is
        // ~1.4 x slower per flop.  Rounds of workgroups x relative time per round; measured with every GEMM forced small: bert-base at
        // 8 sequences per step 1114 -> 1210 seq/s, longformer-base 4 x 2048 323 -> 330; at M = 16384 the deep-pipeline kernel wins everywhere.
        const int t_dp = (a_in.M / 256) * ((a_in.N % 256) == 0 ? a_in.N / 256 : a_in.N / 192);
        const int t_sm = (a_in.M / BM) * (a_in.N / BN);
        // Round 6: the 128 x 128 kernel wins clearly only as the RING form (about one tile per CU: M <= 6826 at N = 768 -- 31 vs 46 us at M = 4096,
        // K = 3072) or on short K; with two tiles on some CUs and a long K the 256 x 192 deep-pipeline tile (its residual / derivative operands now
        // prefetched under the last K tile) is ahead again: M = 10240, K = 3072 / 2304: 52.3 / 43.4 -> 47.3 / 39.1 us, M = 8192: 46.2 / 39.0 -> 45.0 / 37.7;
        // bert-base 16 x 512 +2.6 %, longformer 4 x 2048 +1.5 % per step (tools/dbg/ab_smallm_r06.sh, profiles/r06_smallm_dispatch.md)
        const bool ring = t_sm * 4 <= amdseg_num_cus() * 5;      // (a few CUs with two tiles still favour it: M = 6144, 288 tiles: 7.38 vs 7.46 ms per step)
        const float c_dp = (float)((t_dp + 255) / 256), c_sm = ((ring || a_in.K < 1536) ? 0.715f : 1.05f) * (float)((t_sm + 511) / 512);
        if (!(small_ok && c_sm < c_dp))
            return amdseg_launch_nt_dp<EPIX, OutT>(a_in, s);
        return launch_nt_small<EPIX, OutT>(a_in, s);
    }
    // the ping-pong kernel counts its in-flight stores (two outputs for BIAS_GELU): the single-output form runs on the 128x128 kernel
    const bool single_gelu = EPI == EPI_BIAS_GELU && !a_in.C2;
    if (single_gelu && !small_ok) return AMDSEG_ERR_SHAPE;
    const bool pp_ok = (a_in.M % PP_BM) == 0 && (a_in.N % PP_BN) == 0 && !single_gelu;
    constexpr int pp_min_k = 1536;
    if (pp_ok && !(g_force_small_tile && small_ok) && (a_in.K >= pp_min_k || !small_ok || g_force_small_tile < 0)) {
        static bool attr_set = false;          // > 64 KiB of dynamic LDS is opted into once per kernel instantiation
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_pp_kernel<EPIX, OutT>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        GemmNTArgs a = a_in;
        a.tiles_m = a.M / PP_BM; a.tiles_n = a.N / PP_BN;
        const int T = a.tiles_m * a.tiles_n;
        hipLaunchKernelGGL((gemm_nt_pp_kernel<EPIX, OutT>), dim3(T < 256 ? ((T + 7) / 8) * 8 : 256), dim3(512), PP_LDS, s, a);
        return amdseg_launch_status();
    }
    return launch_nt_small<EPIX, OutT>(a_in, s);
}

int amdseg_gemm_nt_impl(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                        int epi, const float* bias, const void* R, int ldr, void* C2, int ldc2, int out_fp32,
                        hipStream_t stream, const int* zkend, const int* zguard, int zL) {
    if (!A || !B || (!C && (epi & 0xff) != 7)) return AMDSEG_ERR_ARG;      // (BIAS_GELU_SPLIT: C == NULL = no pre-activation output)
    const int act = (epi >> 8) & 1;                         // AMDSEG_EPI_ACT_TANH: gelu_new instead of the erf GELU
    const int keepd = (epi >> 9) & 1;                       // AMDSEG_EPI_KEEP_DERIV: C2 / R is gelu'(pre-activation), not the pre-activation
    const int d8 = (epi >> 10) & 1;                         // AMDSEG_EPI_DERIV_U8: ... as one byte per element
    epi &= 0xff;
    if (keepd && ((act && !d8) || (epi != EPI_BIAS_GELU && epi != EPI_GELU_BWD))) return AMDSEG_ERR_ARG;   // gelu_new: the one-byte form only
    if (d8 && !keepd) return AMDSEG_ERR_ARG;
    const bool big = (M % PP_BM) == 0 && (N % PP_BN) == 0, small = (M % BM) == 0 && (N % BN) == 0;
    if (M <= 0 || N <= 0 || K <= 0 || !(big || small) || (K % BK)) return AMDSEG_ERR_SHAPE;
    if ((lda % 8) || (ldb % 8) || (ldc % 8)) return AMDSEG_ERR_SHAPE;
    GemmNTArgs a;
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.bias = bias; a.R = (const bf16_t*)R; a.C2 = (bf16_t*)C2;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.ldc2 = ldc2; a.M = M; a.N = N; a.K = K;
    a.tiles_m = M / BM; a.tiles_n = N / BN; a.dup_off = 0;
    const bool zok = zkend && zguard && zL > 0 && (zL % PP_BM) == 0 && (M % zL) == 0;      // a 256-row tile lies inside one sequence
    a.zkend = zok ? zkend : nullptr; a.zguard = zok ? zguard : nullptr; a.zL = zok ? zL : 0;
    switch (epi) {
        case EPI_NONE: return out_fp32 ? launch_nt<EPI_NONE, float>(a, stream) : launch_nt<EPI_NONE, bf16_t>(a, stream);
        case EPI_BIAS:
            if (!bias) return AMDSEG_ERR_ARG;
            return out_fp32 ? launch_nt<EPI_BIAS, float>(a, stream) : launch_nt<EPI_BIAS, bf16_t>(a, stream);
        case EPI_BIAS_GELU:
            if (!bias || out_fp32 || (C2 && (ldc2 % 8))) return AMDSEG_ERR_ARG;       // C2 == NULL: gelu output only (inference)
            if (keepd) {
                if ((M % 256) || ((N % 256) && (N % 192)) || K < 128 || (d8 && (N % 256))) return AMDSEG_ERR_SHAPE;
                if (d8) return act ? amdseg_launch_nt_dp<EPI_BIAS_GELU_DG8_TANH, bf16_t>(a, stream) : amdseg_launch_nt_dp<EPI_BIAS_GELU_DG8, bf16_t>(a, stream);
                return amdseg_launch_nt_dp<EPI_BIAS_GELU_DG, bf16_t>(a, stream);
            }
            return act ? launch_nt<EPI_BIAS_GELU_TANH, bf16_t>(a, stream) : launch_nt<EPI_BIAS_GELU, bf16_t>(a, stream);
        case EPI_ADD_RES:
            if (!R || (ldr % 8)) return AMDSEG_ERR_ARG;
            return out_fp32 ? launch_nt<EPI_ADD_RES, float>(a, stream) : launch_nt<EPI_ADD_RES, bf16_t>(a, stream);
        case EPI_GELU_BWD:
            if (!R || out_fp32 || (ldr % 8)) return AMDSEG_ERR_ARG;
            if (keepd) {
                if ((M % 256) || ((N % 256) && (N % 192)) || K < 128 || (d8 && (N % 256))) return AMDSEG_ERR_SHAPE;
                if (d8) return amdseg_launch_nt_dp<EPI_MUL_RES8, bf16_t>(a, stream);
                return amdseg_launch_nt_dp<EPI_MUL_RES, bf16_t>(a, stream);
            }
            return act ? launch_nt<EPI_GELU_BWD_TANH, bf16_t>(a, stream) : launch_nt<EPI_GELU_BWD, bf16_t>(a, stream);
        case 5:                                             // AMDSEG_EPI_BIAS_SPLIT: the 256 x 256 deep-pipeline kernel only
            if (!C2 || out_fp32 || (ldc2 % 8)) return AMDSEG_ERR_ARG;                 // bias may be NULL (no bias added)
            if ((M % 256) || (N % 256) || K < 128) return AMDSEG_ERR_SHAPE;
            return amdseg_launch_nt_dp<EPI_BIAS_SPLIT, bf16_t>(a, stream);
        case 6:                                             // AMDSEG_EPI_GELU_BWD_SPLIT: C = image [M, 3N] = [hi | hi | lo] of (A B^T) * gelu_erf'(R), R fp32
            if (!R || out_fp32 || (ldr % 4) || ldc < 3 * N || act) return AMDSEG_ERR_ARG;
            if ((M % 256) || (N % 256) || K < 128) return AMDSEG_ERR_SHAPE;
            a.C2 = reinterpret_cast<bf16_t*>(C) + 2 * (size_t)N; a.ldc2 = ldc; a.dup_off = N;
            return amdseg_launch_nt_dp<EPI_GELU_BWD_SPLIT, bf16_t>(a, stream);
        case 7:                                             // AMDSEG_EPI_BIAS_GELU_SPLIT: C fp32 pre-activation, C2 = image [M, 3N] of gelu_erf(C)
            if (!bias || !C2 || !out_fp32 || (ldc2 % 8) || ldc2 < 3 * N || act) return AMDSEG_ERR_ARG;
            if ((M % 256) || (N % 256) || K < 128) return AMDSEG_ERR_SHAPE;
            a.dup_off = N;
            return amdseg_launch_nt_dp<EPI_BIAS_GELU_SPLIT, float>(a, stream);
    }
    return AMDSEG_ERR_ARG;
}

// ------------------------------------------------------------------------------------------------ gemm_tn
// Operand tiles are [64 m][128 cols] bf16 (256 B per row, 16 chunks of 16 B); chunk c of row m is stored at slot
// c ^ ((m & 3) << 2) so the four rows gathered by one ds_read_b64_tr_b16 half-wave fall on disjoint banks.

__device__ __forceinline__ void tn_stage(const bf16_t* __restrict__ G, int ld, int m0, int col0, char* lds_tile, int w, int l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int R0 = w * 16 + q * 4;
        int r = R0 + (l >> 4), s = l & 15;
        int c = s ^ ((r & 3) << 2);
        glds16(G + (size_t)(m0 + r) * ld + col0 + c * 8, lds_tile + R0 * 256);
    }
}
// fragment for a 32-wide block of columns starting at col (multiple of 32), k-step kk (16 rows of m)
__device__ __forceinline__ bf16x8 tn_frag(const char* lds_tile, int col, int kk, int l) {
    const int q = l >> 4, i16 = l & 15, nblk = q & 1, g = q >> 1;
    const int c = ((col + nblk * 16) >> 3) + ((i16 & 3) >> 1);
    bf16x8 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int mr = kk * 16 + g * 8 + h * 4 + (i16 >> 2);
        const int off = mr * 256 + ((c ^ ((mr & 3) << 2)) << 4) + (i16 & 1) * 8;
        bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4, lds_tile + off));
        f[h * 4 + 0] = v[0]; f[h * 4 + 1] = v[1]; f[h * 4 + 2] = v[2]; f[h * 4 + 3] = v[3];
    }
    return f;
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTNArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1;
    const int t = xcd_remap(blockIdx.x, a.total_tiles);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < AMDSEG_MAX_GROUP; ++i)
        if (i < a.nprob && t >= a.p[i].tile_begin) pi = i;
    const TNProblem& P = a.p[pi];
    const int lt = t - P.tile_begin;
    const int tn_ = lt / P.tiles_k, tk = lt - tn_ * P.tiles_k;
    const int n0 = tn_ * 128, k0 = tk * 128;
#define bufA(i) (smem + (i) * 32768)
#define bufB(i) (smem + 16384 + (i) * 32768)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nm = a.M / 64;
    tn_stage(P.A, P.lda, 0, n0, bufA(0), w, l);
    tn_stage(P.B, P.ldb, 0, k0, bufB(0), w, l);
    for (int mt = 0; mt < nm; ++mt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = mt & 1;
        if (mt + 1 < nm) {
            tn_stage(P.A, P.lda, (mt + 1) * 64, n0, bufA(cur ^ 1), w, l);
            tn_stage(P.B, P.ldb, (mt + 1) * 64, k0, bufB(cur ^ 1), w, l);
        }
        const char* tA = bufA(cur);
        const char* tB = bufB(cur);
        bf16x8 fa[4][2], fb[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[kk][i] = tn_frag(tA, wr * 64 + i * 32, kk, l);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[kk][j] = tn_frag(tB, wc * 64 + j * 32, kk, l);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
        // 16 transpose reads up front, then two more behind each of the first 8 MFMAs, then the last 8 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    }
    __syncthreads();
    float* sm = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = wc * 64 + j * 32 + (l & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                sm[m * 128 + n] = acc[i][j][r];
            }
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int chunk = it * 256 + tid;
        const int r = chunk >> 5, cc = (chunk & 31) * 4;
        float4 x = *reinterpret_cast<const float4*>(sm + r * 128 + cc);
        float* dst = P.C + (size_t)(n0 + r) * P.ldc + k0 + cc;
        if (a.accumulate) {
            float4 o = *reinterpret_cast<const float4*>(dst);
            x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
        }
        *reinterpret_cast<float4*>(dst) = x;
    }
}

int amdseg_gemm_tn_grouped_impl(int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                float* const* C, const int* ldc, const int* N, const int* Kp, int M, int accumulate,
                                hipStream_t stream) {
    return amdseg_gemm_tn_grouped_bias_impl(nprob, A, lda, B, ldb, C, ldc, N, Kp, M, accumulate, nullptr, nullptr, stream);
}

// the same, plus optional bias gradients: colsum_out[i] (+)= column sums of A[i] over the M rows (colsum_out[i] may be NULL);
// colsum_scratch[i]: >= max(ceil(M/128), Kp[i]/128) * N[i] floats, untouched until the queued reductions have run.  With the deep-pipeline
// kernel the sums come out of the GEMM's own A fragments (no extra pass over dY); otherwise amdseg_colsum runs.
int amdseg_gemm_tn_grouped_bias_impl(int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                     float* const* C, const int* ldc, const int* N, const int* Kp, int M, int accumulate,
                                     float* const* colsum_out, float* const* colsum_scratch, hipStream_t stream,
                                     const int* runs, const int* counts, const int* zguard) {
    if (nprob <= 0 || nprob > AMDSEG_MAX_GROUP || !A || !B || !C) return AMDSEG_ERR_ARG;
    if (M <= 0 || (M % 64)) return AMDSEG_ERR_SHAPE;
    GemmTNArgs a;
    int tiles = 0;
    for (int i = 0; i < nprob; ++i) {
        if (!A[i] || !B[i] || !C[i]) return AMDSEG_ERR_ARG;
        if ((N[i] % 128) || (Kp[i] % 128) || (lda[i] % 8) || (ldb[i] % 8) || (ldc[i] % 4)) return AMDSEG_ERR_SHAPE;
        TNProblem& P = a.p[i];
        P.A = (const bf16_t*)A[i]; P.B = (const bf16_t*)B[i]; P.C = C[i];
        P.N = N[i]; P.Kp = Kp[i]; P.lda = lda[i]; P.ldb = ldb[i]; P.ldc = ldc[i];
        P.tile_begin = tiles; P.tiles_k = Kp[i] / 128;
        P.colsum_part = nullptr;
        if (colsum_out && colsum_out[i] && !(colsum_scratch && colsum_scratch[i])) return AMDSEG_ERR_ARG;
        tiles += (N[i] / 128) * (Kp[i] / 128);
    }
    for (int i = nprob; i < AMDSEG_MAX_GROUP; ++i) a.p[i] = a.p[0];
    a.nprob = nprob; a.M = M; a.accumulate = accumulate; a.total_tiles = tiles;
    const bool zok = runs && counts && zguard;
    a.runs = zok ? runs : nullptr; a.counts = zok ? counts : nullptr; a.zguard = zok ? zguard : nullptr;
    // 256 x 128 deep-pipeline kernel (gemm_dp.hip) when every problem tiles by it: ~0.7x the time of the kernel below
    bool dp = M >= 128 && !g_force_small_tile;
    for (int i = 0; i < nprob; ++i) dp = dp && (N[i] % 256) == 0 && (size_t)M * (size_t)(lda[i] > ldb[i] ? lda[i] : ldb[i]) < ((size_t)1 << 31);
    if (dp) {
        for (int i = 0; i < nprob; ++i)
            if (colsum_out && colsum_out[i]) a.p[i].colsum_part = colsum_scratch[i];
        int rc = amdseg_launch_tn_dp(a, stream);
        if (rc) return rc;
        for (int i = 0; i < nprob; ++i)                     // out[n] (+)= sum over the K' tiles of the per-tile partials
            if (colsum_out && colsum_out[i]) amdseg_reduce_rows(colsum_scratch[i], Kp[i] / 128, N[i], N[i], colsum_out[i], accumulate, stream);
        return amdseg_launch_status();
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles), dim3(256), 0, stream, a);
    for (int i = 0; i < nprob; ++i)
        if (colsum_out && colsum_out[i]) {
            int rc = amdseg_colsum_impl(A[i], lda[i], colsum_scratch[i], colsum_out[i], M, N[i], accumulate, AMDSEG_BF16, stream);
            if (rc) return rc;
        }
    return amdseg_launch_status();
}
