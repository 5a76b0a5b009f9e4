// gemm_nt, deep-pipeline kernel for long-K shapes on gfx950:  C[M,N] = A[M,K] . B[N,K]^T (+ epilogue), M % 256 == 0, N % 256 == 0.
// (same reference lines as gemm.hip: the torch.nn.Linear calls of [hf] models/bert/modeling_bert.py:340-351 (FFN output dense,
// K = intermediate size) and the autograd dgrads dx = dy . W of the FFN / QKV projections.)
//
// One 512-thread workgroup per CU computes a 256 x 256 tile with v_mfma_f32_16x16x32_bf16:
//   * 8 waves = 2 row groups x 4 column waves, wave tile 128 x 64 (8 x 4 fragments, fp32 accumulators in 128 VGPRs);
//   * two 64-KiB LDS stages of [64 rows][64 k] bf16 tile images (4 for A, 4 for B), filled by global_load_lds DMA
//     (16 B per lane) about one K tile ahead, retired with COUNTED s_waitcnt vmcnt so the next refill stays in flight;
//   * two phases per K tile (rows 0-63 | 64-127 of the wave tile x all 64 columns = 32 MFMAs each).  A phase is
//     [fragment reads + DMA issue | s_barrier | MFMAs | s_barrier]; group 1 runs one barrier behind group 0, so one group's
//     LDS reads and DMA issue always run under the other group's MFMAs;
//   * the 16-B chunk c of tile row r is stored at chunk c ^ f((r >> 1) & 7), f(p) = p ^ (p in {2,3,4,5}): ds_read_b128 serves a
//     wave in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (8 rows at chunk c + the other 8 rows at chunk c ^ 1) and
//     this makes every group hit 16 distinct 16-B slots (SQ_LDS_BANK_CONFLICT = 0; the attention-style swizzle had 2-way
//     conflicts on every fragment read = half the LDS rate, 1099 -> 1277 TFLOP/s on 8192 x 7680 x 8192).
// Measured (plain, random bf16 data, tools/ubench/gemm8p.cpp): 16384 x 768 x 3072: 70.6 us = 1096 TFLOP/s (ping-pong 256x192
// kernel: 90 us, hipBLASLt: 61 us); 8192 x 7680 x 8192: 1277 TFLOP/s; 4096 x 3840 x 4096: 1312 TFLOP/s.
// Hazards (MI355X_MICROARCH.md, LDS-DMA ordering): a stage tile is refilled only after every reader retired its fragment
// reads with lgkmcnt(0) BEFORE the barrier that ends its load half; DMA data is read only after the issuing wave's counted
// vmcnt and a later barrier.
#include "common.h"
#include "amdseg_internal.h"
#include "gemm_epi.h"

#define DP_BM 256
#define DP_BN 256
#define DP_STAGE (8 * 8192)
#define DP_LDS (2 * DP_STAGE)
#define DP_ST16(ptr, val) *reinterpret_cast<uint4*>(ptr) = (val)

__device__ __forceinline__ int dp_swz(int r) { const int p = (r >> 1) & 7; return p ^ (((p + 2) >> 2) & 1); }
__device__ __forceinline__ bf16x8 dp_frag(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ dp_swz(r)) << 4));
}
// B images of the 256-wide tile: fragment e of a column wave reads the PERMUTED rows n = (e >> 1)*32 + (i16 >> 2)*8 + (e & 1)*4 + (i16 & 3),
// so that a lane's accumulators of the fragment pair (2j, 2j+1) are 8 CONSECUTIVE output columns (g*8 .. +8 of the pair's 32): the
// epilogue then stores 16 bytes per lane straight from registers (no LDS staging pass).  The row class that the swizzle keys on is
// u(r) = ((r >> 3) & 3)*2 + ((r >> 1) & 1) = i16 >> 1 of the reading lane, i.e. the same lane-group structure dp_swz was derived for.
__device__ __forceinline__ int dp_swzB(int r) { const int p = ((r >> 3) & 3) * 2 + ((r >> 1) & 1); return p ^ (((p + 2) >> 2) & 1); }
__device__ __forceinline__ bf16x8 dp_fragB(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ dp_swzB(r)) << 4));
}
__device__ __forceinline__ void dp_glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(void, lds_wave_base), 16, 0, 0);
}

// NF = 16-column fragments per column wave: 4 -> 256 x 256 tile, 3 -> 256 x 192 tile (wave tile 128 x 48, three B images per stage)
// for widths that are multiples of 192 but not of 256.
// (Rounds 3-5 also carried a persistent form -- one workgroup per CU walking its tiles, the next tile's two stages requested behind the K loop -- the
// early-start prologue, the relaxed lgkmcnt waits, a fused bias + dropout + residual epilogue and a set of wrong-result timing probes; all measured
// neutral or slower (profiles/r03_gemm_epilogue_overlap.md, r04_gemm_epilogue_split.md, r04_gemm_prologue_ablation.md, r04_fused_drop_res.md) and
// were removed from the product in round 6: `git show 17e81c4:spokennlp_amd/csrc/gemm_dp.hip`.)
template <int EPIX, typename OutT, int NF>
__global__ __launch_bounds__(512, 1) void gemm_nt_dp_kernel(GemmNTArgs a) {
    constexpr int EPI = EPI_BASE(EPIX); constexpr int ACT = EPI_ACT(EPIX); (void)ACT;
    constexpr int BN = NF * 64, WN = NF * 16, STAGE = (4 + NF) * 8192;     // tile width, wave-tile width, bytes per LDS stage
    constexpr bool E_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_SPLIT || EPI == EPI_BIAS_GELU_SPLIT ||
                            EPI == EPI_BIAS_GELU_DG || EPI == EPI_BIAS_GELU_DG8;           // a bias vector goes into the accumulators
    constexpr bool E_RES = EPI == EPI_ADD_RES || EPI == EPI_GELU_BWD || EPI == EPI_MUL_RES;    // a bf16 operand R is read
    constexpr bool E_GELU2 = EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_DG;               // two outputs: C2 (pre-activation / derivative) and C
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3, wq = w & 3;
    const int g = l >> 4, i16 = l & 15;
    const int nwg = a.tiles_m * a.tiles_n;
    const int gsz_full = GROUP_M * a.tiles_n;
    // tile of launch index tt (each XCD walks a contiguous range, GROUP_M row tiles share their B panel): (m0, n0)
#define DP_TILE_OF(tt, M0, N0) { const int t_ = xcd_remap(tt, nwg); const int grp_ = t_ / gsz_full, first_m_ = grp_ * GROUP_M; \
        const int gmn_ = min(a.tiles_m - first_m_, GROUP_M); const int rem_ = t_ - grp_ * gsz_full; \
        M0 = (first_m_ + rem_ % gmn_) * DP_BM; N0 = (rem_ / gmn_) * BN; }
#define DP_ZTILE(M0) (a.zkend != nullptr && ((M0) - ((M0) / a.zL) * a.zL) >= a.zkend[(M0) / a.zL] && *a.zguard == 0)
    int m0, n0;
    DP_TILE_OF(blockIdx.x, m0, n0)
#define DP_TILE_A(s, i) (smem + (s) * STAGE + (i) * 8192)
#define DP_TILE_B(s, i) (smem + (s) * STAGE + (4 + (i)) * 8192)
    const bf16_t* pA = a.A + (size_t)(m0 + wr * 128) * a.lda;                 // this group's A rows
    const bf16_t* pB = a.B + (size_t)(n0 + wr * 128) * a.ldb;                 // this group's DMA duty on B: tile images 2wr, 2wr+1 (NF = 3: group 1 has image 2 only)
    const bool b2 = NF == 4 || wr == 0;                                       // two B images for this group
    const bool vm8 = b2;                                                      // pieces per wave and K tile: 4 (A) + 4 or 2 (B)
    uint32_t offA[4], offB[4];                                                // lane BYTE offsets, constant over K (saddr + voffset loads)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wq * 2 + q) * 8 + (l >> 3), c = (l & 7) ^ dp_swz(r), cb = NF == 4 ? ((l & 7) ^ dp_swzB(r)) : c;
            offA[i * 2 + q] = ((i * 64 + r) * a.lda + c * 8) * 2;
            offB[i * 2 + q] = ((i * 64 + r) * a.ldb + cb * 8) * 2;
        }
    // SGPR base + 32-bit lane byte offset: no VALU address arithmetic where the pieces are issued
    // LDS addresses of this wave's DMA pieces as 32-bit scalars (base of the wave's duty + compile-time offsets): m0 is written by one SALU add
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
    const uint32_t ldsAw = __builtin_amdgcn_readfirstlane(lds0 + (wr * 2) * 8192 + (wq * 2) * 1024);
    const uint32_t ldsBw = __builtin_amdgcn_readfirstlane(lds0 + (4 + wr * 2) * 8192 + (wq * 2) * 1024);
#define DP_DMA_A_P(PA, s, i, kt) _Pragma("unroll") for (int q = 0; q < 2; ++q) \
        amdseg_glds16_saddr_lds((PA) + (kt) * 64, offA[(i) * 2 + q], ldsAw + (s) * STAGE + (i) * 8192 + q * 1024);
#define DP_DMA_A(s, i, kt) DP_DMA_A_P(pA, s, i, kt)
#define DP_DMA_B_P(PB, s, kt) _Pragma("unroll") for (int i = 0; i < 2; ++i) if (i == 0 || b2) _Pragma("unroll") for (int q = 0; q < 2; ++q) \
        amdseg_glds16_saddr_lds((PB) + (kt) * 64, offB[i * 2 + q], ldsBw + (s) * STAGE + i * 8192 + q * 1024);
#define DP_DMA_B(s, kt) DP_DMA_B_P(pB, s, kt)
#define DP_WAIT_TILE() do { if (vm8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } while (0)
    // the bias vector is requested HERE, in front of the prologue's DMA, and added at the end (round 6): the epilogue used to open with these four
    // global loads per lane, whose latency every tile of every biased GEMM paid with the matrix pipe idle.  (It is still ADDED last: starting the
    // accumulators from it was measured too -- another 0.5-1 us per launch -- but rounds differently from the 128 x 128 and ping-pong kernels of
    // gemm.hip, and the suite holds this library to bit-identical results across batch compositions, i.e. across kernels.)
    // (256 x 192 tile only: the 256-wide instantiations have no 16 registers to spare across the K loop -- 6 spilled VGPRs in the GELU epilogues -- and
    //  request the vector at the top of the epilogue as before)
    constexpr bool BIAS_EARLY = NF == 3;
#define DP_LOAD_BIAS() _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) \
            bv[nf] = (EPI == EPI_BIAS_SPLIT && !a.bias) ? make_float4(0.f, 0.f, 0.f, 0.f)      /* BIAS_SPLIT without a bias: the plain product as an image */ \
                   : *reinterpret_cast<const float4*>(a.bias + n0 + wc * WN + (NF == 4 ? (nf >> 1) * 32 + g * 8 + (nf & 1) * 4 : nf * 16 + g * 4));
    float4 bv[NF];
    if (E_BIAS && BIAS_EARLY) { DP_LOAD_BIAS() }
    f32x4 acc[8][NF];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = a.K / 64;
    const bool ztile = DP_ZTILE(m0);                          // workgroup-uniform: every row of this tile's A is an exact zero
    // EPI_MUL_RES8 (round 6): the tile's 256 x 256 one-byte derivatives (64 KiB = one LDS stage) are fetched by LDS-DMA UNDER the last K tile into
    // the stage that tile does not use -- every read of it was retired two barriers earlier (both groups: see the hazard notes at the top) --
    // instead of by 16 global loads per lane at the start of the epilogue, where all CUs of a round waited for them with the matrix pipes idle
    // (rocprofv3, round 6: 91 us per launch against 65-70 for the K loops alone; the derivative reads were the exposed part).
    // Piece j of wave w = tile rows w*32 + j*4 .. +4, all 256 columns; the 16-B unit (row r, columns c16*16 .. +16) lands at LDS unit (c16 + r) & 15 of
    // its row, so the epilogue's ds_read_b64 (16 rows x two neighbouring units per wave instruction) touches every 16-B bank group exactly twice.
    constexpr bool R8PF = EPI == EPI_MUL_RES8 && NF == 4;
    // EPI_ADD_RES on the 256 x 192 tile (the residual-adding input-gradient GEMMs dx1 = du W1 + dz2, dx_in = dqkv Wqkv + dz1): the bf16 residual tile is
    // 96 KiB, the free stage 56 KiB -- the FIRST HALF of each row group's rows (mf 0..3: tile rows 0-63 and 128-191, 128 rows x 384 B) comes through
    // LDS under the last K tile, the second half by global loads requested at the top of the epilogue, in front of the arithmetic of the first.
    // LDS rows are 416 B apart (26 units of 16 B, the last two padding): 16 consecutive rows then start in 8 distinct 32-B bank slots, two each, and
    // the epilogue's ds_read_b64 (16 rows x 32 B per wave instruction) is conflict-free.  56 pieces of 1 KiB = the stage exactly, 7 per wave.
    constexpr bool R16PF = EPI == EPI_ADD_RES && NF == 3;
    constexpr bool RPF = R8PF || R16PF;
#define DP_RPF_WAIT() do { if (R8PF) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); } while (0)
#define DP_R16_ISSUE(stage) do { if (R16PF) { \
        const bf16_t* r16b_ = a.R + (size_t)m0 * a.ldr + n0; \
        _Pragma("unroll") for (int j_ = 0; j_ < 7; ++j_) { \
            const int u_ = (w * 7 + j_) * 64 + l, rho_ = min(u_ / 26, 127), un_ = min(u_ - rho_ * 26, 23); \
            const int tr_ = (rho_ & 63) + ((rho_ >> 6) << 7); \
            amdseg_glds16_saddr_lds(r16b_, (uint32_t)((tr_ * a.ldr + un_ * 8) * 2), lds0 + (stage) * STAGE + (w * 7 + j_) * 1024); } } } while (0)
#define DP_RPF_ISSUE(stage) do { DP_R8_ISSUE(stage); DP_R16_ISSUE(stage); } while (0)
#define DP_R8_ISSUE(stage) do { if (R8PF) { \
        const unsigned char* r8b_ = reinterpret_cast<const unsigned char*>(a.R) + (size_t)(m0 + w * 32) * a.ldr + n0; \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) { \
            const int r_ = j_ * 4 + (l >> 4); \
            amdseg_glds16_saddr_lds(r8b_, (uint32_t)(r_ * a.ldr + ((((l & 15) - r_) & 15) << 4)), lds0 + (stage) * STAGE + (w * 32 + j_ * 4) * 256); } } } while (0)
    if (!ztile) {
    // prologue: both stages; the first MFMA phase waits for K tile 1 as well (vmcnt(0) at kt = 0).  (Round 4 tried an early start -- K tile 1 issued in
    // the loop's steady-state order so that the first MFMAs wait for K tile 0 only -- and measured it slower, profiles/r04_gemm_prologue_ablation.md:
    // both stages of a fresh workgroup land together, the fill is one latency, not two transfers.)
    DP_DMA_A(0, 0, 0) DP_DMA_A(0, 1, 0) DP_DMA_B(0, 0)
#define DP_KT0 1
    if (nk > 1) { DP_DMA_A(1, 0, 1) DP_DMA_A(1, 1, 1) DP_DMA_B(1, 1) DP_WAIT_TILE(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();             // stagger: group 1 runs one barrier behind group 0
    bf16x8 fa[4][2], fb[NF][2];
    // column e*16 + i16 of the wave's WN columns -> (B image, row): wave-uniform per fragment (WN = 48 straddles images)
    int fbi[NF], fbr[NF];
#pragma unroll
    for (int e = 0; e < NF; ++e) { const int c = wc * WN + e * 16; fbi[e] = c >> 6; fbr[e] = c & 63; }
#define DP_LOAD_A(s, h) _Pragma("unroll") for (int f = 0; f < 4; ++f) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        fa[f][kk] = dp_frag(DP_TILE_A(s, wr * 2 + (h)), f * 16 + i16, kk * 4 + g);
#define DP_LOAD_B(s) _Pragma("unroll") for (int e = 0; e < NF; ++e) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        fb[e][kk] = NF == 4 ? dp_fragB(DP_TILE_B(s, wc), (e >> 1) * 32 + (i16 >> 2) * 8 + (e & 1) * 4 + (i16 & 3), kk * 4 + g) \
                            : dp_frag(DP_TILE_B(s, fbi[e]), fbr[e] + i16, kk * 4 + g);
    // NF = 4, unrolled K loop: the fragment addresses written out as (stage, k-half)-specific lane bases + compile-time immediates that fit the
    // 16-bit offset field of ds_read_b128 (the stage-1 images start at 64 KiB, so each stage has its own bases): 8 VGPRs, no address
    // arithmetic in the loop.  (c ^ swz) << 4 with c = kk*4 + g splits into (kk << 2) ^ (g ^ swz); both swizzles depend on i16 only.
    const char* la_[2][2];
    const char* lb_[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            la_[st][kk] = smem + st * STAGE + wr * 2 * 8192 + i16 * 128 + ((((g ^ dp_swz(i16)) ^ (kk << 2))) << 4);
            const int rb = (i16 >> 2) * 8 + (i16 & 3);
            lb_[st][kk] = smem + st * STAGE + (4 + wc) * 8192 + rb * 128 + ((((g ^ dp_swzB(rb)) ^ (kk << 2))) << 4);
        }
#define DP_LOAD_A_U(S, h) _Pragma("unroll") for (int f = 0; f < 4; ++f) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        fa[f][kk] = *reinterpret_cast<const bf16x8*>(la_[S][kk] + (h) * 8192 + f * 2048);
#define DP_LOAD_B_U(S) _Pragma("unroll") for (int e = 0; e < NF; ++e) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        fb[e][kk] = *reinterpret_cast<const bf16x8*>(lb_[S][kk] + (e >> 1) * 4096 + (e & 1) * 512);
#define DP_MFMA(ah) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int f = 0; f < 4; ++f) \
        _Pragma("unroll") for (int e = 0; e < NF; ++e) \
        acc[(ah) * 4 + f][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[e][kk], fa[f][kk], acc[(ah) * 4 + f][e], 0, 0, 0);
#define DP_MID() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); } while (0)
#define DP_END() do { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    // one K tile on stage S.  phase 1: rows 0-63 of the wave tile; DMA: the rows-64..127 image of K tile kt+1 (other stage; last read in phase 2
    // of K tile kt-1, retired before that phase's barrier).  phase 2: rows 64-127; DMA: rows-0..63 image + this group's two B images of K tile
    // kt+2 into THIS stage (their last readers -- this group's phase 1 and the other group's phase 1, one slot later -- have retired their reads).
    // (All 6 pieces of phase 2 stay in its load half: issuing the B pieces among the MFMAs measured slower, profiles/r04_gemm_dma_placement.md.)
#define DP_LOOP_DMA(x) x
    // every fragment read is retired (lgkmcnt(0)) before the barrier that ends its load half: the refill order would allow lgkmcnt(8) in front of
    // the first barrier and no wait in front of the second (only group 1's B reads need it), measured neutral on all eight shapes (round 4)
#define DP_LGKM_P1() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define DP_LGKM_P2() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define DP_KTILE(kt, S, LDA, LDB) { \
        LDB(S) __builtin_amdgcn_sched_barrier(0); LDA(S, 0) __builtin_amdgcn_sched_barrier(0); \
        if ((kt) >= DP_KT0 && (kt) + 1 < nk) { DP_LOOP_DMA(DP_DMA_A((S) ^ 1, 1, (kt) + 1)) } \
        else if (RPF && (kt) + 1 == nk) DP_RPF_ISSUE((S) ^ 1); \
        DP_LGKM_P1(); \
        if ((kt) >= DP_KT0 && (kt) + 1 < nk) DP_WAIT_TILE(); \
        else if (RPF && (kt) + 1 == nk) DP_RPF_WAIT();           /* everything but the epilogue operand's pieces */ \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        DP_MID(); \
        DP_MFMA(0) \
        DP_END(); \
        LDA(S, 1) \
        if ((kt) + 2 < nk) { DP_LOOP_DMA(DP_DMA_A(S, 0, (kt) + 2) DP_DMA_B(S, (kt) + 2)) } \
        DP_LGKM_P2(); \
        if ((kt) + 2 < nk) DP_WAIT_TILE(); \
        else if (RPF && (kt) + 1 == nk) DP_RPF_WAIT(); \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        DP_MID(); \
        DP_MFMA(1) \
        DP_END(); }
    int kt = 0;
    // the K loop unrolled by two: the stage is a compile-time constant in each copy, so every fragment address is (a lane term that does
    // not depend on the stage) + an immediate -- round 4: the rolled loop re-derived them per K tile, ~40 vector instructions in the load
    // halves (SQ counters, profiles/r04_gemm_dma_placement.md)
    if (NF == 4) {
        for (; kt + 1 < nk; kt += 2) { DP_KTILE(kt, 0, DP_LOAD_A_U, DP_LOAD_B_U) DP_KTILE(kt + 1, 1, DP_LOAD_A_U, DP_LOAD_B_U) }
        if (kt < nk) DP_KTILE(kt, 0, DP_LOAD_A_U, DP_LOAD_B_U)
    } else {
        for (; kt + 1 < nk; kt += 2) { DP_KTILE(kt, 0, DP_LOAD_A, DP_LOAD_B) DP_KTILE(kt + 1, 1, DP_LOAD_A, DP_LOAD_B) }
        if (kt < nk) DP_KTILE(kt, 0, DP_LOAD_A, DP_LOAD_B)
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();             // group 0 pays back the stagger barrier: every LDS read is retired now
    } else if (RPF) DP_RPF_ISSUE(nk & 1);                  // a skipped tile multiplies zeros by the same derivatives (the sign of a zero is a bit too) / adds the same residual
    if (RPF) {                                             // the operand image has landed and every wave sees it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue.  Lane owns row m = mf*16 + i16 and columns nf*16 + g*4 .. +4 of the wave tile.  bf16 results go through a
    // wave-private 16-KiB LDS image (2 x [64 rows][128 B], swizzled) so every global store instruction writes 8 full 128-B lines.
    // the bias goes INTO the accumulators once: recomputing acc + bias in both passes of BIAS_GELU made the compiler keep all
    // 128 sums of pass 0 alive for pass 1 (common subexpression) next to the 128 accumulators -> 116 spilled VGPRs
    if (E_BIAS && !BIAS_EARLY) { DP_LOAD_BIAS() }
    if (E_BIAS) {
#pragma unroll
        for (int mf = 0; mf < 8; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) { acc[mf][nf][0] += bv[nf].x; acc[mf][nf][1] += bv[nf].y; acc[mf][nf][2] += bv[nf].z; acc[mf][nf][3] += bv[nf].w; }
    }
    if (NF == 4) {
        // ---- direct epilogue (256-wide tile): lane owns row m = mf*16 + i16 and the 8 consecutive columns ep*32 + g*8 .. +8 of its wave's 64.
        // BIAS_GELU writes the pre-activation and the activation of a chunk back to back (the stores of one overlap the GELU math of the next)
        // (round 4, second session: requesting R for the whole tile up front -- 16 loads of 16 B per lane in flight instead of two per row fragment --
        //  measured neutral, 91.9 / 96.8 vs 98.2 / 94.1 us for GELU' at N = 3072: the epilogue's R traffic is HBM time, not latency;
        //  profiles/r04_gemm_epilogue_split.md)
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) {
            const size_t gm = (size_t)(m0 + wr * 128 + mf * 16 + i16);
            uint4 rr[2];
            if (E_RES) {
#pragma unroll
                for (int ep = 0; ep < 2; ++ep) rr[ep] = *reinterpret_cast<const uint4*>(a.R + gm * a.ldr + n0 + wc * 64 + ep * 32 + g * 8);
            }
            uint2 r8[2];
            if (EPI == EPI_MUL_RES8) {                         // the derivative kept by the forward, one byte per element
#pragma unroll
                for (int ep = 0; ep < 2; ++ep) {
                    if (R8PF) {                                // ... from the LDS image fetched under the last K tile (stage nk & 1)
                        const int tr = wr * 128 + mf * 16 + i16, c16 = wc * 4 + ep * 2 + (g >> 1);
                        r8[ep] = *reinterpret_cast<const uint2*>(smem + (nk & 1) * STAGE + tr * 256 + (((c16 + tr) & 15) << 4) + (g & 1) * 8);
                    } else
                    r8[ep] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(a.R) + gm * a.ldr + n0 + wc * 64 + ep * 32 + g * 8);
                }
            }
            float4 ru[2][2];
            if (EPI == EPI_GELU_BWD_SPLIT) {                   // the fp32 pre-activation
#pragma unroll
                for (int ep = 0; ep < 2; ++ep) {
                    const float* up = reinterpret_cast<const float*>(a.R) + gm * a.ldr + n0 + wc * 64 + ep * 32 + g * 8;
                    ru[ep][0] = *reinterpret_cast<const float4*>(up); ru[ep][1] = *reinterpret_cast<const float4*>(up + 4);
                }
            }
#pragma unroll
            for (int ep = 0; ep < 2; ++ep) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = acc[mf][2 * ep][r]; v[4 + r] = acc[mf][2 * ep + 1][r]; }
                const size_t col = (size_t)(n0 + wc * 64 + ep * 32 + g * 8);
                if (EPI == EPI_BIAS_GELU_SPLIT) {              // the fp32 pre-activation, then the activation as a split image
                    if (a.C) {                                     // (inference passes no pre-activation buffer: only backward reads it)
                        float* dst = reinterpret_cast<float*>(a.C) + gm * a.ldc + col;
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = gelu_erf(v[q]);
                    uint4 hi; hi.x = pack2bf(v[0], v[1]); hi.y = pack2bf(v[2], v[3]); hi.z = pack2bf(v[4], v[5]); hi.w = pack2bf(v[6], v[7]);
                    const uint32_t hw[4] = {hi.x, hi.y, hi.z, hi.w};
                    uint4 lo;
                    uint32_t lw[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) lw[q] = pack2bf(v[2 * q] - __uint_as_float(hw[q] << 16), v[2 * q + 1] - __uint_as_float(hw[q] & 0xffff0000u));
                    lo.x = lw[0]; lo.y = lw[1]; lo.z = lw[2]; lo.w = lw[3];
                    bf16_t* img = a.C2 + gm * a.ldc2 + col;
                    *reinterpret_cast<uint4*>(img) = hi; *reinterpret_cast<uint4*>(img + a.dup_off) = hi; *reinterpret_cast<uint4*>(img + 2 * a.dup_off) = lo;
                    continue;
                }
                if (EPI == EPI_GELU_BWD_SPLIT) {
                    const float uu[8] = {ru[ep][0].x, ru[ep][0].y, ru[ep][0].z, ru[ep][0].w, ru[ep][1].x, ru[ep][1].y, ru[ep][1].z, ru[ep][1].w};
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= gelu_erf_grad(uu[q]);
                }
                if (EPI == EPI_BIAS_SPLIT || EPI == EPI_GELU_BWD_SPLIT) {   // x = hi + lo, both bf16: C <- hi, C2 <- lo (same row / column)
                    uint4 hi; hi.x = pack2bf(v[0], v[1]); hi.y = pack2bf(v[2], v[3]); hi.z = pack2bf(v[4], v[5]); hi.w = pack2bf(v[6], v[7]);
                    const uint32_t hw[4] = {hi.x, hi.y, hi.z, hi.w};
                    uint4 lo;
                    uint32_t lw[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) lw[q] = pack2bf(v[2 * q] - __uint_as_float(hw[q] << 16), v[2 * q + 1] - __uint_as_float(hw[q] & 0xffff0000u));
                    lo.x = lw[0]; lo.y = lw[1]; lo.z = lw[2]; lo.w = lw[3];
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + gm * a.ldc + col) = hi;
                    if (EPI == EPI_GELU_BWD_SPLIT) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + gm * a.ldc + col + a.dup_off) = hi;
                    *reinterpret_cast<uint4*>(a.C2 + gm * a.ldc2 + col) = lo;
                    continue;
                }
                if (EPI == EPI_BIAS_GELU_DG8) {                // ... the derivative as one byte per element
                    if (a.C2) {
                        float d[8];
                        uint2 q;
                        if (ACT) {                                 // gelu_new (BigBird): value and derivative from one tanh
#pragma unroll
                            for (int e = 0; e < 8; ++e) { float h_; gelu_tanh_both(v[e], h_, d[e]); v[e] = h_; }
                            q.x = gelu_dq_pack4(d); q.y = gelu_dq_pack4(d + 4);
                        } else {                                   // the derivative comes out in the units of its byte code
                            gelu_both4q(v, d, GELU_DQ_SCALE, GELU_DQ_OFF); gelu_both4q(v + 4, d + 4, GELU_DQ_SCALE, GELU_DQ_OFF);
                            q.x = gelu_dq_pack4_scaled(d); q.y = gelu_dq_pack4_scaled(d + 4);
                        }
                        // (plain stores: as NON-TEMPORAL 8-byte stores -- the tensor is read next by backward -- the launch average of the NT GEMMs
                        //  went 59.8 -> 63.5 us: partial lines that bypass L2's write combining)
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(a.C2) + gm * a.ldc2 + col) = q;
                    } else { gelu_act4(v, ACT); gelu_act4(v + 4, ACT); }
                } else if (EPI == EPI_MUL_RES8) {
                    gelu_dq_mul4(v, r8[ep].x); gelu_dq_mul4(v + 4, r8[ep].y);
                } else if (EPI == EPI_BIAS_GELU_DG) {          // gelu and, for backward, its derivative from one sigmoid (AMDSEG_EPI_KEEP_DERIV)
                    if (a.C2) {
                        float d[8];
                        gelu_both4(v, d); gelu_both4(v + 4, d + 4);
                        uint4 pk; pk.x = pack2bf(d[0], d[1]); pk.y = pack2bf(d[2], d[3]); pk.z = pack2bf(d[4], d[5]); pk.w = pack2bf(d[6], d[7]);
                        DP_ST16(a.C2 + gm * a.ldc2 + col, pk);
                    } else { gelu_act4(v, 0); gelu_act4(v + 4, 0); }
                } else if (EPI == EPI_BIAS_GELU) {
                    if (a.C2) {
                        uint4 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); pk.z = pack2bf(v[4], v[5]); pk.w = pack2bf(v[6], v[7]);
                        DP_ST16(a.C2 + gm * a.ldc2 + col, pk);
                    }
                    gelu_act4(v, ACT); gelu_act4(v + 4, ACT);
                } else if (E_RES) {
                    const uint32_t rw[4] = {rr[ep].x, rr[ep].y, rr[ep].z, rr[ep].w};
                    float rf[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { rf[2 * q] = __uint_as_float(rw[q] << 16); rf[2 * q + 1] = __uint_as_float(rw[q] & 0xffff0000u); }
                    if (EPI == EPI_ADD_RES) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += rf[q];
                    } else if (EPI == EPI_MUL_RES) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] *= rf[q];
                    } else { gelu_grad_mul4(v, rf[0], rf[1], rf[2], rf[3], ACT); gelu_grad_mul4(v + 4, rf[4], rf[5], rf[6], rf[7], ACT); }
                }
                if (sizeof(OutT) == 2) {
                    uint4 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); pk.z = pack2bf(v[4], v[5]); pk.w = pack2bf(v[6], v[7]);
                    DP_ST16(reinterpret_cast<bf16_t*>(a.C) + gm * a.ldc + col, pk);
                } else {
                    float* dst = reinterpret_cast<float*>(a.C) + gm * a.ldc + col;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    constexpr bool STAGED = sizeof(OutT) == 2;
    char* stg = smem + w * 16384;
    uint2 rpf[R16PF ? 8 : 1][NF];
    if (R16PF) {
        // every residual value of the wave tile in registers before any wave writes its staging image (the images overlap the residual image):
        // rows of mf 0..3 from LDS, rows of mf 4..7 from memory (in flight under the arithmetic below)
#pragma unroll
        for (int mf = 4; mf < 8; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                rpf[mf][nf] = *reinterpret_cast<const uint2*>(a.R + (size_t)(m0 + wr * 128 + mf * 16 + i16) * a.ldr + n0 + wc * WN + nf * 16 + g * 4);
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                rpf[mf][nf] = *reinterpret_cast<const uint2*>(smem + (nk & 1) * STAGE + (wr * 64 + mf * 16 + i16) * 416 + wc * 96 + nf * 32 + g * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#define DP_STG_OFF(r, c16) (((r) >> 6) * 8192 + ((r) & 63) * 128 + ((((c16) ^ (((r) & 63) ^ (((r) & 63) >> 3))) & 7) << 4))
    const int col0 = n0 + wc * WN;
#pragma unroll
    for (int pass = 0; pass < (E_GELU2 ? 2 : 1); ++pass) {
        // pass 0 of BIAS_GELU writes the pre-activation (C2; _DG: the derivative), pass 1 the activation; other epilogues have one pass
        if (E_GELU2 && pass == 0 && !a.C2) continue;
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) {
            const size_t gm = (size_t)(m0 + wr * 128 + mf * 16 + i16);
            uint2 rr[NF];
            if (R16PF) {
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) rr[nf] = rpf[mf][nf];
            } else if (E_RES) {
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) rr[nf] = *reinterpret_cast<const uint2*>(a.R + gm * a.ldr + col0 + nf * 16 + g * 4);
            }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                float v[4] = {acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]};
                if (E_GELU2 && pass == 1) {
                    gelu_act4(v, EPI == EPI_BIAS_GELU_DG ? 0 : ACT);
                } else if (EPI == EPI_BIAS_GELU_DG) {             // pass 0: the derivative
                    float d[4] = {1.f, 1.f, 1.f, 1.f};
                    gelu_grad_mul4(d, v[0], v[1], v[2], v[3], 0);
                    v[0] = d[0]; v[1] = d[1]; v[2] = d[2]; v[3] = d[3];
                } else if (E_RES) {
                    const float r0 = __uint_as_float(rr[nf].x << 16), r1 = __uint_as_float(rr[nf].x & 0xffff0000u);
                    const float r2 = __uint_as_float(rr[nf].y << 16), r3 = __uint_as_float(rr[nf].y & 0xffff0000u);
                    if (EPI == EPI_ADD_RES) { v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3; }
                    else if (EPI == EPI_MUL_RES) { v[0] *= r0; v[1] *= r1; v[2] *= r2; v[3] *= r3; }
                    else gelu_grad_mul4(v, r0, r1, r2, r3, ACT);
                }
                if (STAGED) {
                    uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>(stg + DP_STG_OFF(mf * 16 + i16, nf * 2 + (g >> 1)) + (g & 1) * 8) = pk;
                } else {
                    float* dst = reinterpret_cast<float*>(a.C) + gm * a.ldc + col0 + nf * 16 + g * 4;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
            // one row fragment (16 values per lane) at a time: without the fence the scheduler interleaves all 128 GELU
            // evaluations of a pass and the BIAS_GELU instantiation spills 116 VGPRs next to the 128 accumulators
            __builtin_amdgcn_sched_barrier(0);
        }
        if (STAGED) {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int rr0 = l >> 3, cc = l & 7;
            bf16_t* obase = ((E_GELU2 && pass == 0) ? a.C2 : reinterpret_cast<bf16_t*>(a.C));
            const int old = (E_GELU2 && pass == 0) ? a.ldc2 : a.ldc;
            obase += (size_t)(m0 + wr * 128) * old + col0 + cc * 8;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int r = p * 8 + rr0;
                const uint4 val = *reinterpret_cast<const uint4*>(stg + DP_STG_OFF(r, cc));
                if (NF == 4 || cc < 2 * NF) *reinterpret_cast<uint4*>(obase + (size_t)r * old) = val;      // 48-column wave tile: 6 of the 8 chunks
            }
            __builtin_amdgcn_wave_barrier();              // the image is rewritten by the next pass
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// one workgroup per CU for the persistent form
static int dp_num_cus() {
    static int n = 0;
    if (!n) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; }
    return n;
}

template <int EPIX, typename OutT, int NF>
static int launch_nt_dp_nf(const GemmNTArgs& a_in, hipStream_t s) {
    static bool attr_set = false;
    constexpr int LDS = 8 * 16384;          // two stages (NF = 4: exactly this, NF = 3: 112 KiB) or the epilogue's 8 x 16-KiB wave images
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_dp_kernel<EPIX, OutT, NF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    GemmNTArgs a = a_in;
    a.tiles_m = a.M / DP_BM; a.tiles_n = a.N / (NF * 64);
    AMDSEG_LAUNCH_PROF(AMDSEG_PROF_GEMM_NT, 2.0 * a.M * a.N * a.K, (gemm_nt_dp_kernel<EPIX, OutT, NF>), dim3(a.tiles_m * a.tiles_n),
                       dim3(512), LDS, s, a);
    return amdseg_launch_status();
}

// CUs the grids of a call can count on (amdseg_ctx_set_cu_budget; 0 = all of them).  An overlapped RCCL all-reduce keeps one CU per
// channel for the whole of backward -- a deep-pipeline workgroup needs a CU's entire register file, so those CUs are lost to it -- and a grid of
// exactly 256 tiles then runs TWO rounds: measured with tools/dbg/cu_hog.sh, 8 / 16 / 32 occupied CUs all cost the training step 10 % (NT launch
// average 61 -> 72 us), 64 double the weight-gradient GEMM (216 tiles).  The tile-width choice below counts rounds against the budget.
int amdseg_num_cus() { return dp_num_cus(); }
int amdseg_cu_budget() {                                    // the budget of the call's context (amdseg_ctx_set_cu_budget); none, 0 or >= the chip: every CU
    const amdseg_ctx* c = amdseg_current_ctx();
    const int b = c ? c->cu_budget : 0, cus = dp_num_cus();
    return b > 0 && b < cus ? b : cus;
}

// tile width: 256 whenever N allows it, 192 for the other multiples of 192.  Picking 192 for wave quantisation (N = 768: 256 tiles
// instead of 192, N = 2304: 3 full rounds instead of 2.25) measured NO gain in the training step (QKV 67.1 vs 68.6 us, N = 768
// K = 3072 78.3 vs 79.8, N = 768 K = 768 27.0 vs 25.0): the chip is clock/power limited under MFMA load, 192 busy CUs run as fast
// as 256 at 0.75 of the work each.
template <int EPIX, typename OutT>
int amdseg_launch_nt_dp(const GemmNTArgs& a_in, hipStream_t s) {
    const bool ok256 = (a_in.N % 256) == 0, ok192 = (a_in.N % 192) == 0;
    // ... but ROUNDS count (round 3, M = 8192 = the 4 x 2048 launch shape of run_finetune.sh): N = 2304 is 288 tiles of 256 columns = 2 rounds with
    // the second one 1/8 full, and 384 tiles of 192 = 2 rounds of 3/4 the work each: 45.0 -> 41.0 us; N = 3072 with the dual-output GELU epilogue
    // 56.6 -> 52.5 (384 -> 512 tiles).  The residual-reading epilogues lose on the narrow tile's staged stores (GELU' 57.8 -> 59.2) and stay wide.
    // Round 5: the residual-adding input-gradient GEMMs (N = 768 at M = 16384: 192 tiles of 256 columns = three quarters of ONE round) take the narrow
    // tile when that makes the single round (nearly) full: 256 tiles on 256 CUs, 64.4 -> 62 us per launch, 12.91 -> 12.78 ms per step (three
    // interleaved repetitions on one box, profiles/r05_default_switches.md).
    // Round 6 (tools/dbg/nt_addres_shapes.py, K = 3072 / 2304, us per launch wide -> narrow): the same epilogue over MORE than one round follows rounds x
    // width like the plain epilogues -- M = 24576: 126.6 / 98.8 -> 111.2 / 85.7, M = 32768 (the 8 x 4096 models: 384 wide tiles = 1.5 rounds, 512 narrow = 2):
    // 146.2 / 104.7 -> 136.3 / 97.4, M = 49152: 225.8 / 171.5 -> 202.5 / 158.2, M = 65536 (3 wide rounds = 4 narrow): unchanged, stays wide -- and a single
    // partial round takes the narrow tile whenever the narrow tiles still fit into it (M = 12288: 62.3 / 49.4 -> 58.2 / 45.9; its residual tile now arrives
    // by LDS-DMA under the last K tile, R16PF above).  longformer / PoNet / bigbird-base 8 x 4096: +0.8 / +1.2 / +1.0 % per step, same box, interleaved.
    constexpr int EB = EPI_BASE(EPIX);
    bool narrow = false;
    if (ok256 && ok192) {
        const int t256 = (a_in.M / DP_BM) * (a_in.N / 256), t192 = (a_in.M / DP_BM) * (a_in.N / 192), C = amdseg_cu_budget();
        if (EB == EPI_NONE || EB == EPI_BIAS || EB == EPI_BIAS_GELU || EB == EPI_BIAS_GELU_DG)
            narrow = 0.78f * (float)((t192 + C - 1) / C) < (float)((t256 + C - 1) / C);
        else if (EB == EPI_ADD_RES)
            narrow = t256 < C ? (t192 <= C) : 0.78f * (float)((t192 + C - 1) / C) < (float)((t256 + C - 1) / C);
    }
    constexpr bool direct_only = EB == EPI_BIAS_SPLIT || EB == EPI_GELU_BWD_SPLIT || EB == EPI_BIAS_GELU_SPLIT ||
                                 EB == EPI_BIAS_GELU_DG8 || EB == EPI_MUL_RES8;    // epilogues of the 256-wide tile only
    if (direct_only && !ok256) return AMDSEG_ERR_SHAPE;
    const bool use192 = !direct_only && ok192 && (!ok256 || narrow);
    if (use192) return launch_nt_dp_nf<EPIX, OutT, 3>(a_in, s);
    return launch_nt_dp_nf<EPIX, OutT, 4>(a_in, s);
}


#define DP_INST(E, T) template int amdseg_launch_nt_dp<E, T>(const GemmNTArgs&, hipStream_t);
DP_INST(EPI_NONE, bf16_t) DP_INST(EPI_NONE, float) DP_INST(EPI_BIAS, bf16_t) DP_INST(EPI_BIAS, float) DP_INST(EPI_BIAS_GELU, bf16_t)
DP_INST(EPI_ADD_RES, bf16_t) DP_INST(EPI_ADD_RES, float) DP_INST(EPI_GELU_BWD, bf16_t)
DP_INST(EPI_BIAS_GELU_TANH, bf16_t) DP_INST(EPI_GELU_BWD_TANH, bf16_t) DP_INST(EPI_BIAS_SPLIT, bf16_t) DP_INST(EPI_GELU_BWD_SPLIT, bf16_t)
DP_INST(EPI_BIAS_GELU_SPLIT, float) DP_INST(EPI_BIAS_GELU_DG, bf16_t) DP_INST(EPI_MUL_RES, bf16_t) DP_INST(EPI_BIAS_GELU_DG8, bf16_t) DP_INST(EPI_MUL_RES8, bf16_t) DP_INST(EPI_BIAS_GELU_DG8_TANH, bf16_t)


// ==================================================================================================== gemm_tn, deep pipeline
// Weight gradients  C_p[N_p, K_p] (+)= sum_m A_p[m, N_p] . B_p[m, K_p]  (dW = dY^T X of the encoder's four Linears, one grouped
// launch per layer; same reference lines as gemm_tn_kernel in gemm.hip).  256(N) x 128(K') tile per 512-thread workgroup:
//   * BOTH operands are k(= token)-strided: [64 m][64] LDS tile images, fragments gathered with ds_read_b64_tr_b16.  k-slot
//     (g, j) <-> rows kh*32 + g*4 + j and kh*32 + 16 + g*4 + j - 4, so that one 32-lane group touches 8 CONSECUTIVE rows x
//     32 B, which tn_swz spreads over all 64 banks (rows g*8.. gave 2-way conflicts; SQ_LDS_BANK_CONFLICT = 0 now);
//   * 8 waves = 2 K-halves (kh: tokens kh*32..+32 of every 64-token K tile) x 2 N-halves x 2 K'-halves, wave tile 128 x 64 over
//     half the K tile (8 x 4 fragments of v_mfma_f32_16x16x32_bf16, operands swapped so a lane owns one output row and 4
//     consecutive columns): 24 gathers per 32 MFMAs; the two K-halves are summed through LDS once, after the K loop;
//   * software pipeline inside every wave: the fragments of K tile kt+1 are gathered BETWEEN the MFMAs of K tile kt (A
//     fragments single-buffered: reloaded right after their last MFMA; B fragments double-buffered).  A wave issues only one
//     ds_read_b64_tr per ~16 clk, so a separate load phase (the first version of this kernel: 40 gathers, barrier, 32 MFMAs,
//     barrier, two staggered groups) was bound by gather issue: 321 -> 257 us on 256 tiles of M = 16384;
//   * 3-stage ring of 48 KiB (A 4 images + B 2 images) filled by DMA two to three K tiles ahead, one piece between every
//     4 MFMAs, counted vmcnt, ONE barrier per K tile;
//   * the gathers are inline asm: in front of a ds_read_b64_tr_b16 BUILTIN that follows an LDS-DMA the compiler emits
//     s_waitcnt vmcnt(0) (it cannot prove they do not alias), which drained the whole ring once per K tile.  Their lgkmcnt wait
//     is explicit (top of every K tile) and ties every fragment register, so nothing derived from a gathered value can be
//     scheduled before its data arrived (a compiler-inserted v_bfi on a fresh asm result once consumed stale registers).
// Measured (tools/ubench/gemm_tn_dp.cpp, M = 16384, one tile per CU): 72 tiles 194 us, 256 tiles 257 us = 1071 TFLOP/s
// (MFMA-only ablation 145 / 157 us; without DMA 169 / 205 us).
#define TN_STG 49152
#define TN_LDS (3 * TN_STG)
typedef int tn_i32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 tn_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int tn_swz(int r) { return (((r >> 1) & 3) << 1) | ((r >> 3) & 1); }
// wave-uniform pointer the loop optimiser cannot turn into per-lane 64-bit induction variables (24 VGPRs of DMA addresses otherwise)
__device__ __forceinline__ const bf16_t* tn_uniform(const bf16_t* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const bf16_t*)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ const int* tn_uniform_i(const int* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const int*)(((uint64_t)hi << 32) | lo);
}

__global__ __launch_bounds__(512, 1) void gemm_tn_dp_kernel(GemmTNArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dr = w >> 2, wq = w & 3;                       // DMA duty: images 2dr, 2dr+1 of A and dr of B, 8-row pieces 2wq, 2wq+1
    const int kh = w >> 2, wr = (w >> 1) & 1, wc = w & 1;    // compute role
    const int g = l >> 4, i16 = l & 15;
    const int t = xcd_remap(blockIdx.x, a.total_tiles);      // each XCD walks a contiguous run of tiles (shared A panels)
    int pi = 0;
#pragma unroll
    for (int i = 1; i < AMDSEG_MAX_GROUP; ++i)
        if (i < a.nprob && t >= a.p[i].tile_begin) pi = i;
    const TNProblem P = a.p[pi];
    const int lt = t - P.tile_begin;
    // tile order inside a problem: an XCD walks a contiguous run of ~tiles / 8 tiles whose workgroups move through the tokens together, so an operand
    // panel that several of them share is fetched once per XCD.  With K' fastest, a problem with few N tiles and many K' tiles (dW of the FFN
    // down-projection: 3 x 24) has every XCD read ALL K' panels of B (= h, 100 MB) once per N tile it touches: 290 MB for that problem; with N fastest
    // an XCD's 27 tiles are a 3 x 9 block: all of A (25 MB) + 9 of 24 B panels.
    const int tiles_n_ = P.N / 256;
    const bool nfast = tiles_n_ < P.tiles_k;
    const int tn = nfast ? lt % tiles_n_ : lt / P.tiles_k, tk = nfast ? lt / tiles_n_ : lt % P.tiles_k;
    const int n0 = tn * 256, k0 = tk * 128;
#define TN_TILE_A(s, i) (smem + (s) * TN_STG + (i) * 8192)
#define TN_TILE_B(s, j) (smem + (s) * TN_STG + 32768 + (j) * 8192)
    uint32_t offA[4], offB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wq * 2 + q) * 8 + (l >> 3), c = (l & 7) ^ tn_swz(r);
        offA[q] = (r * P.lda + (dr * 2) * 64 + c * 8) * 2;          // byte offsets from the K tile's first row
        offA[2 + q] = (r * P.lda + (dr * 2 + 1) * 64 + c * 8) * 2;
        offB[q] = (r * P.ldb + dr * 64 + c * 8) * 2;
    }
    const bf16_t* pA = P.A + n0;
    const bf16_t* pB = P.B + k0;
    // piece j of this wave's DMA duty for K tile kt: j < 4 -> A image 2dr + (j >> 1), 8-row piece 2wq + (j & 1); j >= 4 -> B image dr
    // (round 4, second session) the wave-uniform parts of a K tile's six pieces are made ONCE per K tile -- the two global row bases (SGPR pairs)
    // and the LDS address of the stage as a 32-bit scalar that the compiler adds the piece's constant to and writes to m0 -- instead of once per
    // piece (64-bit multiplies, two v_readfirstlane and a generic-pointer null check per piece: ~60 scalar + 12 vector instructions per K tile)
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
    const uint32_t ldsAw = __builtin_amdgcn_readfirstlane(lds0 + (dr * 2) * 8192 + (wq * 2) * 1024);
    const uint32_t ldsBw = __builtin_amdgcn_readfirstlane(lds0 + 32768 + dr * 8192 + (wq * 2) * 1024);
    // (element offsets of a K tile's first row fit 32 bits: the launcher sends shapes with M * ld >= 2^31 to the 128 x 128 kernel)
#define TN_DMA_SETUP(s, kt) const bf16_t* const gA_ = tn_uniform(pA + (uint32_t)((kt) * 64) * (uint32_t)P.lda); const bf16_t* const gB_ = tn_uniform(pB + (uint32_t)((kt) * 64) * (uint32_t)P.ldb); \
        const uint32_t lsA_ = ldsAw + (uint32_t)(s) * TN_STG, lsB_ = ldsBw + (uint32_t)(s) * TN_STG
#define TN_DMA_PIECE(j) do { if ((j) < 4) amdseg_glds16_saddr_lds(gA_, offA[j], lsA_ + ((j) >> 1) * 8192 + ((j) & 1) * 1024); \
        else amdseg_glds16_saddr_lds(gB_, offB[(j) - 4], lsB_ + ((j) - 4) * 1024); } while (0)
#define TN_DMA(s, kt) do { TN_DMA_SETUP(s, kt); _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) TN_DMA_PIECE(j_); } while (0)
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K loop: all M / 64 token tiles, or only the caller's runs of tiles that are not known zeros.  Only the DMA prefetch needs tile
    // indices (the compute side consumes the ring in order), so ONE wave-uniform iterator walks the runs: (it_cur, it_end) = the run being
    // prefetched, refilled by a scalar load once per run (waited for inside its own asm, right after the top-of-tile lgkmcnt(0): the
    // compiler turns loads behind the "memory"-clobbering asm of this kernel into vector loads, whose vmcnt wait would drain the ring)
    const int* runs = a.runs;
    if (runs && __builtin_amdgcn_readfirstlane(*a.zguard) != 0) runs = nullptr;
    int nk = a.M / 64, it_cur = 0, it_end = nk, it_r = 0, it_nr = 1;
    if (runs) {
        nk = __builtin_amdgcn_readfirstlane(a.counts[0]); it_nr = __builtin_amdgcn_readfirstlane(a.counts[1]);
        it_cur = it_nr > 0 ? __builtin_amdgcn_readfirstlane(runs[0]) : 0;
        it_end = it_nr > 0 ? __builtin_amdgcn_readfirstlane(runs[1]) : 1;
    }
    // t = the next tile to prefetch, ph = t mod tiles_k (which K' tile of the row sums its columns: below); past the last tile it keeps
    // returning the last one (re-fetched into a stage nobody reads)
    int it_ph = it_cur % P.tiles_k;
#define TN_NEXT(t, ph) do { t = it_cur; ph = it_ph; \
        if (it_cur + 1 < it_end) { ++it_cur; it_ph = it_ph + 1 == P.tiles_k ? 0 : it_ph + 1; } \
        else if (it_r + 1 < it_nr) { ++it_r; tn_i32x2 rv_; \
            asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rv_) : "s"(tn_uniform_i(runs + 2 * it_r)) : "memory"); \
            it_cur = rv_.x; it_end = rv_.y; it_ph = it_cur % P.tiles_k; } } while (0)
    // fused bias gradient: column sums of A (= dY) for free -- the A fragments are in registers anyway.  Tile (tn, tk) sums the K tiles
    // kt = tk (mod tiles_k) with v_dot2c_f32_bf16 against (1, 1) (4 per fragment, only the wc == 0 waves, ~1/tiles_k of the K tiles),
    // writes its partial to colsum_part[tk][n]; the host queues the sum over tk (deterministic second stage)
    const bool cs_on = P.colsum_part != nullptr && wc == 0;
    float accb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) accb[i] = 0.f;
    int ph0, ph1, ph2;                                      // (token tile mod tiles_k) of the K tiles kt, kt+1, kt+2 (set by the prefetch iterator)
    { int t0_, t1_, t2_; TN_NEXT(t0_, ph0); TN_NEXT(t1_, ph1); TN_NEXT(t2_, ph2);   // always three tiles (clamped): the vmcnt arithmetic below is uniform
      TN_DMA(0, t0_); TN_DMA(1, t1_); TN_DMA(2, t2_); }
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // lane addresses of the gathers inside stage 0: lane (i16, g) of fragment column block c4 reads rows kh*32 + g*4 + (i16 >> 2) (+16)
    uint32_t laA[4], laB[4];
    {
        const int row = kh * 32 + g * 4 + (i16 >> 2);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int c = c4 * 2 + ((i16 & 3) >> 1);
            const uint32_t o = (uint32_t)(uintptr_t)LDS_PTR(char, smem) + row * 128 + ((c ^ tn_swz(row)) << 4) + (i16 & 1) * 8;
            laA[c4] = o + wr * 2 * 8192;
            laB[c4] = o + 32768 + wc * 8192;
        }
    }
    tn_i32x2 fa[8][2], fb0[4][2], fb1[4][2];
// the gathers as builtins (this kernel's LDS-DMA is inline asm, so the compiler knows of no vector memory operation it would have to wait for in
// front of them): it tracks their lgkmcnt itself and may place a fragment's two halves in the adjacent registers the MFMA wants (the asm form pays
// two v_mov + s_nop for six of the eight A fragments per K tile; 245 -> 220 VGPRs, stand-alone 266-272 -> 261-264 us, in the step 211.8 -> 209.0)
typedef short tn_v4s __attribute__((ext_vector_type(4)));
#define TN_RD(dst, addr, off) dst = __builtin_bit_cast(tn_i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_v4s*)(uintptr_t)((addr) + (off))))
// (a probe with A fragments 6 and 7 double-buffered -- their gathers issued early in the body instead of behind its last MFMAs -- measured neutral in round 4,
//  profiles/r04b_tn_counters.md: the K tile's barrier waits for the slowest of eight waves, not for this wave's last gathers; removed in round 6)
#define TN_FA(nf, AC) (fa[nf])
#define TN_LDA_TO(dst, nf) do { TN_RD((dst)[0], aA[(nf) & 3], ((nf) >> 2) * 8192); TN_RD((dst)[1], aA[(nf) & 3], ((nf) >> 2) * 8192 + 2048); } while (0)
#define TN_LDA(nf) TN_LDA_TO(fa[nf], nf)
#define TN_LDB(e, FB) do { TN_RD(FB[e][0], aB[e], 0); TN_RD(FB[e][1], aB[e], 2048); } while (0)
#define TN_CAT(x) __builtin_bit_cast(bf16x8, __builtin_shufflevector(x[0], x[1], 0, 1, 2, 3))
#define TN_SB() __builtin_amdgcn_sched_barrier(0)
#define TN_MF(nf, e, FB, AC) acc[nf][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(TN_CAT(FB[e]), TN_CAT(TN_FA(nf, AC)), acc[nf][e], 0, 0, 0)
#define TN_DOT2(x, c) __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(tn_bf2, x), __builtin_bit_cast(tn_bf2, 0x3f803f80), c, false)
// (the elements are copied to plain ints first: __builtin_bit_cast applied directly to an ext-vector ELEMENT expression reads element 0
//  whatever the index -- seen with hipcc 7.2: `bit_cast<bf2>(v.y)` compiled to the same register as `bit_cast<bf2>(v.x)`)
#define TN_CS(nf, AC) do { const tn_i32x2 c0_ = TN_FA(nf, AC)[0], c1_ = TN_FA(nf, AC)[1]; const int e0_ = c0_.x, e1_ = c0_.y, e2_ = c1_.x, e3_ = c1_.y; \
                       accb[nf] = TN_DOT2(e0_, accb[nf]); accb[nf] = TN_DOT2(e1_, accb[nf]); \
                       accb[nf] = TN_DOT2(e2_, accb[nf]); accb[nf] = TN_DOT2(e3_, accb[nf]); } while (0)
#define TN_WAIT_FRAGS(FB) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    uint32_t aA[4], aB[4];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) { aA[c4] = laA[c4]; aB[c4] = laB[c4]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) TN_LDB(e, fb0);
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) TN_LDA(nf);
    int sc = 0, sn = 1;                                     // stages of K tiles kt (free: refilled with kt+3) and kt+1 (gathered now)
    // K tile kt.  Top: this wave's gathers of tile kt (issued during kt-1) and its DMA pieces of tile kt+1 have landed; after the
    // barrier that holds for every wave, so stage sn may be read and stage sc (tile kt, now in registers everywhere) refilled.
    // the bias-gradient column sums run in ~1 of tiles_k K tiles: as ONE block in front of the K tile's MFMAs (all eight A fragments of tile kt are
    // in registers there) behind one branch -- as eight `if (cs_now)` inside the MFMA stream they were eight taken branches per K tile in the
    // common case; a second copy of the body for the tiles that sum spilled registers (256 VGPRs + 604 B of scratch)
#define TN_BODY_CORE(FC, FN, AC) do { \
        _Pragma("unroll") for (int nf = 0; nf < 4; ++nf) { \
            TN_MF(nf, 0, FC, 0); TN_SB(); TN_MF(nf, 1, FC, 0); TN_SB(); TN_LDB(nf, FN); TN_SB(); TN_MF(nf, 2, FC, 0); TN_SB(); TN_MF(nf, 3, FC, 0); TN_SB(); \
            TN_LDA(nf); TN_SB(); \
            if (nf >= 1) { TN_DMA_PIECE(nf - 1); TN_SB(); } } \
        _Pragma("unroll") for (int nf = 4; nf < 8; ++nf) { \
            TN_MF(nf, 0, FC, 0); TN_MF(nf, 1, FC, 0); TN_MF(nf, 2, FC, 0); TN_MF(nf, 3, FC, 0); TN_SB(); \
            TN_LDA(nf); TN_SB(); \
            if (nf <= 6) { TN_DMA_PIECE(nf - 1); TN_SB(); } } } while (0)
#define TN_ACSEL(AC) 0
#define TN_BODY(FC, FN, AC) do { \
        TN_WAIT_FRAGS(FC); \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          /* every K tile issues 6 pieces: tile kt+1 has landed */ \
        TN_SB(); __builtin_amdgcn_s_barrier(); TN_SB(); \
        const bool cs_now = cs_on && ph0 == tk; \
        ph0 = ph1; ph1 = ph2; \
        int ktd_; TN_NEXT(ktd_, ph2);                  /* tile of iteration kt + 3 */ \
        TN_DMA_SETUP(sc, ktd_); \
        _Pragma("unroll") for (int c4 = 0; c4 < 4; ++c4) { aA[c4] = laA[c4] + sn * TN_STG; aB[c4] = laB[c4] + sn * TN_STG; } \
        __builtin_amdgcn_s_setprio(1); \
        if (cs_now) { _Pragma("unroll") for (int nf = 0; nf < 8; ++nf) TN_CS(nf, TN_ACSEL(AC)); TN_SB(); } \
        TN_BODY_CORE(FC, FN, TN_ACSEL(AC)); \
        __builtin_amdgcn_s_setprio(0); \
        sc = sc == 2 ? 0 : sc + 1; sn = sn == 2 ? 0 : sn + 1; } while (0)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        TN_BODY(fb0, fb1, 0);
        ++kt; TN_BODY(fb1, fb0, 1); --kt;
    }
    if (kt < nk) TN_BODY(fb0, fb1, 0);
    TN_WAIT_FRAGS(fb0);                                     // the last K tile gathered a (never used) tile nk: retire it before
    TN_WAIT_FRAGS(fb1);                                     // these registers and the ring are reused
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // ... and the clamped re-fetches of the last tiles
    __builtin_amdgcn_s_barrier();
    // K-half exchange through LDS: wave (kh, wr, wc) keeps row fragments nf = kh*4 .. +4 and hands the other four to its partner
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
    // lane owns row n = (kh*4 + i)*16 + i16 and columns e*16 + g*4 .. +4 of the wave tile: 16-B fp32 read-modify-write
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* crow = P.C + (size_t)(n0 + wr * 128 + (kh * 4 + i) * 16 + i16) * P.ldc + k0 + wc * 64 + g * 4;
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
    if (P.colsum_part != nullptr) {                         // workgroup-uniform
        __syncthreads();                                    // the K-half exchange above has been read
        float* csx = reinterpret_cast<float*>(smem);        // [2 wr][8 nf][16] partial sums of the kh = 1 waves
        if (wc == 0) {
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) {                // the 4 lane groups hold disjoint token slots of the same column
                accb[nf] += __shfl_xor(accb[nf], 16, 64);
                accb[nf] += __shfl_xor(accb[nf], 32, 64);
            }
            if (kh == 1 && g == 0) {
#pragma unroll
                for (int nf = 0; nf < 8; ++nf) csx[(wr * 8 + nf) * 16 + i16] = accb[nf];
            }
        }
        __syncthreads();
        if (wc == 0 && kh == 0 && g == 0) {
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
                P.colsum_part[(size_t)tk * P.N + n0 + wr * 128 + nf * 16 + i16] = accb[nf] + csx[(wr * 8 + nf) * 16 + i16];
        }
    }
}

int amdseg_launch_tn_dp(const GemmTNArgs& a128, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_dp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TN_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    GemmTNArgs a = a128;                                     // re-tile: 256 x 128 tiles, K' fastest
    int tiles = 0;
    for (int i = 0; i < a.nprob; ++i) {
        a.p[i].tile_begin = tiles; a.p[i].tiles_k = a.p[i].Kp / 128;
        tiles += (a.p[i].N / 256) * (a.p[i].Kp / 128);
    }
    a.total_tiles = tiles;
    double work = 0;
    for (int i = 0; i < a.nprob; ++i) work += 2.0 * a.M * a.p[i].N * a.p[i].Kp;
    AMDSEG_LAUNCH_PROF(AMDSEG_PROF_GEMM_TN, work, gemm_tn_dp_kernel, dim3(tiles), dim3(512), TN_LDS, s, a);
    return amdseg_launch_status();
}
