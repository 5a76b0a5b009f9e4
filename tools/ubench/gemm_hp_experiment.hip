// EXPERIMENT, not part of libamdseg (round 3): measured slower than the one-tile-per-workgroup kernel, see profiles/r03_gemm_epilogue_overlap.md.
// To rebuild the measurement: copy this file to spokennlp_amd/csrc/gemm_hp.hip, add it to build.py's SOURCES, declare
// `template <int EPIX> int amdseg_launch_nt_hp(const GemmNTArgs&, hipStream_t);` in gemm_epi.h and dispatch the two GELU epilogues to it in
// gemm.hip's launch_nt (git show of the commit that introduced this file has the hunk).  Numerics were not finished: one of the large-shape
// comparisons against the 256 x 256 kernel still failed when the experiment was stopped.
// gemm_nt for the two GELU epilogues, PERSISTENT with the epilogue of tile i running under the main loop of tile i + 1.
// (same reference lines as gemm_dp.hip: BertIntermediate's Linear + GELU, [hf] models/bert/modeling_bert.py:325-337, and the dgrad
// du = (d_out . W2) * gelu'(u) of its autograd backward.)
//
// Why: at K = 768 the 256 x 256 kernel of gemm_dp.hip spends 56-61 us in its main loop and another 35-43 us in the bias + GELU (two
// outputs) / GELU' x R epilogues -- erf-GELU math on 128 accumulators per lane and the stores, all eight waves at once, MFMA pipes idle
// (profiles/r01_gemm_experiments.md: 99.0 / 89.5 / 56.1 us full / without stores / main loop only).  A second accumulator set for a 256 x 256
// tile does not fit (223 VGPRs already), so here the tile is 128 x 256: wave tile 64 x 64 = 64 accumulators, TWO sets.  One workgroup per
// CU walks its tiles; while set A accumulates tile i + 1, the eight 8-value slices of set B (tile i) are converted and stored one per K
// step, in the half of the phase in which this wave only issues loads and the other row group owns the MFMA pipe.
//   * 8 waves = 2 row groups x 4 column waves, wave tile 64 x 64 (4 x 4 fragments of v_mfma_f32_16x16x32_bf16);
//   * three 48-KiB LDS stages (A: 2 images of [64 rows][64 k], B: 4), filled by LDS-DMA two K steps ahead; ONE phase per K step:
//     [16 fragment reads + 6 DMA pieces + counted vmcnt + epilogue slice | s_barrier | 32 MFMAs | s_barrier], the row groups staggered by
//     one barrier as in gemm_dp.hip;
//   * vmcnt retires loads AND stores in issue order on gfx950 (tools/ubench/vmcnt_order.cpp: 0 of 21 M lane-trials saw a cold load
//     pending after `load; store; s_waitcnt vmcnt(1)`): a phase queues [6 DMA pieces][next slice's R / bias][this slice's stores], and
//     `s_waitcnt vmcnt(6 + stores)` behind the NEXT phase's pieces retires the old pieces and the prefetched values while the stores
//     (their HBM acknowledgement takes longer than a phase) and the new pieces stay in flight;
//   * the slice's R (GELU') or bias (GELU) values are fetched one phase ahead by inline-asm loads: a compiler-visible load would get a
//     compiler-counted wait that knows nothing of the asm DMA pieces behind it and would drain them.
// Costs against the 256 x 256 tile: 16 instead of 12 fragment reads per 32 MFMAs, 1.5 x the DMA bytes per flop; only the last tile of a
// workgroup (1 of 6 at M = 16384, N = 3072) pays its epilogue in the open.
#include "common.h"
#include "amdseg_internal.h"
#include "gemm_epi.h"

#define HP_BM 128
#define HP_BN 256
#define HP_STAGE (6 * 8192)
#define HP_LDS (3 * HP_STAGE)

__device__ __forceinline__ int hp_swz(int r) { const int p = (r >> 1) & 7; return p ^ (((p + 2) >> 2) & 1); }
__device__ __forceinline__ int hp_swzB(int r) { const int p = ((r >> 3) & 3) * 2 + ((r >> 1) & 1); return p ^ (((p + 2) >> 2) & 1); }
__device__ __forceinline__ bf16x8 hp_frag(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ hp_swz(r)) << 4));
}
__device__ __forceinline__ bf16x8 hp_fragB(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ hp_swzB(r)) << 4));
}

// epilogue memory traffic in the SGPR-base + 32-bit lane-offset form: the lane offsets are the same for every slice of every tile (ONE
// VGPR per matrix), the slice / tile part of the address is scalar arithmetic.  With per-lane 64-bit addresses the optimiser hoisted all
// eight slices' store and load addresses out of the phases: ~50 VGPRs live across the whole tile, and the finished accumulators spilled.
typedef uint32_t hp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void hp_aux_load(hp_u32x4& dst, const void* sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void hp_store16(const void* sbase, uint32_t voff, hp_u32x4 d) {
    asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(voff), "v"(d), "s"(sbase) : "memory");
}

// tile t (XCD-contiguous inside a round of G workgroups, GROUP_M-grouped) -> (m0, n0)
__device__ __forceinline__ void hp_coords(const GemmNTArgs& a, int t, int& m0, int& n0) {
    const int gsz_full = GROUP_M * a.tiles_n;
    const int grp = t / gsz_full, first_m = grp * GROUP_M;
    const int gmn = min(a.tiles_m - first_m, GROUP_M);
    const int rem = t - grp * gsz_full;
    m0 = (first_m + rem % gmn) * HP_BM; n0 = (rem / gmn) * HP_BN;
}

struct HpState {
    // DMA prefetch iterator (wave-uniform): the K step two ahead of the one being computed
    const bf16_t* pA; const bf16_t* pB;      // this wave group's A rows / B rows of the prefetch tile
    int p_kt, p_j;                           // its K step and the index of its tile in this workgroup's sequence
    int sc;                                  // LDS stage of the K step being computed
    // epilogue of the previous tile
    int em0, en0;
};

template <int EPIX>
__global__ __launch_bounds__(512, 1) void gemm_nt_hp_kernel(GemmNTArgs a, int ntiles) {
    constexpr int EPI = EPI_BASE(EPIX); constexpr int ACT = EPI_ACT(EPIX); (void)ACT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3, wq = w & 3;
    const int g = l >> 4, i16 = l & 15;
    const int G = gridDim.x;
    const int wg = xcd_remap(blockIdx.x, G);
    const int nk = a.K / 64;
#define HP_TILE_A(s, i) (smem + (s) * HP_STAGE + (i) * 8192)
#define HP_TILE_B(s, i) (smem + (s) * HP_STAGE + (2 + (i)) * 8192)
    uint32_t offA[2], offB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wq * 2 + q) * 8 + (l >> 3), c = (l & 7) ^ hp_swz(r), cb = (l & 7) ^ hp_swzB(r);
        offA[q] = (r * a.lda + c * 8) * 2;
        offB[q] = (r * a.ldb + cb * 8) * 2;
        offB[2 + q] = ((64 + r) * a.ldb + cb * 8) * 2;
    }
    // six 1-KiB pieces per wave and K step: 2 of this group's A image, 4 of its two B images
#define HP_DMA(s, pa, pb) do { \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) amdseg_glds16_saddr(pa, offA[q_], HP_TILE_A(s, wr) + (wq * 2 + q_) * 1024); \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) \
            amdseg_glds16_saddr(pb, offB[i_ * 2 + q_], HP_TILE_B(s, wr * 2 + i_) + (wq * 2 + q_) * 1024); } while (0)

    HpState st;
    st.sc = 0; st.p_kt = 0; st.p_j = 0; st.em0 = 0; st.en0 = 0;
    auto set_prefetch_tile = [&](int j) {                  // (wave-uniform) base pointers of this workgroup's j-th tile; past the end: keep the last
        const int t = j * G + wg;
        if (t < ntiles) {
            int m0, n0;
            hp_coords(a, t, m0, n0);
            st.pA = a.A + (size_t)(m0 + wr * 64) * a.lda;
            st.pB = a.B + (size_t)(n0 + wr * 128) * a.ldb;
        }
    };
    auto advance_prefetch = [&]() {
        if (++st.p_kt == nk) {
            if ((st.p_j + 1) * G + wg < ntiles) { st.p_kt = 0; ++st.p_j; set_prefetch_tile(st.p_j); }
            else st.p_kt = nk - 1;                         // exhausted: re-fetch the last step into a stage nobody reads any more
        }
    };
    if (wg >= ntiles) return;                              // (the launcher never starts more workgroups than tiles)
    set_prefetch_tile(0);
    HP_DMA(0, st.pA, st.pB); advance_prefetch();
    HP_DMA(1, st.pA + st.p_kt * 64, st.pB + st.p_kt * 64); advance_prefetch();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();             // stagger: group 1 runs one barrier behind group 0

    bf16x8 fa[4][2], fb[4][2];
#define HP_LOAD_A(s) _Pragma("unroll") for (int f = 0; f < 4; ++f) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        fa[f][kk] = hp_frag(HP_TILE_A(s, wr), f * 16 + i16, kk * 4 + g);
#define HP_LOAD_B(s) _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        fb[e][kk] = hp_fragB(HP_TILE_B(s, wc), (e >> 1) * 32 + (i16 >> 2) * 8 + (e & 1) * 4 + (i16 & 3), kk * 4 + g);
#define HP_MFMA(ACC) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int f = 0; f < 4; ++f) \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) \
        ACC[f][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[e][kk], fa[f][kk], ACC[f][e], 0, 0, 0);
#ifndef AMDSEG_HP_MFMA_PRIO
#define AMDSEG_HP_MFMA_PRIO 1
#endif
#define HP_MID() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); if (AMDSEG_HP_MFMA_PRIO) __builtin_amdgcn_s_setprio(AMDSEG_HP_MFMA_PRIO); } while (0)
#define HP_END() do { if (AMDSEG_HP_MFMA_PRIO) __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- epilogue slices.  Slice u = (mf, ep) = (u >> 1, u & 1): lane owns row mf*16 + i16 of the wave's 64 and the 8 consecutive columns
    // ep*32 + g*8 .. +8 of its 64 (the permuted B fragment rows of HP_LOAD_B make the accumulators of fragments 2ep, 2ep+1 consecutive).
    // aux = what the slice needs from memory, fetched one phase ahead: 8 bias floats (GELU) or 8 bf16 of R (GELU')
    hp_u32x4 aux0, aux1;
    const uint32_t offC = (uint32_t)(i16 * a.ldc + g * 8) * 2, offC2 = (uint32_t)(i16 * a.ldc2 + g * 8) * 2;
    const uint32_t offR = (uint32_t)(i16 * a.ldr + g * 8) * 2, offBias = (uint32_t)(g * 8) * 4;
    auto aux_fetch = [&](int u, int em0, int en0) {
        const int mf = u >> 1, ep = u & 1;
        if (EPI == EPI_BIAS_GELU) {
            const float* bp = a.bias + en0 + wc * 64 + ep * 32;
            hp_aux_load(aux0, bp, offBias); hp_aux_load(aux1, bp + 4, offBias);
        } else hp_aux_load(aux0, a.R + (size_t)(em0 + wr * 64 + mf * 16) * a.ldr + en0 + wc * 64 + ep * 32, offR);
    };
    // VMEM instructions a slice issues AFTER fetching the next slice's aux values: its stores (C2 + C for GELU -- this kernel is only
    // dispatched with a pre-activation output -- or C alone).  vmcnt retires in issue order, so a phase's queue is
    //     [6 DMA pieces] [aux of the next slice] [this slice's stores]      and the next phase's      [6 DMA pieces]
    // and `s_waitcnt vmcnt(6 + STORES)` behind those pieces means: last phase's pieces and the aux values have landed, while its stores
    // (HBM write acknowledgements take longer than a phase) and the new pieces stay in flight.
    constexpr int STORES = EPI == EPI_BIAS_GELU ? 2 : 1;
#define HP_SLICE(ACC, u, NEXT) do { \
        const int mf_ = (u) >> 1, ep_ = (u) & 1; \
        float v[8]; \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) { v[r] = ACC[mf_][2 * ep_][r]; v[4 + r] = ACC[mf_][2 * ep_ + 1][r]; } \
        const size_t row_ = (size_t)(st.em0 + wr * 64 + mf_ * 16);      /* wave-uniform */ \
        const int col_ = st.en0 + wc * 64 + ep_ * 32; \
        float rf[8]; \
        if (EPI == EPI_BIAS_GELU) { \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) { v[q] += __uint_as_float(aux0[q]); v[4 + q] += __uint_as_float(aux1[q]); } \
        } else { \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) { rf[2 * q] = __uint_as_float(aux0[q] << 16); rf[2 * q + 1] = __uint_as_float(aux0[q] & 0xffff0000u); } \
        } \
        /* the aux registers are consumed: fetch the next slice's values BEFORE this slice's stores enter the queue */ \
        if (EPI == EPI_BIAS_GELU) asm volatile("" : "+v"(v[0]), "+v"(v[3]), "+v"(v[4]), "+v"(v[7]) :: "memory"); \
        else asm volatile("" : "+v"(rf[0]), "+v"(rf[2]), "+v"(rf[4]), "+v"(rf[6]) :: "memory"); \
        if ((NEXT) >= 0) aux_fetch((NEXT) < 0 ? 0 : (NEXT), st.em0, st.en0); \
        if (EPI == EPI_BIAS_GELU) { \
            hp_store16(a.C2 + row_ * a.ldc2 + col_, offC2, (hp_u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])}); \
            gelu_act4(v, ACT); gelu_act4(v + 4, ACT); \
        } else { \
            gelu_grad_mul4(v, rf[0], rf[1], rf[2], rf[3], ACT); gelu_grad_mul4(v + 4, rf[4], rf[5], rf[6], rf[7], ACT); \
        } \
        hp_store16(reinterpret_cast<bf16_t*>(a.C) + row_ * a.ldc + col_, offC, \
                   (hp_u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])}); } while (0)
    // the counted wait of a phase; ties the prefetched registers so no use can move above it
#define HP_WAIT(N) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(aux0), "+v"(aux1) : "n"(N) : "memory")

    // (ablation builds: -DAMDSEG_HP_ABL_NOSLICE drops the overlapped slices -- wrong results, main-loop timing only)
#ifdef AMDSEG_HP_ABL_NOSLICE
#define HP_ABL_SLICES(ACC_E, SLICE)
#else
#define HP_ABL_SLICES(ACC_E, SLICE) \
        if ((SLICE) == -2) aux_fetch(0, st.em0, st.en0); \
        if ((SLICE) >= 0) HP_SLICE(ACC_E, ((SLICE) < 0 ? 0 : (SLICE)), ((SLICE) < 7 ? (SLICE) + 1 : -1));
#endif
    // one K step.  SLICE: -1 = no epilogue work in this phase; -2 = only fetch slice 0's aux values; 0..7 = that slice of ACC_E.
    // PREV_STORES = stores the PREVIOUS phase issued (compile-time: the phases of a tile are unrolled)
#define HP_PHASE(ACC_M, ACC_E, SLICE, PREV_STORES) do { \
        { const int s2_ = st.sc == 0 ? 2 : st.sc - 1; HP_DMA(s2_, st.pA + st.p_kt * 64, st.pB + st.p_kt * 64); advance_prefetch(); } \
        HP_WAIT(6 + (PREV_STORES)); \
        HP_ABL_SLICES(ACC_E, SLICE) \
        /* the fragments are read AFTER the slice: its ~40 temporaries live in the 64 registers the fragments will take (with the reads \
           first the kernel needs > 256 VGPRs and spills the finished accumulators) */ \
        __builtin_amdgcn_sched_barrier(0); \
        HP_LOAD_B(st.sc) HP_LOAD_A(st.sc) \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        HP_MID(); \
        HP_MFMA(ACC_M) \
        HP_END(); \
        st.sc = st.sc == 2 ? 0 : st.sc + 1; } while (0)

    f32x4 acc0[4][4], acc1[4][4];
    aux0 = (hp_u32x4){0u, 0u, 0u, 0u}; aux1 = aux0;
#define HP_ZERO(ACC) _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j2 = 0; j2 < 4; ++j2) ACC[i][j2] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one tile: 9 unrolled phases (the slices need compile-time register indices and the waits compile-time store counts) + the
    // remaining nk - 9 plain ones (K >= 576).  The first tile of a workgroup has no predecessor: plain phases only.
#define HP_TILE(ACC_M, ACC_E) do { \
        HP_ZERO(ACC_M) \
        HP_PHASE(ACC_M, ACC_E, -2, 0); \
        HP_PHASE(ACC_M, ACC_E, 0, 0); HP_PHASE(ACC_M, ACC_E, 1, STORES); HP_PHASE(ACC_M, ACC_E, 2, STORES); HP_PHASE(ACC_M, ACC_E, 3, STORES); \
        HP_PHASE(ACC_M, ACC_E, 4, STORES); HP_PHASE(ACC_M, ACC_E, 5, STORES); HP_PHASE(ACC_M, ACC_E, 6, STORES); HP_PHASE(ACC_M, ACC_E, 7, STORES); \
        if (nk > 9) { HP_PHASE(ACC_M, ACC_E, -1, STORES); for (int kt_ = 10; kt_ < nk; ++kt_) HP_PHASE(ACC_M, ACC_E, -1, 0); } } while (0)
    // (a count that is too SMALL only waits longer -- e.g. the first phase of a tile behind a 9-step tile, whose slice 7 left stores in the
    //  queue; a count that is too large would let DMA pieces be read before they landed: every PREV_STORES above is exact or smaller)
    int j = 0;
    {
        HP_ZERO(acc0)
        for (int kt_ = 0; kt_ < nk; ++kt_) HP_PHASE(acc0, acc1, -1, 0);
        hp_coords(a, wg, st.em0, st.en0);
        j = 1;
    }
    for (;;) {                                             // two tiles per trip: the accumulator sets swap roles without any select / copy
        int t = j * G + wg, m0, n0;
        if (t >= ntiles) break;
        hp_coords(a, t, m0, n0);
        HP_TILE(acc1, acc0);
        st.em0 = m0; st.en0 = n0; ++j;
        t = j * G + wg;
        if (t >= ntiles) break;
        hp_coords(a, t, m0, n0);
        HP_TILE(acc0, acc1);
        st.em0 = m0; st.en0 = n0; ++j;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();             // group 0 pays back the stagger barrier
    // the last tile's epilogue, in the open.  j = number of tiles done: the last one accumulated in acc0 when j is odd
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(aux0), "+v"(aux1) :: "memory");
#define HP_LAST(ACC) do { \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) { \
            aux_fetch(u, st.em0, st.en0); \
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(aux0), "+v"(aux1) :: "memory"); \
            HP_SLICE(ACC, u, -1); } } while (0)
    if (j & 1) HP_LAST(acc0); else HP_LAST(acc1);
}

template <int EPIX>
int amdseg_launch_nt_hp(const GemmNTArgs& a_in, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_hp_kernel<EPIX>), hipFuncAttributeMaxDynamicSharedMemorySize, HP_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    GemmNTArgs a = a_in;
    a.tiles_m = a.M / HP_BM; a.tiles_n = a.N / HP_BN;
    const int ntiles = a.tiles_m * a.tiles_n;
    static int ncu = 0;
    if (!ncu) { int dev = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev); if (ncu <= 0) ncu = 256; }
    const int grid = ntiles < ncu ? ntiles : ncu;
    AMDSEG_LAUNCH_PROF(AMDSEG_PROF_GEMM_NT, 2.0 * a.M * a.N * a.K, (gemm_nt_hp_kernel<EPIX>), dim3(grid), dim3(512), HP_LDS, s, a, ntiles);
    return amdseg_launch_status();
}

template int amdseg_launch_nt_hp<EPI_BIAS_GELU>(const GemmNTArgs&, hipStream_t);
template int amdseg_launch_nt_hp<EPI_GELU_BWD>(const GemmNTArgs&, hipStream_t);
template int amdseg_launch_nt_hp<EPI_BIAS_GELU_TANH>(const GemmNTArgs&, hipStream_t);
template int amdseg_launch_nt_hp<EPI_GELU_BWD_TANH>(const GemmNTArgs&, hipStream_t);
