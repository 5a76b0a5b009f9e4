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

__device__ __forceinline__ int dp_swz(int r) { const int p = (r >> 1) & 7; return p ^ (((p + 2) >> 2) & 1); }
__device__ __forceinline__ bf16x8 dp_frag(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ dp_swz(r)) << 4));
}
__device__ __forceinline__ void dp_glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(void, lds_wave_base), 16, 0, 0);
}

template <int EPI, typename OutT>
__global__ __launch_bounds__(512, 1) void gemm_nt_dp_kernel(GemmNTArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3, wq = w & 3;
    const int g = l >> 4, i16 = l & 15;
    const int nwg = a.tiles_m * a.tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    const int gsz_full = GROUP_M * a.tiles_n;
    const int grp = t / gsz_full, first_m = grp * GROUP_M;
    const int gmn = min(a.tiles_m - first_m, GROUP_M);
    const int rem = t - grp * gsz_full;
    const int tm = first_m + rem % gmn, tn = rem / gmn;
    const int m0 = tm * DP_BM, n0 = tn * DP_BN;
#define DP_TILE_A(s, i) (smem + (s) * DP_STAGE + (i) * 8192)
#define DP_TILE_B(s, i) (smem + (s) * DP_STAGE + (4 + (i)) * 8192)
    const bf16_t* pA = a.A + (size_t)(m0 + wr * 128) * a.lda;                 // this group's A rows
    const bf16_t* pB = a.B + (size_t)(n0 + wr * 128) * a.ldb;                 // this group's DMA duty on B: tile images 2wr, 2wr+1
    int offA[4], offB[4];                                                     // lane offsets, constant over K (saddr + voffset loads)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wq * 2 + q) * 8 + (l >> 3), c = (l & 7) ^ dp_swz(r);
            offA[i * 2 + q] = (i * 64 + r) * a.lda + c * 8;
            offB[i * 2 + q] = (i * 64 + r) * a.ldb + c * 8;
        }
#define DP_DMA_A(s, i, kt) _Pragma("unroll") for (int q = 0; q < 2; ++q) \
        dp_glds16(pA + (kt) * 64 + offA[(i) * 2 + q], DP_TILE_A(s, wr * 2 + (i)) + (wq * 2 + q) * 1024);
#define DP_DMA_B(s, kt) _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int q = 0; q < 2; ++q) \
        dp_glds16(pB + (kt) * 64 + offB[i * 2 + q], DP_TILE_B(s, wr * 2 + i) + (wq * 2 + q) * 1024);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = a.K / 64;
    DP_DMA_A(0, 0, 0) DP_DMA_A(0, 1, 0) DP_DMA_B(0, 0)
    if (nk > 1) { DP_DMA_A(1, 0, 1) DP_DMA_A(1, 1, 1) DP_DMA_B(1, 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();             // stagger: group 1 runs one barrier behind group 0
    bf16x8 fa[4][2], fb[4][2];
#define DP_LOAD_A(s, h) _Pragma("unroll") for (int f = 0; f < 4; ++f) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        fa[f][kk] = dp_frag(DP_TILE_A(s, wr * 2 + (h)), f * 16 + i16, kk * 4 + g);
#define DP_LOAD_B(s) _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
        fb[e][kk] = dp_frag(DP_TILE_B(s, wc), e * 16 + i16, kk * 4 + g);
#define DP_MFMA(ah) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int f = 0; f < 4; ++f) \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) \
        acc[(ah) * 4 + f][e] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[e][kk], fa[f][kk], acc[(ah) * 4 + f][e], 0, 0, 0);
#define DP_MID() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(1); } while (0)
#define DP_END() do { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    for (int kt = 0; kt < nk; ++kt) {
        const int s = kt & 1;
        // ---- phase 1: rows 0-63 of the wave tile.  DMA: the rows-64..127 image of K tile kt+1 (other stage; last read in phase 2
        //      of K tile kt-1, retired before that phase's barrier)
        DP_LOAD_B(s) DP_LOAD_A(s, 0)
        if (kt >= 1 && kt + 1 < nk) { DP_DMA_A(s ^ 1, 1, kt + 1) }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt >= 1 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DP_MID();
        DP_MFMA(0)
        DP_END();
        // ---- phase 2: rows 64-127.  DMA: rows-0..63 image + this group's two B images of K tile kt+2 into THIS stage (their last
        //      readers -- this group's phase 1 and the other group's phase 1, one slot later -- have retired their reads)
        DP_LOAD_A(s, 1)
        if (kt + 2 < nk) { DP_DMA_A(s, 0, kt + 2) DP_DMA_B(s, kt + 2) }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DP_MID();
        DP_MFMA(1)
        DP_END();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();             // group 0 pays back the stagger barrier: every LDS read is retired now

    // ---- epilogue.  Lane owns row m = mf*16 + i16 and columns nf*16 + g*4 .. +4 of the wave tile.  bf16 results go through a
    // wave-private 16-KiB LDS image (2 x [64 rows][128 B], swizzled) so every global store instruction writes 8 full 128-B lines.
    float4 bv[4];
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) bv[nf] = *reinterpret_cast<const float4*>(a.bias + n0 + wc * 64 + nf * 16 + g * 4);
    }
    constexpr bool STAGED = sizeof(OutT) == 2;
    char* stg = smem + w * 16384;
#define DP_STG_OFF(r, c16) (((r) >> 6) * 8192 + ((r) & 63) * 128 + ((((c16) ^ (((r) & 63) ^ (((r) & 63) >> 3))) & 7) << 4))
    const int col0 = n0 + wc * 64;
#pragma unroll
    for (int pass = 0; pass < (EPI == EPI_BIAS_GELU ? 2 : 1); ++pass) {
        // pass 0 of BIAS_GELU writes the pre-activation (C2), pass 1 the activation; other epilogues have one pass
        if (EPI == EPI_BIAS_GELU && pass == 0 && !a.C2) continue;
#pragma unroll
        for (int mf = 0; mf < 8; ++mf) {
            const size_t gm = (size_t)(m0 + wr * 128 + mf * 16 + i16);
            uint2 rr[4];
            if (EPI == EPI_ADD_RES || EPI == EPI_GELU_BWD) {
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) rr[nf] = *reinterpret_cast<const uint2*>(a.R + gm * a.ldr + col0 + nf * 16 + g * 4);
            }
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                float v[4] = {acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]};
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) { v[0] += bv[nf].x; v[1] += bv[nf].y; v[2] += bv[nf].z; v[3] += bv[nf].w; }
                if (EPI == EPI_BIAS_GELU && pass == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
                } else if (EPI == EPI_ADD_RES || EPI == EPI_GELU_BWD) {
                    const float r0 = __uint_as_float(rr[nf].x << 16), r1 = __uint_as_float(rr[nf].x & 0xffff0000u);
                    const float r2 = __uint_as_float(rr[nf].y << 16), r3 = __uint_as_float(rr[nf].y & 0xffff0000u);
                    if (EPI == EPI_ADD_RES) { v[0] += r0; v[1] += r1; v[2] += r2; v[3] += r3; }
                    else { v[0] *= gelu_grad_fast(r0); v[1] *= gelu_grad_fast(r1); v[2] *= gelu_grad_fast(r2); v[3] *= gelu_grad_fast(r3); }
                }
                if (STAGED) {
                    uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>(stg + DP_STG_OFF(mf * 16 + i16, nf * 2 + (g >> 1)) + (g & 1) * 8) = pk;
                } else {
                    float* dst = reinterpret_cast<float*>(a.C) + gm * a.ldc + col0 + nf * 16 + g * 4;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        if (STAGED) {
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int rr0 = l >> 3, cc = l & 7;
            bf16_t* obase = ((EPI == EPI_BIAS_GELU && pass == 0) ? a.C2 : reinterpret_cast<bf16_t*>(a.C));
            const int old = (EPI == EPI_BIAS_GELU && pass == 0) ? a.ldc2 : a.ldc;
            obase += (size_t)(m0 + wr * 128) * old + col0 + cc * 8;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int r = p * 8 + rr0;
                const uint4 val = *reinterpret_cast<const uint4*>(stg + DP_STG_OFF(r, cc));
                *reinterpret_cast<uint4*>(obase + (size_t)r * old) = val;
            }
            __builtin_amdgcn_wave_barrier();              // the image is rewritten by the next pass
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

template <int EPI, typename OutT>
int amdseg_launch_nt_dp(const GemmNTArgs& a_in, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_dp_kernel<EPI, OutT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    GemmNTArgs a = a_in;
    a.tiles_m = a.M / DP_BM; a.tiles_n = a.N / DP_BN;
    hipLaunchKernelGGL((gemm_nt_dp_kernel<EPI, OutT>), dim3(a.tiles_m * a.tiles_n), dim3(512), DP_LDS, s, a);
    return amdseg_launch_status();
}

#define DP_INST(E, T) template int amdseg_launch_nt_dp<E, T>(const GemmNTArgs&, hipStream_t);
DP_INST(EPI_NONE, bf16_t) DP_INST(EPI_NONE, float) DP_INST(EPI_BIAS, bf16_t) DP_INST(EPI_BIAS, float) DP_INST(EPI_BIAS_GELU, bf16_t)
DP_INST(EPI_ADD_RES, bf16_t) DP_INST(EPI_ADD_RES, float) DP_INST(EPI_GELU_BWD, bf16_t)
