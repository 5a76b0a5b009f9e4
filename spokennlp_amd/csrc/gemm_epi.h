// shared by gemm.hip and gemm_dp.hip: epilogue selectors, launch arguments, tile-order helpers of the NT GEMM kernels
#pragma once
#include "prof.h"
#include "common.h"

#define GROUP_M 8

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_ADD_RES = 3, EPI_GELU_BWD = 4,
       EPI_BIAS_GELU_TANH = 5, EPI_GELU_BWD_TANH = 6,       // kernel template values only: the two GELU epilogues with gelu_new
       EPI_BIAS_SPLIT = 7,                                  // C = bf16 hi of (A B^T + bias), C2 = bf16 lo = bf16(x - hi): the result as a split image
                                                            // ("parity" precision: the consumer is another split-bf16 product), 256-wide dp kernel only
       EPI_GELU_BWD_SPLIT = 8,                              // x = (A B^T) * gelu_erf'(R), R fp32: hi -> C and C + dup_off columns, lo -> C2 (the [hi | hi | lo]
                                                            // image the next split GEMM and the weight gradient read); 256-wide dp kernel only
       EPI_BIAS_GELU_SPLIT = 9,                             // C (fp32) = A B^T + bias (the pre-activation backward reads); C2 = bf16 image [hi | hi | lo] of gelu_erf(that)
       // (10 was the fused bias + dropout + residual epilogue of rounds 4-5: measured slower, profiles/r04_fused_drop_res.md, removed in round 6)
       EPI_BIAS_GELU_DG = 11,                               // AMDSEG_EPI_BIAS_GELU | AMDSEG_EPI_KEEP_DERIV: C = gelu(A B^T + bias), C2 = gelu'(A B^T + bias) (deep-pipeline kernel only)
       EPI_MUL_RES = 12,                                    // AMDSEG_EPI_GELU_BWD | AMDSEG_EPI_KEEP_DERIV: C = (A B^T) * R, R = the derivative kept by the forward
       EPI_BIAS_GELU_DG8 = 13, EPI_MUL_RES8 = 14,           // the same pair with the derivative as ONE BYTE per element (AMDSEG_EPI_DERIV_U8), 256-wide tile only
       EPI_BIAS_GELU_DG8_TANH = 15 };
// the kernels are instantiated on the extended value EPIX; EPI = what the epilogue does, ACT = which GELU (a compile-time constant:
// a run-time flag became one scalar branch PER ELEMENT in the epilogue)
#define EPI_BASE(X) ((X) == EPI_BIAS_GELU_TANH ? EPI_BIAS_GELU : (X) == EPI_GELU_BWD_TANH ? EPI_GELU_BWD : (X) == EPI_BIAS_GELU_DG8_TANH ? EPI_BIAS_GELU_DG8 : (X))
#define EPI_ACT(X) (((X) == EPI_BIAS_GELU_TANH || (X) == EPI_GELU_BWD_TANH || (X) == EPI_BIAS_GELU_DG8_TANH) ? 1 : 0)

struct GemmNTArgs {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; const bf16_t* R; bf16_t* C2;
    int lda, ldb, ldc, ldr, ldc2;
    int M, N, K;
    int tiles_m, tiles_n;
    int dup_off;                                            // EPI_GELU_BWD_SPLIT: second copy of the hi block, in columns from C
    // optional (deep-pipeline kernel only): rows >= zkend[row / zL] of A are known to be exact zeros (activation gradients of trailing
    // padding) unless *zguard != 0 -- a 256-row tile made of such rows skips its K loop and runs the epilogue on zero accumulators
    const int* zkend; const int* zguard; int zL;
};

// bijective XCD-aware remap: hardware places workgroup b on XCD b % 8; give each XCD a contiguous tile range
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}


// gelu'(x) lies in [-0.129, 1.129]: kept as q = round((g' + 0.135) * 200), one byte, absolute error <= 0.0025 -- smaller than the bf16 rounding
// of the same value wherever g' >= 0.64 (bf16: 2^-8 relative), larger near zero where the products it enters are small; what it buys is HBM bytes
// in two epilogues that are HBM time (profiles/r04_gemm_epilogue_split.md): 1 B instead of 2 B per element written by the FFN up-projection and
// read by the GELU backward GEMM
#define GELU_DQ_SCALE 200.0f
#define GELU_DQ_OFF 27.0f
#define GELU_DQ_STEP 0.005f
#define GELU_DQ_LO (-0.135f)
__device__ __forceinline__ uint32_t gelu_dq_pack4(const float* d) {
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(d[0] * GELU_DQ_SCALE + GELU_DQ_OFF, 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(d[1] * GELU_DQ_SCALE + GELU_DQ_OFF, 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(d[2] * GELU_DQ_SCALE + GELU_DQ_OFF, 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(d[3] * GELU_DQ_SCALE + GELU_DQ_OFF, 3, w);
    return w;
}
__device__ __forceinline__ uint32_t gelu_dq_pack4_scaled(const float* dq) {      // dq = gelu' * GELU_DQ_SCALE + GELU_DQ_OFF already (gelu_both4q)
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(dq[0], 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(dq[1], 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(dq[2], 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(dq[3], 3, w);
    return w;
}
__device__ __forceinline__ void gelu_dq_mul4(float* v, uint32_t w) {
    v[0] *= (float)(w & 0xffu) * GELU_DQ_STEP + GELU_DQ_LO;
    v[1] *= (float)((w >> 8) & 0xffu) * GELU_DQ_STEP + GELU_DQ_LO;
    v[2] *= (float)((w >> 16) & 0xffu) * GELU_DQ_STEP + GELU_DQ_LO;
    v[3] *= (float)(w >> 24) * GELU_DQ_STEP + GELU_DQ_LO;
}

// deep-pipeline 256 x 256 kernel (gemm_dp.hip)
template <int EPIX, typename OutT> int amdseg_launch_nt_dp(const GemmNTArgs& a, hipStream_t s);

// grouped TN GEMM (weight gradients): launch arguments shared by gemm.hip (128 x 128 kernel) and gemm_dp.hip (256 x 128 kernel)
struct TNProblem { const bf16_t* A; const bf16_t* B; float* C; int N, Kp, lda, ldb, ldc, tile_begin, tiles_k;
                   float* colsum_part; };   // optional [tiles_k][N] scratch: per-K'-tile partial column sums of A (bias gradient), dp kernel only
struct GemmTNArgs { TNProblem p[AMDSEG_MAX_GROUP]; int nprob, M, accumulate, total_tiles;
                    // optional (deep-pipeline kernel only): runs[r] = {first, end} (r < counts[1]) = the runs of 64-token K tiles whose A
                    // rows are not all exact zeros, counts[0] tiles in all; ignored (every tile walked) when *zguard != 0
                    const int* runs; const int* counts; const int* zguard; };
int amdseg_launch_tn_dp(const GemmTNArgs& a128, hipStream_t s);
