// [64 rows][64 cols] bf16 tiles in LDS for gfx950: DMA staging (global_load_lds), conflict-free fragment reads and
// transposed (k-strided) fragment gathers with ds_read_b64_tr_b16.  Shared by attention.hip and longformer.hip.
#pragma once
#include "common.h"

#define HD 64          // tile width (head dim)
#define CH 64          // rows per streamed chunk

// XOR swizzle for [rows][64] bf16 tiles (128 B rows, 8 chunks of 16 B): chunk c of row r lives at chunk c ^ swz(r).
// ds_read_b128 serves a wave in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... = 8 rows at chunk c and the other 8
// rows of the fragment at chunk c ^ 1; f(p) = p ^ (p in {2,3,4,5}) over the row pair p = (r >> 1) & 7 makes the 16 lanes of a
// group hit 16 distinct 16-B slots (the earlier (((r>>1)&3)<<1)|((r>>3)&1) had 2-way conflicts on EVERY b128 fragment read --
// SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE in the GEMM prototype).
__device__ __forceinline__ int swz(int r) { const int p = (r >> 1) & 7; return p ^ (((p + 2) >> 2) & 1); }

__device__ __forceinline__ void at_glds16(const void* g, void* lds_wave_base) {
    amdseg_glds16(g, lds_wave_base);
}
// stage a [64][64] bf16 tile; `base` points at element (row 0, col 0), rows are row_stride elements apart
template <int NW>
__device__ __forceinline__ void at_stage(const bf16_t* base, int row_stride, char* tile, int w, int l) {
#pragma unroll
    for (int q = 0; q < 8 / NW; ++q) {
        const int R0 = (w * (8 / NW) + q) * 8;
        const int r = R0 + (l >> 3), s = l & 7;
        const int c = s ^ swz(r);
        at_glds16(base + (size_t)r * row_stride + c * 8, tile + R0 * 128);
    }
}
// 8 consecutive k-elements (chunk c) of row r
__device__ __forceinline__ bf16x8 at_frag(const char* tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(tile + r * 128 + ((c ^ swz(r)) << 4));
}
// transposed gather: lane (i16 = l&15, g = l>>4) receives tile[r0a + j][col0 + i16] (j<4) and tile[r0b + j-4][..] (j>=4)
__device__ __forceinline__ bf16x8 at_frag_tr(const char* tile, int r0a, int r0b, int col0, int l) {
    const int i16 = l & 15;
    const int c = (col0 >> 3) + ((i16 & 3) >> 1);
    bf16x8 f;
    {
        const int row = r0a + (i16 >> 2);
        bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4, tile + row * 128 + ((c ^ swz(row)) << 4) + (i16 & 1) * 8));
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    }
    {
        const int row = r0b + (i16 >> 2);
        bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4, tile + row * 128 + ((c ^ swz(row)) << 4) + (i16 & 1) * 8));
        f[4] = v[0]; f[5] = v[1]; f[6] = v[2]; f[7] = v[3];
    }
    return f;
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    union { uint32_t u[4]; bf16x8 v; } r;
    r.u[0] = pack2bf(a[0], a[1]); r.u[1] = pack2bf(a[2], a[3]);
    r.u[2] = pack2bf(b[0], b[1]); r.u[3] = pack2bf(b[2], b[3]);
    return r.v;
}
