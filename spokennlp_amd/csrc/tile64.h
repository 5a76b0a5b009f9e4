// [64 rows][64 cols] bf16 tiles in LDS for gfx950: DMA staging (global_load_lds), conflict-free fragment reads and
// transposed (k-strided) fragment gathers with ds_read_b64_tr_b16.  Shared by attention.hip and longformer.hip.
#pragma once
#include "common.h"

#define HD 64          // tile width (head dim)
#define CH 64          // rows per streamed chunk

// XOR swizzle for [rows][64] bf16 tiles (128 B rows, 8 chunks of 16 B): chunk c of row r lives at chunk c ^ swz(r), swz(r) = 2 * ((r >> 1) & 3).
// One function that is conflict free for BOTH ways these tiles are read:
//  * ds_read_b128 fragments (lane = row l&15, chunk kk*4 + (l>>4)) are served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... =
//    the row pairs p = (r>>1)&7 in {0,1,6,7} at chunk c and {2,3,4,5} at chunk c ^ 1 (and the complement): the 16 lanes hit 16 distinct 16-B
//    slots iff {s(0),s(1),s(6),s(7)} u {s(2)^1,..,s(5)^1} and {s(2),..,s(5)} u {s(0)^1,s(1)^1,s(6)^1,s(7)^1} both cover 0..7;
//  * ds_read_b64_tr_b16 gathers touch, per 32-lane group, 8 CONSECUTIVE rows (aligned to 8) x 32 B: conflict free iff s(r) >> 1 takes four
//    different values over the four row pairs of the group.
// s = {0,2,4,6,0,2,4,6} satisfies both (exhaustive search over the 2-bit-permutation x low-bit family: 9216 solutions); the earlier
// (((r>>1)&3)<<1)|((r>>3)&1) failed the first condition (2-way conflicts on every b128 fragment), p ^ (p in {2,3,4,5}) the second
// (SQ_LDS_BANK_CONFLICT = 22-29 % of SQ_LDS_IDX_ACTIVE in the attention kernels, whose tiles are read both ways).
__device__ __forceinline__ int swz(int r) { return ((r >> 1) & 3) << 1; }

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
