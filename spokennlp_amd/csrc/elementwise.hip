// HBM-bound row kernels of the BERT encoder for gfx950: embedding+LayerNorm, bias/dropout/residual+LayerNorm,
// their backward passes, column sums (bias gradients), dropout, casts and the weight cast+transpose.
//
// Reference arithmetic: [hf] models/bert/modeling_bert.py:53-108 (BertEmbeddings), :282-293 (BertSelfOutput),
// :340-351 (BertOutput) -- LayerNorm eps 1e-12, dropout after the dense, residual add before the norm; in-tree copy
// mmvts/src/models/cross_encoder/bert_model.py:166-211,364-375,442-453.
//
// One wave (64 lanes) owns one token row; every global access is a 16 B/lane vector (8 bf16 or 2x float4), the row
// lives in registers between the statistics passes (two-pass mean / variance like torch.nn.LayerNorm), row
// reductions are wave shuffles, column reductions go register -> LDS -> per-block partials -> a deterministic
// second-stage reduce (no atomics on the hot path).
#include "common.h"
#include "amdseg_internal.h"
#include "prof.h"
#include "keepmask.h"

#define MAXCH 4                 // up to 4 chunks of 8 elements per lane -> H <= 2048
#define ROWS_PER_BLOCK 4        // one wave per row, 4 waves per block

template <typename T, int NCH>
__device__ __forceinline__ void row_load(const T* row, int nch, int l, float (&v)[NCH][8]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = l + c * 64;
        if (ch < nch) ld8<T>(row + ch * 8, v[c]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        }
    }
}
template <typename T, int NCH>
__device__ __forceinline__ void row_store(T* row, int nch, int l, const float (&v)[NCH][8]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = l + c * 64;
        if (ch < nch) st8<T>(row + ch * 8, v[c]);
    }
}
// the row as a split-bf16 image [hi | hi | lo] (row stride 3H): what amdseg_split3(order 0) makes of it ("parity" precision: the next GEMM's A operand)
template <int NCH>
__device__ __forceinline__ void row_store_image(bf16_t* img_row, int H, int nch, int l, const float (&v)[NCH][8]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = l + c * 64;
        if (ch < nch) {
            uint4 hi, lo;
            hi.x = pack2bf(v[c][0], v[c][1]); hi.y = pack2bf(v[c][2], v[c][3]); hi.z = pack2bf(v[c][4], v[c][5]); hi.w = pack2bf(v[c][6], v[c][7]);
            lo.x = pack2bf(v[c][0] - __uint_as_float(hi.x << 16), v[c][1] - __uint_as_float(hi.x & 0xffff0000u));
            lo.y = pack2bf(v[c][2] - __uint_as_float(hi.y << 16), v[c][3] - __uint_as_float(hi.y & 0xffff0000u));
            lo.z = pack2bf(v[c][4] - __uint_as_float(hi.z << 16), v[c][5] - __uint_as_float(hi.z & 0xffff0000u));
            lo.w = pack2bf(v[c][6] - __uint_as_float(hi.w << 16), v[c][7] - __uint_as_float(hi.w & 0xffff0000u));
            *reinterpret_cast<uint4*>(img_row + ch * 8) = hi; *reinterpret_cast<uint4*>(img_row + H + ch * 8) = hi;
            *reinterpret_cast<uint4*>(img_row + 2 * H + ch * 8) = lo;
        }
    }
}
template <int NCH>
__device__ __forceinline__ void row_stats(const float (&v)[NCH][8], int nch, int l, int H, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (l + c * 64 < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[c][e];
        }
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (l + c * 64 < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
        }
    rstd = rsqrtf(wave_sum(q) / (float)H + eps);
}

// ------------------------------------------------------------------------------------------------ embeddings + LN
template <typename T, int NCH>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids,
                                                           const float* word, const float* pos, const float* type,
                                                           const float* gamma, const float* beta, T* z, T* out, float* mean,
                                                           float* rstd, int M, int L, int H, int vocab, int type_vocab,
                                                           int npos, float eps, uint32_t thresh, float inv_keep, uint64_t seed,
                                                           int pre_ln_dropout) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int m = blockIdx.x * ROWS_PER_BLOCK + w;
    if (m >= M) return;
    const int nch = H >> 3;
    int64_t id = ids[m]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int64_t tt = type_ids ? type_ids[m] : 0; tt = tt < 0 ? 0 : (tt >= type_vocab ? type_vocab - 1 : tt);
    int64_t pp = pos_ids ? pos_ids[m] : (int64_t)(m % L); pp = pp < 0 ? 0 : (pp >= npos ? npos - 1 : pp);
    float v[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = l + c * 64;
        if (ch < nch) {
            float a[8], b[8], d[8];
            ld8<float>(word + (size_t)id * H + ch * 8, a);
            ld8<float>(pos + (size_t)pp * H + ch * 8, b);
            ld8<float>(type + (size_t)tt * H + ch * 8, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = (a[e] + d[e]) + b[e];     // (word + type) + position, as the reference
            // BigBird: LayerNorm(dropout(sum)) ([hf] models/big_bird/modeling_big_bird.py BigBirdEmbeddings.forward); z is then the
            // dropped sum (the LayerNorm input), which is what ln_bwd needs
            if (pre_ln_dropout && thresh) drop8_apply(seed, (uint64_t)m * nch + ch, thresh, inv_keep, v[c]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
        }
    }
    if (z) row_store<T, NCH>(z + (size_t)m * H, nch, l, v);
    float mu, rs;
    row_stats(v, nch, l, H, eps, mu, rs);
    if (l == 0) { if (mean) mean[m] = mu; if (rstd) rstd[m] = rs; }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = l + c * 64;
        if (ch < nch) {
            float gg[8], bb[8];
            ld8<float>(gamma + ch * 8, gg); ld8<float>(beta + ch * 8, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = (v[c][e] - mu) * rs * gg[e] + bb[e];
            if (thresh && !pre_ln_dropout) drop8_apply(seed, (uint64_t)m * nch + ch, thresh, inv_keep, v[c]);
        }
    }
    row_store<T, NCH>(out + (size_t)m * H, nch, l, v);
}

// scatter the embedding-sum gradient dz[M,H] into the three tables
#define EMB_ROWS 16
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* dz, const int64_t* ids, const int64_t* type_ids,
                                                        const int64_t* pos_ids, float* dword, float* dpos, float* dtype,
                                                        int M, int L, int H, int vocab, int type_vocab, int npos, int pad_id) {
    // block = EMB_ROWS consecutive rows x one 256-column slab; type-embedding grads are reduced in LDS first (2-16 hot rows).
    // (64 rows per block = 64 dependent load -> atomic trips on 3 workgroups per CU: 99 us at M = 16384; 16 rows: see DESIGN section 8)
    // type_vocab < 0: |type_vocab| rows, and row 0 of dtype ALREADY holds the column sum of dz over all rows (the caller let amdseg_ln_bwd write
    // it as its dbias): a row of type t != 0 then moves its g from row 0 to row t, a row of type 0 adds nothing.  The usual inputs (one
    // segment: every type id 0) then issue no type atomics at all -- they were 256-512 same-address atomics per column (385 us at M = 32768)
    __shared__ float tacc[4][256];
    const bool t0sum = type_vocab < 0;
    if (t0sum) type_vocab = -type_vocab;
    const int c = blockIdx.y * 256 + threadIdx.x;
    const int r0 = blockIdx.x * EMB_ROWS;
    for (int t = 0; t < 4; ++t) tacc[t][threadIdx.x] = 0.f;
    if (c >= H) return;
    for (int r = r0; r < r0 + EMB_ROWS && r < M; ++r) {
        const float g = Act<T>::ld(dz + (size_t)r * H + c);
        if (g == 0.f) continue;                             // adds nothing anywhere.  The rows of trailing padding are exact zeros, and
                                                            // with RoBERTa-style position ids they ALL point at one position row (padding_idx)
        int64_t id = ids[r];
        if (id >= 0 && id < vocab && id != pad_id) unsafeAtomicAdd(dword + (size_t)id * H + c, g);
        if (pos_ids) {                                      // (default positions r % L: embed_bwd_pos_kernel, no atomics)
            const int64_t pp = pos_ids[r];
            if (pp >= 0 && pp < npos) unsafeAtomicAdd(dpos + (size_t)pp * H + c, g);
        }
        int64_t tt = type_ids ? type_ids[r] : 0;
        if (t0sum) {
            if (tt == 0) continue;
            tacc[0][threadIdx.x] -= g;                      // (also for ids outside the table: the forward clamps, the reference would raise)
        }
        if (tt >= 0 && tt < 4 && tt < type_vocab) tacc[tt][threadIdx.x] += g;
        else if (tt >= 4 && tt < type_vocab) unsafeAtomicAdd(dtype + (size_t)tt * H + c, g);
    }
    for (int t = 0; t < 4 && t < type_vocab; ++t)
        if (tacc[t][threadIdx.x] != 0.f) unsafeAtomicAdd(dtype + (size_t)t * H + c, tacc[t][threadIdx.x]);
}

// default position ids (row r of the batch sits at position r % L): dpos[p] += sum over the B sequences of dz[b*L + p] -- every (p, column) has
// one owner, so plain adds in batch order (deterministic) instead of B atomics per element (half of the scatter's atomic traffic)
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const T* __restrict__ dz, float* __restrict__ dpos, int M, int L, int H, int npos) {
    const int c = blockIdx.y * 256 + threadIdx.x, p = blockIdx.x;
    if (c >= H || p >= npos) return;
    float acc = 0.f;
    for (int r = p; r < M; r += L) acc += Act<T>::ld(dz + (size_t)r * H + c);
    dpos[(size_t)p * H + c] += acc;
}

// Order-independent form of the table scatter (deterministic mode): `order` is a STABLE argsort of keys, so the rows that share a key are
// consecutive in it and in batch order.  The block at the head of a run sums its run front to back and adds the total to the table row with
// a plain store -- one writer per (row, column), a fixed summation order: the result is bit-reproducible, which the atomics above are not.
// A run of n rows costs n dependent-free loads on one block per 256 columns (a token that fills 5 % of a 16 K-token batch: ~0.1 ms).
template <typename T>
__global__ __launch_bounds__(256) void scatter_rows_sorted_kernel(const T* __restrict__ dz, const int64_t* __restrict__ keys,
                                                                  const int64_t* __restrict__ order, float* table, int M, int H, int nrows,
                                                                  int64_t skip_key) {
    const int j = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    const int64_t key = keys[order[j]];
    if (key < 0 || key >= nrows || key == skip_key) return;
    if (j > 0 && keys[order[j - 1]] == key) return;          // not the head of its run
    if (c >= H) return;
    float acc = 0.f;
    int e = j;
    for (; e + 3 < M; e += 4) {                               // four rows in flight; added in run order
        const int64_t r0 = order[e], r1 = order[e + 1], r2 = order[e + 2], r3 = order[e + 3];
        if (keys[r3] != key) break;                           // (sorted: the four are then all of this run)
        const float g0 = Act<T>::ld(dz + (size_t)r0 * H + c), g1 = Act<T>::ld(dz + (size_t)r1 * H + c);
        const float g2 = Act<T>::ld(dz + (size_t)r2 * H + c), g3 = Act<T>::ld(dz + (size_t)r3 * H + c);
        acc += g0; acc += g1; acc += g2; acc += g3;
    }
    for (; e < M; ++e) {
        const int64_t r = order[e];
        if (keys[r] != key) break;
        acc += Act<T>::ld(dz + (size_t)r * H + c);
    }
    table[(size_t)key * H + c] += acc;
}

// ------------------------------------------------------------------------------------------------ dropout + residual + LN
// y (dense output incl. bias) is overwritten by z = resid + dropout(y) (kept for backward); out = LN(z)
template <typename T, int NCH>
__device__ __forceinline__ void add_ln_fwd_body(int block, T* y_z, const T* resid, const float* gamma, const float* beta, T* out,
                                                float* mean, float* rstd, int M, int H, float eps, uint32_t thresh,
                                                float inv_keep, uint64_t seed, bf16_t* img, uint8_t* keepbits, bool keep_z) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int m = block * ROWS_PER_BLOCK + w;
    if (m >= M) return;
    const int nch = H >> 3;
    float v[NCH][8], x[NCH][8];
    row_load<T, NCH>(y_z + (size_t)m * H, nch, l, v);
    if (resid) {                                           // (resid == NULL: y_z already IS z -- the GEMM's epilogue added the residual: LayerNorm only)
    row_load<T, NCH>(resid + (size_t)m * H, nch, l, x);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = l + c * 64;
        if (ch < nch) {
            if (thresh) {
                if (keepbits) {                            // the decisions of this chunk, kept for ln_bwd (1 byte per 8 elements)
                    const uint32_t bits = drop8_bits(seed, (uint64_t)m * nch + ch, thresh);
                    keepbits[(size_t)m * nch + ch] = (uint8_t)bits;
                    drop8_apply_bits(bits, inv_keep, v[c]);
                } else drop8_apply(seed, (uint64_t)m * nch + ch, thresh, inv_keep, v[c]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = x[c][e] + v[c][e];
        }
    }
    if (keep_z) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = l + c * 64;
        if (ch < nch) {
            if (sizeof(T) == 2) {
                typedef unsigned ew_u4 __attribute__((ext_vector_type(4)));
                ew_u4 q = {pack2bf(v[c][0], v[c][1]), pack2bf(v[c][2], v[c][3]), pack2bf(v[c][4], v[c][5]), pack2bf(v[c][6], v[c][7])};
                __builtin_nontemporal_store(q, reinterpret_cast<ew_u4*>(y_z + (size_t)m * H + ch * 8));
            } else st8<T>(y_z + (size_t)m * H + ch * 8, v[c]);
        }
    }
    }
    }
    float mu, rs;
    row_stats(v, nch, l, H, eps, mu, rs);
    if (l == 0) { if (mean) mean[m] = mu; if (rstd) rstd[m] = rs; }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = l + c * 64;
        if (ch < nch) {
            float gg[8], bb[8];
            ld8<float>(gamma + ch * 8, gg); ld8<float>(beta + ch * 8, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[c][e] = (v[c][e] - mu) * rs * gg[e] + bb[e];
        }
    }
    row_store<T, NCH>(out + (size_t)m * H, nch, l, v);
    if (img) row_store_image<NCH>(img + (size_t)m * 3 * H, H, nch, l, v);
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(T* y_z, const T* resid, const float* gamma, const float* beta, T* out,
                                                         float* mean, float* rstd, int M, int H, float eps, uint32_t thresh,
                                                         float inv_keep, uint64_t seed, bf16_t* img, uint8_t* keepbits, bool keep_z) {
    add_ln_fwd_body<T, NCH>((int)blockIdx.x, y_z, resid, gamma, beta, out, mean, rstd, M, H, eps, thresh, inv_keep, seed, img, keepbits, keep_z);
}

// The same rows AND the attention-dropout keep masks of the NEXT layer in one grid (round 6).  The row kernel is HBM-bound (4 passes over [M, H],
// 20.7 us at 16,384 x 768), the generator VALU-bound (18.5 us for 32 x 12 x 512 x 512 decisions): launched back to back they take the sum, as
// workgroups of ONE launch the generator's waves issue under the rows' memory latency.  The n_km generator blocks are spread evenly through the n_ln
// row blocks (block b is a generator block iff k(b + 1) > k(b), k(b) ~ b n_km / (n_ln + n_km), see the kernel), so every CU holds both kinds at any time;
// each kind computes exactly what it computes alone (add_ln_fwd_body / km_block_body on its own block index): the bits do not depend on the fusion.
struct AddLnArgs {
    void* y_z; const void* resid; const float* gamma; const float* beta; void* out; float* mean; float* rstd;
    int M, H; float eps; uint32_t thresh; float inv_keep; uint64_t seed; uint8_t* keepbits; int keep_z;
};
template <typename T, int NCH>
__global__ __launch_bounds__(256) void add_ln_fwd_km_kernel(AddLnArgs a, KeepMaskArgs k, unsigned ratio) {
    // generator blocks in front of block b: k(b) = (b * ratio) >> 32 with ratio = ceil(2^32 n_km / n) -- ONE scalar multiply-high (a 64-bit division
    // here cost every row wave ~200 vector instructions).  k is monotone with steps of 0 or 1 (ratio < 2^32), k(0) = 0 and k(n) = n_km exactly (the
    // ceiling's excess times n stays below 2^32), so the generator indices 0 .. n_km - 1 and the row-block indices 0 .. n_ln - 1 are each hit once
    const unsigned b = blockIdx.x;
    const unsigned k0 = __umulhi(b, ratio), k1 = __umulhi(b + 1, ratio);
    if (k1 > k0) km_block_body(k, (int)k0);
    else add_ln_fwd_body<T, NCH>((int)(b - k0), (T*)a.y_z, (const T*)a.resid, a.gamma, a.beta, (T*)a.out, a.mean, a.rstd, a.M, a.H, a.eps, a.thresh,
                                 a.inv_keep, a.seed, nullptr, a.keepbits, a.keep_z != 0);
}

// LN backward.  dz = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.  Also emits
//   dbranch = dz * keepmask / (1-p)   (gradient of the dense output; == dz when p == 0 -> pass dbranch = nullptr)
//   per-block column partials of dgamma (sum dy*xhat), dbeta (sum dy) and dbias (sum dbranch)
#define LNB_ROWS 16     // rows per block (4 per wave)
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* dy, const T* z, const float* mean, const float* rstd,
                                                     const float* gamma, T* dz, T* dbranch, float* partials, int M, int H,
                                                     uint32_t thresh, float inv_keep, uint64_t seed,
                                                     const int* zkend, const int* zguard, int zL, bf16_t* img,
                                                     const uint8_t* keepbits) {
    extern __shared__ float red[];         // [3][4 waves][H]
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nch = H >> 3;
    // rows of trailing padding whose dy is a known exact zero (amdseg_bert_cfg.pad_guard): dz = dbranch = 0, nothing added to the column sums
    // (decided per BLOCK of 16 consecutive rows, zL % 16 == 0: a per-row branch inside the row loop kept the compiler from hoisting the next
    //  row's loads over the current row's arithmetic: 30.4 -> 32.9 us)
    if (zkend != nullptr && *zguard == 0) {
        const int m0 = blockIdx.x * LNB_ROWS, zb = m0 / zL;
        if (m0 - zb * zL >= zkend[zb]) {
            float zero[NCH][8];
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) zero[c][e] = 0.f;
            for (int rr = 0; rr < LNB_ROWS / 4; ++rr) {
                const int m = m0 + rr * 4 + w;
                if (m >= M) break;
                row_store<T, NCH>(dz + (size_t)m * H, nch, l, zero);
                if (dbranch) row_store<T, NCH>(dbranch + (size_t)m * H, nch, l, zero);
                if (img) row_store_image<NCH>(img + (size_t)m * 3 * H, H, nch, l, zero);
            }
            if (partials)
                for (int i = threadIdx.x; i < 3 * H; i += 256) {
                    const int k = i / H, c = i - k * H;
                    partials[((size_t)k * gridDim.x + blockIdx.x) * H + c] = 0.f;
                }
            return;
        }
    }
    float ag[NCH][8], ab[NCH][8], abias[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[c][e] = 0.f; ab[c][e] = 0.f; abias[c][e] = 0.f; }
    }
    // gamma lives in the (still unused) reduction area during the row loop: 16 VGPRs less keeps the kernel at 4 waves per SIMD
    for (int i = threadIdx.x; i < H; i += 256) red[i] = gamma[i];
    __syncthreads();
    for (int rr = 0; rr < LNB_ROWS / 4; ++rr) {
        const int m = blockIdx.x * LNB_ROWS + rr * 4 + w;
        if (m >= M) break;
        float g[NCH][8], x[NCH][8];
        row_load<T, NCH>(dy + (size_t)m * H, nch, l, g);
        row_load<T, NCH>(z + (size_t)m * H, nch, l, x);
        uint32_t kb[NCH];                                  // the forward's dropout decisions of this lane's chunks (loaded with the row)
#pragma unroll
        for (int c = 0; c < NCH; ++c) kb[c] = (keepbits && thresh && l + c * 64 < nch) ? keepbits[(size_t)m * nch + l + c * 64] : 0u;
        const float mu = mean[m], rs = rstd[m];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (l + c * 64 < nch) {
                float gg[8];
                ld8<float>(red + (l + c * 64) * 8, gg);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (x[c][e] - mu) * rs;
                    x[c][e] = xh;
                    { ab[c][e] += g[c][e]; ag[c][e] += g[c][e] * xh; }
                    g[c][e] *= gg[e];
                    s1 += g[c][e];
                    s2 += g[c][e] * xh;
                }
            }
        s1 = wave_sum_dpp(s1) / (float)H;
        s2 = wave_sum_dpp(s2) / (float)H;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (l + c * 64 < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) g[c][e] = rs * (g[c][e] - s1 - x[c][e] * s2);
            }
        row_store<T, NCH>(dz + (size_t)m * H, nch, l, g);
        if (img && !dbranch) row_store_image<NCH>(img + (size_t)m * 3 * H, H, nch, l, g);      // no dropout: the dense layer's gradient IS dz
        if (dbranch || partials) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int ch = l + c * 64;
                if (ch < nch) {
                    if (thresh) {
                        if (keepbits) drop8_apply_bits(kb[c], inv_keep, g[c]);
                        else drop8_apply(seed, (uint64_t)m * nch + ch, thresh, inv_keep, g[c]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) abias[c][e] += g[c][e];
                }
            }
            if (dbranch) row_store<T, NCH>(dbranch + (size_t)m * H, nch, l, g);
            if (img && dbranch) row_store_image<NCH>(img + (size_t)m * 3 * H, H, nch, l, g);
        }
    }
    if (!partials) return;
    __syncthreads();                       // every wave is done reading gamma from `red`
    // cross-wave reduce through LDS, then one partial row per block
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int ch = l + c * 64;
        if (ch < nch) {
            // 16-B LDS stores (H % 8 == 0): the per-element form compiled to 48 ds_write_b32 at a 32-B lane stride = 8-way bank conflicts
            st8<float>(red + (0 * 4 + w) * H + ch * 8, ag[c]);
            st8<float>(red + (1 * 4 + w) * H + ch * 8, ab[c]);
            st8<float>(red + (2 * 4 + w) * H + ch * 8, abias[c]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * H; i += 256) {
        const int k = i / H, c = i - k * H;
        const float s = red[(k * 4 + 0) * H + c] + red[(k * 4 + 1) * H + c] + red[(k * 4 + 2) * H + c] + red[(k * 4 + 3) * H + c];
        partials[((size_t)k * gridDim.x + blockIdx.x) * H + c] = s;
    }
}

// ---- ln_bwd for H = 768 in bf16 (bert-base / longformer-base / bigbird-base / PoNet-base: the shape every measured step runs) ----------------
// Round 4: SQ counters of the generic kernel above (profiles/r04_ln_bwd.md): 489 vector instructions per token row and wave, the four waves
// of a SIMD ISSUE-bound (active-instruction time ~27 % per wave), half the wave time parked -- not HBM (3.4 TB/s), not loads in flight.  Where
// the instructions went: 768 = 96 chunks of 8 on 64 lanes means a second chunk that only lanes 0-31 own (a full issue slot for half a wave,
// behind an exec-mask branch), 64-bit per-lane address arithmetic for every load and store, the dropout decode, gamma re-read per row.
// This kernel: TWO rows per wave iteration, THREE chunks per lane -- (row A, chunk l), (row B, chunk l), and chunk 64 + (l & 31) of row A for
// lanes 0-31, of row B for lanes 32-63 -- so every lane owns exactly 24 elements and no issue slot runs half empty; row bases are wave-uniform
// (SGPR base + one 32-bit lane offset per chunk: no vector address arithmetic), mean / rstd are scalar loads, the row sums come back as
// scalars (DPP + readlane).  Same partials layout ([3][blocks][H]) and the same second stage as the generic kernel; same padding-block rule.
__device__ __forceinline__ void lnb_unpack8(const uint4& q, float (&v)[8]) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 lnb_pack8(const float (&v)[8]) {
    uint4 q;
    q.x = pack2bf(v[0], v[1]); q.y = pack2bf(v[2], v[3]); q.z = pack2bf(v[4], v[5]); q.w = pack2bf(v[6], v[7]);
    return q;
}
#define LNP_H 768
#define LNP_NCH 96
#define LNP_BOUNDS __launch_bounds__(256)
// raw buffer access: SGPR resource + wave-uniform byte offset (SGPR) + one 32-bit lane offset -- no per-lane 64-bit addresses (the plain
// pointer form cost 2 VGPRs per access stream and a v_lshl_add_u64 per access; the kernel spilled at 4 waves per SIMD)
typedef int lnb_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lnb_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint4 lnb_ld16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const lnb_v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
}
__device__ __forceinline__ void lnb_st16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, const uint4& q) {
    const lnb_v4i v = {(int)q.x, (int)q.y, (int)q.z, (int)q.w};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, 0);
}
__global__ LNP_BOUNDS void ln_bwd_pair768_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ z,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, bf16_t* __restrict__ dz,
                                                             bf16_t* __restrict__ dbranch, float* __restrict__ partials, int M,
                                                             uint32_t thresh, float inv_keep, const int* __restrict__ zkend,
                                                             const int* __restrict__ zguard, int zL, const uint8_t* __restrict__ keepbits) {
    extern __shared__ float red[];         // gamma [768] during the row loop, then [3][4 waves][768]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    // persistent over row PAIRS (grid: pair_grid(), 2 workgroups per CU), pair q goes to wave (q mod waves): with one 16-row block per
    // workgroup, 1024 blocks on 768 resident slots (148 VGPRs) were a full round plus a third of one
    const bool zpad = zkend != nullptr && *zguard == 0;    // rows of trailing padding hold a known exact-zero dy: zero rows out, nothing summed
    const int half = l >> 5;                                // the third chunk belongs to row A (lanes 0-31) or row B (lanes 32-63)
    const int ch2 = 64 + (l & 31);
    const uint32_t off0 = (uint32_t)l * 16u, off1 = (uint32_t)(LNP_H * 2) + off0, off2 = (uint32_t)half * (LNP_H * 2) + (uint32_t)ch2 * 16u;
    const uint32_t kof0 = (uint32_t)l, kof1 = LNP_NCH + kof0, kof2 = (uint32_t)half * LNP_NCH + (uint32_t)ch2;
    float ag0[8], ab0[8], ai0[8], ag2[8], ab2[8], ai2[8];   // column sums: chunk l (both rows) and chunk ch2 (this lane's row)
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag0[e] = ab0[e] = ai0[e] = ag2[e] = ab2[e] = ai2[e] = 0.f; }
    for (int i = threadIdx.x; i < LNP_H; i += 256) red[i] = gamma[i];
    __syncthreads();
    const bool drop = thresh != 0;
    const __amdgpu_buffer_rsrc_t r_dy = lnb_rsrc(dy), r_z = lnb_rsrc(z), r_dz = lnb_rsrc(dz), r_db = lnb_rsrc(dbranch ? dbranch : dz),
                                 r_kb = lnb_rsrc(keepbits ? (const void*)keepbits : (const void*)dy);
#pragma unroll 1
    for (int pair = blockIdx.x * 4 + w; pair < (M >> 1); pair += gridDim.x * 4) {
        const int mA = 2 * pair;                            // wave-uniform; rows mA (A) and mA + 1 (B)
        if (zpad) {                                         // (zL is even: both rows sit in the same sequence)
            const int zb = mA / zL;
            if (mA - zb * zL >= zkend[zb]) {
                const uint4 zq = make_uint4(0u, 0u, 0u, 0u);
                const uint32_t rowz = (uint32_t)mA * (LNP_H * 2);
                lnb_st16(r_dz, off0, rowz, zq); lnb_st16(r_dz, off1, rowz, zq); lnb_st16(r_dz, off2, rowz, zq);
                if (dbranch) { lnb_st16(r_db, off0, rowz, zq); lnb_st16(r_db, off1, rowz, zq); lnb_st16(r_db, off2, rowz, zq); }
                continue;
            }
        }
        const uint32_t rowb = (uint32_t)mA * (LNP_H * 2);   // byte offset of row A (M * 1536 < 2^31: checked by the launcher)
        const uint4 rd0 = lnb_ld16(r_dy, off0, rowb), rd1 = lnb_ld16(r_dy, off1, rowb), rd2 = lnb_ld16(r_dy, off2, rowb);
        const uint4 rz0 = lnb_ld16(r_z, off0, rowb), rz1 = lnb_ld16(r_z, off1, rowb), rz2 = lnb_ld16(r_z, off2, rowb);
        uint32_t kb0 = 0xffu, kb1 = 0xffu, kb2 = 0xffu;
        if (drop) {
            const uint32_t kbrow = (uint32_t)mA * LNP_NCH;
            kb0 = __builtin_amdgcn_raw_buffer_load_b8(r_kb, (int)kof0, (int)kbrow, 0);
            kb1 = __builtin_amdgcn_raw_buffer_load_b8(r_kb, (int)kof1, (int)kbrow, 0);
            kb2 = __builtin_amdgcn_raw_buffer_load_b8(r_kb, (int)kof2, (int)kbrow, 0);
        }
        const float muA = mean[mA], rsA = rstd[mA], muB = mean[mA + 1], rsB = rstd[mA + 1];
        const float mu2 = half ? muB : muA, rs2 = half ? rsB : rsA;
        // pass 1, chunk by chunk: xhat (kept), the column sums of dy and dy * xhat, the row sums of g = dy * gamma and g * xhat.  g itself is
        // NOT kept: pass 2 rebuilds it from the raw bf16 dy (3 instructions per element against 24 more live registers).  Register budget
        // (148 VGPRs = 3 waves per SIMD; measured alternatives, all slower or spilling: xhat rebuilt as well -> the scheduler interleaves the
        // chunks and needs 173-194; forced to 128 -> 28-62 spills, 24 -> 37 us): 48 accumulators + 24 xhat + 12 raw dy + one chunk of temporaries
        float x0[8], x1[8], x2[8];
        lnb_unpack8(rz0, x0); lnb_unpack8(rz1, x1); lnb_unpack8(rz2, x2);
        float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f, s1c = 0.f, s2c = 0.f;
        const float nmA = -muA * rsA, nmB = -muB * rsB, nm2 = -mu2 * rs2;
        {
            float gm[8], d[8];
            ld8<float>(red + l * 8, gm);
            lnb_unpack8(rd0, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x0[e] = x0[e] * rsA + nmA;
                ab0[e] += d[e]; ag0[e] += d[e] * x0[e];
                const float t = d[e] * gm[e];
                s1a += t; s2a += t * x0[e];
            }
            lnb_unpack8(rd1, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x1[e] = x1[e] * rsB + nmB;
                ab0[e] += d[e]; ag0[e] += d[e] * x1[e];
                const float t = d[e] * gm[e];
                s1b += t; s2b += t * x1[e];
            }
            ld8<float>(red + ch2 * 8, gm);
            lnb_unpack8(rd2, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x2[e] = x2[e] * rs2 + nm2;
                ab2[e] += d[e]; ag2[e] += d[e] * x2[e];
                const float t = d[e] * gm[e];
                s1c += t; s2c += t * x2[e];
            }
        }
        s1a += half ? 0.f : s1c; s2a += half ? 0.f : s2c;
        s1b += half ? s1c : 0.f; s2b += half ? s2c : 0.f;
        const float S1A = wave_sum_dpp(s1a) * (1.0f / LNP_H), S2A = wave_sum_dpp(s2a) * (1.0f / LNP_H);
        const float S1B = wave_sum_dpp(s1b) * (1.0f / LNP_H), S2B = wave_sum_dpp(s2b) * (1.0f / LNP_H);
        const float S12 = half ? S1B : S1A, S22 = half ? S2B : S2A;
        float g0[8], g1[8], g2[8];
        // (opaque to the optimiser: otherwise it keeps pass 1's unpacked dy alive across the reductions instead of re-deriving it)
        uint4 qd0 = rd0, qd1 = rd1, qd2 = rd2;
        asm volatile("" : "+v"(qd0.x), "+v"(qd0.y), "+v"(qd0.z), "+v"(qd0.w), "+v"(qd1.x), "+v"(qd1.y), "+v"(qd1.z), "+v"(qd1.w),
                          "+v"(qd2.x), "+v"(qd2.y), "+v"(qd2.z), "+v"(qd2.w));
        {
            float gm[8];
            ld8<float>(red + l * 8, gm);
            lnb_unpack8(qd0, g0); lnb_unpack8(qd1, g1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                g0[e] = rsA * (g0[e] * gm[e] - S1A - x0[e] * S2A);
                g1[e] = rsB * (g1[e] * gm[e] - S1B - x1[e] * S2B);
            }
            ld8<float>(red + ch2 * 8, gm);
            lnb_unpack8(qd2, g2);
#pragma unroll
            for (int e = 0; e < 8; ++e) g2[e] = rs2 * (g2[e] * gm[e] - S12 - x2[e] * S22);
        }
        lnb_st16(r_dz, off0, rowb, lnb_pack8(g0));
        lnb_st16(r_dz, off1, rowb, lnb_pack8(g1));
        lnb_st16(r_dz, off2, rowb, lnb_pack8(g2));
        if (drop) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // bit e -> 0 / all ones -> 0.0f / inv_keep
                const float k0 = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe(kb0, e, 1) & __float_as_uint(inv_keep));
                const float k1 = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe(kb1, e, 1) & __float_as_uint(inv_keep));
                const float k2 = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe(kb2, e, 1) & __float_as_uint(inv_keep));
                g0[e] *= k0; g1[e] *= k1; g2[e] *= k2;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { ai0[e] += g0[e]; ai0[e] += g1[e]; ai2[e] += g2[e]; }
        if (dbranch) {
            lnb_st16(r_db, off0, rowb, lnb_pack8(g0));
            lnb_st16(r_db, off1, rowb, lnb_pack8(g1));
            lnb_st16(r_db, off2, rowb, lnb_pack8(g2));
        }
    }
    __syncthreads();                       // every wave is done reading gamma from `red`
    // the two lanes l and l ^ 32 hold the sums of the SAME columns of chunk ch2 (one per row of the pair): combine them, lanes 0-31 write
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ag2[e] += __shfl_xor(ag2[e], 32, 64); ab2[e] += __shfl_xor(ab2[e], 32, 64); ai2[e] += __shfl_xor(ai2[e], 32, 64);
    }
    st8<float>(red + (0 * 4 + w) * LNP_H + l * 8, ag0);
    st8<float>(red + (1 * 4 + w) * LNP_H + l * 8, ab0);
    st8<float>(red + (2 * 4 + w) * LNP_H + l * 8, ai0);
    if (l < 32) {
        st8<float>(red + (0 * 4 + w) * LNP_H + ch2 * 8, ag2);
        st8<float>(red + (1 * 4 + w) * LNP_H + ch2 * 8, ab2);
        st8<float>(red + (2 * 4 + w) * LNP_H + ch2 * 8, ai2);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * LNP_H; i += 256) {
        const int k = i / LNP_H, c = i - k * LNP_H;
        const float sm = red[(k * 4 + 0) * LNP_H + c] + red[(k * 4 + 1) * LNP_H + c] + red[(k * 4 + 2) * LNP_H + c] + red[(k * 4 + 3) * LNP_H + c];
        partials[((size_t)k * gridDim.x + blockIdx.x) * LNP_H + c] = sm;
    }
}

// (a forward twin of this pair mapping -- dropout + residual + LayerNorm, two token rows per wave -- gave the same bits and the same time as the one-row
//  kernel in round 4 (20.0 vs 19.9 us per launch: its waves wait on memory, not on issue slots) and was removed from the product in round 6)

// out[c] (+)= sum_b partials[b*stride + offset + c]   (deterministic second stage)
// block = RED_CG float4 column groups (RED_COLS = 32 columns = one 128-B line per partial row) x RED_ROWS = 32 row lanes; each lane strides over
// the partial rows, LDS tree at the end.  (Was 64 columns x 16 lanes: the LayerNorm jobs of a layer -- 6 x 768 columns of 1024 partial rows,
// 19 MB -- ran on 72 workgroups with 64 dependent trips each: 12 us per layer; now 144 workgroups x 32 trips.)
#define RED_CG 8
#define RED_COLS 32
#define RED_ROWS 32
__global__ __launch_bounds__(256) void reduce_partials_strided_kernel(const float* partials, int nblocks, int stride, int offset,
                                                                      int n, float* out, int accumulate) {
    __shared__ float red[RED_ROWS][RED_COLS];
    const int cg = threadIdx.x & (RED_CG - 1), ry = threadIdx.x / RED_CG;
    const int c = blockIdx.x * RED_COLS + cg * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const bool vec = (c + 3 < n) && ((stride & 3) == 0) && ((offset & 3) == 0);
    if (vec) {
#pragma unroll 4
        for (int b = ry; b < nblocks; b += RED_ROWS) {
            const float4 v = *reinterpret_cast<const float4*>(partials + (size_t)b * stride + offset + c);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    } else {
        for (int b = ry; b < nblocks; b += RED_ROWS) {
            const float* p = partials + (size_t)b * stride + offset;
            if (c < n) a0 += p[c];
            if (c + 1 < n) a1 += p[c + 1];
            if (c + 2 < n) a2 += p[c + 2];
            if (c + 3 < n) a3 += p[c + 3];
        }
    }
    red[ry][cg * 4 + 0] = a0; red[ry][cg * 4 + 1] = a1; red[ry][cg * 4 + 2] = a2; red[ry][cg * 4 + 3] = a3;
    __syncthreads();
    if (threadIdx.x < RED_COLS) {
        const int cc = blockIdx.x * RED_COLS + threadIdx.x;
        if (cc < n) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < RED_ROWS; ++r) sum += red[r][threadIdx.x];
            out[cc] = accumulate ? out[cc] + sum : sum;
        }
    }
}
// three reductions in one launch (LayerNorm backward: dgamma, dbeta, dbias); blockIdx.y selects the output
struct Reduce3 { const float* part[3]; float* out[3]; };
__global__ __launch_bounds__(256) void reduce3_kernel(Reduce3 r, int nblocks, int n, int accumulate) {
    __shared__ float red[RED_ROWS][RED_COLS];
    float* out = r.out[blockIdx.y];
    if (!out) return;
    const float* partials = r.part[blockIdx.y];
    const int cg = threadIdx.x & (RED_CG - 1), ry = threadIdx.x / RED_CG;
    const int c = blockIdx.x * RED_COLS + cg * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c + 3 < n) {
#pragma unroll 4
        for (int b = ry; b < nblocks; b += RED_ROWS) {
            const float4 v = *reinterpret_cast<const float4*>(partials + (size_t)b * n + c);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    }
    red[ry][cg * 4 + 0] = a0; red[ry][cg * 4 + 1] = a1; red[ry][cg * 4 + 2] = a2; red[ry][cg * 4 + 3] = a3;
    __syncthreads();
    if (threadIdx.x < RED_COLS) {
        const int cc = blockIdx.x * RED_COLS + threadIdx.x;
        if (cc < n) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < RED_ROWS; ++q) sum += red[q][threadIdx.x];
            out[cc] = accumulate ? out[cc] + sum : sum;
        }
    }
}
// Several second-stage reductions in ONE launch (blockIdx.y = job): the composite layer backward defers the four of a layer
// (LN2: dgamma/dbeta/db2, colsum(du), LN1: dgamma/dbeta/dbo, colsum(dqkv) = 8 jobs) to one kernel at its end instead of four
// ~7 us launches between its GEMMs.  Same arithmetic and summation order as the single-job kernels above.
#define AMDSEG_MAX_REDUCE_JOBS 12
struct ReduceJob { const float* part; float* out; int nblocks, stride, offset, n; };
struct ReduceJobs { ReduceJob job[AMDSEG_MAX_REDUCE_JOBS]; int njobs, accumulate; };
__global__ __launch_bounds__(256) void reduce_jobs_kernel(ReduceJobs jobs) {
    __shared__ float red[RED_ROWS][RED_COLS];
    const ReduceJob j = jobs.job[blockIdx.y];
    if ((int)blockIdx.x * RED_COLS >= j.n) return;
    const int cg = threadIdx.x & (RED_CG - 1), ry = threadIdx.x / RED_CG;
    const int c = blockIdx.x * RED_COLS + cg * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const bool vec = (c + 3 < j.n) && ((j.stride & 3) == 0) && ((j.offset & 3) == 0);
    if (vec) {
#pragma unroll 4
        for (int b = ry; b < j.nblocks; b += RED_ROWS) {
            const float4 v = *reinterpret_cast<const float4*>(j.part + (size_t)b * j.stride + j.offset + c);
            a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
        }
    } else {
        for (int b = ry; b < j.nblocks; b += RED_ROWS) {
            const float* p = j.part + (size_t)b * j.stride + j.offset;
            if (c < j.n) a0 += p[c];
            if (c + 1 < j.n) a1 += p[c + 1];
            if (c + 2 < j.n) a2 += p[c + 2];
            if (c + 3 < j.n) a3 += p[c + 3];
        }
    }
    red[ry][cg * 4 + 0] = a0; red[ry][cg * 4 + 1] = a1; red[ry][cg * 4 + 2] = a2; red[ry][cg * 4 + 3] = a3;
    __syncthreads();
    if (threadIdx.x < RED_COLS) {
        const int cc = blockIdx.x * RED_COLS + threadIdx.x;
        if (cc < j.n) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < RED_ROWS; ++r) sum += red[r][threadIdx.x];
            j.out[cc] = jobs.accumulate ? j.out[cc] + sum : sum;
        }
    }
}
static thread_local ReduceJobs* g_defer = nullptr;       // set by amdseg_reduce_defer_begin: reductions are queued, not launched
static thread_local ReduceJobs g_defer_store;
void amdseg_reduce_defer_begin(int accumulate) { g_defer_store.njobs = 0; g_defer_store.accumulate = accumulate; g_defer = &g_defer_store; }
int amdseg_reduce_defer_flush(hipStream_t s) {
    ReduceJobs* d = g_defer;
    g_defer = nullptr;
    if (!d || d->njobs == 0) return AMDSEG_OK;
    int nmax = 0;
    for (int i = 0; i < d->njobs; ++i) nmax = d->job[i].n > nmax ? d->job[i].n : nmax;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3((nmax + RED_COLS - 1) / RED_COLS, d->njobs), dim3(256), 0, s, *d);
    return amdseg_launch_status();
}
// true if the job was queued (the caller then must keep `partials` untouched until the flush)
static inline bool defer_reduce(const float* partials, int nblocks, int stride, int offset, int n, float* out, int accumulate) {
    if (!g_defer || g_defer->njobs >= AMDSEG_MAX_REDUCE_JOBS || g_defer->accumulate != accumulate) return false;
    g_defer->job[g_defer->njobs++] = ReduceJob{partials, out, nblocks, stride, offset, n};
    return true;
}
static inline void launch_reduce(const float* partials, int nblocks, int stride, int offset, int n, float* out, int accumulate,
                                 hipStream_t s) {
    if (defer_reduce(partials, nblocks, stride, offset, n, out, accumulate)) return;
    hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3((n + RED_COLS - 1) / RED_COLS), dim3(256), 0, s, partials, nblocks, stride, offset, n, out, accumulate);
}

void amdseg_reduce_rows(const float* partials, int nblocks, int stride, int n, float* out, int accumulate, hipStream_t s) {
    launch_reduce(partials, nblocks, stride, 0, n, out, accumulate, s);
}

// ------------------------------------------------------------------------------------------------ column sums
// x[M, ld] (first N columns) -> partials[nblk][N]; block = 256 threads = 32 column-chunks(8) x 8 row lanes
#define CS_ROWS 128
// lo_off != 0: x is a split-bf16 image, the value of column c is x[c] + x[lo_off + c] ("parity" precision: bias gradients from the image the
// backward already holds, no fp32 copy of the gradient)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, int ld, float* partials, int M, int N, int lo_off = 0) {
    __shared__ float red[8][256];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = (blockIdx.y * 32 + cx) * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < N) {
        const int r1 = min(M, (int)(blockIdx.x + 1) * CS_ROWS);
        for (int r = blockIdx.x * CS_ROWS + ry; r < r1; r += 8) {
            float v[8]; ld8<T>(x + (size_t)r * ld + col, v);
            if (lo_off) {
                float u[8]; ld8<T>(x + (size_t)r * ld + lo_off + col, u);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += u[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ry][cx * 8 + e] = acc[e];
    __syncthreads();
    const int c = threadIdx.x;
    const int gc = blockIdx.y * 256 + c;
    if (gc < N) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][c];
        partials[(size_t)blockIdx.x * N + gc] = s;
    }
}

// ------------------------------------------------------------------------------------------------ dropout / cast
template <typename TI, typename TO>
__global__ void dropout_kernel(const TI* x, TO* y, size_t n8, uint32_t thresh, float inv_keep, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        float v[8]; ld8<TI>(x + i * 8, v);
        if (thresh) drop8_apply(seed, i, thresh, inv_keep, v);
        st8<TO>(y + i * 8, v);
    }
}

// W[N,K] fp32 master -> bf16 copy Wb[N,K] and bf16 transpose Wt[K,N]; 64x64 tiles through LDS
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* W, bf16_t* Wb, bf16_t* Wt, int N, int K) {
    __shared__ bf16_t tile[64][66];
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;      // 16 x 16 threads, each 4 columns x 4 rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty * 4 + i;
        const float4 v = *reinterpret_cast<const float4*>(W + (size_t)(n0 + r) * K + k0 + tx * 4);
        const bf16_t b0 = f2bf(v.x), b1 = f2bf(v.y), b2 = f2bf(v.z), b3 = f2bf(v.w);
        tile[r][tx * 4 + 0] = b0; tile[r][tx * 4 + 1] = b1; tile[r][tx * 4 + 2] = b2; tile[r][tx * 4 + 3] = b3;
        if (Wb) {
            uint2 pk; pk.x = (uint32_t)b0 | ((uint32_t)b1 << 16); pk.y = (uint32_t)b2 | ((uint32_t)b3 << 16);
            *reinterpret_cast<uint2*>(Wb + (size_t)(n0 + r) * K + k0 + tx * 4) = pk;
        }
    }
    __syncthreads();
    if (Wt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = ty * 4 + i;        // row of Wt within the tile
            uint2 pk;
            pk.x = (uint32_t)tile[tx * 4 + 0][kr] | ((uint32_t)tile[tx * 4 + 1][kr] << 16);
            pk.y = (uint32_t)tile[tx * 4 + 2][kr] | ((uint32_t)tile[tx * 4 + 3][kr] << 16);
            *reinterpret_cast<uint2*>(Wt + (size_t)(k0 + kr) * N + n0 + tx * 4) = pk;
        }
    }
}

// batched form: every matrix of the model in ONE launch (the per-step shadow refresh after AdamW was 60 launches of ~6 us)
#define CT_MAXB 64
struct CastTransposeBatch {
    const float* W[CT_MAXB]; bf16_t* Wb[CT_MAXB]; bf16_t* Wt[CT_MAXB];
    int N[CT_MAXB], K[CT_MAXB], tile0[CT_MAXB + 1];      // tile0: prefix sum of (N/64)*(K/64)
    int n;
    const int* only_if;                                  // optional device flag: the launch does nothing when *only_if == 0
};
__global__ __launch_bounds__(256) void cast_transpose_batched_kernel(CastTransposeBatch b) {
    __shared__ bf16_t tile[64][66];
    if (b.only_if && *b.only_if == 0) return;
    // (the conditional form is launched with a capped grid and walks the tiles: its usual case is 'nothing changed', and 20736 workgroups that only
    //  read the flag still cost ~15 us per inference forward)
    for (int tb = blockIdx.x; tb < b.tile0[b.n]; tb += gridDim.x) {
    int m = 0;
    while (m + 1 < b.n && tb >= b.tile0[m + 1]) ++m;
    const float* W = b.W[m]; bf16_t* Wb = b.Wb[m]; bf16_t* Wt = b.Wt[m];
    const int N = b.N[m], K = b.K[m], t = tb - b.tile0[m], kt = K / 64;
    const int n0 = (t / kt) * 64, k0 = (t % kt) * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty * 4 + i;
        if (!W) {                                           // transposes only, from the bf16 copies (written by the fused AdamW pass)
            const uint2 q = *reinterpret_cast<const uint2*>(Wb + (size_t)(n0 + r) * K + k0 + tx * 4);
            tile[r][tx * 4 + 0] = (bf16_t)(q.x & 0xffffu); tile[r][tx * 4 + 1] = (bf16_t)(q.x >> 16);
            tile[r][tx * 4 + 2] = (bf16_t)(q.y & 0xffffu); tile[r][tx * 4 + 3] = (bf16_t)(q.y >> 16);
            continue;
        }
        const float4 v = *reinterpret_cast<const float4*>(W + (size_t)(n0 + r) * K + k0 + tx * 4);
        const bf16_t b0 = f2bf(v.x), b1 = f2bf(v.y), b2 = f2bf(v.z), b3 = f2bf(v.w);
        tile[r][tx * 4 + 0] = b0; tile[r][tx * 4 + 1] = b1; tile[r][tx * 4 + 2] = b2; tile[r][tx * 4 + 3] = b3;
        if (Wb) {
            uint2 pk; pk.x = (uint32_t)b0 | ((uint32_t)b1 << 16); pk.y = (uint32_t)b2 | ((uint32_t)b3 << 16);
            *reinterpret_cast<uint2*>(Wb + (size_t)(n0 + r) * K + k0 + tx * 4) = pk;
        }
    }
    __syncthreads();
    if (Wt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = ty * 4 + i;
            uint2 pk;
            pk.x = (uint32_t)tile[tx * 4 + 0][kr] | ((uint32_t)tile[tx * 4 + 1][kr] << 16);
            pk.y = (uint32_t)tile[tx * 4 + 2][kr] | ((uint32_t)tile[tx * 4 + 3][kr] << 16);
            *reinterpret_cast<uint2*>(Wt + (size_t)(k0 + kr) * N + n0 + tx * 4) = pk;
        }
    }
    __syncthreads();                                        // the tile is rewritten by the next iteration
    }
}

// ------------------------------------------------------------------------------------------------ small-C row dot (heads)
// logits[m][c] = x[m,:] . W[c,:] + b[c]   (classifier H->2, TSSP H->3;  modules/loss_calculator.py:17,42, modules/tssp.py:14,31)
template <typename T, int NCH>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const T* x, const float* W, const float* b, float* out, int M, int H, int C) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int m = blockIdx.x * ROWS_PER_BLOCK + w;
    if (m >= M) return;
    const int nch = H >> 3;
    float v[NCH][8];
    row_load<T, NCH>(x + (size_t)m * H, nch, l, v);
    for (int c = 0; c < C; ++c) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = l + k * 64;
            if (ch < nch) {
                float ww[8]; ld8<float>(W + (size_t)c * H + ch * 8, ww);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[k][e] * ww[e];
            }
        }
        s = wave_sum(s);
        if (l == 0) out[(size_t)m * C + c] = s + (b ? b[c] : 0.f);
    }
}
// dx[m,:] = sum_c dl[m][c] W[c,:] ; partial dW[c,:] = sum_m dl[m][c] x[m,:] ; partial db[c] = sum_m dl[m][c]   (C <= 4)
template <typename T, int NCH>
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const T* x, const float* W, const float* dl, T* dx, float* partials,
                                                         int M, int H, int C) {
    extern __shared__ float red[];      // [4 waves][C][H]
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int nch = H >> 3;
    float aw[4][NCH][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) aw[c][k][e] = 0.f;
    float adb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int rr = 0; rr < LNB_ROWS / 4; ++rr) {
        const int m = blockIdx.x * LNB_ROWS + rr * 4 + w;
        if (m >= M) break;
        float v[NCH][8], o[NCH][8];
        row_load<T, NCH>(x + (size_t)m * H, nch, l, v);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[k][e] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < C) {
                const float d = dl[(size_t)m * C + c];
                adb[c] += d;
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    const int ch = l + k * 64;
                    if (ch < nch) {
                        float ww[8]; ld8<float>(W + (size_t)c * H + ch * 8, ww);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { o[k][e] += d * ww[e]; aw[c][k][e] += d * v[k][e]; }
                    }
                }
            }
        }
        if (dx) row_store<T, NCH>(dx + (size_t)m * H, nch, l, o);
    }
    if (!partials) return;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < C) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                const int ch = l + k * 64;
                if (ch < nch) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) red[((size_t)w * C + c) * H + ch * 8 + e] = aw[c][k][e];
                }
            }
        }
    __syncthreads();
    // partial layout: [C*H weights | C biases] per block
    const int PW = C * H + C;
    for (int i = threadIdx.x; i < C * H; i += 256) {
        const float s = red[(size_t)0 * C * H + i] + red[(size_t)1 * C * H + i] + red[(size_t)2 * C * H + i] + red[(size_t)3 * C * H + i];
        partials[(size_t)blockIdx.x * PW + i] = s;
    }
    // bias partials: every lane of a wave holds the same adb; combine the 4 waves via LDS tail
    __syncthreads();
    if (l == 0)
        for (int c = 0; c < C; ++c) red[w * 4 + c] = adb[c];
    __syncthreads();
    if (threadIdx.x < C)
        partials[(size_t)blockIdx.x * PW + C * H + threadIdx.x] = red[0 * 4 + threadIdx.x] + red[1 * 4 + threadIdx.x] + red[2 * 4 + threadIdx.x] + red[3 * 4 + threadIdx.x];
}

#define ROWK(K, T, H, ...) do { if ((H) <= 1024) hipLaunchKernelGGL((K<T, 2>), __VA_ARGS__); else hipLaunchKernelGGL((K<T, 4>), __VA_ARGS__); } while (0)
// the same through the launch timer (prof.h): class, algorithmic bytes of the launch
#define ROWK_PROF(cls, work, K, T, H, ...) do { if ((H) <= 1024) AMDSEG_LAUNCH_PROF(cls, work, (K<T, 2>), __VA_ARGS__); else AMDSEG_LAUNCH_PROF(cls, work, (K<T, 4>), __VA_ARGS__); } while (0)
// ------------------------------------------------------------------------------------------------ launchers
// 16-bit threshold of drop8_apply (common.h); p >= 1 drops everything (inv_keep 0 instead of inf so that 0 * inv_keep stays 0)
static inline void drop_params(float p, uint32_t& thresh, float& inv_keep) {
    if (p <= 0.f) { thresh = 0; inv_keep = 1.f; return; }
    double t = (double)p * 65536.0 + 0.5;
    thresh = t >= 65536.0 ? 65536u : (uint32_t)t;
    if (thresh == 0) thresh = 1;
    inv_keep = thresh >= 65536u ? 0.f : (float)(65536.0 / (65536.0 - (double)thresh));
}

int amdseg_embed_ln_fwd_impl(const int64_t* ids, const int64_t* type_ids, const float* word, const float* pos,
                             const float* type, const float* gamma, const float* beta, void* z, void* out, float* mean,
                             float* rstd, int M, int L, int H, int vocab, int type_vocab, int npos,
                             const int64_t* pos_ids, float eps, float p, uint64_t seed, int dtype, hipStream_t s) {
    if (!ids || !word || !pos || !type || !gamma || !beta || !out) return AMDSEG_ERR_ARG;
    if (M <= 0 || H <= 0 || (H % 8) || H > 8 * 64 * MAXCH || L <= 0) return AMDSEG_ERR_SHAPE;
    const int pre = p < 0.f ? 1 : 0;                        // p < 0: dropout(|p|) BEFORE the LayerNorm (BigBird embeddings)
    uint32_t th; float ik; drop_params(pre ? -p : p, th, ik);
    dim3 grid((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    if (dtype == AMDSEG_BF16)
        ROWK(embed_ln_fwd_kernel, bf16_t, H, grid, dim3(256), 0, s, ids, type_ids, pos_ids, word, pos, type, gamma, beta,
                           (bf16_t*)z, (bf16_t*)out, mean, rstd, M, L, H, vocab, type_vocab, npos, eps, th, ik, seed, pre);
    else
        ROWK(embed_ln_fwd_kernel, float, H, grid, dim3(256), 0, s, ids, type_ids, pos_ids, word, pos, type, gamma, beta,
                           (float*)z, (float*)out, mean, rstd, M, L, H, vocab, type_vocab, npos, eps, th, ik, seed, pre);
    return amdseg_launch_status();
}

int amdseg_embed_bwd_impl(const void* dz, const int64_t* ids, const int64_t* type_ids, const int64_t* pos_ids,
                           float* dword, float* dpos, float* dtype_emb, int M, int L, int H, int vocab, int type_vocab,
                           int npos, int pad_id, int dtype, hipStream_t s) {
    if (!dz || !ids || !dword || !dpos || !dtype_emb) return AMDSEG_ERR_ARG;
    if (M <= 0 || H <= 0) return AMDSEG_ERR_SHAPE;
    if (L <= 0) return AMDSEG_ERR_SHAPE;
    dim3 grid((M + EMB_ROWS - 1) / EMB_ROWS, (H + 255) / 256), gridp(L < M ? L : M, (H + 255) / 256);
    if (dtype == AMDSEG_BF16) {
        hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dz, ids, type_ids, pos_ids, dword, dpos,
                           dtype_emb, M, L, H, vocab, type_vocab, npos, pad_id);
        if (!pos_ids) hipLaunchKernelGGL(embed_bwd_pos_kernel<bf16_t>, gridp, dim3(256), 0, s, (const bf16_t*)dz, dpos, M, L, H, npos);
    } else {
        hipLaunchKernelGGL(embed_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)dz, ids, type_ids, pos_ids, dword, dpos,
                           dtype_emb, M, L, H, vocab, type_vocab, npos, pad_id);
        if (!pos_ids) hipLaunchKernelGGL(embed_bwd_pos_kernel<float>, gridp, dim3(256), 0, s, (const float*)dz, dpos, M, L, H, npos);
    }
    return amdseg_launch_status();
}

int amdseg_scatter_rows_sorted_impl(const void* dz, const int64_t* keys, const int64_t* order, float* table, int M, int H, int nrows,
                                    long skip_key, int dtype, hipStream_t s) {
    if (!dz || !keys || !order || !table) return AMDSEG_ERR_ARG;
    if (M <= 0 || H <= 0 || nrows <= 0) return AMDSEG_ERR_SHAPE;
    dim3 grid(M, (H + 255) / 256);
    if (dtype == AMDSEG_BF16)
        hipLaunchKernelGGL(scatter_rows_sorted_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dz, keys, order, table, M, H, nrows, (int64_t)skip_key);
    else
        hipLaunchKernelGGL(scatter_rows_sorted_kernel<float>, grid, dim3(256), 0, s, (const float*)dz, keys, order, table, M, H, nrows, (int64_t)skip_key);
    return amdseg_launch_status();
}

static int pair_grid(int nblk);
int amdseg_add_ln_fwd_impl(void* y_inout_z, const void* resid, const float* gamma, const float* beta, void* out,
                           float* mean, float* rstd, int M, int H, float eps, float p, uint64_t seed, int dtype,
                           hipStream_t s, void* out_image, void* keepbits, bool keep_z) {
    if (!y_inout_z || !gamma || !beta || !out) return AMDSEG_ERR_ARG;      // resid == NULL: LayerNorm of y_inout_z as it stands (p ignored)
    if (M <= 0 || H <= 0 || (H % 8) || H > 8 * 64 * MAXCH) return AMDSEG_ERR_SHAPE;
    uint32_t th; float ik; drop_params(p, th, ik);
    dim3 grid((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    const double bytes_per_el = resid ? (keep_z ? 4.0 : 3.0) : 2.0;
    if (dtype == AMDSEG_BF16)
        ROWK_PROF(AMDSEG_PROF_ADD_LN_FWD, bytes_per_el * M * H * 2, add_ln_fwd_kernel, bf16_t, H, grid, dim3(256), 0, s, (bf16_t*)y_inout_z, (const bf16_t*)resid, gamma, beta,
                           (bf16_t*)out, mean, rstd, M, H, eps, th, ik, seed, (bf16_t*)out_image, (uint8_t*)keepbits, keep_z);
    else
        ROWK_PROF(AMDSEG_PROF_ADD_LN_FWD, bytes_per_el * M * H * 4, add_ln_fwd_kernel, float, H, grid, dim3(256), 0, s, (float*)y_inout_z, (const float*)resid, gamma, beta,
                           (float*)out, mean, rstd, M, H, eps, th, ik, seed, (bf16_t*)out_image, (uint8_t*)keepbits, keep_z);
    return amdseg_launch_status();
}

// amdseg_add_ln_fwd_impl + amdseg_attn_keepmask_impl(keep, ...) as ONE launch (add_ln_fwd_km_kernel); same arguments, same bits as the two calls
int amdseg_add_ln_fwd_km_impl(void* y_inout_z, const void* resid, const float* gamma, const float* beta, void* out, float* mean, float* rstd, int M, int H,
                              float eps, float p, uint64_t seed, int dtype, hipStream_t s, void* keepbits, bool keep_z,
                              void* keep, int B, int L, int heads, float p_attn, uint64_t seed_attn, const int* kend, int window, int nglobal) {
    if (!y_inout_z || !gamma || !beta || !out) return AMDSEG_ERR_ARG;
    if (M <= 0 || H <= 0 || (H % 8) || H > 8 * 64 * MAXCH) return AMDSEG_ERR_SHAPE;
    if (dtype != AMDSEG_BF16 && dtype != AMDSEG_F32) return AMDSEG_ERR_ARG;
    KeepMaskArgs k;
    const int rc = km_fill(k, keep, B, L, heads, p_attn, seed_attn, kend, window, nglobal);
    if (rc) return rc;
    AddLnArgs a = {};
    drop_params(p, a.thresh, a.inv_keep);
    a.y_z = y_inout_z; a.resid = resid; a.gamma = gamma; a.beta = beta; a.out = out; a.mean = mean; a.rstd = rstd; a.M = M; a.H = H; a.eps = eps;
    a.seed = seed; a.keepbits = (uint8_t*)keepbits; a.keep_z = keep_z ? 1 : 0;
    const unsigned n_ln = (unsigned)((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK), n_km = (unsigned)km_blocks(B, L, heads);
    const double bytes_per_el = resid ? (keep_z ? 4.0 : 3.0) : 2.0, esz = dtype == AMDSEG_BF16 ? 2.0 : 4.0;
    const double work = bytes_per_el * M * H * esz + 2.0 * B * heads * (double)L * L / 8.0;       // the rows' passes + the mask bytes written
    const unsigned n_all = n_ln + n_km;                     // (n_ln >= 1: the ratio is below 2^32)
    const unsigned ratio = (unsigned)((((uint64_t)n_km << 32) + n_all - 1) / n_all);
    if (dtype == AMDSEG_BF16) ROWK_PROF(AMDSEG_PROF_ADD_LN_FWD, work, add_ln_fwd_km_kernel, bf16_t, H, dim3(n_all), dim3(256), 0, s, a, k, ratio);
    else ROWK_PROF(AMDSEG_PROF_ADD_LN_FWD, work, add_ln_fwd_km_kernel, float, H, dim3(n_all), dim3(256), 0, s, a, k, ratio);
    return amdseg_launch_status();
}

// workgroups of the persistent pair kernel: 2 per CU (3 fit at 148 VGPRs; measured in the step at 32 x 512 tokens: 2 -> 20.9 us, 3 -> 21.3,
// one block per 16 rows = 1024 blocks on 768 slots -> 23.9), never more than one per 16 rows
static int pair_grid(int nblk) {
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = cus * 2;
    }
    return nblk < slots ? nblk : slots;
}

int amdseg_ln_bwd_impl(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                       void* dz, void* dbranch, float* partials, float* dgamma, float* dbeta, float* dbias, int M,
                       int H, float p, uint64_t seed, int accumulate, int dtype, hipStream_t s,
                       const int* zkend, const int* zguard, int zL, void* dense_grad_image, const void* keepbits) {
    if (!dy || !z || !mean || !rstd || !gamma || !dz) return AMDSEG_ERR_ARG;
    if (!zkend || !zguard || zL <= 0 || (M % zL) || (zL % LNB_ROWS)) { zkend = nullptr; zguard = nullptr; zL = 1; }
    if ((dgamma || dbeta || dbias) && !partials) return AMDSEG_ERR_ARG;
    if (M <= 0 || H <= 0 || (H % 8) || H > 8 * 64 * MAXCH) return AMDSEG_ERR_SHAPE;
    uint32_t th; float ik; drop_params(p, th, ik);
    const int nblk = (M + LNB_ROWS - 1) / LNB_ROWS;
    const size_t shm = (size_t)3 * 4 * H * sizeof(float);
    int nblk_eff = nblk;                                    // partial rows written per column sum (the pair kernel's grid is persistent)
    if (dtype == AMDSEG_BF16 && H == LNP_H && (M % LNB_ROWS) == 0 && (size_t)M * H * 2 < (1ull << 31) && partials && !dense_grad_image && (th == 0 || keepbits))
    {
        nblk_eff = pair_grid(nblk);
        AMDSEG_LAUNCH_PROF(AMDSEG_PROF_LN_BWD, (dbranch ? 4.0 : 3.0) * M * H * 2, ln_bwd_pair768_kernel, dim3(nblk_eff), dim3(256), shm, s,
                           (const bf16_t*)dy, (const bf16_t*)z, mean, rstd, gamma, (bf16_t*)dz, (bf16_t*)dbranch, partials, M, th, ik, zkend, zguard, zL,
                           (const uint8_t*)keepbits);
    }
    else if (dtype == AMDSEG_BF16)
        ROWK_PROF(AMDSEG_PROF_LN_BWD, (dbranch ? 4.0 : 3.0) * M * H * 2, ln_bwd_kernel, bf16_t, H, dim3(nblk), dim3(256), shm, s, (const bf16_t*)dy, (const bf16_t*)z, mean, rstd,
                           gamma, (bf16_t*)dz, (bf16_t*)dbranch, partials, M, H, th, ik, seed, zkend, zguard, zL, (bf16_t*)dense_grad_image,
                           (const uint8_t*)keepbits);
    else
        ROWK_PROF(AMDSEG_PROF_LN_BWD, (dbranch ? 4.0 : 3.0) * M * H * 4, ln_bwd_kernel, float, H, dim3(nblk), dim3(256), shm, s, (const float*)dy, (const float*)z, mean, rstd, gamma,
                           (float*)dz, (float*)dbranch, partials, M, H, th, ik, seed, zkend, zguard, zL, (bf16_t*)dense_grad_image,
                           (const uint8_t*)keepbits);
    if (partials) {
        Reduce3 r;
        r.part[0] = partials; r.part[1] = partials + (size_t)nblk_eff * H; r.part[2] = partials + (size_t)2 * nblk_eff * H;
        r.out[0] = dgamma; r.out[1] = dbeta; r.out[2] = dbias;
        if ((H % 4) == 0 && (dgamma || dbeta || dbias)) {
            bool queued = g_defer != nullptr && g_defer->njobs + 3 <= AMDSEG_MAX_REDUCE_JOBS && g_defer->accumulate == accumulate;
            if (queued)
                for (int k = 0; k < 3; ++k)
                    if (r.out[k]) queued = defer_reduce(r.part[k], nblk_eff, H, 0, H, r.out[k], accumulate) && queued;
            if (!queued) hipLaunchKernelGGL(reduce3_kernel, dim3((H + RED_COLS - 1) / RED_COLS, 3), dim3(256), 0, s, r, nblk_eff, H, accumulate);
        }
    }
    return amdseg_launch_status();
}

int amdseg_colsum_impl(const void* x, int ld, float* partials, float* out, int M, int N, int accumulate, int dtype,
                       hipStream_t s) {
    if (!x || !partials || !out) return AMDSEG_ERR_ARG;
    if (M <= 0 || N <= 0 || (N % 8) || (ld % 8)) return AMDSEG_ERR_SHAPE;
    const int nblk = (M + CS_ROWS - 1) / CS_ROWS;
    dim3 grid(nblk, (N + 255) / 256);
    if (dtype == AMDSEG_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, ld, partials, M, N);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ld, partials, M, N);
    launch_reduce(partials, nblk, N, 0, N, out, accumulate, s);
    return amdseg_launch_status();
}

int amdseg_colsum_split_impl(const void* x, int ld, int lo_off, float* partials, float* out, int M, int N, int accumulate, hipStream_t s) {
    if (!x || !partials || !out) return AMDSEG_ERR_ARG;
    if (M <= 0 || N <= 0 || (N % 8) || (ld % 8) || (lo_off % 8) || lo_off <= 0) return AMDSEG_ERR_SHAPE;
    const int nblk = (M + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(nblk, (N + 255) / 256), dim3(256), 0, s, (const bf16_t*)x, ld, partials, M, N, lo_off);
    launch_reduce(partials, nblk, N, 0, N, out, accumulate, s);
    return amdseg_launch_status();
}

int amdseg_dropout_impl(const void* x, void* y, size_t n, float p, uint64_t seed, int dtype_in, int dtype_out,
                        hipStream_t s) {
    if (!x || !y) return AMDSEG_ERR_ARG;
    if (n == 0 || (n % 8)) return AMDSEG_ERR_SHAPE;
    uint32_t th; float ik; drop_params(p, th, ik);
    const size_t n8 = n / 8;
    const unsigned grid = (unsigned)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);
    if (dtype_in == AMDSEG_BF16 && dtype_out == AMDSEG_BF16)
        hipLaunchKernelGGL((dropout_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n8, th, ik, seed);
    else if (dtype_in == AMDSEG_BF16 && dtype_out == AMDSEG_F32)
        hipLaunchKernelGGL((dropout_kernel<bf16_t, float>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (float*)y, n8, th, ik, seed);
    else if (dtype_in == AMDSEG_F32 && dtype_out == AMDSEG_BF16)
        hipLaunchKernelGGL((dropout_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, s, (const float*)x, (bf16_t*)y, n8, th, ik, seed);
    else
        hipLaunchKernelGGL((dropout_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, n8, th, ik, seed);
    return amdseg_launch_status();
}

// *guard = 1 if any element of a row at a position >= kend[b] is not an exact zero (NaN counts), else 0.  One wave per row; rows in front
// of kend leave at once, so the pass reads only the padded rows (reference: none -- this is the run-time check behind the
// amdseg_bert_cfg.pad_* fields: the backward may drop work on rows whose gradient is an exact zero only after it has looked)
__global__ __launch_bounds__(256) void pad_rows_guard_kernel(const float* __restrict__ x, const int* __restrict__ kend, int L, int H, int M,
                                                             int* __restrict__ guard) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (row >= M) return;
    const int b = row / L, p = row - b * L;
    if (p < kend[b]) return;
    const float* r = x + (size_t)row * H;
    bool nz = false;
    for (int c = l * 4; c < H; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(r + c);
        nz |= !(v.x == 0.f) | !(v.y == 0.f) | !(v.z == 0.f) | !(v.w == 0.f);
    }
    if (__ballot(nz) != 0ull && l == 0) atomicOr(guard, 1);
}

// The per-batch padding plan the layer calls take (amdseg_bert_cfg.kend / seq_order / pad_runs / pad_counts) from the attention mask, on the
// device: kend[b] = (last position with a non-zero mask) + 1; seq_order = the sequences by decreasing kend (stable); runs = {first, end} 64-token
// tiles in front of kend for every sequence with kend > 0 (in batch order); counts = {tiles in the runs, runs}.
__global__ __launch_bounds__(256) void pad_kend_kernel(const int64_t* __restrict__ mask, int L, int* __restrict__ kend,
                                                       float* __restrict__ mask_bias, float bias) {
    __shared__ int part[4];
    const int b = blockIdx.x, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    int ke = 0;
    for (int p = threadIdx.x; p < L; p += 256) {
        const int64_t mv = mask[(size_t)b * L + p];
        if (mv != 0) ke = p + 1;                                   // p grows: the last hit of this lane
        if (mask_bias) mask_bias[(size_t)b * L + p] = (1.0f - (float)mv) * bias;    // the additive key mask, as the host mirror computed it
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ke = max(ke, __shfl_xor(ke, o, 64));
    if (l == 0) part[w] = ke;
    __syncthreads();
    if (threadIdx.x == 0) kend[b] = max(max(part[0], part[1]), max(part[2], part[3]));
}
__global__ __launch_bounds__(1024) void pad_plan_kernel(const int* __restrict__ kend, int B, int L, int* __restrict__ seq_order,
                                                        int* __restrict__ runs, int* __restrict__ counts) {
    extern __shared__ int sk[];
    for (int b = threadIdx.x; b < B; b += blockDim.x) sk[b] = kend[b];
    __syncthreads();
    const int nt = L / 64;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const int ke = sk[b];
        int rank = 0, live = 0;
        for (int j = 0; j < B; ++j) {
            const int kj = sk[j];
            rank += (kj > ke) || (kj == ke && j < b);
            live += (j < b) && kj > 0;
        }
        seq_order[rank] = b;
        if (ke > 0) { runs[2 * live] = b * nt; runs[2 * live + 1] = b * nt + (ke + 63) / 64; }
    }
    if (threadIdx.x == 0) {
        int tiles = 0, nr = 0;
        for (int j = 0; j < B; ++j) if (sk[j] > 0) { tiles += (sk[j] + 63) / 64; ++nr; }
        counts[0] = tiles; counts[1] = nr;
    }
}

int amdseg_pad_plan_impl(const int64_t* mask, int B, int L, int* kend, int* seq_order, int* runs, int* counts, float* mask_bias, float bias,
                         hipStream_t s) {
    if (!mask || !kend || !seq_order || !runs || !counts) return AMDSEG_ERR_ARG;
    if (B <= 0 || B > 8192 || L <= 0 || (L % 64)) return AMDSEG_ERR_SHAPE;
    hipLaunchKernelGGL(pad_kend_kernel, dim3(B), dim3(256), 0, s, mask, L, kend, mask_bias, bias);
    hipLaunchKernelGGL(pad_plan_kernel, dim3(1), dim3(B < 1024 ? ((B + 63) / 64) * 64 : 1024), (size_t)B * sizeof(int), s, kend, B, L, seq_order,
                       runs, counts);
    return amdseg_launch_status();
}

int amdseg_pad_rows_guard_impl(const float* x, const int* kend, int B, int L, int H, int* guard, hipStream_t s) {
    if (!x || !kend || !guard) return AMDSEG_ERR_ARG;
    if (B <= 0 || L <= 0 || H <= 0 || (H % 4)) return AMDSEG_ERR_SHAPE;
    hipError_t e = hipMemsetAsync(guard, 0, sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    const int M = B * L;
    hipLaunchKernelGGL(pad_rows_guard_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, kend, L, H, M, guard);
    return amdseg_launch_status();
}

int amdseg_cast_impl(const void* x, void* y, size_t n, int dtype_in, int dtype_out, hipStream_t s) {
    return amdseg_dropout_impl(x, y, n, 0.f, 0, dtype_in, dtype_out, s);
}

int amdseg_cast_transpose_impl(const float* W, void* Wb, void* Wt, int N, int K, hipStream_t s) {
    if (!W || (!Wb && !Wt)) return AMDSEG_ERR_ARG;
    if (N <= 0 || K <= 0 || (N % 64) || (K % 64)) return AMDSEG_ERR_SHAPE;
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(K / 64, N / 64), dim3(256), 0, s, W, (bf16_t*)Wb, (bf16_t*)Wt, N, K);
    return amdseg_launch_status();
}

// 64-bit content checksum of a buffer (sum over 32-bit words of word * odd position weight; integer atomics: exact, order independent) and the
// comparison with the one taken last time -- "were these weights written since the bf16 copies were made?" answered on the device, no host read
__global__ __launch_bounds__(256) void checksum_u64_kernel(const uint4* __restrict__ x, size_t n16, unsigned long long* state) {
    unsigned long long acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#define CK_TERM(v, idx) do { const unsigned long long w = (unsigned long long)(((uint32_t)(idx) * 2654435761u) | 1u); \
        acc += (unsigned long long)(v).x * w + (unsigned long long)(v).y * (w + 2) + (unsigned long long)(v).z * (w + 4) + (unsigned long long)(v).w * (w + 6); } while (0)
    // four 16-B loads in flight per lane (one at a time left this pass latency-bound: 154 us for 340 MB)
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 v0 = x[i], v1 = x[i + stride], v2 = x[i + 2 * stride], v3 = x[i + 3 * stride];
        CK_TERM(v0, i); CK_TERM(v1, i + stride); CK_TERM(v2, i + 2 * stride); CK_TERM(v3, i + 3 * stride);
    }
    for (; i < n16; i += stride) { const uint4 v = x[i]; CK_TERM(v, i); }
#undef CK_TERM
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(state, acc);
}
__global__ void checksum_compare_kernel(unsigned long long* state, int* changed) {
    *changed = state[0] != state[1] ? 1 : 0;
    state[1] = state[0];
    state[0] = 0ull;
}
int amdseg_weights_changed_impl(const void* x, size_t nbytes, void* state, int* changed, hipStream_t s) {
    if (!x || !state || !changed) return AMDSEG_ERR_ARG;
    if (nbytes == 0 || (nbytes % 16) || ((uintptr_t)x % 16)) return AMDSEG_ERR_SHAPE;
    const size_t n16 = nbytes / 16;
    const unsigned grid = (unsigned)((n16 + 255) / 256 > 2048 ? 2048 : (n16 + 255) / 256);
    hipLaunchKernelGGL(checksum_u64_kernel, dim3(grid), dim3(256), 0, s, (const uint4*)x, n16, (unsigned long long*)state);
    hipLaunchKernelGGL(checksum_compare_kernel, dim3(1), dim3(1), 0, s, (unsigned long long*)state, changed);
    return amdseg_launch_status();
}

int amdseg_cast_transpose_batched_impl(int n, const float* const* W, void* const* Wb, void* const* Wt, const int* N, const int* K,
                                       hipStream_t s, const int* only_if) {
    if (n <= 0 || !N || !K || (!Wb && !Wt) || (!W && !(Wb && Wt))) return AMDSEG_ERR_ARG;      // W == NULL: Wb is the input, Wt the output
    for (int base = 0; base < n; base += CT_MAXB) {
        CastTransposeBatch b = {};
        b.n = n - base < CT_MAXB ? n - base : CT_MAXB;
        int tiles = 0;
        for (int i = 0; i < b.n; ++i) {
            const int j = base + i;
            if ((W && !W[j]) || (!W && (!Wb[j] || !Wt[j])) || N[j] <= 0 || K[j] <= 0 || (N[j] % 64) || (K[j] % 64)) return AMDSEG_ERR_SHAPE;
            b.W[i] = W ? W[j] : nullptr; b.Wb[i] = Wb ? (bf16_t*)Wb[j] : nullptr; b.Wt[i] = Wt ? (bf16_t*)Wt[j] : nullptr;
            b.N[i] = N[j]; b.K[i] = K[j]; b.tile0[i] = tiles;
            tiles += (N[j] / 64) * (K[j] / 64);
        }
        b.tile0[b.n] = tiles;
        b.only_if = only_if;
        hipLaunchKernelGGL(cast_transpose_batched_kernel, dim3(only_if && tiles > 1024 ? 1024 : tiles), dim3(256), 0, s, b);
    }
    return amdseg_launch_status();
}

int amdseg_rowdot_fwd_impl(const void* x, const float* W, const float* b, float* out, int M, int H, int C, int dtype,
                           hipStream_t s) {
    if (!x || !W || !out) return AMDSEG_ERR_ARG;
    if (M <= 0 || C <= 0 || C > 4 || (H % 8) || H > 8 * 64 * MAXCH) return AMDSEG_ERR_SHAPE;
    dim3 grid((M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK);
    if (dtype == AMDSEG_BF16)
        ROWK(rowdot_fwd_kernel, bf16_t, H, grid, dim3(256), 0, s, (const bf16_t*)x, W, b, out, M, H, C);
    else
        ROWK(rowdot_fwd_kernel, float, H, grid, dim3(256), 0, s, (const float*)x, W, b, out, M, H, C);
    return amdseg_launch_status();
}

int amdseg_rowdot_bwd_impl(const void* x, const float* W, const float* dlogits, void* dx, float* partials, float* dW,
                           float* db, int M, int H, int C, int accumulate, int dtype, hipStream_t s) {
    if (!x || !W || !dlogits) return AMDSEG_ERR_ARG;
    if ((dW || db) && !partials) return AMDSEG_ERR_ARG;
    if (M <= 0 || C <= 0 || C > 4 || (H % 8) || H > 8 * 64 * MAXCH) return AMDSEG_ERR_SHAPE;
    const int nblk = (M + LNB_ROWS - 1) / LNB_ROWS;
    const size_t shm = (size_t)4 * C * H * sizeof(float);
    if (dtype == AMDSEG_BF16)
        ROWK(rowdot_bwd_kernel, bf16_t, H, dim3(nblk), dim3(256), shm, s, (const bf16_t*)x, W, dlogits, (bf16_t*)dx, partials, M, H, C);
    else
        ROWK(rowdot_bwd_kernel, float, H, dim3(nblk), dim3(256), shm, s, (const float*)x, W, dlogits, (float*)dx, partials, M, H, C);
    if (partials) {
        const int PW = C * H + C;
        // partial rows are [C*H | C]; reduce weights and biases with a strided view: treat as N = PW columns
        // (dW and db are contiguous in the flat parameter layout only by convention, so reduce separately)
        if (dW) launch_reduce(partials, nblk, PW, 0, C * H, dW, accumulate, s);
        if (db) launch_reduce(partials, nblk, PW, C * H, C, db, accumulate, s);
    }
    return amdseg_launch_status();
}
