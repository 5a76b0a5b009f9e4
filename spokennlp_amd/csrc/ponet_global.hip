// PoNet global aggregation (the "g" branch of the pooling mixer), forward and backward, as streaming passes over the Hq / Hk column blocks
// of the 5H projection.  Reference: alimeeting4mug/src/models/modeling_ponet.py:34-109 calls ModelScope's PoNetModel, whose source is not in
// the reference tree; the arithmetic is the published one, restated in oracle/ponet_oracle.py `pooling()` (UNPINNED):
//     qbar_b   = mean over the valid tokens j of Hq[b, j, :]
//     s[b,h,j] = qbar_b[h-th 64 columns] . Hk[b, j, h-th 64 columns] / sqrt(64) + mask[b, j]
//     p = softmax_j(s),  pd = dropout(p),  g[b, c] = sum_j pd[b, head(c), j] Hk[b, j, c]
// Rounds 1-2 ran this on the Longformer global-row kernels (lf_wsum / lf_rowvec_dot / lf_softmax / lf_dx_update), which are written for
// heads x ALL H columns products: 12 x the arithmetic and 12 x the partial sums PoNet's head-sliced form needs, 10 launches forward and 12
// backward with torch glue in between (112 / 137 us per layer at PoNet-base, 8 x 4096 tokens).  Here a lane owns 8 consecutive columns of one
// head, the eight lanes of a head reduce their partial dots by DPP-class shuffles, and every pass reads Hk exactly once:
//     forward : pn_colmean (Hq, 50 MB) -> pn_vec -> pn_gflash (Hk, 50 MB: scores, online softmax, dropout, weighted sum) -> pn_gcombine
//     backward: pn_gbwd_dot (Hk) -> pn_gbwd_apply (Hk read, dHk written, t = sum_j ds Hk accumulated) -> pn_vec -> pn_dhq (dHq written)
// A wave takes 8 token rows and issues all of its loads before the first use (8 x 16 B per lane in flight, 16 waves per CU); the 8 waves of a
// workgroup fold their column sums / softmax states through LDS into ONE partial per 64 rows, and the second-stage sums run 4 chunk groups x
// 64 columns per workgroup (a thread per column walking 256 partials serially took 20-60 us per launch).
// Dropout decisions are the ones of amdseg_lf_softmax_fwd (drop_keep(seed, (b*heads + h)*L + j)), so both formulations drop the same
// probabilities.  All partial sums are written per workgroup and summed in a fixed order: results are bit-reproducible run to run.
#include "common.h"
#include "amdseg_internal.h"

#define PG_RW 8                        // token rows per wave: loaded together, 8 x 16 B (x NK) per lane in flight
#define PG_NW 8                        // waves per workgroup
#define PG_ROWS (PG_NW * PG_RW)        // 64 token rows per workgroup = one partial
#define PG_MAXH 1024

struct PgArgs {
    const bf16_t* hq; const bf16_t* hk; int ld;      // column blocks of the projection [B*L, ld]
    const float* coef;                               // [B, L] mean weights (1 / valid count on valid tokens, else 0)
    const float* mask_bias;                          // [B, L]
    int B, L, H, heads, nchunk;                      // nchunk = L / PG_ROWS
    uint32_t thresh; float inv_keep; uint64_t seed;
    float* part_acc; float* part_m; float* part_l;   // scratch: [B, nchunk, H], [B, nchunk, heads] x 2, then [B, H] (backward: dqbar)
    float* small;
    float* vec;                                      // [B, H]  qbar / 8 (forward), also read by backward
    float* scores; float* lse; float* g;             // [B, heads, L], [B, heads], [B, H]
    const float* dg; float* dpd;                     // backward: [B, H], [B, heads, L] scratch
    float* dqbar;                                    // [B, H]
    bf16_t* dhq; bf16_t* dhk; int ldd;
};

__device__ __forceinline__ float pg_sum8(float v) {  // sum over the 8 consecutive lanes of a head
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
    return v;
}
// the wave's PG_RW rows of one column block: every load issued before the first use
template <int NK>
__device__ __forceinline__ void pg_load_rows(const bf16_t* base, int ld, int l, int ncg, uint4 (&raw)[PG_RW][NK]) {
#pragma unroll
    for (int r = 0; r < PG_RW; ++r)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int cg = l + 64 * k;
            raw[r][k] = make_uint4(0u, 0u, 0u, 0u);
            if (cg < ncg) raw[r][k] = *reinterpret_cast<const uint4*>(base + (size_t)r * ld + cg * 8);
        }
}
__device__ __forceinline__ void pg_unpack(const uint4& q, float (&v)[8]) {
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u); v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
    v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xffff0000u); v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xffff0000u);
}
// the 8 waves' column sums -> one partial row: sm [PG_NW][H]
template <int NK>
__device__ __forceinline__ void pg_store_cols(float* sm, int H, int w, int l, int ncg, const float (&acc)[NK][8]) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int cg = l + 64 * k;
        if (cg < ncg) {
            *reinterpret_cast<float4*>(sm + (size_t)w * H + cg * 8) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
            *reinterpret_cast<float4*>(sm + (size_t)w * H + cg * 8 + 4) = make_float4(acc[k][4], acc[k][5], acc[k][6], acc[k][7]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward 1: column mean of Hq
template <int NK>
__global__ __launch_bounds__(512) void pn_colmean_kernel(PgArgs a) {
    extern __shared__ float sm[];                       // [PG_NW][H]
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j0 = blockIdx.x * PG_ROWS + w * PG_RW;
    const int ncg = a.H / 8;
    uint4 raw[PG_RW][NK];
    pg_load_rows<NK>(a.hq + ((size_t)b * a.L + j0) * a.ld, a.ld, l, ncg, raw);
    const float* cf = a.coef + (size_t)b * a.L + j0;
    float acc[NK][8];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
#pragma unroll
    for (int r = 0; r < PG_RW; ++r) {
        const float c = cf[r];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float v[8];
            pg_unpack(raw[r][k], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k][e] = fmaf(c, v[e], acc[k][e]);
        }
    }
    pg_store_cols<NK>(sm, a.H, w, l, ncg, acc);
    __syncthreads();
    float* out = a.part_acc + ((size_t)b * a.nchunk + blockIdx.x) * a.H;
    for (int c = threadIdx.x; c < a.H; c += 512) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PG_NW; ++i) s += sm[(size_t)i * a.H + c];
        out[c] = s;
    }
}

// out[b, c] = scale * sum over the chunks of part[b, chunk, c].  grid (H / 64, B): 64 columns x 4 groups of chunks
__global__ __launch_bounds__(256) void pn_vec_kernel(const float* __restrict__ part, float* __restrict__ out, int H, int nchunk, float scale) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    const float* p = part + (size_t)b * nchunk * H + c;
    float s0 = 0.f, s1 = 0.f;
    int i = grp;
    for (; i + 4 < nchunk; i += 8) { s0 += p[(size_t)i * H]; s1 += p[(size_t)(i + 4) * H]; }
    if (i < nchunk) s0 += p[(size_t)i * H];
    red[grp][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (grp == 0) out[(size_t)b * H + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) * scale;
}

// ---------------------------------------------------------------------------------------------------- forward 2: scores + online softmax + weighted sum
template <int NK>
__global__ __launch_bounds__(512) void pn_gflash_kernel(PgArgs a) {
    extern __shared__ float sm[];                       // [PG_NW][H] acc | [PG_NW][heads] m | [PG_NW][heads] l
    float* sm_m = sm + (size_t)PG_NW * a.H;
    float* sm_l = sm_m + PG_NW * a.heads;
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j0 = blockIdx.x * PG_ROWS + w * PG_RW;
    const int ncg = a.H / 8;
    uint4 raw[PG_RW][NK];
    pg_load_rows<NK>(a.hk + ((size_t)b * a.L + j0) * a.ld, a.ld, l, ncg, raw);
    float vq[NK][8], acc[NK][8], m[NK], ls[NK], srow[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int cg = l + 64 * k;
        m[k] = -INFINITY; ls[k] = 0.f; srow[k] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[k][e] = 0.f; vq[k][e] = 0.f; }
        if (cg < ncg) ld8<float>(a.vec + (size_t)b * a.H + cg * 8, vq[k]);
    }
    const float* mb = a.mask_bias + (size_t)b * a.L + j0;
    // the wave's 8 scores per head first, then ONE maximum and one weight per row: no running rescale inside a wave
    float sr[PG_RW][NK];
#pragma unroll
    for (int r = 0; r < PG_RW; ++r) {
        const float mk = mb[r];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float v[8];
            pg_unpack(raw[r][k], v);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf(vq[k][e], v[e], d);
            sr[r][k] = pg_sum8(d) + mk;
            if ((l & 7) == r) srow[k] = sr[r][k];           // lane r of the head keeps row r's score: one store per 8 rows below
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int cg = l + 64 * k, h = cg >> 3;
        float M = sr[0][k];
#pragma unroll
        for (int r = 1; r < PG_RW; ++r) M = fmaxf(M, sr[r][k]);
        m[k] = M;
#pragma unroll
        for (int r = 0; r < PG_RW; ++r) {
            const float e_ = __expf(sr[r][k] - M);
            const bool keep = !a.thresh || drop_keep(a.seed, ((size_t)b * a.heads + h) * a.L + j0 + r, a.thresh);
            const float wgt = keep ? e_ * a.inv_keep : 0.f;
            ls[k] += e_;
            float v[8];
            pg_unpack(raw[r][k], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k][e] = fmaf(wgt, v[e], acc[k][e]);
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int cg = l + 64 * k;
        if (cg < ncg) {
            a.scores[((size_t)b * a.heads + (cg >> 3)) * a.L + j0 + (l & 7)] = srow[k];
            if ((l & 7) == 0) { sm_m[w * a.heads + (cg >> 3)] = m[k]; sm_l[w * a.heads + (cg >> 3)] = ls[k]; }
        }
    }
    pg_store_cols<NK>(sm, a.H, w, l, ncg, acc);
    __syncthreads();
    const size_t pc = (size_t)b * a.nchunk + blockIdx.x;
    for (int c = threadIdx.x; c < a.H; c += 512) {
        const int h = c >> 6;
        float M = -INFINITY;
#pragma unroll
        for (int i = 0; i < PG_NW; ++i) M = fmaxf(M, sm_m[i * a.heads + h]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int i = 0; i < PG_NW; ++i) {
            const float f = __expf(sm_m[i * a.heads + h] - M);
            num = fmaf(sm[(size_t)i * a.H + c], f, num);
            den = fmaf(sm_l[i * a.heads + h], f, den);
        }
        a.part_acc[pc * a.H + c] = num;
        if ((c & 63) == 0) { a.part_m[pc * a.heads + h] = M; a.part_l[pc * a.heads + h] = den; }
    }
}

// g[b, c] = sum_chunks acc e^(m_chunk - M) / sum_chunks l e^(m_chunk - M);  lse[b, h] = M + log(that denominator).  grid (heads, B): one head's
// 64 columns x 4 groups of chunks
__global__ __launch_bounds__(256) void pn_gcombine_kernel(PgArgs a) {
    __shared__ float red[3][4][64];
    const int b = blockIdx.y, h = blockIdx.x, cl = threadIdx.x & 63, grp = threadIdx.x >> 6, c = h * 64 + cl;
    const float* pm = a.part_m + (size_t)b * a.nchunk * a.heads + h;
    const float* pl = a.part_l + (size_t)b * a.nchunk * a.heads + h;
    const float* pa = a.part_acc + (size_t)b * a.nchunk * a.H + c;
    float M = -INFINITY;
    for (int i = grp; i < a.nchunk; i += 4) M = fmaxf(M, pm[(size_t)i * a.heads]);
    red[0][grp][cl] = M;
    __syncthreads();
    M = fmaxf(fmaxf(red[0][0][cl], red[0][1][cl]), fmaxf(red[0][2][cl], red[0][3][cl]));
    float den = 0.f, num = 0.f;
    for (int i = grp; i < a.nchunk; i += 4) {
        const float f = __expf(pm[(size_t)i * a.heads] - M);
        den = fmaf(pl[(size_t)i * a.heads], f, den);
        num = fmaf(pa[(size_t)i * a.H], f, num);
    }
    red[1][grp][cl] = den; red[2][grp][cl] = num;
    __syncthreads();
    if (grp == 0) {
        den = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
        num = (red[2][0][cl] + red[2][1][cl]) + (red[2][2][cl] + red[2][3][cl]);
        a.g[(size_t)b * a.H + c] = num / den;
        if (cl == 0) a.lse[b * a.heads + h] = M + __logf(den);
    }
}

// ---------------------------------------------------------------------------------------------------- backward 1: dpd = dg . Hk per head, delta partials
template <int NK>
__global__ __launch_bounds__(512) void pn_gbwd_dot_kernel(PgArgs a) {
    __shared__ float sm_d[PG_NW][PG_MAXH / 64];
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j0 = blockIdx.x * PG_ROWS + w * PG_RW;
    const int ncg = a.H / 8;
    uint4 raw[PG_RW][NK];
    pg_load_rows<NK>(a.hk + ((size_t)b * a.L + j0) * a.ld, a.ld, l, ncg, raw);
    float dgv[NK][8], drow[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int cg = l + 64 * k;
        drow[k] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dgv[k][e] = 0.f;
        if (cg < ncg) ld8<float>(a.dg + (size_t)b * a.H + cg * 8, dgv[k]);
    }
#pragma unroll
    for (int r = 0; r < PG_RW; ++r)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float v[8];
            pg_unpack(raw[r][k], v);
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf(dgv[k][e], v[e], d);
            d = pg_sum8(d);
            if ((l & 7) == r) drow[k] = d;                  // lane r of the head finishes row r
        }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int cg = l + 64 * k, h = cg >> 3;
        float del = 0.f;
        if (cg < ncg) {
            const size_t idx = ((size_t)b * a.heads + h) * a.L + j0 + (l & 7);
            a.dpd[idx] = drow[k];
            const bool keep = !a.thresh || drop_keep(a.seed, idx, a.thresh);
            if (keep) del = __expf(a.scores[idx] - a.lse[b * a.heads + h]) * a.inv_keep * drow[k];
        }
        del = pg_sum8(del);
        if (cg < ncg && (l & 7) == 0) sm_d[w][h] = del;
    }
    __syncthreads();
    if ((int)threadIdx.x < a.heads) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PG_NW; ++i) s += sm_d[i][threadIdx.x];
        a.part_m[((size_t)b * a.nchunk + blockIdx.x) * a.heads + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------- backward 2: dHk, t partials
template <int NK>
__global__ __launch_bounds__(512) void pn_gbwd_apply_kernel(PgArgs a) {
    extern __shared__ float sm[];                       // [PG_NW][H]
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j0 = blockIdx.x * PG_ROWS + w * PG_RW;
    const int ncg = a.H / 8;
    uint4 raw[PG_RW][NK];
    pg_load_rows<NK>(a.hk + ((size_t)b * a.L + j0) * a.ld, a.ld, l, ncg, raw);
    float dgv[NK][8], vq[NK][8], acc[NK][8], pdv[NK], dsv[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int cg = l + 64 * k, h = cg >> 3;
        pdv[k] = 0.f; dsv[k] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { dgv[k][e] = 0.f; vq[k][e] = 0.f; acc[k][e] = 0.f; }
        float del = 0.f;
        if (cg < ncg) {
            ld8<float>(a.dg + (size_t)b * a.H + cg * 8, dgv[k]);
            ld8<float>(a.vec + (size_t)b * a.H + cg * 8, vq[k]);
            // delta[b, h] = sum over the chunks of the partials of pn_gbwd_dot: the 8 lanes of the head take every 8th chunk
            const float* pd_ = a.part_m + (size_t)b * a.nchunk * a.heads + h;
            for (int i = l & 7; i < a.nchunk; i += 8) del += pd_[(size_t)i * a.heads];
        }
        del = pg_sum8(del);
        if (cg < ncg) {                                     // lane r of the head: row r's probability terms
            const size_t idx = ((size_t)b * a.heads + h) * a.L + j0 + (l & 7);
            const float p = __expf(a.scores[idx] - a.lse[b * a.heads + h]);
            const bool keep = !a.thresh || drop_keep(a.seed, idx, a.thresh);
            const float kf = keep ? a.inv_keep : 0.f;
            pdv[k] = p * kf; dsv[k] = p * (a.dpd[idx] * kf - del);
        }
    }
    bf16_t* dbase = a.dhk + ((size_t)b * a.L + j0) * a.ldd;
#pragma unroll
    for (int r = 0; r < PG_RW; ++r)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int cg = l + 64 * k;
            const float pd = __shfl(pdv[k], (l & ~7) + r, 64), ds = __shfl(dsv[k], (l & ~7) + r, 64);
            float v[8], o[8];
            pg_unpack(raw[r][k], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = fmaf(pd, dgv[k][e], ds * vq[k][e]); acc[k][e] = fmaf(ds, v[e], acc[k][e]); }
            if (cg < ncg) {
                uint4 q;
                q.x = pack2bf(o[0], o[1]); q.y = pack2bf(o[2], o[3]); q.z = pack2bf(o[4], o[5]); q.w = pack2bf(o[6], o[7]);
                *reinterpret_cast<uint4*>(dbase + (size_t)r * a.ldd + cg * 8) = q;
            }
        }
    pg_store_cols<NK>(sm, a.H, w, l, ncg, acc);
    __syncthreads();
    float* out = a.part_acc + ((size_t)b * a.nchunk + blockIdx.x) * a.H;
    for (int c = threadIdx.x; c < a.H; c += 512) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < PG_NW; ++i) s += sm[(size_t)i * a.H + c];
        out[c] = s;
    }
}

// ---------------------------------------------------------------------------------------------------- backward 3: dHq[b, j, :] = coef[b, j] dqbar[b, :]
template <int NK>
__global__ __launch_bounds__(512) void pn_dhq_kernel(PgArgs a) {
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j0 = blockIdx.x * PG_ROWS + w * PG_RW;
    const int ncg = a.H / 8;
    float dq[NK][8];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int cg = l + 64 * k;
#pragma unroll
        for (int e = 0; e < 8; ++e) dq[k][e] = 0.f;
        if (cg < ncg) ld8<float>(a.dqbar + (size_t)b * a.H + cg * 8, dq[k]);
    }
    bf16_t* dbase = a.dhq + ((size_t)b * a.L + j0) * a.ldd;
    const float* cf = a.coef + (size_t)b * a.L + j0;
#pragma unroll
    for (int r = 0; r < PG_RW; ++r) {
        const float c = cf[r];
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int cg = l + 64 * k;
            if (cg >= ncg) continue;
            uint4 q;
            q.x = pack2bf(c * dq[k][0], c * dq[k][1]); q.y = pack2bf(c * dq[k][2], c * dq[k][3]);
            q.z = pack2bf(c * dq[k][4], c * dq[k][5]); q.w = pack2bf(c * dq[k][6], c * dq[k][7]);
            *reinterpret_cast<uint4*>(dbase + (size_t)r * a.ldd + cg * 8) = q;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- launchers
static int pg_fill(PgArgs& a, int B, int L, int H, int heads, float p, uint64_t seed, float* scratch) {
    if (B <= 0 || L <= 0 || (L % PG_ROWS) || heads <= 0 || H != heads * 64 || H > PG_MAXH) return AMDSEG_ERR_SHAPE;
    if (p < 0.f || p >= 1.f) return AMDSEG_ERR_ARG;
    a.B = B; a.L = L; a.H = H; a.heads = heads; a.nchunk = L / PG_ROWS; a.seed = seed;
    if (p <= 0.f) { a.thresh = 0; a.inv_keep = 1.f; }
    else {                                                // the parameters of amdseg_lf_softmax_fwd (same decisions)
        double t = (double)p * 4294967296.0;
        a.thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
        if (a.thresh == 0) a.thresh = 1;
        a.inv_keep = (float)(4294967296.0 / (4294967296.0 - (double)a.thresh));
    }
    a.part_acc = scratch;
    a.part_m = scratch + (size_t)B * a.nchunk * H;
    a.part_l = a.part_m + (size_t)B * a.nchunk * heads;
    a.small = a.part_l + (size_t)B * a.nchunk * heads;
    return AMDSEG_OK;
}
size_t amdseg_ponet_global_scratch_floats_impl(int B, int L, int H, int heads) {
    return (size_t)B * ((size_t)(L / PG_ROWS) * ((size_t)H + 2 * heads) + H);
}

#define PG_LAUNCH(kern, grid, lds) do { if (H <= 512) hipLaunchKernelGGL((kern<1>), grid, dim3(512), lds, s, a); \
                                        else hipLaunchKernelGGL((kern<2>), grid, dim3(512), lds, s, a); } while (0)

int amdseg_ponet_global_fwd_impl(const void* hq, const void* hk, int ld, const float* coef_mean, const float* mask_bias, int B, int L, int H,
                                 int heads, float p, uint64_t seed, float* scratch, float* vecq, float* scores, float* lse, float* g,
                                 hipStream_t s) {
    if (!hq || !hk || !coef_mean || !mask_bias || !scratch || !vecq || !scores || !lse || !g || ld < H || (ld % 8)) return AMDSEG_ERR_ARG;
    PgArgs a = {};
    int rc = pg_fill(a, B, L, H, heads, p, seed, scratch);
    if (rc) return rc;
    a.hq = (const bf16_t*)hq; a.hk = (const bf16_t*)hk; a.ld = ld; a.coef = coef_mean; a.mask_bias = mask_bias;
    a.vec = vecq; a.scores = scores; a.lse = lse; a.g = g;
    const dim3 rows(L / PG_ROWS, B);
    const size_t lds_cols = (size_t)PG_NW * H * sizeof(float), lds_flash = lds_cols + (size_t)2 * PG_NW * heads * sizeof(float);
    PG_LAUNCH(pn_colmean_kernel, rows, lds_cols);
    hipLaunchKernelGGL(pn_vec_kernel, dim3(H / 64, B), dim3(256), 0, s, a.part_acc, vecq, H, a.nchunk, 0.125f);
    PG_LAUNCH(pn_gflash_kernel, rows, lds_flash);
    hipLaunchKernelGGL(pn_gcombine_kernel, dim3(heads, B), dim3(256), 0, s, a);
    return amdseg_launch_status();
}

int amdseg_ponet_global_bwd_impl(const void* hk, int ld, const float* coef_mean, const float* vecq, const float* scores, const float* lse,
                                 const float* dg, int B, int L, int H, int heads, float p, uint64_t seed, float* scratch, float* dpd_ws,
                                 void* dhq, void* dhk, int ldd, hipStream_t s) {
    if (!hk || !coef_mean || !vecq || !scores || !lse || !dg || !scratch || !dpd_ws || !dhq || !dhk || ld < H || (ld % 8) || ldd < H || (ldd % 8))
        return AMDSEG_ERR_ARG;
    PgArgs a = {};
    int rc = pg_fill(a, B, L, H, heads, p, seed, scratch);
    if (rc) return rc;
    a.hk = (const bf16_t*)hk; a.ld = ld; a.coef = coef_mean; a.vec = (float*)vecq; a.scores = (float*)scores; a.lse = (float*)lse; a.dg = dg;
    a.dpd = dpd_ws; a.dqbar = a.small;
    a.dhq = (bf16_t*)dhq; a.dhk = (bf16_t*)dhk; a.ldd = ldd;
    const dim3 rows(L / PG_ROWS, B);
    const size_t lds_cols = (size_t)PG_NW * H * sizeof(float);
    PG_LAUNCH(pn_gbwd_dot_kernel, rows, 0);
    PG_LAUNCH(pn_gbwd_apply_kernel, rows, lds_cols);
    hipLaunchKernelGGL(pn_vec_kernel, dim3(H / 64, B), dim3(256), 0, s, a.part_acc, a.dqbar, H, a.nchunk, 0.125f);
    PG_LAUNCH(pn_dhq_kernel, rows, 0);
    return amdseg_launch_status();
}
