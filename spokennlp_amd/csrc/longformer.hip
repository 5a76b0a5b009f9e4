// Longformer "global row" path for gfx950: the attention of the (single, leading) global token -- [CLS] in the reference
// wrapper, emnlp2023-topic_segmentation/src/models/longformer_for_ts.py:55-58 -- over ALL tokens of its sequence with the
// separate query_global / key_global / value_global projections ([hf] models/longformer/modeling_longformer.py:964-1058).
//
// HF projects every token with key_global and value_global (two M x H x H GEMMs per layer) to feed ONE query row per
// sequence.  Here the projections are folded onto the query side instead (exact algebra, no approximation):
//     score[h, j] = qg_h . (Wkg_h x_j + bkg_h)  =  (Wkg_h^T qg_h) . x_j + const      ->  r[h] = Wkg_h^T qg_h    ([heads, H])
//     out_h       = sum_j p[h, j] (Wvg_h x_j + bvg_h) = Wvg_h (sum_j p[h, j] x_j) + bvg_h sum_j p[h, j]  ->  y[h] = P x
// (the per-head constant qg_h . bkg_h cancels in the softmax).  What remains on the O(L) side are three HBM-bound
// passes over the layer input x [B*L, H], implemented below; the O(heads * H^2) algebra on [B, heads, H] vectors is
// done by the host mirror (spokennlp_amd/longformer_engine.py).  Algorithmic bytes: one read of x per pass
// (2 B/elem bf16) -- forward 2 passes, backward 3 passes (+ one read-modify-write of dx).
//
//   lf_rowvec_dot : out[b, h, j] = vec[b, h, :] . x[b, j, :] (+ add_tok[b, j]) (+ add_bh[b, h])     scores / d(probs)
//   lf_softmax_*  : softmax over j (fp32), dropout by the stateless hash of common.h, and its backward
//   lf_wsum       : y[b, h, :] = sum_j coef[b, h, j] x[b, j, :]                                        y and d(r)
//   lf_dx_update  : dx[b, j, :] += sum_h coefA[b, h, j] vecA[b, h, :] + coefB[b, h, j] vecB[b, h, :]  rank-2*heads update
#include "common.h"
#include "amdseg_internal.h"
#include "tile64.h"

#define LF_MAXCH 2          // 8-element column chunks per lane: H <= 1024
#define LF_MAXHEADS 16

// ---------------------------------------------------------------------------------------------------- rowvec dot
// grid (L/64, B), 256 threads; wave w handles tokens blk*64 + w*16 .. +16; vec[b] staged in LDS as fp32 [heads][H]
template <typename T>
__global__ __launch_bounds__(256) void lf_rowvec_dot_kernel(const T* __restrict__ x, const float* __restrict__ vec,
                                                            const float* __restrict__ add_tok, const float* __restrict__ add_bh,
                                                            float* __restrict__ out, int L, int H, int heads, int ldx) {
    extern __shared__ __attribute__((aligned(16))) float lf_smem[];
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const float* vb = vec + (size_t)b * heads * H;
    for (int i = threadIdx.x * 4; i < heads * H; i += 256 * 4)
        *reinterpret_cast<float4*>(lf_smem + i) = *reinterpret_cast<const float4*>(vb + i);
    __syncthreads();
    const int nch = H / 8;
    for (int t = 0; t < 16; ++t) {
        const int j = blockIdx.x * 64 + w * 16 + t;
        const T* xp = x + ((size_t)b * L + j) * ldx;
        float xr[LF_MAXCH][8];
#pragma unroll
        for (int i = 0; i < LF_MAXCH; ++i) {
            const int c = l + 64 * i;
            if (c < nch) ld8<T>(xp + c * 8, xr[i]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xr[i][e] = 0.f;
            }
        }
        float res = 0.f;                                     // lane h keeps head h's score
        for (int h = 0; h < heads; ++h) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < LF_MAXCH; ++i) {
                const int c = l + 64 * i;
                if (c < nch) {
                    const float4 v0 = *reinterpret_cast<const float4*>(lf_smem + h * H + c * 8);
                    const float4 v1 = *reinterpret_cast<const float4*>(lf_smem + h * H + c * 8 + 4);
                    acc += xr[i][0] * v0.x + xr[i][1] * v0.y + xr[i][2] * v0.z + xr[i][3] * v0.w
                         + xr[i][4] * v1.x + xr[i][5] * v1.y + xr[i][6] * v1.z + xr[i][7] * v1.w;
                }
            }
            acc = wave_sum(acc);
            if (l == h) res = acc;
        }
        if (l < heads) {
            if (add_tok) res += add_tok[(size_t)b * L + j];
            if (add_bh) res += add_bh[b * heads + l];
            out[((size_t)b * heads + l) * L + j] = res;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- softmax fwd / bwd
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}
// one workgroup per (b, h) row of L scores: p = softmax(s) written over s; pd = dropout(p); sp = sum(pd)
__global__ __launch_bounds__(256) void lf_softmax_fwd_kernel(float* __restrict__ s_p, float* __restrict__ pd, float* __restrict__ sp,
                                                             int L, uint32_t thresh, float inv_keep, uint64_t seed) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    float* sr = s_p + row * L;
    if (L <= 256 * 16) {
        // the row in registers: ONE round of loads (the three passes over global memory were three dependent round trips per 256 elements,
        // 13-16 us for an 8-KB row in the forward chain of every Longformer layer).  Same per-thread element order, same reductions.
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int j = threadIdx.x + u * 256; v[u] = j < L ? sr[j] : -INFINITY; }
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < 16; ++u) mx = fmaxf(mx, v[u]);
        mx = block_reduce(mx, red, true);
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int j = threadIdx.x + u * 256; if (j < L) { v[u] = __expf(v[u] - mx); sum += v[u]; } }
        sum = block_reduce(sum, red, false);
        const float inv = 1.f / sum;
        float sd = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = threadIdx.x + u * 256;
            if (j < L) {
                const float p = v[u] * inv;
                sr[j] = p;
                const float q = (!thresh || drop_keep(seed, row * L + j, thresh)) ? p * inv_keep : 0.f;
                pd[row * L + j] = q;
                sd += q;
            }
        }
        sd = block_reduce(sd, red, false);
        if (threadIdx.x == 0) sp[row] = sd;
        return;
    }
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < L; j += 256) mx = fmaxf(mx, sr[j]);
    mx = block_reduce(mx, red, true);
    float sum = 0.f;
    for (int j = threadIdx.x; j < L; j += 256) sum += __expf(sr[j] - mx);
    sum = block_reduce(sum, red, false);
    const float inv = 1.f / sum;
    float sd = 0.f;
    for (int j = threadIdx.x; j < L; j += 256) {
        const float p = __expf(sr[j] - mx) * inv;
        sr[j] = p;
        const float q = (!thresh || drop_keep(seed, row * L + j, thresh)) ? p * inv_keep : 0.f;
        pd[row * L + j] = q;
        sd += q;
    }
    sd = block_reduce(sd, red, false);
    if (threadIdx.x == 0) sp[row] = sd;
}
// in: p, dpd (gradient w.r.t. the dropped probabilities).  out: ds over dpd, pd (recomputed dropped probabilities)
__global__ __launch_bounds__(256) void lf_softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dpd_ds, float* __restrict__ pd,
                                                             int L, uint32_t thresh, float inv_keep, uint64_t seed) {
    __shared__ float red[4];
    const size_t row = blockIdx.x;
    const float* pr = p + row * L;
    float* dr = dpd_ds + row * L;
    float dot = 0.f;
    for (int j = threadIdx.x; j < L; j += 256) {
        const float k = (!thresh || drop_keep(seed, row * L + j, thresh)) ? inv_keep : 0.f;
        dot += pr[j] * dr[j] * k;
    }
    dot = block_reduce(dot, red, false);
    for (int j = threadIdx.x; j < L; j += 256) {
        const float k = (!thresh || drop_keep(seed, row * L + j, thresh)) ? inv_keep : 0.f;
        const float pj = pr[j];
        pd[row * L + j] = pj * k;
        dr[j] = pj * (dr[j] * k - dot);
    }
}

// ---------------------------------------------------------------------------------------------------- weighted row sum
// grid (L/128, B), 256 threads; wave w owns heads w, w+4, w+8, w+12 and walks the block's 128 tokens; lane owns column
// chunks l, l+64.  part[b][seg][h][H] fp32, reduced over seg by lf_wsum_reduce_kernel (deterministic, no atomics)
#define LF_SEG 128
template <typename T>
__global__ __launch_bounds__(256) void lf_wsum_kernel(const T* __restrict__ x, const float* __restrict__ coef, float* __restrict__ part,
                                                      int L, int H, int heads, int seglen, int ldx) {
    __shared__ float cs[LF_MAXHEADS][LF_SEG];
    const int b = blockIdx.y, seg = blockIdx.x, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j0 = seg * seglen;
    for (int i = threadIdx.x; i < heads * seglen; i += 256) {
        const int h = i / seglen, t = i % seglen;
        cs[h][t] = coef[((size_t)b * heads + h) * L + j0 + t];
    }
    __syncthreads();
    const int nch = H / 8;
    float acc[4][LF_MAXCH][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < LF_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k][i][e] = 0.f;
    for (int t = 0; t < seglen; ++t) {
        const T* xp = x + ((size_t)b * L + j0 + t) * ldx;
        float xr[LF_MAXCH][8];
#pragma unroll
        for (int i = 0; i < LF_MAXCH; ++i) {
            const int c = l + 64 * i;
            if (c < nch) ld8<T>(xp + c * 8, xr[i]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xr[i][e] = 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h = w + 4 * k;
            const float cf = h < heads ? cs[h][t] : 0.f;
#pragma unroll
            for (int i = 0; i < LF_MAXCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[k][i][e] = fmaf(cf, xr[i][e], acc[k][i][e]);
        }
    }
    const int nseg = gridDim.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int h = w + 4 * k;
        if (h >= heads) continue;
        float* pp = part + (((size_t)b * nseg + seg) * heads + h) * H;
#pragma unroll
        for (int i = 0; i < LF_MAXCH; ++i) {
            const int c = l + 64 * i;
            if (c < nch) st8<float>(pp + c * 8, acc[k][i]);
        }
    }
}
__global__ void lf_wsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ y, int nseg, int n, int total) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;       // over B * n
    if (gid >= total) return;
    const int b = gid / n, i = gid % n;
    float s = 0.f;
    int g = 0;
    for (; g + 8 <= nseg; g += 8) {                         // eight partial rows in flight (same order of additions)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[((size_t)b * nseg + g + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; g < nseg; ++g) s += part[((size_t)b * nseg + g) * n + i];
    y[gid] = s;
}

// ---------------------------------------------------------------------------------------------------- dx update
// grid (L/64, B), 256 threads; vecA[b], vecB[b] ([heads][H] fp32 each) staged in LDS; wave handles 16 tokens
template <typename T>
__global__ __launch_bounds__(256) void lf_dx_update_kernel(T* __restrict__ dx, const float* __restrict__ coefA, const float* __restrict__ vecA,
                                                           const float* __restrict__ coefB, const float* __restrict__ vecB,
                                                           int L, int H, int heads, int ldx, int assign) {
    extern __shared__ __attribute__((aligned(16))) float lf_smem[];
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = heads * H;
    for (int i = threadIdx.x * 4; i < n; i += 256 * 4) {
        *reinterpret_cast<float4*>(lf_smem + i) = *reinterpret_cast<const float4*>(vecA + (size_t)b * n + i);
        *reinterpret_cast<float4*>(lf_smem + n + i) = *reinterpret_cast<const float4*>(vecB + (size_t)b * n + i);
    }
    __syncthreads();
    const int nch = H / 8;
    for (int t = 0; t < 16; ++t) {
        const int j = blockIdx.x * 64 + w * 16 + t;
        T* xp = dx + ((size_t)b * L + j) * ldx;
        float ca = 0.f, cb = 0.f;                            // lane h holds the two coefficients of head h
        if (l < heads) {
            ca = coefA[((size_t)b * heads + l) * L + j];
            cb = coefB[((size_t)b * heads + l) * L + j];
        }
#pragma unroll
        for (int i = 0; i < LF_MAXCH; ++i) {
            const int c = l + 64 * i;
            float v[8];
            if (c < nch) {
                if (assign) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                } else ld8<T>(xp + c * 8, v);
            }
            for (int h = 0; h < heads; ++h) {
                const float a_ = __shfl(ca, h, 64), b_ = __shfl(cb, h, 64);
                if (c < nch) {
                    const float* pa = lf_smem + h * H + c * 8;
                    const float* pb = lf_smem + n + h * H + c * 8;
                    const float4 a0 = *reinterpret_cast<const float4*>(pa), a1 = *reinterpret_cast<const float4*>(pa + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(pb), b1 = *reinterpret_cast<const float4*>(pb + 4);
                    v[0] += a_ * a0.x + b_ * b0.x; v[1] += a_ * a0.y + b_ * b0.y; v[2] += a_ * a0.z + b_ * b0.z; v[3] += a_ * a0.w + b_ * b0.w;
                    v[4] += a_ * a1.x + b_ * b1.x; v[5] += a_ * a1.y + b_ * b1.y; v[6] += a_ * a1.z + b_ * b1.z; v[7] += a_ * a1.w + b_ * b1.w;
                }
            }
            if (c < nch) st8<T>(xp + c * 8, v);
        }
    }
}


// ==================================================================================================== MFMA variants
// bf16 activations: the three O(L) passes as skinny v_mfma_f32_16x16x32_bf16 products with the head index padded to
// 16.  They are HBM-bound (one read of x; one read-modify-write of dx): the matrix cores only remove the VALU / LDS
// cost of the per-token dot products.  The fp32 parity mode keeps the scalar kernels above.
__device__ __forceinline__ void split_bf16(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    union { uint32_t u[4]; bf16x8 b; } h, l_;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h.u[i] = pack2bf(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - __uint_as_float(h.u[i] << 16), r1 = v[2 * i + 1] - __uint_as_float(h.u[i] & 0xffff0000u);
        l_.u[i] = pack2bf(r0, r1);
    }
    hi = h.b; lo = l_.b;
}

// scores: out[b, h, j] = vec[b, h, :] . x[b, j, :].  One wave = 16 tokens; A = x rows straight from global (16 B per lane
// per k-step), B = vec as bf16 hi + lo (two MFMAs per k-step keep the fp32 vector exact to 2^-17).  grid (L/64, B)
__global__ __launch_bounds__(256) void lf_rowvec_dot_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ vec,
                                                                 const float* __restrict__ add_tok, const float* __restrict__ add_bh,
                                                                 float* __restrict__ out, int L, int H, int heads, int ldx) {
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, i16 = l & 15;
    const int j0 = blockIdx.x * 64 + w * 16;
    const bf16_t* xp = x + ((size_t)b * L + j0 + i16) * ldx + g * 8;
    const float* vp = vec + ((size_t)b * heads + (i16 < heads ? i16 : 0)) * H + g * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nk = H / 32;
    // k-steps in batches of 8 with every load of a batch in flight together: the rolled loop paid one memory round trip per k-step
    // (24 in a row at H = 768: 20-38 us for 12 MB, on the critical path of every Longformer layer; round 4)
    for (int kb = 0; kb < nk; kb += 8) {
        bf16x8 fx[8];
        float v[8][8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (kb + u < nk) {
                fx[u] = *reinterpret_cast<const bf16x8*>(xp + (kb + u) * 32);
                ld8<float>(vp + (kb + u) * 32, v[u]);
            }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (kb + u < nk) {
                if (i16 >= heads) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
                }
                bf16x8 hi, lo;
                split_bf16(v[u], hi, lo);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[u], hi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[u], lo, acc, 0, 0, 0);
            }
    }
    if (i16 < heads) {                         // lane: head i16, tokens j0 + g*4 .. +4
        const int j = j0 + g * 4;
        float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (add_tok) {
            const float4 t = *reinterpret_cast<const float4*>(add_tok + (size_t)b * L + j);
            o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
        }
        if (add_bh) { const float c = add_bh[b * heads + i16]; o.x += c; o.y += c; o.z += c; o.w += c; }
        *reinterpret_cast<float4*>(out + ((size_t)b * heads + i16) * L + j) = o;
    }
}

// weighted row sum: part[b][chunk][h][slab*64 + c] = sum_{j in chunk} coef[b,h,j] x[b,j,slab*64+c].  One wave = one
// [64 tokens][64 cols] tile of x staged in a wave-private LDS region by DMA; the product is the attention P.V step with
// the 16 (padded) heads in place of the queries.  grid (ceil(L/256), H/64, B), 4 waves = 4 consecutive token chunks
__global__ __launch_bounds__(256) void lf_wsum_mfma_kernel(const bf16_t* __restrict__ x, const float* __restrict__ coef,
                                                           float* __restrict__ part, int L, int H, int heads, int ldx) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 8192];
    const int b = blockIdx.z, slab = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, i16 = l & 15;
    const int chunk = blockIdx.x * 4 + w, j0 = chunk * 64;
    if (j0 >= L) return;                                   // no block-level barrier below: a wave may leave early
    char* tile = smem + w * 8192;
    at_stage<1>(x + ((size_t)b * L + j0) * ldx + slab * 64, ldx, tile, 0, l);
    f32x4 s[4];
    const float* cp = coef + ((size_t)b * heads + (i16 < heads ? i16 : 0)) * L + j0 + g * 4;
#pragma unroll
    for (int fc = 0; fc < 4; ++fc) {
        const float4 c = *reinterpret_cast<const float4*>(cp + fc * 16);
        s[fc] = i16 < heads ? (f32x4){c.x, c.y, c.z, c.w} : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
        const float v8[8] = {s[2 * kp][0], s[2 * kp][1], s[2 * kp][2], s[2 * kp][3], s[2 * kp + 1][0], s[2 * kp + 1][1], s[2 * kp + 1][2], s[2 * kp + 1][3]};
        bf16x8 fp, fp_lo;                                   // coefficients as bf16 hi + lo: exact to 2^-17
        split_bf16(v8, fp, fp_lo);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const bf16x8 fv = at_frag_tr(tile, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4, d * 16, l);
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fp, o[d], 0, 0, 0);
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fp_lo, o[d], 0, 0, 0);
        }
    }
    if (i16 < heads) {
        const int nchunk = L / 64;
        float* pp = part + (((size_t)b * nchunk + chunk) * heads + i16) * H + slab * 64 + g * 4;
#pragma unroll
        for (int d = 0; d < 4; ++d) *reinterpret_cast<float4*>(pp + d * 16) = make_float4(o[d][0], o[d][1], o[d][2], o[d][3]);
    }
}

// vt[b][c][slot] bf16, slot = which*16 + head (which 0 = vecA, 1 = vecB; heads padded to 16 with zeros)
__global__ void lf_vt_prep_kernel(const float* __restrict__ vecA, const float* __restrict__ vecB, bf16_t* __restrict__ vt,
                                  int H, int heads, int total) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;          // over B * H * 32, slot fastest
    if (gid >= total) return;
    const int slot = gid & 31, c = (gid >> 5) % H, b = (gid >> 5) / H;
    const int which = slot >> 4, h = slot & 15;
    float v = 0.f;
    if (h < heads) v = (which ? vecB : vecA)[((size_t)b * heads + h) * H + c];
    vt[gid] = f2bf(v);
}
// dx[b, j, :] += [coefA | coefB][:, j]^T . [vecA ; vecB]   -- one k-step of 32 slots per 16x16 output tile.
// Operands swapped (A = vt rows = columns of dx, B = coefficients of 16 tokens) so that a lane owns one token and 4
// consecutive columns: 8-byte read-modify-write.  One wave = 32 tokens; grid (L/128, B)
__global__ __launch_bounds__(256) void lf_dx_update_mfma_kernel(bf16_t* __restrict__ dx, const float* __restrict__ coefA,
                                                                const float* __restrict__ coefB, const bf16_t* __restrict__ vt,
                                                                int L, int H, int heads, int ldx, int assign, const float* __restrict__ trow) {
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, i16 = l & 15;
    const int j0 = blockIdx.x * 128 + w * 32;
    if (j0 >= L) return;
    bf16x8 fc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float* cp = ((g >> 1) ? coefB : coefA) + (size_t)b * heads * L + j0 + t * 16 + i16;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int h = (g & 1) * 8 + e;
            v[e] = h < heads ? cp[(size_t)h * L] : 0.f;
        }
        union { uint32_t u[4]; bf16x8 q; } pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk.u[i] = pack2bf(v[2 * i], v[2 * i + 1]);
        fc[t] = pk.q;
    }
    const bf16_t* vp = vt + ((size_t)b * H + i16) * 32 + g * 8;
    // the 16-column tiles of a row are dealt over gridDim.z workgroups and walked four at a time with their loads issued together: one wave
    // walking all H / 16 = 48 tiles as dependent load -> MFMA -> store trips took 44 us whatever M was (64 workgroups at M = 8192)
    const int nct = H / 16, per = (nct + gridDim.z - 1) / gridDim.z;
    const int ct0 = blockIdx.z * per, ct1 = min(nct, ct0 + per);
    for (int cb = ct0; cb < ct1; cb += 4) {
        bf16x8 fv[4];
        uint2 old[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ct = min(cb + u, ct1 - 1);
            fv[u] = *reinterpret_cast<const bf16x8*>(vp + (size_t)ct * 16 * 32);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                old[u][t] = make_uint2(0u, 0u);
                if (!assign) old[u][t] = *reinterpret_cast<const uint2*>(dx + ((size_t)b * L + j0 + t * 16 + i16) * ldx + ct * 16 + g * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (cb + u >= ct1) break;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[u], fc[t], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (trow && j0 + t * 16 + i16 == 0) {            // token 0 of the sequence also takes the row vector trow[b, :]
                    const float4 tr = *reinterpret_cast<const float4*>(trow + (size_t)b * H + (cb + u) * 16 + g * 4);
                    d[0] += tr.x; d[1] += tr.y; d[2] += tr.z; d[3] += tr.w;
                }
                uint2 nw;
                nw.x = pack2bf(__uint_as_float(old[u][t].x << 16) + d[0], __uint_as_float(old[u][t].x & 0xffff0000u) + d[1]);
                nw.y = pack2bf(__uint_as_float(old[u][t].y << 16) + d[2], __uint_as_float(old[u][t].y & 0xffff0000u) + d[3]);
                *reinterpret_cast<uint2*>(dx + ((size_t)b * L + j0 + t * 16 + i16) * ldx + (cb + u) * 16 + g * 4) = nw;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- launchers
static int lf_check(int B, int L, int H, int heads) {
    if (B <= 0 || L <= 0 || (L % 64) || (H % 8) || H > 8 * 64 * LF_MAXCH || heads <= 0 || heads > LF_MAXHEADS) return AMDSEG_ERR_SHAPE;
    if ((size_t)heads * H * 4 * 2 > 160 * 1024) return AMDSEG_ERR_SHAPE;
    return AMDSEG_OK;
}
static inline void lf_drop_params(float p, uint32_t& thresh, float& inv_keep) {
    if (p <= 0.f) { thresh = 0; inv_keep = 1.f; return; }
    double t = (double)p * 4294967296.0;
    thresh = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
    if (thresh == 0) thresh = 1;
    inv_keep = (float)(4294967296.0 / (4294967296.0 - (double)thresh));
}

int amdseg_lf_rowvec_dot_impl(const void* x, const float* vec, const float* add_tok, const float* add_bh, float* out, int B, int L,
                              int H, int heads, int dtype, int ldx, hipStream_t s) {
    if (!x || !vec || !out || ldx < H || (ldx % 8)) return AMDSEG_ERR_ARG;
    int rc = lf_check(B, L, H, heads);
    if (rc) return rc;
    if (L % 64) return AMDSEG_ERR_SHAPE;
    const size_t lds = (size_t)heads * H * 4;
    if (dtype == AMDSEG_BF16 && (H % 32) == 0) {
        hipLaunchKernelGGL(lf_rowvec_dot_mfma_kernel, dim3(L / 64, B), dim3(256), 0, s, (const bf16_t*)x, vec, add_tok, add_bh, out, L, H, heads, ldx);
    } else if (dtype == AMDSEG_BF16) {
        (void)hipFuncSetAttribute((const void*)lf_rowvec_dot_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lf_rowvec_dot_kernel<bf16_t>, dim3(L / 64, B), dim3(256), lds, s, (const bf16_t*)x, vec, add_tok, add_bh, out, L, H, heads, ldx);
    } else {
        (void)hipFuncSetAttribute((const void*)lf_rowvec_dot_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lf_rowvec_dot_kernel<float>, dim3(L / 64, B), dim3(256), lds, s, (const float*)x, vec, add_tok, add_bh, out, L, H, heads, ldx);
    }
    return amdseg_launch_status();
}

int amdseg_lf_softmax_fwd_impl(float* s_inout_p, float* pd, float* sp, int rows, int L, float p, uint64_t seed, hipStream_t s) {
    if (!s_inout_p || !pd || !sp) return AMDSEG_ERR_ARG;
    if (rows <= 0 || L <= 0 || p < 0.f || p >= 1.f) return AMDSEG_ERR_SHAPE;
    uint32_t th; float ik; lf_drop_params(p, th, ik);
    hipLaunchKernelGGL(lf_softmax_fwd_kernel, dim3(rows), dim3(256), 0, s, s_inout_p, pd, sp, L, th, ik, seed);
    return amdseg_launch_status();
}

int amdseg_lf_softmax_bwd_impl(const float* p_saved, float* dpd_inout_ds, float* pd, int rows, int L, float p, uint64_t seed,
                               hipStream_t s) {
    if (!p_saved || !dpd_inout_ds || !pd) return AMDSEG_ERR_ARG;
    if (rows <= 0 || L <= 0 || p < 0.f || p >= 1.f) return AMDSEG_ERR_SHAPE;
    uint32_t th; float ik; lf_drop_params(p, th, ik);
    hipLaunchKernelGGL(lf_softmax_bwd_kernel, dim3(rows), dim3(256), 0, s, p_saved, dpd_inout_ds, pd, L, th, ik, seed);
    return amdseg_launch_status();
}

int amdseg_lf_wsum_impl(const void* x, const float* coef, float* partials, float* y, int B, int L, int H, int heads, int dtype,
                        int ldx, hipStream_t s) {
    if (!x || !coef || !partials || !y || ldx < H || (ldx % 8)) return AMDSEG_ERR_ARG;
    int rc = lf_check(B, L, H, heads);
    if (rc) return rc;
    int seglen = (L % LF_SEG) ? 64 : LF_SEG;
    int nseg = L / seglen;
    if (dtype == AMDSEG_BF16 && (H % 64) == 0) {
        seglen = 64; nseg = L / 64;
        hipLaunchKernelGGL(lf_wsum_mfma_kernel, dim3((nseg + 3) / 4, H / 64, B), dim3(256), 0, s, (const bf16_t*)x, coef, partials, L, H, heads, ldx);
    } else if (dtype == AMDSEG_BF16)
        hipLaunchKernelGGL(lf_wsum_kernel<bf16_t>, dim3(nseg, B), dim3(256), 0, s, (const bf16_t*)x, coef, partials, L, H, heads, seglen, ldx);
    else
        hipLaunchKernelGGL(lf_wsum_kernel<float>, dim3(nseg, B), dim3(256), 0, s, (const float*)x, coef, partials, L, H, heads, seglen, ldx);
    const int n = heads * H, total = B * n;
    hipLaunchKernelGGL(lf_wsum_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, partials, y, nseg, n, total);
    return amdseg_launch_status();
}

int amdseg_lf_dx_update_impl(void* dx, const float* coefA, const float* vecA, const float* coefB, const float* vecB, void* vt_ws,
                             int B, int L, int H, int heads, int dtype, int ldx, int assign, hipStream_t s) {
    if (!dx || !coefA || !vecA || !coefB || !vecB || ldx < H || (ldx % 8)) return AMDSEG_ERR_ARG;
    int rc = lf_check(B, L, H, heads);
    if (rc) return rc;
    if (L % 64) return AMDSEG_ERR_SHAPE;
    if (dtype == AMDSEG_BF16 && vt_ws && (H % 16) == 0) {
        const int total = B * H * 32;
        hipLaunchKernelGGL(lf_vt_prep_kernel, dim3((total + 255) / 256), dim3(256), 0, s, vecA, vecB, (bf16_t*)vt_ws, H, heads, total);
        const int wgs = ((L + 127) / 128) * B, nz = wgs >= 1024 ? 2 : wgs >= 256 ? 4 : 8;      // column groups: enough workgroups to hide the row loads
        hipLaunchKernelGGL(lf_dx_update_mfma_kernel, dim3((L + 127) / 128, B, nz), dim3(256), 0, s, (bf16_t*)dx, coefA, coefB, (const bf16_t*)vt_ws, L, H, heads, ldx, assign,
                           (const float*)nullptr);
        return amdseg_launch_status();
    }
    const size_t lds = (size_t)heads * H * 4 * 2;
    if (dtype == AMDSEG_BF16) {
        (void)hipFuncSetAttribute((const void*)lf_dx_update_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lf_dx_update_kernel<bf16_t>, dim3(L / 64, B), dim3(256), lds, s, (bf16_t*)dx, coefA, vecA, coefB, vecB, L, H, heads, ldx, assign);
    } else {
        (void)hipFuncSetAttribute((const void*)lf_dx_update_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lf_dx_update_kernel<float>, dim3(L / 64, B), dim3(256), lds, s, (float*)dx, coefA, vecA, coefB, vecB, L, H, heads, ldx, assign);
    }
    return amdseg_launch_status();
}

// amdseg_lf_dx_update in two calls (bf16): _prep packs the two vector sets into the MFMA operand image vt_ws [B, H, 32] -- everything that does
// not touch dx, so it can run early and elsewhere -- and _apply is the read-modify-write pass over dx alone; with trow [B, H] it also adds
// trow[b, :] to row b*L (the global token's own row: amdseg_lf_global_bwd_dx).
int amdseg_lf_dx_prep_impl(const float* vecA, const float* vecB, void* vt_ws, int B, int L, int H, int heads, hipStream_t s) {
    if (!vecA || !vecB || !vt_ws) return AMDSEG_ERR_ARG;
    int rc = lf_check(B, L, H, heads);
    if (rc) return rc;
    if (H % 16) return AMDSEG_ERR_SHAPE;
    const int total = B * H * 32;
    hipLaunchKernelGGL(lf_vt_prep_kernel, dim3((total + 255) / 256), dim3(256), 0, s, vecA, vecB, (bf16_t*)vt_ws, H, heads, total);
    return amdseg_launch_status();
}
int amdseg_lf_dx_apply_impl(void* dx, int ldx, const float* coefA, const float* coefB, const void* vt_ws, const float* trow, int B, int L, int H,
                            int heads, hipStream_t s) {
    if (!dx || !coefA || !coefB || !vt_ws || ldx < H || (ldx % 8)) return AMDSEG_ERR_ARG;
    int rc = lf_check(B, L, H, heads);
    if (rc) return rc;
    if (H % 16) return AMDSEG_ERR_SHAPE;
    const int wgs = ((L + 127) / 128) * B, nz = wgs >= 1024 ? 2 : wgs >= 256 ? 4 : 8;
    hipLaunchKernelGGL(lf_dx_update_mfma_kernel, dim3((L + 127) / 128, B, nz), dim3(256), 0, s, (bf16_t*)dx, coefA, coefB, (const bf16_t*)vt_ws, L, H, heads, ldx, 0, trow);
    return amdseg_launch_status();
}
