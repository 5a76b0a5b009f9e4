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

#define LF_MAXCH 2          // 8-element column chunks per lane: H <= 1024
#define LF_MAXHEADS 16

// ---------------------------------------------------------------------------------------------------- rowvec dot
// grid (L/64, B), 256 threads; wave w handles tokens blk*64 + w*16 .. +16; vec[b] staged in LDS as fp32 [heads][H]
template <typename T>
__global__ __launch_bounds__(256) void lf_rowvec_dot_kernel(const T* __restrict__ x, const float* __restrict__ vec,
                                                            const float* __restrict__ add_tok, const float* __restrict__ add_bh,
                                                            float* __restrict__ out, int L, int H, int heads) {
    extern __shared__ __attribute__((aligned(16))) float lf_smem[];
    const int b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const float* vb = vec + (size_t)b * heads * H;
    for (int i = threadIdx.x * 4; i < heads * H; i += 256 * 4)
        *reinterpret_cast<float4*>(lf_smem + i) = *reinterpret_cast<const float4*>(vb + i);
    __syncthreads();
    const int nch = H / 8;
    for (int t = 0; t < 16; ++t) {
        const int j = blockIdx.x * 64 + w * 16 + t;
        const T* xp = x + ((size_t)b * L + j) * H;
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
                                                      int L, int H, int heads, int seglen) {
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
        const T* xp = x + ((size_t)b * L + j0 + t) * H;
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
    for (int g = 0; g < nseg; ++g) s += part[((size_t)b * nseg + g) * n + i];
    y[gid] = s;
}

// ---------------------------------------------------------------------------------------------------- dx update
// grid (L/64, B), 256 threads; vecA[b], vecB[b] ([heads][H] fp32 each) staged in LDS; wave handles 16 tokens
template <typename T>
__global__ __launch_bounds__(256) void lf_dx_update_kernel(T* __restrict__ dx, const float* __restrict__ coefA, const float* __restrict__ vecA,
                                                           const float* __restrict__ coefB, const float* __restrict__ vecB,
                                                           int L, int H, int heads) {
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
        T* xp = dx + ((size_t)b * L + j) * H;
        float ca = 0.f, cb = 0.f;                            // lane h holds the two coefficients of head h
        if (l < heads) {
            ca = coefA[((size_t)b * heads + l) * L + j];
            cb = coefB[((size_t)b * heads + l) * L + j];
        }
#pragma unroll
        for (int i = 0; i < LF_MAXCH; ++i) {
            const int c = l + 64 * i;
            float v[8];
            if (c < nch) ld8<T>(xp + c * 8, v);
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
                              int H, int heads, int dtype, hipStream_t s) {
    if (!x || !vec || !out) return AMDSEG_ERR_ARG;
    int rc = lf_check(B, L, H, heads);
    if (rc) return rc;
    if (L % 64) return AMDSEG_ERR_SHAPE;
    const size_t lds = (size_t)heads * H * 4;
    if (dtype == AMDSEG_BF16) {
        (void)hipFuncSetAttribute((const void*)lf_rowvec_dot_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lf_rowvec_dot_kernel<bf16_t>, dim3(L / 64, B), dim3(256), lds, s, (const bf16_t*)x, vec, add_tok, add_bh, out, L, H, heads);
    } else {
        (void)hipFuncSetAttribute((const void*)lf_rowvec_dot_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lf_rowvec_dot_kernel<float>, dim3(L / 64, B), dim3(256), lds, s, (const float*)x, vec, add_tok, add_bh, out, L, H, heads);
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
                        hipStream_t s) {
    if (!x || !coef || !partials || !y) return AMDSEG_ERR_ARG;
    int rc = lf_check(B, L, H, heads);
    if (rc) return rc;
    const int seglen = (L % LF_SEG) ? 64 : LF_SEG;
    const int nseg = L / seglen;
    if (dtype == AMDSEG_BF16)
        hipLaunchKernelGGL(lf_wsum_kernel<bf16_t>, dim3(nseg, B), dim3(256), 0, s, (const bf16_t*)x, coef, partials, L, H, heads, seglen);
    else
        hipLaunchKernelGGL(lf_wsum_kernel<float>, dim3(nseg, B), dim3(256), 0, s, (const float*)x, coef, partials, L, H, heads, seglen);
    const int n = heads * H, total = B * n;
    hipLaunchKernelGGL(lf_wsum_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, partials, y, nseg, n, total);
    return amdseg_launch_status();
}

int amdseg_lf_dx_update_impl(void* dx, const float* coefA, const float* vecA, const float* coefB, const float* vecB, int B, int L,
                             int H, int heads, int dtype, hipStream_t s) {
    if (!dx || !coefA || !vecA || !coefB || !vecB) return AMDSEG_ERR_ARG;
    int rc = lf_check(B, L, H, heads);
    if (rc) return rc;
    if (L % 64) return AMDSEG_ERR_SHAPE;
    const size_t lds = (size_t)heads * H * 4 * 2;
    if (dtype == AMDSEG_BF16) {
        (void)hipFuncSetAttribute((const void*)lf_dx_update_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lf_dx_update_kernel<bf16_t>, dim3(L / 64, B), dim3(256), lds, s, (bf16_t*)dx, coefA, vecA, coefB, vecB, L, H, heads);
    } else {
        (void)hipFuncSetAttribute((const void*)lf_dx_update_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(lf_dx_update_kernel<float>, dim3(L / 64, B), dim3(256), lds, s, (float*)dx, coefA, vecA, coefB, vecB, L, H, heads);
    }
    return amdseg_launch_status();
}
