// "parity" precision mode: fp32 activations, every contraction at fp32-grade accuracy on the bf16 MFMA pipes.
//
// The reference trains and predicts in fp32 (run_finetune.sh:61-96 has no --fp16 / --bf16); the north star asks for logits within
// 1e-3 of it.  A single bf16 rounding of the operands misses that by two orders of magnitude (hard part 1).  Here every
// fp32 operand x is split into two bf16 numbers, hi = bf16(x) and lo = bf16(x - hi) (x = hi + lo to 2^-17 relative), and a product
// A . B^T is evaluated as  Ahi.Bhi + Ahi.Blo + Alo.Bhi  (the dropped Alo.Blo term is 2^-16 relative) -- as ONE bf16 GEMM over a
// three times longer K:  A' = [Ahi | Ahi | Alo]  (M x 3K),  B' = [Bhi | Blo | Bhi]  (N x 3K), fp32 accumulation inside the MFMA and
// an fp32 result.  So the projection / FFN GEMMs of the forward pass and the dgrads of backward run on the SAME deep-pipeline kernel
// as the fast path (gemm_dp.hip) at K' = 3K, the weight gradients on the grouped TN kernel as three accumulating launches over the
// hi / lo column blocks; what is new are the HBM-bound producers of the split images below and an fp32 attention (forward with
// saved log-sum-exp + dropout, backward) on the vector ALUs with LDS-staged 64-row chunks.  The exact v_mfma_f32_32x32x2_f32 kernels
// of gemm_f32.hip (1/16 of the bf16 rate) remain as the bit-faithful inference reference ("fp32" precision).
#include "common.h"
#include "amdseg_internal.h"

// ------------------------------------------------------------------------------------------------ split images
// out row (3K wide):  order 0 (activation / gradient operand A') = [hi | hi | lo];  order 1 (weight operand B') = [hi | lo | hi]
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
    hi.x = pack2bf(v.x, v.y); hi.y = pack2bf(v.z, v.w);
    const float r0 = v.x - __uint_as_float(hi.x << 16), r1 = v.y - __uint_as_float(hi.x & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(hi.y << 16), r3 = v.w - __uint_as_float(hi.y & 0xffff0000u);
    lo.x = pack2bf(r0, r1); lo.y = pack2bf(r2, r3);
}
__device__ __forceinline__ void split_store(bf16_t* orow, int K, int c, const float4 v, int order) {
    uint2 hi, lo;
    split4(v, hi, lo);
    *reinterpret_cast<uint2*>(orow + c) = hi;
    *reinterpret_cast<uint2*>(orow + K + c) = order ? lo : hi;
    *reinterpret_cast<uint2*>(orow + 2 * K + c) = order ? hi : lo;
}

__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int ld, bf16_t* __restrict__ out, int M, int K, int order) {
    const int k4 = K >> 2;
    const size_t total = (size_t)M * k4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / k4;
        const int c = (int)(i - r * k4) * 4;
        split_store(out + r * 3 * (size_t)K, K, c, *reinterpret_cast<const float4*>(x + r * ld + c), order);
    }
}

// W [N, K] fp32 -> out [K, 3N] = [Wt_hi | Wt_lo | Wt_hi] (the B' operand of a dgrad  dX = dY . W  written as an NT product)
__global__ __launch_bounds__(256) void split3_transpose_kernel(const float* __restrict__ W, bf16_t* __restrict__ out, int N, int K) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = W[(size_t)(n0 + ty + 8 * i) * K + k0 + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, n = n0 + tx;
        const float v = tile[tx][ty + 8 * i];
        const bf16_t hi = f2bf(v), lo = f2bf(v - bf2f(hi));
        bf16_t* o = out + (size_t)k * 3 * N;
        o[n] = hi; o[N + n] = lo; o[2 * N + n] = hi;
    }
}

// every weight matrix of the model in ONE launch: W_i [N, K] fp32 -> out_i [N, 3K] = [hi | lo | hi] (forward B' operand) and
// out_t_i [K, 3N] = [Wt_hi | Wt_lo | Wt_hi] (dgrad B' operand).  (One split3 + one split3_transpose launch per matrix were ~100 launches of
// ~6 us behind every optimiser step.)
#define SW_MAXB 64
struct SplitWeightsBatch {
    const float* W[SW_MAXB]; bf16_t* out[SW_MAXB]; bf16_t* out_t[SW_MAXB];
    int N[SW_MAXB], K[SW_MAXB], tile0[SW_MAXB + 1];
    int n;
};
__global__ __launch_bounds__(256) void split3_weights_batched_kernel(SplitWeightsBatch b) {
    __shared__ bf16_t th[64][66], tl[64][66];
    int m = 0;
    while (m + 1 < b.n && (int)blockIdx.x >= b.tile0[m + 1]) ++m;
    const float* W = b.W[m]; bf16_t* out = b.out[m]; bf16_t* out_t = b.out_t[m];
    const int N = b.N[m], K = b.K[m], t = blockIdx.x - b.tile0[m], kt = K / 64;
    const int n0 = (t / kt) * 64, k0 = (t % kt) * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty * 4 + i;
        const float4 v = *reinterpret_cast<const float4*>(W + (size_t)(n0 + r) * K + k0 + tx * 4);
        uint2 hi, lo;
        split4(v, hi, lo);
        th[r][tx * 4 + 0] = (bf16_t)(hi.x & 0xffffu); th[r][tx * 4 + 1] = (bf16_t)(hi.x >> 16);
        th[r][tx * 4 + 2] = (bf16_t)(hi.y & 0xffffu); th[r][tx * 4 + 3] = (bf16_t)(hi.y >> 16);
        tl[r][tx * 4 + 0] = (bf16_t)(lo.x & 0xffffu); tl[r][tx * 4 + 1] = (bf16_t)(lo.x >> 16);
        tl[r][tx * 4 + 2] = (bf16_t)(lo.y & 0xffffu); tl[r][tx * 4 + 3] = (bf16_t)(lo.y >> 16);
        if (out) {
            bf16_t* o = out + (size_t)(n0 + r) * 3 * K + k0 + tx * 4;
            *reinterpret_cast<uint2*>(o) = hi; *reinterpret_cast<uint2*>(o + K) = lo; *reinterpret_cast<uint2*>(o + 2 * K) = hi;
        }
    }
    __syncthreads();
    if (out_t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = ty * 4 + i;
            uint2 hi, lo;
            hi.x = (uint32_t)th[tx * 4 + 0][kr] | ((uint32_t)th[tx * 4 + 1][kr] << 16);
            hi.y = (uint32_t)th[tx * 4 + 2][kr] | ((uint32_t)th[tx * 4 + 3][kr] << 16);
            lo.x = (uint32_t)tl[tx * 4 + 0][kr] | ((uint32_t)tl[tx * 4 + 1][kr] << 16);
            lo.y = (uint32_t)tl[tx * 4 + 2][kr] | ((uint32_t)tl[tx * 4 + 3][kr] << 16);
            bf16_t* o = out_t + (size_t)(k0 + kr) * 3 * N + n0 + tx * 4;
            *reinterpret_cast<uint2*>(o) = hi; *reinterpret_cast<uint2*>(o + N) = lo; *reinterpret_cast<uint2*>(o + 2 * N) = hi;
        }
    }
}

// h = gelu(u) written directly as the split image [M, 3I] (the fp32 h itself is never needed: W2's GEMM and its weight gradient
// read the image);  backward: du = du * gelu'(u) in place (fp32, for the bias gradient) + its split image
__device__ __forceinline__ float gelu_exact(float x, int act) {
    return act ? 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x))) : gelu_erf(x);
}
__device__ __forceinline__ float gelu_exact_grad(float x, int act) {
    if (!act) return gelu_erf_grad(x);
    const float x2 = x * x, t = tanhf(0.7978845608028654f * (x + 0.044715f * x * x2));
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 0.134145f * x2);
}
__global__ __launch_bounds__(256) void gelu_fwd_split_kernel(const float* __restrict__ u, bf16_t* __restrict__ hs, int M, int I, int act) {
    const int k4 = I >> 2;
    const size_t total = (size_t)M * k4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / k4;
        const int c = (int)(i - r * k4) * 4;
        float4 v = *reinterpret_cast<const float4*>(u + r * I + c);
        v.x = gelu_exact(v.x, act); v.y = gelu_exact(v.y, act); v.z = gelu_exact(v.z, act); v.w = gelu_exact(v.w, act);
        split_store(hs + r * 3 * (size_t)I, I, c, v, 0);
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_split_kernel(float* __restrict__ du, const float* __restrict__ u, bf16_t* __restrict__ dus,
                                                             int M, int I, int act) {
    const int k4 = I >> 2;
    const size_t total = (size_t)M * k4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / k4;
        const int c = (int)(i - r * k4) * 4;
        float4 g = *reinterpret_cast<const float4*>(du + r * I + c);
        const float4 x = *reinterpret_cast<const float4*>(u + r * I + c);
        g.x *= gelu_exact_grad(x.x, act); g.y *= gelu_exact_grad(x.y, act); g.z *= gelu_exact_grad(x.z, act); g.w *= gelu_exact_grad(x.w, act);
        *reinterpret_cast<float4*>(du + r * I + c) = g;
        split_store(dus + r * 3 * (size_t)I, I, c, g, 0);
    }
}
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(y)[i];
        const float4 b = reinterpret_cast<const float4*>(x)[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        reinterpret_cast<float4*>(y)[i] = a;
    }
}

static inline unsigned ew_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b == 0 ? 1 : b));
}

int amdseg_split3_impl(const float* x, int ld, void* out, int M, int K, int order, hipStream_t s) {
    if (!x || !out) return AMDSEG_ERR_ARG;
    if (M <= 0 || K <= 0 || (K % 4) || (ld % 4)) return AMDSEG_ERR_SHAPE;
    hipLaunchKernelGGL(split3_kernel, dim3(ew_grid((size_t)M * K / 4)), dim3(256), 0, s, x, ld, (bf16_t*)out, M, K, order);
    return amdseg_launch_status();
}
int amdseg_split3_transpose_impl(const float* W, void* out, int N, int K, hipStream_t s) {
    if (!W || !out) return AMDSEG_ERR_ARG;
    if (N <= 0 || K <= 0 || (N % 32) || (K % 32)) return AMDSEG_ERR_SHAPE;
    hipLaunchKernelGGL(split3_transpose_kernel, dim3(K / 32, N / 32), dim3(256), 0, s, W, (bf16_t*)out, N, K);
    return amdseg_launch_status();
}
int amdseg_split3_weights_batched_impl(int n, const float* const* W, void* const* out, void* const* out_t, const int* N, const int* K, hipStream_t s) {
    if (n <= 0 || !W || !N || !K || (!out && !out_t)) return AMDSEG_ERR_ARG;
    for (int base = 0; base < n; base += SW_MAXB) {
        SplitWeightsBatch b = {};
        b.n = n - base < SW_MAXB ? n - base : SW_MAXB;
        int tiles = 0;
        for (int i = 0; i < b.n; ++i) {
            const int j = base + i;
            if (!W[j] || N[j] <= 0 || K[j] <= 0 || (N[j] % 64) || (K[j] % 64)) return AMDSEG_ERR_SHAPE;
            b.W[i] = W[j]; b.out[i] = out ? (bf16_t*)out[j] : nullptr; b.out_t[i] = out_t ? (bf16_t*)out_t[j] : nullptr;
            b.N[i] = N[j]; b.K[i] = K[j]; b.tile0[i] = tiles;
            tiles += (N[j] / 64) * (K[j] / 64);
        }
        b.tile0[b.n] = tiles;
        hipLaunchKernelGGL(split3_weights_batched_kernel, dim3(tiles), dim3(256), 0, s, b);
    }
    return amdseg_launch_status();
}
int amdseg_gelu_fwd_split_impl(const float* u, void* hs, int M, int I, int act, hipStream_t s) {
    if (!u || !hs) return AMDSEG_ERR_ARG;
    if (M <= 0 || I <= 0 || (I % 4)) return AMDSEG_ERR_SHAPE;
    hipLaunchKernelGGL(gelu_fwd_split_kernel, dim3(ew_grid((size_t)M * I / 4)), dim3(256), 0, s, u, (bf16_t*)hs, M, I, act);
    return amdseg_launch_status();
}
int amdseg_gelu_bwd_split_impl(float* du, const float* u, void* dus, int M, int I, int act, hipStream_t s) {
    if (!du || !u || !dus) return AMDSEG_ERR_ARG;
    if (M <= 0 || I <= 0 || (I % 4)) return AMDSEG_ERR_SHAPE;
    hipLaunchKernelGGL(gelu_bwd_split_kernel, dim3(ew_grid((size_t)M * I / 4)), dim3(256), 0, s, du, u, (bf16_t*)dus, M, I, act);
    return amdseg_launch_status();
}
int amdseg_add_inplace_impl(float* y, const float* x, size_t n, hipStream_t s) {
    if (!y || !x) return AMDSEG_ERR_ARG;
    if (n == 0 || (n % 4)) return AMDSEG_ERR_SHAPE;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, s, y, x, n / 4);
    return amdseg_launch_status();
}

// ------------------------------------------------------------------------------------------------ fp32 attention, forward + backward
// [hf] models/bert/modeling_bert.py:111-136 (eager_attention_forward): scores = q k^T / 8 + mask, softmax in fp32, dropout on the
// probabilities, context = P V; backward by the usual recomputation from the saved log-sum-exp.  Vector-ALU kernels (the fp32 MFMA
// would bring 1/16 of the bf16 rate; these products are 10 % of the layer's FLOPs): a workgroup = 4 waves = 4 consecutive query (or
// key) rows of one (batch, head); the other side is streamed through LDS in 64-row chunks [64][65] (row stride 65 floats: a lane
// reading ITS row and all lanes reading ONE row are both conflict-free).  "lane = row of the chunk" for the dot products (the wave's
// own row broadcast from a VGPR with v_readlane), "lane = feature" for the weighted sums.  Dropout: stateless hash of
// (seed, flat element index), keep-scale 1 / (1 - p) as torch.
struct PAttnArgs {
    const float* qkv; const float* mask_bias; float* ctx; float* lse;
    const float* dctx; float* delta; float* dqkv;
    int B, L, heads;
    float scale, inv_keep; uint32_t thresh; uint64_t seed;
    // trailing padding, as AttnArgs in attention.hip (pattn2 kernels): keys at positions >= kend[b] are masked (exp() == 0 exactly in fp32
    // too) -> their chunks are not visited; seq_order = dispatch order (longest first); *qguard == 0: the dctx rows there are exact zeros
    const int* kend; const int* seq_order; const int* qguard;
};
#define PA_LD 65
__device__ __forceinline__ int pa2_visible_chunks(const PAttnArgs& a, int b, int ns) {
    if (!a.kend) return ns;
    const int ke = a.kend[b];
    return ke > 0 ? min(ns, (ke + 63) >> 6) : ns;
}

__device__ __forceinline__ float pa_keep(const PAttnArgs& a, size_t row_bhq, int key) {
    if (a.thresh == 0) return 1.0f;
    return drop_keep(a.seed, row_bhq * (size_t)a.L + key, a.thresh) ? a.inv_keep : 0.0f;
}

// ------------------------------------------------------------------------------------------------ fp32 attention on the fp32 MFMA
// The same arithmetic as the vector-ALU kernels above, as flash-style kernels on v_mfma_f32_16x16x4_f32 (exact fp32 products and
// accumulation, 1/16 of the bf16 MFMA rate = the fp32 vector rate, but one instruction does 1024 multiply-adds instead of 64): a
// workgroup = 4 waves x 16 query (key) rows, the other side streamed through LDS in 64-row chunks [64][68] (row stride 68 floats: both
// fragment access patterns below hit 64 distinct banks).  TRANSPOSED orientation as in attention.hip: S^T = K Q^T puts the keys on the
// accumulator rows (lane group g, register r -> key 4g + r of a 16-key tile) and one query per lane column, so max / sum / lse / delta
// are lane-local plus two xor-shuffles, and the probabilities feed the next product straight from the accumulator registers: the
// contraction step r of O^T += V^T P^T takes, in lane group g, exactly key 4g + r -- register r of the S^T tile.
// Operand maps (cdna_hip_programming.md): A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D[i = 4 (l >> 4) + r][j = l & 15].
// (the vector-ALU kernels these replaced -- profiles/r02_parity_v1 -- were removed from the product in round 6.)
#define PA2_LD 68
__device__ __forceinline__ void pa2_stage(float (*dst)[PA2_LD], const float* src, int ld) {      // 64 rows x 64 floats, 256 threads
    const int t = threadIdx.x, r = t >> 2, c0 = (t & 3) * 16;
    const float* p = src + (size_t)r * ld + c0;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&dst[r][c0 + 4 * i]) = *reinterpret_cast<const float4*>(p + 4 * i);
}
__device__ __forceinline__ float pa2_max_g(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float pa2_sum_g(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }


// XCD-aware placement (as attn_xcd_remap in attention.hip): the row blocks of one (batch, head) stream the same fp32 K / V (or Q / dO) rows;
// workgroup ids xcd, xcd + 8, ... (one XCD, in dispatch order) walk one pair's row blocks before the next pair
__device__ __forceinline__ void pa2_xcd_remap(int& rb, int& h, int& b) {
    const int nrb = gridDim.x, heads = gridDim.y, nbh = gridDim.y * gridDim.z;
    rb = blockIdx.x; h = blockIdx.y; b = blockIdx.z;
    if (nbh & 7) return;
    const int id = blockIdx.x + nrb * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = id & 7, t = id >> 3, bh = (t / nrb) * 8 + xcd;
    rb = t % nrb; h = bh % heads; b = bh / heads;
}

__global__ __launch_bounds__(256) void pattn2_fwd_kernel(PAttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[64][PA2_LD];
    __shared__ __attribute__((aligned(16))) float Vs[64][PA2_LD];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, i16 = l & 15;
    int rb_, h, b;
    pa2_xcd_remap(rb_, h, b);
    if (a.seq_order) b = a.seq_order[b];
    const int q = rb_ * 64 + w * 16 + i16;
    const int H = a.heads * 64, H3 = 3 * H, ns = pa2_visible_chunks(a, b, a.L / 64);
    const float* base = a.qkv + (size_t)b * a.L * H3 + h * 64;
    const size_t row = ((size_t)b * a.heads + h) * a.L + q;
    float qf[16];                                             // B operand of S^T = K Q^T: Q[q][4i + g]
#pragma unroll
    for (int i = 0; i < 16; ++i) qf[i] = base[(size_t)q * H3 + 4 * i + g];
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_part = 0.f;
    for (int c = 0; c < ns; ++c) {
        __syncthreads();
        pa2_stage(Ks, base + (size_t)c * 64 * H3 + H, H3);
        pa2_stage(Vs, base + (size_t)c * 64 * H3 + 2 * H, H3);
        __syncthreads();
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[16 * kt + i16][4 * i + g], qf[i], s[kt], 0, 0, 0);
        float cmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float4 mb = *reinterpret_cast<const float4*>(a.mask_bias + (size_t)b * a.L + c * 64 + 16 * kt + 4 * g);
            s[kt][0] = fmaf(s[kt][0], a.scale, mb.x); s[kt][1] = fmaf(s[kt][1], a.scale, mb.y);
            s[kt][2] = fmaf(s[kt][2], a.scale, mb.z); s[kt][3] = fmaf(s[kt][3], a.scale, mb.w);
            cmax = fmaxf(fmaxf(cmax, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
        }
        cmax = pa2_max_g(cmax);
        const float m_new = fmaxf(m_run, cmax);
        const float alpha = expf(m_run - m_new);              // exp(-inf) = 0 on the first chunk
        m_run = m_new;
        l_part *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = expf(s[kt][r] - m_run);
                l_part += pv;
                s[kt][r] = pv * pa_keep(a, row, c * 64 + 16 * kt + 4 * g + r);
            }
        // O^T[d][q] += V^T[d][key] P^T[key][q]: contraction step r of key tile kt = keys 16 kt + 4 g + r
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[16 * kt + 4 * g + r][16 * d + i16], s[kt][r], o[d], 0, 0, 0);
    }
    const float lsum = pa2_sum_g(l_part), inv = 1.0f / lsum;
    float* op = a.ctx + ((size_t)b * a.L + q) * H + h * 64;
#pragma unroll
    for (int d = 0; d < 4; ++d) *reinterpret_cast<float4*>(op + 16 * d + 4 * g) = make_float4(o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv);
    if (a.lse && g == 0) a.lse[row] = m_run + logf(lsum);
}

__global__ __launch_bounds__(256) void pattn2_bwd_dq_kernel(PAttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[64][PA2_LD];
    __shared__ __attribute__((aligned(16))) float Vs[64][PA2_LD];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, i16 = l & 15;
    int rb_, h, b;
    pa2_xcd_remap(rb_, h, b);
    if (a.seq_order) b = a.seq_order[b];
    const int q = rb_ * 64 + w * 16 + i16;
    const int H = a.heads * 64, H3 = 3 * H, ns = pa2_visible_chunks(a, b, a.L / 64);
    const float* base = a.qkv + (size_t)b * a.L * H3 + h * 64;
    const size_t tok = (size_t)b * a.L + q, row = ((size_t)b * a.heads + h) * a.L + q;
    if (a.qguard && a.kend) {                               // (workgroup-uniform) a query block of trailing padding with exact-zero dO rows: dQ = 0, delta = 0
        const int ke = a.kend[b];
        if (ke > 0 && rb_ * 64 >= ke && *a.qguard == 0) {
            float* op0 = a.dqkv + tok * H3 + h * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d) *reinterpret_cast<float4*>(op0 + 16 * d + 4 * g) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g == 0) a.delta[row] = 0.f;
            return;
        }
    }
    float qf[16], dof[16], dl = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        qf[i] = base[(size_t)q * H3 + 4 * i + g];
        dof[i] = a.dctx[tok * H + h * 64 + 4 * i + g];
        dl = fmaf(dof[i], a.ctx[tok * H + h * 64 + 4 * i + g], dl);
    }
    const float delta = pa2_sum_g(dl), lse = a.lse[row];
    if (g == 0) a.delta[row] = delta;
    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < ns; ++c) {
        __syncthreads();
        pa2_stage(Ks, base + (size_t)c * 64 * H3 + H, H3);
        pa2_stage(Vs, base + (size_t)c * 64 * H3 + 2 * H, H3);
        __syncthreads();
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[16 * kt + i16][4 * i + g], qf[i], s[kt], 0, 0, 0);
                dp[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[16 * kt + i16][4 * i + g], dof[i], dp[kt], 0, 0, 0);
            }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const float4 mb4 = *reinterpret_cast<const float4*>(a.mask_bias + (size_t)b * a.L + c * 64 + 16 * kt + 4 * g);
            const float mb[4] = {mb4.x, mb4.y, mb4.z, mb4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = expf(fmaf(s[kt][r], a.scale, mb[r]) - lse);
                s[kt][r] = pv * (dp[kt][r] * pa_keep(a, row, c * 64 + 16 * kt + 4 * g + r) - delta);       // dS^T[key][q]
            }
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < 4; ++d) dq[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[16 * kt + 4 * g + r][16 * d + i16], s[kt][r], dq[d], 0, 0, 0);
    }
    float* op = a.dqkv + tok * H3 + h * 64;
#pragma unroll
    for (int d = 0; d < 4; ++d)
        *reinterpret_cast<float4*>(op + 16 * d + 4 * g) = make_float4(dq[d][0] * a.scale, dq[d][1] * a.scale, dq[d][2] * a.scale, dq[d][3] * a.scale);
}

// dK, dV: workgroup = 64 keys (16 per wave, one key per lane column); streams the query chunks (Q rows, dO rows, lse, delta)
__global__ __launch_bounds__(256) void pattn2_bwd_dkv_kernel(PAttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Qs[64][PA2_LD];
    __shared__ __attribute__((aligned(16))) float Ds[64][PA2_LD];
    __shared__ float Ls[64], Dl[64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, i16 = l & 15;
    int rb_, h, b;
    pa2_xcd_remap(rb_, h, b);
    if (a.seq_order) b = a.seq_order[b];
    const int key = rb_ * 64 + w * 16 + i16;
    const int H = a.heads * 64, H3 = 3 * H;
    int ns = a.L / 64;
    const float* base = a.qkv + (size_t)b * a.L * H3 + h * 64;
    const size_t tok = (size_t)b * a.L + key, bh = (size_t)b * a.heads + h;
    if (a.kend) {
        const int ke = a.kend[b];
        if (ke > 0 && rb_ * 64 >= ke) {                     // a key block wholly in the trailing padding: p == 0 for every query -> dK = dV = 0
            float* okp0 = a.dqkv + tok * H3 + H + h * 64;
            float* ovp0 = a.dqkv + tok * H3 + 2 * H + h * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                *reinterpret_cast<float4*>(okp0 + 16 * d + 4 * g) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(ovp0 + 16 * d + 4 * g) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
        if (ke > 0 && a.qguard && *a.qguard == 0) ns = min(ns, (ke + 63) >> 6);      // query chunks of padding: dO = 0 and delta = 0 there
    }
    float kf[16], vf[16];                                     // B operands: K[key][4i + g], V[key][4i + g]
#pragma unroll
    for (int i = 0; i < 16; ++i) { kf[i] = base[(size_t)key * H3 + H + 4 * i + g]; vf[i] = base[(size_t)key * H3 + 2 * H + 4 * i + g]; }
    const float mbk = a.mask_bias[tok];
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) { dk[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[d] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int ci = 0; ci < ns; ++ci) {
        __syncthreads();
        pa2_stage(Qs, base + (size_t)ci * 64 * H3, H3);
        pa2_stage(Ds, a.dctx + ((size_t)b * a.L + ci * 64) * H + h * 64, H);
        if (threadIdx.x < 64) Ls[threadIdx.x] = a.lse[bh * a.L + ci * 64 + threadIdx.x];
        else if (threadIdx.x < 128) Dl[threadIdx.x - 64] = a.delta[bh * a.L + ci * 64 + threadIdx.x - 64];
        __syncthreads();
        f32x4 s[4], dp[4];                                    // S[q][key], dP[q][key]: row q = 16 qt + 4 g + r, column = this lane's key
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) { s[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[qt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                s[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[16 * qt + i16][4 * i + g], kf[i], s[qt], 0, 0, 0);
                dp[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ds[16 * qt + i16][4 * i + g], vf[i], dp[qt], 0, 0, 0);
            }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = 16 * qt + 4 * g + r;
                const float pv = expf(fmaf(s[qt][r], a.scale, mbk) - Ls[qi]);
                const float keep = pa_keep(a, bh * a.L + ci * 64 + qi, key);
                s[qt][r] = pv * (dp[qt][r] * keep - Dl[qi]);      // dS[q][key]
                dp[qt][r] = pv * keep;                            // P_drop[q][key]
            }
        // dV^T[d][key] += dO^T[d][q] P_drop[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    dv[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ds[16 * qt + 4 * g + r][16 * d + i16], dp[qt][r], dv[d], 0, 0, 0);
                    dk[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[16 * qt + 4 * g + r][16 * d + i16], s[qt][r], dk[d], 0, 0, 0);
                }
    }
    float* okp = a.dqkv + tok * H3 + H + h * 64;
    float* ovp = a.dqkv + tok * H3 + 2 * H + h * 64;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        *reinterpret_cast<float4*>(okp + 16 * d + 4 * g) = make_float4(dk[d][0] * a.scale, dk[d][1] * a.scale, dk[d][2] * a.scale, dk[d][3] * a.scale);
        *reinterpret_cast<float4*>(ovp + 16 * d + 4 * g) = make_float4(dv[d][0], dv[d][1], dv[d][2], dv[d][3]);
    }
}

static int pattn_fill(PAttnArgs& a, int B, int L, int heads, float scale, float p, uint64_t seed) {
    if (B <= 0 || L <= 0 || heads <= 0 || (L % 64) || L > 4096) return AMDSEG_ERR_SHAPE;
    if (p < 0.f || p >= 1.f) return AMDSEG_ERR_ARG;
    a.B = B; a.L = L; a.heads = heads; a.scale = scale; a.seed = seed;
    a.thresh = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
    if (p > 0.f && a.thresh == 0) a.thresh = 1;
    a.inv_keep = 1.0f / (1.0f - p);
    return AMDSEG_OK;
}

int amdseg_pattn_fwd_impl(const float* qkv, const float* mask_bias, float* ctx, float* lse, int B, int L, int heads, float scale, float p,
                          uint64_t seed, hipStream_t s, const int* kend, const int* seq_order) {
    if (!qkv || !mask_bias || !ctx) return AMDSEG_ERR_ARG;
    PAttnArgs a = {};
    int rc = pattn_fill(a, B, L, heads, scale, p, seed);
    if (rc) return rc;
    a.qkv = qkv; a.mask_bias = mask_bias; a.ctx = ctx; a.lse = lse;
    a.kend = kend; a.seq_order = kend ? seq_order : nullptr;
    hipLaunchKernelGGL(pattn2_fwd_kernel, dim3(L / 64, heads, B), dim3(256), 0, s, a);
    return amdseg_launch_status();
}

int amdseg_pattn_bwd_impl(const float* qkv, const float* mask_bias, const float* ctx, const float* dctx, const float* lse, float* delta,
                          float* dqkv, int B, int L, int heads, float scale, float p, uint64_t seed, hipStream_t s, const int* kend,
                          const int* seq_order, const int* qguard) {
    if (!qkv || !mask_bias || !ctx || !dctx || !lse || !delta || !dqkv) return AMDSEG_ERR_ARG;
    PAttnArgs a = {};
    int rc = pattn_fill(a, B, L, heads, scale, p, seed);
    if (rc) return rc;
    a.qkv = qkv; a.mask_bias = mask_bias; a.ctx = (float*)ctx; a.lse = (float*)lse; a.dctx = dctx; a.delta = delta; a.dqkv = dqkv;
    a.kend = kend; a.seq_order = kend ? seq_order : nullptr; a.qguard = kend ? qguard : nullptr;
    hipLaunchKernelGGL(pattn2_bwd_dq_kernel, dim3(L / 64, heads, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(pattn2_bwd_dkv_kernel, dim3(L / 64, heads, B), dim3(256), 0, s, a);
    return amdseg_launch_status();
}
