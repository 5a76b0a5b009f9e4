// fp32 "parity mode" kernels for gfx950: exact-fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain at the
// fp32 vector rate = 1/16 of the bf16 MFMA rate) and a plain fp32 attention.  Used for inference when logits must
// match the fp32 CPU reference to ~1e-5 (north star: <= 1e-3); the throughput path is the bf16 kernels of gemm.hip /
// attention.hip.  Same arithmetic as [hf] models/bert/modeling_bert.py:111-136,175-177,282-293,325-351.
//
// GEMM: 128x128 tile, BK = 32 fp32 (128-B rows, the same global_load_lds staging + XOR swizzle as the bf16 kernel),
// 4 waves x (64x64) = 2x2 frags of 32x32.  k-slot trick: lane half h = lane>>5 owns k = 8c + 4h + j for the j-th of 4
// consecutive MFMAs, so each lane fetches its A/B scalars with ONE conflict-free ds_read_b128 per 4 MFMAs.
#include "common.h"
#include "amdseg_internal.h"

enum { F32_EPI_NONE = 0, F32_EPI_BIAS = 1, F32_EPI_BIAS_GELU = 2 };

struct GemmF32Args {
    const float* A; const float* B; float* C; const float* bias;
    int lda, ldb, ldc, M, N, K, tiles_m, tiles_n;
    int act;
};

__device__ __forceinline__ void f32_stage(const float* __restrict__ G, int ld, int row0, int k0, char* lds_tile, int w, int l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int R0 = w * 32 + q * 8;
        int r = R0 + (l >> 3), s = l & 7;
        int c = s ^ ((r >> 1) & 7);
        __builtin_amdgcn_global_load_lds(GLB_PTR(G + (size_t)(row0 + r) * ld + k0 + c * 4), LDS_PTR(void, lds_tile + R0 * 128), 16, 0, 0);
    }
}
__device__ __forceinline__ float4 f32_frag(const char* lds_tile, int r, int c) {
    return *reinterpret_cast<const float4*>(lds_tile + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_nt_kernel(GemmF32Args a) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int t = blockIdx.x;
    const int tm = t / a.tiles_n, tn = t - tm * a.tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
#define fA(i) (smem + (i) * 32768)
#define fB(i) (smem + 16384 + (i) * 32768)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = a.K / 32;
    f32_stage(a.A, a.lda, m0, 0, fA(0), w, l);
    f32_stage(a.B, a.ldb, n0, 0, fB(0), w, l);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            f32_stage(a.A, a.lda, m0, (kt + 1) * 32, fA(cur ^ 1), w, l);
            f32_stage(a.B, a.ldb, n0, (kt + 1) * 32, fB(cur ^ 1), w, l);
        }
        const char* tA = fA(cur);
        const char* tB = fB(cur);
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {                 // 8 k per group: chunk 2*c8 + (l>>5)
            const int c = c8 * 2 + (l >> 5);
            float4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = f32_frag(tA, wr * 64 + i * 32 + (l & 31), c);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = f32_frag(tB, wc * 64 + j * 32 + (l & 31), c);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
    __syncthreads();
    float* sm = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = wc * 64 + j * 32 + (l & 31);
            const float bv = EPI != F32_EPI_NONE ? a.bias[n0 + n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                float v = acc[i][j][r] + bv;
                if (EPI == F32_EPI_BIAS_GELU) v = a.act ? 0.5f * v * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * v * v * v))) : gelu_erf(v);
                sm[m * 128 + n] = v;
            }
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int chunk = it * 256 + tid;
        const int r = chunk >> 5, cc = (chunk & 31) * 4;
        *reinterpret_cast<float4*>(a.C + (size_t)(m0 + r) * a.ldc + n0 + cc) = *reinterpret_cast<const float4*>(sm + r * 128 + cc);
    }
}

int amdseg_gemm_f32_nt_impl(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                            int epi, const float* bias, hipStream_t stream) {
    if (!A || !B || !C) return AMDSEG_ERR_ARG;
    const int act = (epi >> 8) & 1;
    epi &= 0xff;
    if (M <= 0 || N <= 0 || K <= 0 || (M % 128) || (N % 128) || (K % 32)) return AMDSEG_ERR_SHAPE;
    if ((lda % 4) || (ldb % 4) || (ldc % 4)) return AMDSEG_ERR_SHAPE;
    if (epi != F32_EPI_NONE && !bias) return AMDSEG_ERR_ARG;
    GemmF32Args a = {A, B, C, bias, lda, ldb, ldc, M, N, K, M / 128, N / 128, act};
    dim3 grid(a.tiles_m * a.tiles_n);
    switch (epi) {
        case F32_EPI_NONE: hipLaunchKernelGGL(gemm_f32_nt_kernel<F32_EPI_NONE>, grid, dim3(256), 0, stream, a); break;
        case F32_EPI_BIAS: hipLaunchKernelGGL(gemm_f32_nt_kernel<F32_EPI_BIAS>, grid, dim3(256), 0, stream, a); break;
        case F32_EPI_BIAS_GELU: hipLaunchKernelGGL(gemm_f32_nt_kernel<F32_EPI_BIAS_GELU>, grid, dim3(256), 0, stream, a); break;
        default: return AMDSEG_ERR_ARG;
    }
    return amdseg_launch_status();
}

// ------------------------------------------------------------------------------------------------ fp32 attention
// one wave per (b, h, q): lanes own keys j = lane + 64*s for the scores, then own the output column d = lane (d == 64)
#define F32_MAXS 64      // L <= 4096
// window > 0: Longformer band (|q - j| <= window or j < nglobal; see attention.hip), rows of padded queries zeroed
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads,
                                                       float scale, int window, int nglobal) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + w;            // (b*heads + h)*L + q
    const size_t total = (size_t)B * heads * L;
    if (row >= total) return;
    const int q = row % L;
    const int h = (row / L) % heads;
    const int b = row / ((size_t)L * heads);
    const int H = heads * 64, H3 = 3 * H;
    const float* qp = qkv + ((size_t)b * L + q) * H3 + h * 64;
    const float qd = qp[l];                                    // lane holds q[d = lane]
    const int ns = L / 64;
    float s[F32_MAXS];
    float mx = -INFINITY;
#define SLAB_VISIBLE(si) (window <= 0 || (si) * 64 < nglobal || ((si) * 64 + 63 >= q - window && (si) * 64 <= q + window))
    for (int si = 0; si < ns; ++si) {
        if (!SLAB_VISIBLE(si)) { s[si] = -INFINITY; continue; }
        const int j = si * 64 + l;
        const float* kp = qkv + ((size_t)b * L + j) * H3 + H + h * 64;
        float acc = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) acc = fmaf(__shfl(qd, d, 64), kp[d], acc);
        acc = acc * scale + mask_bias[(size_t)b * L + j];
        if (window > 0 && j >= nglobal && (j - q > window || q - j > window)) acc = -INFINITY;
        s[si] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int si = 0; si < ns; ++si) { s[si] = expf(s[si] - mx); sum += s[si]; }
    sum = wave_sum(sum);
    const float inv = (window > 0 && mask_bias[(size_t)b * L + q] < 0.f) ? 0.f : 1.0f / sum;
    float o = 0.f;
    for (int si = 0; si < ns; ++si) {
        if (!SLAB_VISIBLE(si)) continue;
        const float ps = s[si] * inv;
        for (int jj = 0; jj < 64; ++jj) {
            const float pj = __shfl(ps, jj, 64);
            const float* vp = qkv + ((size_t)b * L + si * 64 + jj) * H3 + 2 * H + h * 64;
            o = fmaf(pj, vp[l], o);
        }
    }
    ctx[((size_t)b * L + q) * H + h * 64 + l] = o;
}

int amdseg_attn_f32_impl(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, int d, float scale,
                         int window, int nglobal, hipStream_t s) {
    if (!qkv || !mask_bias || !ctx || window < 0 || nglobal < 0) return AMDSEG_ERR_ARG;
    if (d != 64 || B <= 0 || heads <= 0 || L <= 0 || (L % 64) || L > 64 * F32_MAXS) return AMDSEG_ERR_SHAPE;
    const size_t total = (size_t)B * heads * L;
    hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, qkv, mask_bias, ctx, B, L, heads, scale, window, window > 0 ? nglobal : 0);
    return amdseg_launch_status();
}

// ------------------------------------------------------------------------------------------------ fp32 block-list attention
// BigBird block-sparse attention in fp32 (inference parity mode): for the 64-query block qb of head h ONE softmax over the key
// blocks klist[h][qb][0 .. kcnt[h][qb]) -- duplicates count once per listing, as in the bf16 kernel (attention.hip, LIST mode)
// and the reference's gathered-block formulation.  One wave per (b, h, q), lanes own one key of each listed block.
__global__ __launch_bounds__(256) void attn_list_f32_kernel(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads,
                                                            float scale, const int* klist, const int* kcnt, int stride) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + w;            // (b*heads + h)*L + q
    const size_t total = (size_t)B * heads * L;
    if (row >= total) return;
    const int q = row % L;
    const int h = (row / L) % heads;
    const int b = row / ((size_t)L * heads);
    const int H = heads * 64, H3 = 3 * H, nb = L / 64;
    const int* kl = klist + ((size_t)h * nb + (q >> 6)) * stride;
    const int n = min(kcnt[h * nb + (q >> 6)], F32_MAXS);      // a list never holds more than L/64 <= 64 entries
    const float qd = qkv[((size_t)b * L + q) * H3 + h * 64 + l];
    float s[F32_MAXS];
    float mx = -INFINITY;
    for (int e = 0; e < F32_MAXS; ++e) {
        if (e >= n) { s[e] = -INFINITY; continue; }
        const int j = kl[e] * 64 + l;
        const float* kp = qkv + ((size_t)b * L + j) * H3 + H + h * 64;
        float acc = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) acc = fmaf(__shfl(qd, d, 64), kp[d], acc);
        acc = acc * scale + mask_bias[(size_t)b * L + j];
        s[e] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int e = 0; e < F32_MAXS; ++e) { s[e] = expf(s[e] - mx); sum += s[e]; }
    sum = wave_sum(sum);
    const float inv = mask_bias[(size_t)b * L + q] < 0.f ? 0.f : 1.0f / sum;      // padded query: zero row (context_layer * from_mask)
    float o = 0.f;
    for (int e = 0; e < F32_MAXS; ++e) {
        if (e >= n) break;
        const float ps = s[e] * inv;
        const int j0 = kl[e] * 64;
        for (int jj = 0; jj < 64; ++jj) {
            const float pj = __shfl(ps, jj, 64);
            o = fmaf(pj, qkv[((size_t)b * L + j0 + jj) * H3 + 2 * H + h * 64 + l], o);
        }
    }
    ctx[((size_t)b * L + q) * H + h * 64 + l] = o;
}

int amdseg_attn_list_f32_impl(const float* qkv, const float* mask_bias, float* ctx, int B, int L, int heads, float scale,
                              const int* klist, const int* kcnt, int stride, hipStream_t s) {
    if (!qkv || !mask_bias || !ctx || !klist || !kcnt) return AMDSEG_ERR_ARG;
    if (B <= 0 || heads <= 0 || L <= 0 || (L % 64) || L > 64 * F32_MAXS || stride <= 0) return AMDSEG_ERR_SHAPE;
    const size_t total = (size_t)B * heads * L;
    hipLaunchKernelGGL(attn_list_f32_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, qkv, mask_bias, ctx, B, L, heads, scale, klist, kcnt, stride);
    return amdseg_launch_status();
}
