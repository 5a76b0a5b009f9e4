// Fused optimiser-side kernels over the flat fp32 parameter / gradient buffers (HBM-bound, 16 B/lane streams).
//
// Replaces torch.optim.AdamW.step + torch.nn.utils.clip_grad_norm_ as driven by transformers.Trainer
// ([hf] trainer.py:2539 clip, training_args.py:777-856 defaults; reference launch values run_finetune.sh:29,73:
// lr 5e-5, betas (0.9, 0.999), eps 1e-8, weight_decay 0, max_grad_norm 1.0).  Update rule = torch.optim.AdamW
// (decoupled decay, bias-corrected, eps added to sqrt(v_hat)) -- the oracle for it is torch.optim.AdamW on CPU.
// One pass reads p,g,m,v and writes p,m,v (+ the bf16 compute shadow, + optionally zeroes g): 28-34 B/param.
#include "common.h"
#include "amdseg_internal.h"
#include "prof.h"

// streaming form (round 4): every buffer is touched exactly once per step, so loads and stores carry the non-temporal hint (no reuse to keep in
// L2 / MALL), and a thread handles two float4 quads per trip (8 x 16-B loads in flight).  -DAMDSEG_ADAMW_PLAIN: the plain form.
typedef float aw_f4 __attribute__((ext_vector_type(4)));
typedef unsigned aw_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 aw_ld_nt(const float* p, size_t i) {
    const aw_f4 v = __builtin_nontemporal_load(reinterpret_cast<const aw_f4*>(p) + i);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void aw_st_nt(float* p, size_t i, const float4& v) {
    __builtin_nontemporal_store((aw_f4){v.x, v.y, v.z, v.w}, reinterpret_cast<aw_f4*>(p) + i);
}
__device__ __forceinline__ void aw_st2_nt(bf16_t* p, size_t i, const uint2& v) {
    __builtin_nontemporal_store((aw_u2){v.x, v.y}, reinterpret_cast<aw_u2*>(p) + i);
}
#define AW_LD(p, i) aw_ld_nt(p, i)
#define AW_ST(p, i, v) aw_st_nt(p, i, v)
#define AW_ST2(p, i, v) aw_st2_nt(p, i, v)
__device__ __forceinline__ void adamw_quad(float4& pp, const float4& gg, float4& mm, float4& vv, float gs, float decay, float beta1, float beta2,
                                           float eps, float step_size, float rsqrt_bc2) {
    float P[4] = {pp.x, pp.y, pp.z, pp.w}, G[4] = {gg.x, gg.y, gg.z, gg.w}, Mm[4] = {mm.x, mm.y, mm.z, mm.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gr = G[e] * gs;
        P[e] *= decay;
        Mm[e] = beta1 * Mm[e] + (1.0f - beta1) * gr;
        V[e] = beta2 * V[e] + (1.0f - beta2) * gr * gr;
        const float denom = sqrtf(V[e]) * rsqrt_bc2 + eps;
        P[e] -= step_size * (Mm[e] / denom);
    }
    pp = make_float4(P[0], P[1], P[2], P[3]); mm = make_float4(Mm[0], Mm[1], Mm[2], Mm[3]); vv = make_float4(V[0], V[1], V[2], V[3]);
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ shadow, size_t n4, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float rsqrt_bc2,
                                                    const float* __restrict__ gscale, int zero_grad,
                                                    const unsigned char* __restrict__ chunk_flags) {
    const float gs = gscale ? *gscale : 1.0f;
    const float step_size = lr / bc1;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    // two quads per trip: i and i + stride (both coalesced across the wave)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
        const size_t j = i + stride;
        const bool hj = j < n4;
        // per 64-element chunk (parameters start on 64-element boundaries of the flat buffer): bit 0 = weight decay applies
        // (HF Trainer decays neither biases nor LayerNorm weights), bit 1 = frozen parameter (requires_grad = False): untouched
        const unsigned fi = chunk_flags ? chunk_flags[i >> 4] : 1u;
        const unsigned fj = hj ? (chunk_flags ? chunk_flags[j >> 4] : 1u) : 2u;
        const bool ai = !(fi & 2u), aj = hj && !(fj & 2u);
        float4 pi, gi, mi, vi, pj, gj, mj, vj;
        if (ai) { pi = AW_LD(p, i); gi = AW_LD(g, i); mi = AW_LD(m, i); vi = AW_LD(v, i); }
        if (aj) { pj = AW_LD(p, j); gj = AW_LD(g, j); mj = AW_LD(m, j); vj = AW_LD(v, j); }
        if (ai) {
            adamw_quad(pi, gi, mi, vi, gs, (fi & 1u) ? (1.0f - lr * wd) : 1.0f, beta1, beta2, eps, step_size, rsqrt_bc2);
            AW_ST(p, i, pi); AW_ST(m, i, mi); AW_ST(v, i, vi);
            if (shadow) { uint2 pk; pk.x = pack2bf(pi.x, pi.y); pk.y = pack2bf(pi.z, pi.w); AW_ST2(shadow, i, pk); }
        }
        if (zero_grad) AW_ST(g, i, z4);
        if (aj) {
            adamw_quad(pj, gj, mj, vj, gs, (fj & 1u) ? (1.0f - lr * wd) : 1.0f, beta1, beta2, eps, step_size, rsqrt_bc2);
            AW_ST(p, j, pj); AW_ST(m, j, mj); AW_ST(v, j, vj);
            if (shadow) { uint2 pk; pk.x = pack2bf(pj.x, pj.y); pk.y = pack2bf(pj.z, pj.w); AW_ST2(shadow, j, pk); }
        }
        if (zero_grad && hj) AW_ST(g, j, z4);
    }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, size_t n4, float* partials) {
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = AW_LD(x, i);
        s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_final_kernel(const float* partials, int n, float* out, int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { const float t = red[0] + red[1] + red[2] + red[3]; out[0] = accumulate ? out[0] + t : t; }
}
// coef = min(1, max_norm / (norm + 1e-6)) * extra_scale  (torch.nn.utils.clip_grad_norm_), norm = sqrt(sumsq) * |extra_scale| ... see api
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, float extra_scale, float* coef, float* norm) {
    const float nrm = sqrtf(sumsq[0]) * extra_scale;       // extra_scale: 1/world_size or 1/grad_accum pre-scaling
    if (norm) norm[0] = nrm;
    float c = max_norm > 0.f ? max_norm / (nrm + 1e-6f) : 1.0f;
    c = c > 1.0f ? 1.0f : c;
    coef[0] = c * extra_scale;
}
__global__ void scale_kernel(float* x, size_t n4, const float* coef) {
    const float c = *coef;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(x)[i];
        a.x *= c; a.y *= c; a.z *= c; a.w *= c;
        reinterpret_cast<float4*>(x)[i] = a;
    }
}

static inline unsigned stream_grid(size_t n4) {
    size_t b = (n4 + 255) / 256;
    return (unsigned)(b > 2048 ? 2048 : (b == 0 ? 1 : b));
}

int amdseg_adamw_impl(float* p, const float* g, float* m, float* v, void* shadow, size_t n, float lr, float beta1,
                      float beta2, float eps, float wd, int step, const float* gscale, int zero_grad,
                      const unsigned char* chunk_flags, hipStream_t s) {
    if (!p || !g || !m || !v) return AMDSEG_ERR_ARG;
    if (n == 0 || (n % 4) || step < 1) return AMDSEG_ERR_SHAPE;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    AMDSEG_LAUNCH_PROF(AMDSEG_PROF_ADAMW, (28.0 + (shadow ? 2.0 : 0.0) + (zero_grad ? 4.0 : 0.0)) * (double)n, adamw_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, s, p, (float*)g, m, v, (bf16_t*)shadow, n / 4, lr, beta1,
                       beta2, eps, wd, (float)bc1, (float)(1.0 / sqrt(bc2)), gscale, zero_grad, chunk_flags);
    return amdseg_launch_status();
}

#define SUMSQ_BLOCKS 1024
int amdseg_sumsq_impl(const float* x, size_t n, float* partials, float* out, int accumulate, hipStream_t s) {
    if (!x || !partials || !out) return AMDSEG_ERR_ARG;
    if (n == 0 || (n % 4)) return AMDSEG_ERR_SHAPE;
    unsigned grid = stream_grid(n / 4);
    if (grid > SUMSQ_BLOCKS) grid = SUMSQ_BLOCKS;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, x, n / 4, partials);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, partials, (int)grid, out, accumulate);
    return amdseg_launch_status();
}

int amdseg_clip_coef_impl(const float* sumsq, float max_norm, float extra_scale, float* coef, float* norm, hipStream_t s) {
    if (!sumsq || !coef) return AMDSEG_ERR_ARG;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, s, sumsq, max_norm, extra_scale, coef, norm);
    return amdseg_launch_status();
}

int amdseg_scale_impl(float* x, size_t n, const float* coef, hipStream_t s) {
    if (!x || !coef) return AMDSEG_ERR_ARG;
    if (n == 0 || (n % 4)) return AMDSEG_ERR_SHAPE;
    hipLaunchKernelGGL(scale_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, s, x, n / 4, coef);
    return amdseg_launch_status();
}
