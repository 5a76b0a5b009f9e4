// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libamdseg.
// wave = 64 lanes; MFMA operand/accumulator layouts verified by tools/probe_gfx950.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// (Rounds 1-5 carried compile-time timing probes -- epilogues without stores, operands that cost nothing, a stand-in GELU -- behind a probe-build
//  gate; round 6 removed them with the variants they measured.  The records are under profiles/, the code in the history: 17e81c4.)

typedef uint16_t bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define AMDSEG_OK 0
// launch classes of the in-kernel timer (prof.h / amdseg_prof_read)
#define AMDSEG_PROF_GEMM_NT 0
#define AMDSEG_PROF_GEMM_TN 1
#define AMDSEG_PROF_ATTN_FWD 2
#define AMDSEG_PROF_ATTN_BWD_DQ 3
#define AMDSEG_PROF_ATTN_BWD_DKV 4
#define AMDSEG_PROF_ADD_LN_FWD 5
#define AMDSEG_PROF_LN_BWD 6
#define AMDSEG_PROF_ADAMW 7
#define AMDSEG_PROF_KEEPMASK 8
#define AMDSEG_ERR_SHAPE 1001      // unsupported / misaligned shape
#define AMDSEG_ERR_ARG 1002        // null pointer / bad enum
#define AMDSEG_ERR_LAUNCH 1003     // hip launch error (hipGetLastError non-zero)

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
// LDS-DMA issued as inline asm: 64 lanes x 16 B (or 4 B) from per-lane global addresses to lds_wave_base + lane * size.
// Why not __builtin_amdgcn_global_load_lds: after the builtin the compiler puts s_waitcnt vmcnt(0) in front of the next LDS READ
// BUILTIN with a memory operand it cannot disambiguate (ds_read_b64_tr_b16, plain loads in some kernels) -- i.e. it drains the
// prefetch that was just issued (seen in attention fwd/bwd and the TN GEMM; the waits these kernels need are the explicit
// counted vmcnt + barrier hand-offs in their loops).  m0 is not allocatable; the compiler re-materialises it before its own uses.
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void amdseg_glds16(const void* g, void* lds_wave_base) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)LDS_PTR(char, lds_wave_base));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(m) : "memory", "m0");
}
// the same with the global address split into a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: no VALU
// address arithmetic at the issue point (3 instructions per 1-KiB piece inside an MFMA stream)
__device__ __forceinline__ void amdseg_glds16_saddr(const void* uniform_base, uint32_t lane_byte_offset, void* lds_wave_base) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)LDS_PTR(char, lds_wave_base));
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_byte_offset), "s"(uniform_base), "s"(m) : "memory", "m0");
}
// ... and the LDS destination given as a wave-uniform 32-bit LDS byte address that the compiler places in m0 itself (an "s" operand built from a
// generic pointer costs a null check -- s_cmp_lg_u64 + s_cselect -- and, when the pointer went through a VGPR, two v_readfirstlane per piece)
__device__ __forceinline__ void amdseg_glds16_saddr_lds(const void* uniform_base, uint32_t lane_byte_offset, uint32_t lds_wave_addr) {
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane_byte_offset), "s"(uniform_base), "{m0}"(lds_wave_addr) : "memory");
}
__device__ __forceinline__ void amdseg_glds4(const void* g, void* lds_wave_base) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)LDS_PTR(char, lds_wave_base));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(g), "s"(m) : "memory", "m0");
}

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two fp32 -> packed bf16x2 with the gfx950 hardware conversion (round-to-nearest-even); one VALU op instead of ~14
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// generic load/store of an activation element type T in {float, bf16_t}
template <typename T> struct Act;
template <> struct Act<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Act<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
// 8 consecutive elements <-> 8 floats
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    uint4 q = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
    float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    uint4 q;
    q.x = pack2bf(v[0], v[1]); q.y = pack2bf(v[2], v[3]); q.z = pack2bf(v[4], v[5]); q.w = pack2bf(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = q;
}
template <> __device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// exact-erf GELU ([hf] activations "gelu"; reference mmvts/src/models/cross_encoder/bert_model.py:427-439)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
// bf16 fast path: erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16 output rounding 2^-9) sharing
// ONE exponential e = exp(-x^2/2) between erf(x/sqrt2) and the Gaussian term of the derivative.
__device__ __forceinline__ float erf_as_from_e(float ax_over_sqrt2, float e) {     // erf(z), z >= 0, e = exp(-z^2)
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax_over_sqrt2);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    return 1.0f - poly * e;
}
__device__ __forceinline__ float gelu_fast(float x) {
    const float e = __expf(-0.5f * x * x);
    const float er = erf_as_from_e(fabsf(x) * 0.70710678118654752f, e);
    return 0.5f * x * (1.0f + copysignf(er, x));
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float e = __expf(-0.5f * x * x);
    const float er = erf_as_from_e(fabsf(x) * 0.70710678118654752f, e);
    return 0.5f * (1.0f + copysignf(er, x)) + x * 0.39894228040143268f * e;
}
// Two elements at a time on the packed-fp32 VALU ops (v_pk_mul/add/fma_f32): the GELU epilogues are VALU-bound -- at K = 768 the
// ~25 VALU slots per element of the scalar form cost MORE SIMD time than the 1536 MFMA flops of the element (ablation: 25.6 us
// of the 114 us FFN-up GEMM, 16.5 us of its GELU' backward).  Same formula and constants as gelu_fast / gelu_grad_fast:
//   gelu(x)  = h + |h| (1 - q),  gelu'(x) = 1/2 + s (1 - q) + x e / sqrt(2 pi),   h = x/2, s = copysign(1/2, x),
//   q = poly(t) e,  t = 1 / (1 + p |x| / sqrt 2),  e = exp(-x^2 / 2) = exp2(-x^2 log2(e) / 2)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_pair_core(f32x2 x, f32x2& ax, f32x2& e, f32x2& q) {
    ax = (f32x2){fabsf(x.x), fabsf(x.y)};
    const f32x2 x2 = x * x;
    const f32x2 a = x2 * -0.72134752044448170f;
    e = (f32x2){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f32x2 d = ax * (0.3275911f * 0.70710678118654752f) + 1.0f;
    const f32x2 t = (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2 p = t * 1.061405429f + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    q = (p * t) * e;
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    f32x2 ax, e, q;
    gelu_pair_core(x, ax, e, q);
    const f32x2 h = x * 0.5f, ah = ax * 0.5f;
    return (h + ah) - ah * q;
}
__device__ __forceinline__ f32x2 gelu_grad_fast2(f32x2 x) {
    f32x2 ax, e, q;
    gelu_pair_core(x, ax, e, q);
    const f32x2 s = (f32x2){copysignf(0.5f, x.x), copysignf(0.5f, x.y)};
    return ((s + 0.5f) - s * q) + (x * e) * 0.39894228040143268f;
}
// ---- the form the bf16 GEMM epilogues use.  They are VALU-bound (two waves per SIMD evaluating 128 GELUs per lane with the matrix pipe
// idle; ~33 us of the 99-us FFN-up GEMM, profiles/r01_gemm_experiments.md) and that time cannot be hidden under another wave's MFMAs
// (profiles/r03_gemm_epilogue_overlap.md), so what is left is fewer instructions per element.  gelu(x) = x Phi(x) with
//     Phi(x) ~ sigmoid(x (a + b x^2 + c x^4)),   (a, b, c) = (1.5950157686, 0.0740112920, -0.0007030336)
// a minimax fit to the erf form on |x| <= 9 (x^2 clamped at 64, where Phi is 0 / 1 to fp32 precision): max |error| 2.5e-5 in gelu, 1.1e-4
// in its derivative (the tanh form "gelu_new" with the same cost is off by 4.7e-4).  These are ABSOLUTE errors: they sit below the bf16
// rounding of the stored output (2^-9 relative) only where |gelu(x)| >~ 0.013 (resp. |gelu'(x)| >~ 0.056); for outputs nearer zero (x < -3.3,
// |x| < 0.026) the fit's error is the larger of the two -- still <= 2.5e-5 absolute on values that enter a K = 3072 dot product next to O(1)
// terms.  The model-level effect is measured against the REFERENCE's erf GELU at full size by the bf16 legs of
// tests/test_gpu_fullsize.py and reported per round (profiles/rNN_parity_values.json); -DAMDSEG_GELU_ERF_EPILOGUE (compile-checked by
// tests/test_cpu_host.py, `build.build(extra_flags=[...], out=...)`) restores the erf form in the bf16 epilogues.
// One exp2 and one rcp per element instead of exp2 + rcp + a 5-term polynomial: ~14 instead of ~22 issue slots per element forward, ~19
// instead of ~24 for the derivative.  The fp32 / parity paths keep the exact erf (gemm_f32.hip, parity.hip).
#ifndef AMDSEG_GELU_ERF_EPILOGUE
#define GELU_SIG_A (-1.5950157685725626f * 1.4426950408889634f)      // the constants carry -log2(e): the sigmoid is 1 / (1 + exp2(.))
#define GELU_SIG_B (-0.07401129204482199f * 1.4426950408889634f)
#define GELU_SIG_C (0.0007030335771565074f * 1.4426950408889634f)
__device__ __forceinline__ void gelu_sig_core(f32x2 x, f32x2& r, f32x2& x2c) {       // r = Phi(x)
    const f32x2 x2 = x * x;
    x2c = (f32x2){fminf(x2.x, 64.f), fminf(x2.y, 64.f)};
    const f32x2 w = x * ((x2c * GELU_SIG_C + GELU_SIG_B) * x2c + GELU_SIG_A);
    const f32x2 d = (f32x2){__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)} + 1.0f;
    r = (f32x2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ f32x2 gelu_sig2(f32x2 x) {
    f32x2 r, x2c;
    gelu_sig_core(x, r, x2c);
    return x * r;
}
__device__ __forceinline__ f32x2 gelu_sig_grad2(f32x2 x) {           // d/dx [x Phi] = Phi + x Phi (1 - Phi) s'(x), s' = a + 3 b x^2 + 5 c x^4
    f32x2 r, x2c;
    gelu_sig_core(x, r, x2c);
    const f32x2 sp = (x2c * (-5.0f * GELU_SIG_C * 0.6931471805599453f) + (-3.0f * GELU_SIG_B * 0.6931471805599453f)) * x2c
                     + (-GELU_SIG_A * 0.6931471805599453f);
    return (r - r * r) * (x * sp) + r;
}
#endif

// gelu and its derivative from ONE sigmoid evaluation (the forward GEMM epilogue that keeps the derivative for backward, AMDSEG_EPI_KEEP_DERIV)
__device__ __forceinline__ void gelu_both2(f32x2 x, f32x2& h, f32x2& d) {
#ifndef AMDSEG_GELU_ERF_EPILOGUE
    f32x2 r, x2c;
    gelu_sig_core(x, r, x2c);
    const f32x2 sp = (x2c * (-5.0f * GELU_SIG_C * 0.6931471805599453f) + (-3.0f * GELU_SIG_B * 0.6931471805599453f)) * x2c
                     + (-GELU_SIG_A * 0.6931471805599453f);
    h = x * r;
    d = (r - r * r) * (x * sp) + r;
#else
    h = gelu_fast2(x); d = gelu_grad_fast2(x);
#endif
}
// ... with the derivative already in the units of its one-byte code (gemm_epi.h: q = round(gelu' * 200 + 27)): the scale rides the polynomial's
// constants and the offset the final sum, one packed fma in place of two scalar multiply-adds per pair (the epilogue that calls this is VALU time)
__device__ __forceinline__ void gelu_both2q(f32x2 x, f32x2& h, f32x2& dq, float scale, float off) {
#ifndef AMDSEG_GELU_ERF_EPILOGUE
    f32x2 r, x2c;
    gelu_sig_core(x, r, x2c);
    const f32x2 sp = (x2c * (-5.0f * GELU_SIG_C * 0.6931471805599453f * scale) + (-3.0f * GELU_SIG_B * 0.6931471805599453f * scale)) * x2c
                     + (-GELU_SIG_A * 0.6931471805599453f * scale);
    h = x * r;
    dq = (r - r * r) * (x * sp) + (r * scale + off);
#else
    f32x2 d;
    h = gelu_fast2(x); d = gelu_grad_fast2(x);
    dq = d * scale + off;
#endif
}
__device__ __forceinline__ void gelu_both4q(float* v, float* dq, float scale, float off) {
    f32x2 h0, d0, h1, d1;
    gelu_both2q((f32x2){v[0], v[1]}, h0, d0, scale, off); gelu_both2q((f32x2){v[2], v[3]}, h1, d1, scale, off);
    v[0] = h0.x; v[1] = h0.y; v[2] = h1.x; v[3] = h1.y;
    dq[0] = d0.x; dq[1] = d0.y; dq[2] = d1.x; dq[3] = d1.y;
}
__device__ __forceinline__ void gelu_both4(float* v, float* d) {
    f32x2 h0, d0, h1, d1;
    gelu_both2((f32x2){v[0], v[1]}, h0, d0); gelu_both2((f32x2){v[2], v[3]}, h1, d1);
    v[0] = h0.x; v[1] = h0.y; v[2] = h1.x; v[3] = h1.y;
    d[0] = d0.x; d[1] = d0.y; d[2] = d1.x; d[3] = d1.y;
}

// "gelu_new" (tanh approximation; [hf] activations.py NewGELUActivation -- BigBird's default hidden_act) and its derivative
__device__ __forceinline__ float gelu_tanh_fast(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
    return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float gelu_tanh_grad_fast(float x) {
    const float x2 = x * x;
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 0.134145f * x2);
}
// value and derivative of gelu_new from ONE tanh (one exp + one rcp)
__device__ __forceinline__ void gelu_tanh_both(float x, float& h, float& d) {
    const float x2 = x * x;
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
    const float hp = 0.5f * (1.0f + t);
    h = x * hp;
    d = hp + 0.5f * x * (1.0f - t * t) * 0.7978845608028654f * (1.0f + 0.134145f * x2);
}
// act: 0 = exact (erf) GELU, 1 = gelu_new; wave-uniform
__device__ __forceinline__ float gelu_act(float x, int act) { return act ? gelu_tanh_fast(x) : gelu_fast(x); }
__device__ __forceinline__ float gelu_grad_act(float x, int act) { return act ? gelu_tanh_grad_fast(x) : gelu_grad_fast(x); }
// four values of one accumulator fragment at a time (the erf form runs on the packed ops)
__device__ __forceinline__ void gelu_act4(float* v, int act) {
    if (act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_fast(v[e]);
    } else {
#ifndef AMDSEG_GELU_ERF_EPILOGUE
        const f32x2 a = gelu_sig2((f32x2){v[0], v[1]}), b = gelu_sig2((f32x2){v[2], v[3]});
#else
        const f32x2 a = gelu_fast2((f32x2){v[0], v[1]}), b = gelu_fast2((f32x2){v[2], v[3]});
#endif
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
}
__device__ __forceinline__ void gelu_grad_mul4(float* v, float r0, float r1, float r2, float r3, int act) {
    if (act) {
        v[0] *= gelu_tanh_grad_fast(r0); v[1] *= gelu_tanh_grad_fast(r1); v[2] *= gelu_tanh_grad_fast(r2); v[3] *= gelu_tanh_grad_fast(r3);
    } else {
#ifndef AMDSEG_GELU_ERF_EPILOGUE
        const f32x2 a = gelu_sig_grad2((f32x2){r0, r1}), b = gelu_sig_grad2((f32x2){r2, r3});
#else
        const f32x2 a = gelu_grad_fast2((f32x2){r0, r1}), b = gelu_grad_fast2((f32x2){r2, r3});
#endif
        v[0] *= a.x; v[1] *= a.y; v[2] *= b.x; v[3] *= b.y;
    }
}

// Stateless counter-based dropout RNG: keep(element idx) iff hash(seed, idx) >= threshold.
// The same (seed, idx) is re-evaluated in backward, so no mask is stored.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t idx) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    return mix32(mix32(lo ^ s0) + hi * 0x9e3779b9u + s1);
}
// thresh = floor(p * 2^32); keep when rng >= thresh
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh) { return rng_u32(seed, idx) >= thresh; }
// Hidden-state dropout of the row kernels works on 16-B chunks (8 consecutive elements): ONE two-round hash of (seed, chunk index),
// then one multiply round per element pair, 16 bits per element -- 8 quarter-rate v_mul_lo_u32 per chunk instead of 32 (the
// per-element hash was a quarter of ln_bwd's time at p = 0.1).  keep element e iff its field >= thresh16, thresh16 = round(p * 2^16);
// inv_keep = 2^16 / (2^16 - thresh16), so the scaling is unbiased for the realised rate.
// the 8 keep decisions of a chunk as a bit mask (bit e = element e is kept): what the forward row kernel stores per chunk (1 byte) so that
// the backward reads the decisions instead of re-hashing them (round 4: the hash was +6.4 us of ln_bwd's 29)
__device__ __forceinline__ uint32_t drop8_bits(uint64_t seed, uint64_t chunk, uint32_t thresh16) {
    const uint32_t h = rng_u32(seed, chunk);
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t x = h ^ (0x9e3779b9u * (uint32_t)(i + 1));
        x *= 0x7feb352du; x ^= x >> 15;
        bits |= ((x & 0xffffu) >= thresh16 ? 1u : 0u) << (2 * i);
        bits |= ((x >> 16) >= thresh16 ? 1u : 0u) << (2 * i + 1);
    }
    return bits;
}
__device__ __forceinline__ void drop8_apply_bits(uint32_t bits, float inv_keep, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bits >> e) & 1u ? v[e] * inv_keep : 0.f;
}
__device__ __forceinline__ void drop8_apply(uint64_t seed, uint64_t chunk, uint32_t thresh16, float inv_keep, float (&v)[8]) {
    const uint32_t h = rng_u32(seed, chunk);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t x = h ^ (0x9e3779b9u * (uint32_t)(i + 1));
        x *= 0x7feb352du; x ^= x >> 15;
        v[2 * i] = (x & 0xffffu) >= thresh16 ? v[2 * i] * inv_keep : 0.f;
        v[2 * i + 1] = (x >> 16) >= thresh16 ? v[2 * i + 1] * inv_keep : 0.f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// the same sum without LDS round trips: four DPP butterfly steps inside each row of 16 lanes (quad xor 1, quad xor 2, half-row mirror, row
// mirror: VALU operand modifiers), then the four row sums through v_readlane.  wave_sum's six ds_bpermute trips are a ~700-cycle dependent
// chain; ln_bwd runs two of them per token row (36.9 -> 35.0 us stand-alone).  A different summation order: used where a kernel opts in.
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define AMDSEG_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    AMDSEG_DPP_ADD(0xB1);          // quad_perm [1, 0, 3, 2]
    AMDSEG_DPP_ADD(0x4E);          // quad_perm [2, 3, 0, 1]
    AMDSEG_DPP_ADD(0x141);         // row_half_mirror
    AMDSEG_DPP_ADD(0x140);         // row_mirror
#undef AMDSEG_DPP_ADD
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int amdseg_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMDSEG_OK : (int)e;
}
