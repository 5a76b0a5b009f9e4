// bf16 MFMA GEMMs for the BERT encoder projections on gfx950 (MI355X).
//
//   gemm_nt : C[M,N] = A[M,K] . B[N,K]^T  (+ fused epilogue)      forward projections and dgrad (with W^T shadows)
//             replaces the torch.nn.Linear calls of [hf] models/bert/modeling_bert.py:175-177 (q,k,v), :282-293
//             (attention output), :325-337 (intermediate + GELU), :340-351 (output) and their autograd backward.
//   gemm_tn : C[N,K'] (+)= sum_m A[m,N] . B[m,K']   grouped launch   weight gradients (dW = dY^T X)
//
// Design (MI355X-first, not a port): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per
// wave, v_mfma_f32_32x32x16_bf16), BK = 64, operands staged HBM -> LDS by direct DMA (global_load_lds, 16 B/lane)
// into a double-buffered 64 KiB LDS image, XOR-swizzled on the SOURCE address so the ds_read_b128 / tr_b16
// fragment reads are bank-conflict free; the k-strided operands of gemm_tn are read with the LDS transpose read
// ds_read_b64_tr_b16 so no transposed activation copies ever touch HBM.  The fp32 accumulators go back through
// LDS so every global store (and every epilogue operand load) is a coalesced 16 B/lane access.  Workgroup ids
// are remapped so that each XCD (private 4 MiB L2) walks a contiguous run of tiles sharing operand panels.
#include "common.h"
#include "amdseg_internal.h"

#define BM 128
#define BN 128
#define BK 64
#define GROUP_M 8

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_GELU = 2, EPI_ADD_RES = 3, EPI_GELU_BWD = 4 };

struct GemmNTArgs {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; const bf16_t* R; bf16_t* C2;
    int lda, ldb, ldc, ldr, ldc2;
    int M, N, K;
    int tiles_m, tiles_n;
};

// bijective XCD-aware remap: hardware places workgroup b on XCD b % 8; give each XCD a contiguous tile range
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(void, lds_wave_base), 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------ gemm_nt
// Pipeline (measured motivation, profiles/r01_v0): with a 2-deep LDS double buffer the kernel moved 21 B/clk/CU = 64 KiB in
// flight per CU / ~3000 clk loaded memory latency, i.e. it was bound by the latency x bandwidth product, not by MFMA or
// L2.  So: BK = 32, a 5-slot LDS ring per workgroup (80 KiB, 2 workgroups per CU -> 128 KiB of DMA in flight per CU), loads
// issued 4 K-steps ahead, COUNTED s_waitcnt vmcnt(N) (never 0 in the steady state) + a raw s_barrier so the DMA queue is
// never drained inside the K loop.
// LDS image of an operand stage: [128 rows][32 k] bf16, 64 B per row, 16-B chunk c of row r stored at slot
// c ^ ((r >> 2) & 3): a ds_read_b128 lane group (16 distinct rows mod 16, one k-chunk) covers all 16 slots of the
// 256-B bank row -> conflict free.
#define NT_NS 5
#define NT_STAGE_BYTES 16384          // A 8 KiB + B 8 KiB
__device__ __forceinline__ void nt_stage(const bf16_t* __restrict__ G, int ld, int row0, int k0, char* lds_tile, int w, int l) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int R0 = w * 32 + q * 16;
        int r = R0 + (l >> 2), s = l & 3;
        int c = s ^ ((r >> 2) & 3);
        glds16(G + (size_t)(row0 + r) * ld + k0 + c * 8, lds_tile + R0 * 64);
    }
}
__device__ __forceinline__ bf16x8 nt_frag(const char* lds_tile, int r, int c) {
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * 64 + ((c ^ ((r >> 2) & 3)) << 4));
}
// wait until this wave's loads of the oldest in-flight stage have landed; `younger` = stages issued after it (0..3)
__device__ __forceinline__ void wait_stage(int younger) {
    if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int EPI, typename OutT>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNTArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // NT_NS * 16 KiB ring; reused as fp32 [128][128] in the epilogue
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int nwg = a.tiles_m * a.tiles_n;
    const int t = xcd_remap(blockIdx.x, nwg);
    // grouped tile order inside each XCD's contiguous range: the ~64 tiles resident on one XCD (32 CUs x 2) form a
    // GROUP_M x 8 patch, so each A/B panel fetched into the XCD's 4 MiB L2 is shared by 8 tiles
    const int gsz_full = GROUP_M * a.tiles_n;
    const int grp = t / gsz_full, first_m = grp * GROUP_M;
    const int gm = min(a.tiles_m - first_m, GROUP_M);
    const int rem = t - grp * gsz_full;
    const int tm = first_m + rem % gm, tn = rem / gm;
    const int m0 = tm * BM, n0 = tn * BN;
#define slotA(i) (smem + (i) * NT_STAGE_BYTES)
#define slotB(i) (smem + (i) * NT_STAGE_BYTES + 8192)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = a.K / 32;
    // prologue: fill NT_NS-1 slots
#pragma unroll
    for (int p = 0; p < NT_NS - 1; ++p)
        if (p < nk) {
            nt_stage(a.A, a.lda, m0, p * 32, slotA(p), w, l);
            nt_stage(a.B, a.ldb, n0, p * 32, slotB(p), w, l);
        }
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const int issued_after = min(nk - 1 - kt, NT_NS - 2);   // stages younger than kt currently in flight
        wait_stage(issued_after);
        __builtin_amdgcn_s_barrier();                           // stage kt visible to all waves; slot (kt-1)%NS free
        if (kt + NT_NS - 1 < nk) {
            const int ps = slot == 0 ? NT_NS - 1 : slot - 1;    // == (kt + NS - 1) % NS
            nt_stage(a.A, a.lda, m0, (kt + NT_NS - 1) * 32, slotA(ps), w, l);
            nt_stage(a.B, a.ldb, n0, (kt + NT_NS - 1) * 32, slotB(ps), w, l);
        }
        const char* tA = slotA(slot);
        const char* tB = slotB(slot);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = kk * 2 + (l >> 5);
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = nt_frag(tA, wr * 64 + i * 32 + (l & 31), c);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = nt_frag(tB, wc * 64 + j * 32 + (l & 31), c);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        slot = slot == NT_NS - 1 ? 0 : slot + 1;
    }
    // ---- epilogue: accumulators -> LDS (fp32 [128][128]) -> coalesced 16 B/lane stores
    __syncthreads();
    float* sm = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = wc * 64 + j * 32 + (l & 31);
            float bv = 0.f;
            if (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU) bv = a.bias[n0 + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                sm[m * BN + n] = acc[i][j][r] + bv;
            }
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int chunk = it * 256 + tid;
        const int r = chunk >> 4, cc = (chunk & 15) * 8;
        float v[8];
        {
            float4 x = *reinterpret_cast<const float4*>(sm + r * BN + cc);
            float4 y = *reinterpret_cast<const float4*>(sm + r * BN + cc + 4);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
        }
        const size_t gm_ = (size_t)(m0 + r);
        const int gn = n0 + cc;
        if (EPI == EPI_BIAS_GELU) {
            st8<bf16_t>(a.C2 + gm_ * a.ldc2 + gn, v);     // pre-activation u, kept for backward
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_fast(v[e]);
        } else if (EPI == EPI_ADD_RES) {
            float rr[8]; ld8<bf16_t>(a.R + gm_ * a.ldr + gn, rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rr[e];
        } else if (EPI == EPI_GELU_BWD) {
            float u[8]; ld8<bf16_t>(a.R + gm_ * a.ldr + gn, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_fast(u[e]);
        }
        st8<OutT>(reinterpret_cast<OutT*>(a.C) + gm_ * a.ldc + gn, v);
    }
}

#define NT_LDS_BYTES (NT_NS * NT_STAGE_BYTES)
template <int EPI, typename OutT>
static int launch_nt(const GemmNTArgs& a, hipStream_t s) {
    static bool attr_set = false;          // > 64 KiB of dynamic LDS must be opted into once per kernel
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<EPI, OutT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, OutT>), dim3(a.tiles_m * a.tiles_n), dim3(256), NT_LDS_BYTES, s, a);
    return amdseg_launch_status();
}

int amdseg_gemm_nt_impl(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                        int epi, const float* bias, const void* R, int ldr, void* C2, int ldc2, int out_fp32,
                        hipStream_t stream) {
    if (!A || !B || !C) return AMDSEG_ERR_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || (M % BM) || (N % BN) || (K % 32)) return AMDSEG_ERR_SHAPE;
    if ((lda % 8) || (ldb % 8) || (ldc % 8)) return AMDSEG_ERR_SHAPE;
    GemmNTArgs a;
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)B; a.C = C; a.bias = bias; a.R = (const bf16_t*)R; a.C2 = (bf16_t*)C2;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.ldc2 = ldc2; a.M = M; a.N = N; a.K = K;
    a.tiles_m = M / BM; a.tiles_n = N / BN;
    switch (epi) {
        case EPI_NONE: return out_fp32 ? launch_nt<EPI_NONE, float>(a, stream) : launch_nt<EPI_NONE, bf16_t>(a, stream);
        case EPI_BIAS:
            if (!bias) return AMDSEG_ERR_ARG;
            return out_fp32 ? launch_nt<EPI_BIAS, float>(a, stream) : launch_nt<EPI_BIAS, bf16_t>(a, stream);
        case EPI_BIAS_GELU:
            if (!bias || !C2 || out_fp32 || (ldc2 % 8)) return AMDSEG_ERR_ARG;
            return launch_nt<EPI_BIAS_GELU, bf16_t>(a, stream);
        case EPI_ADD_RES:
            if (!R || (ldr % 8)) return AMDSEG_ERR_ARG;
            return out_fp32 ? launch_nt<EPI_ADD_RES, float>(a, stream) : launch_nt<EPI_ADD_RES, bf16_t>(a, stream);
        case EPI_GELU_BWD:
            if (!R || out_fp32 || (ldr % 8)) return AMDSEG_ERR_ARG;
            return launch_nt<EPI_GELU_BWD, bf16_t>(a, stream);
    }
    return AMDSEG_ERR_ARG;
}

// ------------------------------------------------------------------------------------------------ gemm_tn
// Operand tiles are [64 m][128 cols] bf16 (256 B per row, 16 chunks of 16 B); chunk c of row m is stored at slot
// c ^ ((m & 3) << 2) so the four rows gathered by one ds_read_b64_tr_b16 half-wave fall on disjoint banks.
struct TNProblem { const bf16_t* A; const bf16_t* B; float* C; int N, Kp, lda, ldb, ldc, tile_begin, tiles_k; };
struct GemmTNArgs { TNProblem p[AMDSEG_MAX_GROUP]; int nprob, M, accumulate, total_tiles; };

__device__ __forceinline__ void tn_stage(const bf16_t* __restrict__ G, int ld, int m0, int col0, char* lds_tile, int w, int l) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int R0 = w * 8 + q * 4;
        int r = R0 + (l >> 4), s = l & 15;
        int c = s ^ ((r & 3) << 2);
        glds16(G + (size_t)(m0 + r) * ld + col0 + c * 8, lds_tile + R0 * 256);
    }
}
// fragment for a 32-wide block of columns starting at col (multiple of 32), k-step kk (16 rows of m)
__device__ __forceinline__ bf16x8 tn_frag(const char* lds_tile, int col, int kk, int l) {
    const int q = l >> 4, i16 = l & 15, nblk = q & 1, g = q >> 1;
    const int c = ((col + nblk * 16) >> 3) + ((i16 & 3) >> 1);
    bf16x8 f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int mr = kk * 16 + g * 8 + h * 4 + (i16 >> 2);
        const int off = mr * 256 + ((c ^ ((mr & 3) << 2)) << 4) + (i16 & 1) * 8;
        bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(bf16x4, lds_tile + off));
        f[h * 4 + 0] = v[0]; f[h * 4 + 1] = v[1]; f[h * 4 + 2] = v[2]; f[h * 4 + 3] = v[3];
    }
    return f;
}

// same 5-slot LDS ring / counted-vmcnt pipeline as gemm_nt (stage = [32 m][128] of A + [32 m][128] of B = 16 KiB)
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTNArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int t = xcd_remap(blockIdx.x, a.total_tiles);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < AMDSEG_MAX_GROUP; ++i)
        if (i < a.nprob && t >= a.p[i].tile_begin) pi = i;
    const TNProblem& P = a.p[pi];
    const int lt = t - P.tile_begin;
    const int tn_ = lt / P.tiles_k, tk = lt - tn_ * P.tiles_k;
    const int n0 = tn_ * 128, k0 = tk * 128;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nm = a.M / 32;
#pragma unroll
    for (int p = 0; p < NT_NS - 1; ++p)
        if (p < nm) {
            tn_stage(P.A, P.lda, p * 32, n0, slotA(p), w, l);
            tn_stage(P.B, P.ldb, p * 32, k0, slotB(p), w, l);
        }
    int slot = 0;
    for (int mt = 0; mt < nm; ++mt) {
        wait_stage(min(nm - 1 - mt, NT_NS - 2));
        __builtin_amdgcn_s_barrier();
        if (mt + NT_NS - 1 < nm) {
            const int ps = slot == 0 ? NT_NS - 1 : slot - 1;
            tn_stage(P.A, P.lda, (mt + NT_NS - 1) * 32, n0, slotA(ps), w, l);
            tn_stage(P.B, P.ldb, (mt + NT_NS - 1) * 32, k0, slotB(ps), w, l);
        }
        const char* tA = slotA(slot);
        const char* tB = slotB(slot);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = tn_frag(tA, wr * 64 + i * 32, kk, l);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = tn_frag(tB, wc * 64 + j * 32, kk, l);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        slot = slot == NT_NS - 1 ? 0 : slot + 1;
    }
    __syncthreads();
    float* sm = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = wc * 64 + j * 32 + (l & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                sm[m * 128 + n] = acc[i][j][r];
            }
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int chunk = it * 256 + tid;
        const int r = chunk >> 5, cc = (chunk & 31) * 4;
        float4 x = *reinterpret_cast<const float4*>(sm + r * 128 + cc);
        float* dst = P.C + (size_t)(n0 + r) * P.ldc + k0 + cc;
        if (a.accumulate) {
            float4 o = *reinterpret_cast<const float4*>(dst);
            x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
        }
        *reinterpret_cast<float4*>(dst) = x;
    }
}

int amdseg_gemm_tn_grouped_impl(int nprob, const void* const* A, const int* lda, const void* const* B, const int* ldb,
                                float* const* C, const int* ldc, const int* N, const int* Kp, int M, int accumulate,
                                hipStream_t stream) {
    if (nprob <= 0 || nprob > AMDSEG_MAX_GROUP || !A || !B || !C) return AMDSEG_ERR_ARG;
    if (M <= 0 || (M % 32)) return AMDSEG_ERR_SHAPE;
    GemmTNArgs a;
    int tiles = 0;
    for (int i = 0; i < nprob; ++i) {
        if (!A[i] || !B[i] || !C[i]) return AMDSEG_ERR_ARG;
        if ((N[i] % 128) || (Kp[i] % 128) || (lda[i] % 8) || (ldb[i] % 8) || (ldc[i] % 4)) return AMDSEG_ERR_SHAPE;
        TNProblem& P = a.p[i];
        P.A = (const bf16_t*)A[i]; P.B = (const bf16_t*)B[i]; P.C = C[i];
        P.N = N[i]; P.Kp = Kp[i]; P.lda = lda[i]; P.ldb = ldb[i]; P.ldc = ldc[i];
        P.tile_begin = tiles; P.tiles_k = Kp[i] / 128;
        tiles += (N[i] / 128) * (Kp[i] / 128);
    }
    for (int i = nprob; i < AMDSEG_MAX_GROUP; ++i) a.p[i] = a.p[0];
    a.nprob = nprob; a.M = M; a.accumulate = accumulate; a.total_tiles = tiles;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, NT_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles), dim3(256), NT_LDS_BYTES, stream, a);
    return amdseg_launch_status();
}
