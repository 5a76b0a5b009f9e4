// PoNet token mixing for gfx950 (alimeeting4mug/src/models/modeling_ponet.py:68-79 -> modelscope PoNetModel, NOT in the
// reference tree: semantics restated in oracle/ponet_oracle.py, parity unpinned).  HBM-bound row kernels over the fused
// projection output  proj [M, 5H] bf16 = (Hq | Hk | Ho | Hl | Hs):
//
//   segment max-pooling  S_n = max of Hs over the valid tokens of n's segment (a contiguous run of equal segment_ids),
//   local max-pooling    L_n = max of Hl over the valid tokens n-1, n, n+1,
//   fusion               ctx_n = (g + S_n) * Ho_n + L_n        (g = global-attention aggregate, [B, H] fp32, see host mirror)
// and their backward.  One wave per token row, 16 B per lane (the LayerNorm recipe).  Runs are reduced in two levels so
// that no wave walks more than 64 rows: token n is a SUB-LEADER iff (n - run_start[n]) % 64 == 0; it reduces rows
// [n, min(n + 63, run_end[n])] into row n of a scratch plane, and every token then combines the <= ceil(len / 64)
// sub-leader rows of its run.  Algorithmic bytes per token (forward): read Hs, Ho, Hl + write ctx = 4 * H * 2 B
// (25.2 MB per 4096-token sequence and layer); backward reads dctx, Ho, Hl, Hs and writes dHo, dHl, dHs.
#include "common.h"
#include "amdseg_internal.h"

#define PN_MAXCH 2                // 8-column chunks per lane: H <= 1024
#define PN_SUB 64

struct PnArgs {
    const bf16_t* proj; int ld;                // fused projections, row stride (5H)
    const float* mask_bias;                    // [M], < 0 = padded token
    const int* run_start; const int* run_end;  // [M] index (within the sequence) of the first / last token of the token's run
    bf16_t* part; unsigned short* parg;        // [M, H] sub-leader rows: partial max of Hs and its argmax token index
    bf16_t* part2; unsigned short* parg2;      // [M, H] run-leader rows: the folded run maximum / argmax
    float* psum2;                              // [M, H] run-leader rows: the folded run sum of E
    bf16_t* partA; unsigned short* pargA; float* psumA;    // [M, H] level-A rows (fan-in 8) of the two trees
    const float* g;                            // [B, H]
    bf16_t* ctx;                               // [M, H]
    const bf16_t* dctx;                        // [M, H]
    bf16_t* dproj;                             // [M, 5H] gradient of the projections (Ho, Hl, Hs columns written here)
    bf16_t* E;                                 // [M, H] dctx * Ho (0 on padded rows)
    float* psum;                               // [M, H] sub-leader rows: partial sum of E
    int M, L, H;
};

template <int NCH>
__device__ __forceinline__ void pn_load(const bf16_t* row, int nch, int l, float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) ld8<bf16_t>(row + c * 8, v[i]);
    }
}
template <int NCH>
__device__ __forceinline__ void pn_store(bf16_t* row, int nch, int l, const float (&v)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) st8<bf16_t>(row + c * 8, v[i]);
    }
}
template <int NCH>
__device__ __forceinline__ void pn_fill(float (&v)[NCH][8], float x) {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = x;
}

// ---------------------------------------------------------------------------------------------------- sub-run max of Hs
// Two tree levels with fan-in 8 (all 8 row loads of a wave in flight; a flat 64-row walk left 2 waves per CU streaming
// 6 KB at a time: 1.5 TB/s):  level A: token n with (n - run_start) % 8 == 0 reduces Hs rows [n, n+7] of its run into
// plane A;  level B: token n with (n - run_start) % 64 == 0 reduces the <= 8 plane-A rows n, n+8, .. into `part`.
template <bool LEVEL_B>
__global__ __launch_bounds__(256) void pn_tree_max_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + w;
    if (n >= a.M || a.mask_bias[n] < 0.f) return;
    const int b = n / a.L, pos = n - b * a.L, rs = a.run_start[n], re = a.run_end[n];
    constexpr int STRIDE = LEVEL_B ? 8 : 1;
    if ((pos - rs) % (STRIDE * 8)) return;
    const int nch = a.H >> 3;
    float v[8][PN_MAXCH][8]; uint4 pa[8][PN_MAXCH]; bool ok[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int t = pos + k * STRIDE;
        const size_t row = (size_t)b * a.L + min(t, re);
        ok[k] = t <= re && a.mask_bias[row] >= 0.f;
        if (LEVEL_B) {
            pn_load<PN_MAXCH>(a.partA + row * a.H, nch, l, v[k]);
#pragma unroll
            for (int i = 0; i < PN_MAXCH; ++i)
                if (l + 64 * i < nch) pa[k][i] = *reinterpret_cast<const uint4*>(a.pargA + row * a.H + (l + 64 * i) * 8);
        } else pn_load<PN_MAXCH>(a.proj + row * a.ld + 4 * a.H, nch, l, v[k]);
    }
    float mx[PN_MAXCH][8]; unsigned short arg[PN_MAXCH][8];
    pn_fill<PN_MAXCH>(mx, -INFINITY);
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) arg[i][e] = (unsigned short)pos;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!ok[k]) continue;
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i) {
            const uint32_t pw[4] = {pa[k][i].x, pa[k][i].y, pa[k][i].z, pa[k][i].w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (v[k][i][e] > mx[i][e]) {                                         // first maximum wins
                    mx[i][e] = v[k][i][e];
                    arg[i][e] = LEVEL_B ? (unsigned short)((pw[e >> 1] >> ((e & 1) * 16)) & 0xffffu) : (unsigned short)(pos + k);
                }
        }
    }
    bf16_t* dv = LEVEL_B ? a.part : a.partA;
    unsigned short* da = LEVEL_B ? a.parg : a.pargA;
    pn_store<PN_MAXCH>(dv + (size_t)n * a.H, nch, l, mx);
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) {
            uint4 pk;
            pk.x = arg[i][0] | ((uint32_t)arg[i][1] << 16); pk.y = arg[i][2] | ((uint32_t)arg[i][3] << 16);
            pk.z = arg[i][4] | ((uint32_t)arg[i][5] << 16); pk.w = arg[i][6] | ((uint32_t)arg[i][7] << 16);
            *reinterpret_cast<uint4*>(da + (size_t)n * a.H + c * 8) = pk;
        }
    }
}

// run max S (and, when ARG, its argmax token) from the sub-leader rows of a run (walked by the run LEADER only)
template <bool ARG>
__device__ __forceinline__ void pn_run_max_walk(const PnArgs& a, int b, int rs, int re, int nch, int l, float (&S)[PN_MAXCH][8],
                                                unsigned short (&arg)[PN_MAXCH][8]) {
    pn_fill<PN_MAXCH>(S, -INFINITY);
    for (int k = rs; k <= re; k += PN_SUB) {
        const size_t row = (size_t)b * a.L + k;
        if (a.mask_bias[row] < 0.f) continue;                      // (a padded sub-leader wrote nothing)
        float v[PN_MAXCH][8];
        pn_load<PN_MAXCH>(a.part + row * a.H, nch, l, v);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i) {
            const int c = l + 64 * i;
            uint4 pk = make_uint4(0, 0, 0, 0);
            if (ARG && c < nch) pk = *reinterpret_cast<const uint4*>(a.parg + row * a.H + c * 8);
            const uint32_t pw[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c < nch && v[i][e] > S[i][e]) {
                    S[i][e] = v[i][e];
                    if (ARG) arg[i][e] = (unsigned short)((pw[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
                }
        }
    }
}


// level 2: the run leader (pos == run_start) folds the sub-leader rows of its run into ONE row (part2 / parg2 / psum2 at
// the leader's row), so that every other token of the run reads a single row
__global__ __launch_bounds__(256) void pn_run_fold_max_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + w;
    if (n >= a.M || a.mask_bias[n] < 0.f) return;
    const int b = n / a.L, pos = n - b * a.L, rs = a.run_start[n];
    if (pos != rs) return;
    const int nch = a.H >> 3;
    float S[PN_MAXCH][8]; unsigned short arg[PN_MAXCH][8];
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) arg[i][e] = (unsigned short)pos;
    pn_run_max_walk<true>(a, b, rs, a.run_end[n], nch, l, S, arg);
    pn_store<PN_MAXCH>(a.part2 + (size_t)n * a.H, nch, l, S);
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) {
            uint4 pk;
            pk.x = arg[i][0] | ((uint32_t)arg[i][1] << 16); pk.y = arg[i][2] | ((uint32_t)arg[i][3] << 16);
            pk.z = arg[i][4] | ((uint32_t)arg[i][5] << 16); pk.w = arg[i][6] | ((uint32_t)arg[i][7] << 16);
            *reinterpret_cast<uint4*>(a.parg2 + (size_t)n * a.H + c * 8) = pk;
        }
    }
}
__global__ __launch_bounds__(256) void pn_run_fold_sum_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + w;
    if (n >= a.M || a.mask_bias[n] < 0.f) return;
    const int b = n / a.L, pos = n - b * a.L, rs = a.run_start[n], re = a.run_end[n];
    if (pos != rs) return;
    const int nch = a.H >> 3;
    float G[PN_MAXCH][8];
    pn_fill<PN_MAXCH>(G, 0.f);
    for (int k = rs; k <= re; k += PN_SUB) {
        const size_t row = (size_t)b * a.L + k;
        if (a.mask_bias[row] < 0.f) continue;
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                float v[8];
                ld8<float>(a.psum + row * a.H + c * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) G[i][e] += v[e];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) st8<float>(a.psum2 + (size_t)n * a.H + c * 8, G[i]);
    }
}
// single-row read of the folded run maximum (row of the run leader)
template <bool ARG>
__device__ __forceinline__ void pn_run_max(const PnArgs& a, int b, int rs, int re, int nch, int l, float (&S)[PN_MAXCH][8],
                                           unsigned short (&arg)[PN_MAXCH][8]) {
    const size_t row = (size_t)b * a.L + rs;
    pn_load<PN_MAXCH>(a.part2 + row * a.H, nch, l, S);
    if (ARG) {
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i) {
            const int c = l + 64 * i;
            if (c < nch) {
                const uint4 pk = *reinterpret_cast<const uint4*>(a.parg2 + row * a.H + c * 8);
                const uint32_t pw[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) arg[i][e] = (unsigned short)((pw[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward fusion
__global__ __launch_bounds__(256) void pn_combine_fwd_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + w;
    if (n >= a.M) return;
    const int nch = a.H >> 3;
    float out[PN_MAXCH][8];
    if (a.mask_bias[n] < 0.f) {
        pn_fill<PN_MAXCH>(out, 0.f);
        pn_store<PN_MAXCH>(a.ctx + (size_t)n * a.H, nch, l, out);
        return;
    }
    const int b = n / a.L, pos = n - b * a.L;
    float S[PN_MAXCH][8]; unsigned short dummy[PN_MAXCH][8];
    pn_run_max<false>(a, b, a.run_start[n], a.run_end[n], nch, l, S, dummy);
    const bf16_t* prow = a.proj + (size_t)n * a.ld;
    float ho[PN_MAXCH][8], lm[PN_MAXCH][8], t[PN_MAXCH][8];
    pn_load<PN_MAXCH>(prow + 2 * a.H, nch, l, ho);
    pn_load<PN_MAXCH>(prow + 3 * a.H, nch, l, lm);
    if (pos > 0 && a.mask_bias[n - 1] >= 0.f) {
        pn_load<PN_MAXCH>(prow - a.ld + 3 * a.H, nch, l, t);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) lm[i][e] = fmaxf(lm[i][e], t[i][e]);
    }
    if (pos + 1 < a.L && a.mask_bias[n + 1] >= 0.f) {
        pn_load<PN_MAXCH>(prow + a.ld + 3 * a.H, nch, l, t);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) lm[i][e] = fmaxf(lm[i][e], t[i][e]);
    }
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) {
            float gv[8];
            ld8<float>(a.g + (size_t)b * a.H + c * 8, gv);
#pragma unroll
            for (int e = 0; e < 8; ++e) out[i][e] = (gv[e] + S[i][e]) * ho[i][e] + lm[i][e];
        }
    }
    pn_store<PN_MAXCH>(a.ctx + (size_t)n * a.H, nch, l, out);
}

// ---------------------------------------------------------------------------------------------------- backward, per token
// dHo = dctx * (g + S);  E = dctx * Ho;  dHl_j = sum over the valid neighbours n of j (incl. j) of dctx_n * [argmax of
// {Hl_{n-1}, Hl_n, Hl_{n+1}} (valid ones, first maximum in that order) == j]
__global__ __launch_bounds__(256) void pn_bwd_token_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= a.M) return;
    const int nch = a.H >> 3;
    bf16_t* drow = a.dproj + (size_t)j * a.ld;
    float z[PN_MAXCH][8];
    if (a.mask_bias[j] < 0.f) {
        pn_fill<PN_MAXCH>(z, 0.f);
        pn_store<PN_MAXCH>(drow + 2 * a.H, nch, l, z); pn_store<PN_MAXCH>(drow + 3 * a.H, nch, l, z);
        pn_store<PN_MAXCH>(drow + 4 * a.H, nch, l, z); pn_store<PN_MAXCH>(a.E + (size_t)j * a.H, nch, l, z);
        return;
    }
    const int b = j / a.L, pos = j - b * a.L;
    float S[PN_MAXCH][8]; unsigned short dummy[PN_MAXCH][8];
    pn_run_max<false>(a, b, a.run_start[j], a.run_end[j], nch, l, S, dummy);
    const bf16_t* prow = a.proj + (size_t)j * a.ld;
    float dc[PN_MAXCH][8], ho[PN_MAXCH][8], o1[PN_MAXCH][8], o2[PN_MAXCH][8];
    pn_load<PN_MAXCH>(a.dctx + (size_t)j * a.H, nch, l, dc);
    pn_load<PN_MAXCH>(prow + 2 * a.H, nch, l, ho);
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) {
            float gv[8];
            ld8<float>(a.g + (size_t)b * a.H + c * 8, gv);
#pragma unroll
            for (int e = 0; e < 8; ++e) { o1[i][e] = dc[i][e] * (gv[e] + S[i][e]); o2[i][e] = dc[i][e] * ho[i][e]; }
        }
    }
    pn_store<PN_MAXCH>(drow + 2 * a.H, nch, l, o1);
    pn_store<PN_MAXCH>(a.E + (size_t)j * a.H, nch, l, o2);
    // local max-pool backward: Hl rows j-2 .. j+2 (validity-gated), dctx rows j-1, j, j+1
    bool val[5];
    float hl[5][PN_MAXCH][8];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int p = pos + k - 2;
        val[k] = p >= 0 && p < a.L && a.mask_bias[(size_t)b * a.L + p] >= 0.f;
        if (val[k]) pn_load<PN_MAXCH>(a.proj + ((size_t)b * a.L + p) * a.ld + 3 * a.H, nch, l, hl[k]);
        else pn_fill<PN_MAXCH>(hl[k], -INFINITY);
    }
    float dl[PN_MAXCH][8];
    pn_fill<PN_MAXCH>(dl, 0.f);
#pragma unroll
    for (int k = 1; k <= 3; ++k) {              // neighbour n = pos + k - 2 whose window is hl[k-1], hl[k], hl[k+1]; j is hl[2]
        if (!val[k]) continue;
        float dn[PN_MAXCH][8];
        if (k == 2) {
#pragma unroll
            for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) dn[i][e] = dc[i][e];
        } else pn_load<PN_MAXCH>(a.dctx + ((size_t)b * a.L + pos + k - 2) * a.H, nch, l, dn);
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x0 = hl[k - 1][i][e], x1 = hl[k][i][e], x2 = hl[k + 1][i][e];
                const int am = (x0 >= x1 && x0 >= x2) ? k - 1 : ((x1 >= x2) ? k : k + 1);       // first maximum of the window
                if (am == 2) dl[i][e] += dn[i][e];
            }
    }
    pn_store<PN_MAXCH>(drow + 3 * a.H, nch, l, dl);
}

// ---------------------------------------------------------------------------------------------------- sub-run sum of E
template <bool LEVEL_B>
__global__ __launch_bounds__(256) void pn_tree_sum_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + w;
    if (n >= a.M || a.mask_bias[n] < 0.f) return;
    const int b = n / a.L, pos = n - b * a.L, rs = a.run_start[n], re = a.run_end[n];
    constexpr int STRIDE = LEVEL_B ? 8 : 1;
    if ((pos - rs) % (STRIDE * 8)) return;
    const int nch = a.H >> 3;
    float v[8][PN_MAXCH][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t row = (size_t)b * a.L + min(pos + k * STRIDE, re);
        if (LEVEL_B) {
#pragma unroll
            for (int i = 0; i < PN_MAXCH; ++i)
                if (l + 64 * i < nch) ld8<float>(a.psumA + row * a.H + (l + 64 * i) * 8, v[k][i]);
        } else pn_load<PN_MAXCH>(a.E + row * a.H, nch, l, v[k]);                      // E is 0 on padded rows
    }
    float sm[PN_MAXCH][8];
    pn_fill<PN_MAXCH>(sm, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (pos + k * STRIDE > re) continue;
#pragma unroll
        for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) sm[i][e] += v[k][i][e];
    }
    float* dst = LEVEL_B ? a.psum : a.psumA;
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) st8<float>(dst + (size_t)n * a.H + c * 8, sm[i]);
    }
}
// dHs_j = [argmax of j's run == j] * (sum of E over the run)
__global__ __launch_bounds__(256) void pn_bwd_route_kernel(PnArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + w;
    if (j >= a.M || a.mask_bias[j] < 0.f) return;                       // padded rows were zeroed by pn_bwd_token_kernel
    const int b = j / a.L, pos = j - b * a.L, rs = a.run_start[j], re = a.run_end[j], nch = a.H >> 3;
    float S[PN_MAXCH][8]; unsigned short arg[PN_MAXCH][8];
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) arg[i][e] = 0xffffu;
    pn_run_max<true>(a, b, rs, re, nch, l, S, arg);
    float G[PN_MAXCH][8];
    pn_fill<PN_MAXCH>(G, 0.f);
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i) {
        const int c = l + 64 * i;
        if (c < nch) ld8<float>(a.psum2 + ((size_t)b * a.L + rs) * a.H + c * 8, G[i]);
    }
    float out[PN_MAXCH][8];
#pragma unroll
    for (int i = 0; i < PN_MAXCH; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) out[i][e] = (arg[i][e] == (unsigned short)pos) ? G[i][e] : 0.f;
    pn_store<PN_MAXCH>(a.dproj + (size_t)j * a.ld + 4 * a.H, nch, l, out);
}

// ---------------------------------------------------------------------------------------------------- launchers
static int pn_check(int B, int L, int H, int ld) {
    if (B <= 0 || L <= 0 || L > 65535 || H <= 0 || (H % 8) || H > 8 * 64 * PN_MAXCH || ld < 5 * H || (ld % 8)) return AMDSEG_ERR_SHAPE;
    return AMDSEG_OK;
}

int amdseg_ponet_pool_fwd_impl(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end,
                               const float* g, void* part, void* parg, void* ctx, int B, int L, int H, hipStream_t s) {
    // part / parg hold THREE [M, H] planes each: sub-leader (64) rows, folded run-leader rows, level-A (8) rows
    if (!proj || !mask_bias || !run_start || !run_end || !g || !part || !parg || !ctx) return AMDSEG_ERR_ARG;
    int rc = pn_check(B, L, H, ld);
    if (rc) return rc;
    PnArgs a = {};
    a.proj = (const bf16_t*)proj; a.ld = ld; a.mask_bias = mask_bias; a.run_start = run_start; a.run_end = run_end; a.g = g;
    a.part = (bf16_t*)part; a.parg = (unsigned short*)parg; a.ctx = (bf16_t*)ctx; a.M = B * L; a.L = L; a.H = H;
    a.part2 = a.part + (size_t)a.M * H; a.parg2 = a.parg + (size_t)a.M * H;
    a.partA = a.part + 2 * (size_t)a.M * H; a.pargA = a.parg + 2 * (size_t)a.M * H;
    const dim3 grid((a.M + 3) / 4);
    hipLaunchKernelGGL(pn_tree_max_kernel<false>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_tree_max_kernel<true>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_run_fold_max_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_combine_fwd_kernel, grid, dim3(256), 0, s, a);
    return amdseg_launch_status();
}

int amdseg_ponet_pool_bwd_impl(const void* proj, int ld, const float* mask_bias, const int* run_start, const int* run_end,
                               const float* g, const void* part, const void* parg, const void* dctx, void* dproj, void* E,
                               float* psum, int B, int L, int H, hipStream_t s) {
    if (!proj || !mask_bias || !run_start || !run_end || !g || !part || !parg || !dctx || !dproj || !E || !psum) return AMDSEG_ERR_ARG;
    int rc = pn_check(B, L, H, ld);
    if (rc) return rc;
    PnArgs a = {};
    a.proj = (const bf16_t*)proj; a.ld = ld; a.mask_bias = mask_bias; a.run_start = run_start; a.run_end = run_end; a.g = g;
    a.part = (bf16_t*)part; a.parg = (unsigned short*)parg; a.dctx = (const bf16_t*)dctx; a.dproj = (bf16_t*)dproj;
    a.E = (bf16_t*)E; a.psum = psum; a.M = B * L; a.L = L; a.H = H;
    a.part2 = a.part + (size_t)a.M * H; a.parg2 = a.parg + (size_t)a.M * H; a.psum2 = psum + (size_t)a.M * H;
    a.psumA = psum + 2 * (size_t)a.M * H;
    const dim3 grid((a.M + 3) / 4);
    hipLaunchKernelGGL(pn_bwd_token_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_tree_sum_kernel<false>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_tree_sum_kernel<true>, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_run_fold_sum_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(pn_bwd_route_kernel, grid, dim3(256), 0, s, a);
    return amdseg_launch_status();
}
