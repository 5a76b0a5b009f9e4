// Fused loss heads of the topic-segmentation wrapper (training path): the token cross-entropy over the classifier logits, the CSSL
// InfoNCE over the labelled [BOS] rows and the TSSP linear + cross-entropy, forward and backward in ONE C-ABI call each.
//
// Reference (emnlp2023-topic_segmentation/src/models/modules):
//   loss_calculator.py:25-73   loss = ts_w * CE(logits, labels; weight, ignore -100) [+ cl_w * cssl (anchor half)] [+ tssp_w * tssp (DA half)]
//   utils.py:141-182           weighted CE = sum_i w[y_i] nll_i / sum_i w[y_i];  focal_loss_gamma != 0: FocalLoss overwrites its own `reduction`
//                              with 'mean' before super().forward runs, so what it returns is  mean_i (1 - p_i,t_i)^gamma  x  that SCALAR mean CE
//                              (t_i = the label, 0 on ignored rows; the mean runs over ALL rows) -- reproduced as is
//   cssl.py:82-116,118-228     per anchor i: -log( sum_{l < pk} e^{cos(a_i, x_l,i)/tau} / sum_l e^{cos(a_i, x_l,i)/tau} ), mean over anchors
//                              (lists built on the host with the reference's `random` call order: bert_for_ts.py::_plan_cssl)
//   tssp.py:16-36              rows at sent_token_mask != -100 -> Linear(H, 3) -> CE (mean); weighted tssp_w twice (tssp.py:36 x :71)
// The torch formulation of these heads ran as ~130 launches of a few microseconds per step (0.5 ms of a 15 ms step); here the forward
// is 4 kernels and the backward 4 (+ the classifier's rowdot kernels), none of which synchronises with the host.  Work per step is
// tiny (a few hundred rows of H floats), so the kernels are simple one-wave-per-row loops; what matters is the launch count.
#include "common.h"
#include "amdseg_internal.h"

#define HEADS_MAXC 4
#define HEADS_MAXSEG 2
#define HEADS_MAXLIST 16

// out[]: 0,1 CE mean per segment; 2 CSSL; 3 TSSP (unweighted CE mean); 4 total loss; 5,6 1 / sum of CE weights per segment
struct HeadsArgs {
    const float* x; int M, H;
    const float* logits; const int64_t* labels; const float* class_w; int C, nseg, rows_per_seg;
    float* ce_unit; float* out;
    const int64_t* feat_rows; const int64_t* anchor_idx; const int64_t* lists; int n_anchor, n_list, pk; float inv_temp;
    const float* Wt; const float* bt; const int64_t* t_rows; const int64_t* t_labels; int nt, Ct;
    float w_ts, w_cl, w_tssp2;
    float gamma;           // focal_loss_gamma; != 0: ce_unit is [M, 2C] (CE unit | focal-factor unit), out[8..9] = mean focal factor, out[10..11] = mean CE per segment
    // backward
    const float* gout; float* dx; float* dWt; float* dbt; float* dlogits;
    float* row_loss;       // forward scratch: per-anchor CSSL terms [n_anchor] followed by per-row TSSP terms [nt]
    // backward scatter target: fixed-point sums, [n_feat][H] CSSL feature rows | [nt][H] TSSP rows | [Ct][H] dWt | [Ct] dbt
    unsigned long long* fix; int n_feat;
};

// The row gradients of CSSL / TSSP are scatter sums (a feature row sits in the lists of many anchors; dWt sums over every TSSP row).
// Float atomics would make their value depend on the arrival order; the contributions are instead rounded to 2^-40 and summed as 64-bit
// integers -- integer addition commutes, so the sums (and with them the step) are bit-reproducible -- and converted back by the single
// writer of each destination (heads_fix_apply_kernel).  Supported range: |contribution| < 2^20 (these are loss gradients of O(1) rows; a loss scale
// of 65536 still fits); a contribution outside it -- or a NaN / Inf of a diverged step, which __float2ll_rn would turn into 0 / saturate -- is
// reported by fix_add, and the calling wave then writes NaN into the gradient row it was adding to (heads_poison): the apply pass adds on top
// of it, so a diverged step reaches the gradient norm as NaN exactly as it did with float atomics.
#define HEADS_FIX_SCALE 1099511627776.0f            // 2^40
__device__ __forceinline__ bool fix_add(unsigned long long* p, float v) {
    const bool bad = !(fabsf(v) < 1048576.0f);                                   // also true for NaN
    atomicAdd(p, (unsigned long long)__float2ll_rn(bad ? 0.f : v * HEADS_FIX_SCALE));
    return bad;
}
__device__ __forceinline__ void heads_poison(bool bad, float* row, int l) {     // wave-wide: any lane's bad contribution poisons the row
    if (__any(bad)) row[l] = __builtin_nanf("");
}
// dst[(rows ? rows[slot] : slot)][d] += fix[slot][d] * 2^-40; one wave per slot, each destination element has one writer per launch
__global__ __launch_bounds__(256) void heads_fix_apply_kernel(const unsigned long long* fix, const int64_t* rows, int n, int H, float* dst) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + w;
    if (i >= n) return;
    float* o = dst + (size_t)(rows ? rows[i] : i) * H;
    const unsigned long long* f = fix + (size_t)i * H;
    for (int d = l; d < H; d += 64) {
        const long long q = (long long)f[d];
        if (q) o[d] += (float)((double)q * (1.0 / 1099511627776.0));
    }
}

// ---------------------------------------------------------------------------------------------------- token cross-entropy
// one thread per row; per-wave (num, den) partials per segment, summed in a fixed order by the finalize kernel (no atomics: the loss
// is bit-reproducible run to run)
__global__ __launch_bounds__(256) void heads_ce_fwd_kernel(HeadsArgs a, float* acc) {      // acc[wave][seg*2 + {0: sum w nll, 1: sum w}]
    const int i = blockIdx.x * 256 + threadIdx.x;
    float num = 0.f, den = 0.f, foc = 0.f;
    int seg = 0;
    const int us = a.gamma != 0.f ? 2 * a.C : a.C;          // row stride of ce_unit
    if (i < a.M) {
        seg = i / a.rows_per_seg;
        const int64_t y = a.labels[i];
        float lg[HEADS_MAXC], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < HEADS_MAXC; ++c) if (c < a.C) { lg[c] = a.logits[(size_t)i * a.C + c]; mx = fmaxf(mx, lg[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < HEADS_MAXC; ++c) if (c < a.C) se += expf(lg[c] - mx);
        const float lse = mx + logf(se);
        const bool valid = y >= 0 && y < a.C;
        const float w = valid ? (a.class_w ? a.class_w[y] : 1.0f) : 0.f;
#pragma unroll
        for (int c = 0; c < HEADS_MAXC; ++c)
            if (c < a.C) a.ce_unit[(size_t)i * us + c] = valid ? w * (expf(lg[c] - lse) - (c == y ? 1.0f : 0.0f)) : 0.f;
        float ly = 0.f;
#pragma unroll
        for (int c = 0; c < HEADS_MAXC; ++c) if (c < a.C && c == y) ly = lg[c];
        if (valid) { num = w * (lse - ly); den = w; }
        if (a.gamma != 0.f) {                               // focal factor of the row: (1 - p_t)^gamma, t = label (0 on ignored rows)
            const int t = valid ? (int)y : 0;
            float lt = lg[0];
#pragma unroll
            for (int c = 1; c < HEADS_MAXC; ++c) if (c < a.C && c == t) lt = lg[c];
            const float pt = expf(lt - lse), om = fmaxf(1.0f - pt, 0.f);
            foc = powf(om, a.gamma);
            const float dfac = om > 0.f ? a.gamma * powf(om, a.gamma - 1.0f) * pt : 0.f;      // d foc / d z_c = dfac * (p_c - [c == t])
#pragma unroll
            for (int c = 0; c < HEADS_MAXC; ++c)
                if (c < a.C) a.ce_unit[(size_t)i * 2 * a.C + a.C + c] = dfac * (expf(lg[c] - lse) - (c == t ? 1.0f : 0.0f));
        }
    }
    // rows of one wave may straddle a segment boundary only if rows_per_seg is not a multiple of 64: reduce per segment
#pragma unroll
    for (int sg = 0; sg < HEADS_MAXSEG; ++sg) {
        if (sg >= a.nseg) break;
        const float n1 = wave_sum(seg == sg ? num : 0.f), d1 = wave_sum(seg == sg ? den : 0.f), f1 = wave_sum(seg == sg ? foc : 0.f);
        if ((threadIdx.x & 63) == 0) {
            float* slot = acc + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + sg * 4;
            slot[0] = n1; slot[1] = d1; slot[2] = f1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- CSSL (list form)
__device__ __forceinline__ float row_dot(const float* __restrict__ p, const float* __restrict__ q, int H, int l) {
    float s = 0.f;
    for (int d = l * 4; d < H; d += 256) {
        const float4 u = *reinterpret_cast<const float4*>(p + d), v = *reinterpret_cast<const float4*>(q + d);
        s += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
    }
    return wave_sum(s);
}
// one wave per anchor.  BWD = false: adds its -log ratio / n to out[2].  BWD = true: scatter-adds coef * d(loss_i)/d(rows) into dx.
template <bool BWD>
__global__ __launch_bounds__(256) void heads_cssl_kernel(HeadsArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + w;
    if (i >= a.n_anchor) return;
    const int64_t fa = a.anchor_idx ? a.anchor_idx[i] : i;
    const int64_t ra = a.feat_rows[fa];
    const float* xa = a.x + (size_t)ra * a.H;
    const float na = fmaxf(sqrtf(row_dot(xa, xa, a.H, l)), 1e-8f);
    float cs[HEADS_MAXLIST], no[HEADS_MAXLIST], e[HEADS_MAXLIST];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < HEADS_MAXLIST; ++k) {
        if (k >= a.n_list) break;
        const float* xo = a.x + (size_t)a.feat_rows[a.lists[(size_t)k * a.n_anchor + i]] * a.H;
        no[k] = fmaxf(sqrtf(row_dot(xo, xo, a.H, l)), 1e-8f);
        cs[k] = row_dot(xa, xo, a.H, l) / (na * no[k]);
        mx = fmaxf(mx, cs[k] * a.inv_temp);
    }
    float sall = 0.f, spos = 0.f;
#pragma unroll
    for (int k = 0; k < HEADS_MAXLIST; ++k) {
        if (k >= a.n_list) break;
        e[k] = expf(cs[k] * a.inv_temp - mx);
        sall += e[k];
        if (k < a.pk) spos += e[k];
    }
    if (!BWD) {
        if (l == 0) a.row_loss[i] = -logf(spos / sall);
        return;
    }
    // d loss_i / d cos_k = inv_temp * (e_k / sall - [k < pk] e_k / spos) / n ; chain through the cosine into both rows
    const float g = a.gout[0] * a.w_cl / (float)a.n_anchor * a.inv_temp;
    unsigned long long* dxa = a.fix + (size_t)fa * a.H;
#pragma unroll
    for (int k = 0; k < HEADS_MAXLIST; ++k) {
        if (k >= a.n_list) break;
        const float dc = g * (e[k] / sall - (k < a.pk ? e[k] / spos : 0.f));
        const int64_t fo = a.lists[(size_t)k * a.n_anchor + i];
        const float* xo = a.x + (size_t)a.feat_rows[fo] * a.H;
        unsigned long long* dxo = a.fix + (size_t)fo * a.H;
        const float ca = dc / (na * no[k]), cva = dc * cs[k] / (na * na), cvo = dc * cs[k] / (no[k] * no[k]);
        bool bad = false;
        for (int d = l; d < a.H; d += 64) {
            const float va = xa[d], vo = xo[d];
            bad |= fix_add(dxa + d, ca * vo - cva * va);
            bad |= fix_add(dxo + d, ca * va - cvo * vo);
        }
        heads_poison(bad, a.dx + (size_t)ra * a.H, l);
    }
}

// ---------------------------------------------------------------------------------------------------- TSSP
template <bool BWD>
__global__ __launch_bounds__(256) void heads_tssp_kernel(HeadsArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + w;
    if (i >= a.nt) return;
    const int64_t r = a.t_rows[i];
    const float* xr = a.x + (size_t)r * a.H;
    float lg[HEADS_MAXC], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < HEADS_MAXC; ++c) {
        if (c >= a.Ct) break;
        lg[c] = row_dot(xr, a.Wt + (size_t)c * a.H, a.H, l) + a.bt[c];
        mx = fmaxf(mx, lg[c]);
    }
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < HEADS_MAXC; ++c) if (c < a.Ct) se += expf(lg[c] - mx);
    const float lse = mx + logf(se);
    const int64_t y = a.t_labels[i];
    if (!BWD) {
        float ly = 0.f;
#pragma unroll
        for (int c = 0; c < HEADS_MAXC; ++c) if (c < a.Ct && c == y) ly = lg[c];
        if (l == 0) a.row_loss[a.n_anchor + i] = lse - ly;
        return;
    }
    const float g = a.gout[0] * a.w_tssp2 / (float)a.nt;
    float dl[HEADS_MAXC];
#pragma unroll
    for (int c = 0; c < HEADS_MAXC; ++c) if (c < a.Ct) dl[c] = g * (expf(lg[c] - lse) - (c == y ? 1.0f : 0.0f));
    unsigned long long* dxr = a.fix + (size_t)(a.n_feat + i) * a.H;       // this row's own slot: a plain store would do, kept uniform
    unsigned long long* dW = a.fix + (size_t)(a.n_feat + a.nt) * a.H;
    bool bad = false;
    for (int d = l; d < a.H; d += 64) {
        const float xv = xr[d];
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < HEADS_MAXC; ++c)
            if (c < a.Ct) { acc += dl[c] * a.Wt[(size_t)c * a.H + d]; bad |= fix_add(dW + (size_t)c * a.H + d, dl[c] * xv); }
        bad |= fix_add(dxr + d, acc);
    }
    if (l == 0)
#pragma unroll
        for (int c = 0; c < HEADS_MAXC; ++c) if (c < a.Ct) bad |= fix_add(dW + (size_t)a.Ct * a.H + c, dl[c]);
    heads_poison(bad, a.dx + (size_t)r * a.H, l);
}

// ---------------------------------------------------------------------------------------------------- combine / CE backward
// block-wide sum in a fixed order: thread t adds elements t, t + 256, ...; then the wave / LDS tree
__device__ __forceinline__ float block_sum_fixed(const float* p, int n, int stride, float* red) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += p[(size_t)i * stride];
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void heads_finalize_kernel(HeadsArgs a, const float* acc, int nwaves) {
    __shared__ float red[4];
    float tot = 0.f;
    const float cssl = a.n_anchor > 0 ? block_sum_fixed(a.row_loss, a.n_anchor, 1, red) / (float)a.n_anchor : 0.f;
    const float tssp = a.nt > 0 ? block_sum_fixed(a.row_loss + a.n_anchor, a.nt, 1, red) / (float)a.nt : 0.f;
    float num[HEADS_MAXSEG], dens[HEADS_MAXSEG], focs[HEADS_MAXSEG];
    for (int sg = 0; sg < a.nseg; ++sg) {
        num[sg] = block_sum_fixed(acc + sg * 4, nwaves, 8, red);
        dens[sg] = block_sum_fixed(acc + sg * 4 + 1, nwaves, 8, red);
        focs[sg] = a.gamma != 0.f ? block_sum_fixed(acc + sg * 4 + 2, nwaves, 8, red) : 0.f;
    }
    if (threadIdx.x != 0) return;
    a.out[2] = cssl; a.out[3] = tssp;
    for (int sg = 0; sg < a.nseg; ++sg) {
        const float den = dens[sg];
        float ce = den != 0.f ? num[sg] / den : 0.f / 0.f;     // torch: mean over an empty set = nan
        a.out[5 + sg] = den != 0.f ? 1.0f / den : 0.f;
        if (a.gamma != 0.f) {                                   // focal: mean factor x mean CE (see the header)
            // a segment without a labelled row: the reference's FocalLoss returns a constant 0 when its mean CE is NaN
            // (modules/utils.py:150-156) -- no loss and no gradient from that segment
            const float fbar = den != 0.f ? focs[sg] / (float)a.rows_per_seg : 0.f;
            if (den == 0.f) ce = 0.f;
            a.out[8 + sg] = fbar; a.out[10 + sg] = ce;
            ce *= fbar;
        }
        a.out[sg] = ce;
        tot += a.w_ts * ce;
    }
    tot += a.w_cl * a.out[2] + a.w_tssp2 * a.out[3];
    a.out[4] = tot;
}
__global__ __launch_bounds__(256) void heads_ce_bwd_kernel(HeadsArgs a) {
    const size_t n = (size_t)a.M * a.C;
    const float g = a.gout[0] * a.w_ts;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t row = i / a.C;
        const int seg = (int)(row / a.rows_per_seg), c = (int)(i - row * a.C);
        if (a.gamma == 0.f) { a.dlogits[i] = a.ce_unit[i] * g * a.out[5 + seg]; continue; }
        // d (fbar * ce) = fbar * d ce + ce * d fbar
        const float* u = a.ce_unit + row * 2 * a.C;
        a.dlogits[i] = g * (a.out[8 + seg] * u[c] * a.out[5 + seg] + a.out[10 + seg] * u[a.C + c] / (float)a.rows_per_seg);
    }
}

static int heads_check(const HeadsArgs& a) {
    if (!a.x || !a.logits || !a.labels || !a.ce_unit || !a.out) return AMDSEG_ERR_ARG;
    if (a.M <= 0 || a.H <= 0 || (a.H % 4) || a.C < 1 || a.C > HEADS_MAXC || a.nseg < 1 || a.nseg > HEADS_MAXSEG || a.rows_per_seg <= 0) return AMDSEG_ERR_SHAPE;
    if ((long)a.nseg * a.rows_per_seg != a.M) return AMDSEG_ERR_SHAPE;
    if (a.n_anchor > 0 && (!a.feat_rows || !a.lists || a.n_list < 1 || a.n_list > HEADS_MAXLIST || a.pk < 1 || a.pk > a.n_list)) return AMDSEG_ERR_ARG;
    if (a.nt > 0 && (!a.Wt || !a.bt || !a.t_rows || !a.t_labels || a.Ct < 1 || a.Ct > HEADS_MAXC)) return AMDSEG_ERR_ARG;
    return AMDSEG_OK;
}

static HeadsArgs heads_fill(const float* x, int M, int H, const float* logits, const int64_t* labels, const float* class_w, int C, int nseg,
                            float* ce_unit, float* out, const int64_t* idx, long feat_off, long anchor_off, long lists_off, int n_anchor,
                            int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off, long t_labels_off, int nt,
                            int Ct, float w_ts, float w_cl, float w_tssp2) {
    HeadsArgs a = {};
    a.x = x; a.M = M; a.H = H; a.logits = logits; a.labels = labels; a.class_w = class_w; a.C = C; a.nseg = nseg;
    a.rows_per_seg = nseg > 0 ? M / nseg : 0; a.ce_unit = ce_unit; a.out = out;
    a.feat_rows = (idx && n_anchor > 0) ? idx + feat_off : nullptr;
    a.anchor_idx = (idx && n_anchor > 0 && anchor_off >= 0) ? idx + anchor_off : nullptr;
    a.lists = (idx && n_anchor > 0) ? idx + lists_off : nullptr;
    a.n_anchor = n_anchor; a.n_list = n_list; a.pk = pk; a.inv_temp = temp != 0.f ? 1.0f / temp : 1.0f;
    a.Wt = Wt; a.bt = bt; a.t_rows = (idx && nt > 0) ? idx + t_rows_off : nullptr; a.t_labels = (idx && nt > 0) ? idx + t_labels_off : nullptr;
    a.nt = nt; a.Ct = Ct; a.w_ts = w_ts; a.w_cl = w_cl; a.w_tssp2 = w_tssp2;
    return a;
}

int amdseg_heads_fwd_impl(const float* x, int M, int H, const float* logits, const int64_t* labels, const float* class_w, int C, int nseg,
                          float* ce_unit, float* out8, float* acc, const int64_t* idx, long feat_off, long anchor_off, long lists_off,
                          int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off,
                          long t_labels_off, int nt, int Ct, float w_ts, float w_cl, float w_tssp2, hipStream_t s, float focal_gamma) {
    HeadsArgs a = heads_fill(x, M, H, logits, labels, class_w, C, nseg, ce_unit, out8, idx, feat_off, anchor_off, lists_off, n_anchor, n_list,
                             pk, temp, Wt, bt, t_rows_off, t_labels_off, nt, Ct, w_ts, w_cl, w_tssp2);
    if (!acc) return AMDSEG_ERR_ARG;
    a.gamma = focal_gamma;
    int rc = heads_check(a);
    if (rc) return rc;
    // acc: [4 * ceil(M / 256)][8] per-wave CE partials (per segment: sum w nll, sum w, sum focal factor, -), then n_anchor + nt per-row loss terms (amdseg.h: amdseg_heads_acc_floats)
    const int blocks = (M + 255) / 256, nwaves = blocks * 4;
    a.row_loss = acc + (size_t)nwaves * 8;
    hipLaunchKernelGGL(heads_ce_fwd_kernel, dim3(blocks), dim3(256), 0, s, a, acc);
    if (n_anchor > 0) hipLaunchKernelGGL(heads_cssl_kernel<false>, dim3((n_anchor + 3) / 4), dim3(256), 0, s, a);
    if (nt > 0) hipLaunchKernelGGL(heads_tssp_kernel<false>, dim3((nt + 3) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL(heads_finalize_kernel, dim3(1), dim3(256), 0, s, a, (const float*)acc, nwaves);
    return amdseg_launch_status();
}

// backward: writes dlogits [M,C] (the caller runs the classifier's rowdot backward on it, which WRITES dx), then adds the CSSL / TSSP row
// gradients into dx and ACCUMULATES dWt / dbt (zeroed by the caller)
int amdseg_heads_bwd_ce_impl(const float* gout, int M, int C, int nseg, const float* ce_unit, const float* out8, float w_ts, float* dlogits,
                             hipStream_t s, float focal_gamma) {
    if (!gout || !ce_unit || !out8 || !dlogits) return AMDSEG_ERR_ARG;
    if (M <= 0 || C < 1 || C > HEADS_MAXC || nseg < 1 || nseg > HEADS_MAXSEG || (M % nseg)) return AMDSEG_ERR_SHAPE;
    HeadsArgs a = {};
    a.M = M; a.C = C; a.nseg = nseg; a.rows_per_seg = M / nseg; a.ce_unit = (float*)ce_unit; a.out = (float*)out8; a.w_ts = w_ts;
    a.gout = gout; a.dlogits = dlogits; a.gamma = focal_gamma;
    size_t blocks = ((size_t)M * C + 255) / 256;
    hipLaunchKernelGGL(heads_ce_bwd_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, s, a);
    return amdseg_launch_status();
}
int amdseg_heads_bwd_rows_impl(const float* gout, const float* x, int M, int H, float* dx, const int64_t* idx, long feat_off, long anchor_off,
                               long lists_off, int n_anchor, int n_list, int pk, float temp, const float* Wt, const float* bt, long t_rows_off,
                               long t_labels_off, int nt, int Ct, float* dWt, float* dbt, float w_cl, float w_tssp2, int n_feat, void* fix,
                               size_t fix_bytes, hipStream_t s) {
    if (!gout || !x || !dx || !fix) return AMDSEG_ERR_ARG;
    if (n_anchor <= 0) n_feat = 0;
    if (nt < 0 || n_feat < 0 || (n_anchor > 0 && n_feat < 1)) return AMDSEG_ERR_ARG;
    const size_t fix_elems = (size_t)(n_feat + nt) * H + (nt > 0 ? (size_t)Ct * H + Ct : 0);
    if (fix_bytes < fix_elems * sizeof(unsigned long long)) return AMDSEG_ERR_ARG;
    if (M <= 0 || H <= 0 || (H % 4)) return AMDSEG_ERR_SHAPE;
    HeadsArgs a = heads_fill(x, M, H, nullptr, nullptr, nullptr, 1, 1, nullptr, nullptr, idx, feat_off, anchor_off, lists_off, n_anchor, n_list,
                             pk, temp, Wt, bt, t_rows_off, t_labels_off, nt, Ct, 0.f, w_cl, w_tssp2);
    if (n_anchor > 0 && (!a.feat_rows || !a.lists || n_list < 1 || n_list > HEADS_MAXLIST || pk < 1 || pk > n_list)) return AMDSEG_ERR_ARG;
    if (nt > 0 && (!Wt || !bt || !dWt || !dbt || Ct < 1 || Ct > HEADS_MAXC)) return AMDSEG_ERR_ARG;
    a.gout = gout; a.dx = dx; a.dWt = dWt; a.dbt = dbt; a.fix = (unsigned long long*)fix; a.n_feat = n_feat;
    if (fix_elems == 0) return AMDSEG_OK;
    if (hipMemsetAsync(fix, 0, fix_elems * sizeof(unsigned long long), s) != hipSuccess) return AMDSEG_ERR_LAUNCH;
    if (n_anchor > 0) hipLaunchKernelGGL(heads_cssl_kernel<true>, dim3((n_anchor + 3) / 4), dim3(256), 0, s, a);
    if (nt > 0) hipLaunchKernelGGL(heads_tssp_kernel<true>, dim3((nt + 3) / 4), dim3(256), 0, s, a);
    // the feature rows of one head are distinct, but a row may serve both heads: one launch per head keeps every dx element single-writer
    if (n_anchor > 0)
        hipLaunchKernelGGL(heads_fix_apply_kernel, dim3((n_feat + 3) / 4), dim3(256), 0, s, a.fix, a.feat_rows, n_feat, H, dx);
    if (nt > 0) {
        const unsigned long long* ft = a.fix + (size_t)n_feat * H;
        hipLaunchKernelGGL(heads_fix_apply_kernel, dim3((nt + 3) / 4), dim3(256), 0, s, ft, a.t_rows, nt, H, dx);
        hipLaunchKernelGGL(heads_fix_apply_kernel, dim3((Ct + 3) / 4), dim3(256), 0, s, ft + (size_t)nt * H, (const int64_t*)nullptr, Ct, H, dWt);
        hipLaunchKernelGGL(heads_fix_apply_kernel, dim3(1), dim3(256), 0, s, ft + (size_t)nt * H + (size_t)Ct * H, (const int64_t*)nullptr, 1, Ct, dbt);
    }
    return amdseg_launch_status();
}
