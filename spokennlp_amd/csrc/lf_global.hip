// Longformer global [CLS] row: the O(heads * H^2) algebra on [B, heads, H] vectors that surrounds the O(L) passes of longformer.hip
// ([hf] models/longformer/modeling_longformer.py:964-1058; called through longformer_for_ts.py:55-89 with [CLS] as the only global
// token).  Round 1 issued it as ~25 tiny rocBLAS / elementwise launches per layer and direction (6 % of the longformer-base step in
// launches of a few microseconds); here it is 2 kernels forward and 4 backward, each a handful of small matrix-vector products with the
// weights read once (2.4 MB per H x H matrix, fp32 masters: the global row is computed in fp32).
//   forward :  qg = (Wq x0 + bq) / 8 ;  r_h = Wk_h^T qg_h            (lf_global_q)    -> scores = r . x_j  (longformer.hip)
//              out_h = Wv_h y_h + bv_h * sum_j p_j  -> ctx[:, 0]      (lf_global_out)
//   backward:  dout = dctx[:, 0] (then zeroed); dyv_h = Wv_h^T dout_h ; dsp_h = bv_h . dout_h            (lf_global_bwd_a)
//              dqg_h = Wk_h dr_h / 8                                                                      (lf_global_bwd_b)
//              dWv += dout (x) y, dbv += dout * sp, dWk += qg (x) dr  (one workgroup per weight row)     (lf_global_bwd_w)
//              dWq += dqg (x) x0, dbq += dqg ; dx[:, 0] += Wq^T dqg                                        (lf_global_bwd_q)
#include "common.h"
#include "amdseg_internal.h"

struct LfGArgs {
    const void* x; int x_dtype; int L, H, heads, B;        // layer input [B*L, H]; row b*L is the [CLS] row
    const float* Wq; const float* bq; const float* Wk; const float* Wv; const float* bv;
    float* qg; float* r;                                   // [B, heads, 64], [B, heads, H]
    const float* y; const float* sp;                       // [B, heads, H], [B, heads]
    void* ctx; int ctx_dtype;                              // [B*L, H]
    void* dctx; float* dout; float* dyv; float* dsp;       // backward a
    const float* dr; float* dqg;                           // backward b
    float* dWq; float* dbq; float* dWk; float* dWv; float* dbv;
    void* dx; int dx_dtype;
    float scale;
    int keep_dctx;                                         // lf_global_bwd_a: 1 = read dctx[:, 0] only (do not zero it)
    int qpart;                                             // lf_global_bwd_q: 0 = weight rows + dx rows, 1 = weight rows only, 2 = dx rows only
    float* trow;                                           // dx rows, optional [B, H]: Wq^T dqg[b] is WRITTEN here instead of added to dx[b, 0, :]
};

template <typename T> __device__ __forceinline__ float ldf(const void* p, size_t i) { return Act<T>::ld(reinterpret_cast<const T*>(p) + i); }
__device__ __forceinline__ float ld_any(const void* p, size_t i, int dtype) { return dtype == AMDSEG_BF16 ? ldf<bf16_t>(p, i) : ldf<float>(p, i); }
__device__ __forceinline__ void st_any(void* p, size_t i, float v, int dtype) {
    if (dtype == AMDSEG_BF16) reinterpret_cast<bf16_t*>(p)[i] = f2bf(v); else reinterpret_cast<float*>(p)[i] = v;
}
// dot of a weight row (fp32, length H, H % 4 == 0) with a vector held in LDS, by one wave
__device__ __forceinline__ float wave_dot_lds(const float* __restrict__ wrow, const float* vec, int H, int l) {
    float s = 0.f;
    for (int k = l * 4; k < H; k += 256) {
        const float4 w = *reinterpret_cast<const float4*>(wrow + k);
        s += w.x * vec[k] + w.y * vec[k + 1] + w.z * vec[k + 2] + w.w * vec[k + 3];
    }
    return wave_sum(s);
}

// four consecutive weight rows against the same LDS vector: the 12 row loads of a lane are in flight together (one row at a time left
// each 768-long dot exposed to a full memory round trip: 29 us for 64 rows)
__device__ __forceinline__ void wave_dot4_lds(const float* __restrict__ w0, int H, const float* vec, int l, float (&out)[4]) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = l * 4; k < H; k += 256) {
        float4 w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const float4*>(w0 + (size_t)j * H + k);
        const float v0 = vec[k], v1 = vec[k + 1], v2 = vec[k + 2], v3 = vec[k + 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += w[j].x * v0 + w[j].y * v1 + w[j].z * v2 + w[j].w * v3;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = wave_sum(s[j]);
}

// eight consecutive weight rows at a time (24 row loads of a lane in flight at H = 768): lf_global_q heads the forward chain of every layer
__device__ __forceinline__ void wave_dot8_lds(const float* __restrict__ w0, int H, const float* vec, int l, float (&out)[8]) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (H <= 768) {                                          // every trip's loads issued before the first use
        float4 w[3][8];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int k = l * 4 + t * 256;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[t][j] = k < H ? *reinterpret_cast<const float4*>(w0 + (size_t)j * H + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int k = l * 4 + t * 256;
            if (k < H) {
                const float v0 = vec[k], v1 = vec[k + 1], v2 = vec[k + 2], v3 = vec[k + 3];
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += w[t][j].x * v0 + w[t][j].y * v1 + w[t][j].z * v2 + w[t][j].w * v3;
            }
        }
    } else {
        for (int k = l * 4; k < H; k += 256) {
            float4 w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const float4*>(w0 + (size_t)j * H + k);
            const float v0 = vec[k], v1 = vec[k + 1], v2 = vec[k + 2], v3 = vec[k + 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += w[j].x * v0 + w[j].y * v1 + w[j].z * v2 + w[j].w * v3;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = wave_sum(s[j]);
}

// grid (heads, B, H / 256), 256 threads: every block recomputes its head's qg (64 short dots), block z writes r columns z*256 ..
__global__ __launch_bounds__(256) void lf_global_q_kernel(LfGArgs a) {
    extern __shared__ float sm[];                        // [H] x0 | [64] qg
    float* x0 = sm; float* q = sm + a.H;
    const int h = blockIdx.x, b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int k = threadIdx.x; k < a.H; k += 256) x0[k] = ld_any(a.x, (size_t)b * a.L * a.H + k, a.x_dtype);
    __syncthreads();
    for (int e = w * 16; e < w * 16 + 16; e += 8) {
        const int row = h * 64 + e;
        float v[8];
        wave_dot8_lds(a.Wq + (size_t)row * a.H, a.H, x0, l, v);
        if (l == 0)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = (v[j] + a.bq[row + j]) * a.scale;
                q[e + j] = t;
                if (blockIdx.z == 0) a.qg[((size_t)b * a.heads + h) * 64 + e + j] = t;
            }
    }
    __syncthreads();
    const int c = blockIdx.z * 256 + threadIdx.x;
    if (c < a.H) {
        // the lane's 64 strided Wk elements in four batches of 16 loads in flight (same four partial sums, same order of additions as before)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float* wk = a.Wk + (size_t)(h * 64) * a.H + c;
#pragma unroll
        for (int e0 = 0; e0 < 64; e0 += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = wk[(size_t)(e0 + u) * a.H];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                s0 = fmaf(q[e0 + u], wv[u], s0); s1 = fmaf(q[e0 + u + 1], wv[u + 1], s1);
                s2 = fmaf(q[e0 + u + 2], wv[u + 2], s2); s3 = fmaf(q[e0 + u + 3], wv[u + 3], s3);
            }
        }
        a.r[((size_t)b * a.heads + h) * a.H + c] = (s0 + s1) + (s2 + s3);
    }
}

// grid (heads, B): out_h = Wv_h y_h + bv_h * sp -> ctx row of [CLS]
__global__ __launch_bounds__(256) void lf_global_out_kernel(LfGArgs a) {
    extern __shared__ float sm[];                        // [H] y
    const int h = blockIdx.x, b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t bh = (size_t)b * a.heads + h;
    for (int k = threadIdx.x; k < a.H; k += 256) sm[k] = a.y[bh * a.H + k];
    __syncthreads();
    const float spv = a.sp[bh];
    for (int e = w * 16; e < w * 16 + 16; e += 8) {
        const int row = h * 64 + e;
        float v[8];
        wave_dot8_lds(a.Wv + (size_t)row * a.H, a.H, sm, l, v);
        if (l == 0)
#pragma unroll
            for (int j = 0; j < 8; ++j) st_any(a.ctx, (size_t)b * a.L * a.H + row + j, v[j] + a.bv[row + j] * spv, a.ctx_dtype);
    }
}

// grid (heads, B): consumes + zeroes dctx[:, 0]; dyv, dsp, dout
__global__ __launch_bounds__(256) void lf_global_bwd_a_kernel(LfGArgs a) {
    __shared__ float d[64];
    const int h = blockIdx.x, b = blockIdx.y;
    const size_t bh = (size_t)b * a.heads + h;
    if (threadIdx.x < 64) {
        const size_t i = (size_t)b * a.L * a.H + h * 64 + threadIdx.x;
        const float v = ld_any(a.dctx, i, a.ctx_dtype);
        d[threadIdx.x] = v;
        a.dout[bh * 64 + threadIdx.x] = v;
        if (!a.keep_dctx) st_any(a.dctx, i, 0.f, a.ctx_dtype);    // the band attention's own row 0 was overwritten in forward: no gradient
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.H; c += 256) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float* wv = a.Wv + (size_t)(h * 64) * a.H + c;
#pragma unroll 4
        for (int e = 0; e < 64; e += 4) {
            s0 = fmaf(d[e], wv[(size_t)e * a.H], s0); s1 = fmaf(d[e + 1], wv[(size_t)(e + 1) * a.H], s1);
            s2 = fmaf(d[e + 2], wv[(size_t)(e + 2) * a.H], s2); s3 = fmaf(d[e + 3], wv[(size_t)(e + 3) * a.H], s3);
        }
        a.dyv[bh * a.H + c] = (s0 + s1) + (s2 + s3);
    }
    if (threadIdx.x < 64) {
        float s = d[threadIdx.x] * a.bv[h * 64 + threadIdx.x];
        s = wave_sum(s);
        if (threadIdx.x == 0) a.dsp[bh] = s;
    }
}

// grid (heads, B): dqg_h = Wk_h dr_h * scale
__global__ __launch_bounds__(256) void lf_global_bwd_b_kernel(LfGArgs a) {
    extern __shared__ float sm[];                        // [H] dr
    const int h = blockIdx.x, b = blockIdx.y, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const size_t bh = (size_t)b * a.heads + h;
    for (int k = threadIdx.x; k < a.H; k += 256) sm[k] = a.dr[bh * a.H + k];
    __syncthreads();
    for (int e = w * 16; e < w * 16 + 16; e += 4) {
        float v[4];
        wave_dot4_lds(a.Wk + (size_t)(h * 64 + e) * a.H, a.H, sm, l, v);
        if (l == 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) a.dqg[(size_t)b * a.H + h * 64 + e + j] = v[j] * a.scale;
    }
}

// grid (H): weight row i = h*64 + e:  dWv[i,:] += sum_b dout[b,i] y[b,h,:] ; dbv[i] += sum_b dout[b,i] sp[b,h] ; dWk[i,:] += sum_b qg[b,i] dr[b,h,:]
__global__ __launch_bounds__(256) void lf_global_bwd_w_kernel(LfGArgs a) {
    const int i = blockIdx.x, h = i >> 6;
    for (int c = threadIdx.x; c < a.H; c += 256) {
        float sv = 0.f, sk = 0.f;
        for (int b = 0; b < a.B; ++b) {
            const size_t bh = (size_t)b * a.heads + h;
            sv = fmaf(a.dout[bh * 64 + (i & 63)], a.y[bh * a.H + c], sv);
            sk = fmaf(a.qg[bh * 64 + (i & 63)], a.dr[bh * a.H + c], sk);
        }
        a.dWv[(size_t)i * a.H + c] += sv;
        a.dWk[(size_t)i * a.H + c] += sk;
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < a.B; ++b) s = fmaf(a.dout[((size_t)b * a.heads + h) * 64 + (i & 63)], a.sp[(size_t)b * a.heads + h], s);
        a.dbv[i] += s;
    }
}

// grid (H + B * H / 64): blocks [0, H): dWq[i,:] += sum_b dqg[b,i] x0[b,:], dbq[i] += sum_b dqg[b,i];  the others: dx[b, 0, 64 columns] += Wq^T dqg[b]
// (thread = (column, quarter of the rows), the four quarters summed through LDS -- one block per sequence left 8 workgroups walking 768 x 768: 254 us)
__global__ __launch_bounds__(256) void lf_global_bwd_q_kernel(LfGArgs a) {
    extern __shared__ float sm[];                        // [H] dqg[b]
    const int blk = a.qpart == 2 ? (int)blockIdx.x + a.H : (int)blockIdx.x;
    if (blk < a.H) {
        const int i = blk;
        for (int c = threadIdx.x; c < a.H; c += 256) {
            float s = 0.f;
            for (int b = 0; b < a.B; ++b) s = fmaf(a.dqg[(size_t)b * a.H + i], ld_any(a.x, (size_t)b * a.L * a.H + c, a.x_dtype), s);
            a.dWq[(size_t)i * a.H + c] += s;
        }
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int b = 0; b < a.B; ++b) s += a.dqg[(size_t)b * a.H + i];
            a.dbq[i] += s;
        }
        return;
    }
    __shared__ float red[4][64];
    const int nb = a.H / 64, id = blk - a.H, b = id / nb, c = (id % nb) * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < a.H; k += 256) sm[k] = a.dqg[(size_t)b * a.H + k];
    __syncthreads();
    const int i0 = part * (a.H / 4), i1 = i0 + a.H / 4;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = i0; i < i1; i += 4) {
        s0 = fmaf(sm[i], a.Wq[(size_t)i * a.H + c], s0); s1 = fmaf(sm[i + 1], a.Wq[(size_t)(i + 1) * a.H + c], s1);
        s2 = fmaf(sm[i + 2], a.Wq[(size_t)(i + 2) * a.H + c], s2); s3 = fmaf(sm[i + 3], a.Wq[(size_t)(i + 3) * a.H + c], s3);
    }
    red[part][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (part == 0) {
        const size_t o = (size_t)b * a.L * a.H + c;
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (a.trow) a.trow[(size_t)b * a.H + c] = t;
        else st_any(a.dx, o, ld_any(a.dx, o, a.dx_dtype) + t, a.dx_dtype);
    }
}

static int lfg_check(int B, int L, int H, int heads) {
    if (B <= 0 || L <= 0 || heads <= 0 || H != heads * 64 || (H % 16)) return AMDSEG_ERR_SHAPE;
    return AMDSEG_OK;
}

int amdseg_lf_global_q_impl(const void* x, int x_dtype, const float* Wq, const float* bq, const float* Wk, float* qg, float* r, int B, int L,
                            int H, int heads, float scale, hipStream_t s) {
    if (!x || !Wq || !bq || !Wk || !qg || !r) return AMDSEG_ERR_ARG;
    int rc = lfg_check(B, L, H, heads);
    if (rc) return rc;
    LfGArgs a = {};
    a.x = x; a.x_dtype = x_dtype; a.L = L; a.H = H; a.heads = heads; a.B = B; a.Wq = Wq; a.bq = bq; a.Wk = Wk; a.qg = qg; a.r = r; a.scale = scale;
    hipLaunchKernelGGL(lf_global_q_kernel, dim3(heads, B, (H + 255) / 256), dim3(256), (H + 64) * sizeof(float), s, a);
    return amdseg_launch_status();
}
int amdseg_lf_global_out_impl(const float* Wv, const float* bv, const float* y, const float* sp, void* ctx, int ctx_dtype, int B, int L, int H,
                              int heads, hipStream_t s) {
    if (!Wv || !bv || !y || !sp || !ctx) return AMDSEG_ERR_ARG;
    int rc = lfg_check(B, L, H, heads);
    if (rc) return rc;
    LfGArgs a = {};
    a.L = L; a.H = H; a.heads = heads; a.B = B; a.Wv = Wv; a.bv = bv; a.y = y; a.sp = sp; a.ctx = ctx; a.ctx_dtype = ctx_dtype;
    hipLaunchKernelGGL(lf_global_out_kernel, dim3(heads, B), dim3(256), H * sizeof(float), s, a);
    return amdseg_launch_status();
}
int amdseg_lf_global_bwd_a_impl(void* dctx, int dtype, const float* Wv, const float* bv, float* dout, float* dyv, float* dsp, int B, int L,
                                int H, int heads, hipStream_t s, int keep_dctx) {
    if (!dctx || !Wv || !bv || !dout || !dyv || !dsp) return AMDSEG_ERR_ARG;
    int rc = lfg_check(B, L, H, heads);
    if (rc) return rc;
    LfGArgs a = {};
    a.L = L; a.H = H; a.heads = heads; a.B = B; a.Wv = Wv; a.bv = bv; a.dctx = dctx; a.ctx_dtype = dtype; a.dout = dout; a.dyv = dyv; a.dsp = dsp;
    a.keep_dctx = keep_dctx;
    hipLaunchKernelGGL(lf_global_bwd_a_kernel, dim3(heads, B), dim3(256), 0, s, a);
    return amdseg_launch_status();
}
int amdseg_lf_global_bwd_rest_impl(const void* x, int x_dtype, void* dx, int dx_dtype, const float* Wq, const float* Wk, const float* qg,
                                   const float* dout, const float* y, const float* sp, const float* dr, float* dqg, float* dWq, float* dbq,
                                   float* dWk, float* dWv, float* dbv, int B, int L, int H, int heads, float scale, hipStream_t s) {
    if (!x || !dx || !Wq || !Wk || !qg || !dout || !y || !sp || !dr || !dqg || !dWq || !dbq || !dWk || !dWv || !dbv) return AMDSEG_ERR_ARG;
    int rc = lfg_check(B, L, H, heads);
    if (rc) return rc;
    LfGArgs a = {};
    a.x = x; a.x_dtype = x_dtype; a.dx = dx; a.dx_dtype = dx_dtype; a.L = L; a.H = H; a.heads = heads; a.B = B; a.Wq = Wq; a.Wk = Wk;
    a.qg = (float*)qg; a.dout = (float*)dout; a.y = y; a.sp = sp; a.dr = dr; a.dqg = dqg; a.dWq = dWq; a.dbq = dbq; a.dWk = dWk; a.dWv = dWv;
    a.dbv = dbv; a.scale = scale;
    hipLaunchKernelGGL(lf_global_bwd_b_kernel, dim3(heads, B), dim3(256), H * sizeof(float), s, a);
    hipLaunchKernelGGL(lf_global_bwd_w_kernel, dim3(H), dim3(256), 0, s, a);
    hipLaunchKernelGGL(lf_global_bwd_q_kernel, dim3(H + B * (H / 64)), dim3(256), H * sizeof(float), s, a);
    return amdseg_launch_status();
}

// the same backward in two calls, for a caller that wants the part its critical path waits for early and the weight gradients whenever:
// _dx: dqg (lf_global_bwd_b) and trow[b, :] = Wq^T dqg[b], the global row's contribution to dx[b, 0, :] -- WRITTEN to trow [B, H] fp32, to be added
// by amdseg_lf_dx_apply;  _w: every weight / bias gradient of the three global projections (needs dqg from _dx).
int amdseg_lf_global_bwd_dx_impl(const float* Wq, const float* Wk, const float* dr, float* dqg, float* trow, int B, int L, int H, int heads,
                                 float scale, hipStream_t s) {
    if (!Wq || !Wk || !dr || !dqg || !trow) return AMDSEG_ERR_ARG;
    int rc = lfg_check(B, L, H, heads);
    if (rc) return rc;
    LfGArgs a = {};
    a.L = L; a.H = H; a.heads = heads; a.B = B; a.Wq = Wq; a.Wk = Wk; a.dr = dr; a.dqg = dqg; a.trow = trow; a.scale = scale; a.qpart = 2;
    hipLaunchKernelGGL(lf_global_bwd_b_kernel, dim3(heads, B), dim3(256), H * sizeof(float), s, a);
    hipLaunchKernelGGL(lf_global_bwd_q_kernel, dim3(B * (H / 64)), dim3(256), H * sizeof(float), s, a);
    return amdseg_launch_status();
}
int amdseg_lf_global_bwd_w_impl(const void* x, int x_dtype, const float* qg, const float* dout, const float* y, const float* sp, const float* dr,
                                const float* dqg, float* dWq, float* dbq, float* dWk, float* dWv, float* dbv, int B, int L, int H, int heads,
                                hipStream_t s) {
    if (!x || !qg || !dout || !y || !sp || !dr || !dqg || !dWq || !dbq || !dWk || !dWv || !dbv) return AMDSEG_ERR_ARG;
    int rc = lfg_check(B, L, H, heads);
    if (rc) return rc;
    LfGArgs a = {};
    a.x = x; a.x_dtype = x_dtype; a.L = L; a.H = H; a.heads = heads; a.B = B; a.qg = (float*)qg; a.dout = (float*)dout; a.y = y; a.sp = sp; a.dr = dr;
    a.dqg = (float*)dqg; a.dWq = dWq; a.dbq = dbq; a.dWk = dWk; a.dWv = dWv; a.dbv = dbv; a.qpart = 1;
    hipLaunchKernelGGL(lf_global_bwd_w_kernel, dim3(H), dim3(256), 0, s, a);
    hipLaunchKernelGGL(lf_global_bwd_q_kernel, dim3(H), dim3(256), H * sizeof(float), s, a);
    return amdseg_launch_status();
}
