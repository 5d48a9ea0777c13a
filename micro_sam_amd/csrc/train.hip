// Backward kernels for fine-tuning the mask decoder (reference micro_sam/training/sam_trainer.py:131-425,
// trainable_sam.py:12-114): LayerNorm backward, and softmax attention forward / backward for the
// decoder's shapes (8 heads, head dim 16 or 32, <= 16 prompt tokens on one side and 4096 image tokens or <= 16 tokens on the
// other).  All fp32: these are the small, latency-bound pieces around the GEMMs (which run on the MFMA GEMM kernel in both
// directions: dX = dY W, dW = dY^T X).  No atomics on the data path except the per-column parameter-gradient sums.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);

// ---- reproducible parameter gradients (split-K and the column sums added their partial results with atomicAdd, so every
// fine-tuning run produced a different checkpoint).  Every reduction over workgroups now writes its partial results to a workspace and a
// second launch adds them in a FIXED order (one thread per output element walks the parts 0, 1, 2, ...): the same bits on every run.
// The workspaces are library-owned, one per call site ("slot"), grown on demand (hipMalloc / hipFree synchronise; steady state: no
// allocation); a slot is used by one stream at a time - the trainer's compute stream.
float* msam_det_workspace(size_t floats, int slot) {
    static float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    static size_t cap[4] = {0, 0, 0, 0};
    if (slot < 0 || slot > 3) return nullptr;
    if (cap[slot] < floats) {
        if (buf[slot]) (void)hipFree(buf[slot]);
        buf[slot] = nullptr; cap[slot] = 0;
        const size_t n = floats + floats / 4 + 1024;
        void* pnew = nullptr;
        if (hipMalloc(&pnew, n * sizeof(float)) != hipSuccess) return nullptr;
        buf[slot] = (float*)pnew; cap[slot] = n;
    }
    return buf[slot];
}
__global__ __launch_bounds__(256) void det_reduce_kernel(const float* __restrict__ parts, int nparts, long n, float* __restrict__ out, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    for (int p = 0; p < nparts; ++p) s += parts[(long)p * n + i];
    out[i] = s;
}
// one level of the fixed-order tree: out[g][i] = parts[32 g][i] + parts[32 g + 1][i] + ... (32 consecutive parts, in that order)
__global__ __launch_bounds__(256) void det_reduce_stage_kernel(const float* __restrict__ parts, int nparts, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (i >= n) return;
    const int p0 = g * 32, p1 = p0 + 32 < nparts ? p0 + 32 : nparts;
    float s = 0.f;
    for (int p = p0; p < p1; ++p) s += parts[(long)p * n + i];
    out[(long)g * n + i] = s;
}
// Groups of 32 parts are summed level by level (every level in a fixed order, every element by one thread: the same bits on every run) until at
// most 32 parts are left; returns them.  (One thread per element walking 8192 parts - the bias column sums of a 524288-row product - took
// milliseconds: the first form of round 6 cost fine-tuning 25 % of its step rate.)
const float* msam_det_reduce_tree(const float* parts, int* nparts, long n, void* stream) {
    if (*nparts <= 32) return parts;
    const size_t half = (size_t)((*nparts + 31) / 32) * n;          // the first level's output is the largest
    float* ws = msam_det_workspace(2 * half, 3);
    if (!ws) return nullptr;
    for (int level = 0; *nparts > 32; ++level) {
        const int groups = (*nparts + 31) / 32;
        float* dst = ws + (level & 1) * half;                       // levels alternate between the two halves: a level never writes where it reads
        hipLaunchKernelGGL(det_reduce_stage_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)groups), dim3(256), 0, (hipStream_t)stream, parts, *nparts, n, dst);
        parts = dst; *nparts = groups;
    }
    return parts;
}
// out[i] (+)= the sum of the parts, in a fixed order
void msam_det_reduce(const float* parts, int nparts, long n, float* out, int accumulate, void* stream) {
    parts = msam_det_reduce_tree(parts, &nparts, n, stream);
    if (!parts) return;
    hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, parts, nparts, n, out, accumulate);
}

namespace {

// ---- LayerNorm backward over rows of `dim` <= 256 channels (one wave per row):
//   xhat = (x - mean) * rstd;  g = dy * w;  dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  dw += dy * xhat;  db += dy
template <int V>   // V = dim / 64 values per lane (dim in {64, 128, 256})
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ dy, float eps, long rows,
                                                            float* __restrict__ dx, float* __restrict__ part) {
    // part [gridDim.x][2][DIM]: this workgroup's sums of dy * xhat and dy (added across workgroups in a fixed order by msam_det_reduce)
    constexpr int DIM = V * 64;
    __shared__ float sdw[4][DIM], sdb[4][DIM];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float aw[V], ab[V], wv[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { aw[i] = 0.f; ab[i] = 0.f; wv[i] = w[i * 64 + lane]; }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        float xv[V], gy[V], s = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) { xv[i] = x[row * DIM + i * 64 + lane]; gy[i] = dy[row * DIM + i * 64 + lane]; s += xv[i]; }
        const float mean = wave_sum64(s) * (1.f / DIM);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) { xv[i] -= mean; q += xv[i] * xv[i]; }
        const float rstd = 1.0f / sqrtf(wave_sum64(q) * (1.f / DIM) + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            xv[i] *= rstd;                                   // xhat
            const float g = gy[i] * wv[i];
            sg += g; sgx += g * xv[i];
            aw[i] += gy[i] * xv[i]; ab[i] += gy[i];
        }
        const float mg = wave_sum64(sg) * (1.f / DIM), mgx = wave_sum64(sgx) * (1.f / DIM);
#pragma unroll
        for (int i = 0; i < V; ++i) dx[row * DIM + i * 64 + lane] = rstd * (gy[i] * wv[i] - mg - xv[i] * mgx);
    }
#pragma unroll
    for (int i = 0; i < V; ++i) { sdw[wave][i * 64 + lane] = aw[i]; sdb[wave][i * 64 + lane] = ab[i]; }
    __syncthreads();
    float* const mine = part + (long)blockIdx.x * 2 * DIM;
    for (int c = threadIdx.x; c < DIM; c += 256) {
        mine[c] = sdw[0][c] + sdw[1][c] + sdw[2][c] + sdw[3][c];
        mine[DIM + c] = sdb[0][c] + sdb[1][c] + sdb[2][c] + sdb[3][c];
    }
}

// ---- attention, one thread per query row (forward, dQ) or per key row (dK, dV); q [BH, Nq, D], k / v [BH, Nk, D]
template <int D>
__global__ __launch_bounds__(128) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, int Nq, int Nk, float scale,
                                                       float* __restrict__ out, float* __restrict__ lse) {
    const int bh = blockIdx.y, i = blockIdx.x * 128 + threadIdx.x;
    if (i >= Nq) return;
    float qv[D], acc[D];
    const float* qp = q + ((long)bh * Nq + i) * D;
#pragma unroll
    for (int d = 0; d < D; ++d) { qv[d] = qp[d] * scale; acc[d] = 0.f; }
    const float* kb = k + (long)bh * Nk * D;
    const float* vb = v + (long)bh * Nk * D;
    float m = -3.0e38f, l = 0.f;
    for (int j = 0; j < Nk; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) s = fmaf(qv[d], kb[(long)j * D + d], s);
        const float mn = fmaxf(m, s), a = __expf(m - mn), p = __expf(s - mn);
        l = l * a + p;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fmaf(p, vb[(long)j * D + d], acc[d] * a);
        m = mn;
    }
    const float inv = 1.f / l;
    float* op = out + ((long)bh * Nq + i) * D;
#pragma unroll
    for (int d = 0; d < D; ++d) op[d] = acc[d] * inv;
    lse[(long)bh * Nq + i] = m + __logf(l);
}

// delta_i = sum_d dO_i O_i;  dQ_i = scale * sum_j P_ij (dP_ij - delta_i) K_j,  P_ij = exp(scale q_i k_j - lse_i), dP_ij = dO_i V_j
template <int D>
__global__ __launch_bounds__(128) void attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ out,
                                                         const float* __restrict__ dout, const float* __restrict__ lse, int Nq,
                                                         int Nk, float scale, float* __restrict__ dq, float* __restrict__ delta) {
    const int bh = blockIdx.y, i = blockIdx.x * 128 + threadIdx.x;
    if (i >= Nq) return;
    const long ro = ((long)bh * Nq + i) * D;
    float qv[D], dov[D], acc[D], dl = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { qv[d] = q[ro + d] * scale; dov[d] = dout[ro + d]; dl = fmaf(dov[d], out[ro + d], dl); acc[d] = 0.f; }
    const float L = lse[(long)bh * Nq + i];
    const float* kb = k + (long)bh * Nk * D;
    const float* vb = v + (long)bh * Nk * D;
    for (int j = 0; j < Nk; ++j) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s = fmaf(qv[d], kb[(long)j * D + d], s); dp = fmaf(dov[d], vb[(long)j * D + d], dp); }
        const float ds = __expf(s - L) * (dp - dl);
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fmaf(ds, kb[(long)j * D + d], acc[d]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) dq[ro + d] = acc[d] * scale;
    delta[(long)bh * Nq + i] = dl;
}

// dV_j = sum_i P_ij dO_i;  dK_j = scale * sum_i P_ij (dP_ij - delta_i) Q_i
template <int D>
__global__ __launch_bounds__(128) void attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ delta, int Nq,
                                                          int Nk, float scale, float* __restrict__ dk, float* __restrict__ dv) {
    const int bh = blockIdx.y, j = blockIdx.x * 128 + threadIdx.x;
    if (j >= Nk) return;
    const long ro = ((long)bh * Nk + j) * D;
    float kv[D], vv[D], ak[D], av[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { kv[d] = k[ro + d]; vv[d] = v[ro + d]; ak[d] = 0.f; av[d] = 0.f; }
    const float* qb = q + (long)bh * Nq * D;
    const float* dob = dout + (long)bh * Nq * D;
    for (int i = 0; i < Nq; ++i) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s = fmaf(qb[(long)i * D + d], kv[d], s); dp = fmaf(dob[(long)i * D + d], vv[d], dp); }
        const float p = __expf(s * scale - lse[(long)bh * Nq + i]);
        const float ds = p * (dp - delta[(long)bh * Nq + i]);
#pragma unroll
        for (int d = 0; d < D; ++d) { ak[d] = fmaf(ds, qb[(long)i * D + d], ak[d]); av[d] = fmaf(p, dob[(long)i * D + d], av[d]); }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) { dk[ro + d] = ak[d] * scale; dv[ro + d] = av[d]; }
}

// ---- the same three passes when ONE side of the attention is short (the two-way transformer: 5 - 16 prompt tokens against 4096
// image tokens): one 256-thread workgroup per row of the short side, the loop over the long side split over the threads, results
// merged by a block reduction.  One thread per row left 25 x 8 x 7 = 1 400 threads walking 4096 keys each (1 - 3 ms per call, half of a
// fine-tuning step's kernel time: profiles/r03_train_profile.md).
__device__ __forceinline__ float block_sum256(float v, float* red) {        // red: 4 floats of LDS per concurrent use; all 256 threads call
    v = wave_sum64(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max256(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_rowblock_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, int Nq, int Nk, float scale,
                                                                float* __restrict__ out, float* __restrict__ lse) {
    __shared__ float red[4];
    const int bh = blockIdx.y, i = blockIdx.x;
    float qv[D], acc[D];
    const float* qp = q + ((long)bh * Nq + i) * D;
#pragma unroll
    for (int d = 0; d < D; ++d) { qv[d] = qp[d] * scale; acc[d] = 0.f; }
    const float* kb = k + (long)bh * Nk * D;
    const float* vb = v + (long)bh * Nk * D;
    float m = -3.0e38f, l = 0.f;
    for (int j = threadIdx.x; j < Nk; j += 256) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) s = fmaf(qv[d], kb[(long)j * D + d], s);
        const float mn = fmaxf(m, s), a = __expf(m - mn), p = __expf(s - mn);
        l = l * a + p;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fmaf(p, vb[(long)j * D + d], acc[d] * a);
        m = mn;
    }
    const float M = block_max256(m, red);
    const float r = __expf(m - M);                       // threads without a key: m = -3e38 -> r = 0
    const float L = block_sum256(l * r, red);
    float* op = out + ((long)bh * Nq + i) * D;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float t = block_sum256(acc[d] * r, red);
        if (threadIdx.x == 0) op[d] = t / L;
    }
    if (threadIdx.x == 0) lse[(long)bh * Nq + i] = M + __logf(L);
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_q_rowblock_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                  const float* __restrict__ v, const float* __restrict__ out,
                                                                  const float* __restrict__ dout, const float* __restrict__ lse, int Nq,
                                                                  int Nk, float scale, float* __restrict__ dq, float* __restrict__ delta) {
    __shared__ float red[4];
    const int bh = blockIdx.y, i = blockIdx.x;
    const long ro = ((long)bh * Nq + i) * D;
    float qv[D], dov[D], acc[D], dl = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { qv[d] = q[ro + d] * scale; dov[d] = dout[ro + d]; dl = fmaf(dov[d], out[ro + d], dl); acc[d] = 0.f; }
    const float L = lse[(long)bh * Nq + i];
    const float* kb = k + (long)bh * Nk * D;
    const float* vb = v + (long)bh * Nk * D;
    for (int j = threadIdx.x; j < Nk; j += 256) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s = fmaf(qv[d], kb[(long)j * D + d], s); dp = fmaf(dov[d], vb[(long)j * D + d], dp); }
        const float ds = __expf(s - L) * (dp - dl);
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = fmaf(ds, kb[(long)j * D + d], acc[d]);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float t = block_sum256(acc[d], red);
        if (threadIdx.x == 0) dq[ro + d] = t * scale;
    }
    if (threadIdx.x == 0) delta[(long)bh * Nq + i] = dl;
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_kv_rowblock_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, const float* __restrict__ dout,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta, int Nq,
                                                                   int Nk, float scale, float* __restrict__ dk, float* __restrict__ dv) {
    __shared__ float red[4];
    const int bh = blockIdx.y, j = blockIdx.x;
    const long ro = ((long)bh * Nk + j) * D;
    float kv[D], vv[D], ak[D], av[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { kv[d] = k[ro + d]; vv[d] = v[ro + d]; ak[d] = 0.f; av[d] = 0.f; }
    const float* qb = q + (long)bh * Nq * D;
    const float* dob = dout + (long)bh * Nq * D;
    for (int i = threadIdx.x; i < Nq; i += 256) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s = fmaf(qb[(long)i * D + d], kv[d], s); dp = fmaf(dob[(long)i * D + d], vv[d], dp); }
        const float p = __expf(s * scale - lse[(long)bh * Nq + i]);
        const float ds = p * (dp - delta[(long)bh * Nq + i]);
#pragma unroll
        for (int d = 0; d < D; ++d) { ak[d] = fmaf(ds, qb[(long)i * D + d], ak[d]); av[d] = fmaf(p, dob[(long)i * D + d], av[d]); }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float tk = block_sum256(ak[d], red), tv = block_sum256(av[d], red);
        if (threadIdx.x == 0) { dk[ro + d] = tk * scale; dv[ro + d] = tv; }
    }
}

// ---- attention of the image encoder with its decomposed relative position bias, for fine-tuning the encoder (upstream
// segment_anything/modeling/image_encoder.py add_decomposed_rel_pos; oracle/sam_ref.py _attention_relpos): queries and keys
// are the tokens of ONE Gh x Gw grid (a 14 x 14 window or the 64 x 64 image), key j = (kh, kw) = (j / Gw, j % Gw),
//   s_ij = scale q_i k_j + bias_h[i][kh] + bias_w[i][kw],   P = softmax_j(s),   O = P V.
// bias_h [BH, N, Gh] / bias_w [BH, N, Gw] are the products of the (unscaled) queries with the interpolated rel-pos tables,
// computed by the caller (their gradients flow back to q and to the tables through that product).  Same one-thread-per-row
// fp32 form as the decoder's kernels above; head dim 64 (vit_b / vit_l) or 80 (vit_h).
template <int D>
__global__ __launch_bounds__(128) void relpos_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ bias_h,
                                                         const float* __restrict__ bias_w, int Gh, int Gw, float scale,
                                                         float* __restrict__ out, float* __restrict__ lse) {
    const int N = Gh * Gw;
    const int bh = blockIdx.y, i = blockIdx.x * 128 + threadIdx.x;
    if (i >= N) return;
    const long row = (long)bh * N + i;
    float qv[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { qv[d] = q[row * D + d] * scale; acc[d] = 0.f; }
    const float* kb = k + (long)bh * N * D;
    const float* vb = v + (long)bh * N * D;
    const float* bhp = bias_h + row * Gh;
    const float* bwp = bias_w + row * Gw;
    float m = -3.0e38f, l = 0.f;
    for (int kh = 0; kh < Gh; ++kh) {
        const float bh_ = bhp[kh];
        for (int kw = 0; kw < Gw; ++kw) {
            const long j = (long)kh * Gw + kw;
            float s = bh_ + bwp[kw];
#pragma unroll
            for (int d = 0; d < D; ++d) s = fmaf(qv[d], kb[j * D + d], s);
            const float mn = fmaxf(m, s), a = __expf(m - mn), p = __expf(s - mn);
            l = l * a + p;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = fmaf(p, vb[j * D + d], acc[d] * a);
            m = mn;
        }
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < D; ++d) out[row * D + d] = acc[d] * inv;
    lse[row] = m + __logf(l);
}

// delta_i = sum_d dO_i O_i;  dS_ij = P_ij (dO_i V_j - delta_i);  dQ_i = scale sum_j dS_ij K_j;
// dbias_h[i][kh] = sum_kw dS_i(kh,kw);  dbias_w[i][kw] = sum_kh dS_i(kh,kw)  (the per-thread column of an LDS array)
template <int D>
__global__ __launch_bounds__(128) void relpos_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ bias_h,
                                                           const float* __restrict__ bias_w, const float* __restrict__ out,
                                                           const float* __restrict__ dout, const float* __restrict__ lse, int Gh,
                                                           int Gw, float scale, float* __restrict__ dq, float* __restrict__ dbias_h,
                                                           float* __restrict__ dbias_w, float* __restrict__ delta) {
    __shared__ float sdbw[64 * 128];                           // [kw][thread]
    const int N = Gh * Gw;
    const int bh = blockIdx.y, i = blockIdx.x * 128 + threadIdx.x;
    if (i >= N) return;
    const long row = (long)bh * N + i;
    float qv[D], dov[D], acc[D], dl = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        qv[d] = q[row * D + d] * scale; dov[d] = dout[row * D + d]; dl = fmaf(dov[d], out[row * D + d], dl); acc[d] = 0.f;
    }
    for (int kw = 0; kw < Gw; ++kw) sdbw[kw * 128 + threadIdx.x] = 0.f;
    const float L = lse[row];
    const float* kb = k + (long)bh * N * D;
    const float* vb = v + (long)bh * N * D;
    const float* bhp = bias_h + row * Gh;
    const float* bwp = bias_w + row * Gw;
    for (int kh = 0; kh < Gh; ++kh) {
        const float bh_ = bhp[kh];
        float dbh = 0.f;
        for (int kw = 0; kw < Gw; ++kw) {
            const long j = (long)kh * Gw + kw;
            float s = bh_ + bwp[kw], dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) { s = fmaf(qv[d], kb[j * D + d], s); dp = fmaf(dov[d], vb[j * D + d], dp); }
            const float ds = __expf(s - L) * (dp - dl);
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = fmaf(ds, kb[j * D + d], acc[d]);
            dbh += ds;
            sdbw[kw * 128 + threadIdx.x] += ds;
        }
        dbias_h[row * Gh + kh] = dbh;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) dq[row * D + d] = acc[d] * scale;
    for (int kw = 0; kw < Gw; ++kw) dbias_w[row * Gw + kw] = sdbw[kw * 128 + threadIdx.x];
    delta[row] = dl;
}

// dV_j = sum_i P_ij dO_i;  dK_j = scale sum_i dS_ij Q_i
template <int D>
__global__ __launch_bounds__(128) void relpos_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ bias_h,
                                                            const float* __restrict__ bias_w, const float* __restrict__ dout,
                                                            const float* __restrict__ lse, const float* __restrict__ delta, int Gh,
                                                            int Gw, float scale, float* __restrict__ dk, float* __restrict__ dv) {
    const int N = Gh * Gw;
    const int bh = blockIdx.y, j = blockIdx.x * 128 + threadIdx.x;
    if (j >= N) return;
    const int kh = j / Gw, kw = j - kh * Gw;
    const long ro = ((long)bh * N + j) * D;
    float kv[D], vv[D], ak[D], av[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { kv[d] = k[ro + d]; vv[d] = v[ro + d]; ak[d] = 0.f; av[d] = 0.f; }
    const float* qb = q + (long)bh * N * D;
    const float* dob = dout + (long)bh * N * D;
    for (int i = 0; i < N; ++i) {
        const long row = (long)bh * N + i;
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { s = fmaf(qb[(long)i * D + d], kv[d], s); dp = fmaf(dob[(long)i * D + d], vv[d], dp); }
        const float p = __expf(s * scale + bias_h[row * Gh + kh] + bias_w[row * Gw + kw] - lse[row]);
        const float ds = p * (dp - delta[row]);
#pragma unroll
        for (int d = 0; d < D; ++d) { ak[d] = fmaf(ds, qb[(long)i * D + d], ak[d]); av[d] = fmaf(p, dob[(long)i * D + d], av[d]); }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) { dk[ro + d] = ak[d] * scale; dv[ro + d] = av[d]; }
}

// ---- cast + transpose + column sums of a row-major matrix in one pass (the operands of the weight gradient dW = dY^T X)
// x [M, K] fp32 (SRC16 = false) or bf16 (SRC16 = true), row stride ldx.  One workgroup per 64 x 64 tile:
//   out16 [M, K]  bf16 copy (optional)            - coalesced 8-byte writes,
//   outT  [K, M]  bf16 transpose (optional)       - through a padded LDS tile, 8-byte writes along M,
//   colsum        fp32 column sums (optional)     - per-workgroup partial sums (fixed order inside the workgroup) to part [gridDim.y][K]; the
//                                                   launcher adds the row blocks in order (msam_det_reduce) INTO the caller's zeroed vector.
// The unfused form was three to four torch launches per operand (cast, strided transpose copy at 0.5 TB/s, sum): 30 % of a fine-tuning
// step's device time (profiles/r03_experiments.md section 8).  K % 4 == 0, ldx % 4 == 0; M arbitrary.
template <bool SRC16>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const void* __restrict__ xin, long M, int K, long ldx, unsigned short* __restrict__ out16,
                                                             unsigned short* __restrict__ outT, float* __restrict__ colsum) {
    __shared__ unsigned short tile[64][68];          // [m][k], rows padded to 136 B
    __shared__ float csum[16][64];                   // [row group r0][column]
    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * 64;
    const long m0 = (long)blockIdx.y * 64;
    const int c4 = (tid & 15) * 4, r0 = tid >> 4;     // 4 consecutive columns, rows r0 + 16 p
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = r0 + 16 * p;
        const long m = m0 + r;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (m < M && k0 + c4 < K) {
            if (SRC16) {
                const uint2 u = *(const uint2*)((const unsigned short*)xin + m * ldx + k0 + c4);
                v0 = __uint_as_float(u.x << 16); v1 = __uint_as_float(u.x & 0xffff0000u);
                v2 = __uint_as_float(u.y << 16); v3 = __uint_as_float(u.y & 0xffff0000u);
            } else {
                const float4 f = *(const float4*)((const float*)xin + m * ldx + k0 + c4);
                v0 = f.x; v1 = f.y; v2 = f.z; v3 = f.w;
            }
        }
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
        const unsigned short h0 = f2bf(v0), h1 = f2bf(v1), h2 = f2bf(v2), h3 = f2bf(v3);
        tile[r][c4] = h0; tile[r][c4 + 1] = h1; tile[r][c4 + 2] = h2; tile[r][c4 + 3] = h3;
        if (out16 && m < M && k0 + c4 < K) {
            uint2 pk; pk.x = (unsigned)h0 | ((unsigned)h1 << 16); pk.y = (unsigned)h2 | ((unsigned)h3 << 16);
            *(uint2*)(out16 + m * (long)K + k0 + c4) = pk;
        }
    }
    if (colsum) { csum[r0][c4] = s0; csum[r0][c4 + 1] = s1; csum[r0][c4 + 2] = s2; csum[r0][c4 + 3] = s3; }
    __syncthreads();
    if (outT) {
        const int mq = (tid & 15) * 4, kr0 = tid >> 4;   // 4 consecutive rows m of the source = 4 consecutive columns of outT
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int kk = kr0 + 16 * p;
            if (k0 + kk >= K) continue;
            const long m = m0 + mq;
            unsigned short* dst = outT + (long)(k0 + kk) * M + m;
            if (m + 3 < M && (M & 3) == 0) {
                uint2 pk;
                pk.x = (unsigned)tile[mq][kk] | ((unsigned)tile[mq + 1][kk] << 16);
                pk.y = (unsigned)tile[mq + 2][kk] | ((unsigned)tile[mq + 3][kk] << 16);
                *(uint2*)dst = pk;
            } else {
                for (int i = 0; i < 4; ++i)
                    if (m + i < M) dst[i] = tile[mq + i][kk];
            }
        }
    }
    if (colsum && tid < 64 && k0 + tid < K) {       // colsum here = the partial-sum workspace [gridDim.y][K]
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += csum[r][tid];
        colsum[(long)blockIdx.y * K + k0 + tid] = t;
    }
}

__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nparts, int dim, float* __restrict__ dw, float* __restrict__ db) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * dim) return;
    float* const dst = i < dim ? dw + i : db + (i - dim);
    float s = *dst;                                  // (the reference accumulates parameter gradients: `+=` as the atomic form did)
    for (int p = 0; p < nparts; ++p) s += part[(long)p * 2 * dim + i];
    *dst = s;
}

}  // namespace

extern "C" int msam_layernorm_backward(const float* x, const float* weight, const float* dy, float eps, int64_t rows, int32_t dim,
                                       float* dx, float* dweight, float* dbias, void* stream) {
    if (!x || !weight || !dy || !dx || !dweight || !dbias || rows <= 0) { msam_set_error("msam_layernorm_backward: bad argument"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
    if (dim != 64 && dim != 128 && dim != 256 && dim != 768 && dim != 1024 && dim != 1280) { msam_set_error("msam_layernorm_backward: dim must be 64, 128, 256, 768, 1024 or 1280"); return 1; }
    float* part = msam_det_workspace((size_t)grid * 2 * dim, 0);
    if (!part) { msam_set_error("msam_layernorm_backward: cannot allocate the partial-sum workspace"); return 2; }
    if (dim == 256) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(grid), dim3(256), 0, s, x, weight, dy, eps, (long)rows, dx, part);
    else if (dim == 128) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(grid), dim3(256), 0, s, x, weight, dy, eps, (long)rows, dx, part);
    else if (dim == 64) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, x, weight, dy, eps, (long)rows, dx, part);
    // the image encoder's widths (vit_b / vit_l / vit_h), for un-frozen fine-tuning
    else if (dim == 768) hipLaunchKernelGGL(layernorm_bwd_kernel<12>, dim3(grid), dim3(256), 0, s, x, weight, dy, eps, (long)rows, dx, part);
    else if (dim == 1024) hipLaunchKernelGGL(layernorm_bwd_kernel<16>, dim3(grid), dim3(256), 0, s, x, weight, dy, eps, (long)rows, dx, part);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<20>, dim3(grid), dim3(256), 0, s, x, weight, dy, eps, (long)rows, dx, part);
    // dweight / dbias += the workgroups' sums, in a fixed order (the pair [dw | db] of a workgroup is 2 dim contiguous floats)
    int nparts = grid;
    const float* red = msam_det_reduce_tree(part, &nparts, 2L * dim, stream);
    if (!red) { msam_set_error("msam_layernorm_backward: cannot allocate the reduction workspace"); return 2; }
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * dim + 255) / 256), dim3(256), 0, s, red, nparts, dim, dweight, dbias);
    return msam_check_launch("msam_layernorm_backward");
}

extern "C" int msam_attention_forward(const float* q, const float* k, const float* v, int32_t BH, int32_t Nq, int32_t Nk,
                                      int32_t D, float scale, float* out, float* lse, void* stream) {
    if (!q || !k || !v || !out || !lse || BH <= 0 || Nq <= 0 || Nk <= 0) { msam_set_error("msam_attention_forward: bad argument"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    if (D != 16 && D != 32) { msam_set_error("msam_attention_forward: head dim must be 16 or 32"); return 1; }
    if (Nq <= 64 && Nk >= 1024) {                    // few queries, many keys: a workgroup per query row (see the kernels)
        const dim3 grid(Nq, BH);
        if (D == 16) hipLaunchKernelGGL(attn_fwd_rowblock_kernel<16>, grid, dim3(256), 0, s, q, k, v, Nq, Nk, scale, out, lse);
        else hipLaunchKernelGGL(attn_fwd_rowblock_kernel<32>, grid, dim3(256), 0, s, q, k, v, Nq, Nk, scale, out, lse);
        return msam_check_launch("msam_attention_forward");
    }
    const dim3 grid((Nq + 127) / 128, BH);
    if (D == 16) hipLaunchKernelGGL(attn_fwd_kernel<16>, grid, dim3(128), 0, s, q, k, v, Nq, Nk, scale, out, lse);
    else hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, dim3(128), 0, s, q, k, v, Nq, Nk, scale, out, lse);
    return msam_check_launch("msam_attention_forward");
}

extern "C" int msam_attention_backward(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                       const float* lse, int32_t BH, int32_t Nq, int32_t Nk, int32_t D, float scale, float* dq,
                                       float* dk, float* dv, float* delta, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !delta || BH <= 0 || Nq <= 0 || Nk <= 0) {
        msam_set_error("msam_attention_backward: bad argument");
        return 1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (D != 16 && D != 32) { msam_set_error("msam_attention_backward: head dim must be 16 or 32"); return 1; }
    const dim3 gq((Nq + 127) / 128, BH), gk((Nk + 127) / 128, BH);
    const bool q_rows = Nq <= 64 && Nk >= 1024;     // a workgroup per query row for (delta, dQ)
    const bool k_rows = Nk <= 64 && Nq >= 1024;     // a workgroup per key row for (dK, dV)
#define ATTN_BWD(D_)                                                                                                          \
    do {                                                                                                                      \
        if (q_rows) hipLaunchKernelGGL(attn_bwd_q_rowblock_kernel<D_>, dim3(Nq, BH), dim3(256), 0, s, q, k, v, out, dout, lse, Nq, Nk, scale, dq, delta); \
        else hipLaunchKernelGGL(attn_bwd_q_kernel<D_>, gq, dim3(128), 0, s, q, k, v, out, dout, lse, Nq, Nk, scale, dq, delta);     \
        if (k_rows) hipLaunchKernelGGL(attn_bwd_kv_rowblock_kernel<D_>, dim3(Nk, BH), dim3(256), 0, s, q, k, v, dout, lse, delta, Nq, Nk, scale, dk, dv); \
        else hipLaunchKernelGGL(attn_bwd_kv_kernel<D_>, gk, dim3(128), 0, s, q, k, v, dout, lse, delta, Nq, Nk, scale, dk, dv);      \
    } while (0)
    if (D == 16) ATTN_BWD(16); else ATTN_BWD(32);
#undef ATTN_BWD
    return msam_check_launch("msam_attention_backward");
}

extern "C" int msam_relpos_attention_forward(const float* q, const float* k, const float* v, const float* bias_h,
                                             const float* bias_w, int32_t BH, int32_t Gh, int32_t Gw, int32_t D, float scale,
                                             float* out, float* lse, void* stream) {
    if (!q || !k || !v || !bias_h || !bias_w || !out || !lse || BH <= 0 || Gh <= 0 || Gw <= 0 || Gw > 64) {
        msam_set_error("msam_relpos_attention_forward: bad argument (grid width <= 64)");
        return 1;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((Gh * Gw + 127) / 128, BH);
    if (D == 64) hipLaunchKernelGGL(relpos_fwd_kernel<64>, grid, dim3(128), 0, s, q, k, v, bias_h, bias_w, Gh, Gw, scale, out, lse);
    else if (D == 80) hipLaunchKernelGGL(relpos_fwd_kernel<80>, grid, dim3(128), 0, s, q, k, v, bias_h, bias_w, Gh, Gw, scale, out, lse);
    else { msam_set_error("msam_relpos_attention_forward: head dim must be 64 or 80"); return 1; }
    return msam_check_launch("msam_relpos_attention_forward");
}

extern "C" int msam_relpos_attention_backward(const float* q, const float* k, const float* v, const float* bias_h,
                                              const float* bias_w, const float* out, const float* dout, const float* lse,
                                              int32_t BH, int32_t Gh, int32_t Gw, int32_t D, float scale, float* dq, float* dk,
                                              float* dv, float* dbias_h, float* dbias_w, float* delta, void* stream) {
    if (!q || !k || !v || !bias_h || !bias_w || !out || !dout || !lse || !dq || !dk || !dv || !dbias_h || !dbias_w || !delta ||
        BH <= 0 || Gh <= 0 || Gw <= 0 || Gw > 64) {
        msam_set_error("msam_relpos_attention_backward: bad argument (grid width <= 64)");
        return 1;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((Gh * Gw + 127) / 128, BH);
    if (D == 64) {
        hipLaunchKernelGGL(relpos_bwd_q_kernel<64>, grid, dim3(128), 0, s, q, k, v, bias_h, bias_w, out, dout, lse, Gh, Gw, scale, dq,
                           dbias_h, dbias_w, delta);
        hipLaunchKernelGGL(relpos_bwd_kv_kernel<64>, grid, dim3(128), 0, s, q, k, v, bias_h, bias_w, dout, lse, delta, Gh, Gw, scale,
                           dk, dv);
    } else if (D == 80) {
        hipLaunchKernelGGL(relpos_bwd_q_kernel<80>, grid, dim3(128), 0, s, q, k, v, bias_h, bias_w, out, dout, lse, Gh, Gw, scale, dq,
                           dbias_h, dbias_w, delta);
        hipLaunchKernelGGL(relpos_bwd_kv_kernel<80>, grid, dim3(128), 0, s, q, k, v, bias_h, bias_w, dout, lse, delta, Gh, Gw, scale,
                           dk, dv);
    } else { msam_set_error("msam_relpos_attention_backward: head dim must be 64 or 80"); return 1; }
    return msam_check_launch("msam_relpos_attention_backward");
}

extern "C" int msam_cast_transpose(const void* x, int32_t x_dtype, int64_t M, int32_t K, int64_t ldx, void* out16, void* outT, float* colsum,
                                   void* stream) {
    if (!x || M <= 0 || K <= 0 || (K & 3) || (ldx & 3) || ldx < K) { msam_set_error("msam_cast_transpose: need K % 4 == 0, ldx % 4 == 0, ldx >= K"); return 1; }
    if (x_dtype != MSAM_F32 && x_dtype != MSAM_BF16) { msam_set_error("msam_cast_transpose: source fp32 or bf16"); return 1; }
    if ((M + 63) / 64 > 65535) { msam_set_error("msam_cast_transpose: M <= 4 194 240"); return 1; }
    const dim3 grid((K + 63) / 64, (unsigned)((M + 63) / 64));
    float* part = nullptr;
    if (colsum) {
        part = msam_det_workspace((size_t)grid.y * K, 1);
        if (!part) { msam_set_error("msam_cast_transpose: cannot allocate the partial-sum workspace"); return 2; }
    }
    if (x_dtype == MSAM_F32)
        hipLaunchKernelGGL(cast_transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)M, K, (long)ldx, (unsigned short*)out16,
                           (unsigned short*)outT, part);
    else
        hipLaunchKernelGGL(cast_transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)M, K, (long)ldx, (unsigned short*)out16,
                           (unsigned short*)outT, part);
    if (colsum) msam_det_reduce(part, (int)grid.y, K, colsum, 1, stream);       // colsum += row block 0 + row block 1 + ... (the caller zeroes it)
    return msam_check_launch("msam_cast_transpose");
}
