// SAM prompt encoder + two-way-transformer mask decoder for P prompts on one image embedding.
//
// Dataflow (bf16 MFMA operands, fp32 accumulation / LayerNorm / softmax; see DESIGN.md "decoder"):
//   per model   : dense positional encoding pos[4096,256] and its projections pos*Wk^T, pos*Wq^T for every
//                 image-side attention (added as GEMM "table" epilogues: (x+pos)W = xW + posW).
//   per image   : src = embedding^T + no_mask_embed;  layer-0 K|V^T and image-to-token Q (prompt independent).
//   per prompt  : token-side chain on [P*Nt,256] rows through the shared GEMM kernel; image-side stream
//                 keys[P*4096,256] (bf16) updated by image->token attention + out_proj + LN4; token->image
//                 attention is a flash-style MFMA kernel over the 4096 image tokens; up-scaling =
//                 ConvT(2x2)->LN2d->GELU as GEMM + row-LN(64), ConvT(2x2)->GELU->hyper-network product fused.
#include <stdlib.h>
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);

namespace {

constexpr int T = 4096;        // image tokens
constexpr int C = 256;         // transformer dim
constexpr int CI = 128;        // internal dim of the cross attentions
constexpr float NEG_BIG = -1.0e30f;
constexpr float TWO_PI = 6.283185307179586f;

// ------------------------------------------------------------------------------------------ small kernels

// PositionEmbeddingRandom: pe(c) = [sin(2pi * ((2c-1) @ G)), cos(...)], c in [0,1]^2
MSAM_DEVINL void pe_encode(const float* __restrict__ G, float cx, float cy, int j, float& s, float& c) {
    const float x = 2.f * cx - 1.f, y = 2.f * cy - 1.f;
    const float v = TWO_PI * (x * G[j] + y * G[128 + j]);
    s = sinf(v); c = cosf(v);
}

__global__ __launch_bounds__(256) void dense_pe_kernel(const float* __restrict__ G, float* __restrict__ pos_f32,
                                                       u16* __restrict__ pos_bf16) {
    const int idx = blockIdx.x * 256 + threadIdx.x;      // over 4096 * 128
    if (idx >= T * 128) return;
    const int t = idx >> 7, j = idx & 127;
    const int ty = t >> 6, tx = t & 63;
    float s, c;
    pe_encode(G, (tx + 0.5f) / 64.f, (ty + 0.5f) / 64.f, j, s, c);
    pos_f32[t * C + j] = s; pos_f32[t * C + 128 + j] = c;
    pos_bf16[t * C + j] = f2d(s); pos_bf16[t * C + 128 + j] = f2d(c);
}

// tokens[p][0..4] = output tokens; then points (+0.5, label embeddings), padding point when no box, box corners.
__global__ __launch_bounds__(256) void prompt_tokens_kernel(
    const float* __restrict__ G, const float* __restrict__ point_embed, const float* __restrict__ not_a_point,
    const float* __restrict__ out_tokens, const float* __restrict__ points, const int* __restrict__ labels, int Np,
    const float* __restrict__ boxes, int P, int Nt, float* __restrict__ tokens, float* __restrict__ queries,
    u16* __restrict__ queries_bf16) {
    const int p = blockIdx.x, c = threadIdx.x;           // 256 threads = 256 channels
    if (p >= P) return;
    float* tp = tokens + (long)p * Nt * C;
    for (int i = 0; i < 5; ++i) tp[i * C + c] = out_tokens[i * C + c];
    const int j = c & 127;
    int row = 5;
    for (int i = 0; i < Np; ++i, ++row) {
        const float px = points[((long)p * Np + i) * 2] + 0.5f, py = points[((long)p * Np + i) * 2 + 1] + 0.5f;
        const int lab = labels[(long)p * Np + i];
        float s, co;
        pe_encode(G, px / 1024.f, py / 1024.f, j, s, co);
        float v = c < 128 ? s : co;
        if (lab == -1) v = not_a_point[c];
        else if (lab == 0) v += point_embed[0 * C + c];
        else if (lab == 1) v += point_embed[1 * C + c];
        tp[row * C + c] = v;
    }
    if (Np > 0 && !boxes) { tp[row * C + c] = not_a_point[c]; ++row; }     // padding point (0,0), label -1
    if (boxes) {
        for (int k = 0; k < 2; ++k, ++row) {
            const float bx = boxes[(long)p * 4 + 2 * k] + 0.5f, by = boxes[(long)p * 4 + 2 * k + 1] + 0.5f;
            float s, co;
            pe_encode(G, bx / 1024.f, by / 1024.f, j, s, co);
            tp[row * C + c] = (c < 128 ? s : co) + point_embed[(2 + k) * C + c];
        }
    }
    // the transformer starts from queries = tokens (fp32 copy + the bf16 operand of the first projections)
    for (int i = 0; i < Nt; ++i) {
        const float v = tp[i * C + c];                   // written by this thread above
        queries[((long)p * Nt + i) * C + c] = v;
        queries_bf16[((long)p * Nt + i) * C + c] = f2d(v);
    }
}

MSAM_DEVINL float gelu_exact_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Mask prompts (PromptEncoder._embed_masks, segment_anything/modeling/prompt_encoder.py: mask_downscaling = Conv 2x2/2
// (1->4), LayerNorm2d, GELU, Conv 2x2/2 (4->16), LayerNorm2d, GELU, Conv 1x1 (16->256)): the per-prompt decoder source
// src_p = image embedding + dense prompt embedding, written as the bf16 image-token stream [P, 4096, 256].
// grid (64 token rows, P); phase 1: 64 threads = the tokens of the row, 4x4 input pixels -> 16 channels;
// phase 2: 256 threads = output channels.
// dense_out != NULL: write the dense prompt embedding itself, fp32 NCHW [P, 256, 64, 64] (PromptEncoder.forward(masks=...))
__global__ __launch_bounds__(256) void mask_src_kernel(const float* __restrict__ mask, msam_mask_prompt_t mp,
                                                       const float* __restrict__ src_f32, const float* __restrict__ no_mask,
                                                       u16* __restrict__ keys, float* __restrict__ dense_out) {
    __shared__ float h2s[64][17];
    const int ty = blockIdx.x, p = blockIdx.y, tid = threadIdx.x;
    if (tid < 64) {
        const int tx = tid;
        const float* m = mask + (long)p * 65536 + (long)(4 * ty) * 256 + 4 * tx;
        float px[4][4];
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const float4 r = *(const float4*)(m + y * 256);
            px[y][0] = r.x; px[y][1] = r.y; px[y][2] = r.z; px[y][3] = r.w;
        }
        float h1[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[4], mean = 0.f;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    float a = mp.c1_b[ch];
#pragma unroll
                    for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 2; ++kx) a = fmaf(mp.c1_w[ch * 4 + ky * 2 + kx], px[2 * i + ky][2 * j + kx], a);
                    v[ch] = a; mean += a;
                }
                mean *= 0.25f;
                float var = 0.f;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) { v[ch] -= mean; var += v[ch] * v[ch]; }
                const float rstd = 1.0f / sqrtf(var * 0.25f + 1e-6f);
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) { const float z_ = v[ch] * rstd * mp.ln1_w[ch] + mp.ln1_b[ch]; h1[i][j][ch] = mp.exact_gelu ? gelu_exact_f(z_) : gelu_erf(z_); }
            }
        float v2[16], mean = 0.f;
#pragma unroll
        for (int co = 0; co < 16; ++co) {
            float a = mp.c2_b[co];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 2; ++kx) a = fmaf(mp.c2_w[((co * 4 + ci) * 2 + ky) * 2 + kx], h1[ky][kx][ci], a);
            v2[co] = a; mean += a;
        }
        mean *= (1.f / 16.f);
        float var = 0.f;
#pragma unroll
        for (int co = 0; co < 16; ++co) { v2[co] -= mean; var += v2[co] * v2[co]; }
        const float rstd = 1.0f / sqrtf(var * (1.f / 16.f) + 1e-6f);
#pragma unroll
        for (int co = 0; co < 16; ++co) { const float z_ = v2[co] * rstd * mp.ln2_w[co] + mp.ln2_b[co]; h2s[tx][co] = mp.exact_gelu ? gelu_exact_f(z_) : gelu_erf(z_); }
    }
    __syncthreads();
    const int c = tid;
    float w3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w3[k] = mp.c3_w[c * 16 + k];
    const float base = dense_out ? mp.c3_b[c] : mp.c3_b[c] - no_mask[c];          // src_f32 already contains + no_mask_embed
    for (int t = 0; t < 64; ++t) {
        float a = base;
#pragma unroll
        for (int k = 0; k < 16; ++k) a = fmaf(w3[k], h2s[t][k], a);
        const long token = (long)ty * 64 + t;
        if (dense_out) dense_out[((long)p * C + c) * T + token] = a;
        else keys[((long)p * T + token) * C + c] = f2d(src_f32[token * C + c] + a);
    }
}

// src[t][c] = emb[c][t] + no_mask[c]  (NCHW -> token-major); bf16 copy
__global__ __launch_bounds__(256) void src_prepare_kernel(const float* __restrict__ emb, const float* __restrict__ no_mask,
                                                          float* __restrict__ src_f32, u16* __restrict__ src_bf16) {
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int i = ty; i < 32; i += 8) tile[i][tx] = emb[(long)(c0 + i) * T + t0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const float v = tile[tx][i] + no_mask[c0 + tx];
        src_f32[(long)(t0 + i) * C + c0 + tx] = v;
        src_bf16[(long)(t0 + i) * C + c0 + tx] = f2d(v);
    }
}

// tokens[p] = [5 output tokens; sparse[p][0 .. Ns-1]] for caller-supplied sparse prompt embeddings (MaskDecoder.forward as a
// stand-alone module call, micro_sam/training/trainable_sam.py:100-106)
__global__ __launch_bounds__(256) void tokens_from_sparse_kernel(const float* __restrict__ out_tokens, const float* __restrict__ sparse,
                                                                 int Ns, int P, int Nt, float* __restrict__ tokens,
                                                                 float* __restrict__ queries, u16* __restrict__ queries16) {
    const int p = blockIdx.x, c = threadIdx.x;
    if (p >= P) return;
    for (int i = 0; i < Nt; ++i) {
        const float v = i < 5 ? out_tokens[i * C + c] : sparse[((long)p * Ns + (i - 5)) * C + c];
        const long o = ((long)p * Nt + i) * C + c;
        tokens[o] = v; queries[o] = v; queries16[o] = f2d(v);
    }
}

// per-prompt decoder source from a caller-supplied dense prompt embedding: keys[p][t][c] = emb[c][t] + dense[p][c][t]
// (NCHW fp32 -> token-major 16-bit stream); grid (T / 32, C / 32, P)
__global__ __launch_bounds__(256) void dense_src_kernel(const float* __restrict__ emb, const float* __restrict__ dense,
                                                        u16* __restrict__ keys) {
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32, p = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* dp = dense + (long)p * C * T;
    for (int i = ty; i < 32; i += 8) tile[i][tx] = emb[(long)(c0 + i) * T + t0 + tx] + dp[(long)(c0 + i) * T + t0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) keys[((long)p * T + t0 + i) * C + c0 + tx] = f2d(tile[tx][i]);
}

// sparse prompt embeddings only (PromptEncoder.forward as a stand-alone module call): [P, Ns, 256] fp32,
// Ns = Np + (boxes ? 2 : (Np > 0 ? 1 : 0)), same arithmetic as prompt_tokens_kernel
__global__ __launch_bounds__(256) void sparse_embed_kernel(
    const float* __restrict__ G, const float* __restrict__ point_embed, const float* __restrict__ not_a_point,
    const float* __restrict__ points, const int* __restrict__ labels, int Np, const float* __restrict__ boxes, int P, int Ns,
    float* __restrict__ sparse) {
    const int p = blockIdx.x, c = threadIdx.x;
    if (p >= P) return;
    float* tp = sparse + (long)p * Ns * C;
    const int j = c & 127;
    int row = 0;
    for (int i = 0; i < Np; ++i, ++row) {
        const float px = points[((long)p * Np + i) * 2] + 0.5f, py = points[((long)p * Np + i) * 2 + 1] + 0.5f;
        const int lab = labels[(long)p * Np + i];
        float s, co;
        pe_encode(G, px / 1024.f, py / 1024.f, j, s, co);
        float v = c < 128 ? s : co;
        if (lab == -1) v = not_a_point[c];
        else if (lab == 0) v += point_embed[0 * C + c];
        else if (lab == 1) v += point_embed[1 * C + c];
        tp[row * C + c] = v;
    }
    if (Np > 0 && !boxes) { tp[row * C + c] = not_a_point[c]; ++row; }
    if (boxes) {
        for (int k = 0; k < 2; ++k, ++row) {
            const float bx = boxes[(long)p * 4 + 2 * k] + 0.5f, by = boxes[(long)p * 4 + 2 * k + 1] + 0.5f;
            float s, co;
            pe_encode(G, bx / 1024.f, by / 1024.f, j, s, co);
            tp[row * C + c] = (c < 128 ? s : co) + point_embed[(2 + k) * C + c];
        }
    }
}

// out_bf16 = bf16(a + b) over n elements (b may be NULL)
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       u16* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 x = *(const float4*)(a + i * 4);
        if (b) { float4 y = *(const float4*)(b + i * 4); x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w; }
        uint2 pk; pk.x = pack2d(x.x, x.y); pk.y = pack2d(x.z, x.w);
        *(uint2*)(out + i * 4) = pk;
    }
}

// out_a = bf16(a + b), out_b = bf16(a): the (with PE, without PE) operand pair of an attention, one launch
__global__ __launch_bounds__(256) void add_cast2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        u16* __restrict__ out_a, u16* __restrict__ out_b, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 x = *(const float4*)(a + i * 4), y = *(const float4*)(b + i * 4);
        uint2 pa, pb;
        pa.x = pack2d(x.x + y.x, x.y + y.y); pa.y = pack2d(x.z + y.z, x.w + y.w);
        pb.x = pack2d(x.x, x.y); pb.y = pack2d(x.z, x.w);
        *(uint2*)(out_a + i * 4) = pa;
        *(uint2*)(out_b + i * 4) = pb;
    }
}

// Token self-attention: q,k,v bf16 [P*Nt,256], 8 heads x 32; thread = (head, query i); Nt <= 16.
__global__ __launch_bounds__(128) void token_self_attn_kernel(const u16* __restrict__ q, const u16* __restrict__ k,
                                                              const u16* __restrict__ v, int Nt, u16* __restrict__ out) {
    const int p = blockIdx.x, head = threadIdx.x >> 4, i = threadIdx.x & 15;
    if (i >= Nt) return;
    const long base = (long)p * Nt * C + head * 32;
    float qv[32];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        uint4 w = *(const uint4*)(q + base + (long)i * C + c4 * 8);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int x = 0; x < 4; ++x) { qv[c4 * 8 + 2 * x] = d2f((u16)(ww[x] & 0xffff)); qv[c4 * 8 + 2 * x + 1] = d2f((u16)(ww[x] >> 16)); }
    }
    float s[16];
    float m = NEG_BIG;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float acc = 0.f;
        if (j < Nt) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                uint4 w = *(const uint4*)(k + base + (long)j * C + c4 * 8);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    acc += qv[c4 * 8 + 2 * x] * d2f((u16)(ww[x] & 0xffff));
                    acc += qv[c4 * 8 + 2 * x + 1] * d2f((u16)(ww[x] >> 16));
                }
            }
            acc *= 0.17677669529663687f;     // 1/sqrt(32)
        } else acc = NEG_BIG;
        s[j] = acc; m = fmaxf(m, acc);
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { s[j] = __expf(s[j] - m); l += s[j]; }
    float o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < Nt) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                uint4 w = *(const uint4*)(v + base + (long)j * C + c4 * 8);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    o[c4 * 8 + 2 * x] += s[j] * d2f((u16)(ww[x] & 0xffff));
                    o[c4 * 8 + 2 * x + 1] += s[j] * d2f((u16)(ww[x] >> 16));
                }
            }
        }
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
        uint4 pk;
        pk.x = pack2d(o[c4 * 8 + 0] * inv, o[c4 * 8 + 1] * inv); pk.y = pack2d(o[c4 * 8 + 2] * inv, o[c4 * 8 + 3] * inv);
        pk.z = pack2d(o[c4 * 8 + 4] * inv, o[c4 * 8 + 5] * inv); pk.w = pack2d(o[c4 * 8 + 6] * inv, o[c4 * 8 + 7] * inv);
        *(uint4*)(out + base + (long)i * C + c4 * 8) = pk;
    }
}

// Token -> image cross attention (8 heads x 16, 4096 keys), transposed-score MFMA form (see attention.hip):
//   S^T[t][j] = k_img[t] . q_tok[j],  online softmax over t per column j,  O^T[d][j] += V^T[d][t] P^T[t][j].
// One workgroup per (prompt, head); the 4 waves split the 4096 keys and merge their (m, l, O) at the end.
// k_img: bf16 head-major [Pk][8][4096][16] (each (prompt, head) slice is one contiguous 128 KiB stream),
// vT: bf16 [Pk,128,4096]; kv_shared: all prompts use prompt 0's K/V.
__global__ __launch_bounds__(256) void t2i_attn_kernel(const u16* __restrict__ qtok, const u16* __restrict__ kimg,
                                                       const u16* __restrict__ vT, int kv_shared, int Nt,
                                                       u16* __restrict__ out) {
    __shared__ float red[4][16][20];      // per wave: [j][m, l, o[16]] (+pad)
    const int p = blockIdx.x >> 3, head = blockIdx.x & 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int pk = kv_shared ? 0 : p;
    const u16* kb = kimg + ((long)pk * 8 + head) * T * 16;
    const u16* vb = vT + ((long)pk * CI + head * 16 + fr) * T;
    // B operand: q_tok[j = fr][d = fg*8 ..] for fg < 2 (K = 16 zero-padded to 32)
    uint4 qf = make_uint4(0, 0, 0, 0);
    if (fg < 2 && fr < Nt) qf = *(const uint4*)(qtok + ((long)p * Nt + fr) * CI + head * 16 + fg * 8);
    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
    float m = NEG_BIG, l = 0.f;
    const int t_begin = wave * (T / 4), t_end = t_begin + T / 4;
    for (int t0 = t_begin; t0 < t_end; t0 += 32) {
        uint4 ka0 = make_uint4(0, 0, 0, 0), ka1 = ka0;
        if (fg < 2) {
            ka0 = *(const uint4*)(kb + (long)(t0 + fr) * 16 + fg * 8);
            ka1 = *(const uint4*)(kb + (long)(t0 + 16 + fr) * 16 + fg * 8);
        }
        const uint2 v0 = *(const uint2*)(vb + t0 + fg * 4), v1 = *(const uint2*)(vb + t0 + 16 + fg * 4);
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        s0 = mfma16d(ka0, qf, s0);
        s1 = mfma16d(ka1, qf, s1);
        float mt = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[r] *= 0.25f; s1[r] *= 0.25f; mt = fmaxf(mt, fmaxf(s0[r], s1[r])); }
        mt = fmaxf(mt, __shfl_xor(mt, 16)); mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mn = fmaxf(m, mt), alpha = __expf(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0[r] = __expf(s0[r] - mn); s1[r] = __expf(s1[r] - mn); ps += s0[r] + s1[r]; }
        l = l * alpha + ps;
        uint4 pb;
        pb.x = pack2d(s0[0], s0[1]); pb.y = pack2d(s0[2], s0[3]); pb.z = pack2d(s1[0], s1[1]); pb.w = pack2d(s1[2], s1[3]);
        o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
        o = mfma16d(make_uint4(v0.x, v0.y, v1.x, v1.y), pb, o);
    }
    l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
    // merge the 4 waves: lane (j = fr, fg) holds o[d = fg*4 + r]
    if (fg == 0) { red[wave][fr][0] = m; red[wave][fr][1] = l; }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][fr][2 + fg * 4 + r] = o[r];
    __syncthreads();
    if (wave == 0 && fr < Nt) {
        float mm = fmaxf(fmaxf(red[0][fr][0], red[1][fr][0]), fmaxf(red[2][fr][0], red[3][fr][0]));
        float ll = 0.f, oo[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = __expf(red[w][fr][0] - mm);
            ll += a * red[w][fr][1];
#pragma unroll
            for (int r = 0; r < 4; ++r) oo[r] += a * red[w][fr][2 + fg * 4 + r];
        }
        const float inv = 1.f / ll;
        uint2 pkd; pkd.x = pack2d(oo[0] * inv, oo[1] * inv); pkd.y = pack2d(oo[2] * inv, oo[3] * inv);
        *(uint2*)(out + ((long)p * Nt + fr) * CI + head * 16 + fg * 4) = pkd;
    }
}

// Token -> image attention on the SHARED layer-0 K / V^T (prompt independent, L2 resident) for up to 8 tokens per prompt:
// one workgroup per (4 prompts, head), i.e. two 16-column score tiles (column = (prompt & 1) * 8 + token) per K / V^T
// fragment - a quarter of the L2 traffic of the one-prompt-per-workgroup kernel above, which is L2-bandwidth bound.
__global__ __launch_bounds__(256) void t2i_shared4_kernel(const u16* __restrict__ qtok, const u16* __restrict__ kimg,
                                                          const u16* __restrict__ vT, int P, int Nt, u16* __restrict__ out) {
    __shared__ float red[4][32][20];      // per wave: [column][m, l, o[16]] (+pad)
    const int pg = blockIdx.x >> 3, head = blockIdx.x & 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const u16* kb = kimg + (long)head * T * 16;
    const u16* vb = vT + ((long)head * 16 + fr) * T;
    const int tk = fr & 7;
    uint4 qf[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int p = pg * 4 + n * 2 + (fr >> 3);
        qf[n] = make_uint4(0, 0, 0, 0);
        if (fg < 2 && tk < Nt && p < P) qf[n] = *(const uint4*)(qtok + ((long)p * Nt + tk) * CI + head * 16 + fg * 8);
    }
    f32x4_t o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float m[2] = {NEG_BIG, NEG_BIG}, l[2] = {0.f, 0.f};
    const int t_begin = wave * (T / 4), t_end = t_begin + T / 4;
    for (int t0 = t_begin; t0 < t_end; t0 += 32) {
        uint4 ka0 = make_uint4(0, 0, 0, 0), ka1 = ka0;
        if (fg < 2) {
            ka0 = *(const uint4*)(kb + (long)(t0 + fr) * 16 + fg * 8);
            ka1 = *(const uint4*)(kb + (long)(t0 + 16 + fr) * 16 + fg * 8);
        }
        const uint2 v0 = *(const uint2*)(vb + t0 + fg * 4), v1 = *(const uint2*)(vb + t0 + 16 + fg * 4);
        const uint4 va = make_uint4(v0.x, v0.y, v1.x, v1.y);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            s0 = mfma16d(ka0, qf[n], s0);
            s1 = mfma16d(ka1, qf[n], s1);
            float mt = NEG_BIG;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s0[r] *= 0.25f; s1[r] *= 0.25f; mt = fmaxf(mt, fmaxf(s0[r], s1[r])); }
            mt = fmaxf(mt, __shfl_xor(mt, 16)); mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mn = fmaxf(m[n], mt), alpha = __expf(m[n] - mn);
            m[n] = mn;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s0[r] = __expf(s0[r] - mn); s1[r] = __expf(s1[r] - mn); ps += s0[r] + s1[r]; }
            l[n] = l[n] * alpha + ps;
            uint4 pb;
            pb.x = pack2d(s0[0], s0[1]); pb.y = pack2d(s0[2], s0[3]); pb.z = pack2d(s1[0], s1[1]); pb.w = pack2d(s1[2], s1[3]);
            o[n][0] *= alpha; o[n][1] *= alpha; o[n][2] *= alpha; o[n][3] *= alpha;
            o[n] = mfma16d(va, pb, o[n]);
        }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        float ln = l[n];
        ln += __shfl_xor(ln, 16); ln += __shfl_xor(ln, 32);
        if (fg == 0) { red[wave][n * 16 + fr][0] = m[n]; red[wave][n * 16 + fr][1] = ln; }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][n * 16 + fr][2 + fg * 4 + r] = o[n][r];
    }
    __syncthreads();
    if (wave < 2) {                                  // wave n merges score tile n: lane (column fr, fg) -> o[d = fg*4 + r]
        const int n = wave, c = n * 16 + fr, p = pg * 4 + n * 2 + (fr >> 3);
        if (tk < Nt && p < P) {
            const float mm = fmaxf(fmaxf(red[0][c][0], red[1][c][0]), fmaxf(red[2][c][0], red[3][c][0]));
            float ll = 0.f, oo[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float a = __expf(red[w][c][0] - mm);
                ll += a * red[w][c][1];
#pragma unroll
                for (int r = 0; r < 4; ++r) oo[r] += a * red[w][c][2 + fg * 4 + r];
            }
            const float inv = 1.f / ll;
            uint2 pkd; pkd.x = pack2d(oo[0] * inv, oo[1] * inv); pkd.y = pack2d(oo[2] * inv, oo[3] * inv);
            *(uint2*)(out + ((long)p * Nt + tk) * CI + head * 16 + fg * 4) = pkd;
        }
    }
}

inline int grid_for(long items) { long g = (items + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

// ------------------------------------------------------------------------------------------ host helpers

struct Ctx { hipStream_t s; int use_glds; };

// one product of a grouped launch (msam_gemm_group_bf16): the token-side projections are 7 168-row products, latency-bound
// one launch at a time; independent ones (self-attention q / k / v, the k / v of image->token attention, the five output heads
// layer by layer) go out together
msam_gemm_t mk_gemm(const void* A, long lda, const void* W, int M, int N, int K, const float* bias, void* out, int out_dtype,
                    long ldc, int act = 0) {
    msam_gemm_t g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.bias = bias;
    g.act = act; g.out = out; g.out_dtype = out_dtype; g.ldc = ldc; g.a_dtype = MSAM_D16;
    return g;
}

int gemm(const Ctx& cx, const void* A, long lda, const void* W, int M, int N, int K, const float* bias, void* out,
         int out_dtype, long ldc, int act = 0, const void* resid = nullptr, int resid_dtype = 0, long ldr = 0,
         int resid_rows = 0, const float* table = nullptr, int table_cols = 0, int ln_mode = 0,
         const float* ln_w = nullptr, const float* ln_b = nullptr, float ln_eps = 1e-5f) {
    msam_gemm_t g{};
    g.ln_mode = ln_mode; g.ln_w = ln_w; g.ln_b = ln_b; g.ln_eps = ln_eps;
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.bias = bias;
    g.table = table; g.table_rows = T; g.table_cols = table_cols; g.table_ld = CI;
    g.resid = resid; g.resid_dtype = resid_dtype; g.resid_rows = resid_rows; g.ldr = ldr;
    g.act = act; g.out = out; g.out_dtype = out_dtype; g.ldc = ldc; g.out_mode = 0; g.a_dtype = MSAM_D16;
    g.use_glds = MSAM_DEC_F16 ? 0 : cx.use_glds;
    return msam_gemm_bf16(&g, cx.s);
}

// Token-side product with everything behind it in ONE launch (gemm_ln_kernel<F16>, round 3): queries = LayerNorm(A W^T + bias + resid),
// and the 16-bit operand copies of the next products: out_a = round16(queries + add) (add = the positional encoding of the prompt
// tokens, or NULL), out_b = round16(queries).  Was: GEMM, LayerNorm, add_cast[2] = 3 dependent latency-bound launches.
int gemm_ln_tok(const Ctx& cx, const void* A, long lda, const void* W, int M, int K, const float* bias, const float* resid,
                const float* ln_w, const float* ln_b, float* queries, const float* add, void* out_a, void* out_b) {
    msam_gemm_t g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = K; g.M = M; g.N = C; g.K = K; g.bias = bias;
    g.resid = resid; g.resid_dtype = resid ? MSAM_F32 : 0; g.ldr = C;
    g.out = queries; g.out_dtype = MSAM_F32; g.ldc = C; g.a_dtype = MSAM_D16;
    g.ln_mode = 1; g.ln_w = ln_w; g.ln_b = ln_b; g.ln_eps = 1e-5f;
    g.ln_add = add; g.ln_out_a = out_a; g.ln_out_b = out_b;
    return msam_gemm_bf16(&g, cx.s);
}

// K | V^T projection of the per-prompt stream as two N = 128 launches (k with the positional table, v transposed)
int wsgemm_kv(const Ctx& cx, const void* x, const u16* wkv, const float* bkv, const float* pek, int rows, void* k_out,
              void* vT_out) {
    {
        msam_wsgemm_t k{};
        k.A = x; k.W = wkv; k.M = rows; k.N = CI; k.K = C; k.bias = bkv;
        k.table = pek; k.table_rows = T; k.table_cols = CI; k.table_ld = CI;
        k.out = k_out; k.ldc = CI; k.head_major = 1; k.tokens = T;
        if (int e = msam_wsgemm_bf16(&k, cx.s)) return e;
    }
    msam_wsgemm_t g{};
    g.A = x; g.W = wkv + (long)CI * C; g.M = rows; g.N = CI; g.K = C; g.bias = bkv + CI;
    g.kv_split = 1; g.vT_out = vT_out; g.tokens = T;
    return msam_wsgemm_bf16(&g, cx.s);
}

struct Consts {     // layout of the `consts` buffer
    float* pos_f32; u16* pos_bf16; float* pe_k[3]; float* pe_q[2]; u16* wkv[3]; float* bkv[3];
    u16* tab_k[3];  // bf16(pos Wk^T + bk): score table of the folded token->image attention (decfold.hip)
    u16* tab_q[2];  // bf16(pos Wq^T + bq): score table of the folded image->token attention
    void* chain2;   // weight-only tables of the second form of the chained attention (msam_chain_prepare_const2)
};
constexpr long CONST2_BYTES = 128L * 128 * 4 + 4 * 512 + (long)CI * C * 2;        // == msam_chain_const2_bytes() (checked in prepare_const)
constexpr long CONST_BYTES = (long)T * C * 4 + (long)T * C * 2 + 5L * T * CI * 4 + 3L * C * C * 2 + 3L * C * 4 + 5L * T * CI * 2 + CONST2_BYTES;

Consts carve_consts(void* base) {
    Consts c; char* p = (char*)base;
    c.pos_f32 = (float*)p; p += (long)T * C * 4;
    c.pos_bf16 = (u16*)p; p += (long)T * C * 2;
    for (int i = 0; i < 3; ++i) { c.pe_k[i] = (float*)p; p += (long)T * CI * 4; }
    for (int i = 0; i < 2; ++i) { c.pe_q[i] = (float*)p; p += (long)T * CI * 4; }
    for (int i = 0; i < 3; ++i) { c.wkv[i] = (u16*)p; p += (long)C * C * 2; }
    for (int i = 0; i < 3; ++i) { c.bkv[i] = (float*)p; p += (long)C * 4; }
    for (int i = 0; i < 3; ++i) { c.tab_k[i] = (u16*)p; p += (long)T * CI * 2; }
    for (int i = 0; i < 2; ++i) { c.tab_q[i] = (u16*)p; p += (long)T * CI * 2; }
    c.chain2 = p;
    return c;
}

struct ImageState { float* src_f32; u16* src_bf16; u16* k0; u16* vT0; u16* q0; };
constexpr long IMAGE_BYTES = (long)T * C * 4 + (long)T * C * 2 + 3L * T * CI * 2;

ImageState carve_image(void* base) {
    ImageState s; char* p = (char*)base;
    s.src_f32 = (float*)p; p += (long)T * C * 4;
    s.src_bf16 = (u16*)p; p += (long)T * C * 2;
    s.k0 = (u16*)p; p += (long)T * CI * 2;
    s.vT0 = (u16*)p; p += (long)T * CI * 2;
    s.q0 = (u16*)p; p += (long)T * CI * 2;
    return s;
}

inline long align256(long x) { return (x + 255) & ~255L; }

}  // namespace

extern "C" int64_t msam_decoder_const_bytes(void) { return CONST_BYTES; }
// the 16-bit type of every decoder weight the caller hands over (msam_decoder_t, msam_upscale_fused, msam_wsgemm_bf16, ...)
// and of the decoder's 16-bit intermediates: MSAM_F16 (default build) or MSAM_BF16 (MSAM_DEC_F16 = 0)
extern "C" int msam_decoder_dtype(void) { return MSAM_D16; }
extern "C" int64_t msam_decoder_image_bytes(void) { return IMAGE_BYTES; }

extern "C" int msam_decoder_prepare_const(const msam_decoder_t* dec, void* consts, void* stream) {
    if (!dec || !consts) { msam_set_error("msam_decoder_prepare_const: null argument"); return 1; }
    Ctx cx{(hipStream_t)stream, dec->use_glds};
    Consts c = carve_consts(consts);
    hipLaunchKernelGGL(dense_pe_kernel, dim3(T * 128 / 256), dim3(256), 0, cx.s, dec->pe_gauss, c.pos_f32, c.pos_bf16);
    if (int e = msam_check_launch("dense_pe")) return e;
    // concatenated [Wk; Wv] weights / biases of the three token->image attentions (device-to-device copies)
    const msam_attn_w_t* t2i[3] = {&dec->layer[0].t2i, &dec->layer[1].t2i, &dec->final_attn};
    for (int i = 0; i < 3; ++i) {
        hipMemcpyAsync(c.wkv[i], t2i[i]->k_w, (size_t)CI * C * 2, hipMemcpyDeviceToDevice, cx.s);
        hipMemcpyAsync(c.wkv[i] + (long)CI * C, t2i[i]->v_w, (size_t)CI * C * 2, hipMemcpyDeviceToDevice, cx.s);
        hipMemcpyAsync(c.bkv[i], t2i[i]->k_b, CI * 4, hipMemcpyDeviceToDevice, cx.s);
        hipMemcpyAsync(c.bkv[i] + CI, t2i[i]->v_b, CI * 4, hipMemcpyDeviceToDevice, cx.s);
        // pos . Wk^T (no bias): added to the k half through the GEMM table epilogue
        if (int e = gemm(cx, c.pos_bf16, C, t2i[i]->k_w, T, CI, C, nullptr, c.pe_k[i], MSAM_F32, CI)) return e;
        if (int e = gemm(cx, c.pos_bf16, C, t2i[i]->k_w, T, CI, C, t2i[i]->k_b, c.tab_k[i], MSAM_D16, CI)) return e;
    }
    for (int i = 0; i < 2; ++i) {
        if (int e = gemm(cx, c.pos_bf16, C, dec->layer[i].i2t.q_w, T, CI, C, nullptr, c.pe_q[i], MSAM_F32, CI)) return e;
        if (int e = gemm(cx, c.pos_bf16, C, dec->layer[i].i2t.q_w, T, CI, C, dec->layer[i].i2t.q_b, c.tab_q[i], MSAM_D16, CI))
            return e;
    }
    // weight-only tables of the chained attention's second form (decfold_tok.hip): layer-1 token->image Wv / bv / Wk, norm4 and the
    // image->token out_proj of layer 0
    if (msam_chain_const2_bytes() != CONST2_BYTES) { msam_set_error("msam_decoder_prepare_const: const2 size mismatch"); return 2; }
    return msam_chain_prepare_const2(dec->layer[1].t2i.v_w, dec->layer[1].t2i.v_b, dec->layer[1].t2i.k_w, dec->layer[0].n4_w,
                                     dec->layer[0].n4_b, dec->layer[0].i2t.o_w, dec->layer[0].i2t.o_b, c.chain2, stream);
}

extern "C" int msam_decoder_prepare_image(const msam_decoder_t* dec, const void* consts, const float* embedding,
                                          void* image_state, void* workspace, int64_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    if (!dec || !consts || !embedding || !image_state) { msam_set_error("msam_decoder_prepare_image: null argument"); return 1; }
    Ctx cx{(hipStream_t)stream, dec->use_glds};
    Consts c = carve_consts((void*)consts);
    ImageState im = carve_image(image_state);
    hipLaunchKernelGGL(src_prepare_kernel, dim3(T / 32, C / 32), dim3(256), 0, cx.s, embedding, dec->no_mask, im.src_f32,
                       im.src_bf16);
    if (int e = msam_check_launch("src_prepare")) return e;
    if (int e = wsgemm_kv(cx, im.src_bf16, c.wkv[0], c.bkv[0], c.pe_k[0], T, im.k0, im.vT0)) return e;
    return gemm(cx, im.src_bf16, C, dec->layer[0].i2t.q_w, T, CI, C, dec->layer[0].i2t.q_b, im.q0, MSAM_D16, CI, 0,
                nullptr, 0, 0, 0, c.pe_q[0], CI);
}

extern int g_tune_dec_chain, g_tune_dec_chain_min_p, g_tune_chain_variant;
// msam_tune_set "tok_fuse": 1 = product + LayerNorm + operand copies of the token side in one launch each (gemm_ln_tok: 18 fewer launches
// per decode).  Measured (profiles/r03_experiments.md): 151.7 vs 154.4 tiles/s with one decode lane, 167.9 vs 168.7 with three - the
// row-complete 64 x 256 tile kernel walks K = 2048 of the MLP on 112 workgroups and gives the saved launches back; the token side is
// bound by the latency INSIDE each small product (a dependent chain of 4 - 32 k-tiles on a fraction of the chip), not by launch count.
// Default off; kept as a tested option.
int g_tune_tok_fuse = 0;
int g_tune_mlp_split_fused = 1;           // msam_tune_set "mlp_split_fused": 0 = fp32 hidden + separate cast launch (the A/B and test form)
namespace {
struct Work {
    float *qpe, *queries, *tmp; u16 *a, *b, *qs, *ks, *vs, *attn_tok, *mlp_h;
    u16 *keys, *kimg, *vT, *qimg, *attn_img, *up1; float* pre;
    u16 *hh0, *hh1; float *hyper, *iou_full;
};
long work_bytes(int P, int Nt) {
    const long M = (long)P * Nt, R = (long)P * T;
    long b = 0;
    b += 3 * align256(M * C * 4);                 // qpe, queries, tmp
    b += 6 * align256(M * C * 2);                 // a, b, qs, ks, vs, attn_tok
    b += align256(M * 2048 * 2);                  // mlp_h
    b += align256(R * C * 2);                     // keys
    b += 4 * align256(R * CI * 2);                // kimg, vT, qimg, attn_img
    b += align256(R * C * 2);                     // up1
    b += align256(R * C * 4);                     // pre
    b += 2 * align256(5L * P * C * 2);            // hh0, hh1 (one [P, 256] block per head)
    b += align256((long)P * 4 * 128 * 4) + align256((long)P * 128 * 4);
    return b;
}
Work carve_work(void* base, int P, int Nt) {
    const long M = (long)P * Nt, R = (long)P * T;
    Work w; char* p = (char*)base;
    auto take = [&](long bytes) { char* r = p; p += align256(bytes); return r; };
    w.qpe = (float*)take(M * C * 4); w.queries = (float*)take(M * C * 4); w.tmp = (float*)take(M * C * 4);
    w.a = (u16*)take(M * C * 2); w.b = (u16*)take(M * C * 2); w.qs = (u16*)take(M * C * 2); w.ks = (u16*)take(M * C * 2);
    w.vs = (u16*)take(M * C * 2); w.attn_tok = (u16*)take(M * C * 2);
    w.mlp_h = (u16*)take(M * 2048 * 2);
    w.keys = (u16*)take(R * C * 2);
    w.kimg = (u16*)take(R * CI * 2); w.vT = (u16*)take(R * CI * 2); w.qimg = (u16*)take(R * CI * 2);
    w.attn_img = (u16*)take(R * CI * 2);
    w.up1 = (u16*)take(R * C * 2);
    w.pre = (float*)take(R * C * 4);
    w.hh0 = (u16*)take(5L * P * C * 2); w.hh1 = (u16*)take(5L * P * C * 2);
    w.hyper = (float*)take((long)P * 4 * 128 * 4); w.iou_full = (float*)take((long)P * 128 * 4);
    return w;
}

// token -> image attention over the per-prompt stream w.keys: folded form (one pass over the stream, decfold.hip) for
// up to 8 tokens per prompt, explicit K / V^T projection + attention kernel otherwise
int t2i_stream(const Ctx& cx, const Work& w, const Consts& c, int idx, const msam_attn_w_t& aw, int P, int Nt, bool blocked = false) {
    const long R = (long)P * T;
    if (Nt <= 8) {
        // workspace of the folded form = the (then unused) K / V^T / q / attention stream buffers, contiguous in `Work`
        const int64_t avail = (int64_t)((char*)w.up1 - (char*)w.kimg);
        return msam_t2i_fold_attention(w.keys, blocked ? 2 : 0, w.qs, P, Nt, aw.k_w, c.tab_k[idx], aw.v_w, aw.v_b, w.attn_tok, w.kimg,
                                       avail, cx.s);
    }
    if (int e = wsgemm_kv(cx, w.keys, c.wkv[idx], c.bkv[idx], c.pe_k[idx], (int)R, w.kimg, w.vT)) return e;
    hipLaunchKernelGGL(t2i_attn_kernel, dim3(P * 8), dim3(256), 0, cx.s, w.qs, w.kimg, w.vT, 0, Nt, w.attn_tok);
    return msam_check_launch("t2i_attn");
}

__global__ void gather_iou_kernel(const float* __restrict__ iou_full, int P, int c0, int nc, float* __restrict__ iou) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P * nc) iou[i] = iou_full[(long)(i / nc) * 128 + c0 + (i % nc)];
}
}  // namespace

extern "C" int64_t msam_decoder_workspace_bytes(int32_t P) { return work_bytes(P, 16); }

extern "C" int msam_decoder_forward(const msam_decoder_t* dec, const void* consts, const void* image_state,
                                    const float* points, const int32_t* labels, int32_t Np, const float* boxes, int32_t P,
                                    int32_t multimask, float* low_res, float* iou, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
    return msam_decoder_forward_masks(dec, nullptr, consts, image_state, points, labels, Np, boxes, nullptr, P, multimask,
                                      low_res, iou, workspace, workspace_bytes, stream);
}

namespace {
int decoder_run(const msam_decoder_t* dec, const msam_mask_prompt_t* mask_w, const void* consts,
                const void* image_state, const float* points, const int32_t* labels, int32_t Np,
                const float* boxes, const float* mask_input, const float* sparse, int32_t Ns, const float* dense,
                const float* embedding, int32_t P, int32_t multimask,
                float* low_res, float* iou, void* workspace, int64_t workspace_bytes, void* stream);
}

extern "C" int msam_decoder_forward_masks(const msam_decoder_t* dec, const msam_mask_prompt_t* mask_w, const void* consts,
                                          const void* image_state, const float* points, const int32_t* labels, int32_t Np,
                                          const float* boxes, const float* mask_input, int32_t P, int32_t multimask,
                                          float* low_res, float* iou, void* workspace, int64_t workspace_bytes, void* stream) {
    return decoder_run(dec, mask_w, consts, image_state, points, labels, Np, boxes, mask_input, nullptr, 0, nullptr, nullptr, P,
                       multimask, low_res, iou, workspace, workspace_bytes, stream);
}

extern "C" int msam_decoder_forward_embeddings(const msam_decoder_t* dec, const void* consts, const void* image_state,
                                               const float* sparse, int32_t Ns, const float* dense, const float* embedding,
                                               int32_t P, int32_t multimask, float* low_res, float* iou, void* workspace,
                                               int64_t workspace_bytes, void* stream) {
    if (!sparse && Ns != 0) { msam_set_error("msam_decoder_forward_embeddings: Ns > 0 needs the sparse embeddings"); return 1; }
    if (Ns < 0 || Ns > 11) { msam_set_error("msam_decoder_forward_embeddings: 0 .. 11 sparse tokens per prompt"); return 1; }
    if (dense && !embedding) { msam_set_error("msam_decoder_forward_embeddings: a dense embedding needs the image embedding"); return 1; }
    return decoder_run(dec, nullptr, consts, image_state, nullptr, nullptr, 0, nullptr, nullptr, sparse ? sparse : (const float*)consts,
                       Ns, dense, embedding, P, multimask, low_res, iou, workspace, workspace_bytes, stream);
}

extern "C" int msam_prompt_encode(const msam_decoder_t* dec, const msam_mask_prompt_t* mask_w, const float* points,
                                  const int32_t* labels, int32_t Np, const float* boxes, const float* mask_input, int32_t P,
                                  float* sparse, float* dense, void* stream) {
    if (!dec || P <= 0) { msam_set_error("msam_prompt_encode: null argument"); return 1; }
    if (!points) Np = 0;
    const int Ns = Np + (boxes ? 2 : (Np > 0 ? 1 : 0));
    if (Ns > 0) {
        if (!sparse) { msam_set_error("msam_prompt_encode: null sparse output"); return 1; }
        hipLaunchKernelGGL(sparse_embed_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, dec->pe_gauss, dec->point_embed,
                           dec->not_a_point, points, labels, Np, boxes, P, Ns, sparse);
        if (int e = msam_check_launch("sparse_embed")) return e;
    }
    if (mask_input) {
        if (!mask_w || !dense) { msam_set_error("msam_prompt_encode: mask prompts need the mask_downscaling weights and a dense output"); return 1; }
        hipLaunchKernelGGL(mask_src_kernel, dim3(64, P), dim3(256), 0, (hipStream_t)stream, mask_input, *mask_w, (const float*)nullptr,
                           dec->no_mask, (u16*)nullptr, dense);
        if (int e = msam_check_launch("mask_dense")) return e;
    }
    return 0;
}

namespace {
int decoder_run(const msam_decoder_t* dec, const msam_mask_prompt_t* mask_w, const void* consts,
                const void* image_state, const float* points, const int32_t* labels, int32_t Np,
                const float* boxes, const float* mask_input, const float* sparse, int32_t Ns, const float* dense,
                const float* embedding, int32_t P, int32_t multimask,
                float* low_res, float* iou, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!dec || !consts || !image_state || !low_res || !iou || !workspace || P <= 0) {
        msam_set_error("msam_decoder_forward: null argument");
        return 1;
    }
    // a mask prompt on its own is a prompt (reference prompt_based_segmentation.segment_from_mask(use_box=False, use_points=False):
    // no sparse token, the five output tokens only, the mask enters through the per-prompt source stream)
    if (!sparse && (!points || Np <= 0) && !boxes && !mask_input && !dense) {
        msam_set_error("msam_decoder_forward: need points, boxes and / or a mask prompt");
        return 1;
    }
    if (mask_input && !mask_w) { msam_set_error("msam_decoder_forward: mask prompts need the mask_downscaling weights"); return 1; }
    const bool own_src = mask_input != nullptr || dense != nullptr;   // per-prompt source stream (image embedding + dense embedding)
    if (!points) Np = 0;
    const int Nt = sparse ? 5 + Ns : 5 + Np + (boxes ? 2 : (Np > 0 ? 1 : 0));
    if (Nt > 16) { msam_set_error("msam_decoder_forward: at most 16 tokens per prompt (<= 10 points)"); return 1; }
    if (workspace_bytes < work_bytes(P, Nt)) { msam_set_error("msam_decoder_forward: workspace too small"); return 1; }
    Ctx cx{(hipStream_t)stream, dec->use_glds};
    Consts c = carve_consts((void*)consts);
    ImageState im = carve_image((void*)image_state);
    Work w = carve_work(workspace, P, Nt);
    const int M = P * Nt;
    const long R = (long)P * T;
    const long n4 = (long)M * C / 4;
    int e;
#define CHECK(x) do { if ((e = (x))) return e; } while (0)
#define ADD_CAST(a_, b_, out_) do { hipLaunchKernelGGL(add_cast_kernel, dim3(grid_for(n4)), dim3(256), 0, cx.s, a_, b_, out_, n4); \
                                     CHECK(msam_check_launch("add_cast")); } while (0)
#define ADD_CAST2(a_, b_, outa_, outb_) do { hipLaunchKernelGGL(add_cast2_kernel, dim3(grid_for(n4)), dim3(256), 0, cx.s, a_, b_, outa_, \
                                                                 outb_, n4); CHECK(msam_check_launch("add_cast2")); } while (0)
#define LN(x_, w_, b_, rows_, out_, dt_) CHECK(msam_layernorm(x_, w_, b_, 1e-5f, rows_, C, out_, dt_, 0, 0, cx.s))

    if (dense) {
        hipLaunchKernelGGL(dense_src_kernel, dim3(T / 32, C / 32, P), dim3(256), 0, cx.s, embedding, dense, w.keys);
        CHECK(msam_check_launch("dense_src"));
    } else if (own_src) {
        hipLaunchKernelGGL(mask_src_kernel, dim3(64, P), dim3(256), 0, cx.s, mask_input, *mask_w, im.src_f32, dec->no_mask, w.keys,
                           (float*)nullptr);
        CHECK(msam_check_launch("mask_src"));
    }
    if (sparse) {
        hipLaunchKernelGGL(tokens_from_sparse_kernel, dim3(P), dim3(256), 0, cx.s, dec->out_tokens, sparse, Ns, P, Nt, w.qpe,
                           w.queries, w.a);
        CHECK(msam_check_launch("tokens_from_sparse"));
    } else {
        hipLaunchKernelGGL(prompt_tokens_kernel, dim3(P), dim3(256), 0, cx.s, dec->pe_gauss, dec->point_embed, dec->not_a_point,
                           dec->out_tokens, points, labels, Np, boxes, P, Nt, w.qpe, w.queries, w.a);
        CHECK(msam_check_launch("prompt_tokens"));
    }

    // test hook: MSAM_DEBUG_DEC_LAYERS=0/1 stops the two-way transformer early so that intermediate workspace buffers
    // can be compared with the oracle's per-layer taps (outputs are then NOT the model's outputs)
    const char* dbg = getenv("MSAM_DEBUG_DEC_LAYERS");
    const int nlayers = dbg ? atoi(dbg) : 2;
    // Chained form (decfold_tok.hip): with one shared source the layer-0 output stream is not written; the layer-1
    // token->image attention and the layer-1 image->token block recompute their tiles of it from the L2-resident source.
    const bool chain = g_tune_dec_chain && !own_src && !dbg && Nt <= 8 && P >= g_tune_dec_chain_min_p;
    // token side: product + LayerNorm + operand copies per launch (gemm_ln_tok; msam_tune_set "tok_fuse" 0 = one launch per step)
    const bool fuse_tok = g_tune_tok_fuse != 0 && !dbg;
    // operand images in the (then unused) q / attention stream buffers: layer 0, layer 1; the attention's operands in vT
    void* const oper0 = w.qimg; void* const oper1 = w.attn_img;
    // blocked copies of the shared tables (source, layer-0 q, tabK / tabQ of layer 1) at the end of the attention workspace in vT
    void* const tables = (char*)w.vT + (int64_t)R * CI * 2 - msam_chain_tables_bytes();
    if (chain) CHECK(msam_chain_prepare_tables(im.src_bf16, im.q0, c.tab_k[1], c.tab_q[1], tables, cx.s));
    // second form of the chained attention (msam_tune_set "chain_variant" 9): its prompt-independent tables in front of `tables`,
    // the per-prompt M fragments behind the layer-0 operands
    const bool chain2 = chain && g_tune_chain_variant == 9;
    void* const tables2 = (char*)tables - msam_chain_tables2_bytes();
    void* const mfrag = (char*)oper0 + msam_i2t_fold_operand_bytes(P);
    if (chain2) CHECK(msam_chain_prepare_tables2_c(im.src_bf16, c.chain2, tables2, cx.s));
    for (int li = 0; li < nlayers && li < 2; ++li) {
        const msam_twoway_layer_t& L = dec->layer[li];
        // (1) token self attention
        // layer 0: q = k = v input = bf16(tokens), written by prompt_tokens_kernel; layer 1: q = k input carries the PE
        const u16* sv = w.a;
        if (li > 0) {
            // (w.a, w.b) = round16(queries + pe), round16(queries): written behind norm3 of the previous layer (step 3) and still
            // valid - step 4 only reads them and updates the image-token stream
            if (!fuse_tok) ADD_CAST2(w.queries, w.qpe, w.a, w.b);
            sv = w.b;
        }
        {
            const msam_gemm_t qkv[3] = {mk_gemm(w.a, C, L.self_attn.q_w, M, C, C, L.self_attn.q_b, w.qs, MSAM_D16, C),
                                        mk_gemm(w.a, C, L.self_attn.k_w, M, C, C, L.self_attn.k_b, w.ks, MSAM_D16, C),
                                        mk_gemm(sv, C, L.self_attn.v_w, M, C, C, L.self_attn.v_b, w.vs, MSAM_D16, C)};
            CHECK(msam_gemm_group_bf16(qkv, 3, cx.s));
        }
        hipLaunchKernelGGL(token_self_attn_kernel, dim3(P), dim3(128), 0, cx.s, w.qs, w.ks, w.vs, Nt, w.attn_tok);
        CHECK(msam_check_launch("token_self_attn"));
        if (fuse_tok) {
            CHECK(gemm_ln_tok(cx, w.attn_tok, C, L.self_attn.o_w, M, C, L.self_attn.o_b, li == 0 ? nullptr : w.queries, L.n1_w, L.n1_b,
                              w.queries, w.qpe, w.a, nullptr));
        } else {
        CHECK(gemm(cx, w.attn_tok, C, L.self_attn.o_w, M, C, C, L.self_attn.o_b, w.tmp, MSAM_F32, C, 0,
                   li == 0 ? nullptr : w.queries, li == 0 ? 0 : MSAM_F32, C));
        LN(w.tmp, L.n1_w, L.n1_b, M, w.queries, MSAM_F32);
        // (2) token -> image attention
        ADD_CAST(w.queries, w.qpe, w.a);
        }
        CHECK(gemm(cx, w.a, C, L.t2i.q_w, M, CI, C, L.t2i.q_b, w.qs, MSAM_D16, CI));
        if (li == 0 && !own_src) {
            // prompt-independent K / V^T of the shared embedding (prepare_image): 1 MiB, L2 resident
            if (Nt <= 8)
                hipLaunchKernelGGL(t2i_shared4_kernel, dim3(((P + 3) / 4) * 8), dim3(256), 0, cx.s, w.qs, im.k0, im.vT0, P, Nt,
                                   w.attn_tok);
            else
                hipLaunchKernelGGL(t2i_attn_kernel, dim3(P * 8), dim3(256), 0, cx.s, w.qs, im.k0, im.vT0, 1, Nt, w.attn_tok);
            CHECK(msam_check_launch("t2i_attn"));
        } else if (chain) {
            const msam_twoway_layer_t& L0 = dec->layer[0];
            const int64_t avail = (int64_t)R * CI * 2 - msam_chain_tables_bytes() - msam_chain_tables2_bytes();
            if (chain2)
                CHECK(msam_i2t0_t2i_fused_v2(tables, tables2, oper0, mfrag, L0.n4_w, 1e-5f, w.qs, P, Nt, L.t2i.k_w, w.attn_tok, w.vT,
                                             avail, cx.s));
            else
                CHECK(msam_i2t0_t2i_fused(tables, oper0, L0.n4_w, L0.n4_b, 1e-5f, w.qs, P, Nt, L.t2i.k_w, L.t2i.v_w, L.t2i.v_b,
                                          w.attn_tok, w.vT, avail, cx.s));
        } else {
            CHECK(t2i_stream(cx, w, c, li, L.t2i, P, Nt));
        }
        if (fuse_tok) {
            CHECK(gemm_ln_tok(cx, w.attn_tok, CI, L.t2i.o_w, M, CI, L.t2i.o_b, w.queries, L.n2_w, L.n2_b, w.queries, nullptr, w.a, nullptr));
        } else {
        CHECK(gemm(cx, w.attn_tok, CI, L.t2i.o_w, M, C, CI, L.t2i.o_b, w.tmp, MSAM_F32, C, 0, w.queries, MSAM_F32, C));
        LN(w.tmp, L.n2_w, L.n2_b, M, w.queries, MSAM_F32);
        // (3) token MLP (mlp_split reads the fp32 LayerNorm output itself: no plain 16-bit copy needed)
        if (!(L.mlp1_ws != nullptr && L.mlp2_ws != nullptr)) ADD_CAST(w.queries, nullptr, w.a);
        }
        // token MLP.  mlp_split (round 4): both products on hi + lo operand pairs - the LayerNorm output (fp32, w.queries) and the ReLU
        // hidden (kept in fp32) are laid out as [hi | lo | hi] rows against [Whi | Whi | Wlo] weight rows, one plain 16-bit GEMM over
        // 3 K each.  The 2048-wide hidden is the decoder's most rounding-sensitive tensor (on generic weights its fp16 rounding alone
        // was 60 % of the low-res logit error, profiles/r04_experiments.md section 3); the scratch lives in the otherwise unused w.pre
        const bool mlp_split = L.mlp1_ws != nullptr && L.mlp2_ws != nullptr;
        const void* mlp_h = w.mlp_h; const void* mlp2_w = L.mlp2_w; int mlp2_k = 2048;
        if (mlp_split) {
            u16* a3 = (u16*)w.pre;
            float* h32 = (float*)((char*)a3 + align256((long)M * 3 * C * 2));
            u16* h3 = (u16*)((char*)h32 + align256((long)M * 2048 * 4));
            CHECK(msam_cast_f32_split16(w.queries, MSAM_D16, a3, M, C, cx.s));
            if (g_tune_mlp_split_fused) {
                // the ReLU hidden leaves lin1's epilogue as [hi | lo | hi] rows (msam_gemm_t.out_mode 3): no fp32 copy, no second cast launch
                msam_gemm_t g1 = mk_gemm(a3, 3 * C, L.mlp1_ws, M, 2048, 3 * C, L.mlp1_b, h3, MSAM_D16, 3 * 2048, MSAM_ACT_RELU);
                g1.out_mode = 3;
                CHECK(msam_gemm_bf16(&g1, cx.s));
            } else {
                CHECK(gemm(cx, a3, 3 * C, L.mlp1_ws, M, 2048, 3 * C, L.mlp1_b, h32, MSAM_F32, 2048, MSAM_ACT_RELU));
                CHECK(msam_cast_f32_split16(h32, MSAM_D16, h3, M, 2048, cx.s));
            }
            mlp_h = h3; mlp2_w = L.mlp2_ws; mlp2_k = 3 * 2048;
        } else {
        CHECK(gemm(cx, w.a, C, L.mlp1_w, M, 2048, C, L.mlp1_b, w.mlp_h, MSAM_D16, 2048, MSAM_ACT_RELU));
        }
        if (fuse_tok) {
            CHECK(gemm_ln_tok(cx, mlp_h, mlp2_k, mlp2_w, M, mlp2_k, L.mlp2_b, w.queries, L.n3_w, L.n3_b, w.queries, w.qpe, w.a, w.b));
        } else {
        CHECK(gemm(cx, mlp_h, mlp2_k, mlp2_w, M, C, mlp2_k, L.mlp2_b, w.tmp, MSAM_F32, C, 0, w.queries, MSAM_F32, C));
        LN(w.tmp, L.n3_w, L.n3_b, M, w.queries, MSAM_F32);
        // (4) image -> token attention, updates the image-token stream
        ADD_CAST2(w.queries, w.qpe, w.a, w.b);
        }
        {
            const msam_gemm_t kv[2] = {mk_gemm(w.a, C, L.i2t.k_w, M, CI, C, L.i2t.k_b, w.ks, MSAM_D16, CI),
                                       mk_gemm(w.b, C, L.i2t.v_w, M, CI, C, L.i2t.v_b, w.vs, MSAM_D16, CI)};
            CHECK(msam_gemm_group_bf16(kv, 2, cx.s));
        }
        // image->token attention + out_proj + residual + norm4 in ONE pass over the stream: folded form (decfold.hip)
        // for up to 8 tokens per prompt, weights-stationary fused kernel (declayer.hip) otherwise
        if (chain) {
            if (li == 0 && chain2)
                CHECK(msam_i2t_fold_operands_values(w.ks, w.vs, P, Nt, L.i2t.q_w, L.i2t.o_w, L.i2t.o_b, 0, tables2, oper0, mfrag, cx.s));
            else
                CHECK(msam_i2t_fold_operands(w.ks, w.vs, P, Nt, L.i2t.q_w, L.i2t.o_w, L.i2t.o_b, li, li == 0 ? oper0 : oper1, cx.s));
            if (li == 1) {
                const msam_twoway_layer_t& L0 = dec->layer[0];
                CHECK(msam_i2t01_fused(tables, oper0, L0.n4_w, L0.n4_b, oper1, L.n4_w, L.n4_b, 1e-5f, P, Nt, w.keys, cx.s));   // blocked stream
            }
        } else if (Nt <= 8) {
            const bool shared = li == 0 && !own_src;
            CHECK(msam_i2t_fold_layer(shared ? (const void*)im.src_bf16 : (const void*)w.keys, shared, w.ks, w.vs, P, Nt,
                                      L.i2t.q_w, c.tab_q[li], L.i2t.o_w, L.i2t.o_b, L.n4_w, L.n4_b, 1e-5f, w.keys, w.qimg,
                                      (int64_t)((char*)w.up1 - (char*)w.qimg), cx.s));
        } else {
            msam_image_layer_t g{};
            g.wo = L.i2t.o_w; g.bo = L.i2t.o_b; g.ln_w = L.n4_w; g.ln_b = L.n4_b; g.ln_eps = 1e-5f;
            g.ktok = w.ks; g.vtok = w.vs; g.Nt = Nt; g.out = w.keys; g.rows = (int)R;
            if (li == 0 && !own_src) { g.xin = im.src_bf16; g.q_shared = im.q0; }
            else { g.xin = w.keys; g.wq = L.i2t.q_w; g.bq = L.i2t.q_b; g.peq = c.pe_q[li]; }
            CHECK(msam_decoder_image_layer(&g, cx.s));
        }
    }
    if (dbg) return 0;   // test hook: leave queries / keys of the last executed layer in the workspace
    // final token -> image attention
    if (!fuse_tok) ADD_CAST(w.queries, w.qpe, w.a);          // (fused form: w.a = round16(queries + pe) already, see above)
    CHECK(gemm(cx, w.a, C, dec->final_attn.q_w, M, CI, C, dec->final_attn.q_b, w.qs, MSAM_D16, CI));
    CHECK(t2i_stream(cx, w, c, 2, dec->final_attn, P, Nt, chain));
    if (fuse_tok) {
        CHECK(gemm_ln_tok(cx, w.attn_tok, CI, dec->final_attn.o_w, M, CI, dec->final_attn.o_b, w.queries, dec->nf_w, dec->nf_b, w.queries,
                          nullptr, w.a, nullptr));
    } else {
    CHECK(gemm(cx, w.attn_tok, CI, dec->final_attn.o_w, M, C, CI, dec->final_attn.o_b, w.tmp, MSAM_F32, C, 0, w.queries,
               MSAM_F32, C));
    LN(w.tmp, dec->nf_w, dec->nf_b, M, w.queries, MSAM_F32);

    // heads: hyper-network MLPs on mask tokens 1..4, IoU head on token 0 (bf16 copy of queries, strided rows)
    ADD_CAST(w.queries, nullptr, w.a);
    }
    // the five 3-layer MLPs (IoU head on token 0, hyper-networks on tokens 1..4) layer by layer, one grouped launch per layer
    {
        msam_gemm_t h[5];
        const long PC = (long)P * C;
        for (int i = 0; i < 5; ++i) {
            const void* wt = i == 0 ? dec->iou_w[0] : dec->hyp_w[i - 1][0];
            const float* bs = i == 0 ? dec->iou_b[0] : dec->hyp_b[i - 1][0];
            h[i] = mk_gemm(w.a + (long)i * C, (long)Nt * C, wt, P, C, C, bs, w.hh0 + i * PC, MSAM_D16, C, MSAM_ACT_RELU);
        }
        CHECK(msam_gemm_group_bf16(h, 5, cx.s));
        for (int i = 0; i < 5; ++i) {
            const void* wt = i == 0 ? dec->iou_w[1] : dec->hyp_w[i - 1][1];
            const float* bs = i == 0 ? dec->iou_b[1] : dec->hyp_b[i - 1][1];
            h[i] = mk_gemm(w.hh0 + i * PC, C, wt, P, C, C, bs, w.hh1 + i * PC, MSAM_D16, C, MSAM_ACT_RELU);
        }
        CHECK(msam_gemm_group_bf16(h, 5, cx.s));
        h[0] = mk_gemm(w.hh1, C, dec->iou_w[2], P, 128, C, dec->iou_b[2], w.iou_full, MSAM_F32, 128);
        for (int i = 1; i < 5; ++i)
            h[i] = mk_gemm(w.hh1 + i * PC, C, dec->hyp_w[i - 1][2], P, 128, C, dec->hyp_b[i - 1][2], w.hyper + (long)(i - 1) * 128,
                           MSAM_F32, 4 * 128);
        CHECK(msam_gemm_group_bf16(h, 5, cx.s));
    }
    const int mask0 = multimask ? 1 : 0, nmask = multimask ? 3 : 1;
    hipLaunchKernelGGL(gather_iou_kernel, dim3((P * nmask + 255) / 256), dim3(256), 0, cx.s, w.iou_full, P, mask0, nmask, iou);
    CHECK(msam_check_launch("gather_iou"));

    // up-scaling (ConvT1 + LayerNorm2d + GELU + ConvT2 + GELU) and hyper-network product in one pass over the stream
    // (dec->low_res_dtype == MSAM_F16: `low_res` is an fp16 buffer - the AMG path, whose post-processing reads fp16 back)
    CHECK(msam_upscale_fused_out(w.keys, (chain ? 1 : 0) | (dec->up1_centred ? 2 : 0), P, dec->up1_w, dec->up1_b, dec->up_ln_w, dec->up_ln_b, 1e-6f, dec->up2_w,
                                 dec->up2_b, w.hyper, 128, mask0, nmask, low_res, dec->low_res_dtype == MSAM_F16 ? MSAM_F16 : MSAM_F32,
                                 cx.s));
#undef CHECK
#undef ADD_CAST
#undef ADD_CAST2
#undef LN
    return 0;
}
}  // namespace
