// Image -> token cross attention + out_proj + residual + LayerNorm of a two-way block (reference:
// segment_anything TwoWayAttentionBlock.cross_attn_image_to_token + norm4) in TOKEN-OWNER form: same folding as
// fold_i2t_kernel (decfold.hip),
//     S[j][(h,t)] = keys_j . K'_{h,t} / 4 + tabQ_{j,h} . k_{t,h} / 4,   P = softmax_t(S),
//     keys_j <- LayerNorm(keys_j + sum_{(h,t)} P[j][(h,t)] V'_{h,t} + bo),
// but every wave owns WHOLE image tokens: a wave takes a 16-token tile, holds all 64 score rows and all 256 output
// channels of those tokens in its own accumulators, and never exchanges anything with another wave:
//   * the stream tile goes global -> registers directly in MFMA B-operand layout (lane = token l & 15, 8 consecutive
//     channels 32 s + 8 (l >> 4) ..; 16 B per lane, 64 B contiguous per token and instruction) - no LDS staging, no
//     ds_write pass;
//   * the per-prompt operands K' (64 x 256), V'^T (256 x 64) and the block-diagonal table operand live in LDS in
//     FRAGMENT order (one MFMA A operand = 1 KiB contiguous, lane-major: conflict-free ds_read_b128 at immediate
//     offsets from one base register), written once per prompt; the four waves of a workgroup share them;
//   * the score rows are ordered so that one lane holds the 8 prompt tokens of a head (tiles 2a / 2a+1 = tokens 0..3 /
//     4..7 of heads 4a + (l >> 4)): the softmax over the prompt tokens needs no cross-lane step, and the normalised
//     probabilities ARE the B operand of the second product (k-slot 8 (l >> 4) + i = head 4a + (l >> 4), token i);
//   * the output rows are ordered so that a lane's accumulators of tiles 2c, 2c+1 are 8 CONSECUTIVE channels
//     32 c + 8 (l >> 4) ..: the LayerNorm result is packed and stored with the same 16-byte pattern the tile was loaded
//     with; the LayerNorm statistics are in-lane sums + two cross-lane steps;
//   * residual = one MFMA per output tile with a 0/1 selection matrix as A operand and the tile's own B fragment
//     (x * 1.0 is exact in the fp32 accumulator); out_proj bias: sum_t P[h][t] = 1 for every head, so bo / 8 is added to
//     every V'_{h,t} (t < Nt) by the fold kernel; 1/4 and log2(e) are folded into K' (softmax through v_exp_f32).
// No barrier and no LDS write in the tile loop (fold_i2t_kernel: three barriers, ~230 KB of LDS traffic per 32 tokens).
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);
void msam_profile_mark2(void* stream, int begin, double flops, double bytes, int family);

int g_tune_i2t_variant = 1;          // 1: token-owner kernel (this file), 0: fold_i2t_kernel (decfold.hip)
int g_tune_i2t_wg_per_cu = 2;
int g_tune_chain_tmask = 255;        // experiment: tile index mask of the SHARED-table loads of the ring kernels (255 = off)
int g_tune_chain_variant = 9;        // chained kernels: 9 (default) = second form of the attention (V projection before the LayerNorm) + tile loads
                                     // one phase ahead; 6 = first form, loads one phase ahead; 0 = first form, 4-fragment groups; 1 / 2 = 4 waves
                                     // (one per SIMD), rings of 8 / 16; 3 = 8 waves on the ring code; 4 / 5 / 7 / 8 = ring depth 3 / compact accumulators

namespace {

constexpr int T = 4096, C = 256, CI = 128, NTHR = 256;
constexpr int FRAG = 1024;                                  // one MFMA A operand: 64 lanes x 16 B
constexpr int KT_OFF = 0, KT_BYTES = 4 * 2 * FRAG;          // [m][e]      table operand (block diagonal k_{t,h})
constexpr int KF_OFF = KT_OFF + KT_BYTES, KF_BYTES = 4 * 8 * FRAG;     // [m][ks]     K'
constexpr int VF_OFF = KF_OFF + KF_BYTES, VF_BYTES = 2 * 16 * FRAG;    // [a][ct]     V'^T
constexpr int OPER_BYTES = VF_OFF + VF_BYTES;               // 73728 per prompt
constexpr float NEG_BIG = -1.0e30f;
constexpr float SCALE = 0.25f * 1.4426950408889634f;        // 1 / sqrt(16) and log2(e)

// Per prompt: the three operands in fragment order.  Fragment element (lane = fg * 16 + fr, i): A[row fr][k = 8 fg + i].
//   K' [m][ks]: row (m, fr) = (head 4 (m >> 1) + (fr >> 2), token 4 (m & 1) + (fr & 3)), k = channel 32 ks + 8 fg + i
//   KT [m][e] : same rows, k = table channel 32 (2 (m >> 1) + e) + 8 fg + i; k_{t,h} on the head's own 16 channels, else 0
//   V'^T [a][ct]: row fr = channel 32 (ct >> 1) + 8 (fr >> 2) + 4 (ct & 1) + (fr & 3), k = (head 4a + fg, token i)
__global__ __launch_bounds__(256) void fold_frag_kernel(const u16* __restrict__ ktok, const u16* __restrict__ vtok,
                                                        const u16* __restrict__ wq, const u16* __restrict__ wo,
                                                        const float* __restrict__ bo, int Nt, int with_kf,
                                                        u16* __restrict__ oper, const unsigned char* __restrict__ t2 = nullptr,
                                                        u16* __restrict__ mf = nullptr) {
    __shared__ __attribute__((aligned(16))) u16 img[OPER_BYTES / 2];
    __shared__ float kk[8][CI], vv[8][CI];
    const int p = blockIdx.x, c = threadIdx.x;
    for (int i = c; i < 8 * CI; i += 256) {
        const int t = i >> 7, d = i & (CI - 1);
        kk[t][d] = t < Nt ? d2f(ktok[((long)p * Nt + t) * CI + d]) : 0.f;
        vv[t][d] = t < Nt ? d2f(vtok[((long)p * Nt + t) * CI + d]) : 0.f;
    }
    for (int i = c; i < KT_BYTES / 16; i += 256) ((uint4*)img)[KT_OFF / 16 + i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // ---- K'[(h,t)][c] = SCALE * sum_d k[t][16h + d] Wq[16h + d][c]   (not needed by the layer-0 form on a shared source)
    if (with_kf) {
        const int ks = c >> 5, fg = (c >> 3) & 3, ii = c & 7;
        for (int h = 0; h < 8; ++h) {
            float w[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) w[d] = d2f(wq[(h * 16 + d) * C + c]);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < 16; ++d) acc = fmaf(kk[t][h * 16 + d], w[d], acc);
                const int m = 2 * (h >> 2) + (t >> 2), fr = 4 * (h & 3) + (t & 3);
                img[KF_OFF / 2 + ((m * 8 + ks) * 64 + fg * 16 + fr) * 8 + ii] = f2d(acc * SCALE);
            }
        }
    }
    // ---- table operand: SCALE * k[t][16h + d] at row (h,t), table channel 16h + d
    for (int j = c; j < 8 * 8 * 16; j += 256) {
        const int h = j >> 7, t = (j >> 4) & 7, d = j & 15;
        const int m = 2 * (h >> 2) + (t >> 2), e = (h >> 1) & 1, fr = 4 * (h & 3) + (t & 3), fg = 2 * (h & 1) + (d >> 3);
        img[KT_OFF / 2 + ((m * 2 + e) * 64 + fg * 16 + fr) * 8 + (d & 7)] = f2d(kk[t][h * 16 + d] * SCALE);
    }
    // ---- V'^T[c][(h,t)] = sum_d Wo[c][16h + d] v[t][16h + d] + bo[c] / 8   (t < Nt)
    {
        const int cp = c >> 5, within = c & 31, rho = 4 * (within >> 3) + (within & 3), ct = 2 * cp + ((within >> 2) & 1);
        const float bo8 = bo[c] * 0.125f;
        for (int h = 0; h < 8; ++h) {
            const uint4 w0 = *(const uint4*)(wo + (long)c * CI + h * 16), w1 = *(const uint4*)(wo + (long)c * CI + h * 16 + 8);
            const uint32_t ww[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            float wf[16];
#pragma unroll
            for (int x = 0; x < 8; ++x) { wf[2 * x] = d2f((u16)(ww[x] & 0xffff)); wf[2 * x + 1] = d2f((u16)(ww[x] >> 16)); }
            uint32_t pk[4];
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int d = 0; d < 16; ++d) { a0 = fmaf(wf[d], vv[2 * t2][h * 16 + d], a0); a1 = fmaf(wf[d], vv[2 * t2 + 1][h * 16 + d], a1); }
                pk[t2] = pack2d(2 * t2 < Nt ? a0 + bo8 : 0.f, 2 * t2 + 1 < Nt ? a1 + bo8 : 0.f);
            }
            const int a = h >> 2, fg = h & 3;
            *(uint4*)(img + VF_OFF / 2 + ((a * 16 + ct) * 64 + fg * 16 + rho) * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
    // ---- second form of the chained attention (t2 = its tables): M[(h,t)][d] = sum_e v[t][16h+e] WoWv[d][16h+e] + cd[d] / 8 as B
    // fragments [a2][dv] (see fold_values_kernel; the 64 KiB of WoWv come from L2)
    if (t2) {
        const int d = c & 127, hh = c >> 7;
        const float* wowv = (const float*)t2 + d * 128;                      // T2_WOWV = 0
        const float cd8 = ((const float*)(t2 + 128 * 128 * 4))[d] * 0.125f;  // T2_CD
        u16* mdst = mf + (long)p * (16 * FRAG / 2);
        for (int h = hh * 4; h < hh * 4 + 4; ++h) {
            float wv_[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) wv_[e] = wowv[h * 16 + e];
            uint32_t pk[4];
#pragma unroll
            for (int t2_ = 0; t2_ < 4; ++t2_) {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { a0 = fmaf(wv_[e], vv[2 * t2_][h * 16 + e], a0); a1 = fmaf(wv_[e], vv[2 * t2_ + 1][h * 16 + e], a1); }
                pk[t2_] = pack2d(2 * t2_ < Nt ? a0 + cd8 : 0.f, 2 * t2_ + 1 < Nt ? a1 + cd8 : 0.f);
            }
            *(uint4*)(mdst + ((((h >> 2) * 8 + (d >> 4)) * 64 + (h & 3) * 16 + (d & 15)) * 8)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
    __syncthreads();
    uint4* dst = (uint4*)(oper + (long)p * (OPER_BYTES / 2));
    for (int i = c; i < OPER_BYTES / 16; i += 256) dst[i] = ((const uint4*)img)[i];
}


// ---- per-wave constants shared by the kernels below
struct WaveConst {
    uint4 sel[2];          // residual selection operands: tile 2c + e, row rho picks k = 8 (rho >> 2) + 4 e + (rho & 3) of k-step c
    f32x4_t sinit[2];      // initial score accumulators: 0, or -BIG for the rows of absent prompt tokens (token 4 (m & 1) + r >= Nt)
};
MSAM_DEVINL WaveConst wave_const(int fr, int fg, int Nt) {
    WaveConst k;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        uint32_t wv[4] = {0u, 0u, 0u, 0u};
        if (fg == (fr >> 2)) { const int i = 4 * e + (fr & 3); wv[i >> 1] = (i & 1) ? (MSAM_D16_ONE << 16) : MSAM_D16_ONE; }
        k.sel[e] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int r = 0; r < 4; ++r) k.sinit[par][r] = 4 * par + r < Nt ? 0.f : NEG_BIG;
    return k;
}

#define TK_RD4(dst_, off_, stride_) do { _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) dst_[j_] = *(const uint4*)(L + (off_) + j_ * (stride_)); } while (0)
#define TK_FENCE() __builtin_amdgcn_sched_barrier(0)

// One 16-token tile through the image->token layer.  L = lds + lane * 16; KT / KF / VF = byte offsets of the operand
// images in LDS; gp = ln_w + 8 (lane >> 4) in LDS (ln_b follows C floats later).  b[8]: the tile as B fragments (residual,
// and the K' product when HAS_KF); tb[4]: B fragments of the table term: tabQ tile (HAS_KF) or, for layer 0 on the SHARED
// source, the prompt-independent q = (src + pe) Wq^T + bq tile itself (S = q . k, no K' product).  y[8]: the LayerNorm output
// as packed 16-bit values in the same fragment layout as b.  consumed() runs when b / tb have been read for the last time
// (the caller issues its prefetch into them there).
// The A operands come from LDS four at a time into two alternating register groups: the reads of group g+1 are in flight
// while the MFMAs of group g issue (left to itself the compiler reads one fragment, waits for it and multiplies).
template <int KT, int KF, int VF, bool HAS_KF, class F>
MSAM_DEVINL void i2t_block(const unsigned char* L, const float* gp, float eps, const uint4* b, const uint4* tb,
                           const WaveConst& wc, uint4* y, F&& consumed) {
    f32x4_t s[4];
    uint4 fa[4], fb[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) s[m] = wc.sinit[m & 1];
    constexpr int G0 = HAS_KF ? 0 : 8;                   // groups 0..7: K' k-steps, 8..9: table operand
    if (HAS_KF) TK_RD4(fa, KF, 8 * FRAG); else TK_RD4(fa, KT, 2 * FRAG);
    TK_FENCE();
#pragma unroll
    for (int g = G0; g < 10; ++g) {
        uint4* cur = (g & 1) ? fb : fa;
        uint4* nxt = (g & 1) ? fa : fb;
        if (g + 1 < 8) TK_RD4(nxt, KF + (g + 1) * FRAG, 8 * FRAG);
        else if (g + 1 < 10) TK_RD4(nxt, KT + (g + 1 - 8) * FRAG, 2 * FRAG);
        else TK_RD4(nxt, VF, FRAG);                      // first group of the second product
#pragma unroll
        for (int m = 0; m < 4; ++m)
            s[m] = mfma16d(cur[m], g < 8 ? b[g < 8 ? g : 0] : tb[2 * (m >> 1) + (g - 8)], s[m]);
        TK_FENCE();
    }
    // ---- O^T accumulators start as the residual
    f32x4_t o[16];
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) o[ct] = mfma16d(wc.sel[ct & 1], b[ct >> 1], f32x4_t{0.f, 0.f, 0.f, 0.f});
    consumed();
    TK_FENCE();
    // ---- softmax over the 8 prompt tokens of a head, in lane; P packed = B operand of the second product
    uint4 pk[2];
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {
        const f32x4_t u = s[2 * a2], v = s[2 * a2 + 1];
        const float mx = fmaxf(fmaxf(fmaxf(u[0], u[1]), fmaxf(u[2], u[3])), fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        float eu[4], ev[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { eu[r] = __builtin_amdgcn_exp2f(u[r] - mx); ev[r] = __builtin_amdgcn_exp2f(v[r] - mx); }
        const float l = ((eu[0] + eu[1]) + (eu[2] + eu[3])) + ((ev[0] + ev[1]) + (ev[2] + ev[3]));
        const float inv = __builtin_amdgcn_rcpf(l);          // 1 ulp; P is rounded to 16 bits next
        pk[a2] = make_uint4(pack2d(eu[0] * inv, eu[1] * inv), pack2d(eu[2] * inv, eu[3] * inv),
                            pack2d(ev[0] * inv, ev[1] * inv), pack2d(ev[2] * inv, ev[3] * inv));
    }
    TK_FENCE();
    // ---- O^T += V'^T P^T: 8 groups of four output tiles (group q: k-step a2 = q >> 2, tiles 4 (q & 3) ..)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint4* cur = (q & 1) ? fb : fa;                   // group 0 was read into fa by the last score group (g = 9)
        uint4* nxt = (q & 1) ? fa : fb;
        if (q + 1 < 8) TK_RD4(nxt, VF + (q + 1) * 4 * FRAG, FRAG);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * (q & 3) + j] = mfma16d(cur[j], pk[q >> 2], o[4 * (q & 3) + j]);
        TK_FENCE();
    }
    // ---- LayerNorm over the 256 channels of token fr: 64 values in this lane, the rest in lanes fr + 16 g
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < 16; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1 += o[ct][r]; s2 = fmaf(o[ct][r], o[ct][r], s2); }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    const float mean = s1 * (1.f / C);
    const float rstd = rsqrtf(fmaxf(s2 * (1.f / C) - mean * mean, 0.f) + eps);
    const float nmr = -mean * rstd;
#pragma unroll
    for (int c2 = 0; c2 < 8; ++c2) {
        const float4 g0 = *(const float4*)(gp + c2 * 32), g1 = *(const float4*)(gp + c2 * 32 + 4);
        const float4 h0 = *(const float4*)(gp + C + c2 * 32), h1 = *(const float4*)(gp + C + c2 * 32 + 4);
        const f32x4_t x0 = o[2 * c2], x1 = o[2 * c2 + 1];
        y[c2].x = pack2d(fmaf(fmaf(x0[0], rstd, nmr), g0.x, h0.x), fmaf(fmaf(x0[1], rstd, nmr), g0.y, h0.y));
        y[c2].y = pack2d(fmaf(fmaf(x0[2], rstd, nmr), g0.z, h0.z), fmaf(fmaf(x0[3], rstd, nmr), g0.w, h0.w));
        y[c2].z = pack2d(fmaf(fmaf(x1[0], rstd, nmr), g1.x, h1.x), fmaf(fmaf(x1[1], rstd, nmr), g1.y, h1.y));
        y[c2].w = pack2d(fmaf(fmaf(x1[2], rstd, nmr), g1.z, h1.z), fmaf(fmaf(x1[3], rstd, nmr), g1.w, h1.w));
    }
}

struct TokArgs {
    const u16* xin; int x_shared;        // d16 [Px, 4096, 256]
    const u16* oper;                     // d16 [P, OPER_BYTES / 2]  (fold_frag_kernel)
    const u16* tabq;                     // d16 [4096, 128]
    const float* ln_w; const float* ln_b; float eps;
    int Nt, nitems, KS;
    u16* out;                            // d16 [P, 4096, 256] (may alias xin)
};

__global__ __launch_bounds__(NTHR, 2) void i2t_tok_kernel(TokArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[OPER_BYTES + 2 * C * 4];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    const int ks_sh = __builtin_ctz(a.KS), tpi_sh = 8 - ks_sh;          // 256 16-token tiles per prompt
    const int TPI = 1 << tpi_sh, nt = TPI >> 2;                         // tiles per item / per wave
    float* prm = (float*)(lds + OPER_BYTES);                            // ln_w [256], ln_b [256]
    prm[tid] = a.ln_w[tid]; prm[C + tid] = a.ln_b[tid];
    const WaveConst wc = wave_const(fr, fg, a.Nt);
    const int voff = fr * (C * 2) + fg * 16;                 // stream / output: token row fr, 16-byte slot fg of a k-step
    const int tvoff = fr * (CI * 2) + fg * 16;               // table
    const rsrc_t rtab = make_rsrc(a.tabq, T * CI * 2);
    const unsigned char* const L = lds + lane * 16;
    const float* const gp = prm + fg * 8;

    for (int item = (int)blockIdx.x; item < a.nitems; item += (int)gridDim.x) {
        const int p = __builtin_amdgcn_readfirstlane(item >> ks_sh), split = __builtin_amdgcn_readfirstlane(item & (a.KS - 1));
        __syncthreads();                                     // every wave is done with the previous prompt's operands
        {
            const uint4* src = (const uint4*)(a.oper + (long)p * (OPER_BYTES / 2));
#pragma unroll 6
            for (int i = tid; i < OPER_BYTES / 16; i += NTHR) ((uint4*)lds)[i] = src[i];
        }
        __syncthreads();
        const rsrc_t rx = make_rsrc(a.xin + (long)(a.x_shared ? 0 : p) * T * C, T * C * 2);
        const rsrc_t ro = make_rsrc(a.out + (long)p * T * C, T * C * 2);
        const int tile0 = split * TPI + w;                   // this wave: tiles tile0 + 4 n

        uint4 b[8], tb[4];
#define TK_LOAD(tile_)                                                                             \
        do {                                                                                       \
            const int so_ = (tile_) * (16 * C * 2), to_ = (tile_) * (16 * CI * 2);                 \
            _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) b[s_] = buf_load16(rx, voff, so_ + s_ * 64);      \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) tb[s_] = buf_load16(rtab, tvoff, to_ + s_ * 64);  \
        } while (0)
        TK_LOAD(tile0);
        // wait for the first tile HERE: the waits inside the loop are then set by the back edge alone (prefetch of the next tile
        // with 8 younger stores: vmcnt(8)); entered with loads pending and no stores the loop header would need vmcnt(0),
        // i.e. every iteration would also wait for the stores of the previous tile to reach L2
        wait_vmem_all();
        for (int n = 0; n < nt; ++n) {
            // the operands in LDS are loop invariant: keep their reads inside the loop (hoisted they would need 400+ registers)
            asm volatile("" ::: "memory");
            const int tile = tile0 + 4 * n;
            uint4 y[8];
            i2t_block<KT_OFF, KF_OFF, VF_OFF, true>(L, gp, a.eps, b, tb, wc, y, [&]() {
                // next tile of this wave into the (now free) fragment registers; unconditional (clamped) so that the compiler can
                // count the younger stores and wait with vmcnt(n > 0) at the top of the next iteration
                const int nx = tile0 + 4 * (n + 1 < nt ? n + 1 : nt - 1);
                TK_LOAD(nx);
            });
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2) buf_store16(y[c2], ro, voff, tile * (16 * C * 2) + c2 * 64);
        }
#undef TK_LOAD
    }
}

// ============================================================================================================
// Layer 0 on the SHARED source chained into layer 1 (AMG: every prompt of a tile starts from the same embedding).
// keys1 = layer-0 output is never written: a wave recomputes its 16-token tile of keys1 from the L2-resident source
// (src tile for the residual, q0 = (src + pe) Wq0^T + bq0 tile for the scores: two table-operand MFMAs per score tile instead of
// the folded K' product) and feeds it - packed to 16 bits exactly as the stream would have stored it - straight into
//   * i2t01_kernel: the layer-1 image->token block (its B fragments ARE the packed layer-0 output), result = keys2 -> HBM;
//   * i2t0_t2i_kernel: the layer-1 token->image attention in token-owner form (below).
// HBM traffic of the pair: 2 MiB per prompt written (keys2) instead of 2 written + 2 read + 2 read + 2 written.
//
// BLOCKED layout of everything these kernels load and store per tile.  A B / A operand fragment is "lane l & 15 = token, 8
// channels 32 s + 8 (l >> 4) ..": on a row-major [token][channel] matrix consecutive lanes are a whole row (256 / 512 B) apart,
// every lane is its own 16-byte request (64 per instruction instead of 8 x 128 B; measured: the loads alone set a floor of
// 0.5 ms per 1024-prompt launch, profiles/r02_experiments.md).  The shared tables are therefore copied once per decode into
//     blocked[tile of 16 tokens][k-step s][lane = 16 (l >> 4) + (l & 15)][8 channels]        (msam_chain_prepare_tables)
// so that one fragment load is 1 KiB contiguous, lane-major; and the layer-1 output stream is WRITTEN in the same blocked
// form (a tile stays 8 KiB contiguous): its two consumers, fold_attn_kernel and up_fused_kernel, stage whole tiles with
// linear 16-byte chunks and only map chunk -> (token, channel chunk) differently (their `blocked` argument).
struct ChainArgs {
    const u16* src;                      // d16 blocked [256 tiles][8][64][8]   shared source (image embedding + no-mask embedding)
    const u16* q0;                       // d16 blocked [256 tiles][4][64][8]   (src + pe) Wq0^T + bq0
    const u16* oper0;                    // d16 [P, OPER_BYTES / 2] layer-0 operands (KT and VF parts used)
    const float* ln0_w; const float* ln0_b;
    const u16* oper1;                    // d16 [P, OPER_BYTES / 2] layer-1 operands
    const u16* tabq1;                    // d16 blocked [256 tiles][4][64][8]
    const float* ln1_w; const float* ln1_b;
    float eps; int Nt, P;
    u16* out;                            // d16 [P] blocked [256 tiles][8][64][8]  keys2
};

constexpr int NTHR8 = 512;
// LDS images of the chained kernels (byte offsets)
constexpr int C_KT0 = 0, C_VF0 = C_KT0 + KT_BYTES, C_KT1 = C_VF0 + VF_BYTES, C_KF1 = C_KT1 + KT_BYTES, C_VF1 = C_KF1 + KF_BYTES,
              C_PRM = C_VF1 + VF_BYTES, C_LDS = C_PRM + 4 * C * 4;

__global__ __launch_bounds__(NTHR8, 2) void i2t01_kernel(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[C_LDS];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    float* prm = (float*)(lds + C_PRM);                      // ln0_w, ln0_b, ln1_w, ln1_b
    if (tid < C) { prm[tid] = a.ln0_w[tid]; prm[C + tid] = a.ln0_b[tid]; }
    else { prm[2 * C + tid - C] = a.ln1_w[tid - C]; prm[3 * C + tid - C] = a.ln1_b[tid - C]; }
    const WaveConst wc = wave_const(fr, fg, a.Nt);
    const int voff = lane * 16, tvoff = lane * 16;      // blocked tables / stream: one fragment = 1 KiB contiguous, lane-major
    const rsrc_t rsrc = make_rsrc(a.src, T * C * 2), rq0 = make_rsrc(a.q0, T * CI * 2), rtab = make_rsrc(a.tabq1, T * CI * 2);
    const unsigned char* const L = lds + lane * 16;
    const float* const gp0 = prm + fg * 8;
    const float* const gp1 = prm + 2 * C + fg * 8;
    constexpr int NT = 256 / 8;                              // tiles per wave and prompt

    for (int p = (int)blockIdx.x; p < a.P; p += (int)gridDim.x) {
        __syncthreads();
        {
            const uint4* s0 = (const uint4*)(a.oper0 + (long)p * (OPER_BYTES / 2));
            const uint4* s1 = (const uint4*)(a.oper1 + (long)p * (OPER_BYTES / 2));
            for (int i = tid; i < KT_BYTES / 16; i += NTHR8) ((uint4*)(lds + C_KT0))[i] = s0[KT_OFF / 16 + i];
            for (int i = tid; i < VF_BYTES / 16; i += NTHR8) ((uint4*)(lds + C_VF0))[i] = s0[VF_OFF / 16 + i];
#pragma unroll 3
            for (int i = tid; i < OPER_BYTES / 16; i += NTHR8) ((uint4*)(lds + C_KT1))[i] = s1[i];   // KT, KF, VF contiguous
        }
        __syncthreads();
        const rsrc_t ro = make_rsrc(a.out + (long)p * T * C, T * C * 2);
        uint4 b[8], qi[4], tb[4];
#define CH_LOAD0(tile_)                                                                            \
        do {                                                                                       \
            const int so_ = (tile_) * (16 * C * 2), to_ = (tile_) * (16 * CI * 2);                 \
            _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) b[s_] = buf_load16(rsrc, voff, so_ + s_ * FRAG);    \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) qi[s_] = buf_load16(rq0, tvoff, to_ + s_ * FRAG);   \
        } while (0)
#define CH_LOAD1(tile_)                                                                            \
        do {                                                                                       \
            const int to_ = (tile_) * (16 * CI * 2);                                               \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) tb[s_] = buf_load16(rtab, tvoff, to_ + s_ * FRAG);  \
        } while (0)
        CH_LOAD0(w); CH_LOAD1(w);
        wait_vmem_all();                                     // see i2t_tok_kernel
        for (int n = 0; n < NT; ++n) {
            asm volatile("" ::: "memory");
            const int tile = w + 8 * n, nx = w + 8 * (n + 1 < NT ? n + 1 : NT - 1);
            uint4 y1[8], y2[8];
            i2t_block<C_KT0, 0, C_VF0, false>(L, gp0, a.eps, b, qi, wc, y1, [&]() { CH_LOAD0(nx); });
            i2t_block<C_KT1, C_KF1, C_VF1, true>(L, gp1, a.eps, y1, tb, wc, y2, [&]() { CH_LOAD1(nx); });
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2) buf_store16(y2[c2], ro, voff, tile * (16 * C * 2) + c2 * FRAG);
        }
#undef CH_LOAD0
#undef CH_LOAD1
    }
}

// ---- token -> image attention in token-owner form, on a tile that is already in registers.
// The folded form (fold_attn_kernel) keeps O'[(h,t)][256 channels] per wave - 64 x 256 accumulators - which is why its
// waves own (h,t) columns and all read the whole stream tile from LDS.  A wave that owns TOKENS instead projects its
// tile to V = keys Wv^T first (8 heads x 16 channels) and accumulates O^T[d][(h,t)] per head: 32 accumulator registers.
//   S[j][(h,t)]   = keys_j . Q'_{h,t} + tabK_j . q_{h,t}       A = the tile (rows = tokens), B = Q' / block-diagonal q (LDS)
//   V[j][16h + d] = keys_j . Wv[16h + d]                        A = the tile, B = Wv rows (LDS, prompt independent)
//   O^T_h[d][(h',t)] += sum_j V[j][16h + d] P[j][(h',t)]        A = V's accumulators, B = P's accumulators: an accumulator
//        tile C[row = 4 (l >> 4) + r][col = l & 15] read as an operand has k = row, i.e. k = token for both - the transposes
//        are free; only the columns of head h' = h are kept.  K = 16 tokens of the 32 k-slots (upper slots zero).
// Softmax over the keys is online with a LAZY reference maximum per column: P = 2^(S - m_ref) with m_ref only raised (and
// the accumulators rescaled) when some score exceeds it by more than 2^8 (one ballot per tile instead of cross-lane
// maxima); the per-lane partial sums of P are reduced once per prompt.
struct AttnState { f32x4_t o[8]; float m[4], l[4]; };

template <int QD, int QF, int WV, class F>
MSAM_DEVINL void t2i_block(const unsigned char* L, const uint4* y, const uint4* tk, AttnState& st, F&& consumed) {
    f32x4_t s[4];
    uint4 fa[4], fb[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) s[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    TK_RD4(fa, QF, 8 * FRAG);
    TK_FENCE();
#pragma unroll
    for (int g = 0; g < 9; ++g) {                            // groups 0..7: Q' k-steps, 8: block-diagonal q against the table tile
        uint4* cur = (g & 1) ? fb : fa;
        uint4* nxt = (g & 1) ? fa : fb;
        if (g + 1 < 8) TK_RD4(nxt, QF + (g + 1) * FRAG, 8 * FRAG);
        else if (g + 1 < 9) TK_RD4(nxt, QD, FRAG);
        else TK_RD4(nxt, WV, 8 * FRAG);                      // first group of the V projection
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) s[ct] = mfma16d(g < 8 ? y[g < 8 ? g : 0] : tk[ct], cur[ct], s[ct]);
        TK_FENCE();
    }
    consumed();
    // ---- lazy online softmax over the keys (rows), per column fr of the four column tiles
    float mloc[4];
    bool need = false;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        mloc[ct] = fmaxf(fmaxf(s[ct][0], s[ct][1]), fmaxf(s[ct][2], s[ct][3]));
        need = need || (mloc[ct] > st.m[ct] + 8.f);
    }
    if (__builtin_amdgcn_ballot_w64(need) != 0) {            // wave-uniform: the reference maximum of some column is stale
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float mx = mloc[ct];
            mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(st.m[ct], mx), alpha = __builtin_amdgcn_exp2f(st.m[ct] - mn);
            st.m[ct] = mn; st.l[ct] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) { st.o[2 * ct][r] *= alpha; st.o[2 * ct + 1][r] *= alpha; }
        }
    }
    uint4 pb[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(s[ct][r] - st.m[ct]);
        st.l[ct] += (e[0] + e[1]) + (e[2] + e[3]);
        pb[ct] = make_uint4(pack2d(e[0], e[1]), pack2d(e[2], e[3]), 0u, 0u);
    }
    TK_FENCE();
    // ---- V projection, four heads at a time, then O^T_h += V_h^T P
#pragma unroll
    for (int hv = 0; hv < 2; ++hv) {
        f32x4_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int g = hv * 8 + ks;                       // group g: heads 4 hv .. + 3, k-step ks; group 0 was read into fb
            uint4* cur = (g & 1) ? fa : fb;                   // by the last score group (g = 8: current fa, next fb)
            uint4* nxt = (g & 1) ? fb : fa;
            if (g + 1 < 16) TK_RD4(nxt, WV + (((g + 1) >> 3) * 32 + ((g + 1) & 7)) * FRAG, 8 * FRAG);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = mfma16d(y[ks], cur[j], v[j]);
            TK_FENCE();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int h = 4 * hv + j;
            const uint4 va = make_uint4(pack2d(v[j][0], v[j][1]), pack2d(v[j][2], v[j][3]), 0u, 0u);
            st.o[h] = mfma16d(va, pb[h >> 1], st.o[h]);
        }
        TK_FENCE();
    }
}

struct FuseArgs {
    const u16* src; const u16* q0;       // shared source and its layer-0 q (see ChainArgs)
    const u16* oper0;                    // layer-0 image->token operands
    const float* ln0_w; const float* ln0_b; float eps;
    const u16* aoper;                    // d16 [P, AOPER_BYTES / 2]  layer-1 token->image operands: QD, QF (fold_attnfrag_kernel)
    const u16* wvfrag;                   // d16 [WV_BYTES / 2]        Wv in fragment order (wv_frag_kernel)
    const u16* tabk;                     // d16 blocked [256 tiles][4][64][8]  pe Wk^T + bk
    const float* bv;                     // fp32 [128]
    int Nt, P;
    u16* out;                            // d16 [P, Nt, 128]
    int tmask;                           // experiment knob (255): tile index mask of the shared-table loads
};
constexpr int QD_BYTES = 4 * FRAG, QF_BYTES = 4 * 8 * FRAG, AOPER_BYTES = QD_BYTES + QF_BYTES, WV_BYTES = 8 * 8 * FRAG;
constexpr int F_KT0 = 0, F_VF0 = F_KT0 + KT_BYTES, F_QD = F_VF0 + VF_BYTES, F_QF = F_QD + QD_BYTES, F_WV = F_QF + QF_BYTES,
              F_PRM = F_WV + WV_BYTES, F_LDS = F_PRM + 2 * C * 4;
constexpr int MERGE_FLOATS = 64 + 64 + 64 * 16;              // per wave: m, l, O[(h,t)][d]
static_assert(8 * MERGE_FLOATS * 4 <= F_WV, "merge scratch overlays the per-prompt operands");

__global__ __launch_bounds__(NTHR8, 2) void i2t0_t2i_kernel(FuseArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[F_LDS];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    float* prm = (float*)(lds + F_PRM);
    if (tid < C) prm[tid] = a.ln0_w[tid]; else prm[tid] = a.ln0_b[tid - C];
    for (int i = tid; i < WV_BYTES / 16; i += NTHR8) ((uint4*)(lds + F_WV))[i] = ((const uint4*)a.wvfrag)[i];
    const WaveConst wc = wave_const(fr, fg, a.Nt);
    const int voff = lane * 16, tvoff = lane * 16;      // blocked tables / stream: one fragment = 1 KiB contiguous, lane-major
    const rsrc_t rsrc = make_rsrc(a.src, T * C * 2), rq0 = make_rsrc(a.q0, T * CI * 2), rtab = make_rsrc(a.tabk, T * CI * 2);
    const unsigned char* const L = lds + lane * 16;
    const float* const gp0 = prm + fg * 8;
    constexpr int NT = 256 / 8;

    for (int p = (int)blockIdx.x; p < a.P; p += (int)gridDim.x) {
        __syncthreads();                                     // merge of the previous prompt done
        {
            const uint4* s0 = (const uint4*)(a.oper0 + (long)p * (OPER_BYTES / 2));
            const uint4* s1 = (const uint4*)(a.aoper + (long)p * (AOPER_BYTES / 2));
            for (int i = tid; i < KT_BYTES / 16; i += NTHR8) ((uint4*)(lds + F_KT0))[i] = s0[KT_OFF / 16 + i];
            for (int i = tid; i < VF_BYTES / 16; i += NTHR8) ((uint4*)(lds + F_VF0))[i] = s0[VF_OFF / 16 + i];
            for (int i = tid; i < AOPER_BYTES / 16; i += NTHR8) ((uint4*)(lds + F_QD))[i] = s1[i];        // QD, QF contiguous
        }
        __syncthreads();
        AttnState st;
#pragma unroll
        for (int h = 0; h < 8; ++h) st.o[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { st.m[ct] = NEG_BIG; st.l[ct] = 0.f; }
        uint4 b[8], qi[4], tk[4];
#define FU_LOAD0(tile_)                                                                            \
        do {                                                                                       \
            const int so_ = (tile_) * (16 * C * 2), to_ = (tile_) * (16 * CI * 2);                 \
            _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) b[s_] = buf_load16(rsrc, voff, so_ + s_ * FRAG);    \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) qi[s_] = buf_load16(rq0, tvoff, to_ + s_ * FRAG);   \
        } while (0)
#define FU_LOAD1(tile_)                                                                            \
        do {                                                                                       \
            const int to_ = (tile_) * (16 * CI * 2);                                               \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) tk[s_] = buf_load16(rtab, tvoff, to_ + s_ * FRAG);  \
        } while (0)
        FU_LOAD0(w); FU_LOAD1(w);
        for (int n = 0; n < NT; ++n) {
            asm volatile("" ::: "memory");
            const int nx = w + 8 * (n + 1 < NT ? n + 1 : NT - 1);
            uint4 y1[8];
            i2t_block<F_KT0, 0, F_VF0, false>(L, gp0, a.eps, b, qi, wc, y1, [&]() { FU_LOAD0(nx); });
            t2i_block<F_QD, F_QF, F_WV>(L, y1, tk, st, [&]() { FU_LOAD1(nx); });
        }
#undef FU_LOAD0
#undef FU_LOAD1
        // ---- merge the eight waves' (m, l, O^T) and write out[p][t][16h + d] = O / l + bv
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { st.l[ct] += __shfl_xor(st.l[ct], 16); st.l[ct] += __shfl_xor(st.l[ct], 32); }
        __syncthreads();                                     // every wave is done with the operands this scratch overlays
        float* mg = (float*)lds + w * MERGE_FLOATS;
        if (fg == 0) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { mg[ct * 16 + fr] = st.m[ct]; mg[64 + ct * 16 + fr] = st.l[ct]; }
        }
#pragma unroll
        for (int h = 0; h < 8; ++h)
            if ((fr >> 3) == (h & 1))                        // column fr of column tile h >> 1 belongs to head h
                *(float4*)(mg + 128 + (h * 8 + (fr & 7)) * 16 + fg * 4) = make_float4(st.o[h][0], st.o[h][1], st.o[h][2], st.o[h][3]);
        __syncthreads();
        {
            const int col = tid >> 3, d0 = (tid & 7) * 2, h = col >> 3, t = col & 7;
            const float* g0 = (const float*)lds;
            float mx = NEG_BIG;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) mx = fmaxf(mx, g0[ww * MERGE_FLOATS + col]);
            float lt = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                const float* gw = g0 + ww * MERGE_FLOATS;
                const float sc = __builtin_amdgcn_exp2f(gw[col] - mx);
                lt = fmaf(sc, gw[64 + col], lt);
                o0 = fmaf(sc, gw[128 + col * 16 + d0], o0); o1 = fmaf(sc, gw[128 + col * 16 + d0 + 1], o1);
            }
            if (t < a.Nt) {
                const float inv = 1.f / lt;
                *(uint32_t*)(a.out + ((long)p * a.Nt + t) * CI + h * 16 + d0) =
                    pack2d(fmaf(o0, inv, a.bv[h * 16 + d0]), fmaf(o1, inv, a.bv[h * 16 + d0 + 1]));
            }
        }
    }
}

// ============================================================================================================
// The same two chained kernels with the LDS operand reads organised as ONE stream per tile: every fragment of a tile, in the
// order the MFMAs consume it, is fragment n of the stream; they are read in batches of G into a two-slot register ring, batch
// k+1 while batch k is consumed - across the phase boundaries too (the batch after the last score batch is the first batch of
// the next product), and across the tile boundary (the operands do not change within a prompt).  G = 4 is the scheme of the
// kernels above; G = 8 / 16 keep 8 / 16 ds_read_b128 per wave in flight (the LDS latency under load is ~3 of the 16-cycle
// MFMAs per batch of 4), which needs the registers of a 4-wave workgroup with ONE wave per SIMD (NW = 4: 512 VGPRs per wave).
template <int G, int D = 2> struct Ring { uint4 buf[D][G]; };     // D slots: batches k .. k + D - 2 in flight behind batch k

// ABL (ablation builds for tools/chain_ablation.py only; results are then meaningless): bit 0 no layer-0 score MFMAs, 1 no
// layer-0 V'^T MFMAs, 2 no LayerNorm, 3 no attention score MFMAs, 4 no V-projection MFMAs, 5 no LDS operand reads, 6 no softmax exps
template <int G, class Off, int ABL = 0>
MSAM_DEVINL void ring_fill(Ring<G, Off::D>& r, const unsigned char* L, int batch) {
    if (ABL & 32) return;
#pragma unroll
    for (int j = 0; j < G; ++j) r.buf[batch % Off::D][j] = *(const uint4*)(L + Off::off((batch * G + j) % Off::N));
}
// at fragment n_ of the tile's stream: when it opens a batch, issue the reads of the next one
#define RING_STEP(n_) do { if ((n_) % G == 0) { TK_FENCE(); ring_fill<G, Off, ABL>(ring, L, (n_) / G + Off::D - 1); TK_FENCE(); } } while (0)
#define RING_AT(n_) ring.buf[((n_) / G) % Off::D][(n_) % G]

// image->token block (see i2t_block) on the fragments BASE .. of the stream: [8 or 40 score fragments][32 V'^T fragments]
// AFFINE = false: y = the normalised values WITHOUT the LayerNorm weight / bias (the consumer has them folded into its operands);
// pk_out / stat_out (optional): the packed probabilities of the block and (rstd, -mean rstd) of the lane's token (l & 15).
template <int G, class Off, int BASE, bool HAS_KF, int ABL = 0, bool AFFINE = true, class F>
MSAM_DEVINL void i2t_block_r(Ring<G, Off::D>& ring, const unsigned char* L, const float* gp, float eps, const uint4* b, const uint4* tb,
                             const WaveConst& wc, uint4* y, F&& consumed, uint4* pk_out = nullptr, float* stat_out = nullptr) {
    constexpr int NS = HAS_KF ? 40 : 8, G0 = HAS_KF ? 0 : 8;
    f32x4_t s[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) s[m] = wc.sinit[m & 1];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int n = BASE + i, g = G0 + i / 4, m = i % 4;
        RING_STEP(n);
        if (!(ABL & 1)) s[m] = mfma16d(RING_AT(n), g < 8 ? b[g < 8 ? g : 0] : tb[2 * (m >> 1) + (g < 8 ? 0 : g - 8)], s[m]);
        else if (i < 4) s[m][0] += __uint_as_float((RING_AT(n).x ^ tb[m].x) & 0x3fffffffu);
    }
    f32x4_t o[16];
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) o[ct] = mfma16d(wc.sel[ct & 1], b[ct >> 1], f32x4_t{0.f, 0.f, 0.f, 0.f});
    consumed();
    uint4 pk[2];
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {
        const f32x4_t u = s[2 * a2], v = s[2 * a2 + 1];
        const float mx = fmaxf(fmaxf(fmaxf(u[0], u[1]), fmaxf(u[2], u[3])), fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        float eu[4], ev[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            eu[r] = (ABL & 64) ? u[r] - mx : __builtin_amdgcn_exp2f(u[r] - mx); ev[r] = (ABL & 64) ? v[r] - mx : __builtin_amdgcn_exp2f(v[r] - mx);
        }
        const float l = ((eu[0] + eu[1]) + (eu[2] + eu[3])) + ((ev[0] + ev[1]) + (ev[2] + ev[3]));
        const float inv = __builtin_amdgcn_rcpf(l);
        pk[a2] = make_uint4(pack2d(eu[0] * inv, eu[1] * inv), pack2d(eu[2] * inv, eu[3] * inv),
                            pack2d(ev[0] * inv, ev[1] * inv), pack2d(ev[2] * inv, ev[3] * inv));
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int n = BASE + NS + i;
        RING_STEP(n);
        if (!(ABL & 2) || i % 16 == 0) o[i % 16] = mfma16d(RING_AT(n), pk[i / 16], o[i % 16]);
    }
    if (ABL & 4) {
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2)
            y[c2] = make_uint4(pack2d(o[2 * c2][0], o[2 * c2][1]), pack2d(o[2 * c2][2], o[2 * c2][3]),
                               pack2d(o[2 * c2 + 1][0], o[2 * c2 + 1][1]), pack2d(o[2 * c2 + 1][2], o[2 * c2 + 1][3]));
        return;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < 16; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1 += o[ct][r]; s2 = fmaf(o[ct][r], o[ct][r], s2); }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    const float mean = s1 * (1.f / C);
    const float rstd = rsqrtf(fmaxf(s2 * (1.f / C) - mean * mean, 0.f) + eps);
    const float nmr = -mean * rstd;
    if (pk_out) { pk_out[0] = pk[0]; pk_out[1] = pk[1]; }
    if (stat_out) { stat_out[0] = rstd; stat_out[1] = nmr; }
    if (!AFFINE) {
#pragma unroll
        for (int c2 = 0; c2 < 8; ++c2) {
            const f32x4_t x0 = o[2 * c2], x1 = o[2 * c2 + 1];
            y[c2] = make_uint4(pack2d(fmaf(x0[0], rstd, nmr), fmaf(x0[1], rstd, nmr)), pack2d(fmaf(x0[2], rstd, nmr), fmaf(x0[3], rstd, nmr)),
                               pack2d(fmaf(x1[0], rstd, nmr), fmaf(x1[1], rstd, nmr)), pack2d(fmaf(x1[2], rstd, nmr), fmaf(x1[3], rstd, nmr)));
        }
        return;
    }
#pragma unroll
    for (int c2 = 0; c2 < 8; ++c2) {
        const float4 g0 = *(const float4*)(gp + c2 * 32), g1 = *(const float4*)(gp + c2 * 32 + 4);
        const float4 h0 = *(const float4*)(gp + C + c2 * 32), h1 = *(const float4*)(gp + C + c2 * 32 + 4);
        const f32x4_t x0 = o[2 * c2], x1 = o[2 * c2 + 1];
        y[c2].x = pack2d(fmaf(fmaf(x0[0], rstd, nmr), g0.x, h0.x), fmaf(fmaf(x0[1], rstd, nmr), g0.y, h0.y));
        y[c2].y = pack2d(fmaf(fmaf(x0[2], rstd, nmr), g0.z, h0.z), fmaf(fmaf(x0[3], rstd, nmr), g0.w, h0.w));
        y[c2].z = pack2d(fmaf(fmaf(x1[0], rstd, nmr), g1.x, h1.x), fmaf(fmaf(x1[1], rstd, nmr), g1.y, h1.y));
        y[c2].w = pack2d(fmaf(fmaf(x1[2], rstd, nmr), g1.z, h1.z), fmaf(fmaf(x1[3], rstd, nmr), g1.w, h1.w));
    }
}

// token->image block (see t2i_block) on the fragments BASE ..: [36 score fragments][64 Wv fragments, k-step major: all 8 heads]
// COMPACT: one accumulator per COLUMN tile instead of one per head - the two heads of a column tile own disjoint columns, so head
// 2 ct accumulates with the probabilities of columns 8 .. 15 zeroed and head 2 ct + 1 with columns 0 .. 7 zeroed (16 registers less)
template <int G, class Off, int BASE, int ABL = 0, bool COMPACT = false, class F>
MSAM_DEVINL void t2i_block_r(Ring<G, Off::D>& ring, const unsigned char* L, const uint4* y, const uint4* tk, AttnState& st, F&& consumed) {
    f32x4_t s[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) s[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        const int n = BASE + i, g = i / 4, ct = i % 4;
        RING_STEP(n);
        if (!(ABL & 8)) s[ct] = mfma16d(g < 8 ? y[g < 8 ? g : 0] : tk[ct], RING_AT(n), s[ct]);
        else if (i < 4) s[ct][0] += __uint_as_float((RING_AT(n).x ^ tk[ct].x ^ y[ct].x ^ y[4 + ct].y) & 0x3fffffffu);
    }
    consumed();
    float mloc[4];
    bool need = false;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        mloc[ct] = fmaxf(fmaxf(s[ct][0], s[ct][1]), fmaxf(s[ct][2], s[ct][3]));
        need = need || (mloc[ct] > st.m[ct] + 8.f);
    }
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float mx = mloc[ct];
            mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(st.m[ct], mx), alpha = __builtin_amdgcn_exp2f(st.m[ct] - mn);
            st.m[ct] = mn; st.l[ct] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (COMPACT) st.o[ct][r] *= alpha;
                else { st.o[2 * ct][r] *= alpha; st.o[2 * ct + 1][r] *= alpha; }
            }
        }
    }
    uint4 pb[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = (ABL & 64) ? s[ct][r] - st.m[ct] : __builtin_amdgcn_exp2f(s[ct][r] - st.m[ct]);
        st.l[ct] += (e[0] + e[1]) + (e[2] + e[3]);
        pb[ct] = make_uint4(pack2d(e[0], e[1]), pack2d(e[2], e[3]), 0u, 0u);
    }
    f32x4_t v[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) v[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int n = BASE + 36 + i, ks = i / 8, h = i % 8;
        RING_STEP(n);
        if (!(ABL & 16) || ks == 0) v[h] = mfma16d(y[(ABL & 16) ? h : ks], RING_AT(n), v[h]);
    }
    const bool hi_cols = (__lane_id() & 8) != 0;               // this lane's column (l & 15) belongs to the odd head of its column tile
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const uint4 va = make_uint4(pack2d(v[h][0], v[h][1]), pack2d(v[h][2], v[h][3]), 0u, 0u);
        if (COMPACT) {
            const bool mine = hi_cols == ((h & 1) != 0);
            const uint4 pm = make_uint4(mine ? pb[h >> 1].x : 0u, mine ? pb[h >> 1].y : 0u, 0u, 0u);
            st.o[h >> 1] = mfma16d(va, pm, st.o[h >> 1]);
        } else {
            st.o[h] = mfma16d(va, pb[h >> 1], st.o[h]);
        }
    }
}

// fragment streams (byte offsets in the kernels' LDS images); N = fragments per tile incl. padding reads: a multiple of 2 G
template <int G, int D_ = 2> struct ChainOff {               // i2t01: [KT0 8][VF0 32][KF1 / KT1 40][VF1 32]
    static constexpr int D = D_, USED = 112, N = ((USED + D * G - 1) / (D * G)) * (D * G);
    static __device__ constexpr int off(int n) {
        if (n < 8) return C_KT0 + ((n % 4) * 2 + n / 4) * FRAG;
        if (n < 40) return C_VF0 + (n - 8) * FRAG;
        if (n < 80) { const int i = n - 40, g = i / 4, m = i % 4; return g < 8 ? C_KF1 + (m * 8 + g) * FRAG : C_KT1 + (m * 2 + (g - 8)) * FRAG; }
        if (n < 112) return C_VF1 + (n - 80) * FRAG;
        return C_KT0;                                        // padding
    }
};
template <int G, int D_ = 2> struct FuseOff {                // i2t0_t2i: [KT0 8][VF0 32][QF / QD 36][WV 64]
    static constexpr int D = D_, USED = 140, N = ((USED + D * G - 1) / (D * G)) * (D * G);
    static __device__ constexpr int off(int n) {
        if (n < 8) return F_KT0 + ((n % 4) * 2 + n / 4) * FRAG;
        if (n < 40) return F_VF0 + (n - 8) * FRAG;
        if (n < 76) { const int i = n - 40, g = i / 4, ct = i % 4; return g < 8 ? F_QF + (ct * 8 + g) * FRAG : F_QD + ct * FRAG; }
        if (n < 140) { const int i = n - 76, ks = i / 8, h = i % 8; return F_WV + (h * 8 + ks) * FRAG; }
        return F_KT0;                                        // padding
    }
};

template <int NW, int G, int D = 2, bool LATE = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 1 : 2) void i2t01_ring_kernel(ChainArgs a) {
    typedef ChainOff<G, D> Off;
    constexpr int ABL = 0;
    constexpr int NTH = 64 * NW, NT = 256 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[C_LDS];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    float* prm = (float*)(lds + C_PRM);                      // ln0_w, ln0_b, ln1_w, ln1_b
    for (int i = tid; i < 4 * C; i += NTH) prm[i] = i < C ? a.ln0_w[i] : i < 2 * C ? a.ln0_b[i - C] : i < 3 * C ? a.ln1_w[i - 2 * C] : a.ln1_b[i - 3 * C];
    const WaveConst wc = wave_const(fr, fg, a.Nt);
    const int voff = lane * 16, tvoff = lane * 16;      // blocked tables / stream: one fragment = 1 KiB contiguous, lane-major
    const rsrc_t rsrc = make_rsrc(a.src, T * C * 2), rq0 = make_rsrc(a.q0, T * CI * 2), rtab = make_rsrc(a.tabq1, T * CI * 2);
    const unsigned char* const L = lds + lane * 16;
    const float* const gp0 = prm + fg * 8;
    const float* const gp1 = prm + 2 * C + fg * 8;
    Ring<G, D> ring;
    for (int p = (int)blockIdx.x; p < a.P; p += (int)gridDim.x) {
        __syncthreads();
        {
            const uint4* s0 = (const uint4*)(a.oper0 + (long)p * (OPER_BYTES / 2));
            const uint4* s1 = (const uint4*)(a.oper1 + (long)p * (OPER_BYTES / 2));
            for (int i = tid; i < KT_BYTES / 16; i += NTH) ((uint4*)(lds + C_KT0))[i] = s0[KT_OFF / 16 + i];
            for (int i = tid; i < VF_BYTES / 16; i += NTH) ((uint4*)(lds + C_VF0))[i] = s0[VF_OFF / 16 + i];
#pragma unroll 3
            for (int i = tid; i < OPER_BYTES / 16; i += NTH) ((uint4*)(lds + C_KT1))[i] = s1[i];
        }
        __syncthreads();
        const rsrc_t ro = make_rsrc(a.out + (long)p * T * C, T * C * 2);
        uint4 b[8], qi[4], tb[4];
#define CH_LOAD0(tile_)                                                                            \
        do {                                                                                       \
            const int so_ = (tile_) * (16 * C * 2), to_ = (tile_) * (16 * CI * 2);                 \
            _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) b[s_] = buf_load16(rsrc, voff, so_ + s_ * FRAG);    \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) qi[s_] = buf_load16(rq0, tvoff, to_ + s_ * FRAG);   \
        } while (0)
#define CH_LOAD1(tile_)                                                                            \
        do {                                                                                       \
            const int to_ = (tile_) * (16 * CI * 2);                                               \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) tb[s_] = buf_load16(rtab, tvoff, to_ + s_ * FRAG);  \
        } while (0)
        CH_LOAD0(w);
        if (!LATE) CH_LOAD1(w);
        wait_vmem_all();
#pragma unroll
        for (int k = 0; k + 1 < D; ++k) ring_fill<G, Off>(ring, L, k);
        for (int n = 0; n < NT; ++n) {
            asm volatile("" ::: "memory");
            const int tile = w + NW * n, nx = w + NW * (n + 1 < NT ? n + 1 : NT - 1);
            uint4 y1[8], y2[8];
            i2t_block_r<G, Off, 0, false>(ring, L, gp0, a.eps, b, qi, wc, y1, [&]() { if (LATE) CH_LOAD1(tile); else CH_LOAD0(nx); });
            i2t_block_r<G, Off, 40, true>(ring, L, gp1, a.eps, y1, tb, wc, y2, [&]() { if (LATE) CH_LOAD0(nx); else CH_LOAD1(nx); });
#pragma unroll
            for (int i = Off::USED; i < Off::N; ++i) RING_STEP(i);
#pragma unroll
            for (int c2 = 0; c2 < 8; ++c2) buf_store16(y2[c2], ro, voff, tile * (16 * C * 2) + c2 * FRAG);
        }
#undef CH_LOAD0
#undef CH_LOAD1
    }
}

// LATE: the tile loads are issued one phase (not one tile) ahead of their first use - the table tile of the attention during the
// layer-0 block of the SAME tile, the source / q0 tile of the next tile during the V projection - so that their registers are free
// in the phases that need them most (the loads hit L2: a phase is more than their latency).
template <int NW, int G, int ABL = 0, int D = 2, bool COMPACT = false, bool LATE = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 1 : 2) void i2t0_t2i_ring_kernel(FuseArgs a) {
    typedef FuseOff<G, D> Off;
    constexpr int NTH = 64 * NW, NT = 256 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[F_LDS];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    float* prm = (float*)(lds + F_PRM);
    for (int i = tid; i < 2 * C; i += NTH) prm[i] = i < C ? a.ln0_w[i] : a.ln0_b[i - C];
    for (int i = tid; i < WV_BYTES / 16; i += NTH) ((uint4*)(lds + F_WV))[i] = ((const uint4*)a.wvfrag)[i];
    const WaveConst wc = wave_const(fr, fg, a.Nt);
    const int voff = lane * 16, tvoff = lane * 16;      // blocked tables / stream: one fragment = 1 KiB contiguous, lane-major
    const rsrc_t rsrc = make_rsrc(a.src, T * C * 2), rq0 = make_rsrc(a.q0, T * CI * 2), rtab = make_rsrc(a.tabk, T * CI * 2);
    const unsigned char* const L = lds + lane * 16;
    const float* const gp0 = prm + fg * 8;
    Ring<G, D> ring;
    for (int p = (int)blockIdx.x; p < a.P; p += (int)gridDim.x) {
        __syncthreads();
        {
            const uint4* s0 = (const uint4*)(a.oper0 + (long)p * (OPER_BYTES / 2));
            const uint4* s1 = (const uint4*)(a.aoper + (long)p * (AOPER_BYTES / 2));
            for (int i = tid; i < KT_BYTES / 16; i += NTH) ((uint4*)(lds + F_KT0))[i] = s0[KT_OFF / 16 + i];
            for (int i = tid; i < VF_BYTES / 16; i += NTH) ((uint4*)(lds + F_VF0))[i] = s0[VF_OFF / 16 + i];
            for (int i = tid; i < AOPER_BYTES / 16; i += NTH) ((uint4*)(lds + F_QD))[i] = s1[i];
        }
        __syncthreads();
        AttnState st;
#pragma unroll
        for (int h = 0; h < 8; ++h) st.o[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { st.m[ct] = NEG_BIG; st.l[ct] = 0.f; }
        uint4 b[8], qi[4], tk[4];
#define FU_LOAD0(tile_)                                                                            \
        do {                                                                                       \
            const int so_ = (tile_) * (16 * C * 2), to_ = (tile_) * (16 * CI * 2);                 \
            _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) b[s_] = buf_load16(rsrc, voff, so_ + s_ * FRAG);    \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) qi[s_] = buf_load16(rq0, tvoff, to_ + s_ * FRAG);   \
        } while (0)
#define FU_LOAD1(tile_)                                                                            \
        do {                                                                                       \
            const int to_ = (tile_) * (16 * CI * 2);                                               \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) tk[s_] = buf_load16(rtab, tvoff, to_ + s_ * FRAG);  \
        } while (0)
        FU_LOAD0(w & a.tmask);
        if (!LATE) FU_LOAD1(w & a.tmask);
        if (ABL & 32) {
#pragma unroll
            for (int j = 0; j < G; ++j)
#pragma unroll
                for (int k = 0; k < D; ++k) ring.buf[k][j] = make_uint4(tid, lane, 0x3c003c00u, 0x3c003c00u);
        }
#pragma unroll
        for (int k = 0; k + 1 < D; ++k) ring_fill<G, Off, ABL>(ring, L, k);
        for (int n = 0; n < NT; ++n) {
            asm volatile("" ::: "memory");
            const int nx = (w + NW * (n + 1 < NT ? n + 1 : NT - 1)) & a.tmask, cur = (w + NW * n) & a.tmask;
            uint4 y1[8];
            i2t_block_r<G, Off, 0, false, ABL>(ring, L, gp0, a.eps, b, qi, wc, y1, [&]() { if (LATE) FU_LOAD1(cur); else FU_LOAD0(nx); });
            t2i_block_r<G, Off, 40, ABL, COMPACT>(ring, L, y1, tk, st, [&]() { if (LATE) FU_LOAD0(nx); else FU_LOAD1(nx); });
#pragma unroll
            for (int i = Off::USED; i < Off::N; ++i) RING_STEP(i);
        }
#undef FU_LOAD0
#undef FU_LOAD1
        // ---- merge the waves' (m, l, O^T) and write out[p][t][16h + d] = O / l + bv
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { st.l[ct] += __shfl_xor(st.l[ct], 16); st.l[ct] += __shfl_xor(st.l[ct], 32); }
        __syncthreads();
        float* mg = (float*)lds + w * MERGE_FLOATS;
        if (fg == 0) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { mg[ct * 16 + fr] = st.m[ct]; mg[64 + ct * 16 + fr] = st.l[ct]; }
        }
        if (COMPACT) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)                     // column fr of column tile ct = (head 2 ct + (fr >> 3), token fr & 7)
                *(float4*)(mg + 128 + (ct * 16 + fr) * 16 + fg * 4) = make_float4(st.o[ct][0], st.o[ct][1], st.o[ct][2], st.o[ct][3]);
        } else {
#pragma unroll
            for (int h = 0; h < 8; ++h)
                if ((fr >> 3) == (h & 1))
                    *(float4*)(mg + 128 + (h * 8 + (fr & 7)) * 16 + fg * 4) = make_float4(st.o[h][0], st.o[h][1], st.o[h][2], st.o[h][3]);
        }
        __syncthreads();
        for (int idx = tid; idx < 512; idx += NTH) {
            const int col = idx >> 3, d0 = (idx & 7) * 2, h = col >> 3, t = col & 7;
            const float* g0 = (const float*)lds;
            float mx = NEG_BIG;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) mx = fmaxf(mx, g0[ww * MERGE_FLOATS + col]);
            float lt = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const float* gw = g0 + ww * MERGE_FLOATS;
                const float sc = __builtin_amdgcn_exp2f(gw[col] - mx);
                lt = fmaf(sc, gw[64 + col], lt);
                o0 = fmaf(sc, gw[128 + col * 16 + d0], o0); o1 = fmaf(sc, gw[128 + col * 16 + d0 + 1], o1);
            }
            if (t < a.Nt) {
                const float inv = 1.f / lt;
                *(uint32_t*)(a.out + ((long)p * a.Nt + t) * CI + h * 16 + d0) =
                    pack2d(fmaf(o0, inv, a.bv[h * 16 + d0]), fmaf(o1, inv, a.bv[h * 16 + d0 + 1]));
            }
        }
    }
}

// ============================================================================================================
// Second form of the chained attention (i2t0_t2i_v2_kernel): fewer LDS operand reads and MFMAs per tile (these kernels run at
// the rate the LDS delivers fragments, profiles/r02_experiments.md).
//  * The LayerNorm affine of the layer-0 block is folded into the attention's operands: gamma into Q' and Wv, beta into one
//    constant per score column (beta . Q') and per output channel (Wv beta): the block emits x^ = (x - mean) rstd only - no
//    weight / bias reads (32 of the 172 fragment-sized LDS reads per tile), two FMAs less per value.
//  * The V projection is taken BEFORE the LayerNorm, by linearity:  x = src + P0 V''0 (out_proj bias inside V''0),
//        V = x^ (gamma Wv)^T = rstd (src (gWv)^T + P0 (V''0 (gWv)^T) - mean rowsum(gWv)) = rstd (tabV + P0 M - mean gd)
//    tabV [4096,128] and gd [128] are prompt independent (msam_chain_prepare_tables2); M [64,128] costs 16 MACs per entry per
//    prompt (M[(h,t)][d] = sum_e v0[t][16h+e] WoWv[d][16h+e] + cd[d] / 8 with WoWv = (gWv) Wo, msam_t2i_fold_values).  Per tile:
//    16 MFMAs / fragments (P0 from the layer-0 block's registers x M) instead of 64, plus ~3 VALU per value for the row / column
//    terms; the row statistics of tokens 4 (l >> 4) + r come from the lanes that own those tokens (8 cross-lane reads).
struct AttnOff2 {                                            // LDS image of the v2 kernel
    static constexpr int KT0 = 0, VF0 = KT0 + KT_BYTES, QD2 = VF0 + VF_BYTES, QF2 = QD2 + 4 * FRAG, MF2 = QF2 + 32 * FRAG,
                         COLC = MF2 + 16 * FRAG, GDT = COLC + 64 * 4, LDS = GDT + 128 * 4;
};
constexpr int AOPER2_BYTES = 4 * FRAG + 32 * FRAG;           // per prompt: [QD][QF (gamma folded)] + 64 column constants
constexpr int AOPER2_STRIDE = AOPER2_BYTES + 256;
constexpr int MF_BYTES = 16 * FRAG;                          // per prompt: M fragments [a2][dv]
// tables2 blob: WoWv fp32 [128][128], cd / gd / bwv / kb fp32 [128] each, tabV d16 blocked-C [256 tiles][64 lanes][8 dv][4 rows]
constexpr long T2_WOWV = 0, T2_CD = 128 * 128 * 4, T2_GD = T2_CD + 512, T2_BWV = T2_GD + 512, T2_KB = T2_BWV + 512,
               T2_TABV = T2_KB + 512, T2_WVG = T2_TABV + (long)T * CI * 2 /* gamma Wv d16 [128][256] */,
               T2_TABV_RM = T2_WVG + (long)CI * C * 2 /* row-major product before the relayout */, T2_BYTES = T2_TABV_RM + (long)T * CI * 2;

template <int G, int D_ = 2> struct FuseOff2 {               // [KT0 8][VF0 32][QF / QD 36][MF 16]
    static constexpr int D = D_, USED = 92, N = ((USED + D * G - 1) / (D * G)) * (D * G);
    static __device__ constexpr int off(int n) {
        if (n < 8) return AttnOff2::KT0 + ((n % 4) * 2 + n / 4) * FRAG;
        if (n < 40) return AttnOff2::VF0 + (n - 8) * FRAG;
        if (n < 76) { const int i = n - 40, g = i / 4, ct = i % 4; return g < 8 ? AttnOff2::QF2 + (ct * 8 + g) * FRAG : AttnOff2::QD2 + ct * FRAG; }
        if (n < 92) { const int i = n - 76, dv = i / 2, a2 = i % 2; return AttnOff2::MF2 + (a2 * 8 + dv) * FRAG; }
        return AttnOff2::KT0;
    }
};

struct FuseArgs2 {
    const u16* src; const u16* q0; const u16* tabk;   // blocked tables (msam_chain_prepare_tables)
    const u16* tabv; const float* gd; const float* bwv;        // tables2
    const u16* oper0;                    // layer-0 image->token operands (KT, VF)
    const u16* aoper;                    // per prompt [QD][QF gamma][64 column constants]
    const u16* mf;                       // per prompt M fragments
    float eps; int Nt, P;
    u16* out;
};

// colc / gd: this lane's column constants in LDS (float index 16 ct + (l & 15) / 16 dv + (l & 15), read where they are used)
template <int G, class Off, int BASE, class F0, class F>
MSAM_DEVINL void t2i_block_v2(Ring<G, Off::D>& ring, const unsigned char* L, const uint4* y, const uint4* tk, const uint4* tv,
                              const uint4* pk0, float rstd, float nmr, const float* colc, const float* gd, AttnState& st, F0&& start,
                              F&& consumed) {
    constexpr int ABL = 0;
    start();
    f32x4_t s[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) { const float c_ = colc[ct * 16]; s[ct] = f32x4_t{c_, c_, c_, c_}; }
#pragma unroll
    for (int i = 0; i < 36; ++i) {
        const int n = BASE + i, g = i / 4, ct = i % 4;
        RING_STEP(n);
        s[ct] = mfma16d(g < 8 ? y[g < 8 ? g : 0] : tk[ct], RING_AT(n), s[ct]);
    }
    consumed();
    float mloc[4];
    bool need = false;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        mloc[ct] = fmaxf(fmaxf(s[ct][0], s[ct][1]), fmaxf(s[ct][2], s[ct][3]));
        need = need || (mloc[ct] > st.m[ct] + 8.f);
    }
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            float mx = mloc[ct];
            mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(st.m[ct], mx), alpha = __builtin_amdgcn_exp2f(st.m[ct] - mn);
            st.m[ct] = mn; st.l[ct] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) { st.o[2 * ct][r] *= alpha; st.o[2 * ct + 1][r] *= alpha; }
        }
    }
    uint4 pb[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(s[ct][r] - st.m[ct]);
        st.l[ct] += (e[0] + e[1]) + (e[2] + e[3]);
        pb[ct] = make_uint4(pack2d(e[0], e[1]), pack2d(e[2], e[3]), 0u, 0u);
    }
    // statistics of the rows (tokens) 4 (l >> 4) + r of this lane's accumulators: owned by the lanes whose l & 15 is that token
    float rr[4], nn[4];
    const int row0 = (__lane_id() >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { rr[r] = __shfl(rstd, row0 + r); nn[r] = __shfl(nmr, row0 + r); }
    const uint32_t* tvw = (const uint32_t*)tv;               // [dv][2 words]: rows 0, 1 | rows 2, 3
#pragma unroll
    for (int dv = 0; dv < 8; ++dv) {
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
            const int n = BASE + 36 + dv * 2 + a2;
            RING_STEP(n);
            v = mfma16d(pk0[a2], RING_AT(n), v);
        }
        const uint32_t w0 = tvw[2 * dv], w1 = tvw[2 * dv + 1];
        const float t0 = d2f((u16)(w0 & 0xffff)), t1 = d2f((u16)(w0 >> 16)), t2 = d2f((u16)(w1 & 0xffff)), t3 = d2f((u16)(w1 >> 16));
        const float g = gd[dv * 16];
        const float v0 = fmaf(rr[0], v[0] + t0, nn[0] * g), v1 = fmaf(rr[1], v[1] + t1, nn[1] * g);
        const float v2 = fmaf(rr[2], v[2] + t2, nn[2] * g), v3 = fmaf(rr[3], v[3] + t3, nn[3] * g);
        const uint4 va = make_uint4(pack2d(v0, v1), pack2d(v2, v3), 0u, 0u);
        st.o[dv] = mfma16d(va, pb[dv >> 1], st.o[dv]);
    }
}

template <int NW, int G>
__global__ __launch_bounds__(64 * NW, 2) void i2t0_t2i_v2_kernel(FuseArgs2 a) {
    typedef FuseOff2<G, 2> Off;
    constexpr int ABL = 0;
    constexpr int NTH = 64 * NW, NT = 256 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[AttnOff2::LDS > 8 * MERGE_FLOATS * 4 ? AttnOff2::LDS : 8 * MERGE_FLOATS * 4];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    const WaveConst wc = wave_const(fr, fg, a.Nt);
    const int voff = lane * 16, tvoff = lane * 16, vvoff = lane * 64;
    const rsrc_t rsrc = make_rsrc(a.src, T * C * 2), rq0 = make_rsrc(a.q0, T * CI * 2), rtab = make_rsrc(a.tabk, T * CI * 2),
                 rtv = make_rsrc(a.tabv, T * CI * 2);
    const unsigned char* const L = lds + lane * 16;
    if (tid < 128) ((float*)(lds + AttnOff2::GDT))[tid] = a.gd[tid];
    const float* const colc = (const float*)(lds + AttnOff2::COLC) + fr;
    const float* const gdl = (const float*)(lds + AttnOff2::GDT) + fr;
    Ring<G, 2> ring;
    for (int p = (int)blockIdx.x; p < a.P; p += (int)gridDim.x) {
        __syncthreads();
        {
            const uint4* s0 = (const uint4*)(a.oper0 + (long)p * (OPER_BYTES / 2));
            const uint4* s1 = (const uint4*)((const unsigned char*)a.aoper + (long)p * AOPER2_STRIDE);
            const uint4* s2 = (const uint4*)((const unsigned char*)a.mf + (long)p * MF_BYTES);
            for (int i = tid; i < KT_BYTES / 16; i += NTH) ((uint4*)(lds + AttnOff2::KT0))[i] = s0[KT_OFF / 16 + i];
            for (int i = tid; i < VF_BYTES / 16; i += NTH) ((uint4*)(lds + AttnOff2::VF0))[i] = s0[VF_OFF / 16 + i];
            for (int i = tid; i < AOPER2_BYTES / 16; i += NTH) ((uint4*)(lds + AttnOff2::QD2))[i] = s1[i];     // QD, QF contiguous
            for (int i = tid; i < MF_BYTES / 16; i += NTH) ((uint4*)(lds + AttnOff2::MF2))[i] = s2[i];
            for (int i = tid; i < 16; i += NTH) ((uint4*)(lds + AttnOff2::COLC))[i] = s1[AOPER2_BYTES / 16 + i];
        }
        __syncthreads();
        AttnState st;
#pragma unroll
        for (int h = 0; h < 8; ++h) st.o[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { st.m[ct] = NEG_BIG; st.l[ct] = 0.f; }
        uint4 b[8], qi[4], tk[4], tv[4];
#define F2_LOAD0(tile_)                                                                            \
        do {                                                                                       \
            const int so_ = (tile_) * (16 * C * 2), to_ = (tile_) * (16 * CI * 2);                 \
            _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) b[s_] = buf_load16(rsrc, voff, so_ + s_ * FRAG);    \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) qi[s_] = buf_load16(rq0, tvoff, to_ + s_ * FRAG);   \
        } while (0)
#define F2_LOAD1(tile_)                                                                            \
        do {                                                                                       \
            const int to_ = (tile_) * (16 * CI * 2);                                               \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) tk[s_] = buf_load16(rtab, tvoff, to_ + s_ * FRAG);  \
        } while (0)
#define F2_LOAD2(tile_)                                                                            \
        do {                                                                                       \
            const int to_ = (tile_) * (16 * CI * 2);                                               \
            _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) tv[s_] = buf_load16(rtv, vvoff, to_ + s_ * 16);     \
        } while (0)
        F2_LOAD0(w);
        ring_fill<G, Off>(ring, L, 0);
        for (int n = 0; n < NT; ++n) {
            asm volatile("" ::: "memory");
            const int nx = w + NW * (n + 1 < NT ? n + 1 : NT - 1), cur = w + NW * n;
            uint4 y1[8], pk0[2];
            float stat[2];
            i2t_block_r<G, Off, 0, false, 0, false>(ring, L, nullptr, a.eps, b, qi, wc, y1, [&]() { F2_LOAD1(cur); }, pk0, stat);
            t2i_block_v2<G, Off, 40>(ring, L, y1, tk, tv, pk0, stat[0], stat[1], colc, gdl, st, [&]() { F2_LOAD2(cur); }, [&]() { F2_LOAD0(nx); });
#pragma unroll
            for (int i = Off::USED; i < Off::N; ++i) RING_STEP(i);
        }
#undef F2_LOAD0
#undef F2_LOAD1
#undef F2_LOAD2
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { st.l[ct] += __shfl_xor(st.l[ct], 16); st.l[ct] += __shfl_xor(st.l[ct], 32); }
        __syncthreads();
        float* mg = (float*)lds + w * MERGE_FLOATS;
        if (fg == 0) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) { mg[ct * 16 + fr] = st.m[ct]; mg[64 + ct * 16 + fr] = st.l[ct]; }
        }
#pragma unroll
        for (int h = 0; h < 8; ++h)
            if ((fr >> 3) == (h & 1))
                *(float4*)(mg + 128 + (h * 8 + (fr & 7)) * 16 + fg * 4) = make_float4(st.o[h][0], st.o[h][1], st.o[h][2], st.o[h][3]);
        __syncthreads();
        for (int idx = tid; idx < 512; idx += NTH) {
            const int col = idx >> 3, d0 = (idx & 7) * 2, h = col >> 3, t = col & 7;
            const float* g0 = (const float*)lds;
            float mx = NEG_BIG;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) mx = fmaxf(mx, g0[ww * MERGE_FLOATS + col]);
            float lt = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const float* gw = g0 + ww * MERGE_FLOATS;
                const float sc = __builtin_amdgcn_exp2f(gw[col] - mx);
                lt = fmaf(sc, gw[64 + col], lt);
                o0 = fmaf(sc, gw[128 + col * 16 + d0], o0); o1 = fmaf(sc, gw[128 + col * 16 + d0 + 1], o1);
            }
            if (t < a.Nt) {
                const float inv = 1.f / lt;
                *(uint32_t*)(a.out + ((long)p * a.Nt + t) * CI + h * 16 + d0) =
                    pack2d(fmaf(o0, inv, a.bwv[h * 16 + d0]), fmaf(o1, inv, a.bwv[h * 16 + d0 + 1]));
            }
        }
    }
}

// ---- prompt-independent tables of the v2 form: WoWv = (gamma Wv) Wo0, cd = (gamma Wv) bo0, gd = rowsum(gamma Wv), bwv = bv + Wv beta,
// kb = Wk beta; one block per d (128), thread e (128)
__global__ __launch_bounds__(128) void tables2_small_kernel(const u16* __restrict__ wv, const float* __restrict__ bv,
                                                            const u16* __restrict__ wk, const float* __restrict__ gam,
                                                            const float* __restrict__ bet, const u16* __restrict__ wo0,
                                                            const float* __restrict__ bo0, unsigned char* __restrict__ t2,
                                                            long wvg_off) {
    __shared__ float wg[C], wr[C], kr[C];
    const int d = blockIdx.x, e = threadIdx.x;
    for (int c = e; c < C; c += 128) { wr[c] = d2f(wv[d * C + c]); wg[c] = wr[c] * gam[c]; kr[c] = d2f(wk[d * C + c]); }
    __syncthreads();
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(wg[c], d2f(wo0[c * CI + e]), acc);
    ((float*)(t2 + T2_WOWV))[d * 128 + e] = acc;
    for (int c = e; c < C; c += 128) ((u16*)(t2 + wvg_off))[d * C + c] = f2d(wg[c]);
    if (e < 4) {
        float v = 0.f;
        for (int c = 0; c < C; ++c) v += e == 0 ? wg[c] * bo0[c] : e == 1 ? wg[c] : e == 2 ? wr[c] * bet[c] : kr[c] * bet[c];
        if (e == 2) v += bv[d];
        ((float*)(t2 + (e == 0 ? T2_CD : e == 1 ? T2_GD : e == 2 ? T2_BWV : T2_KB)))[d] = v;
    }
}
// tabV = src (gamma Wv)^T comes from the MFMA GEMM as d16 row-major [4096][128]; this puts it into the blocked-C order the kernel
// loads per tile: [tile][lane = 16 (row >> 2) + (d & 15)][dv = d >> 4][row & 3]; one 8-byte chunk (4 rows of one column) per thread
__global__ __launch_bounds__(256) void tabv_relayout_kernel(const u16* __restrict__ rm, u16* __restrict__ tabv) {
    const int id = blockIdx.x * 256 + threadIdx.x;           // chunk = (tile, lane, dv)
    const int dv = id & 7, lane = (id >> 3) & 63, tile = id >> 9;
    const int row0 = tile * 16 + (lane >> 4) * 4, d = dv * 16 + (lane & 15);
    uint2 pk;
    pk.x = (uint32_t)rm[(long)row0 * CI + d] | ((uint32_t)rm[(long)(row0 + 1) * CI + d] << 16);
    pk.y = (uint32_t)rm[(long)(row0 + 2) * CI + d] | ((uint32_t)rm[(long)(row0 + 3) * CI + d] << 16);
    ((uint2*)tabv)[id] = pk;
}

// per prompt: M fragments [a2][dv]: B[k = (head 4 a2 + fg, token i)][col d = 16 dv + fr] = sum_e v0[t][16h+e] WoWv[d][16h+e] + cd[d] / 8
__global__ __launch_bounds__(256) void fold_values_kernel(const u16* __restrict__ vtok0, int Nt, const unsigned char* __restrict__ t2,
                                                          u16* __restrict__ mf) {
    __shared__ float vv[8][CI];
    const int p = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 8 * CI; i += 256) {
        const int t = i >> 7, dd = i & (CI - 1);
        vv[t][dd] = t < Nt ? d2f(vtok0[((long)p * Nt + t) * CI + dd]) : 0.f;
    }
    __syncthreads();
    const int d = tid & 127, hh = tid >> 7;
    const float* wowv = (const float*)(t2 + T2_WOWV) + d * 128;
    const float cd8 = ((const float*)(t2 + T2_CD))[d] * 0.125f;
    u16* dst = mf + (long)p * (MF_BYTES / 2);
    for (int h = hh * 4; h < hh * 4 + 4; ++h) {
        float wv_[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) wv_[e] = wowv[h * 16 + e];
        uint32_t pk[4];
#pragma unroll
        for (int t2_ = 0; t2_ < 4; ++t2_) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { a0 = fmaf(wv_[e], vv[2 * t2_][h * 16 + e], a0); a1 = fmaf(wv_[e], vv[2 * t2_ + 1][h * 16 + e], a1); }
            pk[t2_] = pack2d(2 * t2_ < Nt ? a0 + cd8 : 0.f, 2 * t2_ + 1 < Nt ? a1 + cd8 : 0.f);
        }
        const int a2 = h >> 2, fg = h & 3, dv = d >> 4, fr = d & 15;
        *(uint4*)(dst + (((a2 * 8 + dv) * 64 + fg * 16 + fr) * 8)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}

// per prompt: [QD][QF with gamma folded][64 column constants SCALE q . kb]  (see fold_attnfrag_kernel)
__global__ __launch_bounds__(256) void fold_attnfrag2_kernel(const u16* __restrict__ qtok, const u16* __restrict__ wk,
                                                             const float* __restrict__ gam, const unsigned char* __restrict__ t2,
                                                             int Nt, unsigned char* __restrict__ aoper) {
    __shared__ __attribute__((aligned(16))) u16 img[AOPER2_BYTES / 2];
    __shared__ float qq[8][CI];
    __shared__ float colc[64];
    const int p = blockIdx.x, c = threadIdx.x;
    for (int i = c; i < 8 * CI; i += 256) {
        const int t = i >> 7, d = i & (CI - 1);
        qq[t][d] = t < Nt ? d2f(qtok[((long)p * Nt + t) * CI + d]) : 0.f;
    }
    for (int i = c; i < 4 * FRAG / 16; i += 256) ((uint4*)img)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    {
        const int ks = c >> 5, fg = (c >> 3) & 3, ii = c & 7;
        const float gs = gam[c] * SCALE;
        for (int h = 0; h < 8; ++h) {
            float w[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) w[d] = d2f(wk[(h * 16 + d) * C + c]);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < 16; ++d) acc = fmaf(qq[t][h * 16 + d], w[d], acc);
                const int ct = h >> 1, fr = (h & 1) * 8 + t;
                img[4 * FRAG / 2 + ((ct * 8 + ks) * 64 + fg * 16 + fr) * 8 + ii] = f2d(acc * gs);
            }
        }
    }
    for (int j = c; j < 8 * 8 * 16; j += 256) {
        const int h = j >> 7, t = (j >> 4) & 7, d = j & 15;
        const int ct = h >> 1, fr = (h & 1) * 8 + t, fg = 2 * (h & 1) + (d >> 3);
        img[(ct * 64 + fg * 16 + fr) * 8 + (d & 7)] = f2d(qq[t][h * 16 + d] * SCALE);
    }
    if (c < 64) {
        const int h = c >> 3, t = c & 7;
        const float* kb = (const float*)(t2 + T2_KB);
        float v = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) v = fmaf(qq[t][h * 16 + d], kb[h * 16 + d], v);
        colc[c] = v * SCALE;
    }
    __syncthreads();
    uint4* dst = (uint4*)(aoper + (long)p * AOPER2_STRIDE);
    for (int i = c; i < AOPER2_BYTES / 16; i += 256) dst[i] = ((const uint4*)img)[i];
    if (c < 16) dst[AOPER2_BYTES / 16 + c] = ((const uint4*)colc)[c];
}

// token -> image operands of one prompt in fragment order (B operands: lane = fg * 16 + fr holds B[k = 8 fg + i][col fr]):
//   QD [ct]     : col fr = (head 2 ct + (fr >> 3), token fr & 7), k = table channel 32 ct + 8 fg + i: SCALE q[t][ch] on the head's
//                 own 16 channels, else 0
//   QF [ct][ks] : same columns, k = channel 32 ks + 8 fg + i: SCALE Q'_{h,t}[c],  Q'_{h,t} = Wk_h^T q_{h,t}
__global__ __launch_bounds__(256) void fold_attnfrag_kernel(const u16* __restrict__ qtok, const u16* __restrict__ wk, int Nt,
                                                            u16* __restrict__ aoper) {
    __shared__ __attribute__((aligned(16))) u16 img[AOPER_BYTES / 2];
    __shared__ float qq[8][CI];
    const int p = blockIdx.x, c = threadIdx.x;
    for (int i = c; i < 8 * CI; i += 256) {
        const int t = i >> 7, d = i & (CI - 1);
        qq[t][d] = t < Nt ? d2f(qtok[((long)p * Nt + t) * CI + d]) : 0.f;
    }
    for (int i = c; i < QD_BYTES / 16; i += 256) ((uint4*)img)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    {
        const int ks = c >> 5, fg = (c >> 3) & 3, ii = c & 7;
        for (int h = 0; h < 8; ++h) {
            float w[16];
#pragma unroll
            for (int d = 0; d < 16; ++d) w[d] = d2f(wk[(h * 16 + d) * C + c]);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < 16; ++d) acc = fmaf(qq[t][h * 16 + d], w[d], acc);
                const int ct = h >> 1, fr = (h & 1) * 8 + t;
                img[QD_BYTES / 2 + ((ct * 8 + ks) * 64 + fg * 16 + fr) * 8 + ii] = f2d(acc * SCALE);
            }
        }
    }
    for (int j = c; j < 8 * 8 * 16; j += 256) {
        const int h = j >> 7, t = (j >> 4) & 7, d = j & 15;
        const int ct = h >> 1, fr = (h & 1) * 8 + t, fg = 2 * (h & 1) + (d >> 3);
        img[(ct * 64 + fg * 16 + fr) * 8 + (d & 7)] = f2d(qq[t][h * 16 + d] * SCALE);
    }
    __syncthreads();
    uint4* dst = (uint4*)(aoper + (long)p * (AOPER_BYTES / 2));
    for (int i = c; i < AOPER_BYTES / 16; i += 256) dst[i] = ((const uint4*)img)[i];
}

// row-major [4096][W] -> blocked [tile][s][lane][8] (W = 256 or 128); one 16-byte chunk per thread
__global__ __launch_bounds__(256) void to_blocked_kernel(const u16* __restrict__ src, const u16* __restrict__ q0,
                                                         const u16* __restrict__ tabk, const u16* __restrict__ tabq,
                                                         u16* __restrict__ out) {
    const int id = blockIdx.x * 256 + threadIdx.x;           // chunk ids: src 4096 * 32, then three tables of 4096 * 16
    const u16* in; int W, q; u16* o;
    if (id < T * 32) { in = src; W = C; q = id; o = out; }
    else {
        const int j = id - T * 32, tb = j / (T * 16);
        in = tb == 0 ? q0 : tb == 1 ? tabk : tabq; W = CI; q = j - tb * (T * 16); o = out + (long)T * C + (long)tb * T * CI;
    }
    // destination chunk q of the blocked image: tile, s, lane (fg, fr)
    const int per_tile = 16 * W / 8, tile = q / per_tile, r = q % per_tile, sidx = r >> 6, lane = r & 63, fg = lane >> 4, fr = lane & 15;
    ((uint4*)o)[q] = *(const uint4*)(in + (long)(tile * 16 + fr) * W + sidx * 32 + fg * 8);
}

// Wv [128][256] in fragment order: [dv][ks][lane][8]: col fr = row 16 dv + fr of Wv, k = channel 32 ks + 8 fg + i
__global__ __launch_bounds__(256) void wv_frag_kernel(const u16* __restrict__ wv, u16* __restrict__ wvfrag) {
    const int id = blockIdx.x * 256 + threadIdx.x;           // one 16-byte chunk each: 128 rows x 32 chunks
    const int row = id >> 5, ch = id & 31, dv = row >> 4, fr = row & 15, ks = ch >> 2, fg = ch & 3;
    ((uint4*)wvfrag)[(dv * 8 + ks) * 64 + fg * 16 + fr] = *(const uint4*)(wv + (long)row * C + ch * 8);
}

}  // namespace

int64_t msam_i2t_tok_workspace_bytes(int32_t P) { return (int64_t)P * OPER_BYTES; }

static int cu_count() {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
}

// same contract as msam_i2t_fold_layer (decfold.hip), which dispatches here
int msam_i2t_tok_layer(const void* xin, int32_t x_shared, const void* ktok, const void* vtok, int32_t P, int32_t Nt,
                       const void* wq, const void* tabq, const void* wo, const float* bo, const float* ln_w,
                       const float* ln_b, float ln_eps, void* out, void* workspace, int KS, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    u16* oper = (u16*)workspace;
    hipLaunchKernelGGL(fold_frag_kernel, dim3(P), dim3(256), 0, s, (const u16*)ktok, (const u16*)vtok, (const u16*)wq,
                       (const u16*)wo, bo, Nt, 1, oper);
    if (int e = msam_check_launch("fold_frag")) return e;
    TokArgs a{};
    a.xin = (const u16*)xin; a.x_shared = x_shared; a.oper = oper; a.tabq = (const u16*)tabq;
    a.ln_w = ln_w; a.ln_b = ln_b; a.eps = ln_eps; a.Nt = Nt; a.KS = KS; a.nitems = P * KS; a.out = (u16*)out;
    const int wgs = g_tune_i2t_wg_per_cu * cu_count();
    const int grid = a.nitems < wgs ? a.nitems : wgs;
    const double flops = (double)P * T * (2.0 * 64 * C * 2 + 2.0 * 64 * 32);
    const double bytes = (double)(x_shared ? 1 : P) * T * C * 2 + (double)P * T * C * 2;
    msam_profile_mark2(stream, 1, flops, bytes, 2);
    hipLaunchKernelGGL(i2t_tok_kernel, dim3(grid), dim3(NTHR), 0, s, a);
    msam_profile_mark2(stream, 0, flops, bytes, 2);
    return msam_check_launch("i2t_tok");
}

// ---- chained forms on a shared source (include/msam_hip.h: msam_i2t_fold_operands, msam_i2t0_t2i_fused, msam_i2t01_fused)
extern "C" int64_t msam_i2t_fold_operand_bytes(int32_t P) { return (int64_t)P * OPER_BYTES; }

extern "C" int msam_i2t_fold_operands(const void* ktok, const void* vtok, int32_t P, int32_t Nt, const void* wq, const void* wo,
                                      const float* bo, int32_t with_kfold, void* operands, void* stream) {
    if (!ktok || !vtok || !wq || !wo || !bo || !operands || P <= 0) { msam_set_error("msam_i2t_fold_operands: null argument"); return 1; }
    if (Nt < 1 || Nt > 8) { msam_set_error("msam_i2t_fold_operands: 1 <= Nt <= 8 tokens per prompt"); return 1; }
    hipLaunchKernelGGL(fold_frag_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, (const u16*)ktok, (const u16*)vtok,
                       (const u16*)wq, (const u16*)wo, bo, Nt, with_kfold, (u16*)operands);
    return msam_check_launch("fold_frag");
}

extern "C" int64_t msam_chain_tables_bytes(void) { return (int64_t)T * C * 2 + 3L * T * CI * 2; }

extern "C" int msam_chain_prepare_tables(const void* src, const void* q0, const void* tabk, const void* tabq1, void* tables,
                                         void* stream) {
    if (!src || !q0 || !tabk || !tabq1 || !tables) { msam_set_error("msam_chain_prepare_tables: null argument"); return 1; }
    hipLaunchKernelGGL(to_blocked_kernel, dim3((T * 32 + 3 * T * 16) / 256), dim3(256), 0, (hipStream_t)stream, (const u16*)src,
                       (const u16*)q0, (const u16*)tabk, (const u16*)tabq1, (u16*)tables);
    return msam_check_launch("to_blocked");
}

// layer-0 operands AND the M fragments of the second attention form in one launch (tables2 from msam_chain_prepare_tables2)
extern "C" int msam_i2t_fold_operands_values(const void* ktok, const void* vtok, int32_t P, int32_t Nt, const void* wq, const void* wo,
                                             const float* bo, int32_t with_kfold, const void* tables2, void* operands, void* mf,
                                             void* stream) {
    if (!ktok || !vtok || !wq || !wo || !bo || !operands || !tables2 || !mf || P <= 0) { msam_set_error("msam_i2t_fold_operands_values: null argument"); return 1; }
    if (Nt < 1 || Nt > 8) { msam_set_error("msam_i2t_fold_operands_values: 1 <= Nt <= 8 tokens per prompt"); return 1; }
    hipLaunchKernelGGL(fold_frag_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, (const u16*)ktok, (const u16*)vtok,
                       (const u16*)wq, (const u16*)wo, bo, Nt, with_kfold, (u16*)operands, (const unsigned char*)tables2, (u16*)mf);
    return msam_check_launch("fold_frag+values");
}

extern "C" int64_t msam_i2t0_t2i_workspace_bytes(int32_t P) { return (int64_t)P * AOPER_BYTES + WV_BYTES; }

extern "C" int msam_i2t0_t2i_fused(const void* tables, const void* operands0, const float* ln0_w, const float* ln0_b,
                                   float ln_eps, const void* qtok, int32_t P, int32_t Nt, const void* wk,
                                   const void* wv, const float* bv, void* out, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    const u16* src = (const u16*)tables;
    const u16* q0 = src ? src + (long)T * C : nullptr;
    const u16* tabk = src ? q0 + (long)T * CI : nullptr;
    if (!src || !operands0 || !ln0_w || !ln0_b || !qtok || !wk || !wv || !bv || !out || !workspace || P <= 0) {
        msam_set_error("msam_i2t0_t2i_fused: null argument");
        return 1;
    }
    if (Nt < 1 || Nt > 8) { msam_set_error("msam_i2t0_t2i_fused: 1 <= Nt <= 8 tokens per prompt"); return 1; }
    if (workspace_bytes < msam_i2t0_t2i_workspace_bytes(P)) { msam_set_error("msam_i2t0_t2i_fused: workspace too small"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    u16* wvfrag = (u16*)workspace;
    u16* aoper = wvfrag + WV_BYTES / 2;
    hipLaunchKernelGGL(wv_frag_kernel, dim3(128 * 32 / 256), dim3(256), 0, s, (const u16*)wv, wvfrag);
    if (int e = msam_check_launch("wv_frag")) return e;
    hipLaunchKernelGGL(fold_attnfrag_kernel, dim3(P), dim3(256), 0, s, (const u16*)qtok, (const u16*)wk, Nt, aoper);
    if (int e = msam_check_launch("fold_attnfrag")) return e;
    FuseArgs a{};
    a.src = (const u16*)src; a.q0 = (const u16*)q0; a.oper0 = (const u16*)operands0; a.ln0_w = ln0_w; a.ln0_b = ln0_b; a.eps = ln_eps;
    a.aoper = aoper; a.wvfrag = wvfrag; a.tabk = (const u16*)tabk; a.bv = bv; a.Nt = Nt; a.P = P; a.out = (u16*)out; a.tmask = g_tune_chain_tmask;
    const int cus = cu_count(), grid = P < cus ? P : cus;
    // executed MFMA work: 164 16x16x32 MFMAs per 16-token tile (8 + 16 + 32 layer-0 block, 36 + 64 + 8 attention)
    const double flops = (double)P * (T / 16) * 164.0 * 16384.0;
    msam_profile_mark2(stream, 1, flops, 0.0, 6);
    switch (g_tune_chain_variant) {
        case 1: hipLaunchKernelGGL((i2t0_t2i_ring_kernel<4, 8>), dim3(grid), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((i2t0_t2i_ring_kernel<4, 16>), dim3(grid), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((i2t0_t2i_ring_kernel<8, 4>), dim3(grid), dim3(512), 0, s, a); break;
        case 4: hipLaunchKernelGGL((i2t0_t2i_ring_kernel<8, 4, 0, 3, true>), dim3(grid), dim3(512), 0, s, a); break;
        case 5: hipLaunchKernelGGL((i2t0_t2i_ring_kernel<8, 4, 0, 2, true>), dim3(grid), dim3(512), 0, s, a); break;
        case 6: case 9: hipLaunchKernelGGL((i2t0_t2i_ring_kernel<8, 4, 0, 2, false, true>), dim3(grid), dim3(512), 0, s, a); break;
        case 7: hipLaunchKernelGGL((i2t0_t2i_ring_kernel<8, 4, 0, 3, false, true>), dim3(grid), dim3(512), 0, s, a); break;
        case 8: hipLaunchKernelGGL((i2t0_t2i_ring_kernel<8, 4, 0, 3, true, true>), dim3(grid), dim3(512), 0, s, a); break;
#define ABL_CASE(m_) case 100 + m_: hipLaunchKernelGGL((i2t0_t2i_ring_kernel<8, 4, m_>), dim3(grid), dim3(512), 0, s, a); break;
        ABL_CASE(1) ABL_CASE(2) ABL_CASE(4) ABL_CASE(8) ABL_CASE(16) ABL_CASE(32) ABL_CASE(64) ABL_CASE(27) ABL_CASE(31) ABL_CASE(127)
#undef ABL_CASE
        default: hipLaunchKernelGGL(i2t0_t2i_kernel, dim3(grid), dim3(NTHR8), 0, s, a);
    }
    msam_profile_mark2(stream, 0, flops, 0.0, 6);
    return msam_check_launch("i2t0_t2i");
}

// ---- second form (v2): tables2 once per decode, M fragments when the layer-0 value tokens exist, then the attention
extern "C" int64_t msam_chain_tables2_bytes(void) { return T2_BYTES; }
extern "C" int msam_chain_prepare_tables2(const void* src, const void* wv, const float* bv, const void* wk, const float* ln0_w,
                                          const float* ln0_b, const void* wo0, const float* bo0, void* tables2, void* stream) {
    if (!src || !wv || !bv || !wk || !ln0_w || !ln0_b || !wo0 || !bo0 || !tables2) { msam_set_error("msam_chain_prepare_tables2: null argument"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tables2_small_kernel, dim3(128), dim3(128), 0, s, (const u16*)wv, bv, (const u16*)wk, ln0_w, ln0_b,
                       (const u16*)wo0, bo0, (unsigned char*)tables2, T2_WVG);
    if (int e = msam_check_launch("tables2_small")) return e;
    msam_gemm_t g{};
    g.A = src; g.lda = C; g.W = (unsigned char*)tables2 + T2_WVG; g.ldw = C; g.M = T; g.N = CI; g.K = C;
    g.out = (unsigned char*)tables2 + T2_TABV_RM; g.out_dtype = MSAM_D16; g.ldc = CI; g.a_dtype = MSAM_D16 == MSAM_F16 ? MSAM_F16 : 0;
    if (int e = msam_gemm_bf16(&g, stream)) return e;
    hipLaunchKernelGGL(tabv_relayout_kernel, dim3(T / 16 * 64 * 8 / 256), dim3(256), 0, s, (const u16*)((unsigned char*)tables2 + T2_TABV_RM),
                       (u16*)((unsigned char*)tables2 + T2_TABV));
    return msam_check_launch("tabv_relayout");
}
// The weight-only part of tables2 (WoWv, cd, gd, bwv, kb and gamma Wv: everything but tabV) once per MODEL: const2 = the first
// T2_TABV bytes of the tables2 layout followed by gamma Wv; msam_chain_prepare_tables2_c then costs one 66 KiB device copy, the
// tabV GEMM (src x gamma Wv from const2) and its relayout per decode instead of the 128-block weight kernel.
extern "C" int64_t msam_chain_const2_bytes(void) { return T2_TABV + (long)CI * C * 2; }
extern "C" int msam_chain_prepare_const2(const void* wv, const float* bv, const void* wk, const float* ln0_w, const float* ln0_b,
                                         const void* wo0, const float* bo0, void* const2, void* stream) {
    if (!wv || !bv || !wk || !ln0_w || !ln0_b || !wo0 || !bo0 || !const2) { msam_set_error("msam_chain_prepare_const2: null argument"); return 1; }
    hipLaunchKernelGGL(tables2_small_kernel, dim3(128), dim3(128), 0, (hipStream_t)stream, (const u16*)wv, bv, (const u16*)wk, ln0_w,
                       ln0_b, (const u16*)wo0, bo0, (unsigned char*)const2, T2_TABV);
    return msam_check_launch("tables2_small(const)");
}
extern "C" int msam_chain_prepare_tables2_c(const void* src, const void* const2, void* tables2, void* stream) {
    if (!src || !const2 || !tables2) { msam_set_error("msam_chain_prepare_tables2_c: null argument"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(tables2, const2, (size_t)T2_TABV, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        msam_set_error("msam_chain_prepare_tables2_c: device copy failed");
        return 2;
    }
    msam_gemm_t g{};
    g.A = src; g.lda = C; g.W = (const unsigned char*)const2 + T2_TABV; g.ldw = C; g.M = T; g.N = CI; g.K = C;
    g.out = (unsigned char*)tables2 + T2_TABV_RM; g.out_dtype = MSAM_D16; g.ldc = CI; g.a_dtype = MSAM_D16 == MSAM_F16 ? MSAM_F16 : 0;
    if (int e = msam_gemm_bf16(&g, stream)) return e;
    hipLaunchKernelGGL(tabv_relayout_kernel, dim3(T / 16 * 64 * 8 / 256), dim3(256), 0, s, (const u16*)((unsigned char*)tables2 + T2_TABV_RM),
                       (u16*)((unsigned char*)tables2 + T2_TABV));
    return msam_check_launch("tabv_relayout");
}
extern "C" int64_t msam_t2i_fold_values_bytes(int32_t P) { return (int64_t)P * MF_BYTES; }
extern "C" int msam_t2i_fold_values(const void* vtok0, int32_t P, int32_t Nt, const void* tables2, void* mf, void* stream) {
    if (!vtok0 || !tables2 || !mf || P <= 0 || Nt < 1 || Nt > 8) { msam_set_error("msam_t2i_fold_values: bad arguments"); return 1; }
    hipLaunchKernelGGL(fold_values_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, (const u16*)vtok0, Nt, (const unsigned char*)tables2,
                       (u16*)mf);
    return msam_check_launch("fold_values");
}
extern "C" int64_t msam_i2t0_t2i_v2_workspace_bytes(int32_t P) { return (int64_t)P * AOPER2_STRIDE; }
extern "C" int msam_i2t0_t2i_fused_v2(const void* tables, const void* tables2, const void* operands0, const void* mf, const float* ln0_w,
                                      float ln_eps, const void* qtok, int32_t P, int32_t Nt, const void* wk, void* out,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
    if (!tables || !tables2 || !operands0 || !mf || !ln0_w || !qtok || !wk || !out || !workspace || P <= 0) {
        msam_set_error("msam_i2t0_t2i_fused_v2: null argument");
        return 1;
    }
    if (Nt < 1 || Nt > 8) { msam_set_error("msam_i2t0_t2i_fused_v2: 1 <= Nt <= 8 tokens per prompt"); return 1; }
    if (workspace_bytes < msam_i2t0_t2i_v2_workspace_bytes(P)) { msam_set_error("msam_i2t0_t2i_fused_v2: workspace too small"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    const unsigned char* t2 = (const unsigned char*)tables2;
    hipLaunchKernelGGL(fold_attnfrag2_kernel, dim3(P), dim3(256), 0, s, (const u16*)qtok, (const u16*)wk, ln0_w, t2, Nt,
                       (unsigned char*)workspace);
    if (int e = msam_check_launch("fold_attnfrag2")) return e;
    const u16* src = (const u16*)tables;
    FuseArgs2 a{};
    a.src = src; a.q0 = src + (long)T * C; a.tabk = a.q0 + (long)T * CI;
    a.tabv = (const u16*)(t2 + T2_TABV); a.gd = (const float*)(t2 + T2_GD); a.bwv = (const float*)(t2 + T2_BWV);
    a.oper0 = (const u16*)operands0; a.aoper = (const u16*)workspace; a.mf = (const u16*)mf; a.eps = ln_eps; a.Nt = Nt; a.P = P;
    a.out = (u16*)out;
    const int cus = cu_count(), grid = P < cus ? P : cus;
    const double flops = (double)P * (T / 16) * 116.0 * 16384.0;          // 8 + 16 + 32 layer-0 block, 36 + 16 + 8 attention
    msam_profile_mark2(stream, 1, flops, 0.0, 6);
    hipLaunchKernelGGL((i2t0_t2i_v2_kernel<8, 4>), dim3(grid), dim3(512), 0, s, a);
    msam_profile_mark2(stream, 0, flops, 0.0, 6);
    return msam_check_launch("i2t0_t2i_v2");
}

extern "C" int msam_i2t01_fused(const void* tables, const void* operands0, const float* ln0_w, const float* ln0_b,
                                const void* operands1, const float* ln1_w, const float* ln1_b, float ln_eps,
                                int32_t P, int32_t Nt, void* out, void* stream) {
    const u16* src = (const u16*)tables;
    const u16* q0 = src ? src + (long)T * C : nullptr;
    const u16* tabq1 = src ? q0 + 2L * T * CI : nullptr;
    if (!src || !operands0 || !ln0_w || !ln0_b || !operands1 || !ln1_w || !ln1_b || !out || P <= 0) {
        msam_set_error("msam_i2t01_fused: null argument");
        return 1;
    }
    if (Nt < 1 || Nt > 8) { msam_set_error("msam_i2t01_fused: 1 <= Nt <= 8 tokens per prompt"); return 1; }
    ChainArgs a{};
    a.src = (const u16*)src; a.q0 = (const u16*)q0; a.oper0 = (const u16*)operands0; a.ln0_w = ln0_w; a.ln0_b = ln0_b;
    a.oper1 = (const u16*)operands1; a.tabq1 = (const u16*)tabq1; a.ln1_w = ln1_w; a.ln1_b = ln1_b; a.eps = ln_eps;
    a.Nt = Nt; a.P = P; a.out = (u16*)out;
    const int cus = cu_count(), grid = P < cus ? P : cus;
    const double flops = (double)P * (T / 16) * 144.0 * 16384.0;        // executed: 56 (layer-0 block) + 88 (layer-1 block) MFMAs per tile
    const double bytes = (double)P * T * C * 2;                        // the layer-1 stream, written once
    msam_profile_mark2(stream, 1, flops, bytes, 7);
    switch (g_tune_chain_variant) {
        case 1: hipLaunchKernelGGL((i2t01_ring_kernel<4, 8>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL((i2t01_ring_kernel<4, 16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a); break;
        case 3: case 5: hipLaunchKernelGGL((i2t01_ring_kernel<8, 4>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a); break;
        case 4: hipLaunchKernelGGL((i2t01_ring_kernel<8, 4, 3>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a); break;
        case 6: case 9: hipLaunchKernelGGL((i2t01_ring_kernel<8, 4, 2, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a); break;
        case 7: case 8: hipLaunchKernelGGL((i2t01_ring_kernel<8, 4, 3, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL(i2t01_kernel, dim3(grid), dim3(NTHR8), 0, (hipStream_t)stream, a);
    }
    msam_profile_mark2(stream, 0, flops, bytes, 7);
    return msam_check_launch("i2t01");
}
