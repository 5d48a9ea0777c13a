// Fused image-side half of a two-way-transformer block for the per-prompt token stream:
//
//     q    = (keys + pos) Wq^T + bq                  (layer 1; layer 0 uses the prompt-independent precomputed q)
//     attn = softmax_j(q . k_tok[j] / 4) v_tok       (image token attends over the <= 16 prompt tokens, 8 heads x 16)
//     keys = LayerNorm(keys + attn Wo^T + bo)        (norm4)
//
// One launch replaces q-projection GEMM + image->token attention kernel + out-projection/LayerNorm GEMM and their
// HBM round trips (q 1 MiB + attn 2 MiB + residual re-read 2 MiB per prompt): the stream is read once (16 KB per
// 32-token tile) and written once.  Both weight matrices stay in registers for the whole launch (weights-stationary,
// see wsgemm.hip): Wq 16 columns x 256 per wave, Wo 32 columns x 128 per wave = 64 VGPRs.
// Per tile: keys tile -> LDS (double buffered, requested two tiles ahead) -> MFMA q-proj -> q tile (bf16, LDS) ->
// per (16 tokens, head) transposed-score MFMA attention (attention.hip form) -> attn tile (bf16, LDS) -> MFMA
// out-proj -> fp32 tile (LDS) -> row-complete epilogue (bias + residual from the LDS keys tile + LayerNorm) -> store.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);
void msam_profile_mark(void* stream, int begin, double flops);

namespace {

constexpr int T = 4096, C = 256, CI = 128, WM = 32, NTHR = 512;
constexpr float NEG_BIG = -1.0e30f;

struct LayerArgs {
    const u16* xin;          // stream [rows, 256] (layer 1) or src [4096, 256] (layer 0, row % 4096)
    const u16* q_shared;     // layer 0: bf16 [4096, 128]
    const u16* wq; const float* bq; const float* peq;     // layer 1
    const u16* wo; const float* bo; const float* ln_w; const float* ln_b; float eps;
    const u16* ktok; const u16* vtok; int Nt;
    u16* out; int rows;
};

MSAM_DEVINL int swzr(int row) { return row & 15; }

template <bool L0>
__global__ __launch_bounds__(NTHR, L0 ? 4 : 3) void dec_image_layer_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 dyn_lds[];
    constexpr int CPR = C / 8;                                   // 32 chunks per keys row
    uint4* ldsA = dyn_lds;                                       // [2][WM*CPR]            32 KB (unused for L0)
    uint4* ldsQ = dyn_lds + (L0 ? 0 : 2 * WM * CPR);             // [WM][16]  bf16 q tile   8 KB
    uint4* ldsP = ldsQ + WM * 16;                                // [WM][16]  bf16 attn     8 KB
    float* ldsC = (float*)(ldsP + WM * 16);                      // [WM][256] fp32         32 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    // ---- stationary weights
    uint4 wqf[8];                                                // Wq rows wave*16 + fr, 8 k-chunks (layer 1)
    if constexpr (!L0) {
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) wqf[kc] = *(const uint4*)(a.wq + (long)(wave * 16 + fr) * C + kc * 32 + fg * 8);
    }
    uint4 wof[2][4];                                             // Wo rows wave*32 + ni*16 + fr, 4 k-chunks
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) wof[ni][kc] = *(const uint4*)(a.wo + (long)(wave * 32 + ni * 16 + fr) * CI + kc * 32 + fg * 8);

    const int ntiles = a.rows / WM;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    uint4 ra0, ra1;                  // next tile's rows (two workgroups per CU overlap the remaining latency)
    (void)ra0; (void)ra1;
#define DL_SRC(p_, tile_) (a.xin + ((long)(tile_) * WM + ((p_) * NTHR + tid) / CPR) * C + (((p_) * NTHR + tid) % CPR) * 8)
#define DL_DST(p_, buf_) ldsA[(buf_) * WM * CPR + (((p_) * NTHR + tid) / CPR) * CPR + \
                              ((((p_) * NTHR + tid) % CPR) ^ swzr(((p_) * NTHR + tid) / CPR))]
#define DL_LOAD(r0_, r1_, tile_) do { r0_ = *(const uint4*)DL_SRC(0, tile_); r1_ = *(const uint4*)DL_SRC(1, tile_); } while (0)
#define DL_STORE(r0_, r1_, buf_) do { DL_DST(0, buf_) = r0_; DL_DST(1, buf_) = r1_; } while (0)
    if constexpr (!L0) {
        DL_LOAD(ra0, ra1, tile);
        DL_STORE(ra0, ra1, 0);
    }
    __syncthreads();
    int buf = 0;

    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const long row0 = (long)tile * WM;
        const int p = (int)(row0 / T), t0 = (int)(row0 - (long)p * T);
        if constexpr (!L0) {
            if (next < ntiles) DL_LOAD(ra0, ra1, next);
            // ---- q projection: this wave's 16 columns for the 32 rows
            const uint4* la = ldsA + buf * WM * CPR;
            f32x4_t qa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int row = mi * 16 + fr;
                    qa[mi] = mfma16(la[row * CPR + ((kc * 4 + fg) ^ swzr(row))], wqf[kc], qa[mi]);
                }
            }
            const int qc = wave * 16 + fr;                        // q column of this lane
            const float qb = a.bq[qc];
            u16* q16 = (u16*)ldsQ;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mi * 16 + fg * 4 + r;
                    const float v = qa[mi][r] + qb + a.peq[(long)(t0 + row) * CI + qc];
                    q16[row * 128 + (((qc >> 3) ^ swzr(row)) << 3) + (qc & 7)] = f2bf(v);
                }
        } else {
            // layer 0: the prompt-independent q rows of this token range
            if (tid < WM * 16) {
                const int row = tid >> 4, c = tid & 15;
                ldsQ[row * 16 + (c ^ swzr(row))] = *(const uint4*)(a.q_shared + (long)(t0 + row) * CI + c * 8);
            }
        }
        __syncthreads();
        // ---- image -> token attention: wave = head, two 16-token tiles
        {
            const int head = wave;
            uint4 ka = make_uint4(0, 0, 0, 0);
            if (fg < 2 && fr < a.Nt) ka = *(const uint4*)(a.ktok + ((long)p * a.Nt + fr) * CI + head * 16 + fg * 8);
            uint4 va = make_uint4(0, 0, 0, 0);
            {
                u16 e0 = 0, e1 = 0, e2 = 0, e3 = 0;
                const u16* vb = a.vtok + (long)p * a.Nt * CI + head * 16 + fr;
                if (fg * 4 + 0 < a.Nt) e0 = vb[(fg * 4 + 0) * CI];
                if (fg * 4 + 1 < a.Nt) e1 = vb[(fg * 4 + 1) * CI];
                if (fg * 4 + 2 < a.Nt) e2 = vb[(fg * 4 + 2) * CI];
                if (fg * 4 + 3 < a.Nt) e3 = vb[(fg * 4 + 3) * CI];
                va.x = (uint32_t)e0 | ((uint32_t)e1 << 16); va.y = (uint32_t)e2 | ((uint32_t)e3 << 16);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = mt * 16 + fr;
                uint4 qf = make_uint4(0, 0, 0, 0);
                if (fg < 2) qf = ldsQ[row * 16 + ((head * 2 + fg) ^ swzr(row))];
                f32x4_t s = {0.f, 0.f, 0.f, 0.f};
                s = mfma16(ka, qf, s);                            // rows j = fg*4 + r, col token = fr
                float mx = NEG_BIG;
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[r] = (fg * 4 + r < a.Nt) ? s[r] * 0.25f : NEG_BIG; mx = fmaxf(mx, s[r]); }
                mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[r] = __expf(s[r] - mx); ps += s[r]; }
                ps += __shfl_xor(ps, 16); ps += __shfl_xor(ps, 32);
                uint4 pb; pb.x = pack2bf(s[0], s[1]); pb.y = pack2bf(s[2], s[3]); pb.z = 0; pb.w = 0;
                f32x4_t o = {0.f, 0.f, 0.f, 0.f};
                o = mfma16(va, pb, o);                            // rows d = fg*4 + r, col token = fr
                const float inv = 1.f / ps;
                uint2 pk; pk.x = pack2bf(o[0] * inv, o[1] * inv); pk.y = pack2bf(o[2] * inv, o[3] * inv);
                // attn[token = row][head*16 + fg*4 .. +3]: chunk head*2 + (fg >> 1), 8-byte half (fg & 1)
                uint2* dst = (uint2*)(ldsP + row * 16 + ((head * 2 + (fg >> 1)) ^ swzr(row)));
                dst[fg & 1] = pk;
            }
        }
        __syncthreads();
        // ---- out projection: this wave's 32 columns
        f32x4_t oc[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) oc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = mi * 16 + fr;
                const uint4 af = ldsP[row * 16 + ((kc * 4 + fg) ^ swzr(row))];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) oc[mi][ni] = mfma16(af, wof[ni][kc], oc[mi][ni]);
            }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) ldsC[(mi * 16 + fg * 4 + r) * C + wave * 32 + ni * 16 + fr] = oc[mi][ni][r];
        __syncthreads();
        // ---- row-complete epilogue: one wave = one 256-wide row per pass
        const int col = lane * 4;
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const float4 b4 = *(const float4*)(a.bo + col), w4 = *(const float4*)(a.ln_w + col), g4 = *(const float4*)(a.ln_b + col);
            const int lr = pass * 8 + wave;
            const float4 c = *(const float4*)(ldsC + lr * C + col);
            uint2 rs;
            if constexpr (!L0) {
                const uint4* la = ldsA + buf * WM * CPR;
                rs = ((const uint2*)(la + lr * CPR + ((lane >> 1) ^ swzr(lr))))[lane & 1];
            } else {
                rs = *(const uint2*)(a.xin + (long)(t0 + lr) * C + col);
            }
            float v0 = c.x + b4.x + bf2f((u16)(rs.x & 0xffff)), v1 = c.y + b4.y + bf2f((u16)(rs.x >> 16));
            float v2 = c.z + b4.z + bf2f((u16)(rs.y & 0xffff)), v3 = c.w + b4.w + bf2f((u16)(rs.y >> 16));
            const float mean = wave_sum64((v0 + v1) + (v2 + v3)) * (1.0f / 256.0f);
            v0 -= mean; v1 -= mean; v2 -= mean; v3 -= mean;
            const float var = wave_sum64((v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3)) * (1.0f / 256.0f);
            const float rstd = 1.0f / sqrtf(var + a.eps);
            uint2 pk;
            pk.x = pack2bf(v0 * rstd * w4.x + g4.x, v1 * rstd * w4.y + g4.y);
            pk.y = pack2bf(v2 * rstd * w4.z + g4.z, v3 * rstd * w4.w + g4.w);
            *(uint2*)(a.out + (row0 + lr) * C + col) = pk;
        }
        if constexpr (!L0) { if (next < ntiles) DL_STORE(ra0, ra1, buf ^ 1); }
        __syncthreads();
        buf ^= 1;
    }
#undef DL_SRC
#undef DL_DST
#undef DL_LOAD
#undef DL_STORE
}

template <bool L0>
int launch(const LayerArgs& a, hipStream_t s) {
    constexpr int LDS_BYTES = (L0 ? 0 : 2 * WM * 32 * 16) + 2 * WM * 16 * 16 + WM * C * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)dec_image_layer_kernel<L0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) != hipSuccess) { msam_set_error("msam_decoder_image_layer: cannot raise the LDS limit"); return 2; }
        attr_set = true;
    }
    const int ntiles = a.rows / WM;
    const int grid = ntiles < 512 ? ntiles : 512;                 // two workgroups per CU
    const double flops = 2.0 * a.rows * ((L0 ? 0.0 : (double)CI * C) + (double)C * CI);
    msam_profile_mark(s, 1, flops);
    hipLaunchKernelGGL((dec_image_layer_kernel<L0>), dim3(grid), dim3(NTHR), LDS_BYTES, s, a);
    msam_profile_mark(s, 0, 0.0);
    return msam_check_launch("msam_decoder_image_layer");
}

}  // namespace

extern "C" int msam_decoder_image_layer(const msam_image_layer_t* p, void* stream) {
    if (!p || !p->xin || !p->wo || !p->bo || !p->ln_w || !p->ln_b || !p->ktok || !p->vtok || !p->out) {
        msam_set_error("msam_decoder_image_layer: null argument");
        return 1;
    }
    if (p->rows <= 0 || p->rows % T || p->Nt <= 0 || p->Nt > 16) { msam_set_error("msam_decoder_image_layer: bad sizes"); return 1; }
    LayerArgs a;
    a.xin = (const u16*)p->xin; a.q_shared = (const u16*)p->q_shared; a.wq = (const u16*)p->wq; a.bq = p->bq; a.peq = p->peq;
    a.wo = (const u16*)p->wo; a.bo = p->bo; a.ln_w = p->ln_w; a.ln_b = p->ln_b; a.eps = p->ln_eps;
    a.ktok = (const u16*)p->ktok; a.vtok = (const u16*)p->vtok; a.Nt = p->Nt; a.out = (u16*)p->out; a.rows = p->rows;
    hipStream_t s = (hipStream_t)stream;
    if (p->q_shared) return launch<true>(a, s);
    if (!p->wq || !p->bq || !p->peq) { msam_set_error("msam_decoder_image_layer: layer-1 form needs wq, bq, peq"); return 1; }
    return launch<false>(a, s);
}
