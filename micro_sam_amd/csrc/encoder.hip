// Host-side sequencing of the SAM ViT image encoder on the HIP kernels (no device code in this file).
// Stands behind `predictor.model.image_encoder(x)` (micro_sam/util.py:674) and, with uint8 input, also fuses
// `predictor.model.preprocess` (util.py:670).  Hyper-parameters: micro_sam/models/build_sam.py:40-113.
//
// Residual stream x: fp32 [B*4096, D].  Every GEMM operand is 16 bit (LN output, attention output, GELU output): bf16, or IEEE
// fp16 when msam_encoder_t.dtype16 == MSAM_F16 (same kernels and MFMA rate, 11 instead of 8 significand bits).
// Windowed blocks run on the 4096 real tokens only: the reference pads LN output with zeros to 70x70, so padding
// tokens are exactly q/k/v = bias, which the window attention kernel synthesises (no partition / un-partition copies).
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);

namespace {
constexpr long TOK = 4096;
inline long al(long x) { return (x + 255) & ~255L; }
struct EncWork { float* x; u16 *xn, *patches, *q, *k, *v, *attn, *hid, *n1, *col; float *n0, *n2;
                 unsigned char *xn8, *attn8, *hid8; float *rs_x, *rs_a, *rs_h; };     // fp8 copies + row scales (fp8 mode)
// DA = heads * stored head_dim: width of q / k / v / attention output (== D unless the heads are zero-padded, vit_h)
long enc_bytes(int D, int DA, int B, int fp8, int split) {
    const long R = (long)B * TOK;
    const long colw = split ? 4608 : 2304;      // the neck's 3 x 3 gather: [hi | lo] in split mode
    long n = al(R * D * 4) + al(R * D * 2) + al(R * 768 * 2) + 3 * al(R * DA * 2) + al(R * DA * 2) + al(R * 4 * D * 2) +
             al(R * 256 * 2) + al(R * colw * 2) + 2 * al(R * 256 * 4);
    if (fp8) n += al(R * D) + al(R * DA) + al(R * 4 * D) + 3 * al(R * 4);
    return n;
}
EncWork carve(void* base, int D, int DA, int B, int fp8, int split) {
    const long R = (long)B * TOK;
    char* p = (char*)base; EncWork w;
    auto take = [&](long b) { char* r = p; p += al(b); return r; };
    w.x = (float*)take(R * D * 4); w.xn = (u16*)take(R * D * 2); w.patches = (u16*)take(R * 768 * 2);
    w.q = (u16*)take(R * DA * 2); w.k = (u16*)take(R * DA * 2); w.v = (u16*)take(R * DA * 2);
    w.attn = (u16*)take(R * DA * 2); w.hid = (u16*)take(R * 4 * D * 2);
    w.n1 = (u16*)take(R * 256 * 2); w.col = (u16*)take(R * (split ? 4608 : 2304) * 2);
    w.n0 = (float*)take(R * 256 * 4); w.n2 = (float*)take(R * 256 * 4);
    w.xn8 = w.attn8 = w.hid8 = nullptr; w.rs_x = w.rs_a = w.rs_h = nullptr;
    if (fp8) {
        w.xn8 = (unsigned char*)take(R * D); w.attn8 = (unsigned char*)take(R * DA); w.hid8 = (unsigned char*)take(R * 4 * D);
        w.rs_x = (float*)take(R * 4); w.rs_a = (float*)take(R * 4); w.rs_h = (float*)take(R * 4);
    }
    return w;
}
// stored head_dim: 64 as is (vit_b / vit_l); 80 (vit_h) must come zero-padded to 96; 0 = unsupported
int stored_head_dim(const msam_encoder_t* enc) {
    if (enc->heads <= 0 || enc->embed_dim % enc->heads) return 0;
    const int hd = enc->embed_dim / enc->heads;
    const int st = enc->head_dim_stored ? enc->head_dim_stored : hd;
    if (hd == 64 && st == 64) return 64;
    if (hd == 80 && st == 96) return 96;
    return 0;
}
}  // namespace

extern "C" int64_t msam_encoder_workspace_bytes(const msam_encoder_t* enc, int32_t B) {
    if (!enc || B <= 0) return 0;
    const int hs = stored_head_dim(enc);
    if (!hs) return 0;
    return enc_bytes(enc->embed_dim, enc->heads * hs, B, enc->fp8, enc->split_io);
}

extern "C" int msam_encoder_forward(const msam_encoder_t* enc, const float* img_f32, const uint8_t* img_u8, int32_t h,
                                    int32_t w_, int32_t B, float* out, void* workspace, int64_t workspace_bytes, float* tap,
                                    int32_t tap_block, void* stream) {
    if (!enc || !out || !workspace || B <= 0 || (!img_f32 && !img_u8)) { msam_set_error("msam_encoder_forward: null argument"); return 1; }
    const int D = enc->embed_dim, H = enc->heads;
    const int HS = stored_head_dim(enc);
    if (D % 128 || !HS || enc->depth > MSAM_MAX_BLOCKS) {
        msam_set_error("msam_encoder_forward: unsupported geometry (need embed_dim % 128 == 0 and head_dim 64, or head_dim 80 "
                       "with head_dim_stored = 96 and zero-padded qkv / rel_pos / proj weights: vit_b / vit_l / vit_h)");
        return 1;
    }
    const int DA = H * HS;
    const float scale = 1.0f / sqrtf((float)(D / H));
    if (workspace_bytes < enc_bytes(D, DA, B, enc->fp8, enc->split_io)) { msam_set_error("msam_encoder_forward: workspace too small"); return 1; }
    const bool fp8 = enc->fp8 != 0;
    const int dt = enc->dtype16 == MSAM_F16 ? MSAM_F16 : MSAM_BF16;        // the 16-bit type of every operand / stored activation
    if (fp8 && dt == MSAM_F16) { msam_set_error("msam_encoder_forward: fp8 projections go with the bf16 mode only"); return 1; }
    if (fp8 && (D % 128 || DA % 128)) { msam_set_error("msam_encoder_forward: fp8 needs embed_dim and heads * head_dim % 128 == 0"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    EncWork w = carve(workspace, D, DA, B, enc->fp8, enc->split_io);
    const bool split = enc->split_io != 0;
    const int R = (int)(B * TOK);
    int e;
#define CHECK(x) do { if ((e = (x))) return e; } while (0)
    auto gemm = [&](const void* A, long lda, const void* W, int N, int K, const float* bias, void* o, int odt, long ldc,
                    int act, const void* resid, int rdt, long ldr, const float* table, int trows, int tcols, long tld,
                    long ldw = 0) {
        msam_gemm_t g{};
        g.A = A; g.lda = lda; g.W = W; g.ldw = ldw ? ldw : K; g.M = R; g.N = N; g.K = K; g.bias = bias;
        g.table = table; g.table_rows = trows; g.table_cols = tcols; g.table_ld = tld;
        g.resid = resid; g.resid_dtype = rdt; g.ldr = ldr; g.act = act; g.out = o; g.out_dtype = odt; g.ldc = ldc;
        g.use_glds = enc->use_glds;
        g.a_dtype = dt == MSAM_F16 ? MSAM_F16 : 0;
        return msam_gemm_bf16(&g, s);
    };
    // patch embedding (+ bias + absolute position embedding)
    if (split) {
        // [hi | lo | hi] patch rows (2304 wide) in the neck's gather buffer (free until the neck) against [Whi | Whi | Wlo]
        if (img_u8) CHECK(msam_patchify_u8_split16(img_u8, B, h, w_, dt, w.col, s));
        else CHECK(msam_patchify_split16(img_f32, B, dt, w.col, s));
        CHECK(gemm(w.col, 2304, enc->patch_w, D, 2304, enc->patch_b, w.x, MSAM_F32, D, 0, nullptr, 0, 0, enc->pos_embed,
                   (int)TOK, D, D));
    } else {
        if (img_u8) CHECK(msam_patchify_u8_16(img_u8, B, h, w_, dt, w.patches, s));
        else CHECK(msam_patchify16(img_f32, B, dt, w.patches, s));
        CHECK(gemm(w.patches, 768, enc->patch_w, D, 768, enc->patch_b, w.x, MSAM_F32, D, 0, nullptr, 0, 0, enc->pos_embed,
                   (int)TOK, D, D));
    }
    // fp8 projection: A fp8 [R, K] with row scales, W fp8 [N, K] with column scales
    auto gemm8 = [&](const void* A8, const float* rs, const void* W8, const float* cs, int N, int K, const float* bias, void* o,
                     int odt, long ldc, int act, const void* resid) {
        msam_gemm_t g{};
        g.A = A8; g.lda = K; g.W = W8; g.ldw = K; g.M = R; g.N = N; g.K = K; g.bias = bias;
        g.resid = resid; g.resid_dtype = resid ? MSAM_F32 : 0; g.ldr = N; g.act = act; g.out = o; g.out_dtype = odt; g.ldc = ldc;
        g.a_dtype = MSAM_FP8; g.row_scale = rs; g.col_scale = cs;
        return msam_gemm_bf16(&g, s);
    };
    for (int i = 0; i < enc->depth; ++i) {
        if (fp8) {
            if (!enc->qkv_w8[i] || !enc->qkv_cs[i] || !enc->proj_w8[i] || !enc->proj_cs[i] || !enc->lin1_w8[i] || !enc->lin1_cs[i] ||
                !enc->lin2_w8[i] || !enc->lin2_cs[i]) { msam_set_error("msam_encoder_forward: fp8 weights / scales missing"); return 1; }
            CHECK(msam_layernorm_fp8(w.x, enc->ln1_w[i], enc->ln1_b[i], 1e-6f, R, D, w.xn8, w.rs_x, s));
            msam_gemm_t g{};
            g.A = w.xn8; g.lda = D; g.W = enc->qkv_w8[i]; g.ldw = D; g.M = R; g.N = 3 * DA; g.K = D; g.bias = enc->qkv_b[i];
            g.out_mode = 1; g.q = w.q; g.k = w.k; g.v = w.v; g.heads = H; g.head_dim = HS; g.tokens = (int)TOK;
            g.a_dtype = MSAM_FP8; g.row_scale = w.rs_x; g.col_scale = enc->qkv_cs[i];
            CHECK(msam_gemm_bf16(&g, s));
        } else {
            CHECK(msam_layernorm(w.x, enc->ln1_w[i], enc->ln1_b[i], 1e-6f, R, D, w.xn, dt, 0, 0, s));
            msam_gemm_t g{};
            g.A = w.xn; g.lda = D; g.W = enc->qkv_w[i]; g.ldw = D; g.M = R; g.N = 3 * DA; g.K = D; g.bias = enc->qkv_b[i];
            g.out_mode = 1; g.q = w.q; g.k = w.k; g.v = w.v; g.heads = H; g.head_dim = HS; g.tokens = (int)TOK;
            g.use_glds = enc->use_glds;
            if (dt == MSAM_F16) { g.a_dtype = MSAM_F16; g.out_dtype = MSAM_F16; }
            CHECK(msam_gemm_bf16(&g, s));
        }
        if (enc->is_global[i])
            CHECK(msam_global_attention16(w.q, w.k, w.v, enc->rel_h[i], enc->rel_w[i], B, H, HS, scale, dt, w.attn, s));
        else
            CHECK(msam_window_attention16(w.q, w.k, w.v, enc->rel_h[i], enc->rel_w[i], enc->qkv_b[i], B, H, HS, scale, dt, w.attn, s));
        if (fp8) {
            CHECK(msam_quant_rows_fp8(w.attn, R, DA, w.attn8, w.rs_a, s));
            CHECK(gemm8(w.attn8, w.rs_a, enc->proj_w8[i], enc->proj_cs[i], D, DA, enc->proj_b[i], w.x, MSAM_F32, D, 0, w.x));
            CHECK(msam_layernorm_fp8(w.x, enc->ln2_w[i], enc->ln2_b[i], 1e-6f, R, D, w.xn8, w.rs_x, s));
            CHECK(gemm8(w.xn8, w.rs_x, enc->lin1_w8[i], enc->lin1_cs[i], 4 * D, D, enc->lin1_b[i], w.hid, MSAM_BF16, 4 * D,
                        MSAM_ACT_GELU, nullptr));
            CHECK(msam_quant_rows_fp8(w.hid, R, 4 * D, w.hid8, w.rs_h, s));
            CHECK(gemm8(w.hid8, w.rs_h, enc->lin2_w8[i], enc->lin2_cs[i], D, 4 * D, enc->lin2_b[i], w.x, MSAM_F32, D, 0, w.x));
        } else {
        CHECK(gemm(w.attn, DA, enc->proj_w[i], D, DA, enc->proj_b[i], w.x, MSAM_F32, D, 0, w.x, MSAM_F32, D, nullptr, 0, 0, 0));
        CHECK(msam_layernorm(w.x, enc->ln2_w[i], enc->ln2_b[i], 1e-6f, R, D, w.xn, dt, 0, 0, s));
        CHECK(gemm(w.xn, D, enc->lin1_w[i], 4 * D, D, enc->lin1_b[i], w.hid, dt, 4 * D, MSAM_ACT_GELU, nullptr, 0, 0,
                   nullptr, 0, 0, 0));
        CHECK(gemm(w.hid, 4 * D, enc->lin2_w[i], D, 4 * D, enc->lin2_b[i], w.x, MSAM_F32, D, 0, w.x, MSAM_F32, D, nullptr, 0,
                   0, 0));
        }
        if (tap && tap_block == i)
            if (hipMemcpyAsync(tap, w.x, (size_t)R * D * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) {
                msam_set_error("msam_encoder_forward: tap copy failed");
                return 2;
            }
    }
    // neck: conv1x1 -> LayerNorm2d -> conv3x3 (pad 1) -> LayerNorm2d, output NCHW fp32
    if (split) {
        // conv1x1: x as [hi | lo | hi] rows (3 D wide, in the MLP hidden buffer: 4 D wide, free here) against [Whi | Whi | Wlo]
        CHECK(msam_cast_f32_split16(w.x, dt, w.hid, R, D, s));
        CHECK(gemm(w.hid, 3L * D, enc->neck0_w, 256, 3 * D, nullptr, w.n0, MSAM_F32, 256, 0, nullptr, 0, 0, nullptr, 0, 0, 0));
        // LayerNorm2d in fp32 (in the LN-output buffer: D * 2 >= 256 * 4 bytes per token), gathered as [hi | lo]
        float* n1f = (float*)w.xn;
        CHECK(msam_layernorm(w.n0, enc->neck1_w, enc->neck1_b, 1e-6f, R, 256, n1f, MSAM_F32, 0, 0, s));
        CHECK(msam_im2col3x3_split16(n1f, B, 256, dt, w.col, s));
        // conv3x3: [hi | lo] x [Whi | Whi], then hi x Wlo accumulated on top (the third block of neck2_w's rows)
        CHECK(gemm(w.col, 4608, enc->neck2_w, 256, 4608, nullptr, w.n2, MSAM_F32, 256, 0, nullptr, 0, 0, nullptr, 0, 0, 0, 6912));
        CHECK(gemm(w.col, 4608, (const u16*)enc->neck2_w + 4608, 256, 2304, nullptr, w.n2, MSAM_F32, 256, 0, w.n2, MSAM_F32, 256,
                   nullptr, 0, 0, 0, 6912));
    } else {
    CHECK(msam_cast_f32_to_16(w.x, dt, w.xn, (long)R * D, s));
    CHECK(gemm(w.xn, D, enc->neck0_w, 256, D, nullptr, w.n0, MSAM_F32, 256, 0, nullptr, 0, 0, nullptr, 0, 0, 0));
    CHECK(msam_layernorm(w.n0, enc->neck1_w, enc->neck1_b, 1e-6f, R, 256, w.n1, dt, 0, 0, s));
    CHECK(msam_im2col3x3(w.n1, B, 256, w.col, s));
    CHECK(gemm(w.col, 2304, enc->neck2_w, 256, 2304, nullptr, w.n2, MSAM_F32, 256, 0, nullptr, 0, 0, nullptr, 0, 0, 0));
    }
    CHECK(msam_layernorm(w.n2, enc->neck3_w, enc->neck3_b, 1e-6f, R, 256, out, MSAM_F32, 0, (int)TOK, s));
#undef CHECK
    return 0;
}
