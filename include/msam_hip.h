/*
 * msam_hip.h - C ABI of libmsam_hip.so, the MI355X (gfx950) SAM inference core behind micro_sam's
 * SamPredictor boundary.
 *
 * The reference has no FFI: its seam is the duck-typed `SamPredictor` returned by
 * `micro_sam.util.get_sam_model` (micro_sam/util.py:318-476).  Every entry point below names the reference
 * call it stands behind.  Conventions:
 *   - plain C: pointers + sizes, no torch / C++ types.  All `void*` / `float*` buffers are DEVICE pointers
 *     owned by the caller; the library never allocates user-visible memory (workspace is caller-provided,
 *     sized by the *_workspace_bytes queries).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous.
 *   - return 0 on success, non-zero on error; msam_last_error() returns a static message.
 *     The Python shim (micro_sam_amd/_lib.py) raises ValueError / RuntimeError from it.
 *   - not re-entrant per model object (the reference's predictor is a stateful singleton,
 *     micro_sam/sam_annotator/_state.py:41-47); safe from any single host thread.
 */
#ifndef MSAM_HIP_H
#define MSAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSAM_MAX_BLOCKS 32
#define MSAM_F32 1
#define MSAM_BF16 2
#define MSAM_FP8 3                       /* OCP e4m3 (gfx950), not the fnuz variant of MI300 */
#define MSAM_F16 4                       /* IEEE fp16: the mask decoder's 16-bit type (msam_decoder_dtype) */
#define MSAM_U8 5                        /* raw image tiles (msam_to_image) */
#define MSAM_U16 6
#define MSAM_ACT_NONE 0
#define MSAM_ACT_GELU 1
#define MSAM_ACT_RELU 2
#define MSAM_ACT_SIGMOID 3               /* msam_strict_gemm only */

const char* msam_last_error(void);
int msam_abi_version(void);

/* ---------------------------------------------------------------------------------------------------
 * Building-block operators (each is what a torch op does inside segment_anything's modules; exported so
 * that tests can pin every kernel against the oracle separately).
 * ------------------------------------------------------------------------------------------------- */

/* out = act(A[M,K] * W[N,K]^T + bias + table + resid); bf16 operands, fp32 accumulate (MFMA).
 * Replaces torch.nn.Linear / 1x1 conv / patch-embed conv / transposed-conv GEMMs of the reference model. */
typedef struct {
    const void* A; int64_t lda;          /* bf16 [M,K], row stride lda (elements) */
    const void* W; int64_t ldw;          /* bf16 [N,K] (torch Linear weight layout) */
    int32_t M, N, K;                     /* N % 128 == 0, K % 64 == 0 */
    const float* bias;                   /* [N] or NULL */
    const float* table;                  /* optional additive table, indexed [(row % table_rows)*table_ld + n] */
    int32_t table_rows, table_cols;      /* applied to columns n < table_cols */
    int64_t table_ld;
    const void* resid;                   /* optional residual, resid_dtype MSAM_F32 / MSAM_BF16, 0 = none */
    int32_t resid_dtype; int32_t resid_rows;  /* resid row = row % resid_rows (resid_rows 0 -> row) */
    int64_t ldr;
    int32_t act;                         /* MSAM_ACT_* applied last */
    void* out; int32_t out_dtype; int64_t ldc;   /* plain output (out_mode 0) */
    int32_t out_mode;                    /* 0 plain, 1 qkv-split (ViT attention layout), 2 kv-split (decoder), 3 = `out` as hi + lo pairs of the
                                          * 16-bit out_dtype, rows [hi | lo | hi] with ldc == 3 N (128 x 128 tile path only: the A operand of a
                                          * product against [Whi | Whi | Wlo] weight rows, msam_twoway_layer_t.mlp2_ws) */
    void* q; void* k; void* v;           /* out_mode 1: q,k,v -> [B,heads,tokens,hd] bf16
                                            out_mode 2 (N == 256): k <- cols 0..127 as [M,128] bf16,
                                            v <- cols 128..255 transposed [M/tokens,128,tokens] bf16 */
    int32_t heads, head_dim, tokens;
    int32_t use_glds;                    /* 1: global_load_lds staging variant, 0: register staging */
    int32_t ln_mode;                     /* fused LayerNorm epilogue (N == 256 only, row-complete 64x256 tile kernel):
                                            0 none, 1 LayerNorm over the 256-wide row, 2 LayerNorm over each 64-column
                                            group followed by exact GELU; applied after bias/table/resid */
    const float* ln_w; const float* ln_b; float ln_eps;    /* [256] (mode 1) or [64] (mode 2) */
    /* fp8 operands (BASELINE config 5): a_dtype = MSAM_FP8 -> A and W are OCP e4m3 bytes (lda / ldw / K in elements,
     * K % 128 == 0, N % 256 == 0), out = act(acc * row_scale[m] * col_scale[n] + bias + resid); 0 / MSAM_BF16 = bf16;
     * MSAM_F16 = A and W are IEEE fp16 (128 x 128 tile kernel, plain / grouped launches; 16-bit outputs follow out_dtype:
     * MSAM_F16 -> fp16, MSAM_BF16 -> bf16) */
    int32_t a_dtype;
    const float* row_scale; const float* col_scale;        /* fp32 [M], [N] */
    /* split_k > 1 (128 x 128-tile bf16 / fp16 path): the contraction is cut into split_k slices, one workgroup per (tile, slice),
     * partial tiles are added into `out` with fp32 atomics (out is zeroed first).  For products with a small output and a very long
     * contraction (the weight gradients of fine-tuning: dW = dY^T X over all rows).  Plain fp32 output with ldc == N, no bias /
     * table / residual / activation; K % (64 * split_k) == 0.  0 / 1 = off.  The slices meet in fp32 atomics: the result is NOT
     * bit-reproducible from run to run (order of the additions; training gradients only - no inference path sets it). */
    int32_t split_k;
    /* ln_mode 1 only (the mask decoder's token side): the normalised row is ALSO written as 16-bit operand copies for the next
     * products - ln_out_a [M,256] = round16(y + ln_add[row]) (ln_add fp32 [M,256] or NULL), ln_out_b [M,256] = round16(y); the 16-bit
     * type is the operands' (a_dtype).  NULL = not written.  One launch instead of GEMM + LayerNorm + add/cast launches. */
    const float* ln_add; void* ln_out_a; void* ln_out_b;
} msam_gemm_t;
int msam_gemm_bf16(const msam_gemm_t* p, void* stream);
/* n <= MSAM_GEMM_GROUP_MAX independent products (each as for msam_gemm_bf16; 128 x 128-tile bf16 path only: no fused LayerNorm,
 * no fp8, no qkv-split output) in ONE launch: the decoder's token-side projections are latency-bound one at a time. */
#define MSAM_GEMM_GROUP_MAX 5
int msam_gemm_group_bf16(const msam_gemm_t* items, int32_t n, void* stream);

/* Weights-stationary streaming GEMM for the decoder's image-token stream (M = P*4096 rows, N,K in {128,256}):
 * out = epi(A[M,K] * W[N,K]^T), A / W / out bf16, contiguous rows (lda = K, ldw = K).  Same epilogue vocabulary as
 * msam_gemm_bf16 (bias, row-indexed table on the first table_cols columns, bf16 residual, ln_mode 1/2 for N == 256),
 * or kv_split: N == 256: k_out <- columns 0..127 as [M,128], vT_out <- columns 128..255 as [M/tokens,128,tokens];
 * N == 128: all columns to vT_out transposed (k_out unused).  ln_mode 2 is also available for N == 128. */
typedef struct {
    const void* A; const void* W; int32_t M, N, K;
    const float* bias;
    const float* table; int32_t table_rows, table_cols; int64_t table_ld;
    const void* resid; int32_t resid_rows; int64_t ldr;        /* bf16 residual, row = row % resid_rows (0 -> row) */
    int32_t ln_mode; const float* ln_w; const float* ln_b; float ln_eps;
    void* out; int64_t ldc;
    int32_t kv_split; void* k_out; void* vT_out; int32_t tokens;
    int32_t head_major;      /* plain output stored as [M/tokens][N/16][tokens][16] (heads of 16 columns contiguous) */
} msam_wsgemm_t;
int msam_wsgemm_bf16(const msam_wsgemm_t* p, void* stream);

/* Fused image-side half of a two-way block on the per-prompt stream (upstream TwoWayAttentionBlock step 4):
 *   q = (x + pos) Wq^T + bq  [layer 1]  or the precomputed prompt-independent q_shared [layer 0];
 *   attn = softmax_j(q . k_tok[j] / 4) v_tok  (8 heads x 16, Nt <= 16 prompt tokens);
 *   out = LayerNorm(x + attn Wo^T + bo).
 * xin: bf16 [rows,256] (layer 1, may alias out) or the shared src [4096,256] (layer 0: q_shared != NULL);
 * peq: fp32 [4096,128] = pos Wq^T; ktok / vtok: bf16 [rows/4096 * Nt, 128]; rows % 4096 == 0. */
typedef struct {
    const void* xin; const void* q_shared;
    const void* wq; const float* bq; const float* peq;
    const void* wo; const float* bo; const float* ln_w; const float* ln_b; float ln_eps;
    const void* ktok; const void* vtok; int32_t Nt;
    void* out; int32_t rows;
} msam_image_layer_t;
int msam_decoder_image_layer(const msam_image_layer_t* p, void* stream);

/* Token -> image cross attention with the K / V projections folded into the token side (reference:
 * segment_anything/modeling/transformer.py Attention as called by TwoWayAttentionBlock.cross_attn_token_to_image and
 * TwoWayTransformer.final_attn_token_to_image):
 *   S = (keys + pe) Wk^T q / 4 is evaluated as keys . (Wk^T q) + (pe Wk^T + bk) . q, and softmax(S) (keys Wv^T + bv) as
 *   (softmax(S) keys) Wv^T + bv, so the per-prompt image-token stream is read once and no K / V stream is written.
 * keys: bf16 [Pk,4096,256] (kv_shared = 1: every prompt uses prompt 0's stream; kv_shared = 2: per-prompt streams in the
 * blocked layout msam_i2t01_fused writes); qtok: bf16 [P,Nt,128] projected
 * queries, 1 <= Nt <= 8; wk, wv: bf16 [128,256]; tabk: bf16 [4096,128] = pe Wk^T + bk; bv fp32 [128];
 * out: bf16 [P,Nt,128] (input of the attention's out_proj).  workspace >= msam_t2i_fold_workspace_bytes(P). */
int64_t msam_t2i_fold_workspace_bytes(int32_t P);
int msam_t2i_fold_attention(const void* keys, int32_t kv_shared, const void* qtok, int32_t P, int32_t Nt, const void* wk,
                            const void* tabk, const void* wv, const float* bv, void* out, void* workspace,
                            int64_t workspace_bytes, void* stream);

/* Image -> token cross attention + out_proj + residual + norm4 of a two-way block with the q / out projections folded
 * into the (<= 8) prompt tokens (same reference code as msam_decoder_image_layer; TwoWayAttentionBlock.
 * cross_attn_image_to_token + norm4):
 *   S = keys . (Wq_h^T k_{t,h}) + (pe Wq^T + bq)_h . k_{t,h},  P = softmax_t(S / 4),
 *   out = LayerNorm(keys + sum_{h,t} P (Wo[:, 16h:16h+16] v_{t,h}) + bo).
 * xin: bf16 [Px,4096,256] (x_shared != 0: every prompt reads prompt 0, layer 0); ktok / vtok: bf16 [P,Nt,128] projected
 * prompt tokens; wq bf16 [128,256]; tabq bf16 [4096,128] = pe Wq^T + bq; wo bf16 [256,128]; out bf16 [P,4096,256], may
 * alias xin.  workspace >= msam_i2t_fold_workspace_bytes(P). */
int64_t msam_i2t_fold_workspace_bytes(int32_t P);
int msam_i2t_fold_layer(const void* xin, int32_t x_shared, const void* ktok, const void* vtok, int32_t P, int32_t Nt,
                        const void* wq, const void* tabq, const void* wo, const float* bo, const float* ln_w,
                        const float* ln_b, float ln_eps, void* out, void* workspace, int64_t workspace_bytes, void* stream);

/* Chained forms of the two-way transformer's image side for prompts that share ONE source (AMG: no mask prompts, every
 * prompt of a tile starts from image embedding + no_mask_embed) - csrc/decfold_tok.hip.  The layer-0 output stream is never
 * written: each wave recomputes its 16-token tile of it from the L2-resident source and feeds it, rounded to 16 bits exactly
 * as the stream would have stored it, into the next stage.
 * BLOCKED layout: [tile of 16 tokens][k-step s][lane = 16 g + token][8 channels 32 s + 8 g ..] (one MFMA operand fragment =
 * 1 KiB contiguous; a tile stays contiguous: 8 KiB for 256 channels).
 *   msam_chain_prepare_tables: blocked copies of the shared tables, once per decode: src d16 [4096,256] (the shared source),
 *       q0 d16 [4096,128] = (src + pe) Wq0^T + bq0, tabk d16 [4096,128] = pe Wk^T + bk of the layer-1 token->image attention,
 *       tabq1 d16 [4096,128] = pe Wq^T + bq of the layer-1 image->token block -> tables (>= msam_chain_tables_bytes());
 *   msam_i2t_fold_operands: per-prompt operands of an image->token layer in MFMA fragment order (K' = Wq_h^T k_{t,h},
 *       V'^T = Wo[:,16h:16h+16] v_{t,h} + bo / 8, block-diagonal k; ktok / vtok d16 [P,Nt,128], wq d16 [128,256], wo d16
 *       [256,128]) into `operands` (>= msam_i2t_fold_operand_bytes(P)); with_kfold = 0 skips K' (layer 0 on a shared source
 *       takes its scores from q0);
 *   msam_i2t0_t2i_fused: layer-0 image->token block (tables, operands0, norm4 of layer 0) chained into the layer-1
 *       token->image attention (qtok d16 [P,Nt,128] projected queries, wk / wv d16 [128,256], bv fp32 [128]): out d16
 *       [P,Nt,128] as msam_t2i_fold_attention on the layer-0 output would give; no per-prompt stream is read or written;
 *   msam_i2t01_fused: the same layer-0 block chained into the layer-1 image->token block (operands1, norm4 of layer 1):
 *       out d16 [P] x blocked [256][8][64][8] = the layer-1 output stream in the BLOCKED layout (msam_t2i_fold_attention with
 *       kv_shared = 2 and msam_upscale_fused_layout with keys_blocked = 1 read it).
 * 1 <= Nt <= 8; one 8-wave workgroup per prompt at a time (meant for P >= ~half the CU count). */
int64_t msam_chain_tables_bytes(void);
int msam_chain_prepare_tables(const void* src, const void* q0, const void* tabk, const void* tabq1, void* tables, void* stream);
int64_t msam_i2t_fold_operand_bytes(int32_t P);
int msam_i2t_fold_operands(const void* ktok, const void* vtok, int32_t P, int32_t Nt, const void* wq, const void* wo,
                           const float* bo, int32_t with_kfold, void* operands, void* stream);
int64_t msam_i2t0_t2i_workspace_bytes(int32_t P);
int msam_i2t0_t2i_fused(const void* tables, const void* operands0, const float* ln0_w, const float* ln0_b, float ln_eps,
                        const void* qtok, int32_t P, int32_t Nt, const void* wk, const void* wv, const float* bv, void* out,
                        void* workspace, int64_t workspace_bytes, void* stream);
/* Second form of msam_i2t0_t2i_fused (fewer LDS operand reads and MFMAs per tile; csrc/decfold_tok.hip "second form"): norm4 of
 * layer 0 is folded into the attention's operands and the value projection is taken before that LayerNorm by linearity.
 *   msam_chain_prepare_tables2: prompt-independent tables, once per decode: src d16 [4096,256] (row-major), wv / bv / wk of the
 *       layer-1 token->image attention, ln0_w / ln0_b = norm4 of layer 0, wo0 d16 [256,128] / bo0 = out_proj of the layer-0
 *       image->token attention -> tables2 (>= msam_chain_tables2_bytes());
 *   msam_t2i_fold_values: per prompt, from the layer-0 image->token value tokens vtok0 d16 [P,Nt,128] (the same tensor that went
 *       into msam_i2t_fold_operands for layer 0) -> mf (>= msam_t2i_fold_values_bytes(P));
 *   msam_i2t0_t2i_fused_v2: same result as msam_i2t0_t2i_fused (other rounding points). */
int64_t msam_chain_tables2_bytes(void);
int msam_chain_prepare_tables2(const void* src, const void* wv, const float* bv, const void* wk, const float* ln0_w,
                               const float* ln0_b, const void* wo0, const float* bo0, void* tables2, void* stream);
/* the weight-only part of tables2 once per model (const2 >= msam_chain_const2_bytes()), then per decode the source-dependent rest */
int64_t msam_chain_const2_bytes(void);
int msam_chain_prepare_const2(const void* wv, const float* bv, const void* wk, const float* ln0_w, const float* ln0_b,
                              const void* wo0, const float* bo0, void* const2, void* stream);
int msam_chain_prepare_tables2_c(const void* src, const void* const2, void* tables2, void* stream);
int64_t msam_t2i_fold_values_bytes(int32_t P);
int msam_t2i_fold_values(const void* vtok0, int32_t P, int32_t Nt, const void* tables2, void* mf, void* stream);
/* msam_i2t_fold_operands and msam_t2i_fold_values in one launch */
int msam_i2t_fold_operands_values(const void* ktok, const void* vtok, int32_t P, int32_t Nt, const void* wq, const void* wo,
                                  const float* bo, int32_t with_kfold, const void* tables2, void* operands, void* mf, void* stream);
int64_t msam_i2t0_t2i_v2_workspace_bytes(int32_t P);
int msam_i2t0_t2i_fused_v2(const void* tables, const void* tables2, const void* operands0, const void* mf, const float* ln0_w,
                           float ln_eps, const void* qtok, int32_t P, int32_t Nt, const void* wk, void* out, void* workspace,
                           int64_t workspace_bytes, void* stream);
int msam_i2t01_fused(const void* tables, const void* operands0, const float* ln0_w, const float* ln0_b, const void* operands1,
                     const float* ln1_w, const float* ln1_b, float ln_eps, int32_t P, int32_t Nt, void* out, void* stream);

/* Output up-scaling + hyper-network product of the mask decoder in one pass over the image-token stream (reference:
 * segment_anything/modeling/mask_decoder.py MaskDecoder.predict_masks: output_upscaling = ConvTranspose2d(256,64,2,2),
 * LayerNorm2d(64), GELU, ConvTranspose2d(64,32,2,2), GELU; masks = hyper_in @ upscaled).
 * keys: bf16 [P,4096,256] (token = 64*ty + tx); w1: bf16 [256 = sub*64 + c1, 256] with sub = 2*dy + dx of the first
 * transposed convolution, b1 fp32 [256]; ln_w / ln_b fp32 [64]; w2: bf16 [128 = sub2*32 + c2, 64], b2 fp32 [32];
 * hyper: fp32 [P,4,hyper_ld], the first 32 entries of mask (mask0 + m) are the hyper-network weights of output mask m;
 * low_res: fp32 [P,nmask,256,256], pixel (4*ty + 2*dy + dy2, 4*tx + 2*dx + dx2). */
int msam_upscale_fused(const void* keys, int32_t P, const void* w1, const float* b1, const float* ln_w, const float* ln_b,
                       float ln_eps, const void* w2, const float* b2, const float* hyper, int32_t hyper_ld, int32_t mask0,
                       int32_t nmask, float* low_res, void* stream);
/* the same with the stream layout stated.  keys_blocked is a bit field: bit 0 = `keys` is in the blocked layout msam_i2t01_fused writes
 * (0: row-major [P,4096,256]); bit 1 (round 6) = w1 / b1 are CENTRED over the 64 output channels of every sub-pixel (rows sub*64 .. sub*64+63
 * of w1 minus their mean row, b1 minus its mean: msam_decoder_t.up1_centred) - LayerNorm2d's mean is then zero by construction and is not
 * computed.  Plain weights with bit 1 clear run the general kernel. */
int msam_upscale_fused_layout(const void* keys, int32_t keys_blocked, int32_t P, const void* w1, const float* b1,
                              const float* ln_w, const float* ln_b, float ln_eps, const void* w2, const float* b2,
                              const float* hyper, int32_t hyper_ld, int32_t mask0, int32_t nmask, float* low_res, void* stream);
/* the same with the output type stated: low_res_dtype MSAM_F32 (fp32 [P,nmask,256,256]) or MSAM_F16 (fp16: what
 * msam_postprocess_masks16 reads - the AMG path keeps its low-res logits in 16 bits between the two kernels) */
int msam_upscale_fused_out(const void* keys, int32_t keys_blocked, int32_t P, const void* w1, const float* b1,
                           const float* ln_w, const float* ln_b, float ln_eps, const void* w2, const float* b2,
                           const float* hyper, int32_t hyper_ld, int32_t mask0, int32_t nmask, void* low_res,
                           int32_t low_res_dtype, void* stream);

/* Live measurement of the GEMM kernel (the dominant kernel of the hot path) for bench.py's roofline leg:
 * after msam_profile_enable(1) every msam_gemm_bf16 launch is bracketed by HIP events on its stream;
 * msam_profile_collect synchronises them and returns the number of launches, their summed duration (ms) and
 * summed 2*M*N*K.  Not thread safe; at most 4096 launches between collects (later ones are not recorded). */
/* tuning hook: s_setprio around the stage-1 MFMA cluster of up_fused_kernel (1 = on, default) */
int msam_upscale_set_prio(int32_t prio);
/* tuning hook: fold_attn_kernel operand staging (0 registers, 2 workgroups per CU; 1 LDS-DMA, 3 workgroups per CU) */
int msam_fold_attn_set_dma(int32_t on);
/* named integer tuning knobs of the decoder stream kernels (A/B experiments, tests): "i2t_variant" (1 = token-owner kernel,
 * default; 0 = 4-wave tile kernel), "i2t_wg_per_cu", "dec_chain" (1 = the decoder takes the chained forms above when the
 * prompts share one source, Nt <= 8 and P >= "dec_chain_min_p"; default 1 / 128), "chain_variant" (builds of the chained
 * kernels: 9 = default, second attention form; 6 = first form; csrc/decfold_tok.hip), "up_gelu16" (1 = the up-scaling's GELUs in packed fp16 arithmetic, default in the fp16 decoder build;
 * 0 = packed fp32).  Returns 0, 1 for an unknown key. */
int msam_tune_set(const char* key, int32_t value);
/* debug hook: phase timing of the folded image->token kernel (see csrc/decfold.hip, tools/i2t_timing.py) */
int msam_debug_i2t_timing(int32_t enable, uint64_t* host_out);
/* tuning / test hook: operand staging of the 256 x 256 tile kernel behind msam_gemm_bf16 (0 registers two tiles ahead,
 * 1 registers with the LDS write behind the barrier, 2 LDS-DMA; -1 = built-in default or MSAM_GEMM256_STAGING). */
int msam_gemm256_set_staging(int staging);
/* debug hook: timeline of the two-workgroups-per-CU kernel (staging 4).  device_words = 64 uint64 per workgroup of the persistent grid
 * (2 per CU): [0] HW_ID, [1] XCC_ID, then s_memrealtime stamps (100 MHz): per tile its start and the end of its k-loop, last = exit;
 * NULL switches it off (tools/gemm_probe.py). */
int msam_gemm_set_trace(void* device_words);
int msam_profile_enable(int on);
int msam_profile_collect(int32_t* launches, double* total_ms, double* total_flops);
/* Per kernel family (arrays of MSAM_PROFILE_FAMILIES: launches, summed ms, flops, algorithmic HBM bytes):
 * [0] gemm256_kernel (256 x 256 tile MFMA GEMM, bf16 or fp8: the encoder's large projections), [1] weights-stationary streaming
 * kernels (wsgemm_kernel / dec_image_layer_kernel), [2] fold_i2t_kernel, [3] fold_attn_kernel, [4] up_fused_kernel (the decoder
 * kernels that stream the per-prompt image-token stream once; HBM-bound), [5] gemm_kernel / gemm_ln_kernel (128 x 128 and
 * 64 x 256 tile MFMA GEMMs: patch embedding, neck, the latency-bound token-side projections). */
#define MSAM_PROFILE_FAMILIES 8
int msam_profile_collect_family(int32_t* launches, double* ms, double* flops, double* bytes);

/* util._to_image (micro_sam/util.py:618-651) on the device, bit for bit: in [H,W,C] (C = 1: gray, replicated; C = 2: third
 * channel zero; C > 3: first three) of dtype MSAM_U8 / MSAM_U16 / MSAM_F32 -> out uint8 [H,W,3] with per-channel
 * ((x - min) / (max - min + 1e-7)) * 255 in float32, truncated.  workspace: 32 bytes. */
int msam_to_image(const void* in, int32_t in_dtype, int32_t H, int32_t W, int32_t C, uint8_t* out, void* workspace, void* stream);

/* Greedy mask NMS (micro_sam/util.py:1589-1668 `_batched_mask_nms`): masks as bit masks uint32
 * [K, ceil(H/32), W], order int32 [K] = mask indices by descending score, boxes fp32 [K,4] xyxy and area int32 [K] indexed by
 * mask.  Pairs whose boxes share no area never suppress; overlap = intersection / union, or intersection / (min area + 1e-6)
 * with intersection_over_min; a mask is dropped when its overlap with a kept, higher-scored mask is > thresh.
 * keep_flags int32 [K] in sorted order; mask_scratch: K * ceil(K/64) uint64. */
int msam_mask_nms(const uint32_t* bits, const int32_t* order, const float* boxes, const int32_t* area, int32_t K, int32_t H,
                  int32_t W, float thresh, int32_t intersection_over_min, uint64_t* mask_scratch, int32_t* keep_flags,
                  void* stream);

/* AutomaticMaskGenerator.generate(output_mode="instance_segmentation") of a single-crop device state in one call
 * (micro_sam/instance_segmentation.py:99-144,463-530 + util.mask_data_to_segmentation micro_sam/util.py:1773-1848):
 * threshold / crop-edge filters, greedy box NMS, paint by descending area, connected components, drop the largest component
 * (with_background) and components below min_object_size, consecutive relabel.  1 <= N <= 4096 candidates: iou / stability
 * fp32 [N], boxes int32 [N,4] xyxy in the crop frame, area int32 [N], bits uint32 [N, ceil(H/32), W]; crop_box: HOST int32[4]
 * (x0, y0, x1, y1); labels int32 [H, W]; flag int32 [1] reads 0 when the component labelling converged.  No host
 * synchronisation; 15 kernels on `stream`. */
int64_t msam_amg_generate_workspace_bytes(int32_t N, int32_t H, int32_t W);
int msam_amg_generate_labels(const float* iou, const float* stability, const int32_t* boxes, const int32_t* area,
                             const uint32_t* bits, int32_t N, int32_t H, int32_t W, const int32_t* crop_box,
                             float pred_iou_thresh, float stability_score_thresh, float box_nms_thresh,
                             int32_t min_object_size, int32_t with_background, int32_t* labels, int32_t* flag,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* util.mask_data_to_segmentation(label_masks=True, merge_exclusively=False) (micro_sam/util.py:1773-1848) for masks that are already
 * selected: paint `order` (int32 [K] mask indices into bits, the paint order: later masks overwrite; K from k_dev int32[1] in device
 * memory when k_dev != NULL), connected components in the reference's numbering, drop components below min_object_size and - with_background -
 * the largest one counting label 0, consecutive relabel.  labels int32 [H, W]; flag int32 [1] reads 0 when the labelling converged.
 * 7 kernels on `stream`, no host synchronisation. */
int64_t msam_labels_from_masks_workspace_bytes(int32_t H, int32_t W);
int msam_labels_from_masks(const uint32_t* bits, const int32_t* order, int32_t K, const int32_t* k_dev, int32_t H, int32_t W,
                           int32_t min_object_size, int32_t with_background, int32_t* labels, int32_t* flag,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* ---- fine-tuning (micro_sam/training/sam_trainer.py:131-425, trainable_sam.py:12-114): backward
 * kernels of the mask decoder's non-GEMM pieces (the GEMMs run msam_gemm_bf16 in both directions: dX = dY W, dW = dY^T X).
 * msam_layernorm_backward: x, dy, dx fp32 [rows, dim] (dim 64 / 128 / 256; 768 / 1024 / 1280 for the encoder), dweight / dbias fp32 [dim] ACCUMULATED
 *   (zero them first).
 * msam_attention_forward / backward: softmax(scale q k^T) v for q fp32 [BH, Nq, D], k / v fp32 [BH, Nk, D], D = 16 or 32;
 *   lse fp32 [BH, Nq] (log-sum-exp of the scaled scores, saved for the backward pass), delta: workspace fp32 [BH, Nq]. */
int msam_layernorm_backward(const float* x, const float* weight, const float* dy, float eps, int64_t rows, int32_t dim,
                            float* dx, float* dweight, float* dbias, void* stream);
int msam_attention_forward(const float* q, const float* k, const float* v, int32_t BH, int32_t Nq, int32_t Nk, int32_t D,
                           float scale, float* out, float* lse, void* stream);
int msam_attention_backward(const float* q, const float* k, const float* v, const float* out, const float* dout,
                            const float* lse, int32_t BH, int32_t Nq, int32_t Nk, int32_t D, float scale, float* dq, float* dk,
                            float* dv, float* delta, void* stream);

/* Attention of the image encoder with its decomposed relative position bias, forward + backward, for un-frozen fine-tuning
 * (segment_anything/modeling/image_encoder.py Attention + add_decomposed_rel_pos; reference training entry
 * micro_sam/training/trainable_sam.py:71-81 image_embeddings_oft): queries = keys = the tokens of ONE Gh x Gw grid (a 14 x 14
 * window or the 64 x 64 image; Gw <= 64), key j = (kh, kw) = (j / Gw, j % Gw),
 *   s_ij = scale q_i k_j + bias_h[i][kh] + bias_w[i][kw],  out = softmax_j(s) v.
 * q / k / v / out / dout / dq / dk / dv fp32 [BH, Gh*Gw, D] (D = 64 or 80), bias_h / dbias_h fp32 [BH, Gh*Gw, Gh],
 * bias_w / dbias_w fp32 [BH, Gh*Gw, Gw] (the caller forms the biases from the unscaled queries and the rel-pos tables and
 * propagates their gradients), lse / delta fp32 [BH, Gh*Gw] (delta: workspace).  msam_layernorm_backward additionally
 * takes the encoder widths 768 / 1024 / 1280. */
int msam_relpos_attention_forward(const float* q, const float* k, const float* v, const float* bias_h, const float* bias_w,
                                  int32_t BH, int32_t Gh, int32_t Gw, int32_t D, float scale, float* out, float* lse,
                                  void* stream);
int msam_relpos_attention_backward(const float* q, const float* k, const float* v, const float* bias_h, const float* bias_w,
                                   const float* out, const float* dout, const float* lse, int32_t BH, int32_t Gh, int32_t Gw,
                                   int32_t D, float scale, float* dq, float* dk, float* dv, float* dbias_h, float* dbias_w,
                                   float* delta, void* stream);

/* Operands of a weight gradient in one pass (training; micro_sam_amd/training/functional.py _Linear): x [M, K] fp32 or bf16 (x_dtype
 * MSAM_F32 / MSAM_BF16), row stride ldx elements -> out16 [M, K] bf16 copy, outT [K, M] bf16 transpose, colsum [K] fp32 column sums
 * ADDED to the caller's buffer (atomics; the bias gradient).  Any of the three outputs may be NULL.  K % 4 == 0, ldx % 4 == 0.
 * Replaces torch's cast + strided transpose copy + sum launches behind the reference's autograd (torch.nn.functional.linear backward). */
int msam_cast_transpose(const void* x, int32_t x_dtype, int64_t M, int32_t K, int64_t ldx, void* out16, void* outT, float* colsum,
                        void* stream);

/* Row LayerNorm over the last dim (torch.nn.LayerNorm / LayerNorm2d on token-major data).
 * x fp32 [rows, dim] -> out (fp32 or bf16) [rows, dim]; optional exact GELU afterwards.
 * out_nchw_hw > 0: write fp32 output transposed to [rows/hw, dim, hw] (the encoder's NCHW result). */
int msam_layernorm(const float* x, const float* weight, const float* bias, float eps, int64_t rows, int32_t dim,
                   void* out, int32_t out_dtype, int32_t gelu, int32_t out_nchw_hw, void* stream);

/* fp8 activations (BASELINE config 5; OCP e4m3, one fp32 scale per row = per token: q = round(y * 448 / amax(row))).
 * msam_layernorm_fp8: LayerNorm (fp32 statistics) of fp32 [rows, dim] straight to fp8 [rows, dim] + row_scale [rows];
 * msam_quant_rows_fp8: bf16 [rows, dim] -> fp8 + row scales (attention output, MLP hidden). */
int msam_layernorm_fp8(const float* x, const float* weight, const float* bias, float eps, int64_t rows, int32_t dim,
                       void* out_fp8, float* row_scale, void* stream);
int msam_quant_rows_fp8(const void* x_bf16, int64_t rows, int32_t dim, void* out_fp8, float* row_scale, void* stream);
/* fp32 [B,3,1024,1024] (output of Sam.preprocess) -> bf16 patch matrix [B*4096, 768] (c,ky,kx order). */
int msam_patchify(const float* img, int32_t B, void* out_bf16, void* stream);
int msam_patchify16(const float* img, int32_t B, int32_t dtype16, void* out16, void* stream);
/* uint8 HWC [B,h,w,3] (h,w <= 1024; output of ResizeLongestSide.apply_image) -> normalised, zero padded bf16 patch
 * matrix [B*4096,768]: fuses Sam.preprocess (micro_sam/util.py:670) into the patch gather. */
int msam_patchify_u8(const uint8_t* img, int32_t B, int32_t h, int32_t w, void* out_bf16, void* stream);
int msam_patchify_u8_16(const uint8_t* img, int32_t B, int32_t h, int32_t w, int32_t dtype16, void* out16, void* stream);
/* bf16 [B,64,64,C] -> bf16 [B*4096, 9*C] rows for the 3x3 / pad 1 neck convolution ((ky,kx,c) column order). */
int msam_im2col3x3(const void* x_bf16, int32_t B, int32_t C, void* out_bf16, void* stream);
int msam_cast_f32_to_bf16(const float* x, void* out_bf16, int64_t n, void* stream);
int msam_cast_f32_to_16(const float* x, int32_t dtype16, void* out16, int64_t n, void* stream);
/* ResizeLongestSide.apply_image (micro_sam/util.py:663: Pillow's BILINEAR resize of the uint8 RGB image) on the device: one
 * fixed-point resampling pass along one axis, out[o] = clip8((2^21 + sum_t in[first[o] + t] * coef[o][t]) >> 22).  in: uint8 [B,H,W,C];
 * axis 1 = along W -> out [B,H,n_out,C], axis 0 = along H -> out [B,n_out,W,C]; bounds int32 [n_out,2] = (first tap, taps), coefs int32
 * [n_out,ksize] (device memory; micro_sam_amd.transforms.pil_bilinear_tables = Pillow's precompute_coeffs + normalize_coeffs_8bpc).
 * Pillow's order: horizontal pass into an 8-bit intermediate image, then the vertical pass. */
int msam_resample_u8(const uint8_t* in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t axis, int32_t n_out,
                     const int32_t* bounds, const int32_t* coefs, int32_t ksize, uint8_t* out, void* stream);
/* hi + lo operand pairs of the encoder's "split" sites (patch embedding, neck: msam_encoder_t.split_io): a value enters its product
 * as hi = round16(v), lo = round16(v - hi); rows are written [hi | lo | hi] (K -> 3K; the 3 x 3 gather writes [hi | lo], K -> 2K)
 * against weight rows [Whi | Whi | Wlo], so plain 16-bit products over the widened K form hi*Whi + lo*Whi + hi*Wlo.  Same arguments
 * as the plain forms; out16 rows are 2304 (patchify), 3 * dim (cast) or 18 * C (im2col) elements wide. */
int msam_patchify_split16(const float* img, int32_t B, int32_t dtype16, void* out16, void* stream);
int msam_patchify_u8_split16(const uint8_t* img, int32_t B, int32_t h, int32_t w, int32_t dtype16, void* out16, void* stream);
int msam_im2col3x3_split16(const float* x_f32, int32_t B, int32_t C, int32_t dtype16, void* out16, void* stream);
int msam_cast_f32_split16(const float* x, int32_t dtype16, void* out16, int64_t rows, int32_t dim, void* stream);

/* ViT attention with decomposed relative position bias (segment_anything ImageEncoderViT Attention).
 * head_dim = STORED channels per head, 64 or 96 (vit_h: true head_dim 80, zero-padded to 96 by the caller);
 * scale = (true head_dim)^-0.5.
 * q,k,v: bf16 [B,heads,4096,head_dim];
 * rel_h/rel_w: bf16 [2S-1,head_dim]; qkv_bias: fp32 [3*heads*head_dim] (padding tokens of windowed blocks carry
 * bias-only q/k/v);
 * out: bf16 [B*4096, heads*head_dim] token-major. */
int msam_window_attention(const void* q, const void* k, const void* v, const void* rel_h, const void* rel_w,
                          const float* qkv_bias, int32_t B, int32_t heads, int32_t head_dim, float scale, void* out,
                          void* stream);
/* the same with the 16-bit type stated (MSAM_BF16 / MSAM_F16: q, k, v, rel_h, rel_w and out) */
int msam_window_attention16(const void* q, const void* k, const void* v, const void* rel_h, const void* rel_w,
                            const float* qkv_bias, int32_t B, int32_t heads, int32_t head_dim, float scale, int32_t dtype16,
                            void* out, void* stream);
int msam_global_attention(const void* q, const void* k, const void* v, const void* rel_h, const void* rel_w,
                          int32_t B, int32_t heads, int32_t head_dim, float scale, void* out, void* stream);
int msam_global_attention16(const void* q, const void* k, const void* v, const void* rel_h, const void* rel_w, int32_t B,
                            int32_t heads, int32_t head_dim, float scale, int32_t dtype16, void* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Image encoder:  predictor.model.image_encoder(x)   (micro_sam/util.py:674)
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t embed_dim, depth, heads;               /* micro_sam/models/build_sam.py:40-84 */
    int32_t is_global[MSAM_MAX_BLOCKS];
    const void* patch_w; const float* patch_b;     /* bf16 [D,768]; fp32 [D] */
    const float* pos_embed;                        /* fp32 [4096, D] */
    const float* ln1_w[MSAM_MAX_BLOCKS]; const float* ln1_b[MSAM_MAX_BLOCKS];
    const void* qkv_w[MSAM_MAX_BLOCKS]; const float* qkv_b[MSAM_MAX_BLOCKS];
    const void* rel_h[MSAM_MAX_BLOCKS]; const void* rel_w[MSAM_MAX_BLOCKS];      /* bf16 */
    const void* proj_w[MSAM_MAX_BLOCKS]; const float* proj_b[MSAM_MAX_BLOCKS];
    const float* ln2_w[MSAM_MAX_BLOCKS]; const float* ln2_b[MSAM_MAX_BLOCKS];
    const void* lin1_w[MSAM_MAX_BLOCKS]; const float* lin1_b[MSAM_MAX_BLOCKS];
    const void* lin2_w[MSAM_MAX_BLOCKS]; const float* lin2_b[MSAM_MAX_BLOCKS];
    const void* neck0_w;                           /* bf16 [256, D] */
    const float* neck1_w; const float* neck1_b;
    const void* neck2_w;                           /* bf16 [256, 9*256], (ky,kx,c) column order */
    const float* neck3_w; const float* neck3_b;
    int32_t use_glds;
    /* stored channels per head: 0 or embed_dim / heads when that is 64 (vit_b / vit_l); 96 for vit_h, whose 80-channel
     * heads are zero-padded by the caller: qkv_w bf16 [3*heads*96, D] / qkv_b fp32 [3*heads*96] (zero rows), rel_h / rel_w
     * bf16 [2S-1, 96] (zero columns), proj_w bf16 [D, heads*96] (zero columns). */
    int32_t head_dim_stored;
    /* BASELINE config 5: fp8 != 0 runs qkv / proj / lin1 / lin2 of every block on fp8 e4m3 operands (MX MFMA): activations are
     * quantised per token by msam_layernorm_fp8 / msam_quant_rows_fp8, the weights below are e4m3 [N, K] (same shapes as the
     * bf16 ones, incl. the head padding) with one fp32 scale per output channel.  Attention, patch embedding and neck stay bf16. */
    int32_t fp8;
    const void* qkv_w8[MSAM_MAX_BLOCKS]; const float* qkv_cs[MSAM_MAX_BLOCKS];
    const void* proj_w8[MSAM_MAX_BLOCKS]; const float* proj_cs[MSAM_MAX_BLOCKS];
    const void* lin1_w8[MSAM_MAX_BLOCKS]; const float* lin1_cs[MSAM_MAX_BLOCKS];
    const void* lin2_w8[MSAM_MAX_BLOCKS]; const float* lin2_cs[MSAM_MAX_BLOCKS];
    /* 16-bit type of every matrix-product operand and stored activation of the encoder (weights above, rel_h / rel_w, LN outputs,
     * q / k / v, attention output, MLP hidden): 0 or MSAM_BF16 = bfloat16 ("vit_b bf16"), MSAM_F16 = IEEE fp16 - the same kernels
     * on the fp16 MFMAs of the same rate; the caller then hands over fp16 copies of the weights.  Not together with fp8. */
    int32_t dtype16;
    /* split_io != 0: the patch embedding and the two neck convolutions - 1.2 % of the encoder's flops, and the sites whose 8-bit
     * operand rounding costs most of the mask parity against the fp32 reference (profiles/r03_enc_ablation.txt) - take their
     * operands as hi + lo pairs of the 16-bit type (msam_*_split16): patch_w is then [D, 3*768] = [Whi | Whi | Wlo], neck0_w
     * [256, 3*D] and neck2_w [256, 3*2304] likewise (columns (ky,kx,c) within each block).  Same MFMA kernels, K widened. */
    int32_t split_io;
} msam_encoder_t;

int64_t msam_encoder_workspace_bytes(const msam_encoder_t* enc, int32_t B);
/* img: fp32 [B,3,1024,1024] (img_u8 == NULL) or uint8 HWC [B,h,w,3] (img_f32 == NULL);
 * out: fp32 [B,256,64,64].  tap (optional, may be NULL): fp32 [B*4096, D] copy of the residual stream after
 * block `tap_block` (test hook). */
int msam_encoder_forward(const msam_encoder_t* enc, const float* img_f32, const uint8_t* img_u8, int32_t h, int32_t w,
                         int32_t B, float* out, void* workspace, int64_t workspace_bytes,
                         float* tap, int32_t tap_block, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Prompt encoder + mask decoder:  predictor.predict_torch(...)  up to the low-res logits
 * (micro_sam/instance_segmentation.py:361-366, micro_sam/inference.py:248-255)
 * ------------------------------------------------------------------------------------------------- */
typedef struct {
    const void* q_w; const float* q_b; const void* k_w; const float* k_b;
    const void* v_w; const float* v_b; const void* o_w; const float* o_b;   /* weights bf16, biases fp32 */
} msam_attn_w_t;
typedef struct {
    msam_attn_w_t self_attn, t2i, i2t;
    const float *n1_w, *n1_b, *n2_w, *n2_b, *n3_w, *n3_b, *n4_w, *n4_b;
    const void* mlp1_w; const float* mlp1_b; const void* mlp2_w; const float* mlp2_b;
    /* optional (NULL = plain 16-bit operands): the two MLP weights as hi + lo pairs of the decoder's 16-bit type, rows
     * [Whi | Whi | Wlo]: lin1 [2048, 3 * 256], lin2 [256, 3 * 2048].  With both set the MLP's activations enter their products as
     * [hi | lo | hi] rows as well (msam_cast_f32_split16; ~22 significand bits with fp16): the ReLU hidden of the token MLP is the
     * decoder's most rounding-sensitive tensor (profiles/r04_experiments.md section 3). */
    const void* mlp1_ws; const void* mlp2_ws;
} msam_twoway_layer_t;
typedef struct {
    /* prompt encoder */
    const float* pe_gauss;           /* fp32 [2,128] */
    const float* point_embed;        /* fp32 [4,256]: point_embeddings.{0..3} */
    const float* not_a_point;        /* fp32 [256] */
    const float* no_mask;            /* fp32 [256] */
    /* mask decoder */
    const float* out_tokens;         /* fp32 [5,256]: iou_token, mask_tokens[0..3] */
    msam_twoway_layer_t layer[2];
    msam_attn_w_t final_attn; const float *nf_w, *nf_b;
    const void* up1_w; const float* up1_b;     /* bf16 [256 (sub*64+co), 256 (ci)]; fp32 [256] (bias[co] tiled x4) */
    const float* up_ln_w; const float* up_ln_b;
    const void* up2_w; const float* up2_b;     /* bf16 [128 (sub*32+co), 64 (ci)]; fp32 [32] */
    const void* hyp_w[4][3]; const float* hyp_b[4][3];   /* 4 hyper-network MLPs, bf16 / fp32; the last layer
                                                             is zero-padded to 128 output rows */
    const void* iou_w[3]; const float* iou_b[3];           /* IoU head, last layer zero-padded to 128 rows */
    int32_t use_glds;
    int32_t low_res_dtype;           /* type of the `low_res` buffer the msam_decoder_forward_* calls write: 0 / MSAM_F32 = fp32 (the
                                      * predict_torch contract), MSAM_F16 = fp16 (the AMG path: msam_postprocess_masks16 reads it back) */
    int32_t up1_centred;             /* 1: up1_w / up1_b are CENTRED over the 64 output channels of every sub-pixel (rows sub*64 .. sub*64+63 of
                                      * up1_w minus their mean row, up1_b minus its mean) - LayerNorm2d's mean is then zero by construction and the
                                      * up-scaling kernel skips it (msam_upscale_fused_out, keys_blocked bit 1).  0: plain weights. */
} msam_decoder_t;

/* Per-image constants of the decoder (dense positional encoding etc.): computed once per model. */
/* 16-bit type of the decoder weights in msam_decoder_t / msam_upscale_fused / msam_wsgemm_bf16 / ... and of the decoder's
 * 16-bit tensors at the operator-level entry points: MSAM_F16 (default build) or MSAM_BF16 (built with -DMSAM_DEC_F16=0). */
int msam_decoder_dtype(void);
int64_t msam_decoder_const_bytes(void);
int msam_decoder_prepare_const(const msam_decoder_t* dec, void* consts, void* stream);
/* Per-tile image-side precompute: src tokens, layer-0 image projections (prompt independent). */
int64_t msam_decoder_image_bytes(void);
int msam_decoder_prepare_image(const msam_decoder_t* dec, const void* consts, const float* embedding /*[256,64,64]*/,
                               void* image_state, void* workspace, int64_t workspace_bytes, void* stream);
int64_t msam_decoder_workspace_bytes(int32_t P);
/* points: fp32 [P,Np,2] in the 1024-frame (after transform.apply_coords), labels: int32 [P,Np] (1/0/-1);
 * boxes: fp32 [P,4] or NULL.  (Mask prompts: msam_decoder_forward_masks below.)
 * low_res: fp32 [P,C,256,256], iou: fp32 [P,C] with C = 3 (multimask) or 1. */
int msam_decoder_forward(const msam_decoder_t* dec, const void* consts, const void* image_state,
                         const float* points, const int32_t* labels, int32_t Np, const float* boxes, int32_t P,
                         int32_t multimask, float* low_res, float* iou,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* Same with mask prompts (reference: PromptEncoder._embed_masks + MaskDecoder with a per-prompt dense embedding;
 * SamPredictor.predict_torch(mask_input=...), micro_sam/inference.py:248-255 `logits_masks`).  mask_input: fp32
 * [P,1,256,256] low-res logits of a previous prediction (NULL: identical to msam_decoder_forward).  Weights of
 * prompt_encoder.mask_downscaling: Conv2d(1,4,2,2) [4,1,2,2], LayerNorm2d(4), Conv2d(4,16,2,2) [16,4,2,2], LayerNorm2d(16),
 * Conv2d(16,256,1) [256,16]; all fp32 device pointers. */
typedef struct {
    const float *c1_w, *c1_b, *ln1_w, *ln1_b, *c2_w, *c2_b, *ln2_w, *ln2_b, *c3_w, *c3_b;
    int32_t exact_gelu;              /* 1: the two GELUs as 0.5 x (1 + erf(x / sqrt 2)) with the library erff (strict mode); 0: the 5.5e-5 form */
} msam_mask_prompt_t;
int msam_decoder_forward_masks(const msam_decoder_t* dec, const msam_mask_prompt_t* mask_w, const void* consts,
                               const void* image_state, const float* points, const int32_t* labels, int32_t Np,
                               const float* boxes, const float* mask_input, int32_t P, int32_t multimask, float* low_res,
                               float* iou, void* workspace, int64_t workspace_bytes, void* stream);

/* The prompt encoder and the mask decoder as stand-alone module calls (the reference calls them directly:
 * micro_sam/training/trainable_sam.py:96-106 `sam.prompt_encoder(points, boxes, masks)` and
 * `sam.mask_decoder(image_embeddings, image_pe, sparse_prompt_embeddings, dense_prompt_embeddings, multimask_output)`).
 * msam_prompt_encode: sparse fp32 [P, Ns, 256] with Ns = Np + (boxes ? 2 : (Np > 0 ? 1 : 0)) (not written when Ns == 0);
 *   dense fp32 [P, 256, 64, 64] is written only for mask prompts (mask_input fp32 [P,1,256,256]); without masks the dense
 *   embedding is the broadcast no_mask_embed, which the caller builds as a view.
 * msam_decoder_forward_embeddings: sparse fp32 [P, Ns, 256] (0 <= Ns <= 11), dense fp32 [P, 256, 64, 64] or NULL = the
 *   broadcast no_mask_embed already folded into image_state; `embedding` fp32 [256, 64, 64] is needed with a dense
 *   embedding.  The positional encoding is the model's own (msam_decoder_prepare_const). */
int msam_prompt_encode(const msam_decoder_t* dec, const msam_mask_prompt_t* mask_w, const float* points,
                       const int32_t* labels, int32_t Np, const float* boxes, const float* mask_input, int32_t P,
                       float* sparse, float* dense, void* stream);
int msam_decoder_forward_embeddings(const msam_decoder_t* dec, const void* consts, const void* image_state,
                                    const float* sparse, int32_t Ns, const float* dense, const float* embedding,
                                    int32_t P, int32_t multimask, float* low_res, float* iou, void* workspace,
                                    int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Mask post-processing:  Sam.postprocess_masks + AMGBase._to_mask_data
 * (micro_sam/instance_segmentation.py:229-255, micro_sam/_vendored.py:33-152)
 * ------------------------------------------------------------------------------------------------- */
/* low_res fp32 [N,256,256] -> for every mask n (fused, logits are never materialised at full resolution):
 *   bilinear x4 to 1024^2 (align_corners=False), crop to (in_h,in_w), bilinear to (out_h,out_w),
 *   counts[n] = {#(v > thr+off), #(v > thr-off), #(v > thr)}   (stability numerator / denominator, area)
 *   boxes[n]  = xyxy inclusive box of (v > thr), {0,0,0,0} if empty        (batched_mask_to_box)
 *   bits      = bit mask [N, ceil(out_h/32), out_w] uint32: bit b of word [n][yw][x] is pixel (y = yw*32 + b, x)
 * logits (optional, may be NULL): fp32 [N,out_h,out_w] full-resolution values (return_logits path). */
int msam_postprocess_masks(const float* low_res, int32_t N, int32_t in_h, int32_t in_w, int32_t out_h, int32_t out_w,
                           float thr, float off, int32_t* counts, int32_t* boxes, uint32_t* bits, float* logits,
                           void* stream);
/* the same from low-res logits of type low_res_dtype: MSAM_F32 or MSAM_F16 (every value widened to fp32 on load, the arithmetic
 * - and so every integer output - is that of msam_postprocess_masks on the widened values) */
int msam_postprocess_masks16(const void* low_res, int32_t low_res_dtype, int32_t N, int32_t in_h, int32_t in_w, int32_t out_h,
                             int32_t out_w, float thr, float off, int32_t* counts, int32_t* boxes, uint32_t* bits, float* logits,
                             void* stream);
/* uncrop_masks (reference micro_sam/instance_segmentation.py:250, segment_anything.utils.amg.uncrop_masks): place the
 * bit masks of a crop / tile, [N, ceil(crop_h/32), crop_w], at (x0, y0) of full-image bit masks [N, ceil(out_h/32), out_w]
 * (zero outside the crop).  N <= 65535. */
int msam_uncrop_bits(const uint32_t* bits_crop, int32_t N, int32_t crop_h, int32_t crop_w, int32_t x0, int32_t y0,
                     int32_t out_h, int32_t out_w, uint32_t* bits_out, void* stream);

/* Column-major run-length encoding of the bit masks (mask_to_rle_pytorch, _vendored.py:114-152).
 * Pass 1 (run_counts): number of runs per mask incl. the leading zero-run convention.
 * Pass 2 (rle_encode): counts written at offsets[n] (exclusive prefix sum of run counts, int64). */
int msam_rle_run_counts(const uint32_t* bits, int32_t N, int32_t out_h, int32_t out_w, int32_t* n_runs, void* stream);
int msam_rle_encode(const uint32_t* bits, int32_t N, int32_t out_h, int32_t out_w, const int64_t* offsets,
                    int32_t* counts_out, void* stream);

/* Greedy box NMS = torchvision.ops.batched_nms with one category (instance_segmentation.py:126): boxes_sorted fp32
 * [K,4] xyxy in descending score order (stable); keep_flags[i] = 1 if box i survives (IoU > threshold suppresses,
 * areas without +1).  mask_scratch: K * ceil(K/64) uint64. */
int msam_box_nms(const float* boxes_sorted, int32_t K, float iou_threshold, uint64_t* mask_scratch, int32_t* keep_flags,
                 void* stream);

/* Same, with boxes that an earlier filter already rejected: valid_sorted int32 [K] (0 = rejected; such boxes neither
 * survive nor suppress).  Lets generate() run threshold filters + NMS without compacting on the host. */
int msam_box_nms_valid(const float* boxes_sorted, const int32_t* valid_sorted, int32_t K, float iou_threshold,
                       uint64_t* mask_scratch, int32_t* keep_flags, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Merge masks to a label image:  util.mask_data_to_segmentation  (micro_sam/util.py:1773-1848)
 * ------------------------------------------------------------------------------------------------- */
/* label[y][x] = r + 1 for the LAST r in `order` (int32 [K], indices into bits[*]) whose mask covers the pixel, else 0:
 * the reference's area-descending painting where later masks overwrite (merge_exclusively=False).
 * bits: [*, ceil(H/32), W] as produced by msam_postprocess_masks; label: int32 [H,W]. */
int msam_paint_label_image(const uint32_t* bits, const int32_t* order, int32_t K, int32_t H, int32_t W, int32_t* label,
                           void* stream);
/* As msam_paint_label_image with the number of masks read from device memory (k_dev: int32[1]). */
int msam_paint_label_image_dev(const uint32_t* bits, const int32_t* order, const int32_t* k_dev, int32_t H, int32_t W,
                               int32_t* label, void* stream);
/* Connected components (4-connectivity) of equal non-zero value (elf.parallel.label(block_shape=(512, 512)) at util.py:1834-1838):
 * roots[i] = KEY of the root of pixel i's component, -1 for background.  A pixel's key is its position in block-major order
 * (512 x 512 blocks in raster order, raster order inside a block), the root is the component's smallest key; ascending root keys
 * reproduce the reference's component numbering (per-block labels with running offsets, union across faces, consecutive by first
 * occurrence).  For H, W <= 512 key == linear index.  changed_flag: int32 device scratch.
 * Synchronises the stream once per union pass (at most max_iters, default 8); iters_done (host, optional). */
int msam_label_components(const int32_t* seg, int32_t H, int32_t W, int32_t* roots, int32_t* changed_flag,
                          int32_t max_iters, int32_t* iters_done, void* stream);

/* sizes[r] = number of pixels whose root is r (int32 [n], zeroed here), bg_count[0] = number of background pixels
 * (roots[i] < 0): the `unique(..., return_counts=True)` of util.py:1837 keyed by root index. */
int msam_component_sizes(const int32_t* roots, int32_t n, int32_t* sizes, int32_t* bg_count, void* stream);
/* Fully asynchronous variant: `passes` union passes (one is complete for the lock-free union, a second one verifies),
 * changed_flag = flag of the last pass to be checked by the caller whenever convenient. */
int msam_label_components_async(const int32_t* seg, int32_t H, int32_t W, int32_t* roots, int32_t* changed_flag,
                                int32_t passes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Overlap of objects between consecutive slices:  merge_instance_segmentation_3d's edge extraction
 * (micro_sam/multi_dimensional_segmentation.py:356-363 -> elf.tracking compute_edges_from_overlap -> nifty.ground_truth.overlap;
 *  SURVEY.md 8(f) rank 3)
 * ------------------------------------------------------------------------------------------------- */
/* labels: int32 [Z, H, W] with ids consecutive across z.  For every pixel p and z < Z - 1 with a = labels[z][p] != 0 the pair
 * (a, b = labels[z + 1][p]) is counted in an open-addressing table the caller provides (table_keys uint64 [capacity], table_counts
 * int32 [capacity]; capacity a power of two >= 4 x the distinct pairs, initialised here).  edges: int32 [max_edges, 3] = (source,
 * target (0 = background), pixels) in arbitrary order; n_edges: int32 [2] = {edges found, table-overflow flag}. */
int msam_slice_overlaps(const int32_t* labels, int32_t Z, int32_t H, int32_t W, uint64_t* table_keys, int32_t* table_counts,
                        int32_t capacity, int32_t* edges, int32_t max_edges, int32_t* n_edges, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Seeded watershed of InstanceSegmentationWithDecoder.generate (micro_sam/instance_segmentation.py:1083-1168 ->
 * torch_em watershed_from_center_and_boundary_distances -> skimage.segmentation.watershed).  HOST function
 * (host pointers, no stream): the reference's CPU step, restated from scikit-image's published priority flood.
 * image fp32 [H,W], markers int32 [H,W] (0 = none), mask uint8 [H,W] or NULL, out int32 [H,W].
 * ------------------------------------------------------------------------------------------------- */
int msam_host_seeded_watershed(const float* image, const int32_t* markers, const uint8_t* mask, int32_t H, int32_t W, int32_t* out);

/* ---------------------------------------------------------------------------------------------------
 * The "strict" precision mode (micro_sam_amd.strict, Sam.set_precision("strict")): the reference's own formulation of
 * ImageEncoderViT / PromptEncoder / MaskDecoder (segment_anything behind micro_sam/util.py:674 `predictor.model.image_encoder(x)` and
 * micro_sam/instance_segmentation.py:361-366 `predictor.predict_torch(...)`) on fp32 kernels - every tensor fp32, products on the
 * f32-input MFMA (exact fp32 products, fp32 accumulation), erf GELU, expf softmax, IEEE divisions - for callers who need the
 * reference's results to fp32 rounding rather than the 16-bit throughput path.  The host side sequences these building blocks.
 * ------------------------------------------------------------------------------------------------- */
/* out[m][n] = act(sum_k (A[m][k] + A2[m % a2_rows][k]) * W[n][k] + bias[n]) + res[m % res_rows][n]   (torch.nn.functional.linear with
 * the surrounding adds of the reference: `(x + pe) @ W.T + b`, `x + mlp(x)`, `conv(x) + pos_embed`).  All fp32; K, lda, ldw, lda2 % 4
 * == 0, operands 16-byte aligned; A2 / bias / res may be NULL; a2_rows / res_rows 0 = M. */
typedef struct {
    const float* A; int64_t lda;
    const float* A2; int64_t lda2; int64_t a2_rows;
    const float* W; int64_t ldw;
    int64_t M; int32_t N, K;
    const float* bias; int32_t act;      /* MSAM_ACT_NONE / MSAM_ACT_GELU (exact erf form) / MSAM_ACT_RELU, applied before the residual */
    const float* res; int64_t ldr; int64_t res_rows;
    float* out; int64_t ldc;
    /* the convolutional decoder of AIS (models/unetr_hip.py; reference micro_sam/instance_segmentation.py:710-733), channels-last fp32: */
    const float* col_scale; const float* col_shift;    /* both or neither: v = (acc + bias) * scale[n] + shift[n] before the activation
                                                        * (BatchNorm2d on its running statistics: scale = w / sqrt(var + eps), shift = b - mean * scale) */
    int32_t conv_h, conv_w, conv_c;      /* conv_c > 0: A is [B, conv_h, conv_w, conv_c] and the product is the 3 x 3 / padding 1 convolution as an implicit
                                          * GEMM (the loader gathers the taps): M == B conv_h conv_w, K == 9 conv_c, weight columns (ky, kx, c); lda / A2 unused */
    int32_t shuffle_h, shuffle_w, shuffle_c;   /* shuffle_c > 0: ConvTranspose2d(kernel 2, stride 2) store - N == 4 shuffle_c columns (ky*2+kx)*shuffle_c + co of input
                                                * pixel (b, y, x) (M == B shuffle_h shuffle_w) go to out[((b*2H + 2y+ky)*2W + 2x+kx) * ldc + co]; no residual */
    int32_t a2_cols;                     /* > 0 (a multiple of 128): A2 is added for the output columns n < a2_cols only - two projections of one input
                                          * in one launch, `k = (x + pe) Wk^T` | `v = x Wv^T` with W = [Wk; Wv] (the input crosses HBM once); 0: every column */
    int32_t split16;                     /* 0: exact fp32 products on the f32-input MFMA (the strict mode).  1: the "split16" mode - both operands
                                          * enter the 16-bit matrix pipe as fp16 pairs (hi = fp16(x), lo = fp16(x - hi)) and a product is a_hi w_hi + a_hi w_lo
                                          * + a_lo w_hi (fp32 accumulation): fp32-level accuracy (the dropped term is 2^-22 of a product) at 5.3 x the
                                          * f32-input MFMA rate.  Same tiles, loaders and epilogues; everything outside the product stays fp32. */
    float a_scale, w_scale;              /* split16: powers of two (0 = 1.0) applied to A (+ A2) and W before the split so that the values sit inside
                                          * fp16's range (|x| < 65504, subnormals below 6e-5): W scaled to max |w| ~ 2^13, activations usually 1; undone exactly
                                          * in the epilogue */
    const void* w_pairs;                 /* reserved, must be NULL (prepared weight pairs in the product kernel measured 5 - 40 % slower - even the unused branch
                                          * cost the kernel 28 % - and were removed; msam_si2t_t.wq_pairs / wo_pairs remain) */
} msam_sgemm_t;
int msam_strict_gemm(const msam_sgemm_t* p, void* stream);
/* torch.nn.LayerNorm / LayerNorm2d rows: x fp32 [rows, dim <= 1280] -> out fp32 (may be x), optional exact GELU afterwards;
 * out_nchw_hw > 0: output transposed to [rows / hw, dim, hw] (the encoder's NCHW result). */
int msam_strict_layernorm(const float* x, const float* weight, const float* bias, float eps, int64_t rows, int32_t dim, float* out,
                          int32_t gelu, int32_t out_nchw_hw, void* stream);
/* Attention of one ViT block with add_decomposed_rel_pos, from the qkv projection's rows: qkv fp32 [B * grid^2, 3 * heads * head_dim]
 * (q | k | v, heads inside), qkv_bias fp32 [3 * heads * head_dim] (= the q / k / v of the zero-padded window border), rel_h / rel_w fp32
 * [2 S - 1, head_dim] with S = window or grid (already resized to that length), window 14 (any grid <= 64: windows of the zero-padded
 * grid) or 0 (global, grid 64), head_dim 64 or 80; scores = (scale q) . k + q . R_h + q . R_w, softmax, @ v -> out fp32
 * [B * grid^2, heads * head_dim].  Both forms run their two products on the f32-input MFMA (srelpos_mfma_kernel, srelpos_win_mfma_kernel);
 * msam_tune_set("srel_mfma", 1) sends the windows, 0 both forms to the vector-unit kernel (srelpos_kernel). */
int msam_strict_relpos_attention(const float* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, int32_t B, int32_t heads,
                                 int32_t head_dim, int32_t grid, int32_t window, float scale, float* out, void* stream);
/* The same attention in the split16 mode: q . k and p @ v on fp16 operand pairs (msam_sgemm_t.split16; the exponentials are scaled by 2^12 into
 * fp16's normal range before the split), scores / softmax / relative-position terms / the division in fp32 as above. */
int msam_split16_relpos_attention(const float* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, int32_t B, int32_t heads,
                                  int32_t head_dim, int32_t grid, int32_t window, float scale, float* out, void* stream);
/* The "image attends to the tokens" step of one TwoWayAttentionBlock on the per-prompt image stream in ONE launch (segment_anything
 * modeling/transformer.py TwoWayAttentionBlock.forward: q = keys + key_pe; attn_out = cross_attn_image_to_token(q, k, v);
 * keys = norm4(keys + attn_out)), 4096 image tokens x 256 channels per prompt, 8 heads x 16 channels in the attention:
 *   out[b] = LayerNorm(keys[b] + softmax(((keys[b] + pos) Wq^T + bq) . tok_k[b] / denom) tok_v[b] Wo^T + bo)
 * keys fp32 [B][4096][256] (key_batch_stride 0: one [4096][256] stream shared by every prompt - layer 0 without mask prompts; then out must
 * be another buffer), pos [4096][256], wq [128][256], wo [256][128], tok_k / tok_v [B][Tk <= 16][128] (row stride ld_tok, batch stride
 * tok_batch_stride; the token side's k / v projections, computed by msam_strict_gemm), out [B][4096][256] (may be keys when per prompt). */
typedef struct msam_si2t {
    const float* keys; int64_t key_batch_stride;
    const float* pos;
    const float* wq; const float* bq;
    const float* tok_k; const float* tok_v; int64_t ld_tok, tok_batch_stride;
    const float* wo; const float* bo;
    const float* ln_weight; const float* ln_bias; float ln_eps; float denom;
    float* out;
    int32_t B, Tk;
    int32_t split16;                     /* 1: both projections on fp16 operand pairs (msam_sgemm_t.split16); attention, residual, LayerNorm stay fp32 */
    float wq_scale, wo_scale;            /* split16: powers of two for wq / wo (0 = 1), undone on the accumulators */
    const void* wq_pairs; const void* wo_pairs;   /* split16, optional: wq / wo already as fp16 pairs (msam_split16_prepare_pairs with the same scales;
                                                   * wo with permute = 1) - every workgroup then copies the weights' k-tiles instead of re-splitting them */
} msam_si2t_t;
int msam_strict_i2t_block(const msam_si2t_t* p, void* stream);
/* Attention of the two-way transformer: q fp32 [B, Nq, H * D] (row stride ldq, batch stride in floats; 0 = one tensor shared by every
 * batch entry), k / v [B, Nk, H * D], out [B, Nq, H * D]; softmax((q . k) / denom) @ v; D = 16 or 32, Nq <= 16 or Nk <= 16. */
int msam_strict_attention(const float* q, int64_t ldq, int64_t q_batch_stride, const float* k, int64_t ldk, int64_t k_batch_stride,
                          const float* v, int64_t ldv, int64_t v_batch_stride, int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t D,
                          float denom, float* out, int64_t ldo, int64_t out_batch_stride, void* stream);
/* Patch gather: img fp32 [B,3,1024,1024] (after Sam.preprocess) or img_u8 uint8 HWC [B,h,w,3] (Sam.preprocess fused: (x - mean) / std,
 * zero padding) -> fp32 [B * 4096, 768], columns (c, ky, kx). */
int msam_strict_patchify(const float* img, const uint8_t* img_u8, int32_t B, int32_t h, int32_t w, float* out, void* stream);
/* x fp32 [B,64,64,C] -> fp32 [B * 4096, 9 C], columns (ky, kx, c): the 3 x 3 / pad 1 neck convolution as a product. */
int msam_strict_im2col3x3(const float* x, int32_t B, int32_t C, float* out, void* stream);
/* src fp32 [P, 4096, 256] = embedding [256, 4096]^T + dense: dense_stride 0 -> the no_mask_embed vector [256] (P = 1: shared by every
 * prompt), else dense [P, 256, 4096] (mask prompts).  P <= 8191. */
int msam_strict_source(const float* embedding, const float* dense, int64_t dense_stride, int32_t P, float* src, void* stream);
/* masks = hyper_in @ upscaled_embedding, un-shuffled: up fp32 [P * 4096 * 4, 128] (row = (prompt, token, first 2 x 2 sub-pixel), column =
 * second sub-pixel * 32 + channel), hyper fp32 [P, 4, hyper_ld] -> low_res fp32 [P, nmask, 256, 256] of masks mask0 .. mask0 + nmask - 1. */
int msam_strict_hyper_masks(const float* up, const float* hyper, int32_t hyper_ld, int32_t mask0, int32_t nmask, int64_t P, float* low_res,
                            void* stream);
/* "The tokens attend to the image" on a PER-PROMPT image stream in the split16 mode, without materialising the stream's k and v projections
 * (segment_anything modeling/transformer.py Attention.forward with q = the tokens' projection, k = k_proj(keys + key_pe), v = v_proj(keys);
 * TwoWayAttentionBlock.cross_attn_token_to_image of layers >= 1 and final_attn_token_to_image):
 *   out[b, j, h] = softmax_t(q[b, j, h] . ((keys[b, t] + pos[t]) Wk_h^T + bk_h) / denom) ((keys[b, t]) Wv_h^T + bv_h)
 * computed as S = (keys + pos) G^T with G = Wk_h^T q / denom folded on the token side (q . bk is constant over t and cancels), an online
 * softmax over the 4096 image tokens, U = P^T keys and out = Wv_h U + bv_h (sum_t p = 1) - exact algebra; the 4096 x 256 stream of a prompt is
 * read once.  keys fp32 [B][4096][256], pos [4096][256], q fp32 rows (b, j) [B * Tk][>= 128] (8 heads x 16 channels, ALREADY projected: q_proj
 * and its bias applied), wk / wv fp32 [128][256], bv [128]; Tk <= 8; out fp32 rows (b, j) [B * Tk][ldo]; workspace: B x 131072 bytes. */
/* The same step as msam_strict_i2t_block in the split16 mode with BOTH projections folded into the prompt's <= 8 tokens (exact algebra):
 * scores = (keys + pos) . (Wq_h^T k[j, h] / denom) + bq_h . k[j, h] / denom, out = sum_{h j} p (Wo[:, h] v[j, h]) + bo, then + residual, LayerNorm.
 * A prompt's folded operands (128 KB of fp16 pairs) are staged once and its 4096 rows walked by one workgroup, instead of W_q / W_o streaming through
 * LDS for every 128-row block.  Same argument struct (split16 / scales / pairs fields unused); Tk <= 8; workspace: B x 131328 bytes. */
int msam_split16_i2t_block(const msam_si2t_t* p, void* workspace, int64_t workspace_bytes, void* stream);
/* A weight matrix as fp16 pairs in the LDS tile layout of the split16 kernels: w fp32 [N][K] (K % 32 == 0), each value times `scale` (a power of two),
 * hi = fp16(x), lo = fp16(x - hi); out: N rows of K / 32 k-tiles of [32 hi | 32 lo] halves (2 K halves per row).  permute = 1: the k order of
 * msam_strict_i2t_block's second projection inside a k-tile. */
int msam_split16_prepare_pairs(const float* w, int64_t N, int32_t K, float scale, int32_t permute, void* out, void* stream);
typedef struct {
    const float* keys; int64_t key_batch_stride;
    const float* pos;
    const float* q; int64_t ldq;
    const float* wk; const float* wv; const float* bv;
    float denom;
    float* out; int64_t ldo;
    int32_t B, Tk;
    void* workspace; int64_t workspace_bytes;
} msam_st2i_t;
int msam_split16_t2i_attention(const msam_st2i_t* p, void* stream);
/* The second half of MaskDecoder.predict_masks' up-scaling + the hyper product in ONE launch, split16 products (segment_anything
 * modeling/mask_decoder.py: output_upscaling[1:] = LayerNorm2d, GELU, ConvTranspose2d(64 -> 32, k 2, s 2), GELU; masks = hyper_in @ upscaled):
 *   low_res[p][m][4 ty + 2 ky + ky2][4 tx + 2 kx + kx2] = sum_c hyper[p][mask0 + m][c] * GELU(W2 GELU(LayerNorm(u1[row])) + b2)[(ky2, kx2), c]
 * u1 fp32 [P * 4096 * 4, 64]: row (prompt, token (ty, tx), first sub-pixel ky * 2 + kx) of the first transposed convolution (msam_strict_gemm
 * over its [256, 256] weight, BEFORE the LayerNorm); w2 fp32 [128, 64] rows (ky2, kx2, c2), b2 [128]; hyper fp32 [P, 4, hyper_ld].
 * The stream is read once; nothing but the masks is written.  Products on fp16 operand pairs (w_scale: power of two for w2, 0 = 1). */
typedef struct {
    const float* u1;
    const float* ln_weight; const float* ln_bias; float ln_eps;
    const float* w2; const float* b2; float w_scale;
    const float* hyper; int32_t hyper_ld, mask0, nmask;
    float* low_res;
    int64_t P;
} msam_sup2_t;
int msam_strict_upscale2(const msam_sup2_t* p, void* stream);
/* torch.nn.InstanceNorm2d (no affine; torch_em ConvBlock2d) on channels-last data: x fp32 [B, HW, C] with pixel pitch ldx (>= C: a column
 * slice of a wider buffer) -> out fp32 [B, HW, C] dense; C % 4 == 0, C <= 1024.  workspace: 2 B ceil(HW / 2048) C + 2 B C floats. */
int msam_strict_instance_norm(const float* x, int64_t ldx, int32_t B, int64_t HW, int32_t C, float eps, float* out, float* workspace,
                              int64_t workspace_floats, void* stream);
/* torch.nn.functional.interpolate(mode="bilinear", align_corners=False) on channels-last fp32: the h x w window of in [B, pitch_h, pitch_w]
 * pixels of pixel_pitch >= C floats each (the first C are read: a column slice of a wider buffer) -> out [B, H2, W2, C] dense, or NCHW
 * [B, C, H2, W2] with out_nchw (UNETR.postprocess_masks: resize, crop the padding, resize; the x2 up-sampling of torch_em's Upsampler2d).
 * scale_h / scale_w = the source step per output pixel (h / H2, or 1 / scale_factor). */
int msam_strict_resize_bilinear(const float* in, int32_t B, int32_t h, int32_t w, int32_t pitch_h, int32_t pitch_w, int64_t pixel_pitch, int32_t C,
                                int32_t H2, int32_t W2, float scale_h, float scale_w, int32_t out_nchw, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSAM_HIP_H */
