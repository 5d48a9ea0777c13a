// The "strict" precision mode (Sam.set_precision("strict")): the reference's OWN formulation of SAM - no folding, no 16-bit streams -
// on fp32 kernels, for callers who need the north-star tolerance (mask IoU >= 0.999 per instance, identical ids vs the reference CPU
// path) instead of the 16-bit throughput path.  What the reference runs through torch's fp32 CPU operators
// (segment_anything ImageEncoderViT / PromptEncoder / MaskDecoder behind micro_sam/util.py:674 and instance_segmentation.py:361-366)
// runs here as:
//   sgemm_kernel          y = act((A + A2) W^T + b) + R on v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation - bit for bit a
//                         k-ordered fmaf chain (MI355X_MICROARCH.md "FP32-input MFMA"), 1/16 of the bf16 MFMA rate
//   sln_kernel            LayerNorm / LayerNorm2d rows, two-pass statistics, optional exact erf GELU
//   srelpos_kernel        the image encoder's attention (14 x 14 windows with the zero-padded border, or the 64 x 64 grid) with the
//                         decomposed relative position bias, straight from the qkv rows: fp32 VALU, accurate expf
//   sattn_short / _long   the two-way transformer's attentions (<= 16 tokens on one side), fp32 VALU
//   shyper_kernel         masks = hyper_in @ upscaled + the un-shuffle of the two transposed convolutions' 4 x 4 pixel blocks
//   spatchify / sim2col / ssrc   the exact gathers (Sam.preprocess fused for uint8 input: (x - mean) / std with IEEE division)
// GELU is 0.5 x (1 + erf(x / sqrt 2)) with the library erff, exponentials are expf, divisions are IEEE: every step is the
// reference's step to fp32 rounding; what differs from the CPU result is the order of the additions inside a product.
// The host side (micro_sam_amd/strict.py) sequences these calls; cost and measured parity: DESIGN.md section 4.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);
int g_tune_sgemm_bufs = 1;               // msam_tune_set("sgemm_bufs", 1 | 2): LDS stages of sgemm_kernel (1: three workgroups per CU, +9 % on the encoder's shapes)
int g_tune_sgemm_small_below = 512;      // msam_tune_set("sgemm_small_below", n): launches of fewer than n 128 x 128 tiles run on 64 x 64 tiles
int g_tune_si2t_late_us = 0;             // msam_tune_set("si2t_late_us", n): start delay of the second workgroup per CU (0: none; measured: no effect)
int g_tune_si2t_dbg = 0;                 // msam_tune_set("si2t_dbg", bits): timing experiments of si2t_kernel (WRONG results when != 0)
int g_tune_sattn_allh = 0;               // msam_tune_set("sattn_allh", 0 | 1): token -> image attention with one workgroup per prompt (all heads) instead of one per (prompt, head); measured 1221 vs 691 us per 512 prompts: off
int g_tune_srel_mfma = 2;                // msam_tune_set("srel_mfma", 0 | 1 | 2): 1 = global attention on srelpos_mfma_kernel, 2 = the windows on srelpos_win_mfma_kernel as well, 0 = the vector-unit kernel for both

// register budget of a kernel as waves per SIMD (the tests' host build of this file - tests/hip_host_shim.py, g++ - has no such attribute)
#if defined(__HIPCC__)
#define MSAM_WAVES_PER_EU(n_) __attribute__((amdgpu_waves_per_eu(n_, n_)))
#else
#define MSAM_WAVES_PER_EU(n_)
#endif
typedef float f32x16_t __attribute__((ext_vector_type(16)));
MSAM_DEVINL f32x16_t mfma32f(float a, float b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// ---- the "split16" precision mode: an fp32 value x is carried into the 16-bit matrix pipe as the fp16 PAIR hi = fp16(x), lo = fp16(x - hi)
// (x - hi is exact in fp32; hi + lo holds 21 - 22 bits of x) and a product a . w is formed as a_hi w_hi + a_hi w_lo + a_lo w_hi on
// v_mfma_f32_32x32x16_f16 (exact fp16 x fp16 products, fp32 accumulation): the dropped a_lo w_lo term is 2^-22 of the product.  Three MFMAs
// at 16 x the f32-input rate = 5.3 x the strict mode's matrix throughput at fp32-level accuracy (measured against fp64 on the CPU: mean
// relative error 7e-8 against 3e-7 of an fp32 GEMM; DESIGN.md "precision modes").  Range: fp16 holds |x| < 65504 and its subnormals stop at
// 2^-24, so weights are pre-scaled by a power of two (SGemmArgs.w_scale, undone exactly in the epilogue) and activations are taken as they
// are (|x| >= 2^-3 keeps 21 bits; smaller values carry an absolute error of 2^-25).
#if defined(__HIPCC__)
MSAM_DEVINL f32x16_t mfma32h(const uint4& a, const uint4& b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
MSAM_DEVINL float sp_h2f(uint32_t bits16) { return (float)__builtin_bit_cast(_Float16, (u16)bits16); }
#else
static inline f32x16_t mfma32h(const uint4& a, const uint4& b, f32x16_t c) { return mfma32<true>(a, b, c); }
static inline float sp_h2f(uint32_t bits16) { return h16_to_f((u16)bits16); }
#endif
// erf GELU to fp32 rounding without erff's branches: gelu(x) = max(x, 0) - |x| Psi(|x|), Psi(t) = erfc(t / sqrt 2) / 2, with log2 Psi as a degree-8
// minimax polynomial on [0, 9] (weighted by t Psi(t), the resulting GELU error; beyond 9 the term is < 1e-18).  Evaluated in fp32 against fp64:
// max |error| 2.5e-7 (half an ulp of the result at x = 4.5), mean 1.8e-8 - torch's own fp32 GELU: 1.2e-6.  12 instructions instead of ~40.
MSAM_DEVINL float gelu_p8(float x) {
    const float t = fminf(fabsf(x), 9.0f);
    float q = fmaf(-1.6902803281482193e-06f, t, 2.5081630155909806e-05f);
    q = fmaf(q, t, -0.00011445332347648218f); q = fmaf(q, t, -0.000323369400575757f); q = fmaf(q, t, 0.0073334285989403725f);
    q = fmaf(q, t, -0.05271423980593681f); q = fmaf(q, t, -0.4591154158115387f); q = fmaf(q, t, -1.151123285293579f);
    q = fmaf(q, t, -0.9999988675117493f);
    return fmaf(-t, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));
}
// two values at once on packed fp32 arithmetic (v_pk_fma_f32: the vector ALU issues one instruction per 4 cycles and SIMD whatever its width,
// so a VALU-bound kernel wants the packed forms - profiles/r06_mfma_valu_overlap.md)
MSAM_DEVINL f32x2_t gelu_p8x2(f32x2_t x) {
    const f32x2_t r = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    f32x2_t t = r * 2.0f - x;                       // |x|
    t.x = fminf(t.x, 9.0f); t.y = fminf(t.y, 9.0f);
    f32x2_t q = t * -1.6902803281482193e-06f + 2.5081630155909806e-05f;
    q = q * t + -0.00011445332347648218f; q = q * t + -0.000323369400575757f; q = q * t + 0.0073334285989403725f;
    q = q * t + -0.05271423980593681f; q = q * t + -0.4591154158115387f; q = q * t + -1.151123285293579f;
    q = q * t + -0.9999988675117493f;
    const f32x2_t e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    return r - t * e;
}
// split16 kernels: exponentials as ONE v_exp_f32 on scores that already carry log2(e) (the factor is folded into a scale the kernel applies
// anyway).  The argument's fp32 rounding (|score| log2 e <= ~30: 2e-6) is the error of the result, the size of the score's own error in this
// mode; expf (the strict kernels) is ~15 instructions, and these kernels are bound by their vector instruction count.
constexpr float SP_LOG2E = 1.4426950408889634f;
template <bool FAST> MSAM_DEVINL float sp_exp(float x) { return FAST ? __builtin_amdgcn_exp2f(x) : expf(x); }      // FAST: x is in units of ln 2
// (a.lo16 << 16) | b.lo16 and (a.hi16 << 16) | b.hi16 in one instruction
#if defined(__HIPCC__)
MSAM_DEVINL uint32_t sp_pack_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(a, b, 0x05040100u); }
MSAM_DEVINL uint32_t sp_pack_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(a, b, 0x07060302u); }
#else
static inline uint32_t sp_pack_lo(uint32_t a, uint32_t b) { return (a << 16) | (b & 0xffffu); }
static inline uint32_t sp_pack_hi(uint32_t a, uint32_t b) { return (a & 0xffff0000u) | (b >> 16); }
#endif
// ds_read_b64_tr_b16: inside a group of 16 lanes, lane r receives element (r & 3) of the four 16-bit values read by the lanes 4 j + (r >> 2),
// j = 0..3 - a 4 x 16 <-> 16 x 4 transpose of the block the group addressed: MFMA operands that run DOWN the columns of a row-major LDS tile
#if defined(__HIPCC__)
typedef short sp_s16x4_t __attribute__((ext_vector_type(4)));
MSAM_DEVINL uint2 sp_tr16(const unsigned char* p) {
    sp_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((sp_s16x4_t __attribute__((address_space(3)))*)p);
    return __builtin_bit_cast(uint2, v);
}
#else
static inline uint2 sp_tr16(const unsigned char* p) { return ds_read_tr16_b64_emu(p); }
#endif
// four consecutive values -> their hi and lo halves (4 x fp16 = 8 bytes each)
MSAM_DEVINL void sp_split4(const float4& v, float scale, uint2& hi, uint2& lo) {
    const float x0 = v.x * scale, x1 = v.y * scale, x2 = v.z * scale, x3 = v.w * scale;
    hi.x = pack2h(x0, x1); hi.y = pack2h(x2, x3);
    lo.x = pack2h(x0 - sp_h2f(hi.x & 0xffffu), x1 - sp_h2f(hi.x >> 16));
    lo.y = pack2h(x2 - sp_h2f(hi.y & 0xffffu), x3 - sp_h2f(hi.y >> 16));
}
// eight values -> one MFMA operand pair (hi, lo)
MSAM_DEVINL void sp_split8(const float* v, float scale, uint4& hi, uint4& lo) {
    uint2 h0, l0, h1, l1;
    sp_split4(make_float4(v[0], v[1], v[2], v[3]), scale, h0, l0);
    sp_split4(make_float4(v[4], v[5], v[6], v[7]), scale, h1, l1);
    hi = uint4{h0.x, h0.y, h1.x, h1.y}; lo = uint4{l0.x, l0.y, l1.x, l1.y};
}

namespace {

MSAM_DEVINL float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
MSAM_DEVINL float4 ld4(const float* p) { return *(const float4*)p; }
MSAM_DEVINL float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// position of key (0..31) inside a V^T row of the split16 attention kernels: k-step s = key >> 4 takes, from lane half b, the keys
// 16 s + 4 b + {0..3, 8..11} - the keys whose probabilities that lane half holds in registers 8 s .. 8 s + 7 of the S^T accumulator
MSAM_DEVINL int sp_vpos(int key) { return (key & 16) | ((key & 4) << 1) | ((key & 8) >> 1) | (key & 3); }
constexpr float SP_PSCALE = 4096.0f;     // probabilities (<= 1) are scaled into fp16's normal range before the split, undone with the 1 / l

// ------------------------------------------------------------------------------------------------------------------ sgemm
// 128 x 128 output tile, k-tiles of 32, 4 waves x (64 x 64 = 2 x 2 MFMA tiles of 32 x 32), register-staged double buffer, one barrier
// per k-tile.  v_mfma_f32_32x32x2_f32 contracts over k = lane / 32: lane (i, h) feeds elements h * 16 + s of the 32-wide k-tile in step
// s = 0..15 (both operands use the same assignment, so the instruction sums the pair {s, 16 + s} - any assignment that is the same on both
// sides is a valid contraction order); a lane's 16 steps are 64 contiguous bytes of its LDS row = 4 ds_read_b128.  Row pitch 36 floats:
// the 16 lanes of a ds_read_b128 service group land on 16 different 16-byte bank groups.
constexpr int SG_PITCH = 36;
struct SGemmArgs {
    const float* A; long lda; const float* A2; long lda2; long a2_rows;
    const float* W; long ldw; long M; int N, K;
    const float* bias; int act; const float* res; long ldr; long res_rows;
    float* out; long ldc;
    const float* col_scale; const float* col_shift;      // v = v * scale[n] + shift[n] after the bias (BatchNorm2d on running statistics)
    int conv_h, conv_w, conv_c;                          // CONV: A = [B, H, W, C] channels-last, K = 9 C, columns (ky, kx, c): 3 x 3 / pad 1
    int shuf_h, shuf_w, shuf_c;                          // 2 x 2 / stride 2 transposed convolution: column (ky*2+kx)*shuf_c + co of input pixel
    int a2_cols;                                         // A2 only for column tiles n0 < a2_cols (0: all)
    float a_scale, w_scale, out_scale;                   // SPLIT: powers of two applied to A / W before the fp16 split; out_scale = 1 / (a_scale w_scale)
};                                                       // (b, y, x) is stored at output pixel (b, 2y+ky, 2x+kx), channel co

// The epilogue of one wave's 64 x 64 block.  D of a 32 x 32 MFMA tile: lane (col = l & 31, half = l >> 5), register r -> row
// (r & 3) + 8 (r >> 2) + 4 half.  Everything that depends on the row only (bounds, the residual's row, the pixel a transposed convolution
// scatters to) is computed once per row, everything that depends on the column only once per column; the activation is a template
// parameter.  (The first version decided all of it per element - 64 copies of erff, expf and four 64-bit divisions in the instruction
// stream, about 50 000 cycles per wave whatever the shape: profiles/r05_experiments.md section 7.)
template <int ACT>
MSAM_DEVINL float sg_act(float v) {
    if (ACT == MSAM_ACT_GELU) return gelu_exact(v);
    if (ACT == MSAM_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == MSAM_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}
// MODE 0: bias + activation only; 1: + residual; 2: + BatchNorm scale / shift and the transposed convolution's scatter
template <int ACT, int MODE, int IT, bool SPLIT = false>
MSAM_DEVINL void sg_store(const SGemmArgs& a, const f32x16_t (&acc)[IT][IT], long m0, int n0, int wm, int wn, int li, int lh) {
    int n[IT], co[IT];
    long coff[IT];
    float bs[IT], cs[IT], ct[IT];
    const bool scaled = MODE == 2 && a.col_scale != nullptr, shuf = MODE == 2 && a.shuf_c > 0;
    const float* const resp = MODE >= 1 ? a.res : nullptr;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        n[j] = n0 + wn * (32 * IT) + j * 32 + li;
        const int nn = n[j] < a.N ? n[j] : a.N - 1;
        bs[j] = a.bias ? a.bias[nn] : 0.f;
        cs[j] = scaled ? a.col_scale[nn] : 1.f;
        ct[j] = scaled ? a.col_shift[nn] : 0.f;
        const int sub = shuf ? nn / a.shuf_c : 0;
        co[j] = shuf ? nn - sub * a.shuf_c : nn;
        coff[j] = shuf ? ((long)(sub >> 1) * (2L * a.shuf_w) + (sub & 1)) * a.ldc + co[j] : (long)co[j];
    }
    const bool wrap = resp && a.res_rows < a.M;
    const long res0 = wrap ? m0 % a.res_rows : m0;
    const long shw = shuf ? (long)a.shuf_h * a.shuf_w : 1;
    const long sb0 = shuf ? m0 / shw : 0, sp0 = shuf ? m0 - sb0 * shw : 0;
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int loc = wm * (32 * IT) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const long m = m0 + loc;
            if (m >= a.M) continue;
            long rr = res0 + loc;
            if (wrap) while (rr >= a.res_rows) rr -= a.res_rows;
            long row = m;
            if (shuf) {
                long pix = sp0 + loc, b = sb0;
                while (pix >= shw) { pix -= shw; ++b; }
                const int y = (int)pix / a.shuf_w, x = (int)pix - y * a.shuf_w;
                row = (b * 2 * a.shuf_h + 2 * y) * (2L * a.shuf_w) + 2 * x;
            }
            const float* rp = resp ? resp + rr * a.ldr : nullptr;
            float* op = a.out + row * a.ldc;
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                if (n[j] >= a.N) continue;
                float v = (SPLIT ? acc[i][j][r] * a.out_scale : acc[i][j][r]) + bs[j];
                if (scaled) v = v * cs[j] + ct[j];
                v = sg_act<ACT>(v);
                if (rp) v += rp[n[j]];
                op[coff[j]] = v;
            }
        }
}

// CONV: the 3 x 3 gather is done by the tile loader (implicit GEMM): row m = pixel (b, y, x), k = (tap, c) -> x[b][y + tap/3 - 1][x + tap%3 - 1][c],
// zero outside the image (the reference's padding=1) - no im2col matrix in HBM (4.8 GB for the 1024^2 x 128-channel head of the UNETR decoder).
// NBUF 2: two LDS stages, one barrier per k-tile, two workgroups per CU; NBUF 1: one stage, two barriers, three workgroups per CU.
// IT 2: 128 x 128 tile (a wave: 2 x 2 MFMA tiles); IT 1: 64 x 64 tile (a wave: one MFMA tile) for the launches that would not fill the
// chip with 128 x 128 tiles - the token side of the two-way transformer (M = 7 tokens x 128 prompts), the heads, the encoder at batch 1:
// a quarter of the serial MFMA chain per wave, four times the workgroups.  An element's k order is the same in every variant.
// SPLIT: the split16 mode - the same tiles, loaders and epilogues; a row of an LDS k-tile holds 32 hi halves | 32 lo halves (128 of its 144
// bytes) instead of 32 floats, and a 32 x 32 tile's k-tile is 2 k-steps x 3 v_mfma_f32_32x32x16_f16 instead of 16 v_mfma_f32_32x32x2_f32.
template <bool CONV, int NBUF, int IT, bool SPLIT = false>
__global__ __launch_bounds__(256, IT == 1 ? 4 : NBUF == 1 ? 3 : 2) void sgemm_kernel(SGemmArgs a) {
    constexpr int TM = 64 * IT, NJ = TM / 32;
    __shared__ __attribute__((aligned(16))) float As[NBUF][TM * SG_PITCH];
    __shared__ __attribute__((aligned(16))) float Ws[NBUF][TM * SG_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ntn = (a.N + TM - 1) / TM;
    const long bid = blockIdx.x;
    const int tn = (int)(bid % ntn);
    const long m0 = (bid / ntn) * TM;
    const int n0 = tn * TM;
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;
    const float* ap[NJ]; const float* a2p[NJ]; const float* wp[NJ];
    int py[NJ], px[NJ];
    // 64-bit remainders once per block (m0 is uniform), the block's 128 consecutive rows by add-and-wrap
    const long conv_hw = CONV ? (long)a.conv_h * a.conv_w : 1;
    const long pix0 = CONV ? m0 % conv_hw : 0;
    const float* const A2 = (!CONV && a.A2 && (a.a2_cols <= 0 || n0 < a.a2_cols)) ? a.A2 : nullptr;      // block-uniform
    const long a2r0 = A2 ? m0 % a.a2_rows : 0;
    const long mlast = a.M - 1 - m0;                    // rows past M re-read the last row (their results are not stored)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        long loc = srow + 32 * j;
        if (loc > mlast) loc = mlast;
        const long m = m0 + loc;
        if (CONV) {
            long pix = pix0 + loc;
            while (pix >= conv_hw) pix -= conv_hw;
            py[j] = (int)pix / a.conv_w; px[j] = (int)pix - py[j] * a.conv_w;
            ap[j] = a.A + m * a.lda;                     // lda = pixel pitch (>= conv_c: the input may be a column slice of a wider buffer)
        } else {
            py[j] = px[j] = 0;
            ap[j] = a.A + m * a.lda + sc4;
        }
        a2p[j] = nullptr;
        if (A2) {
            long r2 = a2r0 + loc;
            while (r2 >= a.a2_rows) r2 -= a.a2_rows;
            a2p[j] = A2 + r2 * a.lda2 + sc4;
        }
        int n = n0 + srow + 32 * j;
        if (n >= a.N) n = a.N - 1;
        wp[j] = a.W + (long)n * a.ldw + sc4;
    }
    float4 ra[NJ], rw[NJ];
    auto gload = [&](int k0) {
        const bool in = k0 + sc4 < a.K;
        int dy = 0, dx = 0, coff = 0;
        if (CONV && in) {
            const int k = k0 + sc4, tap = k / a.conv_c;
            dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1;
            coff = (dy * a.conv_w + dx) * (int)a.lda + (k - tap * a.conv_c);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float4 v = zero4(), u = zero4();
            if (in) {
                if (CONV) {
                    if ((unsigned)(py[j] + dy) < (unsigned)a.conv_h && (unsigned)(px[j] + dx) < (unsigned)a.conv_w) v = ld4(ap[j] + coff);
                } else {
                    v = ld4(ap[j] + k0);
                    if (A2) { const float4 t = ld4(a2p[j] + k0); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                }
                u = ld4(wp[j] + k0);
            }
            ra[j] = v; rw[j] = u;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (SPLIT) {
                uint2 h, l;
                char* const pa_ = (char*)&As[buf][(srow + 32 * j) * SG_PITCH] + sc4 * 2;
                char* const pw_ = (char*)&Ws[buf][(srow + 32 * j) * SG_PITCH] + sc4 * 2;
                sp_split4(ra[j], a.a_scale, h, l);
                *(uint2*)pa_ = h; *(uint2*)(pa_ + 64) = l;
                sp_split4(rw[j], a.w_scale, h, l);
                *(uint2*)pw_ = h; *(uint2*)(pw_ + 64) = l;
            } else {
                *(float4*)&As[buf][(srow + 32 * j) * SG_PITCH + sc4] = ra[j];
                *(float4*)&Ws[buf][(srow + 32 * j) * SG_PITCH + sc4] = rw[j];
            }
        }
    };
    f32x16_t acc[IT][IT];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < IT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
    const int nk = (a.K + 31) / 32;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload((kt + 1) * 32);
        const float* pa = &As[kt & (NBUF - 1)][(wm * (TM / 2) + li) * SG_PITCH + lh * (SPLIT ? 4 : 16)];
        const float* pw = &Ws[kt & (NBUF - 1)][(wn * (TM / 2) + li) * SG_PITCH + lh * (SPLIT ? 4 : 16)];
        if constexpr (SPLIT) {
            // lane (row li, half lh) of a 32 x 32 x 16 step ks: k = 16 ks + 8 lh .. + 7 = 16 bytes at 32 ks + 16 lh of the row's hi (lo: + 64) block
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 ah[IT], al[IT], wh[IT], wl[IT];
#pragma unroll
                for (int i = 0; i < IT; ++i) {
                    const char* qa = (const char*)(pa + i * 32 * SG_PITCH) + ks * 32;
                    const char* qw = (const char*)(pw + i * 32 * SG_PITCH) + ks * 32;
                    ah[i] = *(const uint4*)qa; al[i] = *(const uint4*)(qa + 64);
                    wh[i] = *(const uint4*)qw; wl[i] = *(const uint4*)(qw + 64);
                }
#pragma unroll
                for (int i = 0; i < IT; ++i)
#pragma unroll
                    for (int j = 0; j < IT; ++j) {
                        acc[i][j] = mfma32h(al[i], wh[j], acc[i][j]);
                        acc[i][j] = mfma32h(ah[i], wl[j], acc[i][j]);
                        acc[i][j] = mfma32h(ah[i], wh[j], acc[i][j]);
                    }
            }
        } else
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float av[IT][4], bv[IT][4];
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                const float4 t = ld4(pa + i * 32 * SG_PITCH + s4 * 4), u = ld4(pw + i * 32 * SG_PITCH + s4 * 4);
                av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
                bv[i][0] = u.x; bv[i][1] = u.y; bv[i][2] = u.z; bv[i][3] = u.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < IT; ++i)
#pragma unroll
                    for (int j = 0; j < IT; ++j) acc[i][j] = mfma32f(av[i][e], bv[j][e], acc[i][j]);
        }
        if (NBUF == 1) __syncthreads();
        if (kt + 1 < nk) sstore((kt + 1) & (NBUF - 1));
        __syncthreads();
    }
    const int mode = (a.col_scale || a.shuf_c > 0) ? 2 : a.res ? 1 : 0;
#define MSAM_SG_STORE(ACT_)                                                                                                           \
    do {                                                                                                                              \
        if (mode == 0) sg_store<ACT_, 0, IT, SPLIT>(a, acc, m0, n0, wm, wn, li, lh);                                                      \
        else if (mode == 1) sg_store<ACT_, 1, IT, SPLIT>(a, acc, m0, n0, wm, wn, li, lh);                                                 \
        else sg_store<ACT_, 2, IT, SPLIT>(a, acc, m0, n0, wm, wn, li, lh);                                                                \
    } while (0)
    switch (a.act) {
        case MSAM_ACT_GELU: MSAM_SG_STORE(MSAM_ACT_GELU); break;
        case MSAM_ACT_RELU: MSAM_SG_STORE(MSAM_ACT_RELU); break;
        case MSAM_ACT_SIGMOID: MSAM_SG_STORE(MSAM_ACT_SIGMOID); break;
        default: MSAM_SG_STORE(MSAM_ACT_NONE); break;
    }
#undef MSAM_SG_STORE
}

// ------------------------------------------------------------------------------------------------------------------ LayerNorm
// one wave per row (any dim <= 1280); two-pass statistics as torch.nn.LayerNorm / upstream LayerNorm2d
__global__ __launch_bounds__(256) void sln_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                  float eps, long rows, int dim, float* __restrict__ out, int gelu, int nchw_hw) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * dim;
    float v[20];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        const int c = i * 64 + lane;
        v[i] = c < dim ? xr[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum64(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 20; ++i) { const int c = i * 64 + lane; const float d = c < dim ? v[i] - mean : 0.f; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        const int c = i * 64 + lane;
        if (c < dim) {
            float y = (v[i] - mean) * rstd * w[c] + b[c];
            if (gelu) y = gelu_exact(y);
            if (nchw_hw > 0) {
                const long bimg = row / nchw_hw, t = row - bimg * nchw_hw;
                out[(bimg * dim + c) * (long)nchw_hw + t] = y;
            } else out[row * dim + c] = y;
        }
    }
}
// dim == 64 (LayerNorm2d of the up-scaling: 16.7 M rows per 1024 prompts): 16 lanes per row, float4 per lane
__global__ __launch_bounds__(256) void sln64_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                    float eps, long rows, float* __restrict__ out, int gelu) {
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c = (threadIdx.x & 15) * 4;
    if (row >= rows) return;
    const float4 t = ld4(x + row * 64 + c);
    const float mean = wave_sum_xor16((t.x + t.y) + (t.z + t.w)) * (1.0f / 64.0f);
    const float d0 = t.x - mean, d1 = t.y - mean, d2 = t.z - mean, d3 = t.w - mean;
    const float var = wave_sum_xor16((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / 64.0f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4 ww = ld4(w + c), bb = ld4(b + c);
    float y0 = d0 * rstd * ww.x + bb.x, y1 = d1 * rstd * ww.y + bb.y, y2 = d2 * rstd * ww.z + bb.z, y3 = d3 * rstd * ww.w + bb.w;
    if (gelu) { y0 = gelu_exact(y0); y1 = gelu_exact(y1); y2 = gelu_exact(y2); y3 = gelu_exact(y3); }
    *(float4*)(out + row * 64 + c) = make_float4(y0, y1, y2, y3);
}

// ------------------------------------------------------------------------------------------------------------------ encoder attention
// One thread per query, one workgroup per (image, window, head, block of 256 queries).  The query's two bias rows q . R_h[qh - kh],
// q . R_w[qw - kw] (UNSCALED query, upstream add_decomposed_rel_pos) go to LDS once ([S][256] each, the thread's own column), keys and
// values pass through LDS in chunks of 16 (read as broadcasts), softmax online with expf.  Tokens of a window that lie outside the
// G x G grid are the zero padding the reference applies AFTER norm1: their q / k / v rows are the qkv bias.
struct SRelArgs {
    const float* qkv; const float* bqkv; const float* rel_h; const float* rel_w; float* out;
    int B, heads, G, window, Dm; float scale;
};
constexpr int SR_KC = 16;

template <int HD, int S>
__global__ __launch_bounds__(256) void srelpos_kernel(SRelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sr_lds[];
    float* const bh = sr_lds;                       // [S][256]
    float* const bw = sr_lds + S * 256;             // [S][256]
    float* const kv = sr_lds + 2 * S * 256;         // [2][KC][HD]
    constexpr int S2 = S * S, QB = (S2 + 255) / 256;
    const int tid = threadIdx.x;
    const int nW = a.window ? (a.G + S - 1) / S : 1, nwin = nW * nW;
    int bid = blockIdx.x;
    const int qb = bid % QB; bid /= QB;
    const int h = bid % a.heads; bid /= a.heads;
    const int win = bid % nwin;
    const int b = bid / nwin;
    const int wy = win / nW, wx = win % nW;
    const int T = a.G * a.G;
    const long ld = 3L * a.Dm;
    auto row_ptr = [&](int gi, int which) -> const float* {        // grid index inside the window -> the token's q / k / v slice
        const int ty = wy * S + gi / S, tx = wx * S + gi % S;
        if (ty >= a.G || tx >= a.G) return a.bqkv + (long)which * a.Dm + h * HD;
        return a.qkv + ((long)b * T + ty * a.G + tx) * ld + (long)which * a.Dm + h * HD;
    };
    const int qi = qb * 256 + tid;
    const bool valid = qi < S2;
    const int qic = valid ? qi : S2 - 1;
    const int qh = qic / S, qw = qic % S;
    float q[HD], acc[HD];
    {
        const float* qp = row_ptr(qic, 0);
#pragma unroll
        for (int d = 0; d < HD; d += 4) { const float4 t = ld4(qp + d); q[d] = t.x; q[d + 1] = t.y; q[d + 2] = t.z; q[d + 3] = t.w; }
    }
    for (int r = 0; r < S; ++r) {
        const float* rh = a.rel_h + (long)(qh - r + S - 1) * HD;
        const float* rw = a.rel_w + (long)(qw - r + S - 1) * HD;
        float sh = 0.f, sw = 0.f;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const float4 u = ld4(rh + d), v = ld4(rw + d);
            sh = fmaf(q[d], u.x, sh); sh = fmaf(q[d + 1], u.y, sh); sh = fmaf(q[d + 2], u.z, sh); sh = fmaf(q[d + 3], u.w, sh);
            sw = fmaf(q[d], v.x, sw); sw = fmaf(q[d + 1], v.y, sw); sw = fmaf(q[d + 2], v.z, sw); sw = fmaf(q[d + 3], v.w, sw);
        }
        bh[r * 256 + tid] = sh; bw[r * 256 + tid] = sw;
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] *= a.scale; acc[d] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    constexpr int V4 = HD / 4;
    for (int j0 = 0; j0 < S2; j0 += SR_KC) {
        __syncthreads();
        for (int idx = tid; idx < 2 * SR_KC * V4; idx += 256) {
            const int which = idx / (SR_KC * V4), rem = idx % (SR_KC * V4), kk = rem / V4, c4 = rem % V4;
            const int j = j0 + kk;
            float4 t = zero4();
            if (j < S2) t = ld4(row_ptr(j, 1 + which) + c4 * 4);
            *(float4*)&kv[(which * SR_KC + kk) * HD + c4 * 4] = t;
        }
        __syncthreads();
        if (!valid) continue;
        // eight keys per online-softmax step (the 16-key chunk in two steps: the fully unrolled 16-key form spilled registers)
#pragma unroll 1
        for (int k0 = 0; k0 < SR_KC; k0 += 8) {
            float s[8];
            float cm = -3.0e38f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int j = j0 + k0 + kk;
                const float* kp = &kv[(k0 + kk) * HD];
                float dot = 0.f;
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 t = ld4(kp + d);
                    dot = fmaf(q[d], t.x, dot); dot = fmaf(q[d + 1], t.y, dot); dot = fmaf(q[d + 2], t.z, dot); dot = fmaf(q[d + 3], t.w, dot);
                }
                const int jc = j < S2 ? j : S2 - 1;
                s[kk] = j < S2 ? (dot + bh[(jc / S) * 256 + tid]) + bw[(jc % S) * 256 + tid] : -3.0e38f;
                cm = fmaxf(cm, s[kk]);
            }
            const float mn = fmaxf(m, cm), alpha = expf(m - mn);
            l *= alpha;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] *= alpha;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float p = j0 + k0 + kk < S2 ? expf(s[kk] - mn) : 0.f;
                l += p;
                const float* vp = &kv[(SR_KC + k0 + kk) * HD];
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    const float4 t = ld4(vp + d);
                    acc[d] = fmaf(p, t.x, acc[d]); acc[d + 1] = fmaf(p, t.y, acc[d + 1]);
                    acc[d + 2] = fmaf(p, t.z, acc[d + 2]); acc[d + 3] = fmaf(p, t.w, acc[d + 3]);
                }
            }
            m = mn;
        }
    }
    if (!valid) return;
    const int ty = wy * S + qh, tx = wx * S + qw;
    if (ty >= a.G || tx >= a.G) return;                              // padded query rows are dropped by window_unpartition
    float* op = a.out + ((long)b * T + ty * a.G + tx) * a.Dm + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) *(float4*)(op + d) = make_float4(acc[d] / l, acc[d + 1] / l, acc[d + 2] / l, acc[d + 3] / l);
}

// ------------------------------------------------------------------------------------------------------------------ global attention, MFMA
// The 64 x 64 grid's attention (4096 keys per query, 51.5 GFLOP per block of vit_b and image) on v_mfma_f32_32x32x2_f32 in the
// TRANSPOSED orientation: S^T = K Qs^T and O^T = V^T P^T, so that in both D layouts a LANE is a QUERY (column = lane & 31) and the
// registers run over keys (S^T) or head channels (O^T):
//   * the softmax of a query is register-local - one exchange with lane ^ 32 per tile for the common running maximum; each half keeps
//     the partial sum of its own 16 keys per tile, added once at the end;
//   * the exponentials ARE the B operand of the second product: step s of v_mfma_32x32x2 contracts the key pair {key(s, 0), key(s, 1)}
//     with key(s, h) = (s & 3) + 8 (s >> 2) + 4 h - exactly the key that register s of lane (q, h) holds in the D layout of S^T.  Any
//     pairing is a valid contraction as long as the A operand agrees: lane (d, h) reads V^T[d][key(s, h)], four consecutive keys per
//     ds_read_b128 from the tile stored channel-major.
// One workgroup = 128 queries (two grid rows) of one (image, head), one wave = 32 queries of one grid row; K / V tiles of 32 keys
// (one half grid row) pass through LDS once per workgroup.  Decomposed relative positions (upstream add_decomposed_rel_pos, UNSCALED
// query): the column term q . R_w[qw - kw + 63] of the wave's queries against the 95 table rows they can meet is three MFMA tiles
// before the loop, scattered to LDS as bw[query][kw]; the row term q . R_h[qh - kh + 63] is one value per query and key tile, recomputed
// on the vector unit whenever kh changes (every second tile) from the table row, which is wave-uniform.
// score = ((q * scale) . k + rel_h) + rel_w as upstream; expf; IEEE division by the sum.
constexpr int SM_BWP = 68;                      // bw row pitch (floats)
constexpr int SM_VP = 36;                       // V^T row pitch: 32 keys + 4
// SPLIT (the split16 mode): both products on fp16 operand pairs - K rows and V^T rows sit in LDS as hi | lo halves (same pitches), the
// query's halves and the exponentials (scaled by 2^12) are split in registers; 12 + 12 MFMAs of the 16-bit pipe per key tile and head-dim 64
// instead of 32 + 32 f32-input ones.  Scores, softmax, relative-position terms and the final division: the same fp32 code.
template <int HD, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void srelpos_mfma_kernel(SRelArgs a) {
    constexpr int S = 64, T = S * S, HH = HD / 2, DT = (HD + 31) / 32, KP = HD + 4, V4 = HD / 4;
    extern __shared__ __attribute__((aligned(16))) float sm_lds[];
    float* const bw = sm_lds;                           // [128][SM_BWP]
    float* const Ks = bw + 128 * SM_BWP;                // [32][KP]
    float* const Vt = Ks + 32 * KP;                     // [DT * 32][SM_VP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    // workgroups of one (image, head) run on one XCD (round-robin dispatch): they share that head's K / V in its L2
    const int nb = gridDim.x, per = nb >> 3;
    const int bid = (nb & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    const int qblk = bid & 31;
    const int h = (bid >> 5) % a.heads, b = (bid >> 5) / a.heads;
    const long ld = 3L * a.Dm;
    const float* const base = a.qkv + (long)b * T * ld + h * HD;
    const int qh = qblk * 2 + (w >> 1), qw0 = (w & 1) * 32, tq = qh * S + qw0 + li;
    // the lane's half of its query: channels lh * HH + s
    float qu[HH], qs[HH];
    {
        const float* qp = base + (long)tq * ld + lh * HH;
#pragma unroll
        for (int d = 0; d < HH; d += 4) { const float4 t = ld4(qp + d); qu[d] = t.x; qu[d + 1] = t.y; qu[d + 2] = t.z; qu[d + 3] = t.w; }
    }
    // column term: G^T[jj][q] = R_w[qw0 + jj] . q for jj < 96, kept at bw[q][kw = li + 63 - jj]
#pragma unroll 1
    for (int jt = 0; jt < 3; ++jt) {
        int j = qw0 + jt * 32 + li;
        if (j > 2 * S - 2) j = 2 * S - 2;
        const float* rp = a.rel_w + (long)j * HD + lh * HH;
        f32x16_t g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
        for (int d = 0; d < HH; d += 4) {
            const float4 t = ld4(rp + d);
            g = mfma32f(t.x, qu[d], g); g = mfma32f(t.y, qu[d + 1], g); g = mfma32f(t.z, qu[d + 2], g); g = mfma32f(t.w, qu[d + 3], g);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kw = li + 63 - (jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
            if (kw >= 0 && kw < S) bw[(w * 32 + li) * SM_BWP + kw] = SPLIT ? g[r] * SP_LOG2E : g[r];      // (split16: every score term in units of ln 2)
        }
    }
#pragma unroll
    for (int d = 0; d < HH; ++d) qs[d] = qu[d] * (SPLIT ? a.scale * SP_LOG2E : a.scale);
    constexpr int NKS = HH / 8;                          // SPLIT: k-steps of the S^T product (a lane half's HH channels, eight at a time)
    uint4 qsh[SPLIT ? NKS : 1], qsl[SPLIT ? NKS : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int s8 = 0; s8 < NKS; ++s8) sp_split8(&qs[8 * s8], 1.0f, qsh[s8], qsl[s8]);
    }
    for (int i = tid; i < DT * 32 * SM_VP; i += 256) Vt[i] = 0.f;       // HD = 80: channels 80..95 of the third channel tile stay zero
    f32x16_t oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m = -3.0e38f, l = 0.f, bhv = 0.f;
    // tile loader: K rows as they are, V transposed
    constexpr int NLD = (32 * V4 + 255) / 256;
    float4 rk[NLD], rv[NLD];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 32 * V4) {
                const float* kp = base + (long)(t * 32 + idx / V4) * ld + a.Dm + (idx % V4) * 4;
                rk[i] = ld4(kp); rv[i] = ld4(kp + a.Dm);
            }
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i;
            if (idx < 32 * V4) {
                const int row = idx / V4, c = (idx % V4) * 4;
                if constexpr (SPLIT) {
                    uint2 h, l;
                    sp_split4(rk[i], 1.0f, h, l);
                    char* const pk = (char*)Ks + row * (KP * 4) + c * 2;
                    *(uint2*)pk = h; *(uint2*)(pk + HD * 2) = l;
                    sp_split4(rv[i], 1.0f, h, l);
                    u16* const pv = (u16*)((char*)Vt + c * (SM_VP * 4)) + sp_vpos(row);
                    pv[0] = (u16)(h.x & 0xffffu); pv[32] = (u16)(l.x & 0xffffu);
                    pv[SM_VP * 2] = (u16)(h.x >> 16); pv[SM_VP * 2 + 32] = (u16)(l.x >> 16);
                    pv[SM_VP * 4] = (u16)(h.y & 0xffffu); pv[SM_VP * 4 + 32] = (u16)(l.y & 0xffffu);
                    pv[SM_VP * 6] = (u16)(h.y >> 16); pv[SM_VP * 6 + 32] = (u16)(l.y >> 16);
                } else {
                    *(float4*)&Ks[row * KP + c] = rk[i];
                    Vt[(c + 0) * SM_VP + row] = rv[i].x; Vt[(c + 1) * SM_VP + row] = rv[i].y;
                    Vt[(c + 2) * SM_VP + row] = rv[i].z; Vt[(c + 3) * SM_VP + row] = rv[i].w;
                }
            }
        }
    };
    gload(0);
    __syncthreads();                                    // the zero fill of Vt
    sstore();
    __syncthreads();
    const float* const bwq = bw + (w * 32 + li) * SM_BWP + 4 * lh;
#pragma unroll 1
    for (int t = 0; t < T / 32; ++t) {
        if (t + 1 < T / 32) gload(t + 1);
        // S^T tile: 32 keys x the wave's 32 queries
        f32x16_t sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        const float* kp = Ks + li * KP + lh * HH;
        if constexpr (SPLIT) {
            const char* kb = (const char*)Ks + li * (KP * 4) + lh * (HH * 2);
#pragma unroll
            for (int s8 = 0; s8 < NKS; ++s8) {
                const uint4 kh = *(const uint4*)(kb + 16 * s8), kl = *(const uint4*)(kb + HD * 2 + 16 * s8);
                sc = mfma32h(kl, qsh[s8], sc); sc = mfma32h(kh, qsl[s8], sc); sc = mfma32h(kh, qsh[s8], sc);
            }
            if ((t & 1) == 0) {
                const float* rp = a.rel_h + (long)(qh - (t >> 1) + S - 1) * HD + lh * HH;
                float part = 0.f;
#pragma unroll
                for (int d = 0; d < HH; d += 4) {
                    const float4 u = ld4(rp + d);
                    part = fmaf(qu[d], u.x, part); part = fmaf(qu[d + 1], u.y, part); part = fmaf(qu[d + 2], u.z, part); part = fmaf(qu[d + 3], u.w, part);
                }
                bhv = (part + __shfl_xor(part, 32)) * SP_LOG2E;
            }
        } else
        if ((t & 1) == 0) {                             // new key row kh = t / 2: the row term, under the MFMAs of this tile
            const float* rp = a.rel_h + (long)(qh - (t >> 1) + S - 1) * HD + lh * HH;
            float part = 0.f;
#pragma unroll
            for (int d = 0; d < HH; d += 4) {
                const float4 u = ld4(rp + d);
                const float4 kk = ld4(kp + d);
                sc = mfma32f(kk.x, qs[d], sc); sc = mfma32f(kk.y, qs[d + 1], sc); sc = mfma32f(kk.z, qs[d + 2], sc); sc = mfma32f(kk.w, qs[d + 3], sc);
                part = fmaf(qu[d], u.x, part); part = fmaf(qu[d + 1], u.y, part); part = fmaf(qu[d + 2], u.z, part); part = fmaf(qu[d + 3], u.w, part);
            }
            bhv = part + __shfl_xor(part, 32);
        } else {
#pragma unroll
            for (int d = 0; d < HH; d += 4) {
                const float4 kk = ld4(kp + d);
                sc = mfma32f(kk.x, qs[d], sc); sc = mfma32f(kk.y, qs[d + 1], sc); sc = mfma32f(kk.z, qs[d + 2], sc); sc = mfma32f(kk.w, qs[d + 3], sc);
            }
        }
        // + rel_h + rel_w, running maximum (register r of lane (q, lh) = key (r & 3) + 8 (r >> 2) + 4 lh of the tile)
        const float* bp = bwq + (t & 1) * 32;
        float cm = -3.0e38f;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 u = ld4(bp + 8 * g4);
            sc[4 * g4] = (sc[4 * g4] + bhv) + u.x; sc[4 * g4 + 1] = (sc[4 * g4 + 1] + bhv) + u.y;
            sc[4 * g4 + 2] = (sc[4 * g4 + 2] + bhv) + u.z; sc[4 * g4 + 3] = (sc[4 * g4 + 3] + bhv) + u.w;
            cm = fmaxf(fmaxf(cm, fmaxf(sc[4 * g4], sc[4 * g4 + 1])), fmaxf(sc[4 * g4 + 2], sc[4 * g4 + 3]));
        }
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);
        if (__ballot(mn > m)) {
            const float alpha = sp_exp<SPLIT>(m - mn);
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            m = mn;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = sp_exp<SPLIT>(sc[r] - m); l += sc[r]; }
        // O^T += V^T P^T
        if constexpr (SPLIT) {
            uint4 ph[2], pl[2];
            {
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[r] = sc[r];
                sp_split8(&pv[0], SP_PSCALE, ph[0], pl[0]); sp_split8(&pv[8], SP_PSCALE, ph[1], pl[1]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* vb = (const char*)Vt + (dt * 32 + li) * (SM_VP * 4) + 16 * lh;
#pragma unroll
                for (int s8 = 0; s8 < 2; ++s8) {
                    const uint4 vh = *(const uint4*)(vb + 32 * s8), vl = *(const uint4*)(vb + 64 + 32 * s8);
                    oacc[dt] = mfma32h(vl, ph[s8], oacc[dt]); oacc[dt] = mfma32h(vh, pl[s8], oacc[dt]); oacc[dt] = mfma32h(vh, ph[s8], oacc[dt]);
                }
            }
        } else
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const float* vp = Vt + (dt * 32 + li) * SM_VP + 4 * lh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v = ld4(vp + 8 * g4);
                oacc[dt] = mfma32f(v.x, sc[4 * g4], oacc[dt]); oacc[dt] = mfma32f(v.y, sc[4 * g4 + 1], oacc[dt]);
                oacc[dt] = mfma32f(v.z, sc[4 * g4 + 2], oacc[dt]); oacc[dt] = mfma32f(v.w, sc[4 * g4 + 3], oacc[dt]);
            }
        }
        __syncthreads();
        if (t + 1 < T / 32) sstore();
        __syncthreads();
    }
    l += __shfl_xor(l, 32);
    if constexpr (SPLIT) l *= SP_PSCALE;
    float* op = a.out + ((long)b * T + tq) * a.Dm + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * lh;
            if (d < HD) *(float4*)(op + d) = make_float4(oacc[dt][4 * g4] / l, oacc[dt][4 * g4 + 1] / l, oacc[dt][4 * g4 + 2] / l, oacc[dt][4 * g4 + 3] / l);
        }
}

// ------------------------------------------------------------------------------------------------------------------ window attention, MFMA
// The same transposed formulation for the 14 x 14 windows (196 tokens; windows of the zero-padded grid: tokens outside the grid are the
// qkv bias, as in srelpos_kernel).  One workgroup of 7 waves = one (image, window, head): wave w owns queries 32 w .. 32 w + 31 (the last
// wave: 4 of them), the 196 keys pass LDS in 7 tiles of 32 (the last one masked from key 196 on).  Decomposed relative positions: a
// query meets 14 row and 14 column offsets, so both terms are ONE MFMA tile each before the loop (the 27 table rows against the wave's
// queries), scattered to LDS as bh[query][kh] | bw[query][kw]; a score reads its two entries by the key's (kh, kw).
constexpr int SW_S = 14, SW_T = SW_S * SW_S, SW_KT = (SW_T + 31) / 32;      // 196 tokens, 7 tiles
template <int HD, bool SPLIT = false>
// (head_dim 64: 128 registers = four waves per SIMD = two of these 7-wave workgroups per CU, at the price of 5 spilled dwords in the prologue)
__global__ __launch_bounds__(448) MSAM_WAVES_PER_EU(HD == 64 ? 4 : 3) void srelpos_win_mfma_kernel(SRelArgs a) {
    constexpr int HH = HD / 2, DT = (HD + 31) / 32, KP = HD + 4, V4 = HD / 4;
    extern __shared__ __attribute__((aligned(16))) float sw_lds[];
    float* const bt = sw_lds;                           // [224][32]: bh (16) | bw (16) per query
    float* const Ks = bt + 224 * 32;                    // [32][KP]
    float* const Vt = Ks + 32 * KP;                     // [DT * 32][SM_VP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int nW = (a.G + SW_S - 1) / SW_S, nwin = nW * nW;
    int bid = blockIdx.x;
    const int h = bid % a.heads; bid /= a.heads;
    const int win = bid % nwin, b = bid / nwin;
    const int wy = win / nW, wx = win % nW;
    const int T = a.G * a.G;
    const long ld = 3L * a.Dm;
    auto row_ptr = [&](int gi, int which) -> const float* {        // token index inside the window -> its q / k / v slice (or the bias: padding)
        const int ty = wy * SW_S + gi / SW_S, tx = wx * SW_S + gi % SW_S;
        if (ty >= a.G || tx >= a.G) return a.bqkv + (long)which * a.Dm + h * HD;
        return a.qkv + ((long)b * T + ty * a.G + tx) * ld + (long)which * a.Dm + h * HD;
    };
    const int qi = w * 32 + li, qic = qi < SW_T ? qi : SW_T - 1;
    const int qh = qic / SW_S, qw = qic % SW_S;
    float qu[HH], qs[HH];
    {
        const float* qp = row_ptr(qic, 0) + lh * HH;
#pragma unroll
        for (int d = 0; d < HH; d += 4) { const float4 t = ld4(qp + d); qu[d] = t.x; qu[d + 1] = t.y; qu[d + 2] = t.z; qu[d + 3] = t.w; }
    }
    // G^T[j][q] = R[j] . q for the 27 table rows (one tile), kept at bh[q][kh = qh + 13 - j] / bw[q][kw = qw + 13 - j]
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        const float* tab = which ? a.rel_w : a.rel_h;
        const float* rp = tab + (long)(li < 2 * SW_S - 1 ? li : 2 * SW_S - 2) * HD + lh * HH;
        f32x16_t g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
        for (int d = 0; d < HH; d += 4) {
            const float4 t = ld4(rp + d);
            g = mfma32f(t.x, qu[d], g); g = mfma32f(t.y, qu[d + 1], g); g = mfma32f(t.z, qu[d + 2], g); g = mfma32f(t.w, qu[d + 3], g);
        }
        const int q0 = which ? qw : qh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = q0 + (SW_S - 1) - ((r & 3) + 8 * (r >> 2) + 4 * lh);
            if (kk >= 0 && kk < SW_S) bt[qi * 32 + which * 16 + kk] = SPLIT ? g[r] * SP_LOG2E : g[r];
        }
    }
#pragma unroll
    for (int d = 0; d < HH; ++d) qs[d] = qu[d] * (SPLIT ? a.scale * SP_LOG2E : a.scale);
    constexpr int NKS = HH / 8;
    uint4 qsh[SPLIT ? NKS : 1], qsl[SPLIT ? NKS : 1];
    if constexpr (SPLIT) {
#pragma unroll
        for (int s8 = 0; s8 < NKS; ++s8) sp_split8(&qs[8 * s8], 1.0f, qsh[s8], qsl[s8]);
    }
    for (int i = tid; i < DT * 32 * SM_VP; i += 448) Vt[i] = 0.f;
    f32x16_t oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m = -3.0e38f, l = 0.f;
    constexpr int NLD = (32 * V4 + 447) / 448;
    float4 rk[NLD], rv[NLD];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 448 * i;
            if (idx < 32 * V4) {
                const int key = t * 32 + idx / V4;
                rk[i] = rv[i] = zero4();
                if (key < SW_T) { const float* kp = row_ptr(key, 1) + (idx % V4) * 4; rk[i] = ld4(kp); rv[i] = ld4(kp + a.Dm); }
            }
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 448 * i;
            if (idx < 32 * V4) {
                const int row = idx / V4, c = (idx % V4) * 4;
                if constexpr (SPLIT) {
                    uint2 h, l;
                    sp_split4(rk[i], 1.0f, h, l);
                    char* const pk = (char*)Ks + row * (KP * 4) + c * 2;
                    *(uint2*)pk = h; *(uint2*)(pk + HD * 2) = l;
                    sp_split4(rv[i], 1.0f, h, l);
                    u16* const pv = (u16*)((char*)Vt + c * (SM_VP * 4)) + sp_vpos(row);
                    pv[0] = (u16)(h.x & 0xffffu); pv[32] = (u16)(l.x & 0xffffu);
                    pv[SM_VP * 2] = (u16)(h.x >> 16); pv[SM_VP * 2 + 32] = (u16)(l.x >> 16);
                    pv[SM_VP * 4] = (u16)(h.y & 0xffffu); pv[SM_VP * 4 + 32] = (u16)(l.y & 0xffffu);
                    pv[SM_VP * 6] = (u16)(h.y >> 16); pv[SM_VP * 6 + 32] = (u16)(l.y >> 16);
                } else {
                *(float4*)&Ks[row * KP + c] = rk[i];
                Vt[(c + 0) * SM_VP + row] = rv[i].x; Vt[(c + 1) * SM_VP + row] = rv[i].y;
                Vt[(c + 2) * SM_VP + row] = rv[i].z; Vt[(c + 3) * SM_VP + row] = rv[i].w;
                }
            }
        }
    };
    gload(0);
    __syncthreads();                                    // the zero fill of Vt, the bias tables
    sstore();
    __syncthreads();
    const float* const bq = bt + qi * 32;
#pragma unroll 1
    for (int t = 0; t < SW_KT; ++t) {
        if (t + 1 < SW_KT) gload(t + 1);
        f32x16_t sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        const float* kp = Ks + li * KP + lh * HH;
        if constexpr (SPLIT) {
            const char* kb = (const char*)Ks + li * (KP * 4) + lh * (HH * 2);
#pragma unroll
            for (int s8 = 0; s8 < NKS; ++s8) {
                const uint4 kh = *(const uint4*)(kb + 16 * s8), kl = *(const uint4*)(kb + HD * 2 + 16 * s8);
                sc = mfma32h(kl, qsh[s8], sc); sc = mfma32h(kh, qsl[s8], sc); sc = mfma32h(kh, qsh[s8], sc);
            }
        } else
#pragma unroll
        for (int d = 0; d < HH; d += 4) {
            const float4 kk = ld4(kp + d);
            sc = mfma32f(kk.x, qs[d], sc); sc = mfma32f(kk.y, qs[d + 1], sc); sc = mfma32f(kk.z, qs[d + 2], sc); sc = mfma32f(kk.w, qs[d + 3], sc);
        }
        float cm = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int kh = key / SW_S, kw = key - kh * SW_S;
            sc[r] = key < SW_T ? (sc[r] + bq[kh]) + bq[16 + kw] : -3.0e38f;
            cm = fmaxf(cm, sc[r]);
        }
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);
        if (__ballot(mn > m)) {
            const float alpha = sp_exp<SPLIT>(m - mn);
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            m = mn;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sc[r] = key < SW_T ? sp_exp<SPLIT>(sc[r] - m) : 0.f;
            l += sc[r];
        }
        if constexpr (SPLIT) {
            uint4 ph[2], pl[2];
            {
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[r] = sc[r];
                sp_split8(&pv[0], SP_PSCALE, ph[0], pl[0]); sp_split8(&pv[8], SP_PSCALE, ph[1], pl[1]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* vb = (const char*)Vt + (dt * 32 + li) * (SM_VP * 4) + 16 * lh;
#pragma unroll
                for (int s8 = 0; s8 < 2; ++s8) {
                    const uint4 vh = *(const uint4*)(vb + 32 * s8), vl = *(const uint4*)(vb + 64 + 32 * s8);
                    oacc[dt] = mfma32h(vl, ph[s8], oacc[dt]); oacc[dt] = mfma32h(vh, pl[s8], oacc[dt]); oacc[dt] = mfma32h(vh, ph[s8], oacc[dt]);
                }
            }
        } else
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const float* vp = Vt + (dt * 32 + li) * SM_VP + 4 * lh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v = ld4(vp + 8 * g4);
                oacc[dt] = mfma32f(v.x, sc[4 * g4], oacc[dt]); oacc[dt] = mfma32f(v.y, sc[4 * g4 + 1], oacc[dt]);
                oacc[dt] = mfma32f(v.z, sc[4 * g4 + 2], oacc[dt]); oacc[dt] = mfma32f(v.w, sc[4 * g4 + 3], oacc[dt]);
            }
        }
        __syncthreads();
        if (t + 1 < SW_KT) sstore();
        __syncthreads();
    }
    l += __shfl_xor(l, 32);
    if constexpr (SPLIT) l *= SP_PSCALE;
    if (qi >= SW_T) return;
    const int ty = wy * SW_S + qh, tx = wx * SW_S + qw;
    if (ty >= a.G || tx >= a.G) return;                              // padded query rows are dropped by window_unpartition
    float* op = a.out + ((long)b * T + ty * a.G + tx) * a.Dm + h * HD;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d = dt * 32 + 8 * g4 + 4 * lh;
            if (d < HD) *(float4*)(op + d) = make_float4(oacc[dt][4 * g4] / l, oacc[dt][4 * g4 + 1] / l, oacc[dt][4 * g4 + 2] / l, oacc[dt][4 * g4 + 3] / l);
        }
}

// ------------------------------------------------------------------------------------------------------------------ image -> token block
// One launch for the whole "image attends to the tokens" step of a TwoWayAttentionBlock (upstream transformer.py: q = keys + key_pe;
// attn_out = cross_attn_image_to_token(q, k = queries + query_pe, v = queries); keys = norm4(keys + attn_out)) over the per-prompt image
// stream - as separate launches (q projection, sattn_short, out projection + residual, LayerNorm) the 0.5 GB stream of a 128-prompt
// chunk crosses HBM seven times, here twice.  Both projections run in the TRANSPOSED orientation (the weights are the A operand, the
// image rows the B operand), so a lane is an image row in every D layout and the lane pair (row, half 0 / 1) holds the row's channels:
//   Q^T [128 ch x 32 rows per wave]: k pairing h * 16 + s as sgemm_kernel - the projection's bits are those of the separate launch;
//   attention: 8 heads x Tk <= 16 tokens, a head's 16 channels are 8 registers in each lane of the pair (one exchange per score);
//   out^T = W_o att^T: the attention's registers ARE the B operand (step s contracts the channel pair the two lanes hold in register s);
//   + bias + residual, LayerNorm over the row's 256 channels (two passes, one exchange each), all in registers;
//   residual tiles in and result tiles out pass a wave-private LDS tile so that HBM sees whole 128-byte rows.
MSAM_DEVINL void si_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
struct SI2TArgs {
    const float* keys; long key_bs;                     // [B or 1][4096][256]; batch stride 0: one stream shared by every prompt (layer 0)
    const float* pos;                                   // [4096][256] dense positional encoding
    const float* wq; const float* bq;                   // [128][256], [128]
    const float* tk; const float* tv; long ldt, tok_bs; // token-side k / v projections [B][Tk][128]
    const float* wo; const float* bo;                   // [256][128], [256]
    const float* lnw; const float* lnb; float eps, denom;
    float* out;                                         // [B][4096][256] (may be `keys` when that is per prompt)
    int B, Tk;
    int dbg;                                            // timing experiments (msam_tune_set "si2t_dbg"; WRONG results): 1 no projection 1, 2 no attention,
                                                        // 4 no projection 2, 8 no residual / LayerNorm
    int late_lo, late_hi, late_ticks;                   // workgroups [late_lo, late_hi) - the second one of every CU in the first dispatch round -
                                                        // start late_ticks x 10 ns late (de-phasing, see the kernel)
    float wq_scale, wo_scale;                           // SPLIT: powers of two applied to W_q / W_o before the fp16 split (undone on the accumulators)
    const unsigned short* wq16; const unsigned short* wo16;   // SPLIT, optional: the weights ALREADY as fp16 pairs in the kernel's LDS tile layout
};                                                      // (msam_split16_prepare_pairs: row n = per k-tile of 32 [32 hi | 32 lo]; W_o with the permuted k order)

// SPLIT: the split16 mode - both projections on fp16 operand pairs (3 x v_mfma_f32_32x32x16_f16 per product and 16-deep k-step); LDS rows hold
// 32 hi | 32 lo halves as in sgemm_kernel<SPLIT>.  In projection 2 the attention's registers are still the B operand: k-step s of channel tile kt
// takes a lane's registers 8 s .. 8 s + 7 = channels 16 s + 4 lh + {0..3, 8..11}, so W_o's k-tile is stored with that permutation
// (any assignment of k that is the same on both operands is a valid contraction).  Attention, residual and LayerNorm: the same fp32 code.
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void si2t_kernel(SI2TArgs a) {
    constexpr int C = 256, CI = 128, P = SG_PITCH;
    // one LDS block, the small broadcast tables first (ds offsets are 16-bit immediates: an array placed above 64 KB costs an address
    // register per access)
    // (59 KB: two workgroups per CU.  With the token tables in their own 16 KB the block was 76 KB and only ONE workgroup was resident
    //  per CU - 814 us per 128 prompts, exactly 16 serial workgroups per CU)
    __shared__ __attribute__((aligned(16))) float si_lds[4 * 256 + 128 * P + 256 * P];
    float* const par = si_lds;                          // bq (128, padded) | bo | LayerNorm weight | bias
    float* const Xs = par + 4 * 256;                    // projection 1: the (keys + pos) k-tile; attention: the token tables; epilogue: four wave-private [32][P] tiles
    float* const tks = Xs;                              // token k [16][128] (after projection 1)
    float* const tvs = Xs + 16 * CI;                    // token v
    float* const Wt = Xs + 128 * P;                     // weight k-tile: W_q (128 rows), then W_o (256 rows)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int b = (int)(blockIdx.x >> 5), rb = (int)(blockIdx.x & 31);
    par[tid] = tid < CI ? a.bq[tid] : 0.f; par[256 + tid] = a.bo[tid]; par[512 + tid] = a.lnw[tid]; par[768 + tid] = a.lnb[tid];
    // Two workgroups share a CU (one wave of each per SIMD).  Measured, the launch takes exactly the SUM of its phases (tools/si2t_probe.py:
    // projection 1 286 us, attention 120, projection 2 216 - the MFMA rate of the chip -, residual + LayerNorm 81, loads / stores 121 of
    // 818 us per 128 prompts), and starting the second workgroup of every CU late (this experiment knob, 5 - 50 us) changes nothing:
    // on this chip the matrix pipe and the vector ALU of a SIMD do not run at the same time, whichever waves the instructions come
    // from (tools/mfma_valu_overlap_probe.hip; profiles/r05_experiments.md section 8).
    if (a.late_ticks > 0 && (int)blockIdx.x >= a.late_lo && (int)blockIdx.x < a.late_hi) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)a.late_ticks) __builtin_amdgcn_s_sleep(16);
    }
    const float* const kin = a.keys + (long)b * a.key_bs + (long)rb * 128 * C;
    const float* const pin = a.pos + (long)rb * 128 * C;
    float* const outp = a.out + ((long)b * 4096 + rb * 128) * C;
    // the prompt's token k / v rows wait in registers until projection 1 releases its LDS tile (thread -> rows tid / 32 and 8 + tid / 32)
    float4 tkr[2], tvr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = (tid >> 5) + 8 * i, c4 = (tid & 31) * 4;
        tkr[i] = tvr[i] = zero4();
        if (t < a.Tk) { tkr[i] = ld4(a.tk + (long)b * a.tok_bs + (long)t * a.ldt + c4); tvr[i] = ld4(a.tv + (long)b * a.tok_bs + (long)t * a.ldt + c4); }
    }
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;
    float4 rx[4], rw[8];
    auto load1 = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long o = (long)(srow + 32 * j) * C + kt * 32 + sc4;
            const float4 x = ld4(kin + o), pe = ld4(pin + o);
            rx[j] = make_float4(x.x + pe.x, x.y + pe.y, x.z + pe.z, x.w + pe.w);
            if (SPLIT && a.wq16) {                      // prepared pairs: 16-byte pieces of the tile's 128-byte rows, copied as they are
                const int idx = tid + 256 * j;
                rw[j] = ld4((const float*)((const char*)a.wq16 + (long)(idx >> 3) * 1024 + kt * 128 + (idx & 7) * 16));
            } else rw[j] = ld4(a.wq + o);
        }
    };
    auto store1 = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (SPLIT) {
                uint2 h, l;
                char* const px_ = (char*)&Xs[(srow + 32 * j) * P] + sc4 * 2;
                char* const pw_ = (char*)&Wt[(srow + 32 * j) * P] + sc4 * 2;
                sp_split4(rx[j], 1.0f, h, l);
                *(uint2*)px_ = h; *(uint2*)(px_ + 64) = l;
                if (a.wq16) {
                    const int idx = tid + 256 * j;
                    *(float4*)((char*)Wt + (idx >> 3) * (P * 4) + (idx & 7) * 16) = rw[j];
                } else {
                    sp_split4(rw[j], a.wq_scale, h, l);
                    *(uint2*)pw_ = h; *(uint2*)(pw_ + 64) = l;
                }
            } else {
                *(float4*)&Xs[(srow + 32 * j) * P + sc4] = rx[j];
                *(float4*)&Wt[(srow + 32 * j) * P + sc4] = rw[j];
            }
        }
    };
    // W_o k-tile: thread = output channel, its 32 k values are 128 contiguous bytes (one pointer, immediate offsets: the registers
    // of this phase go to the 64 + 128 accumulators)
    const float* const wop = a.wo + (long)tid * CI;
    auto load3 = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (SPLIT && a.wo16) {
                const int idx = tid + 256 * j;
                rw[j] = ld4((const float*)((const char*)a.wo16 + (long)(idx >> 3) * 512 + kt * 128 + (idx & 7) * 16));
            } else rw[j] = ld4(wop + kt * 32 + 4 * j);
        }
    };
    auto store3 = [&]() {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (SPLIT && a.wo16) {
                const int idx = tid + 256 * j;
                *(float4*)((char*)Wt + (idx >> 3) * (P * 4) + (idx & 7) * 16) = rw[j];
            } else if constexpr (SPLIT) {
                // channels 4 j .. 4 j + 3 of the k-tile: k-step j >> 2, lane half j & 1, first / second group of four (j >> 1) & 1
                uint2 h, l;
                char* const pw_ = (char*)&Wt[tid * P] + ((j >> 2) * 16 + (j & 1) * 8 + ((j >> 1) & 1) * 4) * 2;
                sp_split4(rw[j], a.wo_scale, h, l);
                *(uint2*)pw_ = h; *(uint2*)(pw_ + 64) = l;
            } else *(float4*)&Wt[tid * P + 4 * j] = rw[j];
        }
    };
    // ---- Q^T = W_q (keys + pos)^T
    f32x16_t q[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) q[nt][r] = 0.f;
    load1(0);
    store1();
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < ((a.dbg & 1) ? 0 : C / 32); ++kt) {
        if (kt + 1 < C / 32) load1(kt + 1);
        const float* px = &Xs[(w * 32 + li) * P + lh * (SPLIT ? 4 : 16)];
        const float* pw = &Wt[li * P + lh * (SPLIT ? 4 : 16)];
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 xh = *(const uint4*)((const char*)px + ks * 32), xl = *(const uint4*)((const char*)px + ks * 32 + 64);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const char* qw = (const char*)(pw + nt * 32 * P) + ks * 32;
                    const uint4 wh = *(const uint4*)qw, wl = *(const uint4*)(qw + 64);
                    q[nt] = mfma32h(wl, xh, q[nt]); q[nt] = mfma32h(wh, xl, q[nt]); q[nt] = mfma32h(wh, xh, q[nt]);
                }
            }
        } else
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const float4 xb = ld4(px + s4 * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const float4 wa = ld4(pw + nt * 32 * P + s4 * 4);
                q[nt] = mfma32f(wa.x, xb.x, q[nt]); q[nt] = mfma32f(wa.y, xb.y, q[nt]);
                q[nt] = mfma32f(wa.z, xb.z, q[nt]); q[nt] = mfma32f(wa.w, xb.w, q[nt]);
            }
        }
        __syncthreads();
        if (kt + 1 < C / 32) store1();
        __syncthreads();
    }
    load3(0);                                           // W_o's first k-tile arrives under the attention
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int t = (tid >> 5) + 8 * i, c4 = (tid & 31) * 4;
        *(float4*)&tks[t * CI + c4] = tkr[i];
        *(float4*)&tvs[t * CI + c4] = tvr[i];
    }
    __syncthreads();
    // register r of tile nt = channel nt * 32 + (r & 3) + 8 (r >> 2) + 4 lh: registers 0..7 belong to head 2 nt, 8..15 to head 2 nt + 1
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bb = ld4(&par[nt * 32 + 8 * g4 + 4 * lh]);
            if constexpr (SPLIT) {
                const float iq = 1.0f / a.wq_scale;
                q[nt][4 * g4] = q[nt][4 * g4] * iq + bb.x; q[nt][4 * g4 + 1] = q[nt][4 * g4 + 1] * iq + bb.y;
                q[nt][4 * g4 + 2] = q[nt][4 * g4 + 2] * iq + bb.z; q[nt][4 * g4 + 3] = q[nt][4 * g4 + 3] * iq + bb.w;
            } else {
                q[nt][4 * g4] += bb.x; q[nt][4 * g4 + 1] += bb.y; q[nt][4 * g4 + 2] += bb.z; q[nt][4 * g4 + 3] += bb.w;
            }
        }
    // ---- softmax((q . k) / denom) v per head, in place over q
    const float inv_denom = 1.0f / a.denom;
    if (!(a.dbg & 2))
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int c0 = nt * 32 + 16 * g + 4 * lh;   // the lane's channels of this head: c0 .. c0 + 3 and c0 + 8 .. c0 + 11
            float sc[16];
            float mx = -3.0e38f;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                sc[t] = -3.0e38f;
                if (t < a.Tk) {
                    const float4 k0 = ld4(&tks[t * CI + c0]), k1 = ld4(&tks[t * CI + c0 + 8]);
                    float part = q[nt][8 * g] * k0.x;
                    part = fmaf(q[nt][8 * g + 1], k0.y, part); part = fmaf(q[nt][8 * g + 2], k0.z, part); part = fmaf(q[nt][8 * g + 3], k0.w, part);
                    part = fmaf(q[nt][8 * g + 4], k1.x, part); part = fmaf(q[nt][8 * g + 5], k1.y, part);
                    part = fmaf(q[nt][8 * g + 6], k1.z, part); part = fmaf(q[nt][8 * g + 7], k1.w, part);
                    const float other = __shfl_xor(part, 32);
                    // (split16: one multiplication by the reciprocal instead of an IEEE division per score - 1 / denom is exact for upstream's
                    //  sqrt(16) - and one division per head instead of one per probability below: ~130 of a head's ~360 vector instructions)
                    sc[t] = SPLIT ? (lh ? other + part : part + other) * (inv_denom * SP_LOG2E) : (lh ? other + part : part + other) / a.denom;
                    mx = fmaxf(mx, sc[t]);
                }
            }
            float l = 0.f;
#pragma unroll
            for (int t = 0; t < 16; ++t) { sc[t] = t < a.Tk ? (SPLIT ? __builtin_amdgcn_exp2f(sc[t] - mx) : expf(sc[t] - mx)) : 0.f; l += sc[t]; }
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.f;
            const float inv_l = 1.0f / l;
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (t < a.Tk) {
                    const float pr = SPLIT ? sc[t] * inv_l : sc[t] / l;
                    const float4 v0 = ld4(&tvs[t * CI + c0]), v1 = ld4(&tvs[t * CI + c0 + 8]);
                    o[0] = fmaf(pr, v0.x, o[0]); o[1] = fmaf(pr, v0.y, o[1]); o[2] = fmaf(pr, v0.z, o[2]); o[3] = fmaf(pr, v0.w, o[3]);
                    o[4] = fmaf(pr, v1.x, o[4]); o[5] = fmaf(pr, v1.y, o[5]); o[6] = fmaf(pr, v1.z, o[6]); o[7] = fmaf(pr, v1.w, o[7]);
                }
#pragma unroll
            for (int e = 0; e < 8; ++e) q[nt][8 * g + e] = o[e];
        }
    // ---- out^T = W_o att^T
    f32x16_t oa[8];
#pragma unroll
    for (int ot = 0; ot < 8; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) oa[ot][r] = 0.f;
    store3();
    __syncthreads();
    if (!(a.dbg & 4))
#pragma unroll
    for (int kt = 0; kt < CI / 32; ++kt) {
        const float* pw = &Wt[li * P + 4 * lh];
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // the lane's eight attention values of this k-step as an fp16 pair
                uint4 bh, bl;
                {
                    uint2 h0, l0, h1, l1;
                    sp_split4(make_float4(q[kt][8 * ks], q[kt][8 * ks + 1], q[kt][8 * ks + 2], q[kt][8 * ks + 3]), 1.0f, h0, l0);
                    sp_split4(make_float4(q[kt][8 * ks + 4], q[kt][8 * ks + 5], q[kt][8 * ks + 6], q[kt][8 * ks + 7]), 1.0f, h1, l1);
                    bh = uint4{h0.x, h0.y, h1.x, h1.y}; bl = uint4{l0.x, l0.y, l1.x, l1.y};
                }
#pragma unroll
                for (int ot = 0; ot < 8; ++ot) {
                    const char* qw = (const char*)(pw + ot * 32 * P) + ks * 32;
                    const uint4 wh = *(const uint4*)qw, wl = *(const uint4*)(qw + 64);
                    oa[ot] = mfma32h(wl, bh, oa[ot]); oa[ot] = mfma32h(wh, bl, oa[ot]); oa[ot] = mfma32h(wh, bh, oa[ot]);
                }
            }
        } else {
        // one W_o fragment ahead of the MFMAs, no further (sched_barrier): 192 accumulator registers are live here
        float4 wa = ld4(pw);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) {
                const int nx = g4 * 8 + ot + 1;
                const float4 wn = nx < 32 ? ld4(pw + (nx & 7) * 32 * P + 8 * (nx >> 3)) : wa;
                oa[ot] = mfma32f(wa.x, q[kt][4 * g4], oa[ot]); oa[ot] = mfma32f(wa.y, q[kt][4 * g4 + 1], oa[ot]);
                oa[ot] = mfma32f(wa.z, q[kt][4 * g4 + 2], oa[ot]); oa[ot] = mfma32f(wa.w, q[kt][4 * g4 + 3], oa[ot]);
                __builtin_amdgcn_sched_barrier(0);
                wa = wn;
            }
        }
        // (the next k-tile is fetched only now: 64 + 128 accumulator registers leave no room for a prefetch across the MFMAs - the other
        //  workgroup of the CU runs under this latency)
        if (kt + 1 < CI / 32) load3(kt + 1);
        __syncthreads();
        if (kt + 1 < CI / 32) store3();
        __syncthreads();
    }
    // ---- (+ bias) + residual, LayerNorm, store.  The staging tile is wave-private: LDS instructions of one wave execute in issue
    // order, so a wave-level fence (no workgroup barrier: the four waves drift apart here) orders its writes before its reads.
    float* const st = &Xs[w * 32 * P];
    float sum = 0.f;
    // residual tiles two at a time: their 8 loads are in flight together
    if (!(a.dbg & 8))
#pragma unroll
    for (int ob = 0; ob < 8; ob += 2) {
        __builtin_amdgcn_sched_barrier(0);
        float4 rr[2][4];
#pragma unroll
        for (int o4 = 0; o4 < 2; ++o4)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i, row = idx >> 3, c4 = (idx & 7) * 4;
                rr[o4][i] = ld4(kin + (long)(w * 32 + row) * C + (ob + o4) * 32 + c4);
            }
#pragma unroll
        for (int o4 = 0; o4 < 2; ++o4) {
            const int ot = ob + o4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i, row = idx >> 3, c4 = (idx & 7) * 4;
                *(float4*)&st[row * P + c4] = rr[o4][i];
            }
            si_wave_sync();
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 rv = ld4(&st[li * P + 8 * g4 + 4 * lh]), bb = ld4(&par[256 + ot * 32 + 8 * g4 + 4 * lh]);
                if constexpr (SPLIT) {
                    const float io = 1.0f / a.wo_scale;
                    oa[ot][4 * g4] *= io; oa[ot][4 * g4 + 1] *= io; oa[ot][4 * g4 + 2] *= io; oa[ot][4 * g4 + 3] *= io;
                }
                oa[ot][4 * g4] = (oa[ot][4 * g4] + bb.x) + rv.x; oa[ot][4 * g4 + 1] = (oa[ot][4 * g4 + 1] + bb.y) + rv.y;
                oa[ot][4 * g4 + 2] = (oa[ot][4 * g4 + 2] + bb.z) + rv.z; oa[ot][4 * g4 + 3] = (oa[ot][4 * g4 + 3] + bb.w) + rv.w;
                sum += (oa[ot][4 * g4] + oa[ot][4 * g4 + 1]) + (oa[ot][4 * g4 + 2] + oa[ot][4 * g4 + 3]);
            }
            si_wave_sync();
        }
    }
    {
        const float other = __shfl_xor(sum, 32);
        sum = lh ? other + sum : sum + other;
    }
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int ot = 0; ot < 8; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = oa[ot][r] - mean; sq = fmaf(d, d, sq); }
    {
        const float other = __shfl_xor(sq, 32);
        sq = lh ? other + sq : sq + other;
    }
    const float rstd = 1.0f / sqrtf(sq / (float)C + a.eps);
#pragma unroll
    for (int ot = 0; ot < 8; ++ot) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int c = ot * 32 + 8 * g4 + 4 * lh;
            const float4 ww = ld4(&par[512 + c]), bb = ld4(&par[768 + c]);
            *(float4*)&st[li * P + 8 * g4 + 4 * lh] =
                make_float4((oa[ot][4 * g4] - mean) * rstd * ww.x + bb.x, (oa[ot][4 * g4 + 1] - mean) * rstd * ww.y + bb.y,
                            (oa[ot][4 * g4 + 2] - mean) * rstd * ww.z + bb.z, (oa[ot][4 * g4 + 3] - mean) * rstd * ww.w + bb.w);
        }
        si_wave_sync();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i, row = idx >> 3, c4 = (idx & 7) * 4;
            *(float4*)(outp + (long)(w * 32 + row) * C + ot * 32 + c4) = ld4(&st[row * P + c4]);
        }
        si_wave_sync();
    }
}

// fp32 weight [N][K] (K % 32 == 0) -> fp16 pairs in the LDS tile layout of the split16 kernels: row n = K / 32 k-tiles of [32 hi | 32 lo] halves;
// perm: the k order of si2t_kernel's second projection inside a k-tile (channels 4 j .. 4 j + 3 at (j >> 2) * 16 + (j & 1) * 8 + ((j >> 1) & 1) * 4)
__global__ __launch_bounds__(256) void s16_prepare_pairs_kernel(const float* __restrict__ w, long N, int K, float scale, int perm, unsigned short* __restrict__ out) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int k4 = K / 4;
    if (gid >= N * k4) return;
    const long n = gid / k4;
    const int c4 = (int)(gid - n * k4), kt = c4 >> 3, j = c4 & 7;
    uint2 h, l;
    sp_split4(ld4(w + n * K + c4 * 4), scale, h, l);
    const int pos = perm ? (j >> 2) * 16 + (j & 1) * 8 + ((j >> 1) & 1) * 4 : j * 4;
    unsigned short* dst = out + n * (2L * K) + kt * 64 + pos;
    *(uint2*)dst = h; *(uint2*)(dst + 32) = l;
}

// ------------------------------------------------------------------------------------------------------------------ token -> image attention (split16)
// "The tokens attend to the image" of a TwoWayAttentionBlock / the final attention on a PER-PROMPT image stream, without the k and v projections
// of the stream ever existing (upstream transformer.py Attention.forward: k = k_proj(keys + key_pe), v = v_proj(keys); as launches - one
// product over [Wk; Wv] writing k | v, then sattn_long_kernel reading them - 17 GB cross HBM per layer and 1024 prompts; here the 4.3 GB stream
// is read once).  The projections are folded into the token side, which is exact algebra:
//   score[j, h, t] = (q[j, h] . k[t, h]) / sqrt(d) = (X[t] + pos[t]) . G[j h] + const(j, h),   G[j h] = Wk_h^T q[j, h] / sqrt(d)   (the constant
//                                                                                                 q . bk cancels in the softmax over t)
//   out[j, h]      = sum_t p[j h, t] v[t, h]     = Wv_h U[j h] + bv_h,                          U[j h] = sum_t p[j h, t] X[t]
// so per prompt the kernel forms S = (X + pos) G^T [4096 x 64] and U = P^T X [64 x 256] (64 = 8 heads x 8 token slots, Tk <= 8) with an online
// softmax over the 4096 image tokens in between; two small launches do the folds (s16_t2i_fold_kernel before, s16_t2i_out_kernel after, fp32).
// One workgroup = one prompt, 4 waves = (jt: heads 0-3 / 4-7) x (ih: channels 0-127 / 128-255), one wave per SIMD (G lives in 128 registers):
//   S tile [32 rows x 32 jh]: A = the (X + pos) tile's rows from LDS (row-major fp16 pairs), B = G from registers; D: a LANE is a (token, head)
//     pair jh, its registers are image rows -> the running maximum / sum of the softmax are per lane (one exchange with lane ^ 32 per tile);
//   U tile [32 jh x 128 ch] += P^T X: A = the lane's OWN exponentials (registers 8 s .. 8 s + 7 are k-step s: rows 16 s + 4 lh + {0..3, 8..11}),
//     B = columns of a second, pos-free row-major copy of the tile through the transposing LDS read (ds_read_b64_tr_b16, two per operand: rows
//     16 s + 4 lh + 0..3 and + 8..11); the rare rescale of U when a maximum grows goes through a wave-private LDS vector (U's registers run
//     over jh).  (First form: a transposed copy written with 8-byte stores - 8-way bank conflicts on every one of them.)
// Both waves of a jt compute the same S tile (the price of not exchanging P through LDS): 72 MFMAs per wave and 32-row tile, 0.6 ms of matrix
// time per 1024-prompt layer against 0.9 ms of HBM time for the stream.
struct ST2IArgs {
    const float* keys; long key_bs;                     // [B][4096][256]
    const float* pos;                                   // [4096][256]
    const unsigned short* g;                            // [B][64][512]: G as fp16 pairs (256 hi | 256 lo), scaled by T2I_GSCALE
    float* u;                                           // [B][64][256]: U, normalised by the softmax denominator
    int B;
};
constexpr float T2I_GSCALE = 64.0f;
constexpr int T2I_RP = 1040;                            // (X + pos) row pitch in bytes: 256 hi | 256 lo halves + 16
constexpr int T2I_NP = 1088;                            // X row pitch (the second, pos-free copy): 272 dwords = 16 mod 64 - the transposing reads of a
                                                        // 32-lane group (4 rows x 8 column blocks) then touch 64 different banks
constexpr int T2I_BUF = 32 * T2I_RP + 32 * T2I_NP;      // one stage: 68 096 bytes
constexpr int T2I_LDS = 2 * T2I_BUF + 4 * 32 * 4;       // + one 32-float vector per wave

// G[p][h * 8 + j][i] = sum_{c in head h} q[p, j, c] Wk[c, i] / denom as fp16 pairs (thread = (p, jh, four columns))
__global__ __launch_bounds__(256) void s16_t2i_fold_kernel(const float* __restrict__ q, long ldq, int Tk, const float* __restrict__ wk, float denom,
                                                           long total, unsigned short* __restrict__ g) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int i4 = (int)(gid & 63) * 4, jh = (int)((gid >> 6) & 63);
    const long p = gid >> 12;
    const int h = jh >> 3, j = jh & 7;
    float4 acc = zero4();
    if (j < Tk) {
        const float* qp = q + (p * Tk + j) * ldq + h * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float qc = qp[c];
            const float4 wv = ld4(wk + (long)(h * 16 + c) * 256 + i4);
            acc.x = fmaf(qc, wv.x, acc.x); acc.y = fmaf(qc, wv.y, acc.y); acc.z = fmaf(qc, wv.z, acc.z); acc.w = fmaf(qc, wv.w, acc.w);
        }
        acc.x /= denom; acc.y /= denom; acc.z /= denom; acc.w /= denom;
    }
    uint2 hi, lo;
    sp_split4(acc, T2I_GSCALE, hi, lo);
    unsigned short* dst = g + (p * 64 + jh) * 512 + i4;
    *(uint2*)dst = hi; *(uint2*)(dst + 256) = lo;
}
// att[p, j, c] = Wv[c] . U[p][h(c) * 8 + j] + bv[c]  (thread = one output)
__global__ __launch_bounds__(256) void s16_t2i_out_kernel(const float* __restrict__ u, const float* __restrict__ wv, const float* __restrict__ bv, int Tk,
                                                          long total, float* __restrict__ out, long ldo) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int c = (int)(gid & 127);
    const long pj = gid >> 7;
    const long p = pj / Tk;
    const int j = (int)(pj - p * Tk);
    const float* up = u + (p * 64 + (c >> 4) * 8 + j) * 256;
    const float* wp = wv + (long)c * 256;
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < 256; i += 4) {
        const float4 a = ld4(up + i), b = ld4(wp + i);
        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
    }
    out[pj * ldo + c] = acc + bv[c];
}

__global__ __launch_bounds__(256, 1) void s16_t2i_kernel(ST2IArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t2_lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int jt = w >> 1, ih = w & 1;
    const long p = blockIdx.x;
    float* const av = (float*)(t2_lds + 2 * T2I_BUF) + w * 32;      // wave-private vector (rescale factors / final 1 / l)
    const float* const xin = a.keys + p * a.key_bs;
    // G of the lane's jh = jt * 32 + li: 16 k-steps x (hi, lo) of eight channels
    uint4 gh[16], gl[16];
    {
        const unsigned short* gp = a.g + (p * 64 + jt * 32 + li) * 512 + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) { gh[ks] = *(const uint4*)(gp + ks * 16); gl[ks] = *(const uint4*)(gp + 256 + ks * 16); }
    }
    // staging: thread -> two 4 x 4 blocks (rows 4 rg .. + 3, channels 4 cg .. + 3) of the 32 x 256 tile
    float4 rx[2][4], rp[2][4];
    auto gload = [&](int t) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = tid + 256 * k, rg = b >> 6, cg = b & 63;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long o = (long)(t * 32 + rg * 4 + r) * 256 + cg * 4;
                rx[k][r] = ld4(xin + o); rp[k][r] = ld4(a.pos + o);
            }
        }
    };
    auto sstore = [&](int buf) {
        unsigned char* const xr = t2_lds + buf * T2I_BUF;
        unsigned char* const xn = xr + 32 * T2I_RP;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int b = tid + 256 * k, rg = b >> 6, cg = b & 63;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                uint2 h, l;
                sp_split4(make_float4(rx[k][r].x + rp[k][r].x, rx[k][r].y + rp[k][r].y, rx[k][r].z + rp[k][r].z, rx[k][r].w + rp[k][r].w), 1.0f, h, l);
                unsigned char* q = xr + (rg * 4 + r) * T2I_RP + cg * 8;
                *(uint2*)q = h; *(uint2*)(q + 512) = l;
                sp_split4(rx[k][r], 1.0f, h, l);
                q = xn + (rg * 4 + r) * T2I_NP + cg * 8;
                *(uint2*)q = h; *(uint2*)(q + 512) = l;
            }
        }
    };
    f32x16_t uacc[4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) uacc[it][r] = 0.f;
    float m = -3.0e38f, l = 0.f, lc = 0.f;              // l: compensated (Kahan) sum - one exponential of 1.0 and 2047 tiny ones per lane otherwise
    gload(0);                                           // lose 1e-5 of the small ones' mass to the 6e-8 grid of a sum near 1
    sstore(0);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 128; ++t) {
        if (t + 1 < 128) gload(t + 1);
        const unsigned char* const xr = t2_lds + (t & 1) * T2I_BUF;
        const unsigned char* const xn = xr + 32 * T2I_RP;
        // S tile: rows x jh
        // (two accumulators: the hi x hi terms - 16 roundings at the score's magnitude, as a 16-step fp32 chain has - and the small cross terms,
        //  whose 32 roundings happen at 2^-11 of it; one accumulator rounded the +-20 score 48 times: 5e-5 instead of 5e-6 on the attention output)
        f32x16_t sc, scl;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = 0.f; scl[r] = 0.f; }
        const unsigned char* xa = xr + li * T2I_RP + lh * 16;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const uint4 xh = *(const uint4*)(xa + ks * 32), xl = *(const uint4*)(xa + 512 + ks * 32);
            scl = mfma32h(xl, gh[ks], scl); scl = mfma32h(xh, gl[ks], scl); sc = mfma32h(xh, gh[ks], sc);
        }
        float cm = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sc[r] = (sc[r] + scl[r]) * (SP_LOG2E / T2I_GSCALE); cm = fmaxf(cm, sc[r]); }     // scores in units of ln 2
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);
        if (__ballot(mn > m)) {
            // a maximum grew: U's rows jh are rescaled (U's registers run over jh: the factors travel through the wave-private vector)
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            l *= alpha; lc *= alpha;
            if (lh == 0) av[li] = alpha;
            si_wave_sync();
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 f = ld4(&av[8 * g4 + 4 * lh]);
#pragma unroll
                for (int it = 0; it < 4; ++it) { uacc[it][4 * g4] *= f.x; uacc[it][4 * g4 + 1] *= f.y; uacc[it][4 * g4 + 2] *= f.z; uacc[it][4 * g4 + 3] *= f.w; }
            }
            si_wave_sync();
            m = mn;
        }
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[r] = __builtin_amdgcn_exp2f(sc[r] - m);
        {
            const float ts = (((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]))) +
                             (((pv[8] + pv[9]) + (pv[10] + pv[11])) + ((pv[12] + pv[13]) + (pv[14] + pv[15])));
            const float y = ts - lc, tn = l + y;
            lc = (tn - l) - y; l = tn;
        }
        uint4 ph[2], pl[2];
        sp_split8(&pv[0], SP_PSCALE, ph[0], pl[0]); sp_split8(&pv[8], SP_PSCALE, ph[1], pl[1]);
        // U tile: jh x channels of this wave's half
        // (as a SOURCE lane of the transposing read, lane r of a 16-lane group addresses row 4 lh + (r >> 2) [+ 8, + 16 s] and the four channels
        //  4 (r & 3) .. + 3 of the group's 16)
        const unsigned char* xb = xn + (4 * lh + ((li & 15) >> 2)) * T2I_NP + (ih * 128 + (li >> 4) * 16 + (li & 3) * 4) * 2;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int s8 = 0; s8 < 2; ++s8) {
                const unsigned char* q0 = xb + it * 64 + s8 * 16 * T2I_NP;
                const uint2 h0 = sp_tr16(q0), h1 = sp_tr16(q0 + 8 * T2I_NP), l0 = sp_tr16(q0 + 512), l1 = sp_tr16(q0 + 8 * T2I_NP + 512);
                const uint4 xh = uint4{h0.x, h0.y, h1.x, h1.y}, xl = uint4{l0.x, l0.y, l1.x, l1.y};
                uacc[it] = mfma32h(pl[s8], xh, uacc[it]); uacc[it] = mfma32h(ph[s8], xl, uacc[it]); uacc[it] = mfma32h(ph[s8], xh, uacc[it]);
            }
        }
        if (t + 1 < 128) sstore((t + 1) & 1);
        __syncthreads();
    }
    // 1 / (sum of the exponentials): both halves of a lane pair hold partial sums of the same jh
    l -= lc;
    l += __shfl_xor(l, 32);
    if (lh == 0) av[li] = 1.0f / (l * SP_PSCALE);
    si_wave_sync();
    float* const up = a.u + (p * 64 + jt * 32) * 256 + ih * 128 + li;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const float4 f = ld4(&av[8 * g4 + 4 * lh]);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r0 = 8 * g4 + 4 * lh;
            up[(long)(r0 + 0) * 256 + it * 32] = uacc[it][4 * g4] * f.x; up[(long)(r0 + 1) * 256 + it * 32] = uacc[it][4 * g4 + 1] * f.y;
            up[(long)(r0 + 2) * 256 + it * 32] = uacc[it][4 * g4 + 2] * f.z; up[(long)(r0 + 3) * 256 + it * 32] = uacc[it][4 * g4 + 3] * f.w;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ image -> token block, folded (split16)
// The same step as si2t_kernel - keys = norm4(keys + cross_attn_image_to_token(keys + pos, k, v)) - with BOTH projections folded into the <= 8 tokens
// of a prompt (exact algebra, as the default mode's fold kernels do):
//   score[t, h, j] = ((x_t + pos_t) Wq_h^T + bq_h) . k[j, h] / d = (x_t + pos_t) . G[h j] + beta[h j],   G[h j] = Wq_h^T k[j, h] / d,  beta = bq_h . k[j, h] / d
//   out[t]         = sum_h (softmax_j score[t, h, :]) v[:, h] Wo[:, h]^T + bo = sum_{h j} p[t, h j] VO[h j] + bo,   VO[h j] = Wo[:, 16 h .. 16 h + 15] v[j, h]
// si2t_kernel streams W_q and W_o (256 KB of fp32) through LDS for EVERY 128-row block - 0.77 MB of loads per block, and the launch is bound by that
// (9 B per cycle and CU): here a prompt's two operands (G: 64 x 256, VO: 256 x 64, fp16 pairs, 128 KB) are staged ONCE per prompt and the workgroup
// walks the prompt's 4096 rows.  One workgroup = one prompt, one wave per SIMD (512 registers), a wave owns 32-row tiles and exchanges nothing:
//   S^T [64 hj x 32 rows] = G (A, LDS) x (x + pos)^T (B, straight from global memory: lane (row, half) loads the channels 16 s + 4 half + {0..3, 8..11} of
//     k-step s - the SAME channels its accumulators of the second product hold, so the fp32 values stay in registers as the residual);
//   the hj order puts the 8 tokens of a head into one lane's registers (rows {0-3, 8-11} | {16-19, 24-27} of a tile and lane half): softmax lane-local;
//   out^T [256 ch x 32 rows] = VO^T (A, LDS) x P (B = the lane's own probabilities, registers 8 s .. 8 s + 7 of k-step s);
//   + bias + residual, LayerNorm over the lane pair's 256 channels, rows leave through a wave-private LDS tile as whole 128-byte pieces.
struct SI2FArgs {
    const float* keys; long key_bs;                     // [B or 1][4096][256]
    const float* pos;                                   // [4096][256]
    const unsigned short* g;                            // [B][64 rho][512]: G as fp16 pairs in operand order ([s][half][e] hi | the same lo), scaled by I2F_GSCALE
    const float* beta;                                  // [B][64 rho]: score constants (-3e38 for the unused token slots)
    const unsigned short* vo;                           // [B][256 c][128]: VO^T as fp16 pairs in operand order ([s][half][j] hi | lo), scaled by I2F_VSCALE
    const float* bo; const float* lnw; const float* lnb; float eps;
    float* out;                                         // [B][4096][256] (may be keys when that is per prompt)
    int B;
};
constexpr float I2F_GSCALE = 64.0f, I2F_VSCALE = 64.0f;
constexpr int I2F_GP = 1040;                            // G row pitch in bytes (512 hi + 512 lo + 16)
constexpr int I2F_VP = 272;                             // VO^T row pitch (128 hi + 128 lo + 16)
constexpr int I2F_SP = 144;                             // staging tile row pitch (32 floats + 4)
constexpr int I2F_LDS = 64 * I2F_GP + 256 * I2F_VP + 4 * 32 * I2F_SP + 4 * 256 * 4;     // 66 560 + 69 632 + 18 432 + 4 096 = 158 720 bytes

// thread = (prompt, rho, four channel slots) for G / beta; (prompt, c_out, four hj slots) for VO
__global__ __launch_bounds__(256) void s16_i2t_fold_kernel(const float* __restrict__ tk, const float* __restrict__ tv, long ldt, long tok_bs, int Tk,
                                                           const float* __restrict__ wq, const float* __restrict__ bq, const float* __restrict__ wo, float denom,
                                                           long B, unsigned short* __restrict__ g, float* __restrict__ beta, unsigned short* __restrict__ vo) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long ng = B * 64 * 64;
    if (gid < ng) {
        // G row rho of tile T: lane half (rho >> 2) & 1, register r = (rho & 3) + 4 (rho >> 3): head 4 T + 2 half + (r >> 3), token r & 7
        const int q4 = (int)(gid & 63), row = (int)((gid >> 6) & 63);
        const long p = gid >> 12;
        const int T = row >> 5, rho = row & 31, half = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3);
        const int h = 4 * T + 2 * half + (r >> 3), j = r & 7;
        // slot q4 of the row's 64: k-step s = q4 >> 2, lane half (q4 >> 1) & 1, first / second group: channels 16 s + 4 lh + 8 (q4 & 1) + 0..3
        const int ks = q4 >> 2, lh = (q4 >> 1) & 1, c0 = 16 * ks + 4 * lh + 8 * (q4 & 1);
        float4 acc = zero4();
        if (j < Tk) {
            const float* kp = tk + p * tok_bs + (long)j * ldt + h * 16;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float kc = kp[c];
                const float4 wv = ld4(wq + (long)(h * 16 + c) * 256 + c0);
                acc.x = fmaf(kc, wv.x, acc.x); acc.y = fmaf(kc, wv.y, acc.y); acc.z = fmaf(kc, wv.z, acc.z); acc.w = fmaf(kc, wv.w, acc.w);
            }
            acc.x /= denom; acc.y /= denom; acc.z /= denom; acc.w /= denom;
        }
        uint2 hi, lo;
        sp_split4(acc, I2F_GSCALE, hi, lo);
        unsigned short* dst = g + (p * 64 + row) * 512 + ks * 16 + lh * 8 + (q4 & 1) * 4;
        *(uint2*)dst = hi; *(uint2*)(dst + 256) = lo;
        if (q4 == 0) {
            float b = -3.0e38f;
            if (j < Tk) {
                const float* kp = tk + p * tok_bs + (long)j * ldt + h * 16;
                b = 0.f;
#pragma unroll
                for (int c = 0; c < 16; ++c) b = fmaf(kp[c], bq[h * 16 + c], b);
                b /= denom;
            }
            beta[p * 64 + row] = b;
        }
        return;
    }
    const long gv = gid - ng;
    if (gv >= B * 256 * 16) return;
    // VO^T row c_out, slots [s][half][j]: head 4 (s >> 1) + 2 half + (s & 1); this thread: four consecutive j of one (s, half)
    const int q4 = (int)(gv & 15), co = (int)((gv >> 4) & 255);
    const long p = gv >> 12;
    const int ks = q4 >> 2, lh = (q4 >> 1) & 1, j0 = (q4 & 1) * 4;
    const int h = 4 * (ks >> 1) + 2 * lh + (ks & 1);
    float v4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = j0 + jj;
        if (j < Tk) {
            const float* vp = tv + p * tok_bs + (long)j * ldt + h * 16;
            const float* wp = wo + (long)co * 128 + h * 16;
            float a0 = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) a0 = fmaf(wp[c], vp[c], a0);
            v4[jj] = a0;
        }
    }
    uint2 hi, lo;
    sp_split4(make_float4(v4[0], v4[1], v4[2], v4[3]), I2F_VSCALE, hi, lo);
    unsigned short* dst = vo + (p * 256 + co) * 128 + ks * 16 + lh * 8 + j0;
    *(uint2*)dst = hi; *(uint2*)(dst + 64) = lo;
}

__global__ __launch_bounds__(256, 1) void s16_i2t_kernel(SI2FArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char i2_lds[];
    unsigned char* const gs = i2_lds;                                   // G   [64][I2F_GP]
    unsigned char* const vs = gs + 64 * I2F_GP;                         // VO^T [256][I2F_VP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    float* const st = (float*)(vs + 256 * I2F_VP + w * 32 * I2F_SP);    // wave-private [32][36] floats
    float* const par = (float*)(vs + 256 * I2F_VP + 4 * 32 * I2F_SP);   // beta (64) | bo (256) | ln weight (256) | ln bias (256) (+ pad)
    const long p = blockIdx.x;
    // stage the prompt's operands: 16-byte pieces
    {
        const unsigned char* gsrc = (const unsigned char*)(a.g + p * 64 * 512);
        for (int i = tid; i < 64 * 64; i += 256) *(uint4*)(gs + (i >> 6) * I2F_GP + (i & 63) * 16) = *(const uint4*)(gsrc + (long)i * 16);
        const unsigned char* vsrc = (const unsigned char*)(a.vo + p * 256 * 128);
        for (int i = tid; i < 256 * 16; i += 256) *(uint4*)(vs + (i >> 4) * I2F_VP + (i & 15) * 16) = *(const uint4*)(vsrc + (long)i * 16);
        if (tid < 64) par[tid] = a.beta[p * 64 + tid];
        par[64 + tid] = a.bo[tid]; par[320 + tid] = a.lnw[tid]; par[576 + tid] = a.lnb[tid];
    }
    __syncthreads();
    const float* const xin = a.keys + p * a.key_bs;
    float* const outp = a.out + p * 4096L * 256;
#pragma unroll 1
    for (int tile = w; tile < 128; tile += 4) {
        const int t0 = tile * 32;
        // the lane's half of its row: x[s][e] = channel 16 s + 4 lh + (e & 3) + 8 (e >> 2), kept for the residual
        float x[16][8];
        const float* xr = xin + (long)(t0 + li) * 256 + 4 * lh;
        const float* pr = a.pos + (long)(t0 + li) * 256 + 4 * lh;
#pragma unroll
        for (int s8 = 0; s8 < 16; ++s8) {
            const float4 u = ld4(xr + 16 * s8), v = ld4(xr + 16 * s8 + 8);
            x[s8][0] = u.x; x[s8][1] = u.y; x[s8][2] = u.z; x[s8][3] = u.w; x[s8][4] = v.x; x[s8][5] = v.y; x[s8][6] = v.z; x[s8][7] = v.w;
        }
        // ---- S^T = G (x + pos)^T, two tiles of 32 hj; hi x hi and the cross terms in separate accumulators (s16_t2i_kernel)
        f32x16_t sc[2], scl[2];
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sc[T][r] = 0.f; scl[T][r] = 0.f; }
#pragma unroll
        for (int s8 = 0; s8 < 16; ++s8) {
            const float4 pu = ld4(pr + 16 * s8), pv = ld4(pr + 16 * s8 + 8);
            float xp[8] = {x[s8][0] + pu.x, x[s8][1] + pu.y, x[s8][2] + pu.z, x[s8][3] + pu.w, x[s8][4] + pv.x, x[s8][5] + pv.y, x[s8][6] + pv.z, x[s8][7] + pv.w};
            uint4 bh, bl;
            sp_split8(xp, 1.0f, bh, bl);
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                const unsigned char* ga = gs + (T * 32 + li) * I2F_GP + s8 * 32 + lh * 16;
                const uint4 ah = *(const uint4*)ga, al = *(const uint4*)(ga + 512);
                scl[T] = mfma32h(al, bh, scl[T]); scl[T] = mfma32h(ah, bl, scl[T]); sc[T] = mfma32h(ah, bh, sc[T]);
            }
        }
        // ---- softmax over the 8 tokens of each head (registers 8 g .. 8 g + 7 of a tile), in units of ln 2; probabilities -> B operands
        uint4 ph[4], pl[4];
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float v[8];
                float mx = -3.0e38f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = 8 * g + e, rho = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    v[e] = ((sc[T][r] + scl[T][r]) * (1.0f / I2F_GSCALE) + par[T * 32 + rho]) * SP_LOG2E;
                    mx = fmaxf(mx, v[e]);
                }
                float l = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] = __builtin_amdgcn_exp2f(v[e] - mx); l += v[e]; }
                const float il = SP_PSCALE / l;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= il;
                sp_split8(v, 1.0f, ph[2 * T + g], pl[2 * T + g]);
            }
        // ---- out^T = VO^T P
        f32x16_t oa[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) oa[ct][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                const unsigned char* va = vs + (ct * 32 + li) * I2F_VP + ks * 32 + lh * 16;
                const uint4 ah = *(const uint4*)va, al = *(const uint4*)(va + 128);
                oa[ct] = mfma32h(al, ph[ks], oa[ct]); oa[ct] = mfma32h(ah, pl[ks], oa[ct]); oa[ct] = mfma32h(ah, ph[ks], oa[ct]);
            }
        // ---- (+ bias) + residual (the x registers: channel 32 ct + (r & 3) + 8 (r >> 2) + 4 lh = x[2 ct + (r >> 3)][(r & 3) + 4 ((r >> 2) & 1)]), LayerNorm
        const float io = 1.0f / (I2F_VSCALE * SP_PSCALE);
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 bb = ld4(&par[64 + ct * 32 + 8 * g4 + 4 * lh]);
                const int s8 = 2 * ct + (g4 >> 1), e0 = 4 * (g4 & 1);
                oa[ct][4 * g4] = (oa[ct][4 * g4] * io + bb.x) + x[s8][e0]; oa[ct][4 * g4 + 1] = (oa[ct][4 * g4 + 1] * io + bb.y) + x[s8][e0 + 1];
                oa[ct][4 * g4 + 2] = (oa[ct][4 * g4 + 2] * io + bb.z) + x[s8][e0 + 2]; oa[ct][4 * g4 + 3] = (oa[ct][4 * g4 + 3] * io + bb.w) + x[s8][e0 + 3];
                sum += (oa[ct][4 * g4] + oa[ct][4 * g4 + 1]) + (oa[ct][4 * g4 + 2] + oa[ct][4 * g4 + 3]);
            }
        { const float o = __shfl_xor(sum, 32); sum = lh ? o + sum : sum + o; }
        const float mean = sum * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = oa[ct][r] - mean; sq = fmaf(d, d, sq); }
        { const float o = __shfl_xor(sq, 32); sq = lh ? o + sq : sq + o; }
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 256.0f) + a.eps);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c = ct * 32 + 8 * g4 + 4 * lh;
                const float4 ww = ld4(&par[320 + c]), bb = ld4(&par[576 + c]);
                *(float4*)&st[li * (I2F_SP / 4) + 8 * g4 + 4 * lh] =
                    make_float4((oa[ct][4 * g4] - mean) * rstd * ww.x + bb.x, (oa[ct][4 * g4 + 1] - mean) * rstd * ww.y + bb.y,
                                (oa[ct][4 * g4 + 2] - mean) * rstd * ww.z + bb.z, (oa[ct][4 * g4 + 3] - mean) * rstd * ww.w + bb.w);
            }
            si_wave_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i, row = idx >> 3, c4 = (idx & 7) * 4;
                *(float4*)(outp + (long)(t0 + row) * 256 + ct * 32 + c4) = ld4(&st[row * (I2F_SP / 4) + c4]);
            }
            si_wave_sync();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ decoder attention
// q [B, Nq, H*D] (row stride ldq, batch stride sqb; 0 = shared by every batch entry), k / v [B, Nk, H*D], out [B, Nq, H*D] rows ldo.
// scores = (q . k) / denom (upstream: attn / sqrt(c_per_head)), softmax, @ v.
struct SAttnArgs {
    const float* q; long ldq, sqb; const float* k; long ldk, skb; const float* v; long ldv, svb; float* out; long ldo, sob;
    int B, H, Nq, Nk; float denom;
};

// Nk <= 16 (image -> token attention, self attention of the tokens): one thread per (batch, query, head)
template <int D>
__global__ __launch_bounds__(256) void sattn_short_kernel(SAttnArgs a) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.B * a.Nq * a.H;
    if (gid >= total) return;
    const int h = (int)(gid % a.H);
    const long bi = gid / a.H;
    const int i = (int)(bi % a.Nq);
    const long b = bi / a.Nq;
    const float* qp = a.q + b * a.sqb + (long)i * a.ldq + h * D;
    float q[D];
#pragma unroll
    for (int d = 0; d < D; d += 4) { const float4 t = ld4(qp + d); q[d] = t.x; q[d + 1] = t.y; q[d + 2] = t.z; q[d + 3] = t.w; }
    float s[16];
    float m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        s[j] = -3.0e38f;
        if (j < a.Nk) {
            const float* kp = a.k + b * a.skb + (long)j * a.ldk + h * D;
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const float4 t = ld4(kp + d);
                dot = fmaf(q[d], t.x, dot); dot = fmaf(q[d + 1], t.y, dot); dot = fmaf(q[d + 2], t.z, dot); dot = fmaf(q[d + 3], t.w, dot);
            }
            s[j] = dot / a.denom;
            m = fmaxf(m, s[j]);
        }
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { s[j] = j < a.Nk ? expf(s[j] - m) : 0.f; l += s[j]; }
    float o[D];
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < a.Nk) {
            const float p = s[j] / l;
            const float* vp = a.v + b * a.svb + (long)j * a.ldv + h * D;
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const float4 t = ld4(vp + d);
                o[d] = fmaf(p, t.x, o[d]); o[d + 1] = fmaf(p, t.y, o[d + 1]); o[d + 2] = fmaf(p, t.z, o[d + 2]); o[d + 3] = fmaf(p, t.w, o[d + 3]);
            }
        }
    float* op = a.out + b * a.sob + (long)i * a.ldo + h * D;
#pragma unroll
    for (int d = 0; d < D; d += 4) *(float4*)(op + d) = make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]);
}

// Nq <= 16, long key side (token -> image attention over the 4096 image tokens): one workgroup per (batch, head); thread (t, slice) walks
// the keys slice, slice + NS, ... for query t with an online softmax; the NS partial (m, l, acc) of a query are merged through LDS.
// ALLH (H == 8, head dim 16, <= 8 queries): one workgroup per batch entry - thread = (query t, head h, slice of 4) - so that the workgroup reads
// whole 512-byte k / v rows (8 heads x 64 bytes) instead of one 64-byte piece of every row.  MEASURED SLOWER (1221 vs 691 us per 512 prompts: a
// thread's serial chain of 1024 keys with two expf each outweighs the better access pattern); kept behind msam_tune_set("sattn_allh", 1).
template <int D, int NQP, bool ALLH = false>
__global__ __launch_bounds__(256) void sattn_long_kernel(SAttnArgs a) {
    constexpr int NS = ALLH ? 256 / (NQP * 8) : 256 / NQP;
    __shared__ float red_m[256], red_l[256];
    __shared__ __attribute__((aligned(16))) float red_o[256 * D];
    const int tid = threadIdx.x, t = tid % NQP, sl = ALLH ? tid / (NQP * 8) : tid / NQP;
    const int h = ALLH ? (tid / NQP) & 7 : blockIdx.x % a.H;
    const long b = ALLH ? blockIdx.x : blockIdx.x / a.H;
    const bool act = t < a.Nq;
    float q[D], o[D];
    {
        const float* qp = a.q + b * a.sqb + (long)(act ? t : 0) * a.ldq + h * D;
#pragma unroll
        for (int d = 0; d < D; d += 4) { const float4 x = ld4(qp + d); q[d] = x.x; q[d + 1] = x.y; q[d + 2] = x.z; q[d + 3] = x.w; }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = 0.f;
    float m = -3.0e38f, l = 0.f;
    if (act)
        for (int j = sl; j < a.Nk; j += NS) {
            const float* kp = a.k + b * a.skb + (long)j * a.ldk + h * D;
            const float* vp = a.v + b * a.svb + (long)j * a.ldv + h * D;
            float dot = 0.f;
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const float4 x = ld4(kp + d);
                dot = fmaf(q[d], x.x, dot); dot = fmaf(q[d + 1], x.y, dot); dot = fmaf(q[d + 2], x.z, dot); dot = fmaf(q[d + 3], x.w, dot);
            }
            const float s = dot / a.denom;
            const float mn = fmaxf(m, s), alpha = expf(m - mn), p = expf(s - mn);
            l = l * alpha + p;
#pragma unroll
            for (int d = 0; d < D; d += 4) {
                const float4 x = ld4(vp + d);
                o[d] = fmaf(p, x.x, o[d] * alpha); o[d + 1] = fmaf(p, x.y, o[d + 1] * alpha);
                o[d + 2] = fmaf(p, x.z, o[d + 2] * alpha); o[d + 3] = fmaf(p, x.w, o[d + 3] * alpha);
            }
            m = mn;
        }
    red_m[tid] = m; red_l[tid] = l;
#pragma unroll
    for (int d = 0; d < D; ++d) red_o[tid * D + d] = o[d];
    __syncthreads();
    if constexpr (ALLH) {
        // thread index = (sl * 8 + h) * NQP + t: the NS partial results of (t, h) sit NQP * 8 apart
        for (int idx = tid; idx < a.Nq * 8 * D; idx += 256) {
            const int d = idx % D, hh = (idx / D) & 7, tt = idx / (D * 8);
            const int base = hh * NQP + tt;
            float M = -3.0e38f;
            for (int s2 = 0; s2 < NS; ++s2) M = fmaxf(M, red_m[s2 * NQP * 8 + base]);
            float L = 0.f, O = 0.f;
            for (int s2 = 0; s2 < NS; ++s2) {
                const float wgt = expf(red_m[s2 * NQP * 8 + base] - M);
                L = fmaf(red_l[s2 * NQP * 8 + base], wgt, L);
                O = fmaf(red_o[(s2 * NQP * 8 + base) * D + d], wgt, O);
            }
            a.out[b * a.sob + (long)tt * a.ldo + hh * D + d] = O / L;
        }
        return;
    }
    for (int idx = tid; idx < a.Nq * D; idx += 256) {
        const int tt = idx / D, d = idx % D;
        float M = -3.0e38f;
        for (int s2 = 0; s2 < NS; ++s2) M = fmaxf(M, red_m[s2 * NQP + tt]);
        float L = 0.f, O = 0.f;
        for (int s2 = 0; s2 < NS; ++s2) {
            const float wgt = expf(red_m[s2 * NQP + tt] - M);
            L = fmaf(red_l[s2 * NQP + tt], wgt, L);
            O = fmaf(red_o[(s2 * NQP + tt) * D + d], wgt, O);
        }
        a.out[b * a.sob + (long)tt * a.ldo + h * D + d] = O / L;
    }
}

// ------------------------------------------------------------------------------------------------------------------ gathers
// fp32 [B,3,1024,1024] -> [B*4096, 768] (c, ky, kx);  uint8 HWC [B,h,w,3] -> the same with Sam.preprocess ((x - mean) / std, zero pad)
__global__ __launch_bounds__(256) void spatchify_kernel(const float* __restrict__ img, const uint8_t* __restrict__ img8, int B, int h, int w,
                                                        float* __restrict__ out) {
    const float mean[3] = {123.675f, 116.28f, 103.53f};
    const float stdv[3] = {58.395f, 57.12f, 57.375f};
    const long total = (long)B * 4096 * 192;                  // 768 / 4 chunks per patch row
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int chunk = (int)(i % 192);
        const long prow = i / 192;
        const int b = (int)(prow / 4096), pidx = (int)(prow % 4096);
        const int py = pidx >> 6, px = pidx & 63;
        const int c = chunk >> 6, ky = (chunk >> 2) & 15, kx0 = (chunk & 3) * 4;
        const int y = py * 16 + ky, x0 = px * 16 + kx0;
        float4 v;
        if (img) v = ld4(img + (((long)b * 3 + c) * 1024 + y) * 1024 + x0);
        else {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x0 + j;
                t[j] = (y < h && x < w) ? __fdiv_rn((float)img8[(((long)b * h + y) * w + x) * 3 + c] - mean[c], stdv[c]) : 0.f;
            }
            v = make_float4(t[0], t[1], t[2], t[3]);
        }
        *(float4*)(out + prow * 768 + chunk * 4) = v;
    }
}
// x fp32 [B,64,64,C] -> [B*4096, 9*C], column (ky*3+kx)*C + c, zero padding 1
__global__ __launch_bounds__(256) void sim2col_kernel(const float* __restrict__ x, int B, int C, float* __restrict__ out) {
    const int cpr = 9 * C / 4;
    const long total = (long)B * 4096 * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int chunk = (int)(i % cpr);
        const long row = i / cpr;
        const int b = (int)(row / 4096), t = (int)(row % 4096);
        const int ty = t >> 6, tx = t & 63;
        const int tap = (chunk * 4) / C, c0 = chunk * 4 - tap * C;
        const int yy = ty + tap / 3 - 1, xx = tx + tap % 3 - 1;
        float4 val = zero4();
        if (yy >= 0 && yy < 64 && xx >= 0 && xx < 64) val = ld4(x + (((long)b * 64 + yy) * 64 + xx) * C + c0);
        *(float4*)(out + row * (9L * C) + chunk * 4) = val;
    }
}
// src[p][t][c] = emb[c][t] + dense[p][c][t]  (dense_stride 0: the broadcast no_mask_embed vector dense[c])
__global__ __launch_bounds__(256) void ssrc_kernel(const float* __restrict__ emb, const float* __restrict__ dense, long dense_stride,
                                                   float* __restrict__ src) {
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = (blockIdx.y & 7) * 32, p = blockIdx.y >> 3;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const long o = (long)(c0 + i) * 4096 + t0 + tx;
        tile[i][tx] = emb[o] + (dense_stride ? dense[(long)p * dense_stride + o] : dense[c0 + i]);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) src[((long)p * 4096 + t0 + i) * 256 + c0 + tx] = tile[tx][i];
}
// up [P*4096*4, 128]: row (p, token, s1 = ky*2+kx), column s2*32 + c2 (the two 2 x 2 stride-2 transposed convolutions as per-pixel linear
// maps); low[p][m][4 ty + 2 ky + ky2][4 tx + 2 kx + kx2] = sum_c hyper[p][mask0 + m][c] * up[...][s2*32 + c]
__global__ __launch_bounds__(256) void shyper_kernel(const float* __restrict__ up, const float* __restrict__ hyper, int hyper_ld, int mask0,
                                                     int nmask, long P, float* __restrict__ low) {
    __shared__ float hs[4 * 32];
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long p = gid >> 16;                                       // 65536 pixels per prompt: a block never straddles two prompts
    if (threadIdx.x < nmask * 32) hs[threadIdx.x] = hyper[(p * 4 + mask0 + threadIdx.x / 32) * hyper_ld + (threadIdx.x & 31)];
    __syncthreads();
    if (p >= P) return;
    const int s2 = (int)(gid & 3), s1 = (int)((gid >> 2) & 3), tok = (int)((gid >> 4) & 4095);
    const float* u = up + gid * 32;
    float x[32];
#pragma unroll
    for (int c = 0; c < 32; c += 4) { const float4 t = ld4(u + c); x[c] = t.x; x[c + 1] = t.y; x[c + 2] = t.z; x[c + 3] = t.w; }
    const int y = 4 * (tok >> 6) + 2 * (s1 >> 1) + (s2 >> 1), xx = 4 * (tok & 63) + 2 * (s1 & 1) + (s2 & 1);
    for (int mk = 0; mk < nmask; ++mk) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) acc = fmaf(hs[mk * 32 + c], x[c], acc);
        low[((p * nmask + mk) * 256 + y) * 256 + xx] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------------------ up-scaling, second half (split16)
// LayerNorm2d -> GELU -> ConvTranspose2d(64 -> 32, 2 x 2, stride 2) -> GELU -> hyper_in @ upscaled -> the un-shuffled low-res masks in ONE launch
// (upstream MaskDecoder.predict_masks: output_upscaling[1:] and `masks = (hyper_in @ upscaled_embedding.view(b, c, h * w))`).  As separate launches
// (sln64, sgemm, shyper) the 4.3 GB first-stage stream of a 1024-prompt tile is read and written by the LayerNorm, read by the product, whose 8.6 GB
// result is written and read again by the hyper product: 34 GB per tile; here the stream is read once and 0.8 GB of masks are written.
// Row = (prompt, token, s1): one of the four 64-channel pixels the first transposed convolution makes of a token.  TRANSPOSED product as in
// si2t_kernel: the weights are the A operand (row n = s2 * 32 + c2), the rows are the B operand, so a LANE is a ROW in the D layout and the lane
// pair (row, half 0 / 1) holds the row's 128 outputs: the hyper product is register-local (one exchange per mask).  The B operand comes straight
// from global memory: lane (row, half) needs channels 16 ks + 8 half .. + 7 of its row for k-step ks - 32 of the row's 64 values, the pair holds
// the row (LayerNorm statistics: one exchange each).  No LDS traffic for the stream, no workgroup barrier in the loop; W2 (fp16 pair, 34 KB)
// and the small vectors are staged once per workgroup.  Products on fp16 operand pairs (split16); LayerNorm, both erf GELUs (gelu_p8: the erf
// form to fp32 rounding) and the hyper product in fp32.
struct SUp2Args {
    const float* u1;                                    // [P * 16384, 64] first-stage rows (before the LayerNorm)
    const float* lnw; const float* lnb; float eps;
    const float* w2; const float* b2; float w_scale;    // [128, 64] rows (ky2, kx2, c2), [128]
    const float* hyper; int hyper_ld, mask0, nmask;     // [P, 4, hyper_ld]
    float* low;                                         // [P, nmask, 256, 256]
    long P;
};
constexpr int SU_WP = 272;                              // W2 row pitch in bytes: 64 hi | 64 lo halves + 16
constexpr int SU_TILES = 4;                             // 32-row tiles per wave and workgroup (a workgroup: 512 rows of one prompt)

__global__ __launch_bounds__(256, 4) void s16_up2_kernel(SUp2Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char w2s[128 * SU_WP];
    __shared__ __attribute__((aligned(16))) float vec[64 + 64 + 128 + 128];      // LayerNorm weight | bias | b2 | hyper [4][32]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, lh = lane >> 5;
    const long p = blockIdx.x >> 5;                     // 32 workgroups of 512 rows per prompt
    const int blk = (int)(blockIdx.x & 31);
    // W2 -> fp16 pairs: thread = (row n = tid / 2, 32 columns)
    {
        const int n = tid >> 1, c0 = (tid & 1) * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint2 h, l;
            sp_split4(ld4(a.w2 + n * 64 + c0 + 4 * j), a.w_scale, h, l);
            *(uint2*)(w2s + n * SU_WP + (c0 + 4 * j) * 2) = h;
            *(uint2*)(w2s + n * SU_WP + 128 + (c0 + 4 * j) * 2) = l;
        }
        if (tid < 64) { vec[tid] = a.lnw[tid]; vec[64 + tid] = a.lnb[tid]; }
        if (tid < 128) {
            vec[128 + tid] = a.b2[tid];
            const int m = tid >> 5;
            vec[256 + tid] = m < a.nmask ? a.hyper[(p * 4 + a.mask0 + m) * a.hyper_ld + (tid & 31)] : 0.f;
        }
    }
    __syncthreads();
    const float inv = 1.0f / a.w_scale;
#pragma unroll 1
    for (int t = 0; t < SU_TILES; ++t) {
        const int r0 = blk * 512 + (t * 4 + w) * 32;    // row of lane 0 inside the prompt (16384 rows)
        const float* xr = a.u1 + ((long)p * 16384 + r0 + li) * 64 + lh * 8;
        f32x2_t x[16];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 u = ld4(xr + ks * 16), v = ld4(xr + ks * 16 + 4);
            x[4 * ks] = f32x2_t{u.x, u.y}; x[4 * ks + 1] = f32x2_t{u.z, u.w}; x[4 * ks + 2] = f32x2_t{v.x, v.y}; x[4 * ks + 3] = f32x2_t{v.z, v.w};
        }
        // LayerNorm2d over the pixel's 64 channels (two-pass statistics as sln64_kernel), erf GELU
        f32x2_t s2v = x[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) s2v += x[i];
        float sm = s2v.x + s2v.y;
        { const float o = __shfl_xor(sm, 32); sm = lh ? o + sm : sm + o; }
        const float mean = sm * (1.0f / 64.0f);
        f32x2_t sqv = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) { x[i] -= mean; sqv += x[i] * x[i]; }
        float sq = sqv.x + sqv.y;
        { const float o = __shfl_xor(sq, 32); sq = lh ? o + sq : sq + o; }
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 64.0f) + a.eps);
        uint4 bh[4], bl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int c = ks * 16 + lh * 8;
            const float4 w0 = ld4(&vec[c]), w1 = ld4(&vec[c + 4]), b0 = ld4(&vec[64 + c]), b1 = ld4(&vec[64 + c + 4]);
            const f32x2_t g0 = gelu_p8x2(x[4 * ks] * rstd * f32x2_t{w0.x, w0.y} + f32x2_t{b0.x, b0.y});
            const f32x2_t g1 = gelu_p8x2(x[4 * ks + 1] * rstd * f32x2_t{w0.z, w0.w} + f32x2_t{b0.z, b0.w});
            const f32x2_t g2 = gelu_p8x2(x[4 * ks + 2] * rstd * f32x2_t{w1.x, w1.y} + f32x2_t{b1.x, b1.y});
            const f32x2_t g3 = gelu_p8x2(x[4 * ks + 3] * rstd * f32x2_t{w1.z, w1.w} + f32x2_t{b1.z, b1.w});
            uint2 h0, l0, h1, l1;
            sp_split4(make_float4(g0.x, g0.y, g1.x, g1.y), 1.0f, h0, l0); sp_split4(make_float4(g2.x, g2.y, g3.x, g3.y), 1.0f, h1, l1);
            bh[ks] = uint4{h0.x, h0.y, h1.x, h1.y}; bl[ks] = uint4{l0.x, l0.y, l1.x, l1.y};
            __builtin_amdgcn_sched_barrier(0);          // (one k-step's eight GELUs at a time: the fully interleaved form spilled 180 registers)
        }
        const int row = r0 + li, tok = row >> 2, s1 = row & 3;
        const int y0 = 4 * (tok >> 6) + 2 * (s1 >> 1), x0 = 4 * (tok & 63) + 2 * (s1 & 1);
        // the second transposed convolution, one sub-pixel row (ky2 = hp: s2 = 2 hp, 2 hp + 1) at a time: D[n = s2 * 32 + c2][row]; 32 instead of 64
        // accumulator registers live -> three waves per SIMD (the kernel is bound by its vector instructions: more waves, more issue slots used)
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            f32x16_t acc[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q2][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const unsigned char* qw = w2s + ((2 * hp + q2) * 32 + li) * SU_WP + ks * 32 + lh * 16;
                    const uint4 wh = *(const uint4*)qw, wl = *(const uint4*)(qw + 128);
                    acc[q2] = mfma32h(wl, bh[ks], acc[q2]); acc[q2] = mfma32h(wh, bl[ks], acc[q2]); acc[q2] = mfma32h(wh, bh[ks], acc[q2]);
                }
            // + bias, GELU, hyper product over the lane pair's 32 channels of the sub-pixel; register r = channel (r & 3) + 8 (r >> 2) + 4 lh
            float res[4][2];                             // [mask][kx2]
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                f32x2_t part[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int c = 8 * g4 + 4 * lh;
                    const float4 bb = ld4(&vec[128 + (2 * hp + q2) * 32 + c]);
                    const f32x2_t v01 = gelu_p8x2(f32x2_t{acc[q2][4 * g4], acc[q2][4 * g4 + 1]} * inv + f32x2_t{bb.x, bb.y});
                    const f32x2_t v23 = gelu_p8x2(f32x2_t{acc[q2][4 * g4 + 2], acc[q2][4 * g4 + 3]} * inv + f32x2_t{bb.z, bb.w});
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float4 hh = ld4(&vec[256 + m * 32 + c]);
                        part[m] += f32x2_t{hh.x, hh.y} * v01; part[m] += f32x2_t{hh.z, hh.w} * v23;
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float pm = part[m].x + part[m].y;
                    const float o = __shfl_xor(pm, 32);
                    res[m][q2] = lh ? o + pm : pm + o;
                }
            }
            // the lane pair splits the masks: half 0 stores masks 0 and 2, half 1 masks 1 and 3 (two adjacent pixels kx2 = 0, 1 per store)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int m = 2 * mm + lh;
                if (m < a.nmask) {
                    float* o = a.low + (((long)p * a.nmask + m) * 256 + y0 + hp) * 256 + x0;
                    *(float2*)o = make_float2(lh ? res[2 * mm + 1][0] : res[2 * mm][0], lh ? res[2 * mm + 1][1] : res[2 * mm][1]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ AIS decoder pieces
// InstanceNorm2d (no affine; torch.nn.InstanceNorm2d of torch_em's ConvBlock2d) on channels-last x [B, HW, C]: statistics per (b, c) over the
// HW pixels.  Pass 1: per chunk of pixels the mean and the centred sum of squares (two sweeps over the chunk: no E[x^2] - mean^2
// cancellation), pass 2: Chan's merge of the chunk statistics in double -> mean, 1 / sqrt(var + eps), pass 3: (x - mean) * rstd.
constexpr int IN_CHUNK = 2048;
__global__ __launch_bounds__(256) void sinorm_chunk_kernel(const float* __restrict__ x, long ldx, int HW, int C, float* __restrict__ part) {
    __shared__ float red[256 * 4];
    const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x, nchunk = gridDim.x;
    const int CW = C < 256 ? C : 256, SUB = 256 / CW;
    const int c0 = tid % CW, sub = tid / CW;
    const bool act = sub < SUB;
    const int p0 = chunk * IN_CHUNK, p1 = p0 + IN_CHUNK < HW ? p0 + IN_CHUNK : HW, n = p1 - p0;
    const float* xb = x + ((long)b * HW) * ldx;
    float mean[4];
    for (int pass = 0; pass < 2; ++pass) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (act)
            for (int p = p0 + sub; p < p1; p += SUB)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + i * CW;
                    if (c < C) { const float v = xb[(long)p * ldx + c]; acc[i] += pass ? (v - mean[i]) * (v - mean[i]) : v; }
                }
#pragma unroll
        for (int i = 0; i < 4; ++i) red[tid * 4 + i] = act ? acc[i] : 0.f;
        __syncthreads();
        float tot[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < SUB; ++s2)
#pragma unroll
            for (int i = 0; i < 4; ++i) tot[i] += red[(s2 * CW + c0) * 4 + i];
        __syncthreads();
        if (pass == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) mean[i] = tot[i] / (float)n;
        } else if (act && sub == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c0 + i * CW;
                if (c < C) { float* o = part + (((long)b * nchunk + chunk) * C + c) * 2; o[0] = mean[i]; o[1] = tot[i]; }
            }
        }
    }
}
__global__ __launch_bounds__(256) void sinorm_merge_kernel(const float* __restrict__ part, int nchunk, int HW, int C, int B, float eps,
                                                           float* __restrict__ stats) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx % C;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int k = 0; k < nchunk; ++k) {
        const float* o = part + (((long)b * nchunk + k) * C + c) * 2;
        const int p0 = k * IN_CHUNK;
        const double nk = (double)((p0 + IN_CHUNK < HW ? p0 + IN_CHUNK : HW) - p0), d = (double)o[0] - mean, nt = n + nk;
        mean += d * nk / nt;
        m2 += (double)o[1] + d * d * n * nk / nt;
        n = nt;
    }
    stats[2 * idx] = (float)mean;
    stats[2 * idx + 1] = 1.0f / sqrtf((float)(m2 / n) + eps);
}
__global__ __launch_bounds__(256) void sinorm_apply_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ stats, long HW, int C,
                                                           long total4, float* __restrict__ out) {
    const int c4n = C / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        const long row = i / c4n, b = row / HW;
        const float4 v = ld4(x + row * ldx + c);
        const float* st = stats + (b * C + c) * 2;
        *(float4*)(out + row * C + c) = make_float4((v.x - st[0]) * st[1], (v.y - st[2]) * st[3], (v.z - st[4]) * st[5], (v.w - st[6]) * st[7]);
    }
}
// torch.nn.functional.interpolate(mode="bilinear", align_corners=False) on channels-last data: in [B, h (pitch_h rows), w (pitch_w pixels), C]
// (the logical h x w window of a larger image: the crop of postprocess_masks) -> out [B, H2, W2, C] or NCHW [B, C, H2, W2].
// source index = max(0, scale * (dst + 0.5) - 0.5), separately rounded (area_pixel_compute_source_index); weights 1 - l, l.
__global__ __launch_bounds__(256) void sresize_kernel(const float* __restrict__ in, int B, int h, int w, int pitch_h, int pitch_w, long pix, int C,
                                                      int H2, int W2, float sh, float sw, int nchw, float* __restrict__ out) {
    const int c4n = C / 4;
    const long total = (long)B * H2 * W2 * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % c4n) * 4;
        long r = i / c4n;
        const int x2 = (int)(r % W2); r /= W2;
        const int y2 = (int)(r % H2);
        const long b = r / H2;
        float fy = __fsub_rn(__fmul_rn(sh, (float)y2 + 0.5f), 0.5f), fx = __fsub_rn(__fmul_rn(sw, (float)x2 + 0.5f), 0.5f);
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const float* base = in + (b * pitch_h) * (long)pitch_w * pix + c;
        const float4 p00 = ld4(base + ((long)y0 * pitch_w + x0) * pix), p01 = ld4(base + ((long)y0 * pitch_w + x1) * pix);
        const float4 p10 = ld4(base + ((long)y1 * pitch_w + x0) * pix), p11 = ld4(base + ((long)y1 * pitch_w + x1) * pix);
        float o[4];
        const float a00[4] = {p00.x, p00.y, p00.z, p00.w}, a01[4] = {p01.x, p01.y, p01.z, p01.w};
        const float a10[4] = {p10.x, p10.y, p10.z, p10.w}, a11[4] = {p11.x, p11.y, p11.z, p11.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t0 = fmaf(hx, a00[k], __fmul_rn(lx, a01[k])), t1 = fmaf(hx, a10[k], __fmul_rn(lx, a11[k]));
            o[k] = fmaf(hy, t0, __fmul_rn(ly, t1));
        }
        if (nchw) {
#pragma unroll
            for (int k = 0; k < 4; ++k) out[((b * C + c + k) * H2 + y2) * (long)W2 + x2] = o[k];
        } else *(float4*)(out + ((b * H2 + y2) * (long)W2 + x2) * C + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

int grid1d(long n, int per_block) {
    const long g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 0x7fffffffL ? 0x7fffffffL : g));
}

}  // namespace

extern "C" int msam_strict_gemm(const msam_sgemm_t* p, void* stream) {
    if (!p || !p->A || !p->W || !p->out || p->M <= 0 || p->N <= 0 || p->K <= 0) { msam_set_error("msam_strict_gemm: null argument or empty shape"); return 1; }
    const bool conv = p->conv_c > 0;
    if (p->K % 4 || p->ldw % 4 || ((uintptr_t)p->A | (uintptr_t)p->W) % 16 || p->lda % 4 || (conv && p->lda < p->conv_c) || (p->A2 && (p->lda2 % 4 || (uintptr_t)p->A2 % 16))) {
        msam_set_error("msam_strict_gemm: K, lda, ldw (, lda2) must be multiples of 4 and the operands 16-byte aligned");
        return 1;
    }
    if (p->act < MSAM_ACT_NONE || p->act > MSAM_ACT_SIGMOID) { msam_set_error("msam_strict_gemm: unknown activation"); return 1; }
    if (conv && (p->conv_h <= 0 || p->conv_w <= 0 || p->conv_c % 4 || p->K != 9 * p->conv_c || p->M % ((int64_t)p->conv_h * p->conv_w) || p->A2)) {
        msam_set_error("msam_strict_gemm: 3 x 3 mode needs A = [B, H, W, C] with C % 4 == 0, K == 9 C, M == B H W and no A2");
        return 1;
    }
    if (p->shuffle_c > 0 && (p->shuffle_h <= 0 || p->shuffle_w <= 0 || p->N != 4 * p->shuffle_c || p->M % ((int64_t)p->shuffle_h * p->shuffle_w) || p->res)) {
        msam_set_error("msam_strict_gemm: 2 x 2 transposed-convolution store needs N == 4 shuffle_c, M == B shuffle_h shuffle_w and no residual");
        return 1;
    }
    if ((p->col_scale == nullptr) != (p->col_shift == nullptr)) { msam_set_error("msam_strict_gemm: col_scale and col_shift come together"); return 1; }
    if (p->a2_cols < 0 || p->a2_cols % 128) { msam_set_error("msam_strict_gemm: a2_cols is a multiple of 128 (the column tile)"); return 1; }
    SGemmArgs a{};
    a.A = p->A; a.lda = p->lda; a.A2 = p->A2; a.lda2 = p->lda2; a.a2_rows = p->a2_rows > 0 ? p->a2_rows : p->M;
    a.W = p->W; a.ldw = p->ldw; a.M = p->M; a.N = p->N; a.K = p->K; a.bias = p->bias; a.act = p->act;
    a.res = p->res; a.ldr = p->ldr; a.res_rows = p->res_rows > 0 ? p->res_rows : p->M; a.out = p->out; a.ldc = p->ldc;
    a.col_scale = p->col_scale; a.col_shift = p->col_shift;
    a.conv_h = p->conv_h; a.conv_w = p->conv_w; a.conv_c = p->conv_c;
    a.shuf_h = p->shuffle_h; a.shuf_w = p->shuffle_w; a.shuf_c = p->shuffle_c;
    a.a2_cols = p->a2_cols;
    const bool split = p->split16 != 0;
    if (p->split16 < 0 || p->split16 > 1 || p->a_scale < 0.f || p->w_scale < 0.f) { msam_set_error("msam_strict_gemm: split16 is 0 or 1, the scales are powers of two > 0 (0 = 1)"); return 1; }
    a.a_scale = p->a_scale > 0.f ? p->a_scale : 1.f; a.w_scale = p->w_scale > 0.f ? p->w_scale : 1.f;
    a.out_scale = 1.0f / (a.a_scale * a.w_scale);
    if (p->w_pairs) { msam_set_error("msam_strict_gemm: w_pairs is not supported (measured slower: profiles/r06_experiments.md); leave it NULL"); return 1; }
    long blocks = ((p->M + 127) / 128) * (long)((p->N + 127) / 128);
    if (blocks > 0x7fffffffL) { msam_set_error("msam_strict_gemm: too many tiles for one launch"); return 1; }
    const bool small = blocks < g_tune_sgemm_small_below;        // fewer 128 x 128 tiles than fill the chip: 64 x 64 tiles
    if (small) blocks = ((p->M + 63) / 64) * (long)((p->N + 63) / 64);
    const dim3 grid((unsigned)blocks), wg(256);
    hipStream_t st = (hipStream_t)stream;
    if (split) {
        if (small) {
            if (conv) hipLaunchKernelGGL((sgemm_kernel<true, 1, 1, true>), grid, wg, 0, st, a);
            else hipLaunchKernelGGL((sgemm_kernel<false, 1, 1, true>), grid, wg, 0, st, a);
        } else if (g_tune_sgemm_bufs == 1) {
            if (conv) hipLaunchKernelGGL((sgemm_kernel<true, 1, 2, true>), grid, wg, 0, st, a);
            else hipLaunchKernelGGL((sgemm_kernel<false, 1, 2, true>), grid, wg, 0, st, a);
        } else {
            if (conv) hipLaunchKernelGGL((sgemm_kernel<true, 2, 2, true>), grid, wg, 0, st, a);
            else hipLaunchKernelGGL((sgemm_kernel<false, 2, 2, true>), grid, wg, 0, st, a);
        }
    } else if (small) {
        if (conv) hipLaunchKernelGGL((sgemm_kernel<true, 1, 1>), grid, wg, 0, st, a);
        else hipLaunchKernelGGL((sgemm_kernel<false, 1, 1>), grid, wg, 0, st, a);
    } else if (g_tune_sgemm_bufs == 1) {
        if (conv) hipLaunchKernelGGL((sgemm_kernel<true, 1, 2>), grid, wg, 0, st, a);
        else hipLaunchKernelGGL((sgemm_kernel<false, 1, 2>), grid, wg, 0, st, a);
    } else {
        if (conv) hipLaunchKernelGGL((sgemm_kernel<true, 2, 2>), grid, wg, 0, st, a);
        else hipLaunchKernelGGL((sgemm_kernel<false, 2, 2>), grid, wg, 0, st, a);
    }
    return msam_check_launch("strict_gemm");
}

extern "C" int msam_strict_layernorm(const float* x, const float* weight, const float* bias, float eps, int64_t rows, int32_t dim,
                                     float* out, int32_t gelu, int32_t out_nchw_hw, void* stream) {
    if (!x || !weight || !bias || !out || rows <= 0 || dim <= 0 || dim > 1280) { msam_set_error("msam_strict_layernorm: null argument or dim > 1280"); return 1; }
    if (dim == 64 && out_nchw_hw <= 0 && ((uintptr_t)x | (uintptr_t)out | (uintptr_t)weight | (uintptr_t)bias) % 16 == 0)
        hipLaunchKernelGGL(sln64_kernel, dim3(grid1d(rows, 16)), dim3(256), 0, (hipStream_t)stream, x, weight, bias, eps, (long)rows, out, gelu);
    else
        hipLaunchKernelGGL(sln_kernel, dim3(grid1d(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, weight, bias, eps, (long)rows, dim, out, gelu,
                           out_nchw_hw);
    return msam_check_launch("strict_layernorm");
}

static int relpos_attention_impl(const float* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, int32_t B, int32_t heads,
                                 int32_t head_dim, int32_t grid, int32_t window, float scale, float* out, void* stream, bool split);
extern "C" int msam_strict_relpos_attention(const float* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, int32_t B,
                                            int32_t heads, int32_t head_dim, int32_t grid, int32_t window, float scale, float* out,
                                            void* stream) {
    return relpos_attention_impl(qkv, qkv_bias, rel_h, rel_w, B, heads, head_dim, grid, window, scale, out, stream, false);
}
extern "C" int msam_split16_relpos_attention(const float* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, int32_t B,
                                             int32_t heads, int32_t head_dim, int32_t grid, int32_t window, float scale, float* out,
                                             void* stream) {
    return relpos_attention_impl(qkv, qkv_bias, rel_h, rel_w, B, heads, head_dim, grid, window, scale, out, stream, true);
}
static int relpos_attention_impl(const float* qkv, const float* qkv_bias, const float* rel_h, const float* rel_w, int32_t B, int32_t heads,
                                 int32_t head_dim, int32_t grid, int32_t window, float scale, float* out, void* stream, bool split) {
    if (!qkv || !qkv_bias || !rel_h || !rel_w || !out || B <= 0 || heads <= 0) { msam_set_error("msam_strict_relpos_attention: null argument"); return 1; }
    if ((head_dim != 64 && head_dim != 80) || (window != 0 && window != 14) || (window == 0 && grid != 64) || grid < 1 || grid > 64) {
        msam_set_error("msam_strict_relpos_attention: head_dim 64 / 80; window 14 (any grid <= 64) or 0 = global on the 64 x 64 grid");
        return 1;
    }
    SRelArgs a{qkv, qkv_bias, rel_h, rel_w, out, B, heads, grid, window, heads * head_dim, scale};
    const int S = window ? 14 : 64;
    const int nW = window ? (grid + S - 1) / S : 1, QB = (S * S + 255) / 256;
    const long blocks = (long)B * nW * nW * heads * QB;
    const size_t lds = (size_t)(2 * S * 256 + 2 * SR_KC * head_dim) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define MSAM_SREL(HD_, S_)                                                                                                            \
    do {                                                                                                                              \
        (void)hipFuncSetAttribute((const void*)srelpos_kernel<HD_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL((srelpos_kernel<HD_, S_>), dim3((unsigned)blocks), dim3(256), lds, s, a);                                 \
    } while (0)
    if (split) {                                        // both products on fp16 operand pairs (the split16 mode)
        const size_t lw = (size_t)(224 * 32 + 32 * (head_dim + 4) + ((head_dim + 31) / 32) * 32 * SM_VP) * sizeof(float);
        const size_t lm = (size_t)(128 * SM_BWP + 32 * (head_dim + 4) + ((head_dim + 31) / 32) * 32 * SM_VP) * sizeof(float);
        const int nWw = (grid + SW_S - 1) / SW_S;
        const unsigned gw = (unsigned)(B * nWw * nWw * heads), gm = (unsigned)(B * heads * 32);
#define MSAM_SPLIT_ATT(HD_)                                                                                                                         \
        do {                                                                                                                                        \
            if (window) {                                                                                                                           \
                (void)hipFuncSetAttribute((const void*)srelpos_win_mfma_kernel<HD_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lw);  \
                hipLaunchKernelGGL((srelpos_win_mfma_kernel<HD_, true>), dim3(gw), dim3(448), lw, s, a);                                            \
            } else {                                                                                                                                \
                (void)hipFuncSetAttribute((const void*)srelpos_mfma_kernel<HD_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);       \
                hipLaunchKernelGGL((srelpos_mfma_kernel<HD_, true>), dim3(gm), dim3(256), lm, s, a);                                                \
            }                                                                                                                                       \
        } while (0)
        if (head_dim == 64) MSAM_SPLIT_ATT(64); else MSAM_SPLIT_ATT(80);
#undef MSAM_SPLIT_ATT
        return msam_check_launch("split16_relpos_attention");
    }
    if (window && g_tune_srel_mfma >= 2) {              // the 14 x 14 windows on the f32-input MFMA
        const size_t lw = (size_t)(224 * 32 + 32 * (head_dim + 4) + ((head_dim + 31) / 32) * 32 * SM_VP) * sizeof(float);
        const int nWw = (grid + SW_S - 1) / SW_S;
        const unsigned gw = (unsigned)(B * nWw * nWw * heads);
        if (head_dim == 64) {
            (void)hipFuncSetAttribute((const void*)srelpos_win_mfma_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lw);
            hipLaunchKernelGGL((srelpos_win_mfma_kernel<64>), dim3(gw), dim3(448), lw, s, a);
        } else {
            (void)hipFuncSetAttribute((const void*)srelpos_win_mfma_kernel<80>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lw);
            hipLaunchKernelGGL((srelpos_win_mfma_kernel<80>), dim3(gw), dim3(448), lw, s, a);
        }
        return msam_check_launch("strict_relpos_attention");
    }
    if (!window && g_tune_srel_mfma) {                  // the 64 x 64 grid on the f32-input MFMA
        const size_t lm = (size_t)(128 * SM_BWP + 32 * (head_dim + 4) + ((head_dim + 31) / 32) * 32 * SM_VP) * sizeof(float);
        const unsigned gm = (unsigned)(B * heads * 32);
        if (head_dim == 64) {
            (void)hipFuncSetAttribute((const void*)srelpos_mfma_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
            hipLaunchKernelGGL((srelpos_mfma_kernel<64>), dim3(gm), dim3(256), lm, s, a);
        } else {
            (void)hipFuncSetAttribute((const void*)srelpos_mfma_kernel<80>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
            hipLaunchKernelGGL((srelpos_mfma_kernel<80>), dim3(gm), dim3(256), lm, s, a);
        }
        return msam_check_launch("strict_relpos_attention");
    }
    if (head_dim == 64) { if (window) MSAM_SREL(64, 14); else MSAM_SREL(64, 64); }
    else { if (window) MSAM_SREL(80, 14); else MSAM_SREL(80, 64); }
#undef MSAM_SREL
    return msam_check_launch("strict_relpos_attention");
}

static int strict_num_cus() {
    static int cus = 0;
    if (cus <= 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}

extern "C" int msam_strict_i2t_block(const msam_si2t_t* p, void* stream) {
    if (!p || !p->keys || !p->pos || !p->wq || !p->bq || !p->tok_k || !p->tok_v || !p->wo || !p->bo || !p->ln_weight || !p->ln_bias || !p->out) {
        msam_set_error("msam_strict_i2t_block: null argument"); return 1;
    }
    if (p->B <= 0 || p->Tk <= 0 || p->Tk > 16 || p->ld_tok < 128 || p->ld_tok % 4 || p->tok_batch_stride % 4 || p->key_batch_stride % 4 ||
        ((uintptr_t)p->keys | (uintptr_t)p->pos | (uintptr_t)p->wq | (uintptr_t)p->bq | (uintptr_t)p->tok_k | (uintptr_t)p->tok_v |
         (uintptr_t)p->wo | (uintptr_t)p->bo | (uintptr_t)p->ln_weight | (uintptr_t)p->ln_bias | (uintptr_t)p->out) % 16) {
        msam_set_error("msam_strict_i2t_block: 1..16 tokens, strides in multiples of 4 floats, 16-byte aligned pointers"); return 1;
    }
    if (p->key_batch_stride == 0 && p->out == p->keys) { msam_set_error("msam_strict_i2t_block: a shared stream cannot be updated in place"); return 1; }
    SI2TArgs a{p->keys, p->key_batch_stride, p->pos, p->wq, p->bq, p->tok_k, p->tok_v, p->ld_tok, p->tok_batch_stride, p->wo, p->bo,
               p->ln_weight, p->ln_bias, p->ln_eps, p->denom, p->out, p->B, p->Tk, g_tune_si2t_dbg,
               strict_num_cus(), 2 * strict_num_cus(), g_tune_si2t_late_us * 100,
               p->wq_scale > 0.f ? p->wq_scale : 1.f, p->wo_scale > 0.f ? p->wo_scale : 1.f, (const unsigned short*)p->wq_pairs, (const unsigned short*)p->wo_pairs};
    if (((uintptr_t)p->wq_pairs | (uintptr_t)p->wo_pairs) % 16 || ((p->wq_pairs || p->wo_pairs) && !p->split16)) {
        msam_set_error("msam_strict_i2t_block: prepared weight pairs are 16-byte aligned and go with split16"); return 1;
    }
    if (p->split16 < 0 || p->split16 > 1) { msam_set_error("msam_strict_i2t_block: split16 is 0 or 1"); return 1; }
    if (p->split16) hipLaunchKernelGGL(si2t_kernel<true>, dim3((unsigned)p->B * 32u), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(si2t_kernel<false>, dim3((unsigned)p->B * 32u), dim3(256), 0, (hipStream_t)stream, a);
    return msam_check_launch("strict_i2t_block");
}

extern "C" int msam_split16_i2t_block(const msam_si2t_t* p, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!p || !p->keys || !p->pos || !p->wq || !p->bq || !p->tok_k || !p->tok_v || !p->wo || !p->bo || !p->ln_weight || !p->ln_bias || !p->out || !workspace) {
        msam_set_error("msam_split16_i2t_block: null argument"); return 1;
    }
    if (p->B <= 0 || p->Tk <= 0 || p->Tk > 8 || p->ld_tok < 128 || p->ld_tok % 4 || p->tok_batch_stride % 4 || p->key_batch_stride % 4 || p->denom <= 0.f ||
        ((uintptr_t)p->keys | (uintptr_t)p->pos | (uintptr_t)p->wq | (uintptr_t)p->wo | (uintptr_t)p->out | (uintptr_t)workspace) % 16) {
        msam_set_error("msam_split16_i2t_block: 1..8 tokens, strides in multiples of 4 floats, 16-byte aligned pointers"); return 1;
    }
    if (p->key_batch_stride == 0 && p->out == p->keys) { msam_set_error("msam_split16_i2t_block: a shared stream cannot be updated in place"); return 1; }
    const size_t gb = (size_t)p->B * 64 * 512 * 2, vb = (size_t)p->B * 256 * 128 * 2, bb = (size_t)p->B * 64 * 4;
    if ((size_t)workspace_bytes < gb + vb + bb) { msam_set_error("msam_split16_i2t_block: workspace of B x 131328 bytes needed"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    unsigned short* g = (unsigned short*)workspace;
    unsigned short* vo = (unsigned short*)((char*)workspace + gb);
    float* beta = (float*)((char*)workspace + gb + vb);
    const long nf = (long)p->B * 64 * 64 + (long)p->B * 256 * 16;
    hipLaunchKernelGGL(s16_i2t_fold_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, p->tok_k, p->tok_v, (long)p->ld_tok, (long)p->tok_batch_stride, p->Tk,
                       p->wq, p->bq, p->wo, p->denom, (long)p->B, g, beta, vo);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)s16_i2t_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, I2F_LDS) != hipSuccess) {
            msam_set_error("msam_split16_i2t_block: cannot raise the dynamic LDS limit"); return 2;
        }
        attr = true;
    }
    SI2FArgs a{p->keys, (long)p->key_batch_stride, p->pos, g, beta, vo, p->bo, p->ln_weight, p->ln_bias, p->ln_eps, p->out, p->B};
    hipLaunchKernelGGL(s16_i2t_kernel, dim3((unsigned)p->B), dim3(256), I2F_LDS, s, a);
    return msam_check_launch("split16_i2t_block");
}

extern "C" int msam_split16_prepare_pairs(const float* w, int64_t N, int32_t K, float scale, int32_t permute, void* out, void* stream) {
    if (!w || !out || N <= 0 || K <= 0 || K % 32 || scale <= 0.f || ((uintptr_t)w | (uintptr_t)out) % 16) {
        msam_set_error("msam_split16_prepare_pairs: K % 32 == 0, scale > 0, 16-byte aligned pointers"); return 1;
    }
    const long n = N * (K / 4);
    hipLaunchKernelGGL(s16_prepare_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (long)N, K, scale, permute, (unsigned short*)out);
    return msam_check_launch("split16_prepare_pairs");
}

extern "C" int msam_split16_t2i_attention(const msam_st2i_t* p, void* stream) {
    if (!p || !p->keys || !p->pos || !p->q || !p->wk || !p->wv || !p->bv || !p->out || !p->workspace || p->B <= 0) { msam_set_error("msam_split16_t2i_attention: null argument"); return 1; }
    if (p->Tk < 1 || p->Tk > 8 || p->ldq < 128 || p->ldq % 4 || p->ldo < 128 || p->key_batch_stride % 4 || p->denom <= 0.f ||
        ((uintptr_t)p->keys | (uintptr_t)p->pos | (uintptr_t)p->q | (uintptr_t)p->wk | (uintptr_t)p->wv | (uintptr_t)p->workspace) % 16) {
        msam_set_error("msam_split16_t2i_attention: 1..8 tokens, 8 heads x 16 channels, strides in multiples of 4 floats, 16-byte aligned pointers"); return 1;
    }
    if (p->workspace_bytes < (int64_t)p->B * 64 * (512 * 2 + 256 * 4)) { msam_set_error("msam_split16_t2i_attention: workspace of B x 131072 bytes needed"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    unsigned short* g = (unsigned short*)p->workspace;
    float* u = (float*)((char*)p->workspace + (size_t)p->B * 64 * 512 * 2);
    const long nf = (long)p->B * 64 * 64;
    hipLaunchKernelGGL(s16_t2i_fold_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, p->q, (long)p->ldq, p->Tk, p->wk, p->denom, nf, g);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)s16_t2i_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T2I_LDS) != hipSuccess) {
            msam_set_error("msam_split16_t2i_attention: cannot raise the dynamic LDS limit"); return 2;
        }
        attr = true;
    }
    ST2IArgs a{p->keys, (long)p->key_batch_stride, p->pos, g, u, p->B};
    hipLaunchKernelGGL(s16_t2i_kernel, dim3((unsigned)p->B), dim3(256), T2I_LDS, s, a);
    const long no = (long)p->B * p->Tk * 128;
    hipLaunchKernelGGL(s16_t2i_out_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, s, (const float*)u, p->wv, p->bv, p->Tk, no, p->out, (long)p->ldo);
    return msam_check_launch("split16_t2i_attention");
}

extern "C" int msam_strict_attention(const float* q, int64_t ldq, int64_t q_batch_stride, const float* k, int64_t ldk, int64_t k_batch_stride,
                                     const float* v, int64_t ldv, int64_t v_batch_stride, int32_t B, int32_t H, int32_t Nq, int32_t Nk,
                                     int32_t D, float denom, float* out, int64_t ldo, int64_t out_batch_stride, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) { msam_set_error("msam_strict_attention: null argument or empty shape"); return 1; }
    if ((D != 16 && D != 32) || (Nq > 16 && Nk > 16) || ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4 ||
        q_batch_stride % 4 || k_batch_stride % 4 || v_batch_stride % 4 || out_batch_stride % 4 ||
        ((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16) {
        msam_set_error("msam_strict_attention: head dim 16 or 32, one side <= 16 tokens, strides in multiples of 4 floats, 16-byte aligned");
        return 1;
    }
    SAttnArgs a{q, ldq, q_batch_stride, k, ldk, k_batch_stride, v, ldv, v_batch_stride, out, ldo, out_batch_stride, B, H, Nq, Nk, denom};
    hipStream_t s = (hipStream_t)stream;
    if (Nk <= 16) {
        const int g = grid1d((long)B * Nq * H, 256);
        if (D == 16) hipLaunchKernelGGL(sattn_short_kernel<16>, dim3(g), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(sattn_short_kernel<32>, dim3(g), dim3(256), 0, s, a);
    } else {
        const dim3 g((unsigned)((long)B * H));
        if (D == 16 && Nq <= 8 && H == 8 && g_tune_sattn_allh) hipLaunchKernelGGL((sattn_long_kernel<16, 8, true>), dim3((unsigned)B), dim3(256), 0, s, a);
        else if (D == 16) { if (Nq <= 8) hipLaunchKernelGGL((sattn_long_kernel<16, 8>), g, dim3(256), 0, s, a); else hipLaunchKernelGGL((sattn_long_kernel<16, 16>), g, dim3(256), 0, s, a); }
        else { if (Nq <= 8) hipLaunchKernelGGL((sattn_long_kernel<32, 8>), g, dim3(256), 0, s, a); else hipLaunchKernelGGL((sattn_long_kernel<32, 16>), g, dim3(256), 0, s, a); }
    }
    return msam_check_launch("strict_attention");
}

extern "C" int msam_strict_patchify(const float* img, const uint8_t* img_u8, int32_t B, int32_t h, int32_t w, float* out, void* stream) {
    if ((!img && !img_u8) || !out || B <= 0 || (img_u8 && (h <= 0 || w <= 0 || h > 1024 || w > 1024))) {
        msam_set_error("msam_strict_patchify: one of img / img_u8, 0 < h, w <= 1024");
        return 1;
    }
    hipLaunchKernelGGL(spatchify_kernel, dim3(grid1d((long)B * 4096 * 192, 256 * 4)), dim3(256), 0, (hipStream_t)stream, img, img_u8, B, h, w, out);
    return msam_check_launch("strict_patchify");
}

extern "C" int msam_strict_im2col3x3(const float* x, int32_t B, int32_t C, float* out, void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || C % 4) { msam_set_error("msam_strict_im2col3x3: null argument or C % 4 != 0"); return 1; }
    hipLaunchKernelGGL(sim2col_kernel, dim3(grid1d((long)B * 4096 * (9 * C / 4), 256 * 4)), dim3(256), 0, (hipStream_t)stream, x, B, C, out);
    return msam_check_launch("strict_im2col3x3");
}

extern "C" int msam_strict_source(const float* embedding, const float* dense, int64_t dense_stride, int32_t P, float* src, void* stream) {
    if (!embedding || !dense || !src || P <= 0 || P > 8191) { msam_set_error("msam_strict_source: null argument or more than 8191 prompts per call"); return 1; }
    hipLaunchKernelGGL(ssrc_kernel, dim3(4096 / 32, (256 / 32) * P), dim3(256), 0, (hipStream_t)stream, embedding, dense, (long)dense_stride, src);
    return msam_check_launch("strict_source");
}

extern "C" int msam_strict_hyper_masks(const float* up, const float* hyper, int32_t hyper_ld, int32_t mask0, int32_t nmask, int64_t P,
                                       float* low_res, void* stream) {
    if (!up || !hyper || !low_res || P <= 0 || nmask < 1 || mask0 < 0 || mask0 + nmask > 4 || hyper_ld < 32) {
        msam_set_error("msam_strict_hyper_masks: 1 <= nmask masks out of 4, hyper_ld >= 32");
        return 1;
    }
    hipLaunchKernelGGL(shyper_kernel, dim3((unsigned)(P * 256)), dim3(256), 0, (hipStream_t)stream, up, hyper, hyper_ld, mask0, nmask, (long)P, low_res);
    return msam_check_launch("strict_hyper_masks");
}

extern "C" int msam_strict_upscale2(const msam_sup2_t* p, void* stream) {
    if (!p || !p->u1 || !p->ln_weight || !p->ln_bias || !p->w2 || !p->b2 || !p->hyper || !p->low_res || p->P <= 0) { msam_set_error("msam_strict_upscale2: null argument"); return 1; }
    if (p->nmask < 1 || p->mask0 < 0 || p->mask0 + p->nmask > 4 || p->hyper_ld < 32 || ((uintptr_t)p->u1 | (uintptr_t)p->w2 | (uintptr_t)p->low_res) % 16 || p->P > (1L << 26)) {
        msam_set_error("msam_strict_upscale2: 1 <= nmask masks out of 4, hyper_ld >= 32, 16-byte aligned pointers"); return 1;
    }
    SUp2Args a{p->u1, p->ln_weight, p->ln_bias, p->ln_eps, p->w2, p->b2, p->w_scale > 0.f ? p->w_scale : 1.f, p->hyper, p->hyper_ld, p->mask0, p->nmask, p->low_res, (long)p->P};
    hipLaunchKernelGGL(s16_up2_kernel, dim3((unsigned)(p->P * 32)), dim3(256), 0, (hipStream_t)stream, a);
    return msam_check_launch("strict_upscale2");
}

extern "C" int msam_strict_instance_norm(const float* x, int64_t ldx, int32_t B, int64_t HW, int32_t C, float eps, float* out, float* workspace,
                                         int64_t workspace_floats, void* stream) {
    const long nchunk = (HW + IN_CHUNK - 1) / IN_CHUNK;
    const long need = 2L * B * nchunk * C + 2L * B * C;
    if (!x || !out || !workspace || B <= 0 || HW <= 0 || C <= 0 || C % 4 || C > 1024 || ldx < C || ldx % 4 || B > 65535 || HW > 0x7fffffffL) {
        msam_set_error("msam_strict_instance_norm: null argument, C % 4 != 0, C > 1024 or ldx < C");
        return 1;
    }
    if (workspace_floats < need) { msam_set_error("msam_strict_instance_norm: workspace too small (2 B ceil(HW / 2048) C + 2 B C floats)"); return 1; }
    float* part = workspace; float* stats = workspace + 2L * B * nchunk * C;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sinorm_chunk_kernel, dim3((unsigned)nchunk, B), dim3(256), 0, s, x, (long)ldx, (int)HW, C, part);
    hipLaunchKernelGGL(sinorm_merge_kernel, dim3(grid1d((long)B * C, 256)), dim3(256), 0, s, part, (int)nchunk, (int)HW, C, B, eps, stats);
    const long total4 = (long)B * HW * (C / 4);
    hipLaunchKernelGGL(sinorm_apply_kernel, dim3(grid1d(total4, 256 * 4)), dim3(256), 0, s, x, (long)ldx, stats, (long)HW, C, total4, out);
    return msam_check_launch("strict_instance_norm");
}

extern "C" int msam_strict_resize_bilinear(const float* in, int32_t B, int32_t h, int32_t w, int32_t pitch_h, int32_t pitch_w, int64_t pixel_pitch,
                                           int32_t C, int32_t H2, int32_t W2, float scale_h, float scale_w, int32_t out_nchw, float* out, void* stream) {
    if (!in || !out || B <= 0 || h <= 0 || w <= 0 || h > pitch_h || w > pitch_w || C <= 0 || C % 4 || pixel_pitch < C || pixel_pitch % 4 || H2 <= 0 ||
        W2 <= 0 || ((uintptr_t)in | (uintptr_t)out) % 16) {
        msam_set_error("msam_strict_resize_bilinear: null argument, C % 4 != 0, pixel pitch < C or window larger than the image");
        return 1;
    }
    hipLaunchKernelGGL(sresize_kernel, dim3(grid1d((long)B * H2 * W2 * (C / 4), 256 * 2)), dim3(256), 0, (hipStream_t)stream, in, B, h, w, pitch_h,
                       pitch_w, (long)pixel_pitch, C, H2, W2, scale_h, scale_w, out_nchw, out);
    return msam_check_launch("strict_resize_bilinear");
}
