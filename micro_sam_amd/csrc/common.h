// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of micro_sam_amd.
// Wave = 64 lanes.  MFMA fragment maps used everywhere (v_mfma_f32_16x16x32_bf16):
//   A operand: lane l holds A[row = l & 15][k = (l >> 4) * 8 + i], i = 0..7   (8 bf16 = 16 B)
//   B operand: lane l holds B[k = (l >> 4) * 8 + i][col = l & 15]
//   C/D      : lane l, reg r holds C[row = (l >> 4) * 4 + r][col = l & 15]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define MSAM_DEVINL __device__ __forceinline__

// round-to-nearest-even fp32 -> bf16 through the gfx950 conversion instruction (v_cvt_pk_bf16_f32, NaN stays NaN);
// the integer emulation (bias add + NaN branch) cost ~10 VALU instructions and an exec-mask branch per value
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
MSAM_DEVINL u16 f2bf(float f) { return __builtin_bit_cast(u16, (__bf16)f); }
MSAM_DEVINL float bf2f(u16 h) { return __uint_as_float(((uint32_t)h) << 16); }
MSAM_DEVINL uint32_t pack2bf(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}

// vmcnt is an in-order counter: waiting for ANY global load also waits for every load issued before it.  Registers
// that are loaded once (stationary operands) and first used inside a persistent loop make the compiler place that wait
// INSIDE the loop, where it drains the tile prefetch on every iteration.  wait_vmem_all() (s_waitcnt vmcnt(0), which the
// compiler's wait-count pass takes into account) pins the wait to where the loads are issued; the loop body then only
// waits for what it really needs.
MSAM_DEVINL void wait_vmem_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0), expcnt / lgkmcnt untouched

// Buffer addressing for tile streams: base in a scalar resource descriptor, per-lane byte offset in ONE loop-invariant
// VGPR, per-tile byte offset in an SGPR - no 64-bit VGPR address arithmetic per load (the compiler otherwise builds the
// addresses in the destination registers of the previous prefetch, and that write-after-write on a register with a load
// in flight costs a vmcnt wait at the top of every iteration).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
MSAM_DEVINL rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
MSAM_DEVINL uint4 buf_load16(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
MSAM_DEVINL void buf_store16(const uint4& v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, voff, soff, 0);
}
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
MSAM_DEVINL void buf_store8(const uint2& v, rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), r, voff, soff, 0);
}

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// fp16 operands (11-bit significand) for products whose inputs are not bf16 data: v_mfma_f32_16x16x32_f16
MSAM_DEVINL f32x4_t mfma16h(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

MSAM_DEVINL f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) {
    bf16x8_t av = __builtin_bit_cast(bf16x8_t, a);
    bf16x8_t bv = __builtin_bit_cast(bf16x8_t, b);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
}

// erf-GELU (torch.nn.functional.gelu default) as  gelu(x) = max(x, 0) - |x| * Psi(|x|),  Psi(t) = erfc(t / sqrt 2) / 2.
// log2 Psi is smooth and nearly quadratic: a minimax cubic (weighted by t * Psi, the resulting gelu error) gives
// |gelu error| <= 5.5e-5 absolute for every x (all coefficients negative: the approximation decays monotonically, no
// clamp needed), i.e. ~70x below the bf16 rounding applied to the activations afterwards.  3 FMA + v_exp_f32 + max + FMA:
// the decoder's up-scaling stages evaluate 3.2e9 GELUs per 1024-prompt batch and were VALU-bound on the erf form.
MSAM_DEVINL float gelu_erf(float x) {
    const float t = fabsf(x);
    float q = fmaf(-0.0248758f, t, -0.49884797f);
    q = fmaf(q, t, -1.12922424f);
    q = fmaf(q, t, -1.00353579f);
    return fmaf(-t, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));
}

// two values at once: the cubic and the final FMA as packed fp32 (v_pk_fma_f32)
// max(x, 0) as exactly one instruction: fmaxf() on a value that comes straight out of an MFMA makes the compiler insert a
// canonicalising v_max_f32 x, x, x first (IEEE maxnum semantics), which doubled the max count of the up-scaling kernel.
// v_med3_f32(x, 0, 3e38) needs no canonical input (activations never reach 3e38).  (NOT inline asm: the hazard recognizer does not see inside asm
// statements, so an asm v_max reading a fresh MFMA result ran without the required wait states - wrong results.)
MSAM_DEVINL float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 3.0e38f); }   // (+inf would be folded back to maxnum)
MSAM_DEVINL f32x2_t gelu_erf2(f32x2_t x) {
    const f32x2_t r = {relu1(x.x), relu1(x.y)};
    const f32x2_t t = r * 2.0f - x;                  // |x| = 2 max(x, 0) - x (exact): packed ops have no abs modifier
    f32x2_t q = t * -0.0248758f + -0.49884797f;
    q = q * t + -1.12922424f;
    q = q * t + -1.00353579f;
    const f32x2_t e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    return r - t * e;
}
// round-to-nearest-even fp32 -> packed fp16 (v_cvt_pk_f16_f32)
MSAM_DEVINL uint32_t pack2h(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2_t));
}
MSAM_DEVINL float h2f(uint32_t bits16) { return (float)__builtin_bit_cast(_Float16, (u16)bits16); }

MSAM_DEVINL u16 f2h(float f) { return __builtin_bit_cast(u16, (_Float16)f); }

// ---- the mask decoder's 16-bit type (decoder.hip, decfold.hip, declayer.hip, wsgemm.hip, upfused.hip and the decoder's
// launches of the GEMM kernels).  MSAM_DEC_F16 = 1 (default): IEEE fp16 - 11 significand bits instead of bf16's 8 for the
// per-prompt image-token stream, the folded vectors, the attention probabilities, the token-side activations and every
// decoder weight, same MFMA rate and bytes (v_mfma_f32_16x16x32_f16).  Every decoder activation is a LayerNorm output, a
// softmax probability, a GELU / ReLU of O(1..100) values or a projection of those: far inside fp16's range (65504), which is
// why fp16 inference of SAM is common practice; the ENCODER stays bf16 ("vit_b bf16": un-normalised residual-stream
// products).  Measured effect on the per-instance mask IoU vs the fp32 reference: DESIGN.md section 4.
// MSAM_DEC_F16 = 0 rebuilds the all-bf16 decoder of round 1 (ablation: python -m micro_sam_amd.build --dec-bf16).
#ifndef MSAM_DEC_F16
#define MSAM_DEC_F16 1
#endif
#if MSAM_DEC_F16
MSAM_DEVINL f32x4_t mfma16d(const uint4& a, const uint4& b, f32x4_t c) { return mfma16h(a, b, c); }
MSAM_DEVINL uint32_t pack2d(float lo, float hi) { return pack2h(lo, hi); }
MSAM_DEVINL u16 f2d(float f) { return f2h(f); }
MSAM_DEVINL float d2f(u16 h) { return h2f(h); }
#define MSAM_D16_ONE 0x3C00u
#define MSAM_D16 4                        /* == MSAM_F16 (include/msam_hip.h) */
#else
MSAM_DEVINL f32x4_t mfma16d(const uint4& a, const uint4& b, f32x4_t c) { return mfma16(a, b, c); }
MSAM_DEVINL uint32_t pack2d(float lo, float hi) { return pack2bf(lo, hi); }
MSAM_DEVINL u16 f2d(float f) { return f2bf(f); }
MSAM_DEVINL float d2f(u16 h) { return bf2f(h); }
#define MSAM_D16_ONE 0x3F80u
#define MSAM_D16 2                        /* == MSAM_BF16 */
#endif

// LDS swizzle for a [rows][64] bf16 tile (128-B rows, 8 chunks of 16 B): chunk' = chunk ^ swz(row).
// Chosen so that the 16-lane service groups of ds_read_b128 (MI355X_MICROARCH, LDS table) hit 16 distinct
// 16-B slots when lanes read rows (l & 15) at chunk c0 + (l >> 4).
MSAM_DEVINL int swz(int row) { return ((row >> 1) & 7) ^ (((row + 4) >> 3) & 1); }

MSAM_DEVINL float wave_sum_xor16(float v) {   // sum over the 16 lanes sharing (l >> 4)
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}
MSAM_DEVINL float wave_sum64(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}
MSAM_DEVINL float wave_max64(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8)); v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// ---- Component numbering of the reference (micro_sam/util.py:1834-1838: elf.parallel.label with block_shape (512, 512)): every
// 512 x 512 block is labelled on its own (scikit-image: components in raster order of their first pixel) with a running offset in
// block raster order, components are then united across block faces and the united labelling is made consecutive in order of first
// occurrence over the ascending provisional ids - so a component's rank is the rank of its smallest provisional id, i.e. of its first
// pixel in BLOCK-MAJOR order: key(y, x) = (pixels of all earlier blocks) + (y % 512) * block_width + (x % 512).  The connected-
// component kernels keep that key as the union-find label (the smaller key is the root) and number the roots by ascending key.
// Images of up to 512 x 512 are one block: key = raster index.
MSAM_DEVINL int bm_key(int p, int H, int W) {                 // pixel index -> block-major key
    const int y = p / W, x = p - y * W;
    const int by = y >> 9, bx = x >> 9;
    const int bh = min(512, H - (by << 9)), bw = min(512, W - (bx << 9));
    return (by << 9) * W + (bx << 9) * bh + (y & 511) * bw + (x & 511);
}
MSAM_DEVINL int bm_pix(int q, int H, int W) {                 // block-major key -> pixel index
    const int nby = (H + 511) >> 9, nbx = (W + 511) >> 9;
    const int by = min(q / (W << 9), nby - 1);
    const int rem = q - (by << 9) * W;
    const int bh = min(512, H - (by << 9));
    const int bx = min(rem / (bh << 9), nbx - 1);
    const int rem2 = rem - (bx << 9) * bh;
    const int bw = min(512, W - (bx << 9));
    const int ly = rem2 / bw, lx = rem2 - ly * bw;
    return ((by << 9) + ly) * W + (bx << 9) + lx;
}
