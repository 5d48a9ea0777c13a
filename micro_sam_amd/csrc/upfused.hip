// Fused output up-scaling of the mask decoder (reference: segment_anything MaskDecoder.predict_masks,
// output_upscaling = ConvT(256->64, 2x2) . LayerNorm2d . GELU . ConvT(64->32, 2x2) . GELU, followed by the
// hyper-network product  masks = hyper_in @ upscaled;  SURVEY.md A.4 step (6)) in ONE pass over the per-prompt
// image-token stream: 2 MiB read and 3 x 256 KiB written per prompt, no intermediate stream.
//
// Everything is computed in TRANSPOSED form so that each stage's accumulator registers ARE the next stage's MFMA B
// operand (k-slot map (lane group g, i < 4) <-> row 4g + i of the first 16-row tile, i >= 4 <-> the second tile):
//   stage 1   U^T [(sub, c1)][token] = W1 . keys^T     A = W1 rows (stationary, 128 VGPRs), B = keys tile (LDS, b128)
//             + bias, LayerNorm2d over the 64 c1 of a (token, sub) = registers + 2 cross-lane steps, GELU, pack to bf16
//   stage 2   Y^T [(sub2, c2)][pixel] = W2 . G1        A = W2 (LDS, k-permuted image), B = G1 (registers)
//             + bias, GELU, pack to fp16
//   stage 3   out^T [mask][pixel] = H . G2             A = hyper weights of the prompt (fp16 hi + lo), B = G2 as fp16 (registers)
// 4-wave workgroups on 16-token tiles, wave = sub-pixel of stage 1, two workgroups per CU (one 8-wave workgroup ran all waves in lock
// step: 2.05 ms vs this form; one workgroup per CU: +17 %).  One barrier per tile (tile staging double buffer + output patch double buffer).
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);
void msam_profile_mark2(void* stream, int begin, double flops, double bytes, int family);

namespace {

constexpr int T = 4096, C = 256, TK = 16, NTHR = 256;
int g_uf_prio = 1;                    // tuning hook msam_upscale_set_prio
}
int g_tune_up_gelu16 = 1;             // msam_tune_set "up_gelu16": GELUs in packed fp16 arithmetic (fp16 decoder build; 0 = packed fp32)
int g_tune_up_centred = 1;            // msam_tune_set "up_centred": 0 = centred weights run through the general kernel (which computes their mean of ~0): the A/B of CEN
int g_tune_up_ln_two_pass = 0;        // msam_tune_set "up_ln_two_pass": 1 = centred two-pass LayerNorm2d variance (default: one pass, E[u^2] - mean^2)
namespace {
constexpr int SUB_BYTES = TK * 64 + 64, XT_BYTES = 8 * SUB_BYTES;     // k-step sub-tiles [32 tokens][64 B] (+ pad), see decfold.hip
constexpr int W2_BYTES = 128 * 128;
constexpr int PATCH = 3 * 4 * 64;                                     // [mask][4 rows][64 pixels] fp32

struct UpArgs {
    const u16* keys;                     // bf16 [P, 4096, 256]
    const u16* w1; const float* b1;      // bf16 [256 = sub*64 + c1][256], fp32 [256]
    const float* lnw; const float* lnb; float eps;     // LayerNorm2d over the 64 channels
    const u16* w2; const float* b2;      // bf16 [128 = sub2*32 + c2][64], fp32 [32]
    const float* hyper; int hyper_ld, mask0, nmask;    // fp32 [P, 4, hyper_ld]
    int nitems, KS;
    void* out;                           // fp32 or (out16) fp16 [P, nmask, 256, 256]
    int out16;
    int blocked;                         // keys in the blocked layout of decfold_tok.hip ([16-token tile][k-step][lane][8]) instead of row-major
    int centred;                         // w1 / b1 are centred over the 64 channels of every sub-pixel (see CEN)
};

// sum over the wave's four 16-lane rows, in every lane, without the LDS crossbar: v_permlane16_swap exchanges the odd rows of its first
// operand with the even rows of its second (rows r0 r1 r2 r3 -> (r0 r0 r2 r2), (r1 r1 r3 r3)), v_permlane32_swap the same for 32-lane
// halves; the additions are the ones the xor-16 / xor-32 exchange performs ((r0 + r1) + (r2 + r3)), so the result is the same bits.
MSAM_DEVINL float wave_rows_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float t = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// erf-GELU of two values in PACKED fp16 arithmetic (G16 instantiation, fp16 decoder build only).  The result of either GELU
// is rounded to fp16 anyway - it is the B operand of the next MFMA - so the polynomial can run on v_pk_fma_f16 (two values per
// single-pass instruction; v_pk_fma_f32 takes two passes) and end as the packed operand word: convert, max, fma, 3 fma,
// 2 v_exp_f16, fma = 9 single-pass instructions per pair instead of 2 v_med3 + 5 double-pass packed fp32 + 2 v_exp_f32 + 1 convert.
// Same formula as gelu_erf (common.h): max(x, 0) - |x| 2^cubic(|x|); the exponent's fp16 rounding (|q| <= 8 where the term
// matters) adds <= 1.2e-4 absolute to the 5.5e-5 of the cubic, against an output rounding of 2^-11 relative.
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
// EXPM: 1 = 2^q through v_exp_f16 (quarter-rate transcendental, one per value); 3 = 2^q in packed full-rate arithmetic: q is clamped
// to >= -13, split into r = round(q) (the fp16 magic-number add: q + 1536 has ulp 1) and f = q - r in [-1/2, 1/2], 2^f is a cubic
// (3e-5 relative), and 2^r is added into the exponent field with packed 16-bit integer ops ((bits(q + 1536) << 10) mod 2^16 =
// r << 10); the clamp leaves |x| 2^-13 <= 1e-3 instead of ~0 beyond |x| = 3.45, below half an fp16 ulp of the positive outputs there;
// 2 = timing experiment only (no exponential at all: WRONG results, msam_tune_set("up_gelu16", 2)).
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
template <int EXPM>
MSAM_DEVINL uint32_t gelu_pk_h(float x0, float x1) {
    const f32x2_t xf = {x0, x1};
    const h16x2_t x = __builtin_convertvector(xf, h16x2_t);
    const h16x2_t zero = {(_Float16)0.f, (_Float16)0.f};
    const h16x2_t r = __builtin_elementwise_max(x, zero);
    const h16x2_t t = r * (_Float16)2.f - x;
    h16x2_t q = t * (_Float16)-0.0248758f + (_Float16)-0.49884797f;
    q = q * t + (_Float16)-1.12922424f;
    q = q * t + (_Float16)-1.00353579f;
    h16x2_t e;
    if constexpr (EXPM == 2) {
        e = q;
    } else if constexpr (EXPM == 3) {
        const h16x2_t lim = {(_Float16)-13.f, (_Float16)-13.f};
        const h16x2_t qc = __builtin_elementwise_max(q, lim);
        const h16x2_t sm = qc + (_Float16)1536.f;
        const h16x2_t rr = sm - (_Float16)1536.f;
        const h16x2_t f = qc - rr;
        h16x2_t pz = f * (_Float16)0.0555041f + (_Float16)0.2402265f;
        pz = pz * f + (_Float16)0.6931472f;
        pz = pz * f + (_Float16)1.0f;
        const u16x2_t sh = __builtin_bit_cast(u16x2_t, sm) << (unsigned short)10;
        e = __builtin_bit_cast(h16x2_t, (u16x2_t)(__builtin_bit_cast(u16x2_t, pz) + sh));
    } else {
        // v_exp_f16 has no packed form.  Both exponentials write their half of ONE register (SDWA destination select, the other half
        // preserved): no v_pack_b32_f16 behind them, one vector instruction less per pair (24 per tile).  gfx940-family hazard: a write
        // with a destination select needs one wait state before the register is read again (the second instruction reads it).
        uint32_t ew_;
        asm("v_exp_f16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\ts_nop 0\n\t"
            "v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 0"
            : "=&v"(ew_) : "v"(__builtin_bit_cast(uint32_t, q)));
        e = __builtin_bit_cast(h16x2_t, ew_);
    }
    const h16x2_t g = r - t * e;
    return __builtin_bit_cast(uint32_t, g);
}

// Two GELU pairs per call (round 5; tools/uf_lab.py "R_exp_quad", measured -2.4 % on the launch and bit-identical on the device): the
// four destination-select exponentials are interleaved (A.lo, B.lo, A.hi, B.hi), so the wait state a destination-select write needs before
// its register is read again is filled by the other pair's instruction - one s_nop per two pairs instead of four (59 instead of 169 s_nop
// per tile and wave, profiles/r04_experiments.md section 9).
template <int EXPM>
MSAM_DEVINL void gelu_pk_h2(float x0, float x1, float x2, float x3, uint32_t& g01, uint32_t& g23) {
    if constexpr (EXPM != 1) {                       // the other instantiations (timing / packed-exponential forms): pair by pair
        g01 = gelu_pk_h<EXPM == 0 ? 1 : EXPM>(x0, x1); g23 = gelu_pk_h<EXPM == 0 ? 1 : EXPM>(x2, x3);
        return;
    }
    const h16x2_t zero = {(_Float16)0.f, (_Float16)0.f};
    const h16x2_t xa = __builtin_convertvector(f32x2_t{x0, x1}, h16x2_t), xb = __builtin_convertvector(f32x2_t{x2, x3}, h16x2_t);
    const h16x2_t ra = __builtin_elementwise_max(xa, zero), rb = __builtin_elementwise_max(xb, zero);
    const h16x2_t ta = ra * (_Float16)2.f - xa, tb = rb * (_Float16)2.f - xb;
    h16x2_t qa = ta * (_Float16)-0.0248758f + (_Float16)-0.49884797f, qb = tb * (_Float16)-0.0248758f + (_Float16)-0.49884797f;
    qa = qa * ta + (_Float16)-1.12922424f; qb = qb * tb + (_Float16)-1.12922424f;
    // (round 6, measured and not shipped - profiles/r06_experiments.md: the cubic's last step, the exponentials and the final multiply-adds as ONE inline-assembly
    //  block are 10 instructions less per tile and 1 - 2 % SLOWER: nothing can be scheduled into the block)
    qa = qa * ta + (_Float16)-1.00353579f; qb = qb * tb + (_Float16)-1.00353579f;
    uint32_t ea_, eb_;
    asm("v_exp_f16_sdwa %0, %2 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t"
        "v_exp_f16_sdwa %1, %3 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n\t"
        "v_exp_f16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\t"
        "v_exp_f16_sdwa %1, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 0"
        : "=&v"(ea_), "=&v"(eb_) : "v"(__builtin_bit_cast(uint32_t, qa)), "v"(__builtin_bit_cast(uint32_t, qb)));
    g01 = __builtin_bit_cast(uint32_t, (h16x2_t)(ra - ta * __builtin_bit_cast(h16x2_t, ea_)));
    g23 = __builtin_bit_cast(uint32_t, (h16x2_t)(rb - tb * __builtin_bit_cast(h16x2_t, eb_)));
}

// The tile loop is software-pipelined INSIDE every wave (round 4; measured as tools/uf_lab.py "R_pipe_sdwa_dephase": the pipelining and the
// destination-select exponentials are bit-identical to the stage-after-stage form of rounds 1 - 3 and 6 % shorter, profiles/r04_experiments.md
// section 1; the one-pass LayerNorm statistics shipped with them are NOT bit-identical to rounds 1 - 3 - see msam_tune_set "up_ln_two_pass"):
//     phase A   8 x { B fragment of tile q + 1, 4 MFMAs into `un` }  interleaved with  { sums, centring, rstd, affine + GELU of tile q (`uc`) }
//     phase B   stages 2 and 3 of tile q (a software pipeline over the four second-level sub-pixels)
// so that every MFMA of phase A has independent VALU work of the same wave behind it.  The LayerNorm sums go through v_permlane16/32_swap
// (the same additions as an xor-16 / xor-32 exchange, no LDS crossbar round trip).  Registers: the stage-1 bias is read from the LDS
// parameter block instead of being held (16 VGPRs, 4 more ds_read_b128 per tile), a second accumulator set is added (16).
// LDS: tile q + 1 is read from one staging buffer while tile q + 2 is written into the other (free since the previous barrier).
// The second half of the grid starts ~700 cycles late: co-resident workgroups (i, i + grid / 2) then run a quarter tile period out of
// phase instead of competing for the same pipe in the same stage.
// round 6: phase A's LDS operands are read AHEAD of their use - the B fragment of k-step ks + 2 behind the MFMAs of k-step ks (two register sets
// by turns), the LayerNorm affine parameters of row tile rt + 1 in front of the GELUs of row tile rt, the next tile's first two fragments and its
// stage-1 bias right behind the barrier.  The ISA of the round-5 form had `ds_read_b128; s_waitcnt lgkmcnt(0)` in front of every MFMA group and
// every affine step: 16 exposed LDS round trips per tile and wave (SQ_WAIT_INST_ANY 30 % of the wave cycles).  Plain loads do not stay ahead - the
// optimizer sinks them back to their first use, and the compiler's lgkmcnt counts cannot see an inline-assembly read next to them - so EVERY LDS read
// of phase A is inline assembly with its own `s_waitcnt lgkmcnt(n)`: LDS operations return in order, n = the number of younger reads that may still
// be in flight (derivation next to each wait; a compiler-issued LDS operation in between only makes a wait stricter; there is no scalar memory load
// in the loop - those return out of order and would void the counts: tests/test_upfused_isa.py checks the ISA for it).
// Same arithmetic in the same order: bit-identical to round 5 (tools/uf_lab.py "R_lds_behind" is the A/B).
// 16 VGPRs for it: the stage-2 bias and the prompt's hyper weights (hi, lo) live in LDS between the tiles' phases B (4 more reads per tile).
// CEN (round 6): the caller hands over CENTRED first-layer weights and bias (every sub-pixel's 64 rows of w1 minus their mean row, b1 minus its mean:
// msam_upscale_fused_out, `keys_blocked` bit 1) - the 64 channels of a (token, sub-pixel) then have mean zero by construction, LayerNorm2d's mean, its
// exchange and the centring (16 + 6 + 8 vector instructions per tile and wave) are not computed and the variance is the plain mean of squares
// (the two-pass value).  The other instantiations accept such weights too (they compute a mean of ~0).
template <int UF_PRIO, int G16, int LN2P = 0, int CEN = 0>
__global__ __launch_bounds__(NTHR, 2) void up_fused_kernel(UpArgs a) {
#define UF_LDS_AHEAD 1
#if defined(__HIP_DEVICE_COMPILE__) && UF_LDS_AHEAD
#define UF_R16(dst_, base_, imm_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(base_), "n"(imm_))      // base_: LDS byte address (one VGPR), imm_: constant < 65536
#define UF_LDS_ADDR(ptr_) ((uint32_t)(size_t)(ptr_))
#define UF_ARRIVED2(n_, x0_, tok_) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(x0_), "+v"(tok_))
#define UF_ARRIVED4(n_, x0_, x1_, x2_, tok_) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(x0_), "+v"(x1_), "+v"(x2_), "+v"(tok_))
#define UF_ARRIVED6(n_, x0_, x1_, x2_, x3_, x4_, tok_) asm volatile("s_waitcnt lgkmcnt(" #n_ ")" : "+v"(x0_), "+v"(x1_), "+v"(x2_), "+v"(x3_), "+v"(x4_), "+v"(tok_))
#else                               // (host build of the kernel source: the reads complete at once)
#define UF_R16(dst_, base_, imm_) dst_ = *(const f32x4_t*)((base_) + (imm_))
#define UF_LDS_ADDR(ptr_) ((const unsigned char*)(ptr_))
#define UF_ARRIVED2(n_, x0_, tok_) (void)0
#define UF_ARRIVED4(n_, x0_, x1_, x2_, tok_) (void)0
#define UF_ARRIVED6(n_, x0_, x1_, x2_, x3_, x4_, tok_) (void)0
#endif
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XT_BYTES + W2_BYTES];
    __shared__ __attribute__((aligned(16))) float patch[2][PATCH + 4 * 64];      // (+ one mask's worth: the output stage's idle lanes read a fourth mask)
    __shared__ __attribute__((aligned(16))) float prm[256 + 64 + 64 + 32];       // b1, ln_w, ln_b, b2
#if UF_LDS_AHEAD
    __shared__ __attribute__((aligned(16))) uint4 hyp[4][2][64];                 // per wave: hyper weights of the current prompt as MFMA A fragments (hi, lo)
#endif
    unsigned char* const W2L = lds + 2 * XT_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, fg = lane >> 4;
    const int sub = w;
    const int ks_sh = __builtin_ctz(a.KS), tpi_sh = 8 - ks_sh, TPI = 1 << tpi_sh;        // 256 tiles per prompt
    const int my_items = (a.nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nq = my_items * TPI;
    if (nq <= 0) return;

    for (int i = tid; i < 416; i += NTHR) prm[i] = i < 256 ? a.b1[i] : i < 320 ? a.lnw[i - 256] : i < 384 ? a.lnb[i - 320] : a.b2[i - 384];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                            // W2 image: see the shipped kernel
        const int id = j * NTHR + tid, row = id >> 3, ch = id & 7, kk = ch >> 2, g = ch & 3;
        const uint2 lo = *(const uint2*)(a.w2 + row * 64 + kk * 32 + g * 4);
        const uint2 hi = *(const uint2*)(a.w2 + row * 64 + kk * 32 + 16 + g * 4);
        *(uint4*)(W2L + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    uint4 w1f[4][8];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            w1f[rt][ks] = *(const uint4*)(a.w1 + (long)(sub * 64 + rt * 16 + fr) * C + ks * 32 + fg * 8);
    wait_vmem_all();

    uint4 ra0, ra1, rb0, rb1;
    int kdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = i * NTHR + tid;
        const int row = a.blocked ? id & 15 : id >> 5, c = a.blocked ? (id >> 6) * 4 + ((id >> 4) & 3) : id & 31;
        kdst[i] = (c >> 2) * SUB_BYTES + row * 64 + (((c & 3) ^ ((row >> 2) & 3)) << 4);
    }
    auto tile_pos = [&](int q, int& p, int& key0) {
        const int item = (int)blockIdx.x + (q >> tpi_sh) * (int)gridDim.x;
        p = __builtin_amdgcn_readfirstlane(item >> ks_sh);
        key0 = __builtin_amdgcn_readfirstlane((item & (a.KS - 1)) * (T >> ks_sh) + (q & (TPI - 1)) * TK);
    };
    const int voff = tid * 16;
    // round 6: the load cursor (the tile being fetched, three ahead of the tile being computed, clamped to the last one) and the compute cursor advance
    // INCREMENTALLY - a tile is 8 KiB further in the prompt's stream, 16 tokens further in the output - and are recomputed from the tile number only
    // at a work item's first tile: ~20 scalar instructions less per tile (every instruction of a wave costs it an issue slot, scalar ones included)
    int lq = 0, lsoff;                                   // load cursor: tile number, byte offset of the tile in its prompt's stream
    const u16* lbase;                                    //              the prompt's stream
    {
        int p_, k0_;
        tile_pos(0, p_, k0_);
        lbase = a.keys + (long)p_ * T * C; lsoff = k0_ * C * 2;
    }
    auto load_advance = [&]() {
        if (lq < nq - 1) {
            ++lq;
            if ((lq & (TPI - 1)) == 0) {
                int p_, k0_;
                tile_pos(lq, p_, k0_);
                lbase = a.keys + (long)p_ * T * C; lsoff = k0_ * C * 2;
            } else {
                lsoff += TK * C * 2;
            }
        }
    };
#define UF_LOAD(r0_, r1_)                                                                          \
    do {                                                                                           \
        const rsrc_t rx_ = make_rsrc(lbase, T * C * 2);                                            \
        r0_ = buf_load16(rx_, voff, lsoff); r1_ = buf_load16(rx_, voff, lsoff + NTHR * 16);        \
        load_advance();                                                                            \
    } while (0)
#define UF_STORE(r0_, r1_, buf_)                                                                   \
    do {                                                                                           \
        unsigned char* b_ = lds + (buf_) * XT_BYTES;                                               \
        *(uint4*)(b_ + kdst[0]) = r0_; *(uint4*)(b_ + kdst[1]) = r1_;                              \
    } while (0)

    const int boff = fr * 64 + ((fg ^ ((fr >> 2) & 3)) << 4);                     // stage-1 B fragment (token fr, k-step slot fg)
    const int w2off = fr * 128;
    const int w2sw = (fr >> 1) & 7;
    const float* const b1p = prm + sub * 64 + fg * 4;                             // + 16 rt: the stage-1 bias of rows 16 rt + 4 fg ..
#if !UF_LDS_AHEAD
    uint4 hh = make_uint4(0, 0, 0, 0), hl = hh;
    f32x4_t b2a, b2b;                                                             // stage-2 bias: the initial accumulator of every chain
    {
        const float4 ba = *(const float4*)(a.b2 + fg * 4), bb = *(const float4*)(a.b2 + 16 + fg * 4);
        b2a = f32x4_t{ba.x, ba.y, ba.z, ba.w}; b2b = f32x4_t{bb.x, bb.y, bb.z, bb.w};
    }
#endif
    wait_vmem_all();

    // stage 1 of one tile, alone (prologue only)
    auto stage1 = [&](const unsigned char* B, f32x4_t (&u)[4]) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) u[rt] = *(const f32x4_t*)(b1p + rt * 16);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const uint4 kf = *(const uint4*)(B + boff + ks * SUB_BYTES);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) u[rt] = mfma16d(w1f[rt][ks], kf, u[rt]);
        }
    };

    // tiles 0 and 1 staged, tile 2 in flight, stage 1 of tile 0 done
    f32x4_t ua[4], ub[4];
    UF_LOAD(ra0, ra1);                                      // tile 0
    UF_LOAD(rb0, rb1);                                      // tile min(1, nq - 1)
    __syncthreads();                                         // prm / W2 image written
    UF_STORE(ra0, ra1, 0);
    UF_STORE(rb0, rb1, 1);
    UF_LOAD(ra0, ra1);                                      // tile min(2, nq - 1)
    __syncthreads();
    stage1(lds, ua);
    __syncthreads();                                         // every wave is done with buffer 0 before tile 2 goes there
#if UF_LDS_AHEAD
    f32x4_t kfa, kfb;                                        // B fragments of the even / odd k-steps (bit patterns of 8 fp16)
#define UF_NEXT_TILE(B_, u_)                                                                       \
    do {                                                                                           \
        UF_R16(u_[0], b1a, 0); UF_R16(u_[1], b1a, 64); UF_R16(u_[2], b1a, 128); UF_R16(u_[3], b1a, 192); \
        const auto bn_ = UF_LDS_ADDR((B_) + boff);                                                 \
        UF_R16(kfa, bn_, 0); UF_R16(kfb, bn_, SUB_BYTES);                                          \
    } while (0)
    const auto b1a = UF_LDS_ADDR(b1p), prma = UF_LDS_ADDR(&prm[256 + fg * 4]);
    UF_NEXT_TILE(lds + XT_BYTES, ub);
#endif

    if ((int)blockIdx.x >= ((int)gridDim.x >> 1)) __builtin_amdgcn_s_sleep(11);        // ~700 cycles: see the note above the kernel
    const int oesz = a.out16 ? 1 : 2;                                                  // log2 of the output element size
    const int ptoff = ((lane >> 4) * 4 + w) * 64 + (lane & 15) * 4;                     // output stage: mask lane >> 4, patch row w, pixels 4 (lane & 15) ..
    const int ovoff = ((((lane >> 4) * 256 + w) * 256) + (lane & 15) * 4) << oesz;
    // output stage, one tile behind: every wave stores one patch row of all masks (lane >> 4 = mask, 4 pixels per lane: the same work in all four waves)
    // through buffer addressing - the per-thread part of the address is loop-invariant, and lanes of a mask >= nmask fall outside the descriptor's
    // range: the hardware drops their store (as it drops every lane's before the first tile: range 0)
    float4 o4c = make_float4(0.f, 0.f, 0.f, 0.f);
    const char* obase_c = (const char*)a.out;
    uint32_t orange_c = 0;
    int osoff_c = 0;
    auto store_previous = [&]() {
        // (loop-carried scalars: the compiler keeps some of them in vector registers and would wrap the store in a readfirstlane loop)
        const int oso = __builtin_amdgcn_readfirstlane(osoff_c);
        const unsigned long long ob_ = (unsigned long long)obase_c;
        const char* obu = (const char*)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ob_ >> 32)) << 32) |
                                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ob_));
        const rsrc_t ro = make_rsrc(obu, (uint32_t)__builtin_amdgcn_readfirstlane((int)orange_c));
        // fp16 low-res logits (the AMG path: msam_postprocess_masks16 reads them back - half the round trip of 0.8 GB per tile;
        // 2^-11 relative, i.e. <= 5e-4 where the thresholds 0, +-1 are decided, against a logit error of ~3e-2)
        if (__builtin_amdgcn_readfirstlane(oesz) == 1) buf_store8(make_uint2(pack2h(o4c.x, o4c.y), pack2h(o4c.z, o4c.w)), ro, ovoff, oso);
        else buf_store16(make_uint4(__float_as_uint(o4c.x), __float_as_uint(o4c.y), __float_as_uint(o4c.z), __float_as_uint(o4c.w)), ro, ovoff, oso);
    };
    int q = 0, p = 0, key0 = 0;                                                        // compute cursor: tile number, prompt, first token of the tile
    const char* obase = (const char*)a.out;
    auto iteration = [&](uint4& p0, uint4& p1, uint4& f0, uint4& f1, f32x4_t (&uc)[4], f32x4_t (&un)[4]) {
        UF_LOAD(f0, f1);                                 // tile min(q + 3, nq - 1)
        if ((q & (TPI - 1)) != 0) {
            key0 += TK;
        } else {                                         // new work item: position, output descriptor base, hyper weights (rows = masks, k-slots = c2)
            tile_pos(q, p, key0);
            obase = (const char*)a.out + (((long)p * a.nmask) << (16 + oesz));
            float h8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (fr < a.nmask) {
                const float* hp = a.hyper + ((long)p * 4 + a.mask0 + fr) * a.hyper_ld;
                const float4 x0 = *(const float4*)(hp + fg * 4), x1 = *(const float4*)(hp + 16 + fg * 4);
                h8[0] = x0.x; h8[1] = x0.y; h8[2] = x0.z; h8[3] = x0.w; h8[4] = x1.x; h8[5] = x1.y; h8[6] = x1.z; h8[7] = x1.w;
            }
            wait_vmem_all();
#if UF_LDS_AHEAD
            uint4 hh, hl;
#endif
            hh = make_uint4(pack2h(h8[0], h8[1]), pack2h(h8[2], h8[3]), pack2h(h8[4], h8[5]), pack2h(h8[6], h8[7]));
            const uint32_t hw_[4] = {hh.x, hh.y, hh.z, hh.w};
            float l8[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { l8[2 * i] = h8[2 * i] - h2f(hw_[i] & 0xffff); l8[2 * i + 1] = h8[2 * i + 1] - h2f(hw_[i] >> 16); }
            hl = make_uint4(pack2h(l8[0], l8[1]), pack2h(l8[2], l8[3]), pack2h(l8[4], l8[5]), pack2h(l8[6], l8[7]));
#if UF_LDS_AHEAD
            hyp[w][0][lane] = hh; hyp[w][1][lane] = hl;
#endif
        }
        // ---- phase A: stage 1 of tile q + 1 (staging buffer (q + 1) & 1) under LayerNorm2d + GELU of tile q
        const unsigned char* Bn = lds + ((q + 1) & 1) * XT_BYTES;
#if UF_LDS_AHEAD
        const auto bna = UF_LDS_ADDR(Bn + boff);
#endif
#if !UF_LDS_AHEAD
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) un[rt] = *(const f32x4_t*)(b1p + rt * 16);
#endif
        // (the empty asm makes the k-step's operand fragment depend on `tok_`, the last value of the VALU chunk in front of it: instruction
        //  selection cannot move the MFMA group above that chunk; the sched_barrier keeps the machine scheduler from undoing the placement)
#if UF_LDS_AHEAD
        // k-step ks_: its fragment has arrived once at most n_ younger reads are in flight; the MFMA group; the fragment of k-step ks_ + 2 into the same registers
#define UF_MF_(ks_, between_)                                                                      \
        do {                                                                                       \
            f32x4_t& kf_ = ((ks_) & 1) ? kfb : kfa;                                                \
            const uint4 kw_ = make_uint4(__float_as_uint(kf_[0]), __float_as_uint(kf_[1]), __float_as_uint(kf_[2]), __float_as_uint(kf_[3])); \
            _Pragma("unroll") for (int rt_ = 0; rt_ < 4; ++rt_) un[rt_] = mfma16d(w1f[rt_][ks_], kw_, un[rt_]); \
            between_;                                                                              \
            if ((ks_) + 2 < 8) UF_R16(kf_, bna, (((ks_) + 2) & 7) * SUB_BYTES);                    \
        } while (0)
#define UF_MF(ks_) UF_MF_(ks_, (void)0)
#define UF_S1(ks_, tok_, n_)                                                                       \
        do {                                                                                       \
            if ((ks_) & 1) UF_ARRIVED2(n_, kfb, tok_); else UF_ARRIVED2(n_, kfa, tok_);            \
            UF_MF(ks_);                                                                            \
        } while (0)
#else
#define UF_MF(ks_) (void)0
#define UF_S1(ks_, tok_, n_) UF_S1_(ks_, tok_)
#define UF_S1_(ks_, tok_)                                                                          \
        do {                                                                                       \
            uint4 kf_ = *(const uint4*)(Bn + boff + (ks_) * SUB_BYTES);                            \
            asm volatile("" : "+v"(kf_.x), "+v"(tok_));                                            \
            _Pragma("unroll") for (int rt_ = 0; rt_ < 4; ++rt_) un[rt_] = mfma16d(w1f[rt_][ks_], kf_, un[rt_]); \
        } while (0)
#endif
        // (sched_barrier: the MFMA group of a k-step and the VALU chunk next to it stay together - the scheduler otherwise moves the
        //  GELUs behind all 32 MFMAs, which is the shipped order again)
#if UF_LDS_AHEAD
        // in flight here, oldest first: bias x 4, kfa (k-step 0), kfb (1) [UF_NEXT_TILE behind the last barrier]: the bias and kfa are there with <= 1 left
        { int first_ = 0; UF_ARRIVED6(1, un[0], un[1], un[2], un[3], kfa, first_); }
        // (the previous tile's output store sits between this k-step's MFMAs and the next reads: its patch read - the one compiler-issued LDS read of the
        //  phase, whose wait the compiler can only write as lgkmcnt(0) - is then the only LDS operation in flight, and a whole phase B + barrier old)
        UF_MF_(0, store_previous());                      // + kfa (2)
        f32x4_t gq[2], bq[2];                             // affine parameters, two register sets by turns
        UF_R16(gq[0], prma, 0); UF_R16(bq[0], prma, 256);
#else
        { int first_ = 0; UF_S1(0, first_, 0); }
        store_previous();
#endif
        // LayerNorm2d statistics in one pass: both sums before either exchange (the two exchanges are independent: one latency
        // instead of two in a row), variance = E[u^2] - mean^2 in fp32 (measured -6 % on the launch, profiles/r04_experiments.md)
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if constexpr (!CEN) s += (uc[rt][0] + uc[rt][1]) + (uc[rt][2] + uc[rt][3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) ss += uc[rt][r] * uc[rt][r];
        }
        if constexpr (!CEN) s = wave_rows_sum(s);
        ss = wave_rows_sum(ss);
        float mean = s * (1.f / 64.f);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CEN) UF_S1(1, ss, 3); else UF_S1(1, mean, 3);      // in flight: kfb (1), kfa (2), g (0), b (0)  -> + kfb (3)
        if constexpr (!CEN) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) uc[rt][r] -= mean;
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LN2P) {
            // msam_tune_set("up_ln_two_pass", 1): the centred two-pass variance of rounds 1 - 3 (and of the reference's LayerNorm2d) - a second
            // exchange after the centring.  The one-pass form loses ~2^-23 E[u^2] / var of relative accuracy: negligible while |mean| is of
            // the order of the standard deviation (SAM's up-scaling activations), not when |mean| >> std
            float s2 = 0.f;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s2 += uc[rt][r] * uc[rt][r];
            ss = wave_rows_sum(s2);
        }
        UF_S1(2, ss, 3);                                 // in flight: kfa (2), g (0), b (0), kfb (3)  -> + kfa (4)
        float rstd = (LN2P || CEN) ? rsqrtf(ss * (1.f / 64.f) + a.eps) : rsqrtf(fmaxf(ss * (1.f / 64.f) - mean * mean, 0.f) + a.eps);
        __builtin_amdgcn_sched_barrier(0);
        uint32_t g1w[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
#if UF_LDS_AHEAD
            // in flight, oldest first: [rt = 0] g (0), b (0), kfb (3), kfa (4); [rt > 0] kf (3 + rt), g (rt), b (rt), kf (4 + rt): this step's three with <= 1 left
            // (tied to the previous chunk's last value, or the wait drifts above that chunk's arithmetic)
            f32x4_t& kfr = (rt & 1) ? kfa : kfb;          // k-step 3 + rt
            if (rt == 0) UF_ARRIVED4(1, gq[0], bq[0], kfr, rstd); else UF_ARRIVED4(1, gq[rt & 1], bq[rt & 1], kfr, g1w[rt - 1][1]);
            const float4 g4 = make_float4(gq[rt & 1][0], gq[rt & 1][1], gq[rt & 1][2], gq[rt & 1][3]);
            const float4 b4 = make_float4(bq[rt & 1][0], bq[rt & 1][1], bq[rt & 1][2], bq[rt & 1][3]);
            if (rt < 3) { UF_R16(gq[(rt + 1) & 1], prma, ((rt + 1) & 3) * 64); UF_R16(bq[(rt + 1) & 1], prma, 256 + ((rt + 1) & 3) * 64); }
            UF_MF(3 + rt);                                // + kf (5 + rt) while there is one
#else
            if (rt == 0) UF_S1(3, rstd, 0); else UF_S1(3 + rt, g1w[rt - 1][1], 0);
            const float4 g4 = *(const float4*)&prm[256 + rt * 16 + fg * 4], b4 = *(const float4*)&prm[320 + rt * 16 + fg * 4];
#endif
            if (G16) {
                gelu_pk_h2<G16>(uc[rt][0] * rstd * g4.x + b4.x, uc[rt][1] * rstd * g4.y + b4.y, uc[rt][2] * rstd * g4.z + b4.z,
                                uc[rt][3] * rstd * g4.w + b4.w, g1w[rt][0], g1w[rt][1]);
            } else {
                const f32x2_t g01 = gelu_erf2(f32x2_t{uc[rt][0] * rstd * g4.x + b4.x, uc[rt][1] * rstd * g4.y + b4.y});
                const f32x2_t g23 = gelu_erf2(f32x2_t{uc[rt][2] * rstd * g4.z + b4.z, uc[rt][3] * rstd * g4.w + b4.w});
                g1w[rt][0] = pack2d(g01.x, g01.y); g1w[rt][1] = pack2d(g23.x, g23.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        UF_S1(7, g1w[3][1], 0);                          // in flight: kfb (7)
#undef UF_S1
#undef UF_MF
        uint4 g1[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) g1[kk] = make_uint4(g1w[2 * kk][0], g1w[2 * kk][1], g1w[2 * kk + 1][0], g1w[2 * kk + 1][1]);
        // ---- phase B: stages 2 and 3 of tile q, one second-level sub-pixel at a time (the shipped pipeline)
        float* pt = patch[q & 1];
        f32x4_t ya[4], yb[4], ov[4];
        uint4 gh[4];
#if UF_LDS_AHEAD
        const f32x4_t b2a = *(const f32x4_t*)&prm[384 + fg * 4], b2b = *(const f32x4_t*)&prm[400 + fg * 4];
        const uint4 hh = hyp[w][0][lane], hl = hyp[w][1][lane];
#endif
        auto stage2 = [&](int s2) {                                  // c2 = 4 fg + r (ya) and 16 + 4 fg + r (yb)
            f32x4_t xa = b2a, xb = b2b;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint4 wa = *(const uint4*)(W2L + (s2 * 2) * 2048 + w2off + (((kk * 4 + fg) ^ w2sw) << 4));
                const uint4 wb = *(const uint4*)(W2L + (s2 * 2 + 1) * 2048 + w2off + (((kk * 4 + fg) ^ w2sw) << 4));
                xa = mfma16d(wa, g1[kk], xa);
                xb = mfma16d(wb, g1[kk], xb);
            }
            ya[s2] = xa; yb[s2] = xb;
        };
        auto act2 = [&](int s2) {
            if (G16) {
                gelu_pk_h2<G16>(ya[s2][0], ya[s2][1], ya[s2][2], ya[s2][3], gh[s2].x, gh[s2].y);
                gelu_pk_h2<G16>(yb[s2][0], yb[s2][1], yb[s2][2], yb[s2][3], gh[s2].z, gh[s2].w);
                return;
            }
            const f32x2_t a01 = gelu_erf2(f32x2_t{ya[s2][0], ya[s2][1]}), a23 = gelu_erf2(f32x2_t{ya[s2][2], ya[s2][3]});
            const f32x2_t b01 = gelu_erf2(f32x2_t{yb[s2][0], yb[s2][1]}), b23 = gelu_erf2(f32x2_t{yb[s2][2], yb[s2][3]});
            gh[s2] = make_uint4(pack2h(a01.x, a01.y), pack2h(a23.x, a23.y), pack2h(b01.x, b01.y), pack2h(b23.x, b23.y));
        };
        auto stage3 = [&](int s2) {
            f32x4_t o = {0.f, 0.f, 0.f, 0.f};
            o = mfma16h(hh, gh[s2], o);
            o = mfma16h(hl, gh[s2], o);
            ov[s2] = o;
        };
        stage2(0); stage2(1);
        act2(0);
        stage2(2);
        act2(1);
        stage3(0); stage2(3);
        act2(2);
        stage3(1);
        act2(3);
        stage3(2); stage3(3);
        // (round 6, measured: forcing 24 x (1 MFMA, 5 .. 8 vector instructions) on this phase through sched_group_barrier: +1.5 % .. -0.5 %: the compiler's order stays)
        // (round 6, measured - profiles/r06_uf_lab_call8.txt: a further diet of 21 instructions per tile - sum of squares in packed fp32, a mask-innermost patch written
        //  without register moves, the stage-2 bias held in registers, a loop without mid-exits - is 1.4 % SLOWER than this form: instruction count is not the whole story)
        // (round 6, measured: the outputs straight from these accumulators - six 4 / 8-byte stores of pixel pairs per wave, no patch - is 3 % SLOWER than the patch,
        //  profiles/r06_uf_lab_call6.txt; without ANY output store the launch is 9 % shorter: the stores themselves, not the path to them)
        if (fg == 0) {                                   // rows = masks r, column = token fr
#pragma unroll
            for (int sub2 = 0; sub2 < 4; ++sub2) {
                const int yl = (sub >> 1) * 2 + (sub2 >> 1), xl = fr * 4 + (sub & 1) * 2 + (sub2 & 1);
                pt[(0 * 4 + yl) * 64 + xl] = ov[sub2][0];
                pt[(1 * 4 + yl) * 64 + xl] = ov[sub2][1];
                pt[(2 * 4 + yl) * 64 + xl] = ov[sub2][2];
            }
        }
        UF_STORE(p0, p1, q & 1);                         // tile q + 2 into the buffer tile q was read from (free since the last barrier)
        __syncthreads();                                 // tile q + 2 staged; output patch of this tile complete
#if UF_LDS_AHEAD
        UF_NEXT_TILE(lds + (q & 1) * XT_BYTES, uc);      // the next iteration's stage-1 bias (its accumulators' initial value) and first two B fragments (tile q + 2, staged above)
#endif
        // round 6: the tile's output leaves ONE ITERATION LATER (store_previous, in the next iteration's first k-step): here only its patch read is issued
        // - stored right away, the read's round trip (behind the six reads above) stood between the barrier and everything else of the next iteration
        o4c = *(const float4*)&pt[ptoff];
        obase_c = obase; orange_c = (uint32_t)a.nmask << (16 + oesz);
        osoff_c = (((key0 >> 6) << 10) + ((key0 & 63) << 2)) << oesz;                    // rows 4 ty .., pixels 4 tx0 ..
    };
    while (true) {
        iteration(ra0, ra1, rb0, rb1, ua, ub);
        if (++q >= nq) break;
        iteration(rb0, rb1, ra0, ra1, ub, ua);
        if (++q >= nq) break;
    }
    store_previous();                                        // the last tile's
#undef UF_LOAD
#undef UF_STORE
}

}  // namespace

extern "C" int msam_upscale_fused_out(const void* keys, int32_t keys_blocked, int32_t P, const void* w1, const float* b1,
                                      const float* ln_w, const float* ln_b, float ln_eps, const void* w2, const float* b2,
                                      const float* hyper, int32_t hyper_ld, int32_t mask0, int32_t nmask, void* low_res,
                                      int32_t low_res_dtype, void* stream);
extern "C" int msam_upscale_fused_layout(const void* keys, int32_t keys_blocked, int32_t P, const void* w1, const float* b1,
                                         const float* ln_w, const float* ln_b, float ln_eps, const void* w2, const float* b2,
                                         const float* hyper, int32_t hyper_ld, int32_t mask0, int32_t nmask, float* low_res,
                                         void* stream) {
    return msam_upscale_fused_out(keys, keys_blocked, P, w1, b1, ln_w, ln_b, ln_eps, w2, b2, hyper, hyper_ld, mask0, nmask, low_res,
                                  MSAM_F32, stream);
}
extern "C" int msam_upscale_fused(const void* keys, int32_t P, const void* w1, const float* b1, const float* ln_w,
                                  const float* ln_b, float ln_eps, const void* w2, const float* b2, const float* hyper,
                                  int32_t hyper_ld, int32_t mask0, int32_t nmask, float* low_res, void* stream) {
    return msam_upscale_fused_layout(keys, 0, P, w1, b1, ln_w, ln_b, ln_eps, w2, b2, hyper, hyper_ld, mask0, nmask, low_res, stream);
}

extern "C" int msam_upscale_fused_out(const void* keys, int32_t keys_blocked, int32_t P, const void* w1, const float* b1,
                                      const float* ln_w, const float* ln_b, float ln_eps, const void* w2, const float* b2,
                                      const float* hyper, int32_t hyper_ld, int32_t mask0, int32_t nmask, void* low_res,
                                      int32_t low_res_dtype, void* stream) {
    if (low_res_dtype != MSAM_F32 && low_res_dtype != MSAM_F16) {
        msam_set_error("msam_upscale_fused_out: low_res_dtype is MSAM_F32 or MSAM_F16");
        return 1;
    }
    if (!keys || !w1 || !b1 || !ln_w || !ln_b || !w2 || !b2 || !hyper || !low_res || P <= 0) {
        msam_set_error("msam_upscale_fused: null argument");
        return 1;
    }
    if (nmask < 1 || nmask > 3 || mask0 < 0 || mask0 + nmask > 4 || hyper_ld < 32 || hyper_ld % 4) {
        msam_set_error("msam_upscale_fused: 1 <= nmask <= 3 masks out of 4, hyper_ld >= 32 and a multiple of 4");
        return 1;
    }
    UpArgs a{};
    a.keys = (const u16*)keys; a.w1 = (const u16*)w1; a.b1 = b1; a.lnw = ln_w; a.lnb = ln_b; a.eps = ln_eps;
    a.w2 = (const u16*)w2; a.b2 = b2; a.hyper = hyper; a.hyper_ld = hyper_ld; a.mask0 = mask0; a.nmask = nmask;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int ks = 1;
    while (P * ks < 2 * cus && ks < 16) ks *= 2;
    a.KS = ks; a.nitems = P * ks; a.out = low_res; a.out16 = low_res_dtype == MSAM_F16; a.blocked = keys_blocked & 1; a.centred = (keys_blocked >> 1) & 1;
    const int grid = a.nitems < 2 * cus ? a.nitems : 2 * cus;
    const double rows = (double)P * T;
    const double flops = rows * (2.0 * 256 * 256 + 4 * 2.0 * 128 * 64 + 16 * 3 * 2.0 * 16 * 32);
    const double bytes = rows * C * 2 + (double)P * nmask * 256 * 256 * (low_res_dtype == MSAM_F16 ? 2 : 4);
    msam_profile_mark2(stream, 1, flops, bytes, 4);
#if MSAM_DEC_F16
    if (g_tune_up_gelu16 == 2) hipLaunchKernelGGL((up_fused_kernel<1, 2>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
    else if (g_tune_up_gelu16 == 3) hipLaunchKernelGGL((up_fused_kernel<1, 3>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
    else if (g_tune_up_gelu16) {
        if (g_tune_up_ln_two_pass) hipLaunchKernelGGL((up_fused_kernel<1, 1, 1>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
        else if (g_uf_prio && a.centred && g_tune_up_centred) hipLaunchKernelGGL((up_fused_kernel<1, 1, 0, 1>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
        else if (g_uf_prio) hipLaunchKernelGGL((up_fused_kernel<1, 1>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((up_fused_kernel<0, 1>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
    } else
#endif
    if (g_tune_up_ln_two_pass) hipLaunchKernelGGL((up_fused_kernel<1, 0, 1>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
    else if (g_uf_prio && a.centred && g_tune_up_centred) hipLaunchKernelGGL((up_fused_kernel<1, 0, 0, 1>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
    else if (g_uf_prio) hipLaunchKernelGGL((up_fused_kernel<1, 0>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((up_fused_kernel<0, 0>), dim3(grid), dim3(NTHR), 0, (hipStream_t)stream, a);
    msam_profile_mark2(stream, 0, flops, bytes, 4);
    return msam_check_launch("up_fused");
}

extern "C" int msam_upscale_set_prio(int32_t prio) { g_uf_prio = prio; return 0; }
