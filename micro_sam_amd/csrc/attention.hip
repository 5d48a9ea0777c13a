// ViT attention with decomposed relative-position bias for the SAM image encoder.
// Kernels are templated on the STORED head dimension HD in {64, 96}: vit_b / vit_l have head_dim 64; vit_h (head_dim 80)
// runs with its heads zero-padded to 96 channels (= 3 MFMA k-steps of 32) by the host side - padded q / k channels
// contribute 0 to every score, padded v channels produce 0 outputs that meet zero columns of the padded proj weight -
// and the softmax scale of the TRUE head_dim is passed at run time.
//
// Both kernels compute the TRANSPOSED score tile S^T = K * Q^T with v_mfma_f32_16x16x32_bf16, so that in the
// C layout (row = key = (l>>4)*4 + r, col = query = l & 15) every softmax statistic is a per-lane-column
// quantity (reduce over registers + 2 cross-lane steps), and the un-normalised probabilities P^T can be fed
// straight back as the B operand of O^T = V^T * P^T without touching LDS: for the k-slot (g = l>>4, i) of a
// K=32 MFMA built from two 16-key tiles t0,t1 the key is t*16 + g*4 + (i & 3); the V^T fragment is read
// with the same key map (two 8-byte LDS reads).
//
//   window kernel : one workgroup per (image, 14x14 window, head); all 196 keys (+ bias-only padding tokens
//                   of the 70x70 padded grid) resident in LDS; rel-pos terms T[q][j] = q . R[j] by MFMA.
//   global kernel : one workgroup per (image, head, image row of 64 queries); flash-style online softmax over
//                   128 tiles of 32 keys (half an image row: kh is constant per tile), K / V^T tiles double
//                   buffered in LDS with register prefetch; rel_h[q][kh] and rel_w[q][kw] precomputed by MFMA
//                   into LDS in the prologue.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);

namespace {

constexpr int TOK = 4096;
constexpr float NEG_BIG = -1.0e30f;

// F16 (both kernels): q / k / v / rel-pos tables and the output are IEEE fp16 instead of bf16 (the encoder's fp16 mode,
// msam_encoder_t.dtype16): the fp16 MFMA of the same shape and rate, probabilities packed to fp16.
template <bool F16> MSAM_DEVINL f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
    if constexpr (F16) return mfma16h(a, b, c); else return mfma16(a, b, c);
}
template <bool F16> MSAM_DEVINL uint32_t pk2(float lo, float hi) {
    if constexpr (F16) return pack2h(lo, hi); else return pack2bf(lo, hi);
}
template <bool F16> MSAM_DEVINL uint4 bias_chunk(const float* b) {   // 8 fp32 -> 8 x 16 bit
    uint4 r;
    r.x = pk2<F16>(b[0], b[1]); r.y = pk2<F16>(b[2], b[3]); r.z = pk2<F16>(b[4], b[5]); r.w = pk2<F16>(b[6], b[7]);
    return r;
}

// --------------------------------------------------------------------------------------------- window
constexpr int WS = 14, WN = 196, WKT = 13 /* key tiles */, WKP = 224 /* padded keys for PV pairs */;
constexpr int VT_RS = 232;    // V^T row stride in bf16 (464 B = 116 dwords = 4 * odd -> conflict-free b64 reads)
constexpr int T_RS = 65;      // rel-pos table row stride (floats)

template <int HD, bool F16 = false>
__global__ __launch_bounds__(256, 2) void window_attention_kernel(
    const u16* __restrict__ Q, const u16* __restrict__ K, const u16* __restrict__ V, const u16* __restrict__ relh,
    const u16* __restrict__ relw, const float* __restrict__ qkv_bias, int heads, float scale, u16* __restrict__ out) {
    constexpr int KS = HD / 32;                 // MFMA k-steps of the q.k contraction
    constexpr int DT = HD / 16;                 // 16-channel output tiles
    constexpr int CH = HD / 8;                  // 16-byte chunks per K / V row
    constexpr int CHP = HD == 64 ? 8 : 16;      // chunk pitch of a k_lds row (the XOR swizzle needs a power of two)
    __shared__ __attribute__((aligned(16))) uint4 k_lds[WKT * 16 * CHP];          // 26 KB (HD 64) / 52 KB (HD 96)
    __shared__ __attribute__((aligned(16))) u16 vt_lds[HD * VT_RS];               // 29 KB / 43.5 KB
    __shared__ __attribute__((aligned(16))) float t_lds[4][16 * T_RS];            // 16.3 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    int bid = blockIdx.x;
    const int head = bid % heads; bid /= heads;
    const int win = bid % 25, b = bid / 25;
    const int wy = win / 5, wx = win % 5;
    const int D = heads * HD;
    const long bh = ((long)b * heads + head) * TOK * HD;
    const u16* Qb = Q + bh; const u16* Kb = K + bh; const u16* Vb = V + bh;
    const float* bq = qkv_bias + head * HD;
    const float* bk = qkv_bias + D + head * HD;
    const float* bv = qkv_bias + 2 * D + head * HD;

    // ---- stage K (row-major, swizzled chunks) and V^T.  All global loads of the workgroup are issued before the first LDS write
    // (round 3: a load -> wait -> write loop made 14 dependent memory latencies per workgroup; with 2 workgroups per CU the kernel
    // spent most of its 52 us per workgroup waiting for them).
    constexpr int KIT = (WKT * 16 * CH + 255) / 256, VIT = (WKP * CH + 255) / 256;
    uint4 kreg[KIT], vreg[VIT];
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
        const int c = tid + 256 * i;
        const int row = c / CH, ch = c - row * CH;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (c < WKT * 16 * CH && row < WN) {
            const int y = wy * WS + row / WS, x = wx * WS + row % WS;
            if (y < 64 && x < 64) val = *(const uint4*)(Kb + (long)(y * 64 + x) * HD + ch * 8);
            else val = bias_chunk<F16>(bk + ch * 8);
        }
        kreg[i] = val;
    }
#pragma unroll
    for (int i = 0; i < VIT; ++i) {
        const int c = tid + 256 * i;
        const int key = c % WKP, ch = c / WKP;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (c < WKP * CH && key < WN) {
            const int y = wy * WS + key / WS, x = wx * WS + key % WS;
            if (y < 64 && x < 64) val = *(const uint4*)(Vb + (long)(y * 64 + x) * HD + ch * 8);
            else val = bias_chunk<F16>(bv + ch * 8);
        }
        vreg[i] = val;
    }
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
        const int c = tid + 256 * i;
        const int row = c / CH, ch = c - row * CH;
        if (c < WKT * 16 * CH) k_lds[row * CHP + (ch ^ swz(row))] = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VIT; ++i) {
        const int c = tid + 256 * i;
        const int key = c % WKP, ch = c / WKP;
        if (c < WKP * CH) {
            const uint32_t wv[4] = {vreg[i].x, vreg[i].y, vreg[i].z, vreg[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vt_lds[(ch * 8 + 2 * j) * VT_RS + key] = (u16)(wv[j] & 0xffff);
                vt_lds[(ch * 8 + 2 * j + 1) * VT_RS + key] = (u16)(wv[j] >> 16);
            }
        }
    }

    // ---- rel-pos rows as MFMA A operand: tile jt rows j = jt*16 + fr; j < 27 -> rel_h[j], 32 <= j < 59 -> rel_w[j-32]
    uint4 ra[4][KS];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int j = jt * 16 + fr;
        const u16* src = nullptr;
        if (j < 27) src = relh + j * HD;
        else if (j >= 32 && j < 59) src = relw + (j - 32) * HD;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            ra[jt][ks] = src ? *(const uint4*)(src + ks * 32 + fg * 8) : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    float* tl = t_lds[wave];
    // Q fragments (B operand) of query tile qt: query qi = qt*16 + fr; the NEXT tile's are requested while this one is worked on
    auto load_q = [&](int qt, uint4 (&dstq)[KS]) {
        const int qi = qt * 16 + fr;
        const int qh = (qi < WN ? qi : WN - 1) / WS, qw = (qi < WN ? qi : WN - 1) % WS;
        const int qy = wy * WS + qh, qx = wx * WS + qw;
        const bool q_real = qi < WN && qy < 64 && qx < 64;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (q_real) dstq[ks] = *(const uint4*)(Qb + (long)(qy * 64 + qx) * HD + ks * 32 + fg * 8);
            else if (qi < WN) dstq[ks] = bias_chunk<F16>(bq + ks * 32 + fg * 8);
            else dstq[ks] = make_uint4(0, 0, 0, 0);
        }
    };
    uint4 qnext[KS];
    load_q(wave, qnext);
    for (int qt = wave; qt < WKT; qt += 4) {
        const int qi = qt * 16 + fr;
        const int qh = (qi < WN ? qi : WN - 1) / WS, qw = (qi < WN ? qi : WN - 1) % WS;
        const int qy = wy * WS + qh, qx = wx * WS + qw;
        const bool q_real = qi < WN && qy < 64 && qx < 64;
        uint4 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = qnext[ks];
        if (qt + 4 < WKT) load_q(qt + 4, qnext);
        // T^T[j][q] = R[j] . q  -> t_lds[q][j]
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            f32x4_t t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) t = mma<F16>(ra[jt][ks], qf[ks], t);
#pragma unroll
            for (int r = 0; r < 4; ++r) tl[fr * T_RS + jt * 16 + fg * 4 + r] = t[r];
        }
        // S^T = K Q^T
        f32x4_t s[WKT];
#pragma unroll
        for (int kt = 0; kt < WKT; ++kt) {
            const int row = kt * 16 + fr;
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a = mma<F16>(k_lds[row * CHP + ((ks * 4 + fg) ^ swz(row))], qf[ks], a);
            s[kt] = a;
        }
        __builtin_amdgcn_wave_barrier();
        // scale, rel-pos bias, mask padding keys; column-wise (per query) softmax
        // (the key -> (kh, kw) arithmetic is kept INSIDE the query-tile loop: hoisted, its 104 per-lane values pushed the kernel to 353
        // registers = one wave per SIMD, one workgroup per CU)
        int fgv = fg;
        asm volatile("" : "+v"(fgv));
        float m = NEG_BIG;
#pragma unroll
        for (int kt = 0; kt < WKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + fgv * 4 + r;
                float v = NEG_BIG;
                if (key < WN) {
                    const int kh = key / WS, kw = key - kh * WS;
                    v = s[kt][r] * scale + tl[fr * T_RS + (qh - kh + 13)] + tl[fr * T_RS + 32 + (qw - kw + 13)];
                }
                s[kt][r] = v;
                m = fmaxf(m, v);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < WKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { float p = __expf(s[kt][r] - m); s[kt][r] = p; l += p; }
        l += __shfl_xor(l, 16); l += __shfl_xor(l, 32);
        // O^T = V^T P^T
        f32x4_t o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int t0 = 2 * u, t1 = 2 * u + 1;
            uint4 pb;
            pb.x = pk2<F16>(s[t0][0], s[t0][1]); pb.y = pk2<F16>(s[t0][2], s[t0][3]);
            if (t1 < WKT) { pb.z = pk2<F16>(s[t1][0], s[t1][1]); pb.w = pk2<F16>(s[t1][2], s[t1][3]); }
            else { pb.z = 0; pb.w = 0; }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const u16* vr = vt_lds + (dt * 16 + fr) * VT_RS + fg * 4;
                uint2 lo = *(const uint2*)(vr + t0 * 16), hi = *(const uint2*)(vr + t1 * 16);
                o[dt] = mma<F16>(make_uint4(lo.x, lo.y, hi.x, hi.y), pb, o[dt]);
            }
        }
        if (q_real) {
            const float inv = 1.0f / l;
            u16* dst = out + ((long)b * TOK + qy * 64 + qx) * D + head * HD + fg * 4;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                uint2 pk; pk.x = pk2<F16>(o[dt][0] * inv, o[dt][1] * inv); pk.y = pk2<F16>(o[dt][2] * inv, o[dt][3] * inv);
                *(uint2*)(dst + dt * 16) = pk;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// --------------------------------------------------------------------------------------------- global
// Global attention of the ViT blocks 2, 5, 8, 11 (4096 keys, head_dim 64, decomposed relative position bias).
// One workgroup = 128 queries (two image rows) of one (tile, head); wave = 32 queries = two 16-query score tiles that
// share every K / V fragment (the kernel is bound by the L2 bandwidth of re-reading K / V once per workgroup: 128
// queries per workgroup halve that traffic against one image row).  Transposed-score form: S^T = K . Q^T, online
// softmax per lane column, O^T += V^T . P^T with V^T read from the row-major V tile by ds_read_b64_tr_b16.
// Bias: rel_h[q][kh] (one scalar per query and key ROW) from an LDS table; rel_w[q][kw] only depends on the key column,
// i.e. on the position inside a 32-key tile (2 phases) - the 16 values a lane ever needs live in registers.
constexpr int GQ = 128;           // queries per workgroup
constexpr int GKT = 32;           // keys per tile
constexpr int GB_RS = 68;         // bias table row stride in floats (272 B, multiple of 16 B)

typedef short gs16x4_t __attribute__((ext_vector_type(4)));
MSAM_DEVINL uint2 g_tr16(const unsigned char* p) {
    gs16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gs16x4_t __attribute__((address_space(3)))*)p);
    return __builtin_bit_cast(uint2, v);
}

template <int HD, bool F16 = false>
__global__ __launch_bounds__(256, HD == 64 ? 3 : 2) void global_attention_kernel(
    const u16* __restrict__ Q, const u16* __restrict__ K, const u16* __restrict__ V, const u16* __restrict__ relh,
    const u16* __restrict__ relw, int heads, float scale, u16* __restrict__ out) {
    constexpr int KS = HD / 32, DT = HD / 16;
    constexpr int CHP = HD == 64 ? 8 : 16;      // chunk pitch of a k_lds row
    constexpr int VROW = HD == 64 ? 128 : 256;  // bytes per key of the V tile (32-byte d-tile slots, XOR-swizzled)
    __shared__ __attribute__((aligned(16))) uint4 k_lds[2][GKT * CHP];                 // 8 KB (HD 64) / 16 KB
    __shared__ __attribute__((aligned(16))) unsigned char v_lds[2][GKT * VROW];        // 8 KB / 16 KB, [key][HD d] bf16
    __shared__ __attribute__((aligned(16))) float rh_lds[GQ * GB_RS];                  // 34 KB  rel_h[q][kh] (rel_w scratch first)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    int bid = blockIdx.x;
    const int qpair = bid & 31; bid >>= 5;
    const int head = bid % heads, b = bid / heads;
    const int D = heads * HD;
    const long bh = ((long)b * heads + head) * TOK * HD;
    const u16* Qb = Q + bh; const u16* Kb = K + bh; const u16* Vb = V + bh;

    const int qh = qpair * 2 + (wave >> 1);              // image row of this wave's 32 queries
    const int qw0 = (wave & 1) * 32;                     // first column
    uint4 qf[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[j][ks] = *(const uint4*)(Qb + (long)(qh * 64 + qw0 + j * 16 + fr) * HD + ks * 32 + fg * 8);

    // ---- rel_w[q][kw] = q . rel_pos_w[qw - kw + 63]: T[j'][q] for all 127 rows j', scattered to kw = qw - j' + 63 of this
    // wave's private scratch rows, then the 16 values per score tile this lane needs are pulled into registers
    float* scr = rh_lds + (wave * 32) * GB_RS;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int qw = qw0 + j * 16 + fr;
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) {
            const int jr = jt * 16 + fr;
            f32x4_t c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 a = jr < 127 ? *(const uint4*)(relw + jr * HD + ks * 32 + fg * 8) : make_uint4(0, 0, 0, 0);
                c = mma<F16>(a, qf[j][ks], c);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kw = qw - (jt * 16 + fg * 4 + r) + 63;
                if (kw >= 0 && kw < 64) scr[(j * 16 + fr) * GB_RS + kw] = c[r];
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // Softmax in the base-2 domain (round 3): scores, both bias terms and the running maximum carry the factor log2(e), the
    // exponential is the bare v_exp_f32; the per-element work is written on float pairs (v_pk_fma_f32 / v_pk_add_f32: two values
    // per issue slot) - the loop was bound by its VALU count (171 VALU against 16 MFMA per 32-key tile, 25 % MFMA busy).
    constexpr float LOG2E = 1.4426950408889634f;
    f32x2_t bw[2][2][2][2];                              // [score tile j][kw phase][key block t][pair], times log2(e)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 v4 = *(const float4*)(scr + (j * 16 + fr) * GB_RS + ph * 32 + t * 16 + fg * 4);
                bw[j][ph][t][0] = f32x2_t{v4.x * LOG2E, v4.y * LOG2E}; bw[j][ph][t][1] = f32x2_t{v4.z * LOG2E, v4.w * LOG2E};
            }
    __builtin_amdgcn_wave_barrier();
    // ---- rel_h[q][kh] = q . rel_pos_h[qh - kh + 63]  (A rows indexed by kh), into the same rows
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int kh = t * 16 + fr;
            const u16* src = relh + (qh - kh + 63) * HD;
            f32x4_t c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) c = mma<F16>(*(const uint4*)(src + ks * 32 + fg * 8), qf[j][ks], c);
            *(float4*)(scr + (j * 16 + fr) * GB_RS + t * 16 + fg * 4) = make_float4(c[0] * LOG2E, c[1] * LOG2E, c[2] * LOG2E, c[3] * LOG2E);
        }

    // ---- K / V tile staging (register prefetch, double-buffered LDS); both tiles row-major [32 keys][HD]
    // chunks 0..7 of a row: one per thread; HD 96 has chunks 8..11 as well (second load, threads pair up on rows 0..31)
    const int s_row = tid >> 3, s_ch = tid & 7;
    const int s_row2 = (tid >> 2) & 31, s_ch2 = 8 + (tid & 3);
    uint4 rk, rv, rk2, rv2;
    rk = *(const uint4*)(Kb + (long)s_row * HD + s_ch * 8);
    rv = *(const uint4*)(Vb + (long)s_row * HD + s_ch * 8);
    if constexpr (HD > 64) {
        rk2 = *(const uint4*)(Kb + (long)s_row2 * HD + s_ch2 * 8);
        rv2 = *(const uint4*)(Vb + (long)s_row2 * HD + s_ch2 * 8);
    }
    // V image: 32-byte d-tile slot' = slot ^ ((key >> 1) & 3) so that the 4 keys of a transposing read hit distinct banks
    const int v_dst = s_row * VROW + ((((s_ch >> 1) ^ ((s_row >> 1) & 3)) << 5) | ((s_ch & 1) << 4));
    const int v_dst2 = s_row2 * VROW + ((((s_ch2 >> 1) ^ ((s_row2 >> 1) & 3)) << 5) | ((s_ch2 & 1) << 4));
#define G_COMMIT(buf_)                                                                   \
    do {                                                                                 \
        k_lds[buf_][s_row * CHP + (s_ch ^ swz(s_row))] = rk;                             \
        *(uint4*)(v_lds[buf_] + v_dst) = rv;                                             \
        if constexpr (HD > 64) {                                                         \
            k_lds[buf_][s_row2 * CHP + (s_ch2 ^ swz(s_row2))] = rk2;                     \
            *(uint4*)(v_lds[buf_] + v_dst2) = rv2;                                       \
        }                                                                                \
    } while (0)
    G_COMMIT(0);
    __syncthreads();

    // transposing read of V: lane supplies the address of 4 d of key (fg*4 + (fr >> 2)) [+16 for the second block]
    int troff[2];
#pragma unroll
    for (int bk = 0; bk < 2; ++bk) {
        const int key = bk * 16 + fg * 4 + (fr >> 2);
        troff[bk] = key * VROW + (fr & 3) * 8;           // + ((dt ^ ((key >> 1) & 3)) << 5) per d-tile
    }
    const int vsw = (fg * 2 + (fr >> 3)) & 3;            // (key >> 1) & 3 for both blocks (16 >> 1 = 8 = 0 mod 4)

    f32x4_t o[2][DT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[j][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m[2] = {NEG_BIG, NEG_BIG}, l[2] = {0.f, 0.f};
    const int NT = TOK / GKT;   // 128
    const float c2s = scale * LOG2E;
    const f32x2_t c2 = {c2s, c2s};
    // two 32-key tiles per trip = one key row kh of the 64 x 64 grid: the kw phase and the LDS buffer are compile-time
    for (int kh = 0; kh < NT / 2; ++kh) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
        const int kt = kh * 2 + ph;
        const int buf = ph;
        {
            const int nx = min(kt + 1, NT - 1);
            rk = *(const uint4*)(Kb + (long)(nx * GKT + s_row) * HD + s_ch * 8);
            rv = *(const uint4*)(Vb + (long)(nx * GKT + s_row) * HD + s_ch * 8);
            if constexpr (HD > 64) {
                rk2 = *(const uint4*)(Kb + (long)(nx * GKT + s_row2) * HD + s_ch2 * 8);
                rv2 = *(const uint4*)(Vb + (long)(nx * GKT + s_row2) * HD + s_ch2 * 8);
            }
        }
        uint4 ka[2][KS];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) ka[t][ks] = k_lds[buf][row * CHP + ((ks * 4 + fg) ^ swz(row))];
        }
        uint4 va[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const uint2 lo = g_tr16(v_lds[buf] + troff[0] + ((dt ^ vsw) << 5));
            const uint2 hi = g_tr16(v_lds[buf] + troff[1] + ((dt ^ vsw) << 5));
            va[dt] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4_t s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a = mma<F16>(ka[t][ks], qf[j][ks], a);
                s[t] = a;
            }
            const float rh = scr[(j * 16 + fr) * GB_RS + kh];           // already times log2(e)
            const f32x2_t rh2 = {rh, rh};
            f32x2_t e[2][2];                                            // base-2 logits, pairs (0, 1) and (2, 3) of key block t
            float mt = NEG_BIG;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                e[t][0] = f32x2_t{s[t][0], s[t][1]} * c2 + (rh2 + bw[j][ph][t][0]);
                e[t][1] = f32x2_t{s[t][2], s[t][3]} * c2 + (rh2 + bw[j][ph][t][1]);
                mt = fmaxf(mt, fmaxf(fmaxf(e[t][0].x, e[t][0].y), fmaxf(e[t][1].x, e[t][1].y)));
            }
            mt = fmaxf(mt, __shfl_xor(mt, 16)); mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mn = fmaxf(m[j], mt);
            // the running maximum moves in a few of the 128 tiles: rescale the accumulators only then (alpha = 2^0 = 1 otherwise, exactly)
            if (__builtin_amdgcn_ballot_w64(mn > m[j]) != 0) {
                const float alpha = __builtin_amdgcn_exp2f(m[j] - mn);
                const f32x2_t al2 = {alpha, alpha};
                l[j] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const f32x2_t o01 = f32x2_t{o[j][dt][0], o[j][dt][1]} * al2, o23 = f32x2_t{o[j][dt][2], o[j][dt][3]} * al2;
                    o[j][dt][0] = o01.x; o[j][dt][1] = o01.y; o[j][dt][2] = o23.x; o[j][dt][3] = o23.y;
                }
                m[j] = mn;
            }
            const f32x2_t nm2 = {-mn, -mn};
            f32x2_t psv = {0.f, 0.f};
            float pv[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x2_t x = e[t][h2] + nm2;
                    const f32x2_t p = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
                    psv += p;
                    pv[t][2 * h2] = p.x; pv[t][2 * h2 + 1] = p.y;
                }
            l[j] += psv.x + psv.y;         // per-lane partial row sum; the 4 lane groups are combined after the loop
            uint4 pb;
            pb.x = pk2<F16>(pv[0][0], pv[0][1]); pb.y = pk2<F16>(pv[0][2], pv[0][3]);
            pb.z = pk2<F16>(pv[1][0], pv[1][1]); pb.w = pk2<F16>(pv[1][2], pv[1][3]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[j][dt] = mma<F16>(va[dt], pb, o[j][dt]);
        }
        if (kt + 1 < NT) G_COMMIT(buf ^ 1);
        __syncthreads();
        }
    }
#undef G_COMMIT
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float lj = l[j];
        lj += __shfl_xor(lj, 16); lj += __shfl_xor(lj, 32);
        const float inv = 1.0f / lj;
        u16* dst = out + ((long)b * TOK + qh * 64 + qw0 + j * 16 + fr) * D + head * HD + fg * 4;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            uint2 pk; pk.x = pk2<F16>(o[j][dt][0] * inv, o[j][dt][1] * inv); pk.y = pk2<F16>(o[j][dt][2] * inv, o[j][dt][3] * inv);
            *(uint2*)(dst + dt * 16) = pk;
        }
    }
}

}  // namespace

extern "C" int msam_window_attention16(const void* q, const void* k, const void* v, const void* rel_h, const void* rel_w,
                                       const float* qkv_bias, int32_t B, int32_t heads, int32_t head_dim, float scale,
                                       int32_t dtype16, void* out, void* stream) {
    if (!q || !k || !v || !rel_h || !rel_w || !qkv_bias || !out || B <= 0 || heads <= 0) {
        msam_set_error("msam_window_attention: bad arguments");
        return 1;
    }
    if (head_dim != 64 && head_dim != 96) {
        msam_set_error("msam_window_attention: stored head_dim must be 64 or 96 (vit_h: 80 zero-padded to 96)");
        return 1;
    }
    if (dtype16 != MSAM_BF16 && dtype16 != MSAM_F16) { msam_set_error("msam_window_attention: dtype16 must be MSAM_BF16 or MSAM_F16"); return 1; }
    const dim3 grid(B * 25 * heads), block(256);
    hipStream_t s = (hipStream_t)stream;
#define WA_GO(HD_, F16_) hipLaunchKernelGGL((window_attention_kernel<HD_, F16_>), grid, block, 0, s, (const u16*)q, (const u16*)k, \
                                            (const u16*)v, (const u16*)rel_h, (const u16*)rel_w, qkv_bias, heads, scale, (u16*)out)
    if (dtype16 == MSAM_F16) { if (head_dim == 64) WA_GO(64, true); else WA_GO(96, true); }
    else { if (head_dim == 64) WA_GO(64, false); else WA_GO(96, false); }
#undef WA_GO
    return msam_check_launch("msam_window_attention");
}

extern "C" int msam_window_attention(const void* q, const void* k, const void* v, const void* rel_h, const void* rel_w,
                                     const float* qkv_bias, int32_t B, int32_t heads, int32_t head_dim, float scale, void* out,
                                     void* stream) {
    return msam_window_attention16(q, k, v, rel_h, rel_w, qkv_bias, B, heads, head_dim, scale, MSAM_BF16, out, stream);
}

extern "C" int msam_global_attention16(const void* q, const void* k, const void* v, const void* rel_h, const void* rel_w,
                                       int32_t B, int32_t heads, int32_t head_dim, float scale, int32_t dtype16, void* out,
                                       void* stream) {
    if (!q || !k || !v || !rel_h || !rel_w || !out || B <= 0 || heads <= 0) {
        msam_set_error("msam_global_attention: bad arguments");
        return 1;
    }
    if (head_dim != 64 && head_dim != 96) {
        msam_set_error("msam_global_attention: stored head_dim must be 64 or 96 (vit_h: 80 zero-padded to 96)");
        return 1;
    }
    if (dtype16 != MSAM_BF16 && dtype16 != MSAM_F16) { msam_set_error("msam_global_attention: dtype16 must be MSAM_BF16 or MSAM_F16"); return 1; }
    const dim3 grid(B * heads * 32), block(256);
    hipStream_t s = (hipStream_t)stream;
#define GA_GO(HD_, F16_) hipLaunchKernelGGL((global_attention_kernel<HD_, F16_>), grid, block, 0, s, (const u16*)q, (const u16*)k, \
                                            (const u16*)v, (const u16*)rel_h, (const u16*)rel_w, heads, scale, (u16*)out)
    if (dtype16 == MSAM_F16) { if (head_dim == 64) GA_GO(64, true); else GA_GO(96, true); }
    else { if (head_dim == 64) GA_GO(64, false); else GA_GO(96, false); }
#undef GA_GO
    return msam_check_launch("msam_global_attention");
}

extern "C" int msam_global_attention(const void* q, const void* k, const void* v, const void* rel_h, const void* rel_w,
                                     int32_t B, int32_t heads, int32_t head_dim, float scale, void* out, void* stream) {
    return msam_global_attention16(q, k, v, rel_h, rel_w, B, heads, head_dim, scale, MSAM_BF16, out, stream);
}
