// Token -> image cross attention of the mask decoder with the K / V projections FOLDED into the token side
// (SURVEY.md A.4 step (2) and the final attention; reference: segment_anything two-way transformer `Attention`):
//
//     S[j][(h,t)] = (keys_j + pe_j) Wk_h^T . q_{h,t} / 4  =  keys_j . Q'_{h,t} / 4 + tabK_{j,h} . q_{h,t} / 4
//         Q'_{h,t} = Wk_h^T q_{h,t}  (256 channels),  tabK = pe Wk^T + bk  (prompt independent table)
//     out_{h,t}   = softmax_j(S) (keys_j Wv_h^T + bv_h)  =  (softmax_j(S) keys) Wv_h^T + bv_h
//
// The 4096 x 256 per-prompt image-token stream is therefore read ONCE per attention (2 MiB / prompt) instead of being
// projected to K and V^T (two more streams written and read back): 4x less HBM traffic on this path, and fewer MFMA flops
// (the 8 x 8 = 64 folded queries per prompt are cheaper than two 256 -> 128 projections of 4096 tokens).
//
// Main kernel: 4 waves, wave w owns the 16 score columns of heads 2w, 2w+1 (col = (h & 1) * 8 + t).  Per 32-key tile
// (staged global -> registers two tiles ahead -> LDS double buffer):
//     S^T = keys . Q'^T           8 MFMA 16x16x32 per 16-key block, A = keys rows (ds_read_b128), B = Q' (32 VGPRs, resident)
//         + tabK . q (block diag) 1 MFMA, K = 32 = the two heads' 16 channels
//     online softmax per lane column (transposed-score form, see attention.hip)
//     O'^T[c][(h,t)] += keys^T P^T   16 MFMA, A = keys^T through ds_read_b64_tr_b16 (hardware 4x4 transpose, LDS image is
//                                    [16 channel tiles][32 keys][16 channels] with the 16-B halves of a row swapped on
//                                    key bit 3 so that the b128 score reads are bank-conflict free as well), B = P^T
//                                    straight from the score registers
// (m, l, O') partials go to a fp32 workspace ([P][KS][64][256], KS key splits when there are few prompts); the finish
// kernel merges the splits, normalises and applies the per-head 256 -> 16 value projection.
#include "common.h"
#include "../../include/msam_hip.h"
#include <string>

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);
void msam_profile_mark2(void* stream, int begin, double flops, double bytes, int family);

namespace {

constexpr int T = 4096, C = 256, CI = 128, TK = 32, NTHR = 256;
// stream tile image of fold_attn_kernel: 16 channel sub-tiles [32 keys][16 channels] of 1024 B, padded to 1056 B so that the
// staging stores of one token row (16 lanes -> 8 sub-tiles) spread over all LDS banks instead of hitting one 32-byte window
constexpr int SUBT = 1024 + 32;
constexpr int KEYS_BYTES = 16 * SUBT, TAB_BYTES = TK * CI * 2, BUF_BYTES = KEYS_BYTES + TAB_BYTES;
constexpr float NEG_BIG = -1.0e30f;
#ifndef FOLD_ATTN_DMA_DEFAULT
#define FOLD_ATTN_DMA_DEFAULT 0
#endif

typedef short s16x4_t __attribute__((ext_vector_type(4)));

MSAM_DEVINL uint2 lds_tr16(const unsigned char* p) {
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)p);
    return __builtin_bit_cast(uint2, v);
}

// Q'[p][h*8 + t][c] = sum_d q[p][t][h*16 + d] Wk[h*16 + d][c]   (zero rows for t >= Nt)
__global__ __launch_bounds__(256) void fold_q_kernel(const u16* __restrict__ qtok, const u16* __restrict__ wk, int Nt,
                                                     u16* __restrict__ qprime) {
    __shared__ float q[8][16];
    const int p = blockIdx.x >> 3, h = blockIdx.x & 7, c = threadIdx.x;
    if (c < 128) {
        const int t = c >> 4, d = c & 15;
        q[t][d] = t < Nt ? d2f(qtok[((long)p * Nt + t) * CI + h * 16 + d]) : 0.f;
    }
    __syncthreads();
    float w[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) w[d] = d2f(wk[(h * 16 + d) * C + c]);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) a = fmaf(q[t][d], w[d], a);
        qprime[(((long)p * 64) + h * 8 + t) * C + c] = f2d(a);
    }
}

struct FoldArgs {
    const u16* keys; int kv_shared;      // bf16 [Pk, 4096, 256]
    const u16* qprime;                   // bf16 [P, 64, 256]
    const u16* qtok; int Nt;             // bf16 [P, Nt, 128]
    const u16* tabk;                     // bf16 [4096, 128]
    int nitems, KS;                      // work items (prompt, key split)
    float* opart;                        // fp32 [P, KS, 64, 256]
    float* stats;                        // fp32 [P, KS, 64, 2]  (m, l)
    const u16* wv; const float* bv; u16* out;   // KS == 1: value projection fused into the item epilogue, bf16 [P, Nt, 128]
    int blocked;                         // keys in the blocked layout of decfold_tok.hip ([16-token tile][k-step][lane][8]) instead of row-major
};

// DMA: the stream and table tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16, the
// tile image's sub-tile order / half swap / chunk swizzle applied on the per-lane SOURCE address) into the other buffer while
// the current tile is consumed: no staging registers (48 VGPRs) and no ds_write pass, which lets THREE workgroups share a CU.
int g_fold_attn_dma = FOLD_ATTN_DMA_DEFAULT;         // tuning hook msam_fold_attn_set_dma

template <bool DMA>
__global__ __launch_bounds__(NTHR, DMA ? 3 : 2) void fold_attn_kernel(FoldArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int ks_sh = __builtin_ctz(a.KS), tpi_sh = 7 - ks_sh;
    const int TPI = 1 << tpi_sh;                                      // tiles per item ((T / TK) / KS)
    const int my_items = (a.nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nq = my_items * TPI;
    if (nq <= 0) return;

    // ---- staging: 4 keys chunks + 2 table chunks (16 B) per thread and tile
    uint4 ra0, ra1, ra2, ra3, ra4, ra5, rb0, rb1, rb2, rb3, rb4, rb5;
    int kdst[4], tdst[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // 16-byte chunk id of the (contiguous) 32-key tile -> key row, channel chunk c: row-major [32][32 chunks], or blocked
        // [2 blocks of 16 keys][k-step 8][lane group 4][key 16]
        const int id = i * NTHR + tid;
        const int row = a.blocked ? (id >> 9) * 16 + (id & 15) : id >> 5;
        const int c = a.blocked ? ((id >> 6) & 7) * 4 + ((id >> 4) & 3) : id & 31;
        kdst[i] = (c >> 1) * SUBT + row * 32 + (((c & 1) ^ ((row >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = i * NTHR + tid, row = id >> 4, c = id & 15;
        tdst[i] = KEYS_BYTES + row * 256 + ((c ^ (row & 15)) << 4);
    }
    // tile q of this workgroup -> (prompt stream base, byte offset of the tile in the stream / table)
    const int voff = tid * 16;
    const rsrc_t rtab = make_rsrc(a.tabk, T * CI * 2);
    auto tile_src = [&](int q, rsrc_t& rk, int& koffs, int& toffs) {   // KS, TPI are powers of two
        const int item = (int)blockIdx.x + (q >> tpi_sh) * (int)gridDim.x;
        const int p = item >> ks_sh, ks = item & (a.KS - 1);
        const int key0 = ks * (T >> ks_sh) + (q & (TPI - 1)) * TK;
        rk = make_rsrc(a.keys + (long)(a.kv_shared ? 0 : p) * T * C, T * C * 2);
        koffs = key0 * C * 2; toffs = key0 * CI * 2;
    };
#define FA_LOAD(r0_, r1_, r2_, r3_, r4_, r5_, q_)                                                  \
    do {                                                                                           \
        rsrc_t rk_; int ko_, to_;                                                                  \
        tile_src(q_, rk_, ko_, to_);                                                               \
        r0_ = buf_load16(rk_, voff, ko_); r1_ = buf_load16(rk_, voff, ko_ + NTHR * 16);            \
        r2_ = buf_load16(rk_, voff, ko_ + 2 * NTHR * 16); r3_ = buf_load16(rk_, voff, ko_ + 3 * NTHR * 16); \
        r4_ = buf_load16(rtab, voff, to_); r5_ = buf_load16(rtab, voff, to_ + NTHR * 16);          \
    } while (0)
#define FA_STORE(r0_, r1_, r2_, r3_, r4_, r5_, buf_)                                               \
    do {                                                                                           \
        unsigned char* b_ = lds + (buf_) * BUF_BYTES;                                              \
        *(uint4*)(b_ + kdst[0]) = r0_; *(uint4*)(b_ + kdst[1]) = r1_; *(uint4*)(b_ + kdst[2]) = r2_; \
        *(uint4*)(b_ + kdst[3]) = r3_; *(uint4*)(b_ + tdst[0]) = r4_; *(uint4*)(b_ + tdst[1]) = r5_; \
    } while (0)

    // ---- LDS-DMA issue of tile q into buffer `dbuf` (DMA instantiation): wave w, piece n: stream sub-tile ct = 4w + n
    // (lane l -> key l >> 1, stored half l & 1 = source half ^ key bit 3), table rows (2w + n) * 4 .. + 3 (lane l -> row l >> 4,
    // stored chunk l & 15 = source chunk ^ (row & 15))
    // (buffer form: descriptor + 32-bit per-lane offset + scalar tile offset - no 64-bit per-lane addresses to keep alive)
    const int wv_ = __builtin_amdgcn_readfirstlane(w);
    int dk_off[4], dt_off[2];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int ct = wv_ * 4 + n, row = lane >> 1, half = (lane & 1) ^ ((row >> 3) & 1);
        dk_off[n] = row * (C * 2) + (ct * 2 + half) * 16;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int row = (wv_ * 2 + n) * 4 + (lane >> 4), c = (lane & 15) ^ (row & 15);
        dt_off[n] = row * (CI * 2) + c * 16;
    }
    auto dma_tile = [&](int q, int dbuf) {
        rsrc_t rk_; int ko_, to_;
        tile_src(q, rk_, ko_, to_);
        unsigned char* base = lds + dbuf * BUF_BYTES;
#pragma unroll
        for (int n = 0; n < 4; ++n)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk_, (__attribute__((address_space(3))) void*)(base + (wv_ * 4 + n) * SUBT), 16,
                                                 dk_off[n], ko_, 0, 0);
#pragma unroll
        for (int n = 0; n < 2; ++n)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rtab, (__attribute__((address_space(3))) void*)(base + KEYS_BYTES + (wv_ * 2 + n) * 1024), 16,
                                                 dt_off[n], to_, 0, 0);
    };

    // ---- per-lane LDS read offsets
    int koff[2], toff[2], troff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int key = b * 16 + fr;
        koff[b] = (fg >> 1) * SUBT + key * 32 + (((fg & 1) ^ ((key >> 3) & 1)) << 4);
        toff[b] = KEYS_BYTES + key * 256 + (((w * 4 + fg) ^ fr) << 4);
        const int tkey = b * 16 + fg * 4 + (fr >> 2), cc = fr & 3;
        troff[b] = tkey * 32 + ((((cc >> 1) ^ (fg >> 1)) & 1) << 4) + (cc & 1) * 8;
    }

    uint4 qb[8], qd;
    f32x4_t acc[16];
    float m = NEG_BIG, l = 0.f;
    int item = 0;

    // every global load of the steady state is unconditional (tile index clamped to the last tile): the compiler can
    // then count the younger loads / stores and wait with vmcnt(n > 0) instead of draining the prefetch (common.h)
    if constexpr (DMA) {
        dma_tile(0, 0);
    } else {
        FA_LOAD(ra0, ra1, ra2, ra3, ra4, ra5, 0);
        FA_STORE(ra0, ra1, ra2, ra3, ra4, ra5, 0);
        FA_LOAD(ra0, ra1, ra2, ra3, ra4, ra5, min(1, nq - 1));
    }
    __syncthreads();

    int q = 0, buf = 0;
    auto iteration = [&](uint4& p0, uint4& p1, uint4& p2, uint4& p3, uint4& p4, uint4& p5, uint4& f0, uint4& f1, uint4& f2,
                         uint4& f3, uint4& f4, uint4& f5) {
        if constexpr (DMA) { if (q + 1 < nq) dma_tile(q + 1, buf ^ 1); }
        else FA_LOAD(f0, f1, f2, f3, f4, f5, min(q + 2, nq - 1));
        const int tt = q & (TPI - 1);
        if (tt == 0) {                                   // new work item: folded queries of this wave's two heads
            item = (int)blockIdx.x + (q >> tpi_sh) * (int)gridDim.x;
            const int p = item >> ks_sh;
            const u16* qp = a.qprime + (((long)p * 64) + w * 16 + fr) * C + fg * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qb[ks] = *(const uint4*)(qp + ks * 32);
            const int hh = fr >> 3, t = fr & 7;
            qd = make_uint4(0, 0, 0, 0);
            if ((fg >> 1) == hh && t < a.Nt)
                qd = *(const uint4*)(a.qtok + ((long)p * a.Nt + t) * CI + (2 * w + hh) * 16 + (fg & 1) * 8);
            wait_vmem_all();                                 // wait here (once per item), not in the steady state
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) acc[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            m = NEG_BIG; l = 0.f;
        }
        const unsigned char* B = lds + buf * BUF_BYTES;
        // ---- scores of the two 16-key blocks
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const uint4 a0 = *(const uint4*)(B + koff[0] + ks * 2 * SUBT);
            const uint4 a1 = *(const uint4*)(B + koff[1] + ks * 2 * SUBT);
            s0 = mfma16d(a0, qb[ks], s0);
            s1 = mfma16d(a1, qb[ks], s1);
        }
        s0 = mfma16d(*(const uint4*)(B + toff[0]), qd, s0);
        s1 = mfma16d(*(const uint4*)(B + toff[1]), qd, s1);
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
        // ---- O'^T += keys^T P^T over the 16 channel tiles
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const uint2 t0 = lds_tr16(B + ct * SUBT + troff[0]);
            const uint2 t1 = lds_tr16(B + ct * SUBT + troff[1]);
            f32x4_t o = acc[ct];
            o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
            acc[ct] = mfma16d(make_uint4(t0.x, t0.y, t1.x, t1.y), pb, o);
        }
        if (tt == TPI - 1) {                             // work item complete
            float lt = l;
            lt += __shfl_xor(lt, 16); lt += __shfl_xor(lt, 32);
            if (a.KS == 1) {
                // whole prompt in this workgroup: out_{h,t} = (O'_{h,t} / l) Wv_h^T + bv_h right here (no fp32 partials, no
                // finish kernel).  O'^T (rows = channels, columns = (h,t)) is already the MFMA B operand (k-slot map as
                // above: slots i < 4 <-> channel tile 2 ks, i >= 4 <-> tile 2 ks + 1); it enters as bf16 hi + lo, the A
                // operand = 16 rows of Wv of one of the wave's two heads, read with the same channel permutation.
                const float inv = 1.f / lt;
                const int p = item;
                f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
                const u16* w0 = a.wv + (long)((2 * w) * 16 + fr) * C + fg * 4;
                const u16* w1 = w0 + 16 * C;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    float c8[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { c8[r] = acc[2 * ks][r] * inv; c8[4 + r] = acc[2 * ks + 1][r] * inv; }
                    const uint4 bh = make_uint4(pack2d(c8[0], c8[1]), pack2d(c8[2], c8[3]), pack2d(c8[4], c8[5]), pack2d(c8[6], c8[7]));
                    const uint32_t hw[4] = {bh.x, bh.y, bh.z, bh.w};
                    float l8[8];
#pragma unroll
                    for (int x = 0; x < 4; ++x) { l8[2 * x] = c8[2 * x] - d2f((u16)(hw[x] & 0xffff)); l8[2 * x + 1] = c8[2 * x + 1] - d2f((u16)(hw[x] >> 16)); }
                    const uint4 bl = make_uint4(pack2d(l8[0], l8[1]), pack2d(l8[2], l8[3]), pack2d(l8[4], l8[5]), pack2d(l8[6], l8[7]));
                    const uint2 x0 = *(const uint2*)(w0 + (2 * ks) * 16), x1 = *(const uint2*)(w0 + (2 * ks + 1) * 16);
                    const uint2 y0 = *(const uint2*)(w1 + (2 * ks) * 16), y1 = *(const uint2*)(w1 + (2 * ks + 1) * 16);
                    const uint4 a0 = make_uint4(x0.x, x0.y, x1.x, x1.y), a1 = make_uint4(y0.x, y0.y, y1.x, y1.y);
                    o0 = mfma16d(a0, bh, o0); o0 = mfma16d(a0, bl, o0);
                    o1 = mfma16d(a1, bh, o1); o1 = mfma16d(a1, bl, o1);
                }
                const int hh = fr >> 3, t = fr & 7;
                if (t < a.Nt) {                          // rows d = 4 fg + r of head 2w + hh, column = this lane's (head, token)
                    const f32x4_t o = hh ? o1 : o0;
                    const float4 b4 = *(const float4*)(a.bv + (2 * w + hh) * 16 + fg * 4);
                    uint2 pk; pk.x = pack2d(o[0] + b4.x, o[1] + b4.y); pk.y = pack2d(o[2] + b4.z, o[3] + b4.w);
                    *(uint2*)(a.out + ((long)p * a.Nt + t) * CI + (2 * w + hh) * 16 + fg * 4) = pk;
                }
                wait_vmem_all();                         // pins the waits of this (rare) branch, see common.h
            } else {                                     // (m, l, O') partial of this key split
                const long col = (long)item * 64 + w * 16 + fr;
                if (fg == 0) { a.stats[col * 2] = m; a.stats[col * 2 + 1] = lt; }
                float* op = a.opart + col * C + fg * 4;
#pragma unroll
                for (int ct = 0; ct < 16; ++ct)
                    *(float4*)(op + ct * 16) = make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]);
            }
        }
        if constexpr (!DMA) { if (q + 1 < nq) FA_STORE(p0, p1, p2, p3, p4, p5, buf ^ 1); }
        __syncthreads();                                 // (with a DMA in flight the barrier's fence waits vmcnt(0): tile q+1 landed)
        buf ^= 1;
    };
    while (true) {
        iteration(ra0, ra1, ra2, ra3, ra4, ra5, rb0, rb1, rb2, rb3, rb4, rb5);
        if (++q >= nq) break;
        iteration(rb0, rb1, rb2, rb3, rb4, rb5, ra0, ra1, ra2, ra3, ra4, ra5);
        if (++q >= nq) break;
    }
#undef FA_LOAD
#undef FA_STORE
}

// merge the key splits, normalise, per-head value projection: out[p][t][h*16 + d] = ctx_{h,t} . Wv[h*16 + d] + bv
__global__ __launch_bounds__(128) void fold_finish_kernel(const float* __restrict__ opart, const float* __restrict__ stats,
                                                          int KS, int Nt, const u16* __restrict__ wv,
                                                          const float* __restrict__ bv, u16* __restrict__ out) {
    __shared__ float ctx[8][C + 4];
    __shared__ float scale[8][16];
    const int p = blockIdx.x >> 3, h = blockIdx.x & 7, tid = threadIdx.x;
    if (tid < 8) {
        const int t = tid;
        float mm = NEG_BIG;
        for (int k = 0; k < KS; ++k) mm = fmaxf(mm, stats[(((long)p * KS + k) * 64 + h * 8 + t) * 2]);
        float ll = 0.f;
        for (int k = 0; k < KS; ++k) {
            const float* st = stats + (((long)p * KS + k) * 64 + h * 8 + t) * 2;
            const float e = __expf(st[0] - mm);
            scale[t][k] = e; ll += e * st[1];
        }
        const float inv = 1.f / ll;
        for (int k = 0; k < KS; ++k) scale[t][k] *= inv;
    }
    __syncthreads();
    for (int idx = tid; idx < 8 * C; idx += 128) {
        const int t = idx >> 8, c = idx & (C - 1);
        float v = 0.f;
        for (int k = 0; k < KS; ++k) v += scale[t][k] * opart[(((long)p * KS + k) * 64 + h * 8 + t) * C + c];
        ctx[t][c] = v;
    }
    __syncthreads();
    const int t = tid >> 4, d = tid & 15;
    if (t >= Nt) return;
    const u16* wr = wv + (long)(h * 16 + d) * C;
    float acc = bv[h * 16 + d];
#pragma unroll 4
    for (int c8 = 0; c8 < C / 8; ++c8) {
        const uint4 wq = *(const uint4*)(wr + c8 * 8);
        const uint32_t ww[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            acc = fmaf(ctx[t][c8 * 8 + 2 * x], d2f((u16)(ww[x] & 0xffff)), acc);
            acc = fmaf(ctx[t][c8 * 8 + 2 * x + 1], d2f((u16)(ww[x] >> 16)), acc);
        }
    }
    out[((long)p * Nt + t) * CI + h * 16 + d] = f2d(acc);
}

// ============================================================================================================
// Image -> token cross attention + out-projection + residual + LayerNorm, same folding:
//     S[j][(h,t)] = ((keys_j + pe_j) Wq_h^T + bq_h) . k_{t,h} / 4 = keys_j . K'_{h,t} / 4 + tabQ_{j,h} . k_{t,h} / 4
//         K'_{h,t} = Wq_h^T k_{t,h},  tabQ = pe Wq^T + bq
//     P = softmax over the <= 8 prompt tokens t of each head
//     keys_j <- LayerNorm(keys_j + sum_{(h,t)} P[j][(h,t)] V'_{h,t} + bo),   V'_{h,t} = Wo[:, 16h : 16h+16] v_{t,h}
// i.e. per 32-token tile two 32 x 256 x 64 products against per-prompt operands K', V' (32 KB each) instead of the two
// 256 x 128 weight matrices - half the MFMA work, and the stationary operands of one prompt fit the registers of a
// 4-wave workgroup (32 + 32 VGPRs per lane), so two workgroups per CU stream independently (the weights-stationary
// form needed a 16-wave wave-specialised workgroup, declayer.hip, kept for > 8 tokens per prompt).
//
// Wave w: scores of heads 2w, 2w+1 (transposed form: rows (h,t), columns image tokens; the softmax over t is 4 registers
// + one cross-lane exchange), normalised P^T as bf16 to LDS [token][64]; then the 64 output channels 64w .. 64w+63
// (O^T = V'^T P^T, rows channels, columns tokens), residual from the LDS stream tile, LayerNorm statistics across the
// four waves through LDS, result written into the stream tile in place and copied out with coalesced 16-byte stores.
struct I2tArgs {
    const u16* xin; int x_shared;        // bf16 [Px, 4096, 256] (x_shared: every prompt reads prompt 0)
    const u16* kfold;                    // bf16 [P, 64, 256]   K'
    const u16* vfoldT;                   // bf16 [P, 256, 64]   V'^T
    const u16* ktok; int Nt;             // bf16 [P, Nt, 128]
    const u16* tabq;                     // bf16 [4096, 128]
    const float* bo; const float* ln_w; const float* ln_b; float eps;
    int nitems, KS;
    u16* out;                            // bf16 [P, 4096, 256] (may alias xin)
};

constexpr int PT_BYTES = TK * 64 * 2;
// stream tile in LDS: 8 k-step sub-tiles [32 tokens][64 B] (+64 B pad so that the staging stores of one row spread over
// the four bank windows), 16-byte slot' = slot ^ ((token >> 2) & 3): conflict-free b128 operand reads with the k-step as
// an immediate offset
constexpr int SUB_BYTES = TK * 64 + 64, XT_BYTES = 8 * SUB_BYTES;

// phase stamps of workgroup 0 / wave 0 (TIMING instantiation only; msam_debug_i2t_timing): [tile][12] shader-clock values
__device__ unsigned long long g_i2t_stamps[64 * 12];
int g_i2t_timing = 0;

template <bool TIMING>
__global__ __launch_bounds__(NTHR, 2) void fold_i2t_kernel(I2tArgs a) {
#define I2T_STAMP(k_) do { if constexpr (TIMING) { if (blockIdx.x == 0 && threadIdx.x == 0 && q < 64)           \
                                                       g_i2t_stamps[q * 12 + (k_)] = __builtin_readcyclecounter(); } } while (0)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * XT_BYTES + PT_BYTES];
    __shared__ float red[4][TK][2];
    __shared__ __attribute__((aligned(16))) float prm[3][C];                  // bo, ln_w, ln_b
    unsigned char* const PT = lds + 2 * XT_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int ks_sh = __builtin_ctz(a.KS), tpi_sh = 7 - ks_sh;
    const int TPI = 1 << tpi_sh;                                      // tiles per item ((T / TK) / KS)
    const int my_items = (a.nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nq = my_items * TPI;
    if (nq <= 0) return;
    prm[0][tid] = a.bo[tid]; prm[1][tid] = a.ln_w[tid]; prm[2][tid] = a.ln_b[tid];

    // ---- staging: 4 chunks (16 B) of the stream tile per thread
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    int kdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = i * NTHR + tid, row = id >> 5, c = id & 31;
        kdst[i] = (c >> 2) * SUB_BYTES + row * 64 + (((c & 3) ^ ((row >> 2) & 3)) << 4);
    }
    auto tile_pos = [&](int q, int& p, int& key0) {        // KS, TPI are powers of two; results are wave-uniform
        const int item = (int)blockIdx.x + (q >> tpi_sh) * (int)gridDim.x;
        p = __builtin_amdgcn_readfirstlane(item >> ks_sh);
        key0 = __builtin_amdgcn_readfirstlane((item & (a.KS - 1)) * (T >> ks_sh) + (q & (TPI - 1)) * TK);
    };
    const int voff = tid * 16;
    const rsrc_t rtab = make_rsrc(a.tabq, T * CI * 2);
    const int tvoff = (fr * CI + w * 32 + fg * 8) * 2;     // table B fragment: token fr, channels 32w + fg*8 ..
#define FI_LOAD(r0_, r1_, r2_, r3_, q_)                                                            \
    do {                                                                                           \
        int p_, k0_;                                                                               \
        tile_pos(q_, p_, k0_);                                                                     \
        const rsrc_t rx_ = make_rsrc(a.xin + (long)(a.x_shared ? 0 : p_) * T * C, T * C * 2);      \
        const int so_ = k0_ * C * 2;                                                               \
        r0_ = buf_load16(rx_, voff, so_); r1_ = buf_load16(rx_, voff, so_ + NTHR * 16);            \
        r2_ = buf_load16(rx_, voff, so_ + 2 * NTHR * 16); r3_ = buf_load16(rx_, voff, so_ + 3 * NTHR * 16); \
    } while (0)
#define FI_STORE(r0_, r1_, r2_, r3_, buf_)                                                         \
    do {                                                                                           \
        unsigned char* b_ = lds + (buf_) * XT_BYTES;                                               \
        *(uint4*)(b_ + kdst[0]) = r0_; *(uint4*)(b_ + kdst[1]) = r1_; *(uint4*)(b_ + kdst[2]) = r2_; \
        *(uint4*)(b_ + kdst[3]) = r3_;                                                             \
    } while (0)
    // table operand of the NEXT tile (B fragments of tokens fr and 16 + fr), straight from L2
#define FI_TAB(q_)                                                                                 \
    do {                                                                                           \
        int p_, k0_;                                                                               \
        tile_pos(q_, p_, k0_);                                                                     \
        tb0 = buf_load16(rtab, tvoff, k0_ * CI * 2); tb1 = buf_load16(rtab, tvoff, (k0_ + 16) * CI * 2); \
    } while (0)

    uint4 kq[8], vq[4][2], kd, tb0, tb1;
    const int hh_row = fr >> 3, t_row = fr & 7;           // A-operand row fr of this wave's score tile = (head 2w+hh, token t)
    // The residual and the out_proj bias enter the O^T accumulators through the MFMA instead of the VALU (the kernel is
    // VALU-issue bound: PMC): bias = initial accumulator value (C operand of the first MFMA); residual = one more MFMA per
    // output tile with a 16 x 32 slice of the identity as A operand (row r picks channel (i & 1) * 16 + r of the 32-channel
    // k-step 2w + (i >> 1)) and the stream tile's own operand fragment as B - bf16 x 1.0 is exact in the fp32 accumulator.
    uint4 ida[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int target = par * 16 + fr;                 // k index of the one in row fr
        uint32_t wv[4] = {0u, 0u, 0u, 0u};
        if ((target >> 3) == fg) wv[(target & 7) >> 1] = (target & 1) ? (MSAM_D16_ONE << 16) : MSAM_D16_ONE;
        ida[par] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
    f32x4_t bo_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 b4 = *(const float4*)(a.bo + (w * 4 + i) * 16 + fg * 4);
        bo_acc[i] = f32x4_t{b4.x, b4.y, b4.z, b4.w};
    }
    wait_vmem_all();
    // per-lane LDS offsets: operand reads (token fr, slot fg), residual / result (token fr, channels (4w+i)*16 + fg*4 ..)
    const int boff = fr * 64 + ((fg ^ ((fr >> 2) & 3)) << 4);
    int xoff[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
        xoff[e] = (2 * w) * SUB_BYTES + fr * 64 + ((((e * 2 + (fg >> 1)) ^ ((fr >> 2) & 3)) & 3) << 4) + (fg & 1) * 8;

    // unconditional (clamped) loads in the steady state, see fold_attn_kernel
    FI_LOAD(ra0, ra1, ra2, ra3, 0);
    FI_TAB(0);
    FI_STORE(ra0, ra1, ra2, ra3, 0);
    FI_LOAD(ra0, ra1, ra2, ra3, min(1, nq - 1));
    __syncthreads();

    int q = 0, buf = 0;
    auto iteration = [&](uint4& p0, uint4& p1, uint4& p2, uint4& p3, uint4& f0, uint4& f1, uint4& f2, uint4& f3) {
        I2T_STAMP(0);
        FI_LOAD(f0, f1, f2, f3, min(q + 2, nq - 1));
        int p, key0;
        tile_pos(q, p, key0);
        if ((q & (TPI - 1)) == 0) {                      // new work item: this prompt's folded operands
            const u16* kp = a.kfold + (((long)p * 64) + w * 16 + fr) * C + fg * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) kq[ks] = *(const uint4*)(kp + ks * 32);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u16* vp = a.vfoldT + (((long)p * C) + (w * 4 + i) * 16 + fr) * 64 + fg * 8;
                vq[i][0] = *(const uint4*)vp; vq[i][1] = *(const uint4*)(vp + 32);
            }
            kd = make_uint4(0, 0, 0, 0);
            if ((fg >> 1) == hh_row && t_row < a.Nt)
                kd = *(const uint4*)(a.ktok + ((long)p * a.Nt + t_row) * CI + (2 * w + hh_row) * 16 + (fg & 1) * 8);
            wait_vmem_all();                                 // wait here (once per item), not in the steady state
        }
        unsigned char* B = lds + buf * XT_BYTES;
        // ---- S^T (rows (h,t) of heads 2w, 2w+1; two 16-token column tiles)
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const uint4 b0 = *(const uint4*)(B + boff + ks * SUB_BYTES);
            const uint4 b1 = *(const uint4*)(B + boff + ks * SUB_BYTES + 16 * 64);
            s0 = mfma16d(kq[ks], b0, s0);
            s1 = mfma16d(kq[ks], b1, s1);
        }
        s0 = mfma16d(kd, tb0, s0);
        s1 = mfma16d(kd, tb1, s1);
        FI_TAB(min(q + 1, nq - 1));
        I2T_STAMP(1);
        // softmax over the 8 tokens of a head: rows fg*4 + r, i.e. token (fg & 1) * 4 + r of head 2w + (fg >> 1)
        {
            float m0 = NEG_BIG, m1 = NEG_BIG;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = (fg & 1) * 4 + r < a.Nt;
                s0[r] = ok ? s0[r] * 0.25f : NEG_BIG; s1[r] = ok ? s1[r] * 0.25f : NEG_BIG;
                m0 = fmaxf(m0, s0[r]); m1 = fmaxf(m1, s1[r]);
            }
            m0 = fmaxf(m0, __shfl_xor(m0, 16)); m1 = fmaxf(m1, __shfl_xor(m1, 16));
            float l0 = 0.f, l1 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { s0[r] = __expf(s0[r] - m0); s1[r] = __expf(s1[r] - m1); l0 += s0[r]; l1 += s1[r]; }
            l0 += __shfl_xor(l0, 16); l1 += __shfl_xor(l1, 16);
            const float i0 = __builtin_amdgcn_rcpf(l0), i1 = __builtin_amdgcn_rcpf(l1);     // 1 ulp; P is rounded to bf16 next
            uint2 q0, q1;
            q0.x = pack2d(s0[0] * i0, s0[1] * i0); q0.y = pack2d(s0[2] * i0, s0[3] * i0);
            q1.x = pack2d(s1[0] * i1, s1[1] * i1); q1.y = pack2d(s1[2] * i1, s1[3] * i1);
            // P^T tile [token][64 (h,t)] bf16, 128-byte rows, chunk' = chunk ^ ((token >> 1) & 7)  ((16 + fr) >> 1 & 7 == fr >> 1 & 7)
            const int po = fr * 128 + (((2 * w + (fg >> 1)) ^ ((fr >> 1) & 7)) << 4) + (fg & 1) * 8;
            *(uint2*)(PT + po) = q0;
            *(uint2*)(PT + po + 16 * 128) = q1;
        }
        I2T_STAMP(2);
        __syncthreads();                                 // (A) P^T complete
        I2T_STAMP(3);
        // ---- O^T for channels 64w .. 64w+63
        f32x4_t o[4][2];
        {
            uint4 pf[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int po = fr * 128 + (((kk * 4 + fg) ^ ((fr >> 1) & 7)) << 4);
                pf[0][kk] = *(const uint4*)(PT + po);
                pf[1][kk] = *(const uint4*)(PT + po + 16 * 128);
            }
            uint4 xk[2][2];                              // stream-tile operand fragments of this wave's k-steps 2w, 2w+1
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int n = 0; n < 2; ++n) xk[j][n] = *(const uint4*)(B + boff + (2 * w + j) * SUB_BYTES + n * 16 * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4_t c = bo_acc[i];
                    c = mfma16d(vq[i][0], pf[n][0], c);
                    c = mfma16d(vq[i][1], pf[n][1], c);
                    c = mfma16d(ida[i & 1], xk[i >> 1][n], c);
                    o[i][n] = c;
                }
        }
        I2T_STAMP(4);
        // LayerNorm partial sums over this wave's 64 channels (o already holds attention + bias + residual).  lane: token
        // n*16 + fr, channels (4w + i)*16 + fg*4 + r  (sub-tile 2w + (i >> 1), slot (i & 1)*2 + (fg >> 1), byte (fg & 1)*8)
        float s1a[2] = {0.f, 0.f}, s2a[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s1a[n] += o[i][n][r]; s2a[n] += o[i][n][r] * o[i][n][r]; }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            s1a[n] += __shfl_xor(s1a[n], 16); s1a[n] += __shfl_xor(s1a[n], 32);
            s2a[n] += __shfl_xor(s2a[n], 16); s2a[n] += __shfl_xor(s2a[n], 32);
            if (fg == 0) { red[w][n * 16 + fr][0] = s1a[n]; red[w][n * 16 + fr][1] = s2a[n]; }
        }
        I2T_STAMP(5);
        __syncthreads();                                 // (B) statistics of all four channel quarters
        I2T_STAMP(6);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int tok = n * 16 + fr;
            const float su = red[0][tok][0] + red[1][tok][0] + red[2][tok][0] + red[3][tok][0];
            const float sq = red[0][tok][1] + red[1][tok][1] + red[2][tok][1] + red[3][tok][1];
            const float mean = su * (1.f / C);
            const float rstd = rsqrtf(fmaxf(sq * (1.f / C) - mean * mean, 0.f) + a.eps);
            const float nmr = -mean * rstd;              // (x - mean) rstd = fma(x, rstd, -mean rstd): two FMAs per value
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int cbase = (w * 4 + i) * 16 + fg * 4;
                const float4 g4 = *(const float4*)&prm[1][cbase], b4 = *(const float4*)&prm[2][cbase];
                uint2 y;
                y.x = pack2d(fmaf(fmaf(o[i][n][0], rstd, nmr), g4.x, b4.x), fmaf(fmaf(o[i][n][1], rstd, nmr), g4.y, b4.y));
                y.y = pack2d(fmaf(fmaf(o[i][n][2], rstd, nmr), g4.z, b4.z), fmaf(fmaf(o[i][n][3], rstd, nmr), g4.w, b4.w));
                *(uint2*)(B + xoff[i & 1] + (i >> 1) * SUB_BYTES + n * 16 * 64) = y;     // in place
            }
        }
        I2T_STAMP(7);
        if (q + 1 < nq) FI_STORE(p0, p1, p2, p3, buf ^ 1);
        I2T_STAMP(8);
        __syncthreads();                                 // (C) updated tile complete, next tile staged
        I2T_STAMP(9);
        {
            const rsrc_t ro = make_rsrc(a.out + (long)p * T * C, T * C * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) buf_store16(*(const uint4*)(B + kdst[i]), ro, voff, key0 * C * 2 + i * NTHR * 16);
        }
        I2T_STAMP(10);
        buf ^= 1;
    };
    while (true) {
        iteration(ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3);
        if (++q >= nq) break;
        iteration(rb0, rb1, rb2, rb3, ra0, ra1, ra2, ra3);
        if (++q >= nq) break;
    }
#undef FI_LOAD
#undef FI_STORE
#undef FI_TAB
#undef I2T_STAMP
}

// V'^T[p][c][h*8 + t] = sum_d Wo[c][h*16 + d] v[p][t][h*16 + d]   (zero for t >= Nt)
__global__ __launch_bounds__(256) void fold_v_kernel(const u16* __restrict__ vtok, const u16* __restrict__ wo, int Nt,
                                                     u16* __restrict__ vfoldT) {
    __shared__ float v[8][CI];
    const int p = blockIdx.x, c = threadIdx.x;
    for (int i = c; i < 8 * CI; i += 256) {
        const int t = i >> 7, d = i & (CI - 1);
        v[t][d] = t < Nt ? d2f(vtok[((long)p * Nt + t) * CI + d]) : 0.f;
    }
    __syncthreads();
    u16* dst = vfoldT + ((long)p * C + c) * 64;
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
            for (int d = 0; d < 16; ++d) { a0 = fmaf(wf[d], v[2 * t2][h * 16 + d], a0); a1 = fmaf(wf[d], v[2 * t2 + 1][h * 16 + d], a1); }
            pk[t2] = pack2d(a0, a1);
        }
        *(uint4*)(dst + h * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
}

int key_splits(int P) {
    int ks = 1;
    while (P * ks < 512 && ks < 16) ks *= 2;
    return ks;
}

}  // namespace

extern "C" int64_t msam_t2i_fold_workspace_bytes(int32_t P) {
    const long ks = key_splits(P);
    return (long)P * 64 * C * 2 + (long)P * ks * 64 * C * 4 + (long)P * ks * 64 * 2 * 4;
}

extern "C" int msam_t2i_fold_attention(const void* keys, int32_t kv_shared, const void* qtok, int32_t P, int32_t Nt,
                                       const void* wk, const void* tabk, const void* wv, const float* bv, void* out,
                                       void* workspace, int64_t workspace_bytes, void* stream) {
    if (!keys || !qtok || !wk || !tabk || !wv || !bv || !out || !workspace || P <= 0) {
        msam_set_error("msam_t2i_fold_attention: null argument");
        return 1;
    }
    if (Nt < 1 || Nt > 8) { msam_set_error("msam_t2i_fold_attention: 1 <= Nt <= 8 tokens per prompt"); return 1; }
    if (workspace_bytes < msam_t2i_fold_workspace_bytes(P)) { msam_set_error("msam_t2i_fold_attention: workspace too small"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    const int KS = key_splits(P);
    char* wp = (char*)workspace;
    u16* qprime = (u16*)wp; wp += (long)P * 64 * C * 2;
    float* opart = (float*)wp; wp += (long)P * KS * 64 * C * 4;
    float* stats = (float*)wp;
    hipLaunchKernelGGL(fold_q_kernel, dim3(P * 8), dim3(256), 0, s, (const u16*)qtok, (const u16*)wk, Nt, qprime);
    if (int e = msam_check_launch("fold_q")) return e;
    FoldArgs a{};
    a.keys = (const u16*)keys; a.kv_shared = kv_shared & 1; a.blocked = (kv_shared >> 1) & 1; a.qprime = qprime; a.qtok = (const u16*)qtok; a.Nt = Nt;
    a.tabk = (const u16*)tabk; a.nitems = P * KS; a.KS = KS; a.opart = opart; a.stats = stats;
    a.wv = (const u16*)wv; a.bv = bv; a.out = (u16*)out;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = a.nitems < 2 * cus ? a.nitems : 2 * cus;
    const double flops = (double)P * T * (2.0 * 64 * C * 2 + 2.0 * 64 * 32);
    const double bytes = (double)((kv_shared & 1) ? 1 : P) * T * C * 2 + (KS > 1 ? (double)P * KS * 64 * C * 4 : 0.0);
    msam_profile_mark2(stream, 1, flops, bytes, 3);
    if (g_fold_attn_dma && !a.blocked) {
        const int grid3 = a.nitems < 3 * cus ? a.nitems : 3 * cus;
        hipLaunchKernelGGL(fold_attn_kernel<true>, dim3(grid3), dim3(NTHR), 0, s, a);
    } else {
        hipLaunchKernelGGL(fold_attn_kernel<false>, dim3(grid), dim3(NTHR), 0, s, a);
    }
    msam_profile_mark2(stream, 0, flops, bytes, 3);
    if (int e = msam_check_launch("fold_attn")) return e;
    if (KS == 1) return 0;                               // the value projection ran in the attention kernel's epilogue
    hipLaunchKernelGGL(fold_finish_kernel, dim3(P * 8), dim3(128), 0, s, opart, stats, KS, Nt, (const u16*)wv, bv, (u16*)out);
    return msam_check_launch("fold_finish");
}

// token-owner form of the same layer (decfold_tok.hip), the default; g_tune_i2t_variant = 0 selects fold_i2t_kernel
int64_t msam_i2t_tok_workspace_bytes(int32_t P);
int msam_i2t_tok_layer(const void* xin, int32_t x_shared, const void* ktok, const void* vtok, int32_t P, int32_t Nt,
                       const void* wq, const void* tabq, const void* wo, const float* bo, const float* ln_w,
                       const float* ln_b, float ln_eps, void* out, void* workspace, int KS, void* stream);
extern int g_tune_i2t_variant;

extern "C" int64_t msam_i2t_fold_workspace_bytes(int32_t P) {
    const int64_t a = (int64_t)P * 64 * C * 2 * 2, b = msam_i2t_tok_workspace_bytes(P);
    return a > b ? a : b;
}

extern "C" int msam_i2t_fold_layer(const void* xin, int32_t x_shared, const void* ktok, const void* vtok, int32_t P,
                                   int32_t Nt, const void* wq, const void* tabq, const void* wo, const float* bo,
                                   const float* ln_w, const float* ln_b, float ln_eps, void* out, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    if (!xin || !ktok || !vtok || !wq || !tabq || !wo || !bo || !ln_w || !ln_b || !out || !workspace || P <= 0) {
        msam_set_error("msam_i2t_fold_layer: null argument");
        return 1;
    }
    if (Nt < 1 || Nt > 8) { msam_set_error("msam_i2t_fold_layer: 1 <= Nt <= 8 tokens per prompt"); return 1; }
    if (workspace_bytes < msam_i2t_fold_workspace_bytes(P)) { msam_set_error("msam_i2t_fold_layer: workspace too small"); return 1; }
    if (g_tune_i2t_variant == 1)
        return msam_i2t_tok_layer(xin, x_shared, ktok, vtok, P, Nt, wq, tabq, wo, bo, ln_w, ln_b, ln_eps, out, workspace,
                                  key_splits(P), stream);
    hipStream_t s = (hipStream_t)stream;
    u16* kfold = (u16*)workspace;
    u16* vfoldT = kfold + (long)P * 64 * C;
    hipLaunchKernelGGL(fold_q_kernel, dim3(P * 8), dim3(256), 0, s, (const u16*)ktok, (const u16*)wq, Nt, kfold);
    if (int e = msam_check_launch("fold_k")) return e;
    hipLaunchKernelGGL(fold_v_kernel, dim3(P), dim3(256), 0, s, (const u16*)vtok, (const u16*)wo, Nt, vfoldT);
    if (int e = msam_check_launch("fold_v")) return e;
    I2tArgs a{};
    a.xin = (const u16*)xin; a.x_shared = x_shared; a.kfold = kfold; a.vfoldT = vfoldT; a.ktok = (const u16*)ktok; a.Nt = Nt;
    a.tabq = (const u16*)tabq; a.bo = bo; a.ln_w = ln_w; a.ln_b = ln_b; a.eps = ln_eps;
    a.KS = key_splits(P); a.nitems = P * a.KS; a.out = (u16*)out;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = a.nitems < 2 * cus ? a.nitems : 2 * cus;
    const double flops = (double)P * T * (2.0 * 64 * C * 2 + 2.0 * 64 * 32);
    const double bytes = (double)(x_shared ? 1 : P) * T * C * 2 + (double)P * T * C * 2;
    msam_profile_mark2(stream, 1, flops, bytes, 2);
    if (g_i2t_timing) hipLaunchKernelGGL(fold_i2t_kernel<true>, dim3(grid), dim3(NTHR), 0, s, a);
    else hipLaunchKernelGGL(fold_i2t_kernel<false>, dim3(grid), dim3(NTHR), 0, s, a);
    msam_profile_mark2(stream, 0, flops, bytes, 2);
    return msam_check_launch("fold_i2t");
}

// Debug hook (tools/i2t_timing.py): enable != 0 routes msam_i2t_fold_layer to the instrumented instantiation; with a non-null
// host buffer [64 * 12] the phase stamps of the last instrumented launch (workgroup 0, wave 0, first 64 tiles) are copied out.
extern "C" int msam_debug_i2t_timing(int32_t enable, uint64_t* host_out) {
    g_i2t_timing = enable;
    if (host_out) {
        if (hipDeviceSynchronize() != hipSuccess ||
            hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_i2t_stamps), sizeof(unsigned long long) * 64 * 12) != hipSuccess) {
            msam_set_error("msam_debug_i2t_timing: copy failed");
            return 2;
        }
    }
    return 0;
}

extern "C" int msam_fold_attn_set_dma(int32_t on) { g_fold_attn_dma = on; return 0; }

// Tuning / A-B hook (tools/, tests): named integer knobs of the decoder stream kernels.  Returns 0, or 1 for an unknown key.
//   "i2t_variant"     1 = token-owner kernel (decfold_tok.hip, default), 0 = fold_i2t_kernel
//   "i2t_wg_per_cu"   workgroups per CU of the token-owner kernel's persistent grid (default 2)
//   "dec_chain"       1 = decoder_run chains layer 0 into layer 1 on a shared source (decfold_tok.hip), 0 = one kernel per stage
//   "dec_chain_min_p" smallest number of prompts for which it does (default 128)
extern int g_tune_i2t_wg_per_cu, g_tune_chain_variant, g_tune_chain_tmask, g_tune_up_gelu16, g_tune_up_ln_two_pass, g_tune_up_centred;
void msam_gemm_set_dbg(int v);
void msam_gemm_set_gw(int delay, int cls);
void msam_gemm_set_g3(int delay);
void msam_gemm_set_g3_epi(int v);
extern int g_tune_tok_fuse;
extern int g_tune_mlp_split_fused;
extern int g_tune_sgemm_bufs, g_tune_srel_mfma, g_tune_sgemm_small_below, g_tune_si2t_dbg, g_tune_si2t_late_us, g_tune_sattn_allh;
int g_tune_dec_chain = 1, g_tune_dec_chain_min_p = 128;
extern "C" int msam_tune_set(const char* key, int32_t value) {
    const std::string k = key ? key : "";
    if (k == "i2t_variant") g_tune_i2t_variant = value;
    else if (k == "i2t_wg_per_cu") g_tune_i2t_wg_per_cu = value;
    else if (k == "dec_chain") g_tune_dec_chain = value;
    else if (k == "chain_variant") g_tune_chain_variant = value;
    else if (k == "chain_tmask") g_tune_chain_tmask = value;
    else if (k == "up_gelu16") g_tune_up_gelu16 = value;
    else if (k == "up_ln_two_pass") g_tune_up_ln_two_pass = value;
    else if (k == "up_centred") g_tune_up_centred = value;
    else if (k == "gemm_dbg") msam_gemm_set_dbg(value);
    else if (k == "gw_delay") msam_gemm_set_gw(value, -1);
    else if (k == "gw_class") msam_gemm_set_gw(-2, value);
    else if (k == "g3_delay") msam_gemm_set_g3(value);
    else if (k == "g3_epi") msam_gemm_set_g3_epi(value);
    else if (k == "tok_fuse") g_tune_tok_fuse = value;
    else if (k == "mlp_split_fused") g_tune_mlp_split_fused = value;
    else if (k == "dec_chain_min_p") g_tune_dec_chain_min_p = value;
    else if (k == "sgemm_bufs") g_tune_sgemm_bufs = value;
    else if (k == "srel_mfma") g_tune_srel_mfma = value;
    else if (k == "sattn_allh") g_tune_sattn_allh = value;
    else if (k == "sgemm_small_below") g_tune_sgemm_small_below = value;
    else if (k == "si2t_dbg") g_tune_si2t_dbg = value;
    else if (k == "si2t_late_us") g_tune_si2t_late_us = value;
    else { msam_set_error("msam_tune_set: unknown key"); return 1; }
    return 0;
}
