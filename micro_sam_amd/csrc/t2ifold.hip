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

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);
void msam_profile_mark2(void* stream, int begin, double flops, double bytes, int family);

namespace {

constexpr int T = 4096, C = 256, CI = 128, TK = 32, NTHR = 256;
constexpr int KEYS_BYTES = TK * C * 2, TAB_BYTES = TK * CI * 2, BUF_BYTES = KEYS_BYTES + TAB_BYTES;
constexpr float NEG_BIG = -1.0e30f;

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
        q[t][d] = t < Nt ? bf2f(qtok[((long)p * Nt + t) * CI + h * 16 + d]) : 0.f;
    }
    __syncthreads();
    float w[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) w[d] = bf2f(wk[(h * 16 + d) * C + c]);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) a = fmaf(q[t][d], w[d], a);
        qprime[(((long)p * 64) + h * 8 + t) * C + c] = f2bf(a);
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
};

__global__ __launch_bounds__(NTHR, 2) void fold_attn_kernel(FoldArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int TPI = (T / TK) / a.KS;                                  // tiles per item
    const int my_items = (a.nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nq = my_items * TPI;
    if (nq <= 0) return;

    // ---- staging: 4 keys chunks + 2 table chunks (16 B) per thread and tile
    uint4 ra0, ra1, ra2, ra3, ra4, ra5, rb0, rb1, rb2, rb3, rb4, rb5;
    int kdst[4], tdst[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = i * NTHR + tid, row = id >> 5, c = id & 31;
        kdst[i] = (c >> 1) * 1024 + row * 32 + (((c & 1) ^ ((row >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = i * NTHR + tid, row = id >> 4, c = id & 15;
        tdst[i] = KEYS_BYTES + row * 256 + ((c ^ (row & 15)) << 4);
    }
    auto tile_src = [&](int q, const u16*& kp, const u16*& tp) {
        const int item = (int)blockIdx.x + (q / TPI) * (int)gridDim.x;
        const int p = item / a.KS, ks = item - p * a.KS;
        const int key0 = ks * (T / a.KS) + (q % TPI) * TK;
        kp = a.keys + ((long)(a.kv_shared ? 0 : p) * T + key0) * C;
        tp = a.tabk + (long)key0 * CI;
    };
#define FA_LOAD(r0_, r1_, r2_, r3_, r4_, r5_, q_)                                                  \
    do {                                                                                           \
        const u16 *kp_, *tp_;                                                                      \
        tile_src(q_, kp_, tp_);                                                                    \
        r0_ = *(const uint4*)(kp_ + (0 * NTHR + tid) * 8); r1_ = *(const uint4*)(kp_ + (1 * NTHR + tid) * 8); \
        r2_ = *(const uint4*)(kp_ + (2 * NTHR + tid) * 8); r3_ = *(const uint4*)(kp_ + (3 * NTHR + tid) * 8); \
        r4_ = *(const uint4*)(tp_ + (0 * NTHR + tid) * 8); r5_ = *(const uint4*)(tp_ + (1 * NTHR + tid) * 8); \
    } while (0)
#define FA_STORE(r0_, r1_, r2_, r3_, r4_, r5_, buf_)                                               \
    do {                                                                                           \
        unsigned char* b_ = lds + (buf_) * BUF_BYTES;                                              \
        *(uint4*)(b_ + kdst[0]) = r0_; *(uint4*)(b_ + kdst[1]) = r1_; *(uint4*)(b_ + kdst[2]) = r2_; \
        *(uint4*)(b_ + kdst[3]) = r3_; *(uint4*)(b_ + tdst[0]) = r4_; *(uint4*)(b_ + tdst[1]) = r5_; \
    } while (0)

    // ---- per-lane LDS read offsets
    int koff[2], toff[2], troff[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int key = b * 16 + fr;
        koff[b] = (fg >> 1) * 1024 + key * 32 + (((fg & 1) ^ ((key >> 3) & 1)) << 4);
        toff[b] = KEYS_BYTES + key * 256 + (((w * 4 + fg) ^ fr) << 4);
        const int tkey = b * 16 + fg * 4 + (fr >> 2), cc = fr & 3;
        troff[b] = tkey * 32 + ((((cc >> 1) ^ (fg >> 1)) & 1) << 4) + (cc & 1) * 8;
    }

    uint4 qb[8], qd;
    f32x4_t acc[16];
    float m = NEG_BIG, l = 0.f;
    int item = 0;

    FA_LOAD(ra0, ra1, ra2, ra3, ra4, ra5, 0);
    FA_STORE(ra0, ra1, ra2, ra3, ra4, ra5, 0);
    if (1 < nq) FA_LOAD(ra0, ra1, ra2, ra3, ra4, ra5, 1);
    __syncthreads();

    int q = 0, buf = 0;
    auto iteration = [&](uint4& p0, uint4& p1, uint4& p2, uint4& p3, uint4& p4, uint4& p5, uint4& f0, uint4& f1, uint4& f2,
                         uint4& f3, uint4& f4, uint4& f5) {
        if (q + 2 < nq) FA_LOAD(f0, f1, f2, f3, f4, f5, q + 2);
        const int tt = q % TPI;
        if (tt == 0) {                                   // new work item: folded queries of this wave's two heads
            item = (int)blockIdx.x + (q / TPI) * (int)gridDim.x;
            const int p = item / a.KS;
            const u16* qp = a.qprime + (((long)p * 64) + w * 16 + fr) * C + fg * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qb[ks] = *(const uint4*)(qp + ks * 32);
            const int hh = fr >> 3, t = fr & 7;
            qd = make_uint4(0, 0, 0, 0);
            if ((fg >> 1) == hh && t < a.Nt)
                qd = *(const uint4*)(a.qtok + ((long)p * a.Nt + t) * CI + (2 * w + hh) * 16 + (fg & 1) * 8);
#pragma unroll
            for (int ct = 0; ct < 16; ++ct) acc[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            m = NEG_BIG; l = 0.f;
        }
        const unsigned char* B = lds + buf * BUF_BYTES;
        // ---- scores of the two 16-key blocks
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const uint4 a0 = *(const uint4*)(B + koff[0] + ks * 2048);
            const uint4 a1 = *(const uint4*)(B + koff[1] + ks * 2048);
            s0 = mfma16(a0, qb[ks], s0);
            s1 = mfma16(a1, qb[ks], s1);
        }
        s0 = mfma16(*(const uint4*)(B + toff[0]), qd, s0);
        s1 = mfma16(*(const uint4*)(B + toff[1]), qd, s1);
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
        pb.x = pack2bf(s0[0], s0[1]); pb.y = pack2bf(s0[2], s0[3]); pb.z = pack2bf(s1[0], s1[1]); pb.w = pack2bf(s1[2], s1[3]);
        // ---- O'^T += keys^T P^T over the 16 channel tiles
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            const uint2 t0 = lds_tr16(B + ct * 1024 + troff[0]);
            const uint2 t1 = lds_tr16(B + ct * 1024 + troff[1]);
            f32x4_t o = acc[ct];
            o[0] *= alpha; o[1] *= alpha; o[2] *= alpha; o[3] *= alpha;
            acc[ct] = mfma16(make_uint4(t0.x, t0.y, t1.x, t1.y), pb, o);
        }
        if (tt == TPI - 1) {                             // work item complete: (m, l, O') partial of this key split
            float lt = l;
            lt += __shfl_xor(lt, 16); lt += __shfl_xor(lt, 32);
            const long col = (long)item * 64 + w * 16 + fr;
            if (fg == 0) { a.stats[col * 2] = m; a.stats[col * 2 + 1] = lt; }
            float* op = a.opart + col * C + fg * 4;
#pragma unroll
            for (int ct = 0; ct < 16; ++ct)
                *(float4*)(op + ct * 16) = make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]);
        }
        if (q + 1 < nq) FA_STORE(p0, p1, p2, p3, p4, p5, buf ^ 1);
        __syncthreads();
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
            acc = fmaf(ctx[t][c8 * 8 + 2 * x], bf2f((u16)(ww[x] & 0xffff)), acc);
            acc = fmaf(ctx[t][c8 * 8 + 2 * x + 1], bf2f((u16)(ww[x] >> 16)), acc);
        }
    }
    out[((long)p * Nt + t) * CI + h * 16 + d] = f2bf(acc);
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
    a.keys = (const u16*)keys; a.kv_shared = kv_shared; a.qprime = qprime; a.qtok = (const u16*)qtok; a.Nt = Nt;
    a.tabk = (const u16*)tabk; a.nitems = P * KS; a.KS = KS; a.opart = opart; a.stats = stats;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = a.nitems < 2 * cus ? a.nitems : 2 * cus;
    const double flops = (double)P * T * (2.0 * 64 * C * 2 + 2.0 * 64 * 32);
    const double bytes = (double)(kv_shared ? 1 : P) * T * C * 2 + (double)P * KS * 64 * C * 4;
    msam_profile_mark2(stream, 1, flops, bytes, 1);
    hipLaunchKernelGGL(fold_attn_kernel, dim3(grid), dim3(NTHR), 0, s, a);
    msam_profile_mark2(stream, 0, flops, bytes, 1);
    if (int e = msam_check_launch("fold_attn")) return e;
    hipLaunchKernelGGL(fold_finish_kernel, dim3(P * 8), dim3(128), 0, s, opart, stats, KS, Nt, (const u16*)wv, bv, (u16*)out);
    return msam_check_launch("fold_finish");
}
