// bf16 x bf16 -> fp32 MFMA GEMM with fused epilogues, C = act(A[M,K] * W[N,K]^T + bias + table + resid).
// Both operands are K-contiguous ("NT"), so A and B fragments are single 16-byte reads.
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// LDS: 2 buffers x (A 16 KB + B 16 KB), rows of 128 B XOR-swizzled per 16-B chunk (common.h swz()).
// Two staging variants (msam_gemm_t.use_glds):
//   0: global_load_dwordx4 -> VGPR -> ds_write_b128, next tile's loads in flight during the MFMA phase
//   1: global_load_lds_dwordx4 (LDS-DMA), swizzle applied on the per-lane SOURCE address (LDS image linear)
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "../../include/msam_hip.h"

float* msam_det_workspace(size_t floats, int slot);                                           // train.hip: library-owned workspaces of the fixed-order reductions
void msam_det_reduce(const float* parts, int nparts, long n, float* out, int accumulate, void* stream);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_CHUNKS = BM * 8;   // 16-B chunks per operand tile

struct Epi {
    const float* bias; const float* table; int table_rows, table_cols; long table_ld;
    const void* resid; int resid_dtype, resid_rows; long ldr;
    int act;
    void* out; int out_dtype; long ldc;
    int out_mode; u16* q; u16* k; u16* v; int heads, head_dim, tokens;
    const float* row_scale; const float* col_scale;      // fp8 operands: C = acc * row_scale[m] * col_scale[n]
    int splitk_len = 0;                                  // > 0: split-K launch (msam_gemm_t.split_k): workgroup blockIdx.y contracts over
                                                         // k in [y * splitk_len, (y + 1) * splitk_len) and stores its fp32 tile to part y of a
                                                         // workspace; the launcher adds the parts in order (no atomics: reproducible)
    unsigned long long* trace = nullptr;                 // gemm2w_kernel debug timeline (msam_gemm_set_trace): 64 words per workgroup
    int direct_epi = 0;                                  // gemm256_kernel: epilogue from the accumulators (epi_direct) instead of the LDS transposition
    int gw_delay = 0, gw_class = 0;                      // gemm2w_kernel: start delay of the second workgroup class (units of s_sleep 100), class rule
    int dbg = 0;                                         // timing experiments (msam_tune_set "gemm_dbg"; WRONG results when != 0): gemm256 only -
                                                         // 1 = no global stores, 2 = no epilogue at all, 4 = no k-loop, 8 / 16 = operands always from k-tile 0
};

// body of the 128 x 128 tile kernel for workgroup `bid_in` of the product (shared by the plain and the grouped launch)
// F16: operands are IEEE fp16 (the mask decoder's 16-bit type, common.h MSAM_DEC_F16) - same tile, staging and epilogue, the
// fp16 MFMA; 16-bit OUTPUTS follow e.out_dtype (MSAM_F16 -> fp16, else bf16) in both instantiations.
MSAM_DEVINL uint32_t pack16(float lo, float hi, int dt) { return dt == MSAM_F16 ? pack2h(lo, hi) : pack2bf(lo, hi); }
MSAM_DEVINL float load16(u16 v, int dt) { return dt == MSAM_F16 ? h2f(v) : bf2f(v); }

template <bool GLDS, bool F16 = false>
__device__ __forceinline__ void gemm_body(const u16* __restrict__ A, long lda, const u16* __restrict__ W, long ldw, int M, int N,
                                          int K, const Epi& e, int bid_in) {
    __shared__ __attribute__((aligned(16))) uint4 lds[2][2][TILE_CHUNKS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM;
    // XCD-aware remap (bijective): consecutive tiles (sharing the A panel) run on one XCD / one L2
    int nwg = tiles_m * tiles_n, bid = bid_in;
    {
        int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int wm = wave >> 1, wn = wave & 1;

    // staging map: thread t covers LDS chunk position (row = pass*32 + t/8, c' = t%8), which holds global
    // chunk c' ^ swz(row)
    const int srow = tid >> 3, scp = tid & 7;
    const u16* a_src[4]; const u16* w_src[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int row = p * 32 + srow;
        int gc = scp ^ swz(row);
        int ar = m0 + row; if (ar > M - 1) ar = M - 1;
        a_src[p] = A + (long)ar * lda + gc * 8;
        w_src[p] = W + (long)(n0 + row) * ldw + gc * 8;
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
    (void)ra0; (void)rw0;

#define MSAM_ISSUE(kt_, buf_)                                                                          \
    do {                                                                                               \
        if constexpr (GLDS) {                                                                          \
            _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                            \
                /* wave-uniform LDS base for this wave & pass; lane l lands at base + l*16 */          \
                uint4* la = &lds[buf_][0][(p * 32 + wave * 8) * 8];                                    \
                uint4* lw = &lds[buf_][1][(p * 32 + wave * 8) * 8];                                    \
                __builtin_amdgcn_global_load_lds(                                                      \
                    (const __attribute__((address_space(1))) void*)(a_src[p] + (long)(kt_) * BK),      \
                    (__attribute__((address_space(3))) void*)la, 16, 0, 0);                            \
                __builtin_amdgcn_global_load_lds(                                                      \
                    (const __attribute__((address_space(1))) void*)(w_src[p] + (long)(kt_) * BK),      \
                    (__attribute__((address_space(3))) void*)lw, 16, 0, 0);                            \
            }                                                                                          \
        } else {                                                                                       \
            ra0 = *(const uint4*)(a_src[0] + (long)(kt_) * BK);                                        \
            ra1 = *(const uint4*)(a_src[1] + (long)(kt_) * BK);                                        \
            ra2 = *(const uint4*)(a_src[2] + (long)(kt_) * BK);                                        \
            ra3 = *(const uint4*)(a_src[3] + (long)(kt_) * BK);                                        \
            rw0 = *(const uint4*)(w_src[0] + (long)(kt_) * BK);                                        \
            rw1 = *(const uint4*)(w_src[1] + (long)(kt_) * BK);                                        \
            rw2 = *(const uint4*)(w_src[2] + (long)(kt_) * BK);                                        \
            rw3 = *(const uint4*)(w_src[3] + (long)(kt_) * BK);                                        \
        }                                                                                              \
    } while (0)
#define MSAM_COMMIT(buf_)                                                                              \
    do {                                                                                               \
        if constexpr (!GLDS) {                                                                         \
            lds[buf_][0][(0 * 32 + srow) * 8 + scp] = ra0;                                             \
            lds[buf_][0][(1 * 32 + srow) * 8 + scp] = ra1;                                             \
            lds[buf_][0][(2 * 32 + srow) * 8 + scp] = ra2;                                             \
            lds[buf_][0][(3 * 32 + srow) * 8 + scp] = ra3;                                             \
            lds[buf_][1][(0 * 32 + srow) * 8 + scp] = rw0;                                             \
            lds[buf_][1][(1 * 32 + srow) * 8 + scp] = rw1;                                             \
            lds[buf_][1][(2 * 32 + srow) * 8 + scp] = rw2;                                             \
            lds[buf_][1][(3 * 32 + srow) * 8 + scp] = rw3;                                             \
        }                                                                                              \
    } while (0)

    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int row = wm * 64 + i * 16 + fr;
                a[i] = lds[buf][0][row * 8 + ((ks * 4 + fg) ^ swz(row))];
                int col = wn * 64 + i * 16 + fr;
                b[i] = lds[buf][1][col * 8 + ((ks * 4 + fg) ^ swz(col))];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = F16 ? mfma16h(a[i], b[j], acc[i][j]) : mfma16(a[i], b[j], acc[i][j]);
        }
    };
    if constexpr (GLDS) {
        MSAM_ISSUE(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) MSAM_ISSUE(kt + 1, buf ^ 1);
            compute(buf);
            __syncthreads();
        }
    } else {
        // register staging TWO k-tiles ahead (two register sets, roles swapped by the 2x unrolled loop): with one tile in
        // flight the k-loop ran at HBM/L2 latency per tile (PMC: 72 % of the wave cycles waiting).  Buffer addressing and
        // unconditional (clamped) loads keep the waits at vmcnt(8), see common.h.
        const rsrc_t ra = make_rsrc(A, (uint32_t)min((long)M * lda * 2, 0xffffffffL));
        const rsrc_t rw = make_rsrc(W, (uint32_t)min((long)N * ldw * 2, 0xffffffffL));
        int aoff[4], woff[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            aoff[p] = (int)((const char*)a_src[p] - (const char*)A);
            woff[p] = (int)((const char*)w_src[p] - (const char*)W);
        }
        uint4 sa0, sa1, sa2, sa3, sw0, sw1, sw2, sw3;          // second register set (first: ra0..rw3)
#define G_LOAD(a0_, a1_, a2_, a3_, w0_, w1_, w2_, w3_, kt_)                                            \
        do {                                                                                           \
            const int so_ = (kt_) * BK * 2;                                                            \
            a0_ = buf_load16(ra, aoff[0], so_); a1_ = buf_load16(ra, aoff[1], so_);                    \
            a2_ = buf_load16(ra, aoff[2], so_); a3_ = buf_load16(ra, aoff[3], so_);                    \
            w0_ = buf_load16(rw, woff[0], so_); w1_ = buf_load16(rw, woff[1], so_);                    \
            w2_ = buf_load16(rw, woff[2], so_); w3_ = buf_load16(rw, woff[3], so_);                    \
        } while (0)
#define G_COMMIT(a0_, a1_, a2_, a3_, w0_, w1_, w2_, w3_, buf_)                                         \
        do {                                                                                           \
            lds[buf_][0][(0 * 32 + srow) * 8 + scp] = a0_; lds[buf_][0][(1 * 32 + srow) * 8 + scp] = a1_; \
            lds[buf_][0][(2 * 32 + srow) * 8 + scp] = a2_; lds[buf_][0][(3 * 32 + srow) * 8 + scp] = a3_; \
            lds[buf_][1][(0 * 32 + srow) * 8 + scp] = w0_; lds[buf_][1][(1 * 32 + srow) * 8 + scp] = w1_; \
            lds[buf_][1][(2 * 32 + srow) * 8 + scp] = w2_; lds[buf_][1][(3 * 32 + srow) * 8 + scp] = w3_; \
        } while (0)
        G_LOAD(ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3, 0);
        G_COMMIT(ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3, 0);
        G_LOAD(ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3, min(1, nk - 1));
        __syncthreads();
        int kt = 0;
        while (true) {
            G_LOAD(sa0, sa1, sa2, sa3, sw0, sw1, sw2, sw3, min(kt + 2, nk - 1));
            compute(kt & 1);
            if (kt + 1 < nk) G_COMMIT(ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3, (kt & 1) ^ 1);
            __syncthreads();
            if (++kt >= nk) break;
            G_LOAD(ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3, min(kt + 2, nk - 1));
            compute(kt & 1);
            if (kt + 1 < nk) G_COMMIT(sa0, sa1, sa2, sa3, sw0, sw1, sw2, sw3, (kt & 1) ^ 1);
            __syncthreads();
            if (++kt >= nk) break;
        }
        wait_vmem_all();                                 // the clamped tail prefetches must not outlive the LDS reuse below
#undef G_LOAD
#undef G_COMMIT
    }

    // ---- epilogue: stage the fp32 C tile through LDS (reusing the operand buffers), then every thread
    // handles 4 consecutive columns of one row per pass -> 16-byte loads of bias/table/resid, 8/16-byte stores
    float* ldsC = (float*)&lds[0][0][0];          // [128][128] fp32 = 64 KB
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ldsC[(wm * 64 + i * 16 + fg * 4 + r) * BN + wn * 64 + j * 16 + fr] = acc[i][j][r];
    __syncthreads();

    const int c4 = (tid & 31) * 4;                // column offset inside the tile
    const int col = n0 + c4;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (e.bias) { float4 b = *(const float4*)(e.bias + col); bias4[0] = b.x; bias4[1] = b.y; bias4[2] = b.z; bias4[3] = b.w; }
    const bool use_table = e.table && col < e.table_cols;   // table_cols % 4 == 0
    // qkv-split constants (4 consecutive columns stay inside one head: head_dim % 4 == 0)
    int which = 0, head = 0, d = 0;
    if (e.out_mode == 1) {
        const int D = e.heads * e.head_dim;
        which = col / D; int rem = col - which * D; head = rem / e.head_dim; d = rem - head * e.head_dim;
    }
    // residual / table rows of all 16 passes are requested up front (the accumulators are dead, their registers are free):
    // issued one pass at a time, every pass paid a full HBM round trip (measured: 25 us of a 37 us tile round for the
    // K = 768 projections)
    const int lr0 = tid >> 5;
    float4 rt[16];
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
        const int row = min(m0 + pass * 8 + lr0, M - 1);
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (use_table) t = *(const float4*)(e.table + (long)(row % e.table_rows) * e.table_ld + col);
        if (e.resid_dtype) {
            const int rr = e.resid_rows ? (row % e.resid_rows) : row;
            if (e.resid_dtype == MSAM_F32) {
                const float4 u = *(const float4*)((const float*)e.resid + (long)rr * e.ldr + col);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            } else {
                const uint2 u = *(const uint2*)((const u16*)e.resid + (long)rr * e.ldr + col);
                t.x += load16((u16)(u.x & 0xffff), e.resid_dtype); t.y += load16((u16)(u.x >> 16), e.resid_dtype);
                t.z += load16((u16)(u.y & 0xffff), e.resid_dtype); t.w += load16((u16)(u.y >> 16), e.resid_dtype);
            }
        }
        rt[pass] = t;
    }
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
        const int lr = pass * 8 + lr0;
        const int row = m0 + lr;
        if (row >= M) break;      // rows only grow with pass: uniform tail, no barrier inside the loop
        float4 c = *(const float4*)(ldsC + lr * BN + c4);
        float v[4] = {(c.x + bias4[0]) + rt[pass].x, (c.y + bias4[1]) + rt[pass].y, (c.z + bias4[2]) + rt[pass].z,
                      (c.w + bias4[3]) + rt[pass].w};
        if (e.act == MSAM_ACT_GELU) {
            const f32x2_t g01 = gelu_erf2(f32x2_t{v[0], v[1]}), g23 = gelu_erf2(f32x2_t{v[2], v[3]});
            v[0] = g01.x; v[1] = g01.y; v[2] = g23.x; v[3] = g23.y;
        } else if (e.act == MSAM_ACT_RELU) {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] = fmaxf(v[x], 0.f);
        }
        if (e.out_mode == 0) {
            if (e.splitk_len > 0) {
                // split-K: this slice's tile goes to ITS part of the workspace (e.out = parts [split_k][M][N]); the launcher adds the parts in
                // slice order (train.hip msam_det_reduce) - no atomics: the weight gradients are the same bits on every run
                *(float4*)((float*)e.out + ((long)blockIdx.y * M + row) * e.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (e.out_dtype == MSAM_F32) {
                *(float4*)((float*)e.out + (long)row * e.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint2 pk; pk.x = pack16(v[0], v[1], e.out_dtype); pk.y = pack16(v[2], v[3], e.out_dtype);
                *(uint2*)((u16*)e.out + (long)row * e.ldc + col) = pk;
            }
        } else if (e.out_mode == 3) {
            // hi + lo operand pairs of the 16-bit output type, rows [hi | lo | hi] of 3 N (ldc = 3 N): the A operand of a product
            // against [Whi | Whi | Wlo] weight rows - the same values norm.hip's cast_split_kernel writes from an fp32 copy of this
            // tile, without the fp32 round trip and the extra launch (the token MLP's ReLU hidden, decoder.hip mlp_split)
            uint2 hi, lo;
            hi.x = pack16(v[0], v[1], e.out_dtype); hi.y = pack16(v[2], v[3], e.out_dtype);
            lo.x = pack16(v[0] - load16((u16)(hi.x & 0xffff), e.out_dtype), v[1] - load16((u16)(hi.x >> 16), e.out_dtype), e.out_dtype);
            lo.y = pack16(v[2] - load16((u16)(hi.y & 0xffff), e.out_dtype), v[3] - load16((u16)(hi.y >> 16), e.out_dtype), e.out_dtype);
            u16* o = (u16*)e.out + (long)row * e.ldc + col;
            const long third = e.ldc / 3;
            *(uint2*)o = hi; *(uint2*)(o + third) = lo; *(uint2*)(o + 2 * third) = hi;
        } else if (e.out_mode == 1) {
            u16* dst = which == 0 ? e.q : (which == 1 ? e.k : e.v);
            const int b = row / e.tokens, t = row - b * e.tokens;
            uint2 pk; pk.x = pack16(v[0], v[1], e.out_dtype); pk.y = pack16(v[2], v[3], e.out_dtype);
            *(uint2*)(dst + ((long)(b * e.heads + head) * e.tokens + t) * e.head_dim + d) = pk;
        } else {
            // out_mode 2 (decoder K|V projection, N == 256): k half stored row-major [M,128]; the v half is
            // written back to LDS (activated, bf16-rounded values) and stored transposed below
            if (col < 128) {
                uint2 pk; pk.x = pack16(v[0], v[1], e.out_dtype); pk.y = pack16(v[2], v[3], e.out_dtype);
                *(uint2*)(e.k + (long)row * 128 + col) = pk;
            } else {
                *(float4*)(ldsC + lr * BN + c4) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    if (e.out_mode == 2 && n0 == 128) {
        // vT[b][d][t]: thread -> (d = tid & 127, token half = tid >> 7): 64 consecutive tokens = 128 B contiguous
        __syncthreads();
        const int dd = tid & 127, th = tid >> 7;
        const int b = m0 / e.tokens, t0 = m0 - b * e.tokens + th * 64;     // tokens % 128 == 0: tile inside one b
        u16* dst = e.v + ((long)b * 128 + dd) * e.tokens + t0;
        for (int i = 0; i < 64; i += 8) {
            if (m0 + th * 64 + i >= M) break;                               // M % 8 == 0 enforced by the launcher
            uint4 pk;
            const float* src = ldsC + (th * 64 + i) * BN + dd;
            pk.x = pack16(src[0], src[BN], e.out_dtype); pk.y = pack16(src[2 * BN], src[3 * BN], e.out_dtype);
            pk.z = pack16(src[4 * BN], src[5 * BN], e.out_dtype); pk.w = pack16(src[6 * BN], src[7 * BN], e.out_dtype);
            *(uint4*)(dst + i) = pk;
        }
    }
}

template <bool GLDS, bool F16 = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const u16* __restrict__ A, long lda, const u16* __restrict__ W,
                                                   long ldw, int M, int N, int K, Epi e) {
    if (e.splitk_len > 0) {                      // split-K: this workgroup's slice of the contraction (operands are K-contiguous)
        A += (long)blockIdx.y * e.splitk_len; W += (long)blockIdx.y * e.splitk_len; K = e.splitk_len;
    }
    gemm_body<GLDS, F16>(A, lda, W, ldw, M, N, K, e, (int)blockIdx.x);
}

// Grouped launch: up to MSAM_GEMM_GROUP_MAX independent small products in ONE launch (blockIdx.y = product, blockIdx.x = its
// tile; the decoder's token-side projections are 7 168-row products of 112 tiles each - latency-bound one at a time).
struct GroupItem { const u16* A; long lda; const u16* W; long ldw; int M, N, K; Epi e; };
struct GroupArgs { GroupItem it[MSAM_GEMM_GROUP_MAX]; };
template <bool F16>
__global__ __launch_bounds__(256, 2) void gemm_group_kernel(GroupArgs g) {
    // the product's parameters are read from the kernel-argument segment itself (constant address space, scalar loads): indexing
    // the by-value argument with blockIdx.y made the compiler copy the whole array to scratch
    (void)g;
#if defined(__HIP_DEVICE_COMPILE__)                  // (address-space-qualified struct copies do not parse in the host pass)
    typedef const __attribute__((address_space(4))) GroupItem* ItemPtr;
    ItemPtr it = (ItemPtr)__builtin_amdgcn_kernarg_segment_ptr() + blockIdx.y;
    const int M = it->M, N = it->N, K = it->K;
    const int tiles = ((M + BM - 1) / BM) * (N / BN);
    if ((int)blockIdx.x >= tiles) return;
    const Epi e = it->e;
    gemm_body<false, F16>(it->A, it->lda, it->W, it->ldw, M, N, K, e, (int)blockIdx.x);
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Large-shape variant (encoder projections, M = tiles * 4096): tile 256 x 256 x 64, 8 waves as 2 (M) x 4 (N), wave tile
// 128 x 64 on v_mfma_f32_32x32x16_bf16 (twice the flops per LDS byte of the 16x16x32 form; the 128 x 128 tile above needs
// 64 B/cycle/CU of operand traffic at MFMA peak, this one 32).  The product is formed TRANSPOSED (A operand = W rows,
// B operand = A rows) so that a lane holds 4 consecutive output columns of one row.  Operands: register staging two
// k-tiles ahead (buffer loads), double-buffered LDS (2 x 64 KB, dynamic); epilogue: the fp32 tile goes through the same
// LDS in two column halves (256 x 128 fp32 = 128 KB, 16-byte chunks XOR-swizzled with the row) and is written out
// row-wise with the residual rows requested up front, exactly like the 128 x 128 kernel.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
template <bool F16 = false>
MSAM_DEVINL f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// 32 x 32 x 64 fp8 (OCP e4m3) through the block-scaled MX instruction with unit block scales (e8m0 127 = 2^0 in every byte):
// the only large-K fp8 MFMA of gfx950 and the only one that runs at twice the bf16 rate.  A lane holds 32 consecutive
// bytes of K for its row (lane >> 5 selects the 32-byte half of the 64-deep step); both operands use the same map, so any
// consistent assignment of tile bytes to (lane half, byte) computes the same contraction.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
MSAM_DEVINL f32x16_t mfma32_f8(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1, f32x16_t c) {
    const i32x8_t av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    const i32x8_t bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0 /* A fp8 e4m3 */, 0 /* B fp8 e4m3 */, 0, 0x7F7F7F7F,
                                                            0, 0x7F7F7F7F);
}
// Epilogue straight from the accumulators of the 128 x 64 wave tile (transposed 32x32x16 product): acc[i][j][4 g + x] =
// C[m0 + wm*128 + j*32 + l31][n0 + wn*64 + i*32 + 8 g + 4 lh + x] - a lane holds 4 consecutive columns of one row (16 B fp32 / 8 B
// 16-bit stores; the lane pair (l, l + 32) and the four g cover 128 / 64 contiguous bytes of a row).  No LDS, no barrier.
template <bool F16>
MSAM_DEVINL void epi_direct(const f32x16_t (&acc)[2][4], int m0, int n0, int wm, int wn, int l31, int lh, int M, const Epi& e) {
    // loads first (one latency per tile, not one per value): the bias of the lane's 8 column groups
    const int colb = n0 + wn * 64 + lh * 4;
    const int Dqkv = e.heads * e.head_dim;
    const bool has_res = e.resid_dtype == MSAM_F32 && e.resid;
    float4 bias4[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bias4[i][g] = e.bias ? *(const float4*)(e.bias + colb + i * 32 + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    // steps (i, j): the residual of step s + 1 is requested before step s is worked on
    float4 rt[2][4];
    auto load_res = [&](int i, int j, float4 (&r)[4]) {
        const int rc = min(m0 + wm * 128 + j * 32 + l31, M - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) r[g] = *(const float4*)((const float*)e.resid + (long)rc * e.ldr + colb + i * 32 + 8 * g);
    };
    if (has_res) load_res(0, 0, rt[0]);
#pragma unroll
    for (int st8 = 0; st8 < 8; ++st8) {
        const int i = st8 >> 2, j = st8 & 3;
        const int row = m0 + wm * 128 + j * 32 + l31;
        const int rc = min(row, M - 1);
        if (has_res && st8 + 1 < 8) load_res((st8 + 1) >> 2, (st8 + 1) & 3, rt[(st8 + 1) & 1]);
        long rowoff = 0;
        if (e.out_mode == 1) {
            const int b = rc / e.tokens, t = rc - b * e.tokens;
            rowoff = ((long)b * e.heads * e.tokens + t) * e.head_dim;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = colb + i * 32 + 8 * g;
            const float4 b4 = bias4[i][g];
            float v[4] = {acc[i][j][4 * g] + b4.x, acc[i][j][4 * g + 1] + b4.y, acc[i][j][4 * g + 2] + b4.z, acc[i][j][4 * g + 3] + b4.w};
            if (has_res) {
                const float4 r4 = rt[st8 & 1][g];
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if (e.act == MSAM_ACT_GELU) {
                const f32x2_t g01 = gelu_erf2(f32x2_t{v[0], v[1]}), g23 = gelu_erf2(f32x2_t{v[2], v[3]});
                v[0] = g01.x; v[1] = g01.y; v[2] = g23.x; v[3] = g23.y;
            } else if (e.act == MSAM_ACT_RELU) {
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = fmaxf(v[x], 0.f);
            }
            if (row < M && !(e.dbg & 1)) {
                if (e.out_mode == 0 && e.out_dtype == MSAM_F32) {
                    *(float4*)((float*)e.out + (long)row * e.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 pk;
                    if constexpr (F16) { pk.x = pack2h(v[0], v[1]); pk.y = pack2h(v[2], v[3]); }
                    else { pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); }
                    if (e.out_mode == 0) {
                        *(uint2*)((u16*)e.out + (long)row * e.ldc + col) = pk;
                    } else {                              // q / k / v head split: column -> (which, head, d)
                        const int which = col / Dqkv, rem = col - which * Dqkv, head = rem / e.head_dim, d = rem - head * e.head_dim;
                        u16* dst = which == 0 ? e.q : (which == 1 ? e.k : e.v);
                        *(uint2*)(dst + rowoff + (long)head * e.tokens * e.head_dim + d) = pk;
                    }
                }
            }
        }
    }
}

constexpr int G2 = 256;                       // tile edge
constexpr int G2_DEFAULT_STAGING = 3;   // measured (tools/gemm_bench.py): 3 > 1 > 0 by 3 - 8 % each on the encoder shapes, LDS-DMA (2) no better
// Timing experiments and rejected kernel forms (gemm_dbg bits - WRONG results by construction -, the two-workgroups-per-CU kernel = staging 4,
// the accumulator-direct epilogue "g3_epi") are reachable only in a library built with -DMSAM_EXPERIMENTS=1
// (`python -m micro_sam_amd.build --experiments`: tools/gemm_probe.py, tools/gemm_bench.py); the production build ignores the knobs.
#ifndef MSAM_EXPERIMENTS
#define MSAM_EXPERIMENTS 0
#endif
int g_tune_gemm_dbg = 0;                      // msam_tune_set "gemm_dbg" (Epi.dbg)
unsigned long long* g_gw_trace = nullptr;
int g_tune_gw_delay = -1, g_tune_gw_class = 0, g_tune_g3_delay = -1, g_tune_g3_epi = 0;   // "g3_epi": 1 = gemm256_kernel's epilogue from the accumulators; "g3_delay": start-phase spread of gemm256_kernel's first wave // msam_tune_set "gw_delay" (-1 = from K) / "gw_class" (gemm2w_kernel)
int g_gemm256_staging = -1;                   // test / tuning hook (msam_gemm256_set_staging), -1 = default / environment
constexpr int G2_LDS = 2 * 2 * G2 * 8 * 16;   // 2 stages x (A, W) x 256 rows x 8 chunks x 16 B = 128 KB

// STAGING: 0 = registers, two k-tiles ahead, LDS write behind the MFMAs (before the barrier);
//          1 = registers, ONE register set: LDS write of tile kt+1 right after the barrier (start of the iteration), loads of
//              tile kt+2 re-issued at once - the write pass overlaps the MFMA phase instead of sitting in front of the barrier;
//          2 = global_load_lds_dwordx4 (LDS-DMA) into the other LDS buffer while the MFMAs run (no staging registers, no
//              ds_write pass; swizzle applied on the per-lane source address)
// FP8: A and W are fp8 e4m3 bytes (lda / ldw / K in elements = bytes), a k-tile is 128 elements (the same 128-byte LDS
// rows, swizzle and staging map as bf16), two 32 x 32 x 64 MX MFMAs per 128-byte row instead of four 32 x 32 x 16 bf16 ones:
// the same kernel time per byte, twice the elements per byte.  Row / column scales are applied in the epilogue.
// F16: IEEE fp16 operands and 16-bit outputs (the image encoder's fp16 mode, msam_encoder_t.dtype16): same kernel, the fp16 MFMA.
template <int STAGING, bool FP8 = false, bool F16 = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const u16* __restrict__ A, long lda, const u16* __restrict__ W,
                                                         long ldw, int M, int N, int K, Epi e) {
    constexpr int ESZ = FP8 ? 1 : 2;              // bytes per operand element
    constexpr int BKE = 128 / ESZ;                // elements per k-tile (128-byte rows)
    extern __shared__ __attribute__((aligned(16))) uint4 dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles_n = N / G2, tiles_m = (M + G2 - 1) / G2;
    int nwg = tiles_m * tiles_n, bid = blockIdx.x;
    {
        int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * G2, n0 = tile_n * G2;
    const int wm = wave >> 2, wn = wave & 3;
    auto stage = [&](int buf, int op) -> uint4* { return dyn + (buf * 2 + op) * (G2 * 8); };
    // De-phasing of the CUs (round 3): every workgroup of the first dispatch wave (one per CU) starts a class-dependent fraction of a
    // tile period late.  Started together, all CUs walk their tiles in lockstep: nothing is written during the k-loops and every CU
    // writes its 128 KB tile (or reads and writes its fp32 residual tile) at the same moment - an HBM burst that the k-loops of the
    // other classes now cover.  Later workgroups inherit the phase of the workgroup whose CU they take over.
    if (e.gw_delay > 0 && (int)blockIdx.x < e.gw_class) {
        const int cls = (blockIdx.x >> 3) & 3;
        for (int d = 0; d < cls * e.gw_delay; ++d) __builtin_amdgcn_s_sleep(100);
    }

    // staging map: thread t covers LDS chunk position (row = p*64 + t/8, c' = t%8) <- global chunk c' ^ swz(row)
    const int srow = tid >> 3, scp = tid & 7;
    int aoff[4], woff[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = p * 64 + srow, gc = scp ^ swz(row);
        const int ar = min(m0 + row, M - 1);
        aoff[p] = (int)((long)ar * lda * ESZ + gc * 16);
        woff[p] = (int)((long)(n0 + row) * ldw * ESZ + gc * 16);
    }
    const rsrc_t ra = make_rsrc(A, (uint32_t)min((long)M * lda * ESZ, 0xffffffffL));
    const rsrc_t rw = make_rsrc(W, (uint32_t)min((long)N * ldw * ESZ, 0xffffffffL));
    uint4 xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, ya0, ya1, ya2, ya3, yw0, yw1, yw2, yw3;
#define G2_LOAD(a0_, a1_, a2_, a3_, w0_, w1_, w2_, w3_, kt_)                                           \
    do {                                                                                               \
        const int so_ = (kt_) * 128;                                                                   \
        a0_ = buf_load16(ra, aoff[0], so_); a1_ = buf_load16(ra, aoff[1], so_);                        \
        a2_ = buf_load16(ra, aoff[2], so_); a3_ = buf_load16(ra, aoff[3], so_);                        \
        w0_ = buf_load16(rw, woff[0], so_); w1_ = buf_load16(rw, woff[1], so_);                        \
        w2_ = buf_load16(rw, woff[2], so_); w3_ = buf_load16(rw, woff[3], so_);                        \
    } while (0)
#define G2_COMMIT(a0_, a1_, a2_, a3_, w0_, w1_, w2_, w3_, buf_)                                        \
    do {                                                                                               \
        uint4* sa_ = stage(buf_, 0) + srow * 8 + scp; uint4* sw_ = stage(buf_, 1) + srow * 8 + scp;    \
        sa_[0] = a0_; sa_[64 * 8] = a1_; sa_[128 * 8] = a2_; sa_[192 * 8] = a3_;                       \
        sw_[0] = w0_; sw_[64 * 8] = w1_; sw_[128 * 8] = w2_; sw_[192 * 8] = w3_;                       \
    } while (0)

    f32x16_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;

    const int nk = (e.dbg & 4) ? 1 : K / BKE;
    auto compute = [&](int buf) {
        const uint4* la = stage(buf, 0);
        const uint4* lw = stage(buf, 1);
        if constexpr (FP8) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {                 // two 64-byte steps per row; lane half lh owns 32 bytes of a step
                const int c0 = s2 * 4 + lh * 2;
                uint4 wf0[2], wf1[2], af0[4], af1[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = wn * 64 + i * 32 + l31;
                    wf0[i] = lw[row * 8 + (c0 ^ swz(row))]; wf1[i] = lw[row * 8 + ((c0 + 1) ^ swz(row))];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wm * 128 + j * 32 + l31;
                    af0[j] = la[row * 8 + (c0 ^ swz(row))]; af1[j] = la[row * 8 + ((c0 + 1) ^ swz(row))];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma32_f8(wf0[i], wf1[i], af0[j], af1[j], acc[i][j]);
            }
            return;
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            uint4 wf[2], af[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wn * 64 + i * 32 + l31;
                wf[i] = lw[row * 8 + ((s4 * 2 + lh) ^ swz(row))];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wm * 128 + j * 32 + l31;
                af[j] = la[row * 8 + ((s4 * 2 + lh) ^ swz(row))];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma32<F16>(wf[i], af[j], acc[i][j]);
        }
    };
    if constexpr (STAGING == 0) {
        G2_LOAD(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, 0);
        G2_COMMIT(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, 0);
        G2_LOAD(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, min(1, nk - 1));
        __syncthreads();
        int kt = 0;
        while (true) {
            G2_LOAD(ya0, ya1, ya2, ya3, yw0, yw1, yw2, yw3, min(kt + 2, nk - 1));
            compute(kt & 1);
            if (kt + 1 < nk) G2_COMMIT(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, (kt & 1) ^ 1);
            __syncthreads();
            if (++kt >= nk) break;
            G2_LOAD(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, min(kt + 2, nk - 1));
            compute(kt & 1);
            if (kt + 1 < nk) G2_COMMIT(ya0, ya1, ya2, ya3, yw0, yw1, yw2, yw3, (kt & 1) ^ 1);
            __syncthreads();
            if (++kt >= nk) break;
        }
        wait_vmem_all();
    } else if constexpr (STAGING == 1) {
        (void)ya0; (void)yw0;
        G2_LOAD(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, 0);
        G2_COMMIT(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, 0);
        G2_LOAD(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, min(1, nk - 1));
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt+1 (in flight during the previous iteration) -> the buffer every wave finished reading before the barrier
            if (kt + 1 < nk) G2_COMMIT(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, (kt & 1) ^ 1);
            G2_LOAD(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, min(kt + 2, nk - 1));
            compute(kt & 1);
            __syncthreads();
        }
        wait_vmem_all();
    } else if constexpr (STAGING == 3) {
        // Ping-pong of the two M-halves of the workgroup (waves 0-3 = group A, waves 4-7 = group B; every SIMD holds one wave
        // of each).  A k-tile is four slots per wave: R0 M0 R1 M1 - R = the LDS traffic of one k-half (12 ds_read_b128 of the
        // fragments, a quarter of the wave's staging writes of tile k+1 and the re-issue of its loads for tile k+2), M = the
        // 16 MFMAs of that k-half on fragments already in registers.  One barrier per slot; group B runs ONE slot behind
        // group A (an extra barrier in front of its loop, one behind A's), so on every SIMD one wave is in an M slot (MFMA
        // pipe busy back to back, raised priority) while the other does the LDS / global work of an R slot, instead of both
        // reading and then both multiplying.  LDS protocol: tile k+1 is written in the R slots of iteration k (slots 4k .. 4k+3)
        // into the buffer whose last read (group B, R1 of tile k-1) ended in slot 4k-1; its first read is in slot 4k+4.
        (void)ya0; (void)yw0; (void)ya1; (void)yw1; (void)ya2; (void)yw2; (void)ya3; (void)yw3;
        G2_LOAD(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, 0);
        G2_COMMIT(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, 0);
        G2_LOAD(xa0, xa1, xa2, xa3, xw0, xw1, xw2, xw3, min(1, nk - 1));
        __syncthreads();
        const bool group_b = wm == 1;
        if (group_b) { __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
        uint4 wfr[2][2], afr[2][4];
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            const uint4* la = stage(buf, 0);
            const uint4* lw = stage(buf, 1);
            uint4* sa = stage(buf ^ 1, 0) + srow * 8 + scp;
            uint4* sw = stage(buf ^ 1, 1) + srow * 8 + scp;
            const int so = (e.dbg & 8) ? 0 : min(kt + 2, nk - 1) * 128;      // dbg 8: every k-tile = tile 0 (cache hits: is the k-loop bound by the operand stream?)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // ---- R slot of k-half h
                if (kt + 1 < nk) {
                    if (h == 0) { sa[0] = xa0; sa[64 * 8] = xa1; sw[0] = xw0; sw[64 * 8] = xw1; }
                    else { sa[128 * 8] = xa2; sa[192 * 8] = xa3; sw[128 * 8] = xw2; sw[192 * 8] = xw3; }
                }
                if (h == 0) {
                    xa0 = buf_load16(ra, aoff[0], so); xa1 = buf_load16(ra, aoff[1], so);
                    xw0 = buf_load16(rw, woff[0], so); xw1 = buf_load16(rw, woff[1], so);
                } else {
                    xa2 = buf_load16(ra, aoff[2], so); xa3 = buf_load16(ra, aoff[3], so);
                    xw2 = buf_load16(rw, woff[2], so); xw3 = buf_load16(rw, woff[3], so);
                }
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    // 16-byte chunk of the 128-byte row: bf16 = k-step (2h + q2) of 16 elements, lane half lh; fp8 = the lane
                    // half's 32 bytes (chunks 2 lh, 2 lh + 1) of the 64-byte step h
                    const int ch = FP8 ? h * 4 + lh * 2 + q2 : (2 * h + q2) * 2 + lh;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int row = wn * 64 + i * 32 + l31;
                        wfr[q2][i] = lw[row * 8 + (ch ^ swz(row))];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = wm * 128 + j * 32 + l31;
                        afr[q2][j] = la[row * 8 + (ch ^ swz(row))];
                    }
                }
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                // ---- M slot
                __builtin_amdgcn_s_setprio(1);
                if constexpr (FP8) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = mfma32_f8(wfr[0][i], wfr[1][i], afr[0][j], afr[1][j], acc[i][j]);
                } else {
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[i][j] = mfma32<F16>(wfr[q2][i], afr[q2][j], acc[i][j]);
                }
                __builtin_amdgcn_s_setprio(0);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!group_b) { __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
        wait_vmem_all();
    } else {
        (void)xa0; (void)xw0; (void)ya0; (void)yw0;
        // per-lane source pointers (swizzled chunk), wave-uniform LDS destination: wave w, pass p covers rows p*64 + w*8 .. +7
        auto issue = [&](int kt, int buf) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                uint4* la = stage(buf, 0) + (p * 64 + wave * 8) * 8;
                uint4* lw = stage(buf, 1) + (p * 64 + wave * 8) * 8;
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)((const char*)A + aoff[p] + (long)kt * 128),
                    (__attribute__((address_space(3))) void*)la, 16, 0, 0);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)((const char*)W + woff[p] + (long)kt * 128),
                    (__attribute__((address_space(3))) void*)lw, 16, 0, 0);
            }
        };
        issue(0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) issue(kt + 1, (kt & 1) ^ 1);
            compute(kt & 1);
            __syncthreads();
        }
    }
#undef G2_LOAD
#undef G2_COMMIT

    if (e.dbg & 2) return;
    if constexpr (!FP8) {
        if (e.direct_epi) { epi_direct<F16>(acc, m0, n0, wm, wn, l31, lh, M, e); return; }
    }
    // ---- epilogue in two column halves of 128: C^T accumulators (row = n, column = m) -> ldsC[m][128] fp32, chunk-swizzled
    float* ldsC = (float*)dyn;
    const int c4 = tid & 31, rg = tid >> 5;                  // 16-byte column chunk, row group (16 rows per pass)
    for (int h = 0; h < 2; ++h) {
        if ((wn >> 1) == h) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = wm * 128 + j * 32 + l31;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int chunk = ((wn & 1) * 64 + i * 32 + 8 * g + lh * 4) >> 2;
                        *(float4*)(ldsC + m * 128 + ((chunk ^ (m & 31)) << 2)) =
                            make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    }
                }
        }
        __syncthreads();
        const int col = n0 + h * 128 + c4 * 4;
        float bias4[4] = {0.f, 0.f, 0.f, 0.f};
        if (e.bias) { const float4 b = *(const float4*)(e.bias + col); bias4[0] = b.x; bias4[1] = b.y; bias4[2] = b.z; bias4[3] = b.w; }
        float cs4[4] = {1.f, 1.f, 1.f, 1.f};
        if constexpr (FP8) { const float4 c = *(const float4*)(e.col_scale + col); cs4[0] = c.x; cs4[1] = c.y; cs4[2] = c.z; cs4[3] = c.w; }
        int which = 0, head = 0, d = 0;
        if (e.out_mode == 1) {
            const int D = e.heads * e.head_dim;
            which = col / D; const int rem = col - which * D; head = rem / e.head_dim; d = rem - head * e.head_dim;
        }
        // per-thread row bookkeeping ONCE per column half (round 6): a pass advances 16 rows - pointers advance by 16 rows' pitch, the q / k / v
        // store's (image, token) pair by 16 tokens with a wrap - instead of a 64-bit multiplication and a division per pass (the epilogue was
        // 30 - 45 % on top of a K = 768 k-loop, much of it this index arithmetic; full tiles only: the ragged last tile takes the clamped form)
        const bool full_tile = m0 + G2 <= M;
        const long step_r = 16L * e.ldr, step_c = 16L * e.ldc;
        const float* rptr = e.resid_dtype == MSAM_F32 ? (const float*)e.resid + (long)(m0 + rg) * e.ldr + col : nullptr;
        char* optr = e.out_mode == 0 ? (char*)e.out + ((long)(m0 + rg) * e.ldc + col) * (e.out_dtype == MSAM_F32 ? 4 : 2) : nullptr;
        const long step_o = step_c * (e.out_dtype == MSAM_F32 ? 4 : 2);
        int tb = 0, tt = 0;
        if (e.out_mode != 0) { tb = (m0 + rg) / e.tokens; tt = (m0 + rg) - tb * e.tokens; }
        for (int grp = 0; grp < 2; ++grp) {              // residual rows of 8 passes at a time (register budget)
        float4 rt[8];
        if (e.resid_dtype == MSAM_F32) {
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                if (full_tile) rt[ps] = *(const float4*)(rptr + (grp * 8 + ps) * step_r);
                else {
                    const int row = min(m0 + (grp * 8 + ps) * 16 + rg, M - 1);
                    rt[ps] = *(const float4*)((const float*)e.resid + (long)row * e.ldr + col);
                }
            }
        }
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int pass = grp * 8 + ps;
            const int lr = pass * 16 + rg, row = m0 + lr;
            const float4 c = *(const float4*)(ldsC + lr * 128 + ((c4 ^ (lr & 31)) << 2));
            float v[4];
            if constexpr (FP8) {
                const float rs = e.row_scale[min(row, M - 1)];
                v[0] = fmaf(c.x * rs, cs4[0], bias4[0]); v[1] = fmaf(c.y * rs, cs4[1], bias4[1]);
                v[2] = fmaf(c.z * rs, cs4[2], bias4[2]); v[3] = fmaf(c.w * rs, cs4[3], bias4[3]);
            } else {
                v[0] = c.x + bias4[0]; v[1] = c.y + bias4[1]; v[2] = c.z + bias4[2]; v[3] = c.w + bias4[3];
            }
            if (e.resid_dtype == MSAM_F32) { v[0] += rt[ps].x; v[1] += rt[ps].y; v[2] += rt[ps].z; v[3] += rt[ps].w; }
            if (e.act == MSAM_ACT_GELU) {
                const f32x2_t g01 = gelu_erf2(f32x2_t{v[0], v[1]}), g23 = gelu_erf2(f32x2_t{v[2], v[3]});
                v[0] = g01.x; v[1] = g01.y; v[2] = g23.x; v[3] = g23.y;
            } else if (e.act == MSAM_ACT_RELU) {
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = fmaxf(v[x], 0.f);
            }
            if (row < M && !(e.dbg & 1)) {
                if (e.out_mode == 0) {
                    char* const op = optr + pass * step_o;          // == out + (row * ldc + col) elements
                    if (e.out_dtype == MSAM_F32) {
                        *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        uint2 pk;
                        if constexpr (F16) { pk.x = pack2h(v[0], v[1]); pk.y = pack2h(v[2], v[3]); }
                        else { pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); }
                        *(uint2*)op = pk;
                    }
                } else {
                    u16* dst = which == 0 ? e.q : (which == 1 ? e.k : e.v);
                    uint2 pk;
                    if constexpr (F16) { pk.x = pack2h(v[0], v[1]); pk.y = pack2h(v[2], v[3]); }
                    else { pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); }
                    *(uint2*)(dst + ((long)(tb * e.heads + head) * e.tokens + tt) * e.head_dim + d) = pk;
                }
            }
            if (e.out_mode != 0) {                                  // the next pass: 16 tokens further (a tile may straddle several images)
                tt += 16;
                while (tt >= e.tokens) { tt -= e.tokens; ++tb; }
            }
        }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Two-workgroups-per-CU variant of the large-shape kernel (staging 4; round 3): tile 256 (M) x 128 (N) x 32, 4 waves as
// 2 (M) x 2 (N) - the same 128 x 64 wave tile on the 32x32x16 MFMA and the same transposed product as gemm256_kernel - but
// TWO independent workgroups per CU (one wave of each on every SIMD, 256 registers per wave, 72 KB of LDS each) instead of one
// 8-wave workgroup, so that a workgroup's epilogue (and its k-loop's LDS latencies) run under the OTHER workgroup's MFMAs: with one
// workgroup per CU the epilogue of the K = 768 shapes was 30 - 45 % on top of the k-loop and nothing could hide it
// (profiles/r03_experiments.md section 1).
// * operands: LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass) into a ring of THREE stages of
//   (256 + 128) rows x 64 B, two k-tiles ahead; ONE barrier per k-tile, in front of it a COUNTED s_waitcnt vmcnt(6) (the six DMA
//   loads of the newest tile stay in flight).  The DMA instruction is inline assembly: through the builtin the compiler's wait-count
//   pass puts a vmcnt(0) in front of every ds_read that may alias the DMA's LDS destination, which serialises the ring.
//   Ordering: tile kt is waited for (vmcnt) by every wave at the end of iteration kt - 1, then the barrier, then read in iteration
//   kt; stage (kt + 2) % 3 is re-filled at the top of iteration kt, after the barrier that ended its last reads (iteration kt - 1).
// * LDS rows of 64 B (4 chunks of 16 B), chunk' = chunk ^ ((row >> 2) & 3): the 16-lane service groups of ds_read_b128
//   ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... MI355X_MICROARCH LDS table) hit 16 distinct 16-byte slots when the lanes read
//   rows (l & 31) of a 32-row fragment at one logical chunk; applied on the DMA's per-lane SOURCE address (LDS image linear).
// * epilogue straight from the accumulators: a lane holds 4 consecutive columns of one row (16 B fp32 / 8 B 16-bit stores; the
//   lane pair (l, l + 32) and the four g cover 128 / 64 contiguous bytes of a row) - no LDS transposition, no barrier.
// bf16 / fp16 operands only (fp8 keeps gemm256_kernel<1, true>).
constexpr int GW_BM = 256, GW_BN = 128, GW_BK = 32;
constexpr int GW_STAGE = (GW_BM + GW_BN) * 64;        // bytes per stage
constexpr int GW_LDS = 3 * GW_STAGE;                  // 73 728 B: two workgroups per CU
typedef unsigned int u32x4_t_gw __attribute__((ext_vector_type(4)));
MSAM_DEVINL void gw_dma16(const u32x4_t_gw& rsrc, int voff, int soff, unsigned lds_addr) {
    // m0 = LDS base of this wave-instruction (lane l lands at m0 + 16 l); one wait state between the SALU write of m0 and its use
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
                 : "memory");
}
MSAM_DEVINL u32x4_t_gw gw_rsrc(const void* base, long bytes) {
    const unsigned long long b = (unsigned long long)base;
    const unsigned n = (unsigned)(bytes < 0xffffffffL ? bytes : 0xffffffffL);
    return u32x4_t_gw{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) & 0xffffu,
                      (unsigned)__builtin_amdgcn_readfirstlane((int)n), 0x00020000u};
}

template <bool F16>
__global__ __launch_bounds__(256, 2) void gemm2w_kernel(const u16* __restrict__ A, long lda, const u16* __restrict__ W, long ldw,
                                                        int M, int N, int K, Epi e) {
    extern __shared__ __attribute__((aligned(16))) uint4 dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles_n = N / GW_BN, tiles_m = (M + GW_BM - 1) / GW_BM;
    const int nwg = tiles_m * tiles_n;
    const int wm = wave >> 1, wn = wave & 1;
    const u32x4_t_gw ra = gw_rsrc(A, (long)M * lda * 2), rw = gw_rsrc(W, (long)N * ldw * 2);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)dyn;
    const unsigned lds_a = lds0 + (unsigned)wave * (64 * 64), lds_w = lds0 + GW_BM * 64 + (unsigned)wave * (32 * 64);
    // fragment reads: row = 32-row fragment base + l31, logical chunk 2 s + lh, physical chunk ^ ((l31 >> 2) & 3)
    const int fl = (l31 >> 2) & 3;
    const int c0 = (lh ^ fl) << 4, c1 = ((2 + lh) ^ fl) << 4;
    const char* ldsb = (const char*)dyn;
    const int fa = (wm * 128 + l31) * 64, fw = GW_BM * 64 + (wn * 64 + l31) * 64;
    const int nk = (e.dbg & 4) ? 1 : K / GW_BK;

    // Persistent workgroups (grid = 2 per CU), static schedule: workgroup b takes the virtual ids b, b + grid, ... (id -> tile by the
    // XCD-aware map).  The SECOND workgroup of every CU starts half a tile late, so that from then on its epilogues fall into the
    // other workgroup's k-loops and vice versa (started together, the two reach their epilogues together and nothing overlaps).
    {
        bool late;
        if (e.gw_class == 1) late = (blockIdx.x >> 3) & 1;
        else if (e.gw_class == 2) late = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1;          // HW_ID.WAVE_ID parity
        else late = (int)(blockIdx.x >> 3) >= (int)(gridDim.x >> 4);                             // dispatch order: one per CU first
        if (late)
            for (int d = 0; d < e.gw_delay; ++d) __builtin_amdgcn_s_sleep(100);
    }
    int tr_n = 0;
    auto stamp = [&]() {
        if (e.trace && tid == 0 && tr_n < 62) e.trace[(long)blockIdx.x * 64 + 2 + tr_n++] = __builtin_amdgcn_s_memrealtime();
    };
    if (e.trace && tid == 0) {
        e.trace[(long)blockIdx.x * 64] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID
        e.trace[(long)blockIdx.x * 64 + 1] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
    for (int vid = blockIdx.x; vid < nwg; vid += gridDim.x) {
    stamp();                                                       // tile start
    int bid;
    {
        int xcd = vid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * GW_BM, n0 = tile_n * GW_BN;
    if (vid != (int)blockIdx.x) __builtin_amdgcn_s_barrier();      // the previous tile's last fragment reads are done: the ring is free

    // DMA map: wave w, A instruction q (0..3) covers rows w*64 + q*16 + lane/4, W instruction q (0..1) rows w*32 + q*16 + lane/4;
    // lane l writes physical chunk l & 3 of its row <- global chunk (l & 3) ^ ((row >> 2) & 3)
    int aoff[4], woff[2];
    {
        const int r4 = lane >> 2, pc = lane & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = wave * 64 + q * 16 + r4;
            aoff[q] = (int)((long)min(m0 + row, M - 1) * lda * 2 + ((pc ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = wave * 32 + q * 16 + r4;
            woff[q] = (int)((long)(n0 + row) * ldw * 2 + ((pc ^ ((row >> 2) & 3)) << 4));
        }
    }
    auto issue = [&](int kt, int st) {
        const int so = (e.dbg & 8) ? 0 : kt * 64;
        const int sow = (e.dbg & 24) ? 0 : kt * 64;          // dbg 16: only the weights from k-tile 0
        const unsigned sb = (unsigned)st * GW_STAGE;
        gw_dma16(ra, aoff[0], so, lds_a + sb); gw_dma16(ra, aoff[1], so, lds_a + sb + 1024);
        gw_dma16(rw, woff[0], sow, lds_w + sb);
        gw_dma16(ra, aoff[2], so, lds_a + sb + 2048); gw_dma16(ra, aoff[3], so, lds_a + sb + 3072);
        gw_dma16(rw, woff[1], sow, lds_w + sb + 1024);
    };

    f32x16_t acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;

    issue(0, 0);
    if (nk > 1) { issue(1, 1); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 2 < nk;
        if (more) issue(kt + 2, st >= 1 ? st - 1 : 2);
        const char* sa = ldsb + st * GW_STAGE + fa;
        const char* sw = ldsb + st * GW_STAGE + fw;
        uint4 wf[2][2], af[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[0][i] = *(const uint4*)(sw + i * 2048 + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) af[0][j] = *(const uint4*)(sa + j * 2048 + c0);
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[1][i] = *(const uint4*)(sw + i * 2048 + c1);
#pragma unroll
        for (int j = 0; j < 4; ++j) af[1][j] = *(const uint4*)(sa + j * 2048 + c1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma32<F16>(wf[s2][i], af[s2][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        if (kt + 1 < nk) {
            if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        st = st == 2 ? 0 : st + 1;
    }
    stamp();                                                       // k-loop done
    if (e.dbg & 2) continue;

    // ---- epilogue from the accumulators: acc[i][j][4 g + x] = C[m0 + wm*128 + j*32 + l31][n0 + wn*64 + i*32 + 8 g + 4 lh + x]
    epi_direct<F16>(acc, m0, n0, wm, wn, l31, lh, M, e);
    }                                                      // persistent tile loop
    stamp();
}

// ------------------------------------------------------------------------------------------------------------------
// Row-complete variant for N == 256 with a fused LayerNorm epilogue (decoder image-token stream):
// tile 64 x 256 x 64, 4 waves side by side along N (each 64 x 64), so one workgroup owns whole output rows and the
// epilogue can normalise them in place:   ln_mode 1: LayerNorm over the 256-wide row (norm4 of the two-way block)
//                                          ln_mode 2: LayerNorm over each 64-column group + GELU (LayerNorm2d + GELU of
//                                                     the first up-scaling stage; columns = (sub-pixel, channel))
// This removes the fp32 pre-norm round trip through HBM (write 4 B + read 4 B per element) of the unfused form.
// add / out_a / out_b (msam_gemm_t.ln_add / ln_out_a / ln_out_b; the decoder's token side, round 3): besides the plain output the
// normalised row is also written as 16-bit operand copies for the NEXT products - out_a = round16(v + add[row]) (the queries with
// their positional encoding), out_b = round16(v) - which were separate add_cast launches behind a separate LayerNorm launch.
struct EpiLN { const float* ln_w; const float* ln_b; float eps; int mode; const float* add; u16* out_a; u16* out_b; int dt16; };

template <bool F16>
__global__ __launch_bounds__(256) void gemm_ln_kernel(const u16* __restrict__ A, long lda, const u16* __restrict__ W,
                                                      long ldw, int M, int K, Epi e, EpiLN ln) {
    extern __shared__ __attribute__((aligned(16))) uint4 dyn_lds[];
    constexpr int LM = 64, LN_ = 256;
    uint4* ldsA = dyn_lds;                       // [2][64*8]
    uint4* ldsB = dyn_lds + 2 * LM * 8;          // [2][256*8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * LM;
    const int srow = tid >> 3, scp = tid & 7;

    const u16 *a_s0, *a_s1, *w_s0, *w_s1, *w_s2, *w_s3, *w_s4, *w_s5, *w_s6, *w_s7;
    {
        int r0 = srow, r1 = 32 + srow;
        int ar0 = min(m0 + r0, M - 1), ar1 = min(m0 + r1, M - 1);
        a_s0 = A + (long)ar0 * lda + (scp ^ swz(r0)) * 8;
        a_s1 = A + (long)ar1 * lda + (scp ^ swz(r1)) * 8;
#define WSRC(p_) (W + (long)((p_) * 32 + srow) * ldw + (scp ^ swz((p_) * 32 + srow)) * 8)
        w_s0 = WSRC(0); w_s1 = WSRC(1); w_s2 = WSRC(2); w_s3 = WSRC(3); w_s4 = WSRC(4); w_s5 = WSRC(5); w_s6 = WSRC(6); w_s7 = WSRC(7);
#undef WSRC
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK;
    uint4 ra0, ra1, rw0, rw1, rw2, rw3, rw4, rw5, rw6, rw7;
#define LN_ISSUE(kt_)                                                                               \
    do {                                                                                            \
        const long ko_ = (long)(kt_) * BK;                                                          \
        ra0 = *(const uint4*)(a_s0 + ko_); ra1 = *(const uint4*)(a_s1 + ko_);                       \
        rw0 = *(const uint4*)(w_s0 + ko_); rw1 = *(const uint4*)(w_s1 + ko_);                       \
        rw2 = *(const uint4*)(w_s2 + ko_); rw3 = *(const uint4*)(w_s3 + ko_);                       \
        rw4 = *(const uint4*)(w_s4 + ko_); rw5 = *(const uint4*)(w_s5 + ko_);                       \
        rw6 = *(const uint4*)(w_s6 + ko_); rw7 = *(const uint4*)(w_s7 + ko_);                       \
    } while (0)
#define LN_COMMIT(buf_)                                                                             \
    do {                                                                                            \
        uint4* la_ = ldsA + (buf_) * LM * 8; uint4* lb_ = ldsB + (buf_) * LN_ * 8;                  \
        la_[(0 * 32 + srow) * 8 + scp] = ra0; la_[(1 * 32 + srow) * 8 + scp] = ra1;                 \
        lb_[(0 * 32 + srow) * 8 + scp] = rw0; lb_[(1 * 32 + srow) * 8 + scp] = rw1;                 \
        lb_[(2 * 32 + srow) * 8 + scp] = rw2; lb_[(3 * 32 + srow) * 8 + scp] = rw3;                 \
        lb_[(4 * 32 + srow) * 8 + scp] = rw4; lb_[(5 * 32 + srow) * 8 + scp] = rw5;                 \
        lb_[(6 * 32 + srow) * 8 + scp] = rw6; lb_[(7 * 32 + srow) * 8 + scp] = rw7;                 \
    } while (0)
    LN_ISSUE(0);
    LN_COMMIT(0);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) LN_ISSUE(kt + 1);
        const uint4* la = ldsA + buf * LM * 8; const uint4* lb = ldsB + buf * LN_ * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 16 + fr;
                a[i] = la[row * 8 + ((ks * 4 + fg) ^ swz(row))];
                const int col = wave * 64 + i * 16 + fr;
                b[i] = lb[col * 8 + ((ks * 4 + fg) ^ swz(col))];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = F16 ? mfma16h(a[i], b[j], acc[i][j]) : mfma16(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < nk) LN_COMMIT(buf ^ 1);
        __syncthreads();
    }
#undef LN_ISSUE
#undef LN_COMMIT
    float* ldsC = (float*)dyn_lds;                // [64][256] fp32 = 64 KB (<= 80 KB staging area)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ldsC[(i * 16 + fg * 4 + r) * LN_ + wave * 64 + j * 16 + fr] = acc[i][j][r];
    __syncthreads();

    const int col = lane * 4;                      // one wave = one full 256-wide row per pass
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (e.bias) { float4 b = *(const float4*)(e.bias + col); bias4[0] = b.x; bias4[1] = b.y; bias4[2] = b.z; bias4[3] = b.w; }
    const bool use_table = e.table && col < e.table_cols;
    float lw[4] = {1.f, 1.f, 1.f, 1.f}, lb4[4] = {0.f, 0.f, 0.f, 0.f};
    if (ln.mode) {
        const int lc = ln.mode == 2 ? (col & 63) : col;
        float4 w4 = *(const float4*)(ln.ln_w + lc), b4 = *(const float4*)(ln.ln_b + lc);
        lw[0] = w4.x; lw[1] = w4.y; lw[2] = w4.z; lw[3] = w4.w; lb4[0] = b4.x; lb4[1] = b4.y; lb4[2] = b4.z; lb4[3] = b4.w;
    }
    for (int pass = 0; pass < 16; ++pass) {
        const int lr = pass * 4 + wave;
        const int row = m0 + lr;
        if (row >= M) break;                       // uniform per wave
        float4 c = *(const float4*)(ldsC + lr * LN_ + col);
        float v[4] = {c.x + bias4[0], c.y + bias4[1], c.z + bias4[2], c.w + bias4[3]};
        if (use_table) {
            float4 t = *(const float4*)(e.table + (long)(row % e.table_rows) * e.table_ld + col);
            v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
        }
        if (e.resid_dtype) {
            const int rr = e.resid_rows ? (row % e.resid_rows) : row;
            if (e.resid_dtype == MSAM_F32) {
                float4 t = *(const float4*)((const float*)e.resid + (long)rr * e.ldr + col);
                v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
            } else {
                uint2 t = *(const uint2*)((const u16*)e.resid + (long)rr * e.ldr + col);
                v[0] += load16((u16)(t.x & 0xffff), e.resid_dtype); v[1] += load16((u16)(t.x >> 16), e.resid_dtype);
                v[2] += load16((u16)(t.y & 0xffff), e.resid_dtype); v[3] += load16((u16)(t.y >> 16), e.resid_dtype);
            }
        }
        if (ln.mode) {
            const float s = (v[0] + v[1]) + (v[2] + v[3]);
            const float inv_n = ln.mode == 1 ? (1.0f / 256.0f) : (1.0f / 64.0f);
            const float mean = (ln.mode == 1 ? wave_sum64(s) : wave_sum_xor16(s)) * inv_n;
            const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
            const float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            const float var = (ln.mode == 1 ? wave_sum64(q) : wave_sum_xor16(q)) * inv_n;
            const float rstd = 1.0f / sqrtf(var + ln.eps);
            v[0] = d0 * rstd * lw[0] + lb4[0]; v[1] = d1 * rstd * lw[1] + lb4[1];
            v[2] = d2 * rstd * lw[2] + lb4[2]; v[3] = d3 * rstd * lw[3] + lb4[3];
            if (ln.mode == 2) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
        }
        if (e.act == MSAM_ACT_GELU) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
        else if (e.act == MSAM_ACT_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (e.out_dtype == MSAM_F32) {
            *(float4*)((float*)e.out + (long)row * e.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint2 pk; pk.x = pack16(v[0], v[1], e.out_dtype); pk.y = pack16(v[2], v[3], e.out_dtype);
            *(uint2*)((u16*)e.out + (long)row * e.ldc + col) = pk;
        }
        if (ln.out_b) {
            uint2 pk; pk.x = pack16(v[0], v[1], ln.dt16); pk.y = pack16(v[2], v[3], ln.dt16);
            *(uint2*)(ln.out_b + (long)row * LN_ + col) = pk;
        }
        if (ln.out_a) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ln.add) t = *(const float4*)(ln.add + (long)row * LN_ + col);
            uint2 pk; pk.x = pack16(v[0] + t.x, v[1] + t.y, ln.dt16); pk.y = pack16(v[2] + t.z, v[3] + t.w, ln.dt16);
            *(uint2*)(ln.out_a + (long)row * LN_ + col) = pk;
        }
    }
}

thread_local char g_err[512] = "";

}  // namespace

extern "C" const char* msam_last_error(void) { return g_err; }
extern "C" int msam_abi_version(void) { return 1; }

void msam_set_error(const char* msg) {
    int i = 0;
    for (; msg[i] && i < 510; ++i) g_err[i] = msg[i];
    g_err[i] = 0;
}

int msam_check_launch(const char* what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        char buf[400];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(err));
        msam_set_error(buf);
        return 2;
    }
    return 0;
}

// ---- optional live profiling of the GEMM kernel with HIP events on the launch stream (bench.py roofline leg)
namespace {
struct ProfSlot { hipEvent_t a, b; double flops, bytes; int family; };
constexpr int PROF_MAX = 4096;
ProfSlot g_prof[PROF_MAX];
int g_prof_n = 0, g_prof_on = 0, g_prof_init = 0;
}  // namespace

extern "C" int msam_profile_collect_family(int32_t* launches, double* ms, double* flops, double* bytes);

static int msam_num_cus() {
    static int cus = 0;
    if (cus <= 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}
extern "C" int msam_gemm_set_trace(void* p) { g_gw_trace = (unsigned long long*)p; return 0; }
void msam_gemm_set_gw(int delay, int cls) { if (delay >= -1) g_tune_gw_delay = delay; if (cls >= 0) g_tune_gw_class = cls; }
void msam_gemm_set_g3(int delay) { g_tune_g3_delay = delay; }
void msam_gemm_set_g3_epi(int v) { g_tune_g3_epi = v; }
void msam_gemm_set_dbg(int v) { g_tune_gemm_dbg = v; }     // msam_tune_set "gemm_dbg" (decfold.hip)
extern "C" int msam_gemm256_set_staging(int staging) {
    if (staging < -1 || staging > 4) { msam_set_error("msam_gemm256_set_staging: -1 (default), 0, 1, 2, 3 or 4"); return 1; }
    g_gemm256_staging = staging;
    return 0;
}

extern "C" int msam_profile_enable(int on) {
    if (on && !g_prof_init) {
        for (int i = 0; i < PROF_MAX; ++i) {
            if (hipEventCreate(&g_prof[i].a) != hipSuccess || hipEventCreate(&g_prof[i].b) != hipSuccess) {
                msam_set_error("msam_profile_enable: hipEventCreate failed");
                return 2;
            }
        }
        g_prof_init = 1;
    }
    g_prof_on = on; g_prof_n = 0;
    return 0;
}

// Synchronises the recorded events; returns launches, total milliseconds and total flops (2*M*N*K) since enable.
extern "C" int msam_profile_collect(int32_t* launches, double* total_ms, double* total_flops) {
    int32_t n[MSAM_PROFILE_FAMILIES]; double ms[MSAM_PROFILE_FAMILIES], fl[MSAM_PROFILE_FAMILIES], by[MSAM_PROFILE_FAMILIES];
    if (int e = msam_profile_collect_family(n, ms, fl, by)) return e;
    int32_t nn = 0; double tm = 0, tf = 0;
    for (int f = 0; f < MSAM_PROFILE_FAMILIES; ++f) { nn += n[f]; tm += ms[f]; tf += fl[f]; }
    if (launches) *launches = nn;
    if (total_ms) *total_ms = tm;
    if (total_flops) *total_flops = tf;
    return 0;
}

extern "C" int msam_profile_collect_family(int32_t* launches, double* ms, double* flops, double* bytes) {
    for (int f = 0; f < MSAM_PROFILE_FAMILIES; ++f) { launches[f] = 0; ms[f] = 0; flops[f] = 0; bytes[f] = 0; }
    for (int i = 0; i < g_prof_n; ++i) {
        if (hipEventSynchronize(g_prof[i].b) != hipSuccess) { msam_set_error("msam_profile_collect: sync failed"); return 2; }
        float t = 0.f;
        hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b);
        const int f = (unsigned)g_prof[i].family < MSAM_PROFILE_FAMILIES ? g_prof[i].family : 1;
        ++launches[f]; ms[f] += t; flops[f] += g_prof[i].flops; bytes[f] += g_prof[i].bytes;
    }
    g_prof_n = 0;
    return 0;
}

// used by the other GEMM-family kernels (wsgemm.hip) so that they are part of the same live measurement
void msam_profile_mark2(void* stream, int begin, double flops, double bytes, int family) {
    if (!g_prof_on || g_prof_n >= PROF_MAX) return;
    if (begin) {
        g_prof[g_prof_n].flops = flops; g_prof[g_prof_n].bytes = bytes; g_prof[g_prof_n].family = family;
        (void)hipEventRecord(g_prof[g_prof_n].a, (hipStream_t)stream);
    } else { (void)hipEventRecord(g_prof[g_prof_n].b, (hipStream_t)stream); ++g_prof_n; }
}
void msam_profile_mark(void* stream, int begin, double flops) { msam_profile_mark2(stream, begin, flops, 0.0, 1); }

extern "C" int msam_gemm_bf16(const msam_gemm_t* p, void* stream) {
    if (!p || !p->A || !p->W) { msam_set_error("msam_gemm_bf16: null operand"); return 1; }
    if (p->M <= 0 || p->N <= 0 || p->K <= 0 || p->N % BN != 0 || p->K % BK != 0) {
        msam_set_error("msam_gemm_bf16: need M > 0, N % 128 == 0, K % 64 == 0");
        return 1;
    }
    if ((p->lda % 8) || (p->ldw % 8)) { msam_set_error("msam_gemm_bf16: lda/ldw must be multiples of 8"); return 1; }
    if (p->out_mode == 0 && (p->ldc % 4)) { msam_set_error("msam_gemm_bf16: ldc must be a multiple of 4"); return 1; }
    if (p->out_mode == 3 && (!p->out || p->ldc != 3L * p->N || p->N % 4 || p->out_dtype == MSAM_F32 || p->ln_mode || p->split_k > 1 || p->a_dtype == MSAM_FP8)) {
        msam_set_error("msam_gemm_bf16: out_mode 3 (hi + lo pairs) needs a 16-bit output with ldc == 3 N, no fused LayerNorm / split-K / fp8");
        return 1;
    }
    if (p->out_mode < 0 || p->out_mode > 3) { msam_set_error("msam_gemm_bf16: out_mode is 0 .. 3"); return 1; }
    if ((p->table && ((p->table_cols % 4) || (p->table_ld % 4))) || (p->resid && (p->ldr % 4))) {
        msam_set_error("msam_gemm_bf16: table_cols/table_ld/ldr must be multiples of 4");
        return 1;
    }
    if (p->out_mode == 0 && !p->out) { msam_set_error("msam_gemm_bf16: null output"); return 1; }
    if (p->out_mode == 1 && (!p->q || !p->k || !p->v || p->heads * p->head_dim * 3 != p->N || p->head_dim % 4)) {
        msam_set_error("msam_gemm_bf16: bad qkv-split arguments");
        return 1;
    }
    if (p->out_mode == 2 && (!p->k || !p->v || p->N != 256 || p->tokens % 128 || p->M % p->tokens)) {
        msam_set_error("msam_gemm_bf16: bad kv-split arguments (N == 256, tokens % 128 == 0, M % tokens == 0)");
        return 1;
    }
    Epi e;
    e.bias = p->bias; e.table = p->table; e.table_rows = p->table_rows > 0 ? p->table_rows : 1;
    e.table_cols = p->table_cols; e.table_ld = p->table_ld;
    e.resid = p->resid; e.resid_dtype = p->resid ? p->resid_dtype : 0; e.resid_rows = p->resid_rows; e.ldr = p->ldr;
    e.act = p->act; e.out = p->out; e.out_dtype = p->out_dtype; e.ldc = p->ldc;
    e.out_mode = p->out_mode; e.q = (u16*)p->q; e.k = (u16*)p->k; e.v = (u16*)p->v;
    e.heads = p->heads; e.head_dim = p->head_dim; e.tokens = p->tokens;
    e.row_scale = nullptr; e.col_scale = nullptr;
    e.splitk_len = 0;
    e.dbg = MSAM_EXPERIMENTS ? g_tune_gemm_dbg : 0;
    hipStream_t s = (hipStream_t)stream;
    if (p->ln_mode && p->a_dtype == MSAM_FP8) { msam_set_error("msam_gemm_bf16(fp8): no fused LayerNorm epilogue"); return 1; }
    const bool f16 = p->a_dtype == MSAM_F16;
    if (p->ln_mode) {
        if (p->N != 256 || p->out_mode != 0 || !p->ln_w || !p->ln_b || p->ln_mode < 0 || p->ln_mode > 2) {
            msam_set_error("msam_gemm_bf16: fused LayerNorm needs N == 256, plain output and ln_w / ln_b");
            return 1;
        }
        static bool attr_set = false;
        constexpr int LN_LDS = 2 * (64 * 8 + 256 * 8) * 16;     // 80 KB dynamic LDS
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)gemm_ln_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LN_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)gemm_ln_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LN_LDS) != hipSuccess) {
                msam_set_error("msam_gemm_bf16: cannot raise the dynamic LDS limit");
                return 2;
            }
            attr_set = true;
        }
        if ((p->ln_out_a || p->ln_out_b) && p->ln_mode != 1) { msam_set_error("msam_gemm_bf16: ln_out_a / ln_out_b go with ln_mode 1"); return 1; }
        EpiLN ln{p->ln_w, p->ln_b, p->ln_eps, p->ln_mode, p->ln_add, (u16*)p->ln_out_a, (u16*)p->ln_out_b, f16 ? MSAM_F16 : MSAM_BF16};
        const bool prof_ln = g_prof_on && g_prof_n < PROF_MAX;
        if (prof_ln) {
            g_prof[g_prof_n].flops = 2.0 * p->M * (double)p->N * p->K; g_prof[g_prof_n].bytes = 0; g_prof[g_prof_n].family = 5;
            (void)hipEventRecord(g_prof[g_prof_n].a, s);
        }
        if (f16) hipLaunchKernelGGL(gemm_ln_kernel<true>, dim3((p->M + 63) / 64), dim3(256), LN_LDS, s, (const u16*)p->A, (long)p->lda,
                                    (const u16*)p->W, (long)p->ldw, p->M, p->K, e, ln);
        else hipLaunchKernelGGL(gemm_ln_kernel<false>, dim3((p->M + 63) / 64), dim3(256), LN_LDS, s, (const u16*)p->A, (long)p->lda,
                                (const u16*)p->W, (long)p->ldw, p->M, p->K, e, ln);
        if (prof_ln) { (void)hipEventRecord(g_prof[g_prof_n].b, s); ++g_prof_n; }
        return msam_check_launch("msam_gemm_bf16(ln)");
    }
    const bool prof = g_prof_on && g_prof_n < PROF_MAX;
    if (p->a_dtype == MSAM_FP8) {
        // fp8 e4m3 operands: 256 x 256 tile kernel only (the encoder's projections)
        if (p->N % G2 || p->K % 128 || (p->lda % 16) || (p->ldw % 16) || p->ln_mode || p->out_mode == 2 || p->table ||
            p->use_glds || !p->row_scale || !p->col_scale ||
            (p->resid && (p->resid_dtype != MSAM_F32 || p->resid_rows))) {
            msam_set_error("msam_gemm_bf16(fp8): need N % 256 == 0, K % 128 == 0, lda / ldw % 16 == 0, row_scale and col_scale, plain or qkv-split "
                           "output, fp32 residual without row wrap, no table");
            return 1;
        }
        static bool attr8 = false;
        if (!attr8) {
            if (hipFuncSetAttribute((const void*)gemm256_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess) {
                msam_set_error("msam_gemm_bf16: cannot raise the dynamic LDS limit");
                return 2;
            }
            attr8 = true;
        }
        e.row_scale = p->row_scale; e.col_scale = p->col_scale;
        if (prof) {
            g_prof[g_prof_n].flops = 2.0 * p->M * (double)p->N * p->K; g_prof[g_prof_n].bytes = 0; g_prof[g_prof_n].family = 0;
            (void)hipEventRecord(g_prof[g_prof_n].a, s);
        }
        const int tiles8 = ((p->M + G2 - 1) / G2) * (p->N / G2);
        hipLaunchKernelGGL((gemm256_kernel<1, true>), dim3(tiles8), dim3(512), G2_LDS, s, (const u16*)p->A, (long)p->lda,
                           (const u16*)p->W, (long)p->ldw, p->M, p->N, p->K, e);
        if (prof) { (void)hipEventRecord(g_prof[g_prof_n].b, s); ++g_prof_n; }
        return msam_check_launch("msam_gemm_bf16(fp8)");
    }
    int tiles = ((p->M + BM - 1) / BM) * (p->N / BN);
    // large shapes (the encoder's projections): 256 x 256 tile kernel.  MSAM_GEMM256=0 keeps the 128 x 128 kernel (A/B runs)
    static int use256 = -1, staging256 = 0;
    if (use256 < 0) {
        const char* v = getenv("MSAM_GEMM256"); use256 = v ? atoi(v) : 1;
        const char* st = getenv("MSAM_GEMM256_STAGING"); staging256 = st ? atoi(st) : G2_DEFAULT_STAGING;
    }
    if (g_gemm256_staging >= 0) staging256 = g_gemm256_staging;
    if (!MSAM_EXPERIMENTS && (staging256 < 0 || staging256 > 3)) staging256 = G2_DEFAULT_STAGING;
    // staging 4: the two-workgroups-per-CU kernel (256 x 128 tiles, LDS-DMA ring, epilogue from the accumulators)
    // (its operand offsets are 32-bit: operands of 2 GiB and more stay on the 256 x 256 kernel)
    if (MSAM_EXPERIMENTS && use256 && staging256 == 4 && (double)p->M * p->lda * 2 < 2147483648.0 && (double)p->N * p->ldw * 2 < 2147483648.0 && p->split_k <= 1 && (!f16 || p->out_dtype != MSAM_BF16) && !p->use_glds &&
        ((p->M + G2 - 1) / G2) * (p->N / G2) >= 256 && p->N % GW_BN == 0 && p->K % GW_BK == 0 && p->lda % 8 == 0 && p->ldw % 8 == 0 &&
        p->out_mode != 2 && p->out_mode != 3 && !p->table && (!p->resid || (p->resid_dtype == MSAM_F32 && !p->resid_rows))) {
        static bool attr2w = false;
        if (!attr2w) {
            if (hipFuncSetAttribute((const void*)gemm2w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)gemm2w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, GW_LDS) != hipSuccess) {
                msam_set_error("msam_gemm_bf16: cannot raise the dynamic LDS limit");
                return 2;
            }
            attr2w = true;
            if (getenv("MSAM_GEMM_OCC")) {
                int nb = -1;
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gemm2w_kernel<false>, 256, GW_LDS);
                fprintf(stderr, "gemm2w_kernel: %d workgroups per CU (dynamic LDS %d)\n", nb, GW_LDS);
            }
        }
        if (prof) {
            g_prof[g_prof_n].flops = 2.0 * p->M * (double)p->N * p->K; g_prof[g_prof_n].bytes = 0; g_prof[g_prof_n].family = 0;
            (void)hipEventRecord(g_prof[g_prof_n].a, s);
        }
        const int tiles2w = ((p->M + GW_BM - 1) / GW_BM) * (p->N / GW_BN);
        const int gmul = (g_tune_gemm_dbg >> 8) & 15 ? (g_tune_gemm_dbg >> 8) & 15 : 2;         // gemm_dbg bits 8-11: workgroups per CU of the persistent grid
        const int grid2w = tiles2w < gmul * msam_num_cus() ? tiles2w : gmul * msam_num_cus();
        e.gw_class = g_tune_gw_class;
        e.trace = g_gw_trace;
        e.gw_delay = g_tune_gw_delay >= 0 ? g_tune_gw_delay : (p->K / GW_BK) / 4;
        if (f16)
            hipLaunchKernelGGL(gemm2w_kernel<true>, dim3(grid2w), dim3(256), GW_LDS, s, (const u16*)p->A, (long)p->lda, (const u16*)p->W,
                               (long)p->ldw, p->M, p->N, p->K, e);
        else
            hipLaunchKernelGGL(gemm2w_kernel<false>, dim3(grid2w), dim3(256), GW_LDS, s, (const u16*)p->A, (long)p->lda, (const u16*)p->W,
                               (long)p->ldw, p->M, p->N, p->K, e);
        if (prof) { (void)hipEventRecord(g_prof[g_prof_n].b, s); ++g_prof_n; }
        return msam_check_launch("msam_gemm_bf16(2w)");
    }
    // (measured: 3 - 14 % faster than the 128 x 128 kernel from one workgroup per CU upwards, slower below)
    // (fp16 operands: 16-bit outputs of this kernel are then fp16 as well - the encoder's fp16 mode; a bf16 output is not offered)
    if (use256 && p->split_k <= 1 && (!f16 || p->out_dtype != MSAM_BF16) && !p->use_glds && ((p->M + G2 - 1) / G2) * (p->N / G2) >= 256 && p->N % G2 == 0 &&
        p->out_mode != 2 && p->out_mode != 3 && !p->table && (!p->resid || (p->resid_dtype == MSAM_F32 && !p->resid_rows))) {
        static bool attr256 = false;
        if (!attr256) {
            if (hipFuncSetAttribute((const void*)gemm256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)gemm256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)gemm256_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)gemm256_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)gemm256_kernel<3, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS) != hipSuccess) {
                msam_set_error("msam_gemm_bf16: cannot raise the dynamic LDS limit");
                return 2;
            }
            attr256 = true;
        }
        if (prof) {
            g_prof[g_prof_n].flops = 2.0 * p->M * (double)p->N * p->K; g_prof[g_prof_n].bytes = 0; g_prof[g_prof_n].family = 0;
            (void)hipEventRecord(g_prof[g_prof_n].a, s);
        }
        const int tiles256 = ((p->M + G2 - 1) / G2) * (p->N / G2);
        e.gw_class = msam_num_cus();                                  // workgroups of the first dispatch wave
        e.gw_delay = g_tune_g3_delay >= 0 ? g_tune_g3_delay : 0;
        // (epi_direct packs 16-bit outputs by the operand type: offered only where that IS the output type)
        e.direct_epi = (MSAM_EXPERIMENTS && (p->out_dtype == MSAM_F32 || (p->out_dtype == MSAM_F16) == f16)) ? g_tune_g3_epi : 0;
#define G2_GO(ST_) hipLaunchKernelGGL(gemm256_kernel<ST_>, dim3(tiles256), dim3(512), G2_LDS, s, (const u16*)p->A, (long)p->lda, \
                                     (const u16*)p->W, (long)p->ldw, p->M, p->N, p->K, e)
        if (f16)
            hipLaunchKernelGGL((gemm256_kernel<3, false, true>), dim3(tiles256), dim3(512), G2_LDS, s, (const u16*)p->A, (long)p->lda,
                               (const u16*)p->W, (long)p->ldw, p->M, p->N, p->K, e);
        else if (staging256 == 1) G2_GO(1); else if (staging256 == 2) G2_GO(2); else if (staging256 == 3) G2_GO(3); else G2_GO(0);
#undef G2_GO
        if (prof) { (void)hipEventRecord(g_prof[g_prof_n].b, s); ++g_prof_n; }
        return msam_check_launch("msam_gemm_bf16(256)");
    }
    if (p->split_k > 1) {
        // split-K (training: dW = dY^T X contracts over hundreds of thousands of rows into a 128 x 256 tile or two - a handful of
        // workgroups would walk the whole contraction one after the other): split_k slices of K, one workgroup per (tile, slice), fp32
        // atomic accumulation into the zeroed output
        if (p->out_mode != 0 || p->out_dtype != MSAM_F32 || p->bias || p->table || p->resid || p->act || p->ldc != p->N ||
            p->K % (p->split_k * BK)) {
            msam_set_error("msam_gemm_bf16(split_k): plain fp32 output with ldc == N, no bias / table / residual / activation, K % (split_k * 64) == 0");
            return 1;
        }
        float* parts = msam_det_workspace((size_t)p->split_k * p->M * p->N, 2);
        if (!parts) { msam_set_error("msam_gemm_bf16(split_k): cannot allocate the partial-tile workspace"); return 2; }
        e.out = parts;
        e.splitk_len = p->K / p->split_k;
        if (f16) hipLaunchKernelGGL((gemm_kernel<false, true>), dim3(tiles, p->split_k), dim3(256), 0, s, (const u16*)p->A, (long)p->lda,
                                    (const u16*)p->W, (long)p->ldw, p->M, p->N, p->K, e);
        else hipLaunchKernelGGL(gemm_kernel<false>, dim3(tiles, p->split_k), dim3(256), 0, s, (const u16*)p->A, (long)p->lda,
                                (const u16*)p->W, (long)p->ldw, p->M, p->N, p->K, e);
        msam_det_reduce(parts, p->split_k, (long)p->M * p->N, (float*)p->out, 0, stream);       // out = slice 0 + slice 1 + ... in that order
        return msam_check_launch("msam_gemm_bf16(split_k)");
    }
    if (prof) {
        g_prof[g_prof_n].flops = 2.0 * p->M * (double)p->N * p->K; g_prof[g_prof_n].bytes = 0; g_prof[g_prof_n].family = 5;
        (void)hipEventRecord(g_prof[g_prof_n].a, s);
    }
    if (f16)
        hipLaunchKernelGGL((gemm_kernel<false, true>), dim3(tiles), dim3(256), 0, s, (const u16*)p->A, (long)p->lda,
                           (const u16*)p->W, (long)p->ldw, p->M, p->N, p->K, e);
    else if (p->use_glds)
        hipLaunchKernelGGL(gemm_kernel<true>, dim3(tiles), dim3(256), 0, s, (const u16*)p->A, (long)p->lda,
                           (const u16*)p->W, (long)p->ldw, p->M, p->N, p->K, e);
    else
        hipLaunchKernelGGL(gemm_kernel<false>, dim3(tiles), dim3(256), 0, s, (const u16*)p->A, (long)p->lda,
                           (const u16*)p->W, (long)p->ldw, p->M, p->N, p->K, e);
    if (prof) { (void)hipEventRecord(g_prof[g_prof_n].b, s); ++g_prof_n; }
    return msam_check_launch("msam_gemm_bf16");
}

// Grouped launch of independent 128 x 128-tile products (plain bf16 path only: no fused LayerNorm, no fp8, no LDS-DMA staging)
extern "C" int msam_gemm_group_bf16(const msam_gemm_t* items, int32_t n, void* stream) {
    if (!items || n <= 0 || n > MSAM_GEMM_GROUP_MAX) { msam_set_error("msam_gemm_group_bf16: 1 .. MSAM_GEMM_GROUP_MAX products"); return 1; }
    GroupArgs g{};
    int max_tiles = 0;
    double flops = 0.0;
    for (int i = 0; i < n; ++i) {
        const msam_gemm_t* p = items + i;
        if (!p->A || !p->W || p->M <= 0 || p->N <= 0 || p->K <= 0 || p->N % BN || p->K % BK || (p->lda % 8) || (p->ldw % 8) ||
            p->ln_mode || p->a_dtype == MSAM_FP8 || p->use_glds || (p->a_dtype == MSAM_F16) != (items[0].a_dtype == MSAM_F16)) {
            msam_set_error("msam_gemm_group_bf16: every product needs N % 128 == 0, K % 64 == 0, lda / ldw % 8 == 0, bf16 operands "
                           "(or fp16 for all of them), no fused LayerNorm");
            return 1;
        }
        if ((p->out_mode == 0 && (!p->out || (p->ldc % 4))) || p->out_mode == 1 ||
            (p->out_mode == 2 && (!p->k || !p->v || p->N != 256 || p->tokens % 128 || p->M % p->tokens)) || p->out_mode > 3 || p->out_mode < 0 ||
            (p->out_mode == 3 && (!p->out || p->ldc != 3L * p->N || p->N % 4 || p->out_dtype == MSAM_F32)) ||
            (p->table && ((p->table_cols % 4) || (p->table_ld % 4))) || (p->resid && (p->ldr % 4))) {
            msam_set_error("msam_gemm_group_bf16: bad output / table / residual arguments (qkv-split output is not grouped; out_mode 3 needs a "
                           "16-bit output with ldc == 3 N)");
            return 1;
        }
        GroupItem& it = g.it[i];
        it.A = (const u16*)p->A; it.lda = p->lda; it.W = (const u16*)p->W; it.ldw = p->ldw; it.M = p->M; it.N = p->N; it.K = p->K;
        Epi& e = it.e;
        e.bias = p->bias; e.table = p->table; e.table_rows = p->table_rows > 0 ? p->table_rows : 1;
        e.table_cols = p->table_cols; e.table_ld = p->table_ld;
        e.resid = p->resid; e.resid_dtype = p->resid ? p->resid_dtype : 0; e.resid_rows = p->resid_rows; e.ldr = p->ldr;
        e.act = p->act; e.out = p->out; e.out_dtype = p->out_dtype; e.ldc = p->ldc;
        e.out_mode = p->out_mode; e.q = (u16*)p->q; e.k = (u16*)p->k; e.v = (u16*)p->v;
        e.heads = p->heads; e.head_dim = p->head_dim; e.tokens = p->tokens;
        e.row_scale = nullptr; e.col_scale = nullptr;
        e.splitk_len = 0; e.dbg = 0;
        const int tiles = ((p->M + BM - 1) / BM) * (p->N / BN);
        if (tiles > max_tiles) max_tiles = tiles;
        flops += 2.0 * p->M * (double)p->N * p->K;
    }
    for (int i = n; i < MSAM_GEMM_GROUP_MAX; ++i) g.it[i] = g.it[0];
    hipStream_t s = (hipStream_t)stream;
    const bool prof = g_prof_on && g_prof_n < PROF_MAX;
    if (prof) {
        g_prof[g_prof_n].flops = flops; g_prof[g_prof_n].bytes = 0; g_prof[g_prof_n].family = 5;
        (void)hipEventRecord(g_prof[g_prof_n].a, s);
    }
    if (items[0].a_dtype == MSAM_F16) hipLaunchKernelGGL(gemm_group_kernel<true>, dim3(max_tiles, n), dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm_group_kernel<false>, dim3(max_tiles, n), dim3(256), 0, s, g);
    if (prof) { (void)hipEventRecord(g_prof[g_prof_n].b, s); ++g_prof_n; }
    return msam_check_launch("msam_gemm_group_bf16");
}
