// Weights-stationary streaming GEMM for the decoder's image-token stream:  C[M,N] = epi(A[M,K] * W[N,K]^T)
// with M = P*4096 (millions of rows), N in {128,256}, K in {128,256}: the whole weight matrix (32-128 KB bf16) is
// tiny, the activations are the traffic.  So the weights never touch LDS: each of the 8 waves of a persistent
// workgroup keeps ITS N/8 output columns of W as MFMA B fragments in registers for the whole launch (<= 64 VGPRs),
// and only the activation row blocks (64 rows x K) stream HBM -> VGPR -> LDS (double buffered, 16-lane-group
// conflict-free XOR swizzle) -> A fragments.  Per 64-row tile and wave: K/32 x 4 ds_read_b128 feed K/32 x 4 x N/128
// MFMAs; nothing but the A tile is ever written to LDS in the main loop.
// Epilogue: the fp32 tile goes through LDS once so that every wave then owns whole rows: bias, positional table,
// residual, LayerNorm(256) / LayerNorm(64 groups)+GELU, coalesced bf16 stores; or the K | V^T split store.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);
void msam_profile_mark2(void* stream, int begin, double flops, double bytes, int family);

namespace {

constexpr int WM = 32;          // rows per tile (small tiles + 2-3 workgroups per CU: the kernel is HBM / latency bound)

struct WsEpi {
    const float* bias; const float* table; int table_rows, table_cols; long table_ld;
    const u16* resid; int resid_rows; long ldr;
    const float* ln_w; const float* ln_b; float ln_eps; int ln_mode;
    u16* out; long ldc;              // plain bf16 output [M, N]
    int kv_split; u16* k_out; u16* vT_out; int tokens;
    int head_major;                  // plain output as [M/tokens][N/16][tokens][16] (per-head contiguous K for t2i attention)
};

// 4-bit chunk swizzle: the 16 rows of a ds_read_b128 service group hit 16 distinct 16-byte slots
MSAM_DEVINL int swz16(int row) { return (row & 15) ^ (((row + 4) >> 3) & 1); }

// 8 waves; N = 128 variants keep <= 128 VGPRs so that two workgroups share a CU and overlap their MFMA / epilogue
// phases (the decoder therefore issues its N = 256, K = 256 products as two N = 128 launches)
// EPI: the epilogue reads per-row operands from global memory (positional table and / or residual rows); compiled out
// otherwise so that the steady state contains no load but the tile prefetch (common.h wait_vmem_all())
template <int N, int K, bool EPI>
__global__ __launch_bounds__(512, (N == 256 && K == 256) ? 2 : 4) void wsgemm_kernel(const u16* __restrict__ A, const u16* __restrict__ W, int M, WsEpi e) {
    constexpr int NWAVES = 8;
    constexpr int NTHR = NWAVES * 64;
    constexpr int NT = N / NWAVES / 16;          // n-tiles per wave (1 or 2)
    constexpr int KC = K / 32;                   // 32-deep k chunks
    constexpr int CPR = K / 8;                   // 16-byte chunks per A row
    constexpr int APT = WM * CPR / NTHR;         // A chunks staged per thread (1 or 2)
    constexpr int MT = WM / 16;                  // m-tiles per wave
    extern __shared__ __attribute__((aligned(16))) uint4 dyn_lds[];
    uint4* ldsA = dyn_lds;                                        // [2][WM * CPR]
    float* ldsC = (float*)(dyn_lds + 2 * WM * CPR);               // [WM][N] fp32
    float* prm = ldsC + WM * N;                                   // [3][N]: bias, ln_w, ln_b per output column

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    // ---- stationary weights: B fragments of this wave's columns
    uint4 bfr[NT][KC];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
            bfr[ni][kc] = *(const uint4*)(W + (long)(wave * (NT * 16) + ni * 16 + fr) * K + kc * 32 + fg * 8);
    // epilogue parameters live in LDS: a global load inside the loop would have to be waited for with vmcnt(0), which
    // also waits for the tile prefetch issued before it (see common.h touch())
    for (int c = threadIdx.x; c < N; c += NTHR) {
        const int lc = e.ln_mode == 2 ? (c & 63) : c;
        prm[c] = e.bias ? e.bias[c] : 0.f;
        prm[N + c] = e.ln_mode ? e.ln_w[lc] : 1.f;
        prm[2 * N + c] = e.ln_mode ? e.ln_b[lc] : 0.f;
    }
    wait_vmem_all();

    const int ntiles = M / WM;
    // staging map: chunk id q = p*512 + tid -> (row = q / CPR, c = q % CPR)
    uint4 ra0, ra1, rb0, rb1;       // two register sets: tiles are requested TWO iterations ahead (HBM latency under load
    (void)ra1; (void)rb1;           // is several microseconds; one workgroup per CU has nothing else to hide it with)
    const int voff = tid * 16;      // a tile is WM * K * 2 contiguous bytes: chunk p*512 + tid at byte (p*512 + tid) * 16
#define WS_DST(p_, buf_) ldsA[(buf_) * WM * CPR + (((p_) * NTHR + tid) / CPR) * CPR + \
                              ((((p_) * NTHR + tid) % CPR) ^ swz16(((p_) * NTHR + tid) / CPR))]
#define WS_LOAD(r0_, r1_, tile_)                                                          \
    do {                                                                                  \
        const rsrc_t ra_ = make_rsrc(A + (long)(tile_) * WM * K, WM * K * 2);             \
        r0_ = buf_load16(ra_, voff, 0);                                                   \
        if constexpr (APT == 2) { r1_ = buf_load16(ra_, voff, NTHR * 16); }               \
    } while (0)
#define WS_STORE(r0_, r1_, buf_)                                                          \
    do {                                                                                  \
        WS_DST(0, buf_) = r0_;                                                            \
        if constexpr (APT == 2) { WS_DST(1, buf_) = r1_; }                                \
    } while (0)

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    WS_LOAD(ra0, ra1, tile);
    WS_STORE(ra0, ra1, 0);
    WS_LOAD(ra0, ra1, min(tile + (int)gridDim.x, ntiles - 1));     // pending set of the first iteration (clamped: loads
                                                                    // of the steady state are unconditional)
    __syncthreads();
    int buf = 0;
    // one iteration: `pend` holds tile+stride (requested one iteration ago, stored to LDS at the end of this one),
    // `fresh` receives tile+2*stride now
    auto iteration = [&](uint4& pend0, uint4& pend1, uint4& fresh0, uint4& fresh1) {
        const int next = tile + gridDim.x, next2 = tile + 2 * gridDim.x;
        // epilogue operands of THIS tile (residual rows) are requested before the prefetch of tile + 2 strides: vmcnt is
        // in order, so waiting for them later leaves the (younger) prefetch in flight
        const long row0 = (long)tile * WM;
        constexpr int LPR = N / 4;                 // lanes per row (64 for N = 256, 32 for N = 128)
        constexpr int RPP = 64 / LPR;              // rows per wave pass (1 or 2)
        constexpr int NPASS = WM / (NWAVES * RPP); // 4 (N = 256) or 2 (N = 128)
        const int col = (lane % LPR) * 4;
        const bool use_table = EPI && e.table && col < e.table_cols;
        uint2 res[NPASS];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const long row = row0 + (pass * NWAVES + wave) * RPP + lane / LPR;
            res[pass] = (EPI && e.resid) ? *(const uint2*)(e.resid + (e.resid_rows ? (row % e.resid_rows) : row) * e.ldr + col)
                                         : make_uint2(0u, 0u);
        }
        WS_LOAD(fresh0, fresh1, min(next2, ntiles - 1));
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const uint4* la = ldsA + buf * WM * CPR;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            uint4 a[MT];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const int row = mi * 16 + fr;
                a[mi] = la[row * CPR + ((kc * 4 + fg) ^ swz16(row))];
            }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma16d(a[mi], bfr[ni][kc], acc[mi][ni]);
        }
        // ---- fp32 tile -> LDS (row-complete epilogue); ldsC is private to the epilogue, ldsA[buf^1] gets the next tile
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ldsC[(mi * 16 + fg * 4 + r) * N + wave * (NT * 16) + ni * 16 + fr] = acc[mi][ni][r];
        if (next < ntiles) WS_STORE(pend0, pend1, buf ^ 1);
        __syncthreads();

        float bias4[4];
        { const float4 b = *(const float4*)(prm + col); bias4[0] = b.x; bias4[1] = b.y; bias4[2] = b.z; bias4[3] = b.w; }
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int lr = (pass * NWAVES + wave) * RPP + lane / LPR;
            const long row = row0 + lr;
            const float4 c = *(const float4*)(ldsC + lr * N + col);
            float v[4] = {c.x + bias4[0], c.y + bias4[1], c.z + bias4[2], c.w + bias4[3]};
            if (use_table) {       // 2 MB table, L2 resident
                const float4 t = *(const float4*)(e.table + (row % e.table_rows) * e.table_ld + col);
                v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
            }
            v[0] += d2f((u16)(res[pass].x & 0xffff)); v[1] += d2f((u16)(res[pass].x >> 16));
            v[2] += d2f((u16)(res[pass].y & 0xffff)); v[3] += d2f((u16)(res[pass].y >> 16));
            if ((N == 256 && e.ln_mode) || (N == 128 && e.ln_mode == 2)) {
                const float s = (v[0] + v[1]) + (v[2] + v[3]);
                const float inv_n = e.ln_mode == 1 ? (1.0f / 256.0f) : (1.0f / 64.0f);
                const float mean = (e.ln_mode == 1 ? wave_sum64(s) : wave_sum_xor16(s)) * inv_n;
                const float d0 = v[0] - mean, d1 = v[1] - mean, d2 = v[2] - mean, d3 = v[3] - mean;
                const float q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                const float var = (e.ln_mode == 1 ? wave_sum64(q) : wave_sum_xor16(q)) * inv_n;
                const float rstd = 1.0f / sqrtf(var + e.ln_eps);
                const float4 w4 = *(const float4*)(prm + N + col), b4 = *(const float4*)(prm + 2 * N + col);
                v[0] = d0 * rstd * w4.x + b4.x; v[1] = d1 * rstd * w4.y + b4.y;
                v[2] = d2 * rstd * w4.z + b4.z; v[3] = d3 * rstd * w4.w + b4.w;
                if (e.ln_mode == 2) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
            }
            if (N == 128 && e.kv_split) {
                *(float4*)(ldsC + lr * N + col) = make_float4(v[0], v[1], v[2], v[3]);     // all columns go out transposed
            } else if (!e.kv_split) {
                uint2 pk; pk.x = pack2d(v[0], v[1]); pk.y = pack2d(v[2], v[3]);
                if (e.head_major) {
                    const long b = row / e.tokens, t = row - b * e.tokens;
                    *(uint2*)(e.out + ((b * (N / 16) + (col >> 4)) * e.tokens + t) * 16 + (col & 15)) = pk;
                } else {
                    *(uint2*)(e.out + row * e.ldc + col) = pk;
                }
            } else if (col < 128) {
                uint2 pk; pk.x = pack2d(v[0], v[1]); pk.y = pack2d(v[2], v[3]);
                *(uint2*)(e.k_out + row * 128 + col) = pk;
            } else {
                *(float4*)(ldsC + lr * N + col) = make_float4(v[0], v[1], v[2], v[3]);     // finished v values back
            }
        }
        if (e.kv_split) {
            __syncthreads();
            // V^T[b][d][t]: thread -> (d = tid & 127, 8-row group g = tid >> 7 < 4): 8 tokens = 16 contiguous bytes
            const int d = tid & 127, g = tid >> 7;
            if (g < WM / 8) {
                const long b = row0 / e.tokens, t0 = row0 - b * e.tokens + g * 8;
                const float* src = ldsC + (g * 8) * N + (N == 256 ? 128 : 0) + d;
                uint4 p0;
                p0.x = pack2d(src[0 * N], src[1 * N]); p0.y = pack2d(src[2 * N], src[3 * N]);
                p0.z = pack2d(src[4 * N], src[5 * N]); p0.w = pack2d(src[6 * N], src[7 * N]);
                *(uint4*)(e.vT_out + (b * 128 + d) * e.tokens + t0) = p0;
            }
        }
        __syncthreads();        // ldsC free again; next tile's A (stored above) visible
        buf ^= 1;
    };
    while (tile < ntiles) {
        iteration(ra0, ra1, rb0, rb1);
        tile += gridDim.x;
        if (tile >= ntiles) break;
        iteration(rb0, rb1, ra0, ra1);
        tile += gridDim.x;
    }
}

template <int N, int K, bool EPI>
int launch_epi(const u16* A, const u16* W, int M, const WsEpi& e, hipStream_t s) {
    constexpr int LDS_BYTES = 2 * WM * (K / 8) * 16 + WM * N * 4 + 3 * N * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wsgemm_kernel<N, K, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) !=
            hipSuccess) { msam_set_error("msam_wsgemm_bf16: cannot raise the dynamic LDS limit"); return 2; }
        attr_set = true;
    }
    const int ntiles = M / WM;
    constexpr int NTHR = 512;
    const int per_cu = 2;                                   // resident workgroups per CU (LDS / VGPR budget)
    const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
    // algorithmic HBM bytes: read A (bf16) [+ residual], write the bf16 output
    const double bytes = (double)M * (K * 2.0 + N * 2.0 + (e.resid && !e.resid_rows ? N * 2.0 : 0.0));
    msam_profile_mark2(s, 1, 2.0 * M * (double)N * K, bytes, 1);
    hipLaunchKernelGGL((wsgemm_kernel<N, K, EPI>), dim3(grid), dim3(NTHR), LDS_BYTES, s, A, W, M, e);
    msam_profile_mark2(s, 0, 0.0, 0.0, 1);
    return msam_check_launch("msam_wsgemm_bf16");
}

template <int N, int K>
int launch(const u16* A, const u16* W, int M, const WsEpi& e, hipStream_t s) {
    return (e.table || e.resid) ? launch_epi<N, K, true>(A, W, M, e, s) : launch_epi<N, K, false>(A, W, M, e, s);
}

}  // namespace

extern "C" int msam_wsgemm_bf16(const msam_wsgemm_t* p, void* stream) {
    if (!p || !p->A || !p->W) { msam_set_error("msam_wsgemm_bf16: null operand"); return 1; }
    if (p->M <= 0 || p->M % 64) { msam_set_error("msam_wsgemm_bf16: M must be a positive multiple of 64"); return 1; }
    if (p->kv_split && (!p->vT_out || (p->N == 256 && !p->k_out) || p->tokens % 64 || p->M % p->tokens || p->ln_mode)) {
        msam_set_error("msam_wsgemm_bf16: bad kv-split / transposed-output arguments");
        return 1;
    }
    if (!p->kv_split && !p->out) { msam_set_error("msam_wsgemm_bf16: null output"); return 1; }
    if (p->head_major && (p->kv_split || p->tokens <= 0 || p->M % p->tokens)) { msam_set_error("msam_wsgemm_bf16: bad head_major arguments"); return 1; }
    if (p->ln_mode && (!p->ln_w || !p->ln_b || (p->ln_mode == 1 && p->N != 256) || p->ln_mode < 0 || p->ln_mode > 2)) {
        msam_set_error("msam_wsgemm_bf16: LayerNorm(256) needs N == 256; LayerNorm(64 groups) needs ln_w / ln_b");
        return 1;
    }
    WsEpi e;
    e.bias = p->bias; e.table = p->table; e.table_rows = p->table_rows > 0 ? p->table_rows : 1; e.table_cols = p->table_cols;
    e.table_ld = p->table_ld; e.resid = (const u16*)p->resid; e.resid_rows = p->resid_rows; e.ldr = p->ldr;
    e.ln_w = p->ln_w; e.ln_b = p->ln_b; e.ln_eps = p->ln_eps; e.ln_mode = p->ln_mode;
    e.out = (u16*)p->out; e.ldc = p->ldc; e.kv_split = p->kv_split; e.k_out = (u16*)p->k_out; e.vT_out = (u16*)p->vT_out;
    e.tokens = p->tokens > 0 ? p->tokens : 64;
    e.head_major = p->head_major;
    hipStream_t s = (hipStream_t)stream;
    const u16* A = (const u16*)p->A; const u16* W = (const u16*)p->W;
    if (p->N == 256 && p->K == 256) return launch<256, 256>(A, W, p->M, e, s);
    if (p->N == 256 && p->K == 128) return launch<256, 128>(A, W, p->M, e, s);
    if (p->N == 128 && p->K == 256) return launch<128, 256>(A, W, p->M, e, s);
    if (p->N == 128 && p->K == 128) return launch<128, 128>(A, W, p->M, e, s);
    msam_set_error("msam_wsgemm_bf16: N and K must be 128 or 256");
    return 1;
}
