// Fused image-side half of a two-way-transformer block for the per-prompt token stream:
//
//     q    = (keys + pos) Wq^T + bq                  (layer 1; layer 0 uses the prompt-independent precomputed q)
//     attn = softmax_j(q . k_tok[j] / 4) v_tok       (image token attends over the <= 16 prompt tokens, 8 heads x 16)
//     keys = LayerNorm(keys + attn Wo^T + bo)        (norm4)
//
// One launch replaces q-projection GEMM + image->token attention kernel + out-projection/LayerNorm GEMM and their
// HBM round trips: the stream is read once (16 KB per 32-token tile) and written once.
//
// Wave-specialised software pipeline, one 16-wave workgroup per CU (a CU's register file cannot hold both weight
// matrices twice, so two independent 8-wave workgroups do not fit; two cooperating groups do):
//   group A (waves 0-7, wave = head): keeps its 16 rows of Wq as MFMA A fragments (32 VGPRs); per tile computes
//       q^T = Wq . keys^T straight into the C layout (row = d, col = token), which after bf16 packing IS the B operand
//       of S^T = k_tok . q^T (k-slot map (lane group g, i<4) <-> d = 4g+i, k_tok fragments loaded with the same map), then
//       softmax over the <= 16 prompt tokens in registers and O^T = V_tok^T . P^T; writes the attention tile to LDS.
//   group B (waves 8-15): keeps its 32 columns of Wo as B fragments (32 VGPRs); per tile out-projection from the
//       attention tile of the PREVIOUS iteration, fp32 tile through LDS, then row-complete epilogue (bias + residual
//       from the LDS keys tile + LayerNorm) and coalesced bf16 stores.
// While A works on tile i, B finishes tile i-1 and the loads of tile i+1 are in flight: three keys buffers, two
// attention buffers, two block barriers per tile.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);
void msam_profile_mark2(void* stream, int begin, double flops, double bytes, int family);

namespace {

constexpr int T = 4096, C = 256, CI = 128, WM = 32, NTHR = 1024, CPR = C / 8;
constexpr float NEG_BIG = -1.0e30f;

struct LayerArgs {
    const u16* xin;          // stream [rows, 256] (layer 1) or src [4096, 256] (layer 0, row % 4096)
    const u16* q_shared;     // layer 0: bf16 [4096, 128]
    const u16* wq; const float* bq; const float* peq;     // layer 1
    const u16* wo; const float* bo; const float* ln_w; const float* ln_b; float eps;
    const u16* ktok; const u16* vtok; int Nt;
    u16* out; int rows;
};

MSAM_DEVINL int swzr(int row) { return row & 15; }

template <bool L0>
__global__ __launch_bounds__(NTHR, 4) void dec_image_layer_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 dyn_lds[];
    uint4* ldsK = dyn_lds;                                        // [3][WM*CPR] keys tiles (layer 1)   48 KB
    uint4* ldsP = dyn_lds + (L0 ? 0 : 3 * WM * CPR);              // [2][WM*16]  bf16 attention tiles   16 KB
    float* ldsC = (float*)(ldsP + 2 * WM * 16);                   // [WM][256]   fp32                   32 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const bool grpA = wave < 8;
    const int ntiles = a.rows / WM;
    const int stride = gridDim.x;
    const int first = blockIdx.x;
    if (first >= ntiles) return;
    const int my_tiles = (ntiles - first + stride - 1) / stride;   // tiles of this workgroup

    // ---- stationary weights
    uint4 wst[8];
    if (grpA) {
        if constexpr (!L0) {
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) wst[kc] = *(const uint4*)(a.wq + (long)(wave * 16 + fr) * C + kc * 32 + fg * 8);
        } else {
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) wst[kc] = make_uint4(0, 0, 0, 0);
        }
    } else {
        const int wb = wave - 8;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
                wst[ni * 4 + kc] = *(const uint4*)(a.wo + (long)(wb * 32 + ni * 16 + fr) * CI + kc * 32 + fg * 8);
    }

    // keys tile staging (group A threads, 2 chunks each)
    uint4 ra0 = make_uint4(0, 0, 0, 0), ra1 = ra0;
#define DL_SRC(p_, tile_) (a.xin + ((long)(tile_) * WM + ((p_) * 512 + tid) / CPR) * C + (((p_) * 512 + tid) % CPR) * 8)
#define DL_DST(p_, buf_) ldsK[(buf_) * WM * CPR + (((p_) * 512 + tid) / CPR) * CPR + \
                              ((((p_) * 512 + tid) % CPR) ^ swzr(((p_) * 512 + tid) / CPR))]
    if constexpr (!L0) {
        if (grpA) {
            ra0 = *(const uint4*)DL_SRC(0, first); ra1 = *(const uint4*)DL_SRC(1, first);
            DL_DST(0, 0) = ra0; DL_DST(1, 0) = ra1;
        }
    }
    __syncthreads();

    // iteration i: A handles tile i (i < my_tiles), B handles tile i-1 (i >= 1)
    for (int i = 0; i <= my_tiles; ++i) {
        if (grpA) {
            const bool work = i < my_tiles;
            const int tile = first + i * stride;
            const long row0 = (long)tile * WM;
            const int p = (int)(row0 / T), t0 = (int)(row0 - (long)p * T);
            const int head = wave;
            const bool have_next = (i + 1 < my_tiles);
            if constexpr (!L0) {
                if (have_next) { ra0 = *(const uint4*)DL_SRC(0, tile + stride); ra1 = *(const uint4*)DL_SRC(1, tile + stride); }
            }
            uint4 qb0 = make_uint4(0, 0, 0, 0), qb1 = qb0;       // B operands of S^T for the two token tiles
            uint4 ka = make_uint4(0, 0, 0, 0), va = make_uint4(0, 0, 0, 0);
            if (work) {
                if constexpr (!L0) {
                    // q^T = Wq . keys^T : rows d = fg*4 + r of this head, cols = tokens
                    const uint4* lk = ldsK + (i % 3) * WM * CPR;
                    f32x4_t qa0 = {0.f, 0.f, 0.f, 0.f}, qa1 = qa0;
#pragma unroll
                    for (int kc = 0; kc < 8; ++kc) {
                        const int r0 = fr, r1 = 16 + fr;
                        qa0 = mfma16d(wst[kc], lk[r0 * CPR + ((kc * 4 + fg) ^ swzr(r0))], qa0);
                        qa1 = mfma16d(wst[kc], lk[r1 * CPR + ((kc * 4 + fg) ^ swzr(r1))], qa1);
                    }
                    const float4 b4 = *(const float4*)(a.bq + head * 16 + fg * 4);
                    const float4 pe0 = *(const float4*)(a.peq + (long)(t0 + fr) * CI + head * 16 + fg * 4);
                    const float4 pe1 = *(const float4*)(a.peq + (long)(t0 + 16 + fr) * CI + head * 16 + fg * 4);
                    qb0.x = pack2d(qa0[0] + b4.x + pe0.x, qa0[1] + b4.y + pe0.y);
                    qb0.y = pack2d(qa0[2] + b4.z + pe0.z, qa0[3] + b4.w + pe0.w);
                    qb1.x = pack2d(qa1[0] + b4.x + pe1.x, qa1[1] + b4.y + pe1.y);
                    qb1.y = pack2d(qa1[2] + b4.z + pe1.z, qa1[3] + b4.w + pe1.w);
                    // k_tok fragment with the SAME k-slot map: slots i < 4 <-> d = fg*4 + i
                    if (fr < a.Nt) {
                        const uint2 k2 = *(const uint2*)(a.ktok + ((long)p * a.Nt + fr) * CI + head * 16 + fg * 4);
                        ka.x = k2.x; ka.y = k2.y;
                    }
                } else {
                    // prompt-independent q rows and k_tok in the natural map: slots (fg < 2, i) <-> d = fg*8 + i
                    if (fg < 2) {
                        qb0 = *(const uint4*)(a.q_shared + (long)(t0 + fr) * CI + head * 16 + fg * 8);
                        qb1 = *(const uint4*)(a.q_shared + (long)(t0 + 16 + fr) * CI + head * 16 + fg * 8);
                        if (fr < a.Nt) ka = *(const uint4*)(a.ktok + ((long)p * a.Nt + fr) * CI + head * 16 + fg * 8);
                    }
                }
                // V_tok^T fragment: lane (fr = d, fg): slots i < 4 <-> j = fg*4 + i
                {
                    u16 e0 = 0, e1 = 0, e2 = 0, e3 = 0;
                    const u16* vb = a.vtok + (long)p * a.Nt * CI + head * 16 + fr;
                    if (fg * 4 + 0 < a.Nt) e0 = vb[(fg * 4 + 0) * CI];
                    if (fg * 4 + 1 < a.Nt) e1 = vb[(fg * 4 + 1) * CI];
                    if (fg * 4 + 2 < a.Nt) e2 = vb[(fg * 4 + 2) * CI];
                    if (fg * 4 + 3 < a.Nt) e3 = vb[(fg * 4 + 3) * CI];
                    va.x = (uint32_t)e0 | ((uint32_t)e1 << 16); va.y = (uint32_t)e2 | ((uint32_t)e3 << 16);
                }
            }
            __syncthreads();                                   // barrier 1 (B: fp32 tile written)
            if (work) {
                uint4* lp = ldsP + (i & 1) * WM * 16;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int row = mt * 16 + fr;
                    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
                    s = mfma16d(ka, mt == 0 ? qb0 : qb1, s);    // rows j = fg*4 + r, col token = fr
                    float mx = NEG_BIG;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s[r] = (fg * 4 + r < a.Nt) ? s[r] * 0.25f : NEG_BIG; mx = fmaxf(mx, s[r]); }
                    mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));
                    float ps = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s[r] = __expf(s[r] - mx); ps += s[r]; }
                    ps += __shfl_xor(ps, 16); ps += __shfl_xor(ps, 32);
                    uint4 pb; pb.x = pack2d(s[0], s[1]); pb.y = pack2d(s[2], s[3]); pb.z = 0; pb.w = 0;
                    f32x4_t o = {0.f, 0.f, 0.f, 0.f};
                    o = mfma16d(va, pb, o);                     // rows d = fg*4 + r, col token = fr
                    const float inv = 1.f / ps;
                    uint2 pk; pk.x = pack2d(o[0] * inv, o[1] * inv); pk.y = pack2d(o[2] * inv, o[3] * inv);
                    uint2* dst = (uint2*)(lp + row * 16 + ((head * 2 + (fg >> 1)) ^ swzr(row)));
                    dst[fg & 1] = pk;
                }
            }
            if constexpr (!L0) {
                if (have_next) { DL_DST(0, (i + 1) % 3) = ra0; DL_DST(1, (i + 1) % 3) = ra1; }
            }
            __syncthreads();                                   // barrier 2 (end of iteration)
        } else {
            const bool work = i >= 1;
            const int tile = first + (i - 1) * stride;
            const long row0 = (long)tile * WM;
            const int t0 = (int)(row0 % T);
            const int wb = wave - 8;
            if (work) {
                const uint4* lp = ldsP + ((i - 1) & 1) * WM * 16;
                f32x4_t oc00 = {0.f, 0.f, 0.f, 0.f}, oc01 = oc00, oc10 = oc00, oc11 = oc00;
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const int r0 = fr, r1 = 16 + fr;
                    const uint4 af0 = lp[r0 * 16 + ((kc * 4 + fg) ^ swzr(r0))];
                    const uint4 af1 = lp[r1 * 16 + ((kc * 4 + fg) ^ swzr(r1))];
                    oc00 = mfma16d(af0, wst[kc], oc00); oc01 = mfma16d(af0, wst[4 + kc], oc01);
                    oc10 = mfma16d(af1, wst[kc], oc10); oc11 = mfma16d(af1, wst[4 + kc], oc11);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ldsC[(fg * 4 + r) * C + wb * 32 + fr] = oc00[r];
                    ldsC[(fg * 4 + r) * C + wb * 32 + 16 + fr] = oc01[r];
                    ldsC[(16 + fg * 4 + r) * C + wb * 32 + fr] = oc10[r];
                    ldsC[(16 + fg * 4 + r) * C + wb * 32 + 16 + fr] = oc11[r];
                }
            }
            __syncthreads();                                   // barrier 1
            if (work) {
                const int col = lane * 4;
#pragma unroll 2
                for (int pass = 0; pass < 4; ++pass) {
                    const float4 b4 = *(const float4*)(a.bo + col), w4 = *(const float4*)(a.ln_w + col), g4 = *(const float4*)(a.ln_b + col);
                    const int lr = pass * 8 + wb;
                    const float4 c = *(const float4*)(ldsC + lr * C + col);
                    uint2 rs;
                    if constexpr (!L0) {
                        const uint4* lk = ldsK + ((i - 1) % 3) * WM * CPR;
                        rs = ((const uint2*)(lk + lr * CPR + ((lane >> 1) ^ swzr(lr))))[lane & 1];
                    } else {
                        rs = *(const uint2*)(a.xin + (long)(t0 + lr) * C + col);
                    }
                    float v0 = c.x + b4.x + d2f((u16)(rs.x & 0xffff)), v1 = c.y + b4.y + d2f((u16)(rs.x >> 16));
                    float v2 = c.z + b4.z + d2f((u16)(rs.y & 0xffff)), v3 = c.w + b4.w + d2f((u16)(rs.y >> 16));
                    const float mean = wave_sum64((v0 + v1) + (v2 + v3)) * (1.0f / 256.0f);
                    v0 -= mean; v1 -= mean; v2 -= mean; v3 -= mean;
                    const float var = wave_sum64((v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3)) * (1.0f / 256.0f);
                    const float rstd = 1.0f / sqrtf(var + a.eps);
                    uint2 pk;
                    pk.x = pack2d(v0 * rstd * w4.x + g4.x, v1 * rstd * w4.y + g4.y);
                    pk.y = pack2d(v2 * rstd * w4.z + g4.z, v3 * rstd * w4.w + g4.w);
                    *(uint2*)(a.out + (row0 + lr) * C + col) = pk;
                }
            }
            __syncthreads();                                   // barrier 2
        }
    }
#undef DL_SRC
#undef DL_DST
}

template <bool L0>
int launch(const LayerArgs& a, hipStream_t s) {
    constexpr int LDS_BYTES = (L0 ? 0 : 3 * WM * CPR * 16) + 2 * WM * 16 * 16 + WM * C * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)dec_image_layer_kernel<L0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) != hipSuccess) { msam_set_error("msam_decoder_image_layer: cannot raise the LDS limit"); return 2; }
        attr_set = true;
    }
    const int ntiles = a.rows / WM;
    const int grid = ntiles < 256 ? ntiles : 256;                 // one 16-wave workgroup per CU
    const double flops = 2.0 * a.rows * ((L0 ? 0.0 : (double)CI * C) + (double)C * CI);
    const double bytes = (double)a.rows * ((L0 ? 0.0 : C * 2.0) + C * 2.0);   // read the stream (layer 1), write it
    msam_profile_mark2(s, 1, flops, bytes, 1);
    hipLaunchKernelGGL((dec_image_layer_kernel<L0>), dim3(grid), dim3(NTHR), LDS_BYTES, s, a);
    msam_profile_mark2(s, 0, 0.0, 0.0, 1);
    return msam_check_launch("msam_decoder_image_layer");
}

}  // namespace

extern "C" int msam_decoder_image_layer(const msam_image_layer_t* p, void* stream) {
    if (!p || !p->xin || !p->wo || !p->bo || !p->ln_w || !p->ln_b || !p->ktok || !p->vtok || !p->out) {
        msam_set_error("msam_decoder_image_layer: null argument");
        return 1;
    }
    if (p->rows <= 0 || p->rows % T || p->Nt <= 0 || p->Nt > 16) { msam_set_error("msam_decoder_image_layer: bad sizes"); return 1; }
    LayerArgs a;
    a.xin = (const u16*)p->xin; a.q_shared = (const u16*)p->q_shared; a.wq = (const u16*)p->wq; a.bq = p->bq; a.peq = p->peq;
    a.wo = (const u16*)p->wo; a.bo = p->bo; a.ln_w = p->ln_w; a.ln_b = p->ln_b; a.eps = p->ln_eps;
    a.ktok = (const u16*)p->ktok; a.vtok = (const u16*)p->vtok; a.Nt = p->Nt; a.out = (u16*)p->out; a.rows = p->rows;
    hipStream_t s = (hipStream_t)stream;
    if (p->q_shared) return launch<true>(a, s);
    if (!p->wq || !p->bq || !p->peq) { msam_set_error("msam_decoder_image_layer: layer-1 form needs wq, bq, peq"); return 1; }
    return launch<false>(a, s);
}
