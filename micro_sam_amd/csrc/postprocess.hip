// Mask post-processing on the device (HBM-bound integer / byte work):
//   Sam.postprocess_masks (two bilinear resamplings) + calculate_stability_score + threshold +
//   batched_mask_to_box + column-major RLE  (micro_sam/instance_segmentation.py:229-255,
//   micro_sam/_vendored.py:33-152).  Full-resolution fp32 logits are never written unless asked for.
//
// The bilinear arithmetic reproduces torch's CPU kernel bit for bit (verified in tests/): for each axis
//   scale = in / out (fp32), src = max(scale * (dst + 0.5) - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, in - 1),
//   w1 = src - i0, w0 = 1 - w1,   value = fma(w0, p0, w1 * p1)   (inner axis x first, then y).
// so all integer outputs (counts, boxes, bit masks, RLE) are exact given the same low-res logits.
//
// Bit-mask layout: bits[n][yw][x], one uint32 = 32 consecutive rows y = yw*32 + b of column x -> coalesced stores
// here and coalesced column-major walks in the RLE kernels.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);

namespace {

struct Axis { int i0, i1; float w0, w1; };
// fp16 low-res logits as stored by up_fused_kernel (msam_upscale_fused_out, MSAM_F16): widened to fp32 on load
struct lowh_t { unsigned short bits; };
MSAM_DEVINL float lowf(float v) { return v; }
MSAM_DEVINL float lowf(lowh_t v) { return h2f(v.bits); }

MSAM_DEVINL Axis axis_weights(int dst, float scale, int in_size) {
    // torch: scale * (dst + 0.5) - 0.5 with separately rounded mul / sub
    float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
    if (src < 0.f) src = 0.f;
    Axis a;
    a.i0 = (int)src;                                            // floor for src >= 0
    a.i1 = a.i0 + (a.i0 < in_size - 1 ? 1 : 0);
    a.w1 = __fsub_rn(src, (float)a.i0);
    a.w0 = __fsub_rn(1.0f, a.w1);
    return a;
}

MSAM_DEVINL float lerp_torch(float w0, float p0, float w1, float p1) { return __fmaf_rn(w0, p0, __fmul_rn(w1, p1)); }

// value of the 1024x1024 intermediate (x4 up-sampling of the 256x256 low-res mask) at (Y, X)
template <typename TL>
MSAM_DEVINL float stage1(const TL* __restrict__ low, int Y, int X) {
    const Axis ay = axis_weights(Y, 0.25f, 256), ax = axis_weights(X, 0.25f, 256);
    const TL* r0 = low + ay.i0 * 256; const TL* r1 = low + ay.i1 * 256;
    const float t0 = lerp_torch(ax.w0, lowf(r0[ax.i0]), ax.w1, lowf(r0[ax.i1]));
    const float t1 = lerp_torch(ax.w0, lowf(r1[ax.i0]), ax.w1, lowf(r1[ax.i1]));
    return lerp_torch(ay.w0, t0, ay.w1, t1);
}

__global__ void init_stats_kernel(int* __restrict__ counts, int* __restrict__ boxes, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    counts[n * 3] = 0; counts[n * 3 + 1] = 0; counts[n * 3 + 2] = 0;
    boxes[n * 4] = 0x7fffffff; boxes[n * 4 + 1] = 0x7fffffff; boxes[n * 4 + 2] = -1; boxes[n * 4 + 3] = -1;
}

__global__ void finalize_boxes_kernel(int* __restrict__ boxes, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    if (boxes[n * 4 + 2] < boxes[n * 4] || boxes[n * 4 + 3] < boxes[n * 4 + 1]) {
        boxes[n * 4] = 0; boxes[n * 4 + 1] = 0; boxes[n * 4 + 2] = 0; boxes[n * 4 + 3] = 0;
    }
}

// grid (ceil(out_w/256), N); thread = column x, loops over all 32-row words of its column (so a mask contributes
// only ceil(W/256) x 7 atomics instead of one set per 32x256 patch - the atomics were the bottleneck)
// LOGITS (store the up-sampled logits as well) is a compile-time switch: as a run-time test it put a scalar branch and
// the address arithmetic of the store between every two of the 32 unrolled pixels of the hot loop
// TL: type of the low-res logits (float, or lowh_t = the AMG path's 16-bit hand-over from up_fused_kernel; widened on load)
template <bool TWO_STAGE, bool LOGITS, typename TL>
__global__ __launch_bounds__(256) void postprocess_kernel(const TL* __restrict__ low_res, int in_h, int in_w, int out_h,
                                                          int out_w, float thr, float off, int* __restrict__ counts,
                                                          int* __restrict__ boxes, uint32_t* __restrict__ bits,
                                                          float* __restrict__ logits) {
    __shared__ int red[4][7];
    __shared__ float tile[2][10 * 66];
    const int x = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    const int wpc = (out_h + 31) >> 5;
    const TL* low = low_res + (long)n * 65536;
    const float hi_t = thr + off, lo_t = thr - off;
    int c_hi = 0, c_lo = 0, c_m = 0, ymin = 0x7fffffff, ymax = -1;
    bool any = false;
    if (!TWO_STAGE) {
        // x4 fast path: the horizontal lerp of the 10 low-res rows under a 32-row word is shared by its 32 pixels
        // (same arithmetic as stage1(), so still bit-exact): pixel b uses rows c, c+1 with c = (b + 2) / 4;
        // vertical weights are compile-time constants: frac(src) in {.625,.875,.125,.375} by b % 4 (exactly what
        // axis_weights computes in fp32), except the two clamped rows at the top of the image.
        // The 10 x 66 low-res patch under the block's 256 columns is staged through LDS (one coalesced load of 660
        // floats per word instead of 20 mostly redundant dword loads per thread), next word's patch in flight during
        // the 32-row loop.
        const int cbase = max((int)blockIdx.x * 64 - 1, 0);
        int sr[3], sc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = threadIdx.x + k * 256;
            sr[k] = idx / 66; sc[k] = min(cbase + idx % 66, 255);
        }
        float nx[3] = {0.f, 0.f, 0.f};
#define PP_LOAD(yw_)                                                                               \
        _Pragma("unroll") for (int k = 0; k < 3; ++k)                                              \
            if (threadIdx.x + k * 256 < 660) nx[k] = lowf(low[min(max((yw_) * 8 - 1 + sr[k], 0), 255) * 256 + sc[k]]);
#define PP_STORE(buf_)                                                                             \
        _Pragma("unroll") for (int k = 0; k < 3; ++k)                                              \
            if (threadIdx.x + k * 256 < 660) tile[buf_][threadIdx.x + k * 256] = nx[k];
        PP_LOAD(0);
        PP_STORE(0);
        __syncthreads();
        const Axis ax = axis_weights(min(x, out_w - 1), 0.25f, 256);
        const int l0 = ax.i0 - cbase, l1 = ax.i1 - cbase;
        const float W1[4] = {0.625f, 0.875f, 0.125f, 0.375f};
        for (int yw = 0; yw < wpc; ++yw) {
            if (yw + 1 < wpc) { PP_LOAD(yw + 1); }
            const float* tl = tile[yw & 1];
            if (x < out_w) {
                float t[10];
#pragma unroll
                for (int k = 0; k < 10; ++k) t[k] = lerp_torch(ax.w0, tl[k * 66 + l0], ax.w1, tl[k * 66 + l1]);
                const int nb = min(32, out_h - yw * 32);
                // the three predicates v > thr, v > thr + off, v > thr - off are collected as sign bits of (t - v) (strictly
                // negative <=> v > t; t - v == +0 for equality) shifted into three words: 2 instructions per predicate and
                // pixel instead of compare + add-with-carry / select + or; pixel 0 ends up in bit 31 -> bit reverse
                uint32_t wm = 0, wh = 0, wl = 0;
                uint32_t word;
                if (nb == 32) {
                    // full word (every word of a 1024-row mask).  Only the mask bits are needed per lane; the two stability
                    // counts are per-MASK totals, so each is one v_cmp into a scalar lane mask + a scalar popcount (SALU, a
                    // separate issue port) instead of a subtract + alignbit per pixel on the VALU: 6 instead of 8 VALU
                    // instructions per pixel in this VALU-bound loop (same comparisons: v > t  <=>  sign(t - v) set)
                    int s_hi = 0, s_lo = 0;
                    // Decided words: every pixel of the word is a convex combination of two of the ten t[k] (vertical weights
                    // w0 + w1 = 1 exactly), so it lies in [min t, max t] up to two fp32 roundings (< 0.003 for |t| < 8192).  If
                    // that interval clears all three thresholds by 0.01 on the same side the 32 comparisons are known without
                    // interpolating - for a real mask that is every word away from the object boundary.  The whole wave must be
                    // decided (ballot): otherwise the exact per-pixel path below runs for all of its lanes.
                    bool decided = false;
                    if (!LOGITS) {
                        const float tmin = fminf(fminf(fminf(t[0], t[1]), fminf(t[2], t[3])), fminf(fminf(t[4], t[5]), fminf(fminf(t[6], t[7]), fminf(t[8], t[9]))));
                        const float tmax = fmaxf(fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3])), fmaxf(fmaxf(t[4], t[5]), fmaxf(fmaxf(t[6], t[7]), fmaxf(t[8], t[9]))));
                        const bool one = tmin > hi_t + 0.01f && tmax < 8192.f, zero = tmax < lo_t - 0.01f && tmin > -8192.f;
                        if (__ballot(!(one || zero)) == 0) {
                            decided = true;
                            word = one ? 0xffffffffu : 0u;
                            s_hi = s_lo = 32 * __popcll(__ballot(one));
                        }
                    }
                    if (!decided) {
#pragma unroll
                        for (int b = 0; b < 32; ++b) {
                            float w1 = W1[b & 3];
                            if (b < 2 && yw == 0) w1 = 0.0f;
                            const float v = lerp_torch(1.0f - w1, t[(b + 2) / 4], w1, t[(b + 2) / 4 + 1]);
                            if (LOGITS) logits[((long)n * out_h + yw * 32 + b) * out_w + x] = v;
                            wm = __builtin_amdgcn_alignbit(wm, __float_as_uint(__fsub_rn(thr, v)), 31);
                            s_hi += __popcll(__ballot(v > hi_t));
                            s_lo += __popcll(__ballot(v > lo_t));
                        }
                        word = __brev(wm);
                    }
                    if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) { c_hi += s_hi; c_lo += s_lo; }   // first active lane
                } else {
#pragma unroll
                    for (int b = 0; b < 32; ++b) {
                        float w1 = W1[b & 3];
                        if (b < 2 && yw == 0) w1 = 0.0f;
                        const float v = lerp_torch(1.0f - w1, t[(b + 2) / 4], w1, t[(b + 2) / 4 + 1]);
                        if (LOGITS) { if (b < nb) logits[((long)n * out_h + yw * 32 + b) * out_w + x] = v; }
                        wm = __builtin_amdgcn_alignbit(wm, __float_as_uint(__fsub_rn(thr, v)), 31);
                        wh = __builtin_amdgcn_alignbit(wh, __float_as_uint(__fsub_rn(hi_t, v)), 31);
                        wl = __builtin_amdgcn_alignbit(wl, __float_as_uint(__fsub_rn(lo_t, v)), 31);
                    }
                    const uint32_t valid = (1u << nb) - 1u;
                    word = __brev(wm) & valid;
                    c_hi += __popc(__brev(wh) & valid); c_lo += __popc(__brev(wl) & valid);
                }
                if (word) {
                    c_m += __popc(word); any = true;
                    ymin = min(ymin, yw * 32 + __ffs(word) - 1); ymax = yw * 32 + 31 - __clz(word);
                }
                bits[((long)n * wpc + yw) * out_w + x] = word;
            }
            if (yw + 1 < wpc) { PP_STORE((yw + 1) & 1); }
            __syncthreads();
        }
#undef PP_LOAD
#undef PP_STORE
    } else {
        // (every lane of the wave walks the loop - the wave-wide vote below needs them all; lanes right of the image mirror the last
        //  column and neither count nor store)
        const bool live = x < out_w;
        const int x = min((int)(blockIdx.x * 256 + threadIdx.x), out_w - 1);
        // General case (the image is not 1024 x 1024): stage 2 resamples the in_h x in_w corner of the x4 intermediate to out_h x out_w.
        // Same arithmetic as stage1() + two lerps per pixel, but the column x only ever needs two columns X0, X1 of the intermediate,
        // and both the low-res rows and the intermediate rows are visited in ascending order - so the horizontal lerps of a low-res row
        // (for X0 and X1) and the two intermediate values of a row Y are kept in two-entry caches instead of being recomputed from 16
        // global loads per pixel (round 3: this path was 16x slower than the x4 path and is the one every non-1024^2 image takes).
        const float sx = (float)in_w / (float)out_w, sy = (float)in_h / (float)out_h;
        const Axis ax2 = axis_weights(x, sx, in_w);
        const Axis axa = axis_weights(ax2.i0, 0.25f, 256), axb = axis_weights(ax2.i1, 0.25f, 256);     // low-res columns under X0 / X1
        int lr[2] = {-1, -1};                    // cached low-res rows
        float la[2] = {0.f, 0.f}, lb[2] = {0.f, 0.f};           // their horizontal lerps at X0 / X1
        int lnext = 0;
        auto low_row = [&](int r, float& va, float& vb) {
            if (r == lr[0]) { va = la[0]; vb = lb[0]; return; }
            if (r == lr[1]) { va = la[1]; vb = lb[1]; return; }
            const TL* row = low + r * 256;
            va = lerp_torch(axa.w0, lowf(row[axa.i0]), axa.w1, lowf(row[axa.i1]));
            vb = lerp_torch(axb.w0, lowf(row[axb.i0]), axb.w1, lowf(row[axb.i1]));
            lr[lnext] = r; la[lnext] = va; lb[lnext] = vb; lnext ^= 1;
        };
        int ir[2] = {-1, -1};                    // cached intermediate rows Y
        float ia[2] = {0.f, 0.f}, ib[2] = {0.f, 0.f};           // I(Y, X0), I(Y, X1)
        int inext = 0;
        auto inter_row = [&](int Y, float& va, float& vb) {
            if (Y == ir[0]) { va = ia[0]; vb = ib[0]; return; }
            if (Y == ir[1]) { va = ia[1]; vb = ib[1]; return; }
            const Axis ay = axis_weights(Y, 0.25f, 256);
            float a0, b0, a1, b1;
            low_row(ay.i0, a0, b0);
            low_row(ay.i1, a1, b1);
            va = lerp_torch(ay.w0, a0, ay.w1, a1);
            vb = lerp_torch(ay.w0, b0, ay.w1, b1);
            ir[inext] = Y; ia[inext] = va; ib[inext] = vb; inext ^= 1;
        };
        // low-res columns under the workgroup's output columns (monotone in x): first thread's left one .. last thread's right one
        int gclo, gchi;
        {
            const int xf = min((int)blockIdx.x * 256, out_w - 1), xl = min((int)blockIdx.x * 256 + 255, out_w - 1);
            const Axis f2 = axis_weights(xf, sx, in_w), l2 = axis_weights(xl, sx, in_w);
            gclo = axis_weights(f2.i0, 0.25f, 256).i0; gchi = axis_weights(l2.i1, 0.25f, 256).i1;
        }
        for (int yw = 0; yw < wpc; ++yw) {
            uint32_t word = 0;
            // Decided words (as in the x4 path): every pixel of the word is a convex combination (three nested lerps, weights w0 + w1 = 1)
            // of the low-res values under it - rows [first pixel's upper source row, last pixel's lower one], columns [X0's left, X1's
            // right] - so it lies in their [min, max] up to a few fp32 roundings.  If that interval clears the three thresholds by
            // 0.01 on one side for the whole wave, the 32 comparisons are known without interpolating: ~35 loads + min / max instead
            // of 32 pixels x ~50 instructions.  Away from the object boundaries that is every word (round 5: this path is the one every
            // image that is not 1024 pixels long takes, and 4 of the 9 tiles of a 2048^2 slice at tile 768 + halo 128).
            if (!LOGITS) {
                const int y0 = yw * 32, nb = min(32, out_h - y0);
                const int Ylo = axis_weights(y0, sy, in_h).i0, Yhi = axis_weights(y0 + nb - 1, sy, in_h).i1;
                const int rlo = axis_weights(Ylo, 0.25f, 256).i0, rhi = axis_weights(Yhi, 0.25f, 256).i1;
                // the rows are the same for the whole workgroup: every thread reduces ONE low-res column over them into LDS (the
                // columns under the workgroup's 256 output columns), then reads the two to four columns of its own pixel
                float* cm = &tile[yw & 1][0];                         // [2][<= 256]: min | max (the x4 path's LDS, unused here)
                if ((int)threadIdx.x <= gchi - gclo) {
                    const int c = gclo + threadIdx.x;
                    float a = 3.0e38f, b = -3.0e38f;
                    for (int r = rlo; r <= rhi; ++r) { const float v = lowf(low[r * 256 + c]); a = fminf(a, v); b = fmaxf(b, v); }
                    cm[threadIdx.x] = a; cm[256 + threadIdx.x] = b;
                }
                __syncthreads();
                const int clo = min(axa.i0, axb.i0), chi = max(axa.i1, axb.i1);
                float mn = 3.0e38f, mx = -3.0e38f;
                for (int c = clo; c <= chi; ++c) { mn = fminf(mn, cm[c - gclo]); mx = fmaxf(mx, cm[256 + c - gclo]); }
                const bool one = mn > hi_t + 0.01f && mx < 8192.f, zero = mx < lo_t - 0.01f && mn > -8192.f;
                if (__ballot(!(one || zero)) == 0) {
                    if (!live) continue;
                    if (one) {
                        word = nb == 32 ? 0xffffffffu : (1u << nb) - 1u;
                        c_hi += nb; c_lo += nb; c_m += nb; any = true;
                        ymin = min(ymin, y0); ymax = y0 + nb - 1;
                    }
                    bits[((long)n * wpc + yw) * out_w + x] = word;
                    continue;
                }
            }
            for (int b = 0; b < 32; ++b) {
                const int y = yw * 32 + b;
                if (y >= out_h) break;
                const Axis ay2 = axis_weights(y, sy, in_h);
                float p00, p01, p10, p11;
                inter_row(ay2.i0, p00, p01);
                inter_row(ay2.i1, p10, p11);
                const float t0 = lerp_torch(ax2.w0, p00, ax2.w1, p01);
                const float t1 = lerp_torch(ax2.w0, p10, ax2.w1, p11);
                const float v = lerp_torch(ay2.w0, t0, ay2.w1, t1);
                if (!live) continue;
                if (LOGITS) logits[((long)n * out_h + y) * out_w + x] = v;
                c_hi += v > hi_t; c_lo += v > lo_t;
                if (v > thr) { word |= 1u << b; ++c_m; any = true; ymin = min(ymin, y); ymax = y; }
            }
            if (live) bits[((long)n * wpc + yw) * out_w + x] = word;
        }
    }
    int xmin = any ? x : 0x7fffffff, xmax = any ? x : -1;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        c_hi += __shfl_xor(c_hi, s); c_lo += __shfl_xor(c_lo, s); c_m += __shfl_xor(c_m, s);
        ymin = min(ymin, __shfl_xor(ymin, s)); ymax = max(ymax, __shfl_xor(ymax, s));
        xmin = min(xmin, __shfl_xor(xmin, s)); xmax = max(xmax, __shfl_xor(xmax, s));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave][0] = c_hi; red[wave][1] = c_lo; red[wave][2] = c_m; red[wave][3] = xmin; red[wave][4] = ymin;
        red[wave][5] = xmax; red[wave][6] = ymax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int hi = 0, lo = 0, m = 0, x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -1, y1 = -1;
        for (int w = 0; w < 4; ++w) {
            hi += red[w][0]; lo += red[w][1]; m += red[w][2];
            x0 = min(x0, red[w][3]); y0 = min(y0, red[w][4]); x1 = max(x1, red[w][5]); y1 = max(y1, red[w][6]);
        }
        if (hi) atomicAdd(&counts[n * 3], hi);
        if (lo) atomicAdd(&counts[n * 3 + 1], lo);
        if (m) {
            atomicAdd(&counts[n * 3 + 2], m);
            atomicMin(&boxes[n * 4], x0); atomicMin(&boxes[n * 4 + 1], y0);
            atomicMax(&boxes[n * 4 + 2], x1); atomicMax(&boxes[n * 4 + 3], y1);
        }
    }
}

// ---------------------------------------------------------------------------------------------- RLE
// Flattened column-major sequence s = x*H + y.  A "transition" at s means bit[s] != bit[s-1] (bit[-1] := 0, so a
// mask starting with 1 yields a transition at s = 0, i.e. the leading 0-length run of the reference format).
// counts = differences of consecutive transition positions, plus the final run L - last.
// One workgroup (256 threads) per mask; thread handles columns x = tid, tid+256, ...

MSAM_DEVINL uint32_t col_prev_bit(const uint32_t* __restrict__ bm, int wpc, int out_w, int out_h, int x) {
    if (x == 0) return 0u;
    const int y = out_h - 1;
    return (bm[(long)(y >> 5) * out_w + (x - 1)] >> (y & 31)) & 1u;
}

MSAM_DEVINL int col_transitions(const uint32_t* __restrict__ bm, int wpc, int out_w, int out_h, int x, uint32_t prev) {
    int cnt = 0;
    for (int yw = 0; yw < wpc; ++yw) {
        uint32_t w = bm[(long)yw * out_w + x];
        const int nb = min(32, out_h - yw * 32);
        uint32_t shifted = (w << 1) | prev;
        uint32_t diff = w ^ shifted;
        if (nb < 32) diff &= (1u << nb) - 1u;
        cnt += __popc(diff);
        prev = (w >> (nb - 1)) & 1u;
    }
    return cnt;
}

__global__ __launch_bounds__(256) void rle_count_kernel(const uint32_t* __restrict__ bits, int out_h, int out_w,
                                                        int* __restrict__ n_runs) {
    __shared__ int red[4];
    const int n = blockIdx.x, wpc = (out_h + 31) >> 5;
    const uint32_t* bm = bits + (long)n * wpc * out_w;
    int cnt = 0;
    for (int x = threadIdx.x; x < out_w; x += 256) cnt += col_transitions(bm, wpc, out_w, out_h, x, col_prev_bit(bm, wpc, out_w, out_h, x));
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) cnt += __shfl_xor(cnt, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) n_runs[n] = red[0] + red[1] + red[2] + red[3] + 1;   // + the final run
}

__global__ __launch_bounds__(256) void rle_encode_kernel(const uint32_t* __restrict__ bits, int out_h, int out_w,
                                                         const long* __restrict__ offsets, int* __restrict__ out) {
    __shared__ int scan[256];
    __shared__ int carry;
    const int n = blockIdx.x, wpc = (out_h + 31) >> 5, tid = threadIdx.x;
    const uint32_t* bm = bits + (long)n * wpc * out_w;
    int* dst = out + offsets[n];
    if (tid == 0) carry = 0;
    __syncthreads();
    // pass A: transition positions, written to dst[k] (k-th transition of the mask)
    for (int x0 = 0; x0 < out_w; x0 += 256) {
        const int x = x0 + tid;
        uint32_t prev = 0; int cnt = 0;
        if (x < out_w) { prev = col_prev_bit(bm, wpc, out_w, out_h, x); cnt = col_transitions(bm, wpc, out_w, out_h, x, prev); }
        // exclusive scan of cnt over the 256 threads
        scan[tid] = cnt;
        __syncthreads();
        for (int s = 1; s < 256; s <<= 1) {
            int v = tid >= s ? scan[tid - s] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        int k = carry + scan[tid] - cnt;
        if (x < out_w) {
            for (int yw = 0; yw < wpc; ++yw) {
                uint32_t w = bm[(long)yw * out_w + x];
                const int nb = min(32, out_h - yw * 32);
                uint32_t diff = w ^ ((w << 1) | prev);
                if (nb < 32) diff &= (1u << nb) - 1u;
                while (diff) {
                    const int b = __ffs(diff) - 1;
                    diff &= diff - 1;
                    dst[k++] = x * out_h + yw * 32 + b;
                }
                prev = (w >> (nb - 1)) & 1u;
            }
        }
        __syncthreads();
        if (tid == 255) carry += scan[255];
        __syncthreads();
    }
    const int ntrans = carry;
    // pass B: in-place differences; element k needs pos[k] and pos[k-1] -> process chunks from the END so that a
    // chunk's predecessors are still untouched
    const int L = out_w * out_h;
    const int nchunks = (ntrans + 1 + 255) / 256;      // ntrans transitions + 1 final run
    for (int ch = nchunks - 1; ch >= 0; --ch) {
        const int k = ch * 256 + tid;
        int val = 0; bool ok = k <= ntrans;
        if (ok) {
            const int cur = k < ntrans ? dst[k] : L;
            const int prv = k > 0 ? dst[k - 1] : 0;
            val = cur - prv;
        }
        __syncthreads();
        if (ok) dst[k] = val;
        __syncthreads();
    }
}

}  // namespace

extern "C" int msam_postprocess_masks16(const void* low_res, int32_t low_res_dtype, int32_t N, int32_t in_h, int32_t in_w, int32_t out_h,
                                        int32_t out_w, float thr, float off, int32_t* counts, int32_t* boxes, uint32_t* bits,
                                        float* logits, void* stream) {
    if (!low_res || !counts || !boxes || !bits || N <= 0) { msam_set_error("msam_postprocess_masks: null argument"); return 1; }
    if (low_res_dtype != MSAM_F32 && low_res_dtype != MSAM_F16) { msam_set_error("msam_postprocess_masks16: low_res_dtype is MSAM_F32 or MSAM_F16"); return 1; }
    if (in_h <= 0 || in_w <= 0 || in_h > 1024 || in_w > 1024 || out_h <= 0 || out_w <= 0) {
        msam_set_error("msam_postprocess_masks: bad sizes");
        return 1;
    }
    if (N > 65535) { msam_set_error("msam_postprocess_masks: at most 65535 masks per call"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(init_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, s, counts, boxes, N);
    dim3 grid((out_w + 255) / 256, N);
#define PP_LAUNCH(TS_, LG_)                                                                                                              \
    do {                                                                                                                                 \
        if (low_res_dtype == MSAM_F16)                                                                                                   \
            hipLaunchKernelGGL((postprocess_kernel<TS_, LG_, lowh_t>), grid, dim3(256), 0, s, (const lowh_t*)low_res, in_h, in_w,        \
                               out_h, out_w, thr, off, counts, boxes, bits, logits);                                                     \
        else                                                                                                                             \
            hipLaunchKernelGGL((postprocess_kernel<TS_, LG_, float>), grid, dim3(256), 0, s, (const float*)low_res, in_h, in_w, out_h,   \
                               out_w, thr, off, counts, boxes, bits, logits);                                                            \
    } while (0)
    const bool direct = in_h == out_h && in_w == out_w;
    if (direct && !logits) PP_LAUNCH(false, false);
    else if (direct) PP_LAUNCH(false, true);
    else if (!logits) PP_LAUNCH(true, false);
    else PP_LAUNCH(true, true);
#undef PP_LAUNCH
    hipLaunchKernelGGL(finalize_boxes_kernel, dim3((N + 255) / 256), dim3(256), 0, s, boxes, N);
    return msam_check_launch("msam_postprocess_masks");
}

extern "C" int msam_postprocess_masks(const float* low_res, int32_t N, int32_t in_h, int32_t in_w, int32_t out_h,
                                      int32_t out_w, float thr, float off, int32_t* counts, int32_t* boxes, uint32_t* bits,
                                      float* logits, void* stream) {
    return msam_postprocess_masks16(low_res, MSAM_F32, N, in_h, in_w, out_h, out_w, thr, off, counts, boxes, bits, logits, stream);
}

// uncrop_masks of the reference's AMGBase._to_mask_data (instance_segmentation.py:250): bit masks of a crop
// [N, ceil(ch/32), cw] are placed at (x0, y0) of full-image bit masks [N, ceil(H/32), W] (zero outside the crop).  The
// row offset is arbitrary, so an output word combines two source words shifted by y0 mod 32.
__global__ __launch_bounds__(256) void uncrop_bits_kernel(const uint32_t* __restrict__ in, int ch, int cw, int x0, int y0,
                                                          int H, int W, uint32_t* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, yw = blockIdx.y, n = blockIdx.z;
    if (x >= W) return;
    const int wpc_in = (ch + 31) >> 5, wpc_out = (H + 31) >> 5;
    uint32_t word = 0u;
    const int xs = x - x0;
    if (xs >= 0 && xs < cw) {
        const int base = yw * 32 - y0;                       // source row of bit 0 of this output word
        const uint32_t* src = in + (long)n * wpc_in * cw + xs;
        if (base >= 0) {
            const int wl = base >> 5, sh = base & 31;
            if (wl < wpc_in) word = src[(long)wl * cw] >> sh;
            if (sh && wl + 1 < wpc_in) word |= src[(long)(wl + 1) * cw] << (32 - sh);
        } else if (base > -32) {
            word = src[0] << (-base);
        }
        const int rows = min(32, H - yw * 32);                 // bits beyond the image stay zero
        if (rows < 32) word &= (1u << rows) - 1u;
    }
    out[((long)n * wpc_out + yw) * W + x] = word;
}

extern "C" int msam_uncrop_bits(const uint32_t* bits_crop, int32_t N, int32_t crop_h, int32_t crop_w, int32_t x0, int32_t y0,
                                int32_t out_h, int32_t out_w, uint32_t* bits_out, void* stream) {
    if (!bits_crop || !bits_out || N <= 0 || crop_h <= 0 || crop_w <= 0 || out_h <= 0 || out_w <= 0 || x0 < 0 || y0 < 0 ||
        x0 + crop_w > out_w || y0 + crop_h > out_h) {
        msam_set_error("msam_uncrop_bits: bad arguments (the crop must lie inside the output image)");
        return 1;
    }
    if (N > 65535) { msam_set_error("msam_uncrop_bits: at most 65535 masks per call"); return 1; }
    dim3 grid((out_w + 255) / 256, (out_h + 31) / 32, N);
    hipLaunchKernelGGL(uncrop_bits_kernel, grid, dim3(256), 0, (hipStream_t)stream, bits_crop, crop_h, crop_w, x0, y0, out_h,
                       out_w, bits_out);
    return msam_check_launch("msam_uncrop_bits");
}

extern "C" int msam_rle_run_counts(const uint32_t* bits, int32_t N, int32_t out_h, int32_t out_w, int32_t* n_runs,
                                   void* stream) {
    if (!bits || !n_runs || N <= 0 || out_h <= 0 || out_w <= 0) { msam_set_error("msam_rle_run_counts: bad arguments"); return 1; }
    hipLaunchKernelGGL(rle_count_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, bits, out_h, out_w, n_runs);
    return msam_check_launch("msam_rle_run_counts");
}

extern "C" int msam_rle_encode(const uint32_t* bits, int32_t N, int32_t out_h, int32_t out_w, const int64_t* offsets,
                               int32_t* counts_out, void* stream) {
    if (!bits || !offsets || !counts_out || N <= 0) { msam_set_error("msam_rle_encode: bad arguments"); return 1; }
    hipLaunchKernelGGL(rle_encode_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, bits, out_h, out_w,
                       (const long*)offsets, counts_out);
    return msam_check_launch("msam_rle_encode");
}
