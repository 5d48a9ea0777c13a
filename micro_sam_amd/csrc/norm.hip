// HBM-bound helper kernels of the encoder / decoder: LayerNorm (row-wise, fp32 statistics), patch gather
// (+ fused Sam.preprocess for uint8 input), 3x3 im2col for the neck, casts.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);

namespace {
// 16-bit outputs: MSAM_F16 -> IEEE fp16 (mask decoder; the image encoder's fp16 mode), anything else -> bf16
MSAM_DEVINL uint32_t pk16(float lo, float hi, int dt) { return dt == MSAM_F16 ? pack2h(lo, hi) : pack2bf(lo, hi); }
MSAM_DEVINL u16 cv16(float v, int dt) { return dt == MSAM_F16 ? f2h(v) : f2bf(v); }

// One wave per row.  VEC = float4 loads per lane (dim == 256 * VEC) or 0 for the scalar path (dim <= 64*20).
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, long rows, int dim,
                                                        void* out, int out_dtype, int gelu, int nchw_hw) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * dim;
    constexpr int NV = VEC > 0 ? VEC * 4 : 20;
    float v[NV];
    float s = 0.f;
    if constexpr (VEC > 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float4 t = *(const float4*)(xr + (i * 64 + lane) * 4);
            v[i * 4 + 0] = t.x; v[i * 4 + 1] = t.y; v[i * 4 + 2] = t.z; v[i * 4 + 3] = t.w;
            s += (t.x + t.y) + (t.z + t.w);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int c = i * 64 + lane;
            v[i] = c < dim ? xr[c] : 0.f;
            s += v[i];
        }
    }
    const float mean = wave_sum64(s) / (float)dim;
    float q = 0.f;
    if constexpr (VEC > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { float d = v[i] - mean; q += d * d; }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) { int c = i * 64 + lane; float d = c < dim ? v[i] - mean : 0.f; q += d * d; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)dim + eps);

    auto emit = [&](int c, float val) {
        float y = (val - mean) * rstd * w[c] + b[c];
        if (gelu) y = gelu_erf(y);
        return y;
    };
    if constexpr (VEC > 0) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const int c = (i * 64 + lane) * 4;
            float y0 = emit(c, v[i * 4]), y1 = emit(c + 1, v[i * 4 + 1]), y2 = emit(c + 2, v[i * 4 + 2]),
                  y3 = emit(c + 3, v[i * 4 + 3]);
            if (nchw_hw > 0) {
                const long bimg = row / nchw_hw, t = row - bimg * nchw_hw;
                float* o = (float*)out + (bimg * dim + c) * (long)nchw_hw + t;
                o[0] = y0; o[nchw_hw] = y1; o[2L * nchw_hw] = y2; o[3L * nchw_hw] = y3;
            } else if (out_dtype == MSAM_F32) {
                *(float4*)((float*)out + row * dim + c) = make_float4(y0, y1, y2, y3);
            } else {
                uint2 pk; pk.x = pk16(y0, y1, out_dtype); pk.y = pk16(y2, y3, out_dtype);
                *(uint2*)((u16*)out + row * dim + c) = pk;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < dim) {
                float y = emit(c, v[i]);
                if (nchw_hw > 0) {
                    const long bimg = row / nchw_hw, t = row - bimg * nchw_hw;
                    ((float*)out)[(bimg * dim + c) * (long)nchw_hw + t] = y;
                } else if (out_dtype == MSAM_F32) ((float*)out)[row * dim + c] = y;
                else ((u16*)out)[row * dim + c] = cv16(y, out_dtype);
            }
        }
    }
}

// ---- fp8 (OCP e4m3) activations with one scale per row (= per token), BASELINE config 5 ------------------------------
// q = round_e4m3(y * 448 / amax(row)), scale = amax / 448  (448 = largest e4m3 value; an all-zero row gets scale 1).
MSAM_DEVINL uint32_t pack4_fp8(float a, float b, float c, float d) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
MSAM_DEVINL float wave_max64(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8)); v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

// LayerNorm with fp8 output: one wave per row, dim == 256 * VEC; the normalised row never leaves the registers
template <int VEC>
__global__ __launch_bounds__(256) void layernorm_fp8_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps, long rows, int dim,
                                                            uint32_t* __restrict__ out, float* __restrict__ row_scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * dim;
    float v[VEC * 4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const float4 t = *(const float4*)(xr + (i * 64 + lane) * 4);
        v[i * 4 + 0] = t.x; v[i * 4 + 1] = t.y; v[i * 4 + 2] = t.z; v[i * 4 + 3] = t.w;
        s += (t.x + t.y) + (t.z + t.w);
    }
    const float mean = wave_sum64(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VEC * 4; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)dim + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int c = (i * 64 + lane) * 4;
        const float4 w4 = *(const float4*)(w + c), b4 = *(const float4*)(b + c);
        v[i * 4 + 0] = (v[i * 4 + 0] - mean) * rstd * w4.x + b4.x; v[i * 4 + 1] = (v[i * 4 + 1] - mean) * rstd * w4.y + b4.y;
        v[i * 4 + 2] = (v[i * 4 + 2] - mean) * rstd * w4.z + b4.z; v[i * 4 + 3] = (v[i * 4 + 3] - mean) * rstd * w4.w + b4.w;
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[i * 4]), fabsf(v[i * 4 + 1])), fmaxf(fabsf(v[i * 4 + 2]), fabsf(v[i * 4 + 3]))));
    }
    amax = wave_max64(amax);
    const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / scale;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
        out[(row * dim) / 4 + i * 64 + lane] = pack4_fp8(v[i * 4] * inv, v[i * 4 + 1] * inv, v[i * 4 + 2] * inv, v[i * 4 + 3] * inv);
    if (lane == 0) row_scale[row] = scale;
}

// bf16 [rows, dim] -> fp8 [rows, dim] + row scales; one wave per row, 8 values (16 B) per lane and step, dim % 8 == 0, dim <= 5120
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const u16* __restrict__ x, long rows, int dim,
                                                             uint2* __restrict__ out, float* __restrict__ row_scale) {
    constexpr int MAXS = 10;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const u16* xr = x + row * dim;
    uint4 raw[MAXS];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        const int c = (i * 64 + lane) * 8;
        raw[i] = make_uint4(0, 0, 0, 0);
        if (c < dim) {
            raw[i] = *(const uint4*)(xr + c);
            const uint32_t ww[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                amax = fmaxf(amax, fmaxf(fabsf(bf2f((u16)(ww[k] & 0xffff))), fabsf(bf2f((u16)(ww[k] >> 16)))));
        }
    }
    amax = wave_max64(amax);
    const float scale = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / scale;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < dim) {
            const uint32_t ww[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
            float f[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { f[2 * k] = bf2f((u16)(ww[k] & 0xffff)) * inv; f[2 * k + 1] = bf2f((u16)(ww[k] >> 16)) * inv; }
            uint2 o; o.x = pack4_fp8(f[0], f[1], f[2], f[3]); o.y = pack4_fp8(f[4], f[5], f[6], f[7]);
            out[(row * dim + c) / 8] = o;
        }
    }
    if (lane == 0) row_scale[row] = scale;
}

// dim == 64 fast path: one 16-lane group per row (4 rows per wave), float4 per lane.
__global__ __launch_bounds__(256) void layernorm64_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float eps, long rows,
                                                          void* out, int out_dtype, int gelu) {
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c = (threadIdx.x & 15) * 4;
    if (row >= rows) return;   // rows % 16 == 0 is not required: whole 16-lane groups exit together
    float4 t = *(const float4*)(x + row * 64 + c);
    float mean = wave_sum_xor16((t.x + t.y) + (t.z + t.w)) * (1.0f / 64.0f);
    float d0 = t.x - mean, d1 = t.y - mean, d2 = t.z - mean, d3 = t.w - mean;
    float var = wave_sum_xor16((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) * (1.0f / 64.0f);
    float rstd = 1.0f / sqrtf(var + eps);
    float4 ww = *(const float4*)(w + c), bb = *(const float4*)(b + c);
    float y0 = d0 * rstd * ww.x + bb.x, y1 = d1 * rstd * ww.y + bb.y, y2 = d2 * rstd * ww.z + bb.z,
          y3 = d3 * rstd * ww.w + bb.w;
    if (gelu) { y0 = gelu_erf(y0); y1 = gelu_erf(y1); y2 = gelu_erf(y2); y3 = gelu_erf(y3); }
    if (out_dtype == MSAM_F32) *(float4*)((float*)out + row * 64 + c) = make_float4(y0, y1, y2, y3);
    else { uint2 pk; pk.x = pk16(y0, y1, out_dtype); pk.y = pk16(y2, y3, out_dtype); *(uint2*)((u16*)out + row * 64 + c) = pk; }
}

// hi + lo operand pairs ("split" sites: patch embedding and neck): a value enters its product as
// hi = round16(v), lo = round16(v - hi) (~16 significand bits for bf16); rows are laid out [hi | lo | hi] (K -> 3K) against weight
// rows [Whi | Whi | Wlo], so that ONE plain 16-bit GEMM over 3K forms hi*Whi + lo*Whi + hi*Wlo (the lo*Wlo term is 2^-16 relative).
MSAM_DEVINL float rt16(float v, int dt) { return dt == MSAM_F16 ? h2f(f2h(v)) : bf2f(f2bf(v)); }
MSAM_DEVINL void store8_split(u16* row, int K, int col, const float* v, int dt) {
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = v[j] - rt16(v[j], dt);
    uint4 hi, lo;
    hi.x = pk16(v[0], v[1], dt); hi.y = pk16(v[2], v[3], dt); hi.z = pk16(v[4], v[5], dt); hi.w = pk16(v[6], v[7], dt);
    lo.x = pk16(l[0], l[1], dt); lo.y = pk16(l[2], l[3], dt); lo.z = pk16(l[4], l[5], dt); lo.w = pk16(l[6], l[7], dt);
    *(uint4*)(row + col) = hi; *(uint4*)(row + K + col) = lo; *(uint4*)(row + 2 * K + col) = hi;
}

// thread -> 8 consecutive kx of one (patch, c, ky): out col = c*256 + ky*16 + kx
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int B, u16* __restrict__ out, int dt, int split) {
    const long total = (long)B * 4096 * 96;   // 768 / 8 chunks per patch row
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int chunk = (int)(i % 96);
        const long prow = i / 96;
        const int b = (int)(prow / 4096), pidx = (int)(prow % 4096);
        const int py = pidx >> 6, px = pidx & 63;
        const int c = chunk >> 5, ky = (chunk >> 1) & 15, kx0 = (chunk & 1) * 8;
        const float* src = img + (((long)b * 3 + c) * 1024 + (py * 16 + ky)) * 1024 + px * 16 + kx0;
        float4 a = *(const float4*)src, d = *(const float4*)(src + 4);
        if (split) {
            const float v[8] = {a.x, a.y, a.z, a.w, d.x, d.y, d.z, d.w};
            store8_split(out + prow * 2304, 768, chunk * 8, v, dt);
            continue;
        }
        uint4 pk; pk.x = pk16(a.x, a.y, dt); pk.y = pk16(a.z, a.w, dt); pk.z = pk16(d.x, d.y, dt); pk.w = pk16(d.z, d.w, dt);
        *(uint4*)(out + prow * 768 + chunk * 8) = pk;
    }
}

// Sam.preprocess fused: (u8 - mean) / std, zero pad to 1024 (micro_sam/models/build_sam.py:132-133)
__global__ __launch_bounds__(256) void patchify_u8_kernel(const uint8_t* __restrict__ img, int B, int h, int w,
                                                          u16* __restrict__ out, int dt, int split) {
    const float mean[3] = {123.675f, 116.28f, 103.53f};
    const float stdv[3] = {58.395f, 57.12f, 57.375f};
    const long total = (long)B * 4096 * 96;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int chunk = (int)(i % 96);
        const long prow = i / 96;
        const int b = (int)(prow / 4096), pidx = (int)(prow % 4096);
        const int py = pidx >> 6, px = pidx & 63;
        const int c = chunk >> 5, ky = (chunk >> 1) & 15, kx0 = (chunk & 1) * 8;
        const int y = py * 16 + ky, x0 = px * 16 + kx0;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int x = x0 + j;
            v[j] = (y < h && x < w)
                       ? __fdiv_rn((float)img[(((long)b * h + y) * w + x) * 3 + c] - mean[c], stdv[c]) : 0.f;
        }
        if (split) { store8_split(out + prow * 2304, 768, chunk * 8, v, dt); continue; }
        uint4 pk; pk.x = pk16(v[0], v[1], dt); pk.y = pk16(v[2], v[3], dt); pk.z = pk16(v[4], v[5], dt); pk.w = pk16(v[6], v[7], dt);
        *(uint4*)(out + prow * 768 + chunk * 8) = pk;
    }
}

// x bf16 [B,64,64,C] -> [B*4096, 9*C], column (ky*3+kx)*C + c, zero padding 1
__global__ __launch_bounds__(256) void im2col3x3_kernel(const u16* __restrict__ x, int B, int C, u16* __restrict__ out) {
    const int cpr = 9 * C / 8;   // chunks per row
    const long total = (long)B * 4096 * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int chunk = (int)(i % cpr);
        const long row = i / cpr;
        const int b = (int)(row / 4096), t = (int)(row % 4096);
        const int ty = t >> 6, tx = t & 63;
        const int tap = (chunk * 8) / C, c0 = chunk * 8 - tap * C;
        const int yy = ty + tap / 3 - 1, xx = tx + tap % 3 - 1;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (yy >= 0 && yy < 64 && xx >= 0 && xx < 64) val = *(const uint4*)(x + (((long)b * 64 + yy) * 64 + xx) * C + c0);
        *(uint4*)(out + row * (9L * C) + chunk * 8) = val;
    }
}

// split form of the neck's 3 x 3 gather: x fp32 [B,64,64,C] -> [B*4096, 2 * 9C] = [hi | lo] (the third [hi] block of the
// [hi | lo | hi] scheme is the first one again: the caller runs a second product on columns 0 .. 9C-1 with lda = 18C)
__global__ __launch_bounds__(256) void im2col3x3_split_kernel(const float* __restrict__ x, int B, int C, u16* __restrict__ out, int dt) {
    const int cpr = 9 * C / 8;
    const long total = (long)B * 4096 * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int chunk = (int)(i % cpr);
        const long row = i / cpr;
        const int b = (int)(row / 4096), t = (int)(row % 4096);
        const int ty = t >> 6, tx = t & 63;
        const int tap = (chunk * 8) / C, c0 = chunk * 8 - tap * C;
        const int yy = ty + tap / 3 - 1, xx = tx + tap % 3 - 1;
        uint4 hi = make_uint4(0, 0, 0, 0), lo = hi;
        if (yy >= 0 && yy < 64 && xx >= 0 && xx < 64) {
            const float* src = x + (((long)b * 64 + yy) * 64 + xx) * C + c0;
            const float4 a = *(const float4*)src, d = *(const float4*)(src + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, d.x, d.y, d.z, d.w};
            float l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) l[j] = v[j] - rt16(v[j], dt);
            hi.x = pk16(v[0], v[1], dt); hi.y = pk16(v[2], v[3], dt); hi.z = pk16(v[4], v[5], dt); hi.w = pk16(v[6], v[7], dt);
            lo.x = pk16(l[0], l[1], dt); lo.y = pk16(l[2], l[3], dt); lo.z = pk16(l[4], l[5], dt); lo.w = pk16(l[6], l[7], dt);
        }
        u16* dst = out + row * (18L * C) + chunk * 8;
        *(uint4*)dst = hi; *(uint4*)(dst + 9L * C) = lo;
    }
}

// x fp32 [rows, dim] -> [rows, 3 * dim] = [hi | lo | hi]
__global__ __launch_bounds__(256) void cast_split_kernel(const float* __restrict__ x, u16* __restrict__ out, long rows, int dim, int dt) {
    const int cpr = dim / 8;
    const long total = rows * cpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int chunk = (int)(i % cpr);
        const long row = i / cpr;
        const float* src = x + row * dim + chunk * 8;
        const float4 a = *(const float4*)src, d = *(const float4*)(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, d.x, d.y, d.z, d.w};
        store8_split(out + row * (3L * dim), dim, chunk * 8, v, dt);
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, u16* __restrict__ out, long n, int dt) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 t = *(const float4*)(x + i * 4);
        uint2 pk; pk.x = pk16(t.x, t.y, dt); pk.y = pk16(t.z, t.w, dt);
        *(uint2*)(out + i * 4) = pk;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[n4 * 4 + threadIdx.x] = cv16(x[n4 * 4 + threadIdx.x], dt);
}

inline int grid_for(long work_items) {
    long g = (work_items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int msam_layernorm(const float* x, const float* weight, const float* bias, float eps, int64_t rows,
                              int32_t dim, void* out, int32_t out_dtype, int32_t gelu, int32_t out_nchw_hw,
                              void* stream) {
    if (!x || !weight || !bias || !out || rows <= 0 || dim <= 0) { msam_set_error("msam_layernorm: bad arguments"); return 1; }
    if (out_nchw_hw > 0 && (out_dtype != MSAM_F32 || rows % out_nchw_hw)) {
        msam_set_error("msam_layernorm: NCHW output needs fp32 and rows % hw == 0");
        return 1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dim == 64 && out_nchw_hw == 0) {
        hipLaunchKernelGGL(layernorm64_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, x, weight, bias, eps,
                           (long)rows, out, out_dtype, gelu);
        return msam_check_launch("msam_layernorm");
    }
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define LN_LAUNCH(V) hipLaunchKernelGGL(layernorm_kernel<V>, grid, block, 0, s, x, weight, bias, eps, (long)rows, dim, \
                                        out, out_dtype, gelu, out_nchw_hw)
    if (dim == 256) LN_LAUNCH(1);
    else if (dim == 768) LN_LAUNCH(3);
    else if (dim == 1024) LN_LAUNCH(4);
    else if (dim == 1280) LN_LAUNCH(5);
    else if (dim <= 1280) LN_LAUNCH(0);
    else { msam_set_error("msam_layernorm: dim > 1280 unsupported"); return 1; }
#undef LN_LAUNCH
    return msam_check_launch("msam_layernorm");
}

extern "C" int msam_layernorm_fp8(const float* x, const float* weight, const float* bias, float eps, int64_t rows, int32_t dim,
                                  void* out_fp8, float* row_scale, void* stream) {
    if (!x || !weight || !bias || !out_fp8 || !row_scale || rows <= 0) { msam_set_error("msam_layernorm_fp8: bad arguments"); return 1; }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define LN8_LAUNCH(V) hipLaunchKernelGGL(layernorm_fp8_kernel<V>, grid, block, 0, s, x, weight, bias, eps, (long)rows, dim, \
                                         (uint32_t*)out_fp8, row_scale)
    if (dim == 768) LN8_LAUNCH(3);
    else if (dim == 1024) LN8_LAUNCH(4);
    else if (dim == 1280) LN8_LAUNCH(5);
    else { msam_set_error("msam_layernorm_fp8: dim must be 768, 1024 or 1280"); return 1; }
#undef LN8_LAUNCH
    return msam_check_launch("msam_layernorm_fp8");
}

extern "C" int msam_quant_rows_fp8(const void* x_bf16, int64_t rows, int32_t dim, void* out_fp8, float* row_scale, void* stream) {
    if (!x_bf16 || !out_fp8 || !row_scale || rows <= 0 || dim <= 0 || dim % 8 || dim > 5120) {
        msam_set_error("msam_quant_rows_fp8: bad arguments (dim % 8 == 0, dim <= 5120)");
        return 1;
    }
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)x_bf16, (long)rows, dim, (uint2*)out_fp8, row_scale);
    return msam_check_launch("msam_quant_rows_fp8");
}

extern "C" int msam_patchify16(const float* img, int32_t B, int32_t dtype16, void* out16, void* stream) {
    if (!img || !out16 || B <= 0) { msam_set_error("msam_patchify: bad arguments"); return 1; }
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((long)B * 4096 * 96)), dim3(256), 0, (hipStream_t)stream, img, B,
                       (u16*)out16, dtype16, 0);
    return msam_check_launch("msam_patchify");
}
extern "C" int msam_patchify_split16(const float* img, int32_t B, int32_t dtype16, void* out16, void* stream) {
    if (!img || !out16 || B <= 0) { msam_set_error("msam_patchify_split16: bad arguments"); return 1; }
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((long)B * 4096 * 96)), dim3(256), 0, (hipStream_t)stream, img, B,
                       (u16*)out16, dtype16, 1);
    return msam_check_launch("msam_patchify_split16");
}
extern "C" int msam_patchify(const float* img, int32_t B, void* out_bf16, void* stream) { return msam_patchify16(img, B, MSAM_BF16, out_bf16, stream); }

extern "C" int msam_patchify_u8_16(const uint8_t* img, int32_t B, int32_t h, int32_t w, int32_t dtype16, void* out16, void* stream) {
    if (!img || !out16 || B <= 0 || h <= 0 || w <= 0 || h > 1024 || w > 1024) {
        msam_set_error("msam_patchify_u8: bad arguments (need 0 < h,w <= 1024)");
        return 1;
    }
    hipLaunchKernelGGL(patchify_u8_kernel, dim3(grid_for((long)B * 4096 * 96)), dim3(256), 0, (hipStream_t)stream, img,
                       B, h, w, (u16*)out16, dtype16, 0);
    return msam_check_launch("msam_patchify_u8");
}
extern "C" int msam_patchify_u8_split16(const uint8_t* img, int32_t B, int32_t h, int32_t w, int32_t dtype16, void* out16, void* stream) {
    if (!img || !out16 || B <= 0 || h <= 0 || w <= 0 || h > 1024 || w > 1024) {
        msam_set_error("msam_patchify_u8_split16: bad arguments (need 0 < h,w <= 1024)");
        return 1;
    }
    hipLaunchKernelGGL(patchify_u8_kernel, dim3(grid_for((long)B * 4096 * 96)), dim3(256), 0, (hipStream_t)stream, img,
                       B, h, w, (u16*)out16, dtype16, 1);
    return msam_check_launch("msam_patchify_u8_split16");
}
extern "C" int msam_patchify_u8(const uint8_t* img, int32_t B, int32_t h, int32_t w, void* out_bf16, void* stream) {
    return msam_patchify_u8_16(img, B, h, w, MSAM_BF16, out_bf16, stream);
}

extern "C" int msam_im2col3x3(const void* x_bf16, int32_t B, int32_t C, void* out_bf16, void* stream) {
    if (!x_bf16 || !out_bf16 || B <= 0 || C <= 0 || C % 8) { msam_set_error("msam_im2col3x3: bad arguments"); return 1; }
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for((long)B * 4096 * 9 * C / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)x_bf16, B, C, (u16*)out_bf16);
    return msam_check_launch("msam_im2col3x3");
}

extern "C" int msam_im2col3x3_split16(const float* x_f32, int32_t B, int32_t C, int32_t dtype16, void* out16, void* stream) {
    if (!x_f32 || !out16 || B <= 0 || C <= 0 || C % 8) { msam_set_error("msam_im2col3x3_split16: bad arguments"); return 1; }
    hipLaunchKernelGGL(im2col3x3_split_kernel, dim3(grid_for((long)B * 4096 * 9 * C / 8)), dim3(256), 0, (hipStream_t)stream,
                       x_f32, B, C, (u16*)out16, dtype16);
    return msam_check_launch("msam_im2col3x3_split16");
}

extern "C" int msam_cast_f32_split16(const float* x, int32_t dtype16, void* out16, int64_t rows, int32_t dim, void* stream) {
    if (!x || !out16 || rows <= 0 || dim <= 0 || dim % 8) { msam_set_error("msam_cast_f32_split16: bad arguments (dim % 8 == 0)"); return 1; }
    hipLaunchKernelGGL(cast_split_kernel, dim3(grid_for(rows * (dim / 8))), dim3(256), 0, (hipStream_t)stream, x, (u16*)out16,
                       (long)rows, dim, dtype16);
    return msam_check_launch("msam_cast_f32_split16");
}

extern "C" int msam_cast_f32_to_16(const float* x, int32_t dtype16, void* out16, int64_t n, void* stream) {
    if (!x || !out16 || n <= 0) { msam_set_error("msam_cast_f32_to_16: bad arguments"); return 1; }
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, (u16*)out16, (long)n, dtype16);
    return msam_check_launch("msam_cast_f32_to_16");
}
extern "C" int msam_cast_f32_to_bf16(const float* x, void* out_bf16, int64_t n, void* stream) {
    return msam_cast_f32_to_16(x, MSAM_BF16, out_bf16, n, stream);
}
