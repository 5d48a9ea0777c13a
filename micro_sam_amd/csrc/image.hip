// util._to_image on the device (reference micro_sam/util.py:618-651): any [H,W] / [H,W,C] tile -> uint8 RGB with the
// reference's per-channel min-max normalisation, bit for bit:
//     x = float32(v);  x -= min_c;  x /= (max_c(x - min_c) + 1e-7f);  out = uint8(x * 255)        (all float32, IEEE division)
// gray -> replicated, 2 channels -> third channel of zeros, > 3 channels -> first three.  The raw tile crosses PCIe (1 MB for a
// uint8 1024^2 tile instead of the 3 MB RGB copy) and the conversion costs two HBM passes instead of ~15 ms of numpy.
#include "common.h"
#include "../../include/msam_hip.h"

void msam_set_error(const char* msg);
int msam_check_launch(const char* what);

namespace {

MSAM_DEVINL uint32_t f2ord(float f) {            // order-preserving map float -> uint32 (for atomicMin / atomicMax)
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
MSAM_DEVINL float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

template <typename T>
__global__ __launch_bounds__(256) void minmax_kernel(const T* __restrict__ in, long npix, int C, int nch, uint32_t* __restrict__ mm) {
    // mm[c] = ordered min, mm[4 + c] = ordered max of channel c < nch
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        for (int c = 0; c < nch; ++c) {
            const float v = (float)in[i * C + c];
            mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v);
        }
    }
    for (int c = 0; c < nch; ++c) {
        float a = mn[c], b = mx[c];
        for (int o = 32; o > 0; o >>= 1) { a = fminf(a, __shfl_xor(a, o)); b = fmaxf(b, __shfl_xor(b, o)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(mm + c, f2ord(a)); atomicMax(mm + 4 + c, f2ord(b)); }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void to_image_kernel(const T* __restrict__ in, long npix, int C, int nch,
                                                       const uint32_t* __restrict__ mm, uint8_t* __restrict__ out) {
    float mn[3], den[3];
    for (int c = 0; c < 3; ++c) {
        const int sc = nch == 1 ? 0 : c;
        if (sc < nch) {
            mn[c] = ord2f(mm[sc]);
            den[c] = (ord2f(mm[4 + sc]) - mn[c]) + 1e-7f;          // max of (x - min) == max - min (subtraction is monotonic)
        } else { mn[c] = 0.f; den[c] = 1e-7f; }
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
        uint8_t o[3];
        for (int c = 0; c < 3; ++c) {
            const int sc = nch == 1 ? 0 : c;
            const float v = sc < nch ? (float)in[i * C + sc] : 0.f;
            o[c] = (uint8_t)(((v - mn[c]) / den[c]) * 255.f);
        }
        out[i * 3] = o[0]; out[i * 3 + 1] = o[1]; out[i * 3 + 2] = o[2];
    }
}

// ---- Pillow's BILINEAR resize of an 8-bit image (ResizeLongestSide.apply_image, reference micro_sam/util.py:663; libImaging/Resample.c):
// one pass along one axis - out[o] = clip8((2^21 + sum_t in[first[o] + t] * coef[o][t]) >> 22) with the fixed-point tables of
// micro_sam_amd/transforms.py pil_bilinear_tables (computed exactly as Pillow computes them; tests/test_resize_host.py pins tables and
// passes to Pillow itself).  The caller runs the horizontal pass into an 8-bit intermediate image and then the vertical pass, as
// Pillow does.  in: uint8 [B, H, W, C]; axis 1 = along W (out [B, H, n_out, C]), axis 0 = along H (out [B, n_out, W, C]).
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ in, int B, int H, int W, int C, int axis, int n_out,
                                                          const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize,
                                                          uint8_t* __restrict__ out) {
    const int oh = axis == 0 ? n_out : H, ow = axis == 1 ? n_out : W;
    const long total = (long)B * oh * ow * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long r = i / C;
        const int x = (int)(r % ow); r /= ow;
        const int y = (int)(r % oh);
        const int b = (int)(r / oh);
        const int o = axis == 1 ? x : y;
        const int first = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = coefs + (long)o * ksize;
        int acc = 1 << 21;
        if (axis == 1) {
            const uint8_t* src = in + (((long)b * H + y) * W + first) * C + c;
            for (int t = 0; t < n; ++t) acc += (int)src[(long)t * C] * k[t];
        } else {
            const uint8_t* src = in + (((long)b * H + first) * W + x) * C + c;
            for (int t = 0; t < n; ++t) acc += (int)src[(long)t * W * C] * k[t];
        }
        const int v = acc >> 22;
        out[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

template <typename T>
int run(const void* in, long npix, int C, uint8_t* out, uint32_t* mm, hipStream_t s) {
    const int nch = C > 3 ? 3 : C;
    if (hipMemsetAsync(mm, 0xFF, 16, s) != hipSuccess || hipMemsetAsync(mm + 4, 0x00, 16, s) != hipSuccess) {
        msam_set_error("msam_to_image: memset failed");
        return 2;
    }
    const int grid = (int)((npix + 255) / 256 < 2048 ? (npix + 255) / 256 : 2048);
    hipLaunchKernelGGL(minmax_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)in, npix, C, nch, mm);
    hipLaunchKernelGGL(to_image_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)in, npix, C, nch, mm, out);
    return msam_check_launch("msam_to_image");
}

}  // namespace

extern "C" int msam_to_image(const void* in, int32_t in_dtype, int32_t H, int32_t W, int32_t C, uint8_t* out, void* workspace,
                             void* stream) {
    if (!in || !out || !workspace || H <= 0 || W <= 0 || C <= 0) { msam_set_error("msam_to_image: bad argument"); return 1; }
    const long npix = (long)H * W;
    hipStream_t s = (hipStream_t)stream;
    uint32_t* mm = (uint32_t*)workspace;
    switch (in_dtype) {
        case MSAM_U8: return run<uint8_t>(in, npix, C, out, mm, s);
        case MSAM_U16: return run<uint16_t>(in, npix, C, out, mm, s);
        case MSAM_F32: return run<float>(in, npix, C, out, mm, s);
        default: msam_set_error("msam_to_image: input dtype must be MSAM_U8, MSAM_U16 or MSAM_F32"); return 1;
    }
}

extern "C" int msam_resample_u8(const uint8_t* in, int32_t B, int32_t H, int32_t W, int32_t C, int32_t axis, int32_t n_out,
                                const int32_t* bounds, const int32_t* coefs, int32_t ksize, uint8_t* out, void* stream) {
    if (!in || !out || !bounds || !coefs || B <= 0 || H <= 0 || W <= 0 || C <= 0 || n_out <= 0 || ksize <= 0 || (axis != 0 && axis != 1)) {
        msam_set_error("msam_resample_u8: bad argument");
        return 1;
    }
    const long total = (long)B * (axis == 0 ? n_out : H) * (axis == 1 ? n_out : W) * C;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, axis, n_out, bounds, coefs, ksize, out);
    return msam_check_launch("msam_resample_u8");
}
