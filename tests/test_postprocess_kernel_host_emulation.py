"""postprocess_kernel (csrc/postprocess.hip: Sam.postprocess_masks + stability counts + threshold + boxes + bit packing) executed on the
CPU: the kernel SOURCE compiles as host C++ behind a shim (256 host threads per workgroup, barriers, wave collectives through exchange
buffers) and must reproduce torch's CPU bilinear resampling BIT FOR BIT - the x4 path (1024 x 1024 images, with and without the
"decided words" shortcut) and the general two-pass path with its row caches (any other image size).  The same comparison runs on the
device in tests/test_gpu_postprocess.py; this one pins the integer stage without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "micro_sam_amd", "csrc", "postprocess.hip")

SHIM = r"""
#include <algorithm>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
using std::min; using std::max;
struct idx3 { int x, y, z; };
static thread_local idx3 threadIdx, blockIdx;
static std::barrier<>* wave_bar[4];
static std::barrier<>* block_bar;
static long long wave_buf[4][64];
static std::mutex atomic_mutex;
static long long wave_xchg(long long v, int src_lane_xor) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = v;
    wave_bar[w]->arrive_and_wait();
    const long long r = wave_buf[w][l ^ src_lane_xor];
    wave_bar[w]->arrive_and_wait();
    return r;
}
static int __shfl_xor(int v, int o) { return (int)wave_xchg(v, o); }
static unsigned long long __ballot(int pred) {                 // every lane of the wave takes part (the tests keep whole waves active)
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = pred ? 1 : 0;
    wave_bar[w]->arrive_and_wait();
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= (unsigned long long)(wave_buf[w][i] != 0) << i;
    wave_bar[w]->arrive_and_wait();
    return m;
}
static float __fadd_rn(float a, float b) { return a + b; }
static float __fsub_rn(float a, float b) { return a - b; }
static float __fmul_rn(float a, float b) { return a * b; }
static float __fmaf_rn(float a, float b, float c) { return std::fmaf(a, b, c); }
static uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31)); }
static uint32_t __brev(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i); return r; }
static int __popc(uint32_t x) { return __builtin_popcount(x); }
static int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static int __ffs(uint32_t x) { return __builtin_ffs((int)x); }
static int __ffsll(long long x) { return __builtin_ffsll(x); }
static int __clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }
static void atomicAdd(int* p, int v) { std::lock_guard<std::mutex> g(atomic_mutex); *p += v; }
static void atomicMin(int* p, int v) { std::lock_guard<std::mutex> g(atomic_mutex); *p = std::min(*p, v); }
static void atomicMax(int* p, int v) { std::lock_guard<std::mutex> g(atomic_mutex); *p = std::max(*p, v); }
#define __syncthreads() block_bar->arrive_and_wait()
#define MSAM_DEVINL static inline
static float h2f(uint32_t b) {                                // IEEE half -> float (csrc/common.h h2f), bit by bit
    const uint32_t sg = (b >> 15) & 1u, e = (b >> 10) & 31u, m = b & 1023u;
    float v = e == 0 ? std::ldexp((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : std::ldexp((float)(m | 1024u), (int)e - 25);
    return sg ? -v : v;
}
#define __global__
#define __launch_bounds__(n)
#define __restrict__
#define __shared__ static
%s
template <bool TS, bool LG> static void run(const float* low, int in_h, int in_w, int out_h, int out_w, float thr, float off, int* counts,
                                             int* boxes, uint32_t* bits, float* logits, int N) {
    for (int n = 0; n < N; ++n) for (int bx = 0; bx < (out_w + 255) / 256; ++bx) {
        std::barrier<> b0(64), b1(64), b2(64), b3(64), bb(256);
        wave_bar[0] = &b0; wave_bar[1] = &b1; wave_bar[2] = &b2; wave_bar[3] = &b3; block_bar = &bb;
        std::vector<std::thread> ts;
        for (int tx = 0; tx < 256; ++tx)
            ts.emplace_back([=] {
                threadIdx = {tx, 0, 0}; blockIdx = {bx, n, 0};
                postprocess_kernel<TS, LG, float>(low, in_h, in_w, out_h, out_w, thr, off, counts, boxes, bits, logits);
            });
        for (auto& t : ts) t.join();
    }
}
extern "C" void emu_postprocess(int two_stage, int want_logits, const float* low, int in_h, int in_w, int out_h, int out_w, float thr,
                                float off, int* counts, int* boxes, uint32_t* bits, float* logits, int N) {
    if (two_stage) { if (want_logits) run<true, true>(low, in_h, in_w, out_h, out_w, thr, off, counts, boxes, bits, logits, N);
                     else run<true, false>(low, in_h, in_w, out_h, out_w, thr, off, counts, boxes, bits, logits, N); }
    else { if (want_logits) run<false, true>(low, in_h, in_w, out_h, out_w, thr, off, counts, boxes, bits, logits, N);
           else run<false, false>(low, in_h, in_w, out_h, out_w, thr, off, counts, boxes, bits, logits, N); }
}
"""


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    text = open(SRC).read()
    start = text.index("struct Axis {")
    end = text.index("// ---------------------------------------------------------------------------------------------- RLE")
    body = text[start:end]
    assert "postprocess_kernel" in body and "init_stats_kernel" in body
    d = tmp_path_factory.mktemp("emu_pp")
    cpp, so = os.path.join(d, "pp.cpp"), os.path.join(d, "pp.so")
    open(cpp, "w").write(SHIM % body)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-pthread", "-Wno-unknown-pragmas", cpp, "-o", so])
    lib = ctypes.CDLL(so)
    lib.emu_postprocess.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _torch_reference(low, in_hw, out_hw):
    """Sam.postprocess_masks on the CPU (two torch bilinear interpolations, crop in between)."""
    x = F.interpolate(low[None], (1024, 1024), mode="bilinear", align_corners=False)
    x = x[..., : in_hw[0], : in_hw[1]]
    return F.interpolate(x, out_hw, mode="bilinear", align_corners=False)[0]


def _axis(n_out, scale, n_in):
    """torch's area_pixel_compute_source_index with SEPARATELY rounded fp32 operations (the kernel's contract, csrc/postprocess.hip)."""
    f = np.float32
    src = (f(scale) * (np.arange(n_out, dtype=f) + f(0.5))).astype(f) - f(0.5)
    src = np.maximum(src, f(0)).astype(f)
    i0 = src.astype(np.int64)
    i1 = np.minimum(i0 + 1, n_in - 1)
    w1 = (src - i0.astype(f)).astype(f)
    return i0, i1, (f(1) - w1).astype(f), w1


def _lerp(w0, p0, w1, p1):
    """fma(w0, p0, w1 * p1) in fp32: the product w1 * p1 rounded to fp32, the fused multiply-add through float64 (exact product)."""
    t = (w1 * p1).astype(np.float32)
    return (w0.astype(np.float64) * p0.astype(np.float64) + t.astype(np.float64)).astype(np.float32)


def _resize(x, out_h, out_w, sy, sx):
    i0, i1, wy0, wy1 = _axis(out_h, sy, x.shape[1])
    j0, j1, wx0, wx1 = _axis(out_w, sx, x.shape[2])
    top = _lerp(wx0[None, None, :], x[:, i0][:, :, j0], wx1[None, None, :], x[:, i0][:, :, j1])          # inner axis first
    bot = _lerp(wx0[None, None, :], x[:, i1][:, :, j0], wx1[None, None, :], x[:, i1][:, :, j1])
    return _lerp(wy0[None, :, None], top, wy1[None, :, None], bot)


def _reference(low, in_hw, out_hw):
    """The two resamplings restated in numpy fp32.  For scales that are exact in fp32 (x4, identity, x 1/2) this IS torch's CPU result bit
    for bit (asserted below); for other scales torch's CPU kernels differ between hosts in the last bit of the source index (builds
    that contract ``scale * (dst + 0.5) - 0.5`` into one fused operation: the AVX-512 path of this container does, the GPU box's host
    does not - tests/test_gpu_postprocess.py passes there), so the separately rounded form the kernel documents is the reference here and
    torch must agree within one rounding of the weights."""
    f = np.float32
    x = _resize(low.numpy().astype(f), 1024, 1024, f(0.25), f(0.25))[:, : in_hw[0], : in_hw[1]]
    y = _resize(x, out_hw[0], out_hw[1], f(in_hw[0]) / f(out_hw[0]), f(in_hw[1]) / f(out_hw[1]))
    t = _torch_reference(low, in_hw, out_hw).numpy()
    exact_scale = all((i * 1024) % o == 0 and ((i * 1024) // o) & ((i * 1024) // o - 1) == 0 for i, o in zip(in_hw, out_hw))
    if exact_scale:
        assert np.array_equal(y, t)
    else:
        assert np.abs(y - t).max() <= 2e-5 * max(1.0, float(np.abs(y).max())) and (y != t).mean() < 0.02      # (one rounding of a weight x the logit scale)
    return torch.from_numpy(y)


def _run(emu, low, in_hw, out_hw, want_logits):
    N = low.shape[0]
    H, W = out_hw
    two_stage = not (tuple(in_hw) == (1024, 1024) and tuple(out_hw) == (1024, 1024))
    lo = low.numpy().astype(np.float32).copy()
    counts = np.zeros((N, 3), np.int32)
    boxes = np.tile(np.array([0x7fffffff, 0x7fffffff, -1, -1], np.int32), (N, 1))           # init_stats_kernel's values
    bits = np.zeros((N, (H + 31) // 32, W), np.uint32)
    logits = np.full((N, H, W), np.nan, np.float32) if want_logits else np.zeros(1, np.float32)
    emu.emu_postprocess(int(two_stage), int(want_logits), _ptr(lo), in_hw[0], in_hw[1], H, W, 0.0, 1.0, _ptr(counts), _ptr(boxes), _ptr(bits),
                        _ptr(logits), N)
    return counts, boxes, bits, logits


def _check(emu, low, in_hw, out_hw, want_logits):
    ref = _reference(low, in_hw, out_hw)
    counts, boxes, bits, logits = _run(emu, low, in_hw, out_hw, want_logits)
    if want_logits:
        assert np.array_equal(logits, ref.numpy()), "bilinear resampling must be bit-exact"
    m = (ref > 0.0).numpy()
    H = out_hw[0]
    unpacked = ((bits[:, :, None, :] >> np.arange(32, dtype=np.uint32)[None, None, :, None]) & 1).reshape(bits.shape[0], -1, bits.shape[2])[:, :H]
    assert np.array_equal(unpacked.astype(bool), m)
    want = np.stack([(ref > 1.0).sum((1, 2)).numpy(), (ref > -1.0).sum((1, 2)).numpy(), m.sum((1, 2))], 1)
    assert np.array_equal(counts, want)
    for n in range(m.shape[0]):
        ys, xs = np.nonzero(m[n])
        if len(ys):                                   # (empty masks: finalize_boxes_kernel turns the init values into zeros)
            assert boxes[n].tolist() == [xs.min(), ys.min(), xs.max(), ys.max()]
        else:
            assert boxes[n].tolist() == [0x7fffffff, 0x7fffffff, -1, -1]


@pytest.fixture(scope="module")
def low_res():
    g = torch.Generator().manual_seed(6)
    low = torch.randn(3, 256, 256, generator=g) * 3
    low = F.avg_pool2d(low[None], 5, stride=1, padding=2)[0] * 4
    low[2] = -5.0                                     # an empty mask
    return low


@pytest.mark.parametrize("in_hw,out_hw", [((1024, 768), (1024, 768)), ((1024, 1024), (512, 512)), ((683, 1024), (400, 600)), ((1024, 640), (1500, 938))])
def test_general_resampling_path_source_on_the_cpu(emu, low_res, in_hw, out_hw):
    """Identity, x 1/2, a non-integer reduction and an enlargement: logits bit for bit, then bits / counts / boxes."""
    n = 1 if out_hw[0] * out_hw[1] > 600_000 else 3
    _check(emu, low_res[:n], in_hw, out_hw, want_logits=True)
    _check(emu, low_res[:1], in_hw, out_hw, want_logits=False)


def test_x4_path_source_on_the_cpu(emu, low_res):
    _check(emu, low_res[:1], (1024, 1024), (1024, 1024), want_logits=True)


def _object_like_logits():
    g = torch.Generator().manual_seed(16)
    yy, xx = torch.meshgrid(torch.arange(256.0), torch.arange(256.0), indexing="ij")
    low = torch.empty(3, 256, 256)
    for i in range(3):
        f = torch.zeros(256, 256)
        for _ in range(2 + 3 * i):
            cy, cx, r = (torch.rand(3, generator=g) * torch.tensor([256.0, 256.0, 30.0]) + torch.tensor([0.0, 0.0, 4.0])).tolist()
            f = torch.maximum(f, torch.clamp(1.5 - torch.sqrt((yy - cy) ** 2 + (xx - cx) ** 2) / r, 0.0, 1.0))
        low[i] = f * 45.0 - 14.0 + torch.randn(256, 256, generator=g) * 0.3
    low[1] = low[1].clamp(-0.995, 0.995)
    low[2, :128] = 1.005
    return low


def test_x4_path_decided_words_source_on_the_cpu(emu):
    """Object-like logits: most 32-row words are decided from the range of their ten low-res rows; a plateau inside the +-1 band and a
    region within 0.01 of a threshold must take the exact path - bits and counts equal the per-pixel reference either way."""
    _check(emu, _object_like_logits(), (1024, 1024), (1024, 1024), want_logits=False)


@pytest.mark.parametrize("in_hw,out_hw", [((1024, 1024), (896, 896)), ((731, 1024), (640, 896)), ((1024, 683), (300, 200))])
def test_general_path_decided_words_source_on_the_cpu(emu, in_hw, out_hw):
    """The same for the two-stage path (round 5: a word is decided from the range of the low-res rows / columns under it): the border
    tiles of a tiled volume (896, 640 pixels long) and a reduction whose last word is partial and whose last workgroup has idle lanes."""
    _check(emu, _object_like_logits(), in_hw, out_hw, want_logits=False)
