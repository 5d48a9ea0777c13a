"""The rel-pos attention training kernels of csrc/train.hip, executed on the CPU: their threads are independent (no cross-lane
operations, a per-thread LDS column), so the kernel SOURCE compiles as host C++ behind a few macros and runs thread by thread.
This checks the code that will run on the GPU - indexing, online softmax, the backward formulas - against torch autograd before
its first GPU run (tests/test_gpu_training_encoders.py checks the same on the device)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "micro_sam_amd", "csrc", "train.hip")

SHIM = r"""
#include <cmath>
#include <cstdint>
struct idx3 { int x, y, z; };
static idx3 threadIdx, blockIdx;
#define __global__
#define __launch_bounds__(n)
#define __restrict__
#define __shared__ static
#define __expf expf
#define __logf logf
namespace {
%s
}
template <int D> static void run_fwd(const float* q, const float* k, const float* v, const float* bh, const float* bw, int BH, int Gh,
                                     int Gw, float scale, float* out, float* lse) {
    for (int by = 0; by < BH; ++by) for (int bx = 0; bx < (Gh * Gw + 127) / 128; ++bx) for (int tx = 0; tx < 128; ++tx) {
        blockIdx = {bx, by, 0}; threadIdx = {tx, 0, 0};
        relpos_fwd_kernel<D>(q, k, v, bh, bw, Gh, Gw, scale, out, lse);
    }
}
template <int D> static void run_bwd(const float* q, const float* k, const float* v, const float* bh, const float* bw, const float* out,
                                     const float* dout, const float* lse, int BH, int Gh, int Gw, float scale, float* dq, float* dk,
                                     float* dv, float* dbh, float* dbw, float* delta) {
    for (int by = 0; by < BH; ++by) for (int bx = 0; bx < (Gh * Gw + 127) / 128; ++bx) for (int tx = 0; tx < 128; ++tx) {
        blockIdx = {bx, by, 0}; threadIdx = {tx, 0, 0};
        relpos_bwd_q_kernel<D>(q, k, v, bh, bw, out, dout, lse, Gh, Gw, scale, dq, dbh, dbw, delta);
    }
    for (int by = 0; by < BH; ++by) for (int bx = 0; bx < (Gh * Gw + 127) / 128; ++bx) for (int tx = 0; tx < 128; ++tx) {
        blockIdx = {bx, by, 0}; threadIdx = {tx, 0, 0};
        relpos_bwd_kv_kernel<D>(q, k, v, bh, bw, dout, lse, delta, Gh, Gw, scale, dk, dv);
    }
}
extern "C" void emu_fwd(int D, const float* q, const float* k, const float* v, const float* bh, const float* bw, int BH, int Gh, int Gw,
                        float scale, float* out, float* lse) {
    if (D == 64) run_fwd<64>(q, k, v, bh, bw, BH, Gh, Gw, scale, out, lse); else run_fwd<80>(q, k, v, bh, bw, BH, Gh, Gw, scale, out, lse);
}
extern "C" void emu_bwd(int D, const float* q, const float* k, const float* v, const float* bh, const float* bw, const float* out,
                        const float* dout, const float* lse, int BH, int Gh, int Gw, float scale, float* dq, float* dk, float* dv,
                        float* dbh, float* dbw, float* delta) {
    if (D == 64) run_bwd<64>(q, k, v, bh, bw, out, dout, lse, BH, Gh, Gw, scale, dq, dk, dv, dbh, dbw, delta);
    else run_bwd<80>(q, k, v, bh, bw, out, dout, lse, BH, Gh, Gw, scale, dq, dk, dv, dbh, dbw, delta);
}
"""


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    text = open(SRC).read()
    start = text.index("template <int D>\n__global__ __launch_bounds__(128) void relpos_fwd_kernel")
    end = text.index("// ---- cast + transpose + column sums", start)
    kernels = text[start:end]
    assert all(f"relpos_{n}_kernel" in kernels for n in ("fwd", "bwd_q", "bwd_kv")) and "wave_sum64" not in kernels
    d = tmp_path_factory.mktemp("emu")
    cpp, so = os.path.join(d, "emu.cpp"), os.path.join(d, "emu.so")
    open(cpp, "w").write(SHIM % kernels)
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-Wno-unknown-pragmas", cpp, "-o", so])
    return ctypes.CDLL(so)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("BH,Gh,Gw,D", [(3, 14, 14, 64), (2, 5, 9, 64), (2, 14, 14, 80), (1, 12, 11, 64), (1, 3, 64, 64)])
def test_relpos_kernels_source_on_the_cpu(emu, BH, Gh, Gw, D):
    g = torch.Generator().manual_seed(BH + Gh * Gw + D)
    N = Gh * Gw
    q, k, v = (torch.randn(BH, N, D, generator=g) for _ in range(3))
    bh, bw = torch.randn(BH, N, Gh, generator=g), torch.randn(BH, N, Gw, generator=g)
    dout = torch.randn(BH, N, D, generator=g)
    scale = D ** -0.5
    arrs = [t.numpy().astype(np.float32).copy() for t in (q, k, v, bh, bw)]
    out, lse = np.zeros((BH, N, D), np.float32), np.zeros((BH, N), np.float32)
    emu.emu_fwd(D, *map(_ptr, arrs), BH, Gh, Gw, ctypes.c_float(scale), _ptr(out), _ptr(lse))
    ref_in = [t.double().clone().requires_grad_() for t in (q, k, v, bh, bw)]
    s = (ref_in[0] * scale) @ ref_in[1].transpose(1, 2)
    s = (s.view(BH, N, Gh, Gw) + ref_in[3][:, :, :, None] + ref_in[4][:, :, None, :]).view(BH, N, N)
    ref = s.softmax(dim=-1) @ ref_in[2]
    assert np.abs(out - ref.detach().numpy()).max() <= 1e-4 * ref.abs().max().item()
    assert np.abs(lse - torch.logsumexp(s, dim=-1).detach().numpy()).max() <= 1e-4
    ref.backward(dout.double())
    do = dout.numpy().astype(np.float32).copy()
    dq, dk, dv = (np.full((BH, N, D), np.nan, np.float32) for _ in range(3))
    dbh, dbw, delta = np.full((BH, N, Gh), np.nan, np.float32), np.full((BH, N, Gw), np.nan, np.float32), np.zeros((BH, N), np.float32)
    emu.emu_bwd(D, *map(_ptr, arrs), _ptr(out), _ptr(do), _ptr(lse), BH, Gh, Gw, ctypes.c_float(scale), _ptr(dq), _ptr(dk), _ptr(dv),
                _ptr(dbh), _ptr(dbw), _ptr(delta))
    for name, got, want in zip(("dq", "dk", "dv", "dbias_h", "dbias_w"), (dq, dk, dv, dbh, dbw), ref_in):
        w = want.grad.numpy()
        assert np.isfinite(got).all(), name                                                # every element was written
        assert np.abs(got - w).max() <= 2e-4 * np.abs(w).max(), name


# ---------------------------------------------------------------------------------------------------------------
# layernorm_bwd_kernel<V> at the image encoder's widths: the kernel uses wave-wide sums and one __syncthreads, so a workgroup
# is emulated with 256 host threads (4 waves) and barriers.
# ---------------------------------------------------------------------------------------------------------------

LN_SHIM = r"""
#include <barrier>
#include <cmath>
#include <mutex>
#include <thread>
#include <vector>
struct idx3 { int x, y, z; };
static thread_local idx3 threadIdx, blockIdx;
static idx3 gridDim;
static std::barrier<>* wave_bar[4];
static std::barrier<>* block_bar;
static float wave_buf[4][64];
static std::mutex atomic_mutex;
static float wave_sum64(float v) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = v;
    wave_bar[w]->arrive_and_wait();
    float s = 0.f;
    for (int i = 0; i < 64; ++i) s += wave_buf[w][i];
    wave_bar[w]->arrive_and_wait();
    return s;
}
static void atomicAdd(float* p, float v) { std::lock_guard<std::mutex> g(atomic_mutex); *p += v; }
#define __syncthreads() block_bar->arrive_and_wait()
#define __global__
#define __launch_bounds__(n)
#define __restrict__
#define __shared__ static
%s
template <int V> static void run(const float* x, const float* w, const float* dy, float eps, long rows, float* dx, float* dw, float* db,
                                 int grid) {
    gridDim = {grid, 1, 1};
    constexpr int DIM = V * 64;
    std::vector<float> part((size_t)grid * 2 * DIM, NAN);          // the kernel's per-workgroup sums (round 6: no atomics)
    float* pp = part.data();
    for (int bx = 0; bx < grid; ++bx) {
        std::barrier<> b0(64), b1(64), b2(64), b3(64), bb(256);
        wave_bar[0] = &b0; wave_bar[1] = &b1; wave_bar[2] = &b2; wave_bar[3] = &b3; block_bar = &bb;
        std::vector<std::thread> ts;
        for (int tx = 0; tx < 256; ++tx)
            ts.emplace_back([=] { threadIdx = {tx, 0, 0}; blockIdx = {bx, 0, 0}; layernorm_bwd_kernel<V>(x, w, dy, eps, rows, dx, pp); });
        for (auto& t : ts) t.join();
    }
    for (int c = 0; c < DIM; ++c)                                    // ln_bwd_reduce_kernel: workgroup order
        for (int bx = 0; bx < grid; ++bx) { dw[c] += part[((size_t)bx * 2) * DIM + c]; db[c] += part[((size_t)bx * 2 + 1) * DIM + c]; }
}
extern "C" int emu_ln_bwd(int dim, const float* x, const float* w, const float* dy, float eps, long rows, float* dx, float* dw, float* db,
                          int grid) {
    switch (dim) {
        case 256: run<4>(x, w, dy, eps, rows, dx, dw, db, grid); return 0;
        case 768: run<12>(x, w, dy, eps, rows, dx, dw, db, grid); return 0;
        case 1024: run<16>(x, w, dy, eps, rows, dx, dw, db, grid); return 0;
        case 1280: run<20>(x, w, dy, eps, rows, dx, dw, db, grid); return 0;
    }
    return 1;
}
"""


@pytest.fixture(scope="module")
def emu_ln(tmp_path_factory):
    text = open(SRC).read()
    start = text.index("template <int V>   // V = dim / 64 values per lane")
    end = text.index("// ---- attention, one thread per query row", start)
    kernel = text[start:end]
    assert "layernorm_bwd_kernel" in kernel and "wave_sum64" in kernel
    d = tmp_path_factory.mktemp("emu_ln")
    cpp, so = os.path.join(d, "ln.cpp"), os.path.join(d, "ln.so")
    open(cpp, "w").write(LN_SHIM % kernel)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-shared", "-fPIC", "-pthread", "-Wno-unknown-pragmas", cpp, "-o", so])
    lib = ctypes.CDLL(so)
    lib.emu_ln_bwd.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_long,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return lib


@pytest.mark.parametrize("dim", [256, 768, 1024, 1280])
def test_layernorm_backward_kernel_source_on_the_cpu(emu_ln, dim):
    """rows = 11 on a grid of 2 workgroups: waves take rows 0..3 / 4..7 and then loop (rows 8, 9, 10), one wave idles at the end."""
    g = torch.Generator().manual_seed(dim)
    rows = 11
    x = (torch.randn(rows, dim, generator=g) * 2 + 0.5).requires_grad_()
    w = (torch.randn(dim, generator=g) * 0.2 + 1).requires_grad_()
    b = torch.randn(dim, generator=g).requires_grad_()
    dy = torch.randn(rows, dim, generator=g)
    torch.nn.functional.layer_norm(x, (dim,), w, b, 1e-6).backward(dy)
    xa, wa, dya = (t.detach().numpy().astype(np.float32).copy() for t in (x, w, dy))
    dx = np.full((rows, dim), np.nan, np.float32)
    dw, db = np.zeros(dim, np.float32), np.zeros(dim, np.float32)           # accumulated by the kernel (the caller zeroes them)
    assert emu_ln.emu_ln_bwd(dim, _ptr(xa), _ptr(wa), _ptr(dya), 1e-6, rows, _ptr(dx), _ptr(dw), _ptr(db), 2) == 0
    for name, got, want in (("dx", dx, x.grad), ("dw", dw, w.grad), ("db", db, b.grad)):
        want = want.numpy()
        assert np.isfinite(got).all(), name
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max(), name


# ---------------------------------------------------------------------------------------------------------------
# The decoder's attention when one side is short (workgroup per row of the short side, block reductions): attn_fwd_rowblock /
# attn_bwd_q_rowblock / attn_bwd_kv_rowblock.  Same thread + barrier emulation; __shfl_xor through a per-wave exchange buffer.
# ---------------------------------------------------------------------------------------------------------------

RB_SHIM = r"""
#include <barrier>
#include <cmath>
#include <thread>
#include <vector>
struct idx3 { int x, y, z; };
static thread_local idx3 threadIdx, blockIdx;
static std::barrier<>* wave_bar[4];
static std::barrier<>* block_bar;
static float wave_buf[4][64];
static float wave_sum64(float v) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = v;
    wave_bar[w]->arrive_and_wait();
    float s = 0.f;
    for (int i = 0; i < 64; ++i) s += wave_buf[w][i];
    wave_bar[w]->arrive_and_wait();
    return s;
}
static float __shfl_xor(float v, int o) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = v;
    wave_bar[w]->arrive_and_wait();
    const float r = wave_buf[w][l ^ o];
    wave_bar[w]->arrive_and_wait();
    return r;
}
#define __syncthreads() block_bar->arrive_and_wait()
#define __global__
#define __device__
#define __forceinline__ inline
#define __launch_bounds__(n)
#define __restrict__
#define __shared__ static
#define __expf expf
#define __logf logf
%s
template <class F> static void launch(int gx, int gy, F f) {
    for (int by = 0; by < gy; ++by) for (int bx = 0; bx < gx; ++bx) {
        std::barrier<> b0(64), b1(64), b2(64), b3(64), bb(256);
        wave_bar[0] = &b0; wave_bar[1] = &b1; wave_bar[2] = &b2; wave_bar[3] = &b3; block_bar = &bb;
        std::vector<std::thread> ts;
        for (int tx = 0; tx < 256; ++tx) ts.emplace_back([=] { threadIdx = {tx, 0, 0}; blockIdx = {bx, by, 0}; f(); });
        for (auto& t : ts) t.join();
    }
}
template <int D> static void run(int mode, const float* q, const float* k, const float* v, const float* out_in, const float* dout,
                                 const float* lse_in, const float* delta_in, int BH, int Nq, int Nk, float scale, float* o0, float* o1) {
    if (mode == 0) launch(Nq, BH, [=] { attn_fwd_rowblock_kernel<D>(q, k, v, Nq, Nk, scale, o0, o1); });
    if (mode == 1) launch(Nq, BH, [=] { attn_bwd_q_rowblock_kernel<D>(q, k, v, out_in, dout, lse_in, Nq, Nk, scale, o0, o1); });
    if (mode == 2) launch(Nk, BH, [=] { attn_bwd_kv_rowblock_kernel<D>(q, k, v, dout, lse_in, delta_in, Nq, Nk, scale, o0, o1); });
}
extern "C" void emu_rb(int D, int mode, const float* q, const float* k, const float* v, const float* out_in, const float* dout,
                       const float* lse_in, const float* delta_in, int BH, int Nq, int Nk, float scale, float* o0, float* o1) {
    if (D == 16) run<16>(mode, q, k, v, out_in, dout, lse_in, delta_in, BH, Nq, Nk, scale, o0, o1);
    else run<32>(mode, q, k, v, out_in, dout, lse_in, delta_in, BH, Nq, Nk, scale, o0, o1);
}
"""


@pytest.fixture(scope="module")
def emu_rb(tmp_path_factory):
    text = open(SRC).read()
    start = text.index("__device__ __forceinline__ float block_sum256")
    end = text.index("// ---- attention of the image encoder with its decomposed relative position bias", start)
    kernels = text[start:end]
    assert all(n in kernels for n in ("attn_fwd_rowblock_kernel", "attn_bwd_q_rowblock_kernel", "attn_bwd_kv_rowblock_kernel"))
    d = tmp_path_factory.mktemp("emu_rb")
    cpp, so = os.path.join(d, "rb.cpp"), os.path.join(d, "rb.so")
    open(cpp, "w").write(RB_SHIM % kernels)
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-shared", "-fPIC", "-pthread", "-Wno-unknown-pragmas", cpp, "-o", so])
    return ctypes.CDLL(so)


@pytest.mark.parametrize("D", [16, 32])
def test_rowblock_attention_kernels_source_on_the_cpu(emu_rb, D):
    g = torch.Generator().manual_seed(D)
    scale = ctypes.c_float(D ** -0.5)

    def check(BH, Nq, Nk, modes):
        q, k, v = torch.randn(BH, Nq, D, generator=g), torch.randn(BH, Nk, D, generator=g), torch.randn(BH, Nk, D, generator=g)
        dout = torch.randn(BH, Nq, D, generator=g)
        ref_in = [t.double().clone().requires_grad_() for t in (q, k, v)]
        s = (ref_in[0] * D ** -0.5) @ ref_in[1].transpose(1, 2)
        ref = s.softmax(-1) @ ref_in[2]
        ref.backward(dout.double())
        lse_ref = torch.logsumexp(s, -1).detach().float().numpy().copy()
        out_ref = ref.detach().float().numpy().copy()
        delta_ref = (dout * ref.detach().float()).sum(-1).numpy().copy()
        qa, ka, va, do = (t.numpy().astype(np.float32).copy() for t in (q, k, v, dout))
        if 0 in modes:
            out, lse = np.full((BH, Nq, D), np.nan, np.float32), np.full((BH, Nq), np.nan, np.float32)
            emu_rb.emu_rb(D, 0, _ptr(qa), _ptr(ka), _ptr(va), None, None, None, None, BH, Nq, Nk, scale, _ptr(out), _ptr(lse))
            assert np.abs(out - out_ref).max() <= 1e-4 * np.abs(out_ref).max() and np.abs(lse - lse_ref).max() <= 1e-4
        if 1 in modes:
            dq, delta = np.full((BH, Nq, D), np.nan, np.float32), np.full((BH, Nq), np.nan, np.float32)
            emu_rb.emu_rb(D, 1, _ptr(qa), _ptr(ka), _ptr(va), _ptr(out_ref), _ptr(do), _ptr(lse_ref), None, BH, Nq, Nk, scale, _ptr(dq), _ptr(delta))
            w = ref_in[0].grad.numpy()
            assert np.abs(dq - w).max() <= 2e-4 * np.abs(w).max() and np.abs(delta - delta_ref).max() <= 1e-4 * np.abs(delta_ref).max()
        if 2 in modes:
            dk, dv = np.full((BH, Nk, D), np.nan, np.float32), np.full((BH, Nk, D), np.nan, np.float32)
            emu_rb.emu_rb(D, 2, _ptr(qa), _ptr(ka), _ptr(va), None, _ptr(do), _ptr(lse_ref), _ptr(delta_ref), BH, Nq, Nk, scale, _ptr(dk), _ptr(dv))
            for got, want in ((dk, ref_in[1].grad.numpy()), (dv, ref_in[2].grad.numpy())):
                assert np.isfinite(got).all() and np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
    check(2, 3, 1100, (0, 1))          # few queries, many keys (not a multiple of 256; the last threads see 4 keys, some 5)
    check(1, 2, 200, (0, 1))           # fewer keys than threads: idle threads must drop out of the merge
    check(2, 1100, 3, (2,))            # few keys, many queries


# ---------------------------------------------------------------------------------------------------------------
# cast_transpose_kernel (operands of the weight gradients): 256 host threads per workgroup, barriers, LDS atomics under a mutex.
# ---------------------------------------------------------------------------------------------------------------

CT_SHIM = r"""
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
struct idx3 { int x, y, z; };
static thread_local idx3 threadIdx, blockIdx;
static std::barrier<>* block_bar;
static std::mutex atomic_mutex;
struct uint2 { unsigned x, y; };
struct float4 { float x, y, z, w; };
static float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) {                 // round to nearest even (what the device's (__bf16) conversion does)
    unsigned u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static void atomicAdd(float* p, float v) { std::lock_guard<std::mutex> g(atomic_mutex); *p += v; }
#define __syncthreads() block_bar->arrive_and_wait()
#define __global__
#define __launch_bounds__(n)
#define __restrict__
#define __shared__ static
%s
extern "C" void emu_cast_transpose(int src16, const void* x, long M, int K, long ldx, unsigned short* out16, unsigned short* outT, float* colsum) {
    for (int by = 0; by < (M + 63) / 64; ++by) for (int bx = 0; bx < (K + 63) / 64; ++bx) {
        std::barrier<> bb(256);
        block_bar = &bb;
        std::vector<std::thread> ts;
        for (int tx = 0; tx < 256; ++tx)
            ts.emplace_back([=] {
                threadIdx = {tx, 0, 0}; blockIdx = {bx, by, 0};
                if (src16) cast_transpose_kernel<true>(x, M, K, ldx, out16, outT, colsum);
                else cast_transpose_kernel<false>(x, M, K, ldx, out16, outT, colsum);
            });
        for (auto& t : ts) t.join();
    }
}
"""


@pytest.fixture(scope="module")
def emu_ct(tmp_path_factory):
    text = open(SRC).read()
    start = text.index("template <bool SRC16>\n__global__ __launch_bounds__(256) void cast_transpose_kernel")
    end = text.index("}  // namespace", start)
    d = tmp_path_factory.mktemp("emu_ct")
    cpp, so = os.path.join(d, "ct.cpp"), os.path.join(d, "ct.so")
    open(cpp, "w").write(CT_SHIM % text[start:end])
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-shared", "-fPIC", "-pthread", "-Wno-unknown-pragmas", cpp, "-o", so])
    lib = ctypes.CDLL(so)
    lib.emu_cast_transpose.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p]
    return lib


@pytest.mark.parametrize("M,K,ldx,src16", [(130, 68, 68, False), (64, 128, 132, False), (37, 8, 8, True), (200, 64, 64, True), (129, 256, 256, False)])
def test_cast_transpose_kernel_source_on_the_cpu(emu_ct, M, K, ldx, src16):
    """Ragged M (with and without M % 4 == 0: vector / scalar stores of the transpose), K below and across a tile, a row stride > K."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, ldx, generator=g) * 3
    if src16:
        x = x.to(torch.bfloat16)
        src = x.view(torch.int16).numpy().copy()
    else:
        src = x.numpy().astype(np.float32).copy()
    ref16 = x[:, :K].to(torch.bfloat16)
    out16 = np.full((M, K), 0x7fc0, np.uint16)
    outT = np.full((K, M), 0x7fc0, np.uint16)
    # round 6: the kernel writes per-row-block column sums [ceil(M / 64)][K] (no atomics); the launcher adds the blocks in order (msam_det_reduce)
    parts = np.full(((M + 63) // 64, K), np.nan, np.float32)
    emu_ct.emu_cast_transpose(int(src16), _ptr(src), M, K, ldx, _ptr(out16), _ptr(outT), _ptr(parts))
    cs = np.zeros(K, np.float32)
    for blk in parts:
        cs += blk
    want = ref16.view(torch.int16).numpy().view(np.uint16)
    assert np.isfinite(parts).all() and (out16 == want).all() and (outT == want.T).all()
    assert np.abs(cs - x[:, :K].float().sum(0).numpy()).max() <= 1e-4 * max(1.0, float(x.float().abs().sum(0).max()))
    # outputs are optional
    outT2 = np.zeros((K, M), np.uint16)
    emu_ct.emu_cast_transpose(int(src16), _ptr(src), M, K, ldx, None, _ptr(outT2), None)
    assert (outT2 == want.T).all()
