"""Host execution of HIP kernel SOURCE for the CPU-side tests: a C++ prelude that lets a kernel of csrc/*.hip compile with g++ and run
with one host thread per lane - 256 threads per workgroup, ``__syncthreads`` / wave barriers as std::barrier, wave collectives
(``__shfl_xor``, ``__ballot``, the 16x16x32 and 32x32x16 MFMA of gfx950 in their register layouts) through per-wave exchange buffers.

TEST INFRASTRUCTURE.  It checks indexing, staging, masking and the arithmetic of a kernel before (and independently of) its GPU run; the
MFMA emulation accumulates in double and rounds once, so results agree with the device to accumulation-order accuracy, not bit for bit.

Register layouts (MI355X_MICROARCH / cdna_hip_programming guides; the kernels' own comments):
* v_mfma_f32_16x16x32_{bf16,f16}: lane l holds A[row = l % 16][k = 8 (l / 16) .. + 7], B[k = 8 (l / 16) .. + 7][col = l % 16] and
  D[row = 4 (l / 16) + i][col = l % 16], i = 0..3;
* ds_read_b64_tr_b16 (16-bit transposing LDS read): see ds_read_tr16_b64_emu below - restated from how the device-tested kernels use it;
* v_mfma_f32_32x32x16_{bf16,f16}: lane l holds A[row = l % 32][k = 8 (l / 32) .. + 7], B likewise for col = l % 32, and
  D[row = 8 g + 4 (l / 32) + x][col = l % 32] in register 4 g + x.
"""
import ctypes
import os
import subprocess

PRELUDE = r"""
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
static idx3 gridDim;
static std::barrier<>* wave_bar[8];
static std::barrier<>* block_bar;
static std::mutex atomic_mutex;
typedef unsigned short u16;
typedef float f32x4_t __attribute__((vector_size(16)));
struct f32x2_t {                                             // clang's ext_vector_type(2) as far as the kernels use it (.x / .y, * + +=)
    float x, y;
    f32x2_t operator*(const f32x2_t& o) const { return {x * o.x, y * o.y}; }
    f32x2_t operator+(const f32x2_t& o) const { return {x + o.x, y + o.y}; }
    f32x2_t operator-(const f32x2_t& o) const { return {x - o.x, y - o.y}; }
    f32x2_t operator*(float s) const { return {x * s, y * s}; }
    f32x2_t operator+(float s) const { return {x + s, y + s}; }
    f32x2_t& operator+=(const f32x2_t& o) { x += o.x; y += o.y; return *this; }
};
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
typedef float f32x16_t __attribute__((vector_size(64)));
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct float4 { float x, y, z, w; };
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float bf2f(u16 h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
static inline u16 f2bf(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
    return (u16)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float h16_to_f(u16 h) {                       // IEEE binary16 -> binary32 (exact)
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, ex = (h >> 10) & 31u, man = h & 1023u;
    uint32_t u;
    if (ex == 0) {
        if (man == 0) u = sign;
        else { int e = -1; uint32_t m = man; do { ++e; m <<= 1; } while (!(m & 1024u)); u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 1023u) << 13); }
    } else if (ex == 31) u = sign | 0x7f800000u | (man << 13);
    else u = sign | ((ex + 112u) << 23) | (man << 13);
    float f; std::memcpy(&f, &u, 4); return f;
}
static inline u16 f2h(float f) {                             // binary32 -> binary16, round to nearest even
    uint32_t u; std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (u16)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));
    if (u >= 0x477ff000u) return (u16)(sign | 0x7c00u);      // rounds to infinity
    if (u < 0x33000001u) return (u16)sign;                   // rounds to zero
    const int e = (int)(u >> 23) - 127;
    uint32_t man = (u & 0x7fffffu) | 0x800000u;
    int shift = e >= -14 ? 13 : 13 + (-14 - e);              // subnormal halves lose more bits
    uint32_t half = man >> shift, rem = man & ((1u << shift) - 1u), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1u))) ++half;
    if (e >= -14) return (u16)(sign | (((uint32_t)(e + 15) << 10) + (half - 0x400u)));
    return (u16)(sign | half);
}
static inline float h2f(uint32_t b) { return h16_to_f((u16)b); }
static inline uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
static inline uint32_t pack2h(float lo, float hi) { return (uint32_t)f2h(lo) | ((uint32_t)f2h(hi) << 16); }
static inline int swz(int row) { return ((row >> 1) & 7) ^ (((row + 4) >> 3) & 1); }      // common.h
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
// ---- wave collectives: every lane of the wave takes part
static long long wave_buf[8][64];
static uint4 wave_a[8][64], wave_b[8][64], wave_a2[8][64], wave_b2[8][64];
static inline long long wave_xchg(long long v, int src_lane_xor) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = v;
    wave_bar[w]->arrive_and_wait();
    const long long r = wave_buf[w][l ^ src_lane_xor];
    wave_bar[w]->arrive_and_wait();
    return r;
}
static inline int __shfl_xor(int v, int o) { return (int)wave_xchg(v, o); }
static inline float __shfl_xor(float v, int o) { return __uint_as_float((uint32_t)wave_xchg(__float_as_uint(v), o)); }
static inline unsigned long long __ballot(int pred) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = pred ? 1 : 0;
    wave_bar[w]->arrive_and_wait();
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= (unsigned long long)(wave_buf[w][i] != 0) << i;
    wave_bar[w]->arrive_and_wait();
    return m;
}
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool pred) { return __ballot(pred ? 1 : 0); }
static inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
// ds_read_b64_tr_b16 as the kernels use it: inside every group of 16 lanes, lane r receives element (r % 4) of the four 16-bit values
// read by lanes 4 j + r / 4 (j = 0..3) - a 16 x 4 <-> 4 x 16 transpose of the block the group addressed
static uint2 wave_tr[8][64];
static inline uint2 ds_read_tr16_b64_emu(const unsigned char* p) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    std::memcpy(&wave_tr[w][l], p, 8);
    wave_bar[w]->arrive_and_wait();
    const int g0 = l & ~15, r = l & 15;
    u16 e[4];
    for (int j = 0; j < 4; ++j) {
        const uint2& src = wave_tr[w][g0 + 4 * j + (r >> 2)];
        const uint32_t wd = (r & 2) ? src.y : src.x;
        e[j] = (u16)((r & 1) ? (wd >> 16) : (wd & 0xffffu));
    }
    wave_bar[w]->arrive_and_wait();
    return uint2{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16)};
}
static inline void __builtin_amdgcn_wave_barrier() { wave_bar[threadIdx.x >> 6]->arrive_and_wait(); }
static inline float elem16(const uint4& v, int j, bool f16) {
    const uint32_t wd = (&v.x)[j >> 1];
    const u16 b = (u16)((j & 1) ? (wd >> 16) : (wd & 0xffffu));
    return f16 ? h16_to_f(b) : bf2f(b);
}
static inline f32x4_t mfma16_emu(const uint4& a, const uint4& b, f32x4_t c, bool f16) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_a[w][l] = a; wave_b[w][l] = b;
    wave_bar[w]->arrive_and_wait();
    f32x4_t d;
    for (int i = 0; i < 4; ++i) {
        const int row = (l >> 4) * 4 + i, col = l & 15;
        double s = c[i];
        for (int kb = 0; kb < 4; ++kb)
            for (int j = 0; j < 8; ++j) s += (double)elem16(wave_a[w][kb * 16 + row], j, f16) * (double)elem16(wave_b[w][kb * 16 + col], j, f16);
        d[i] = (float)s;
    }
    wave_bar[w]->arrive_and_wait();
    return d;
}
static inline f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) { return mfma16_emu(a, b, c, false); }
static inline f32x4_t mfma16h(const uint4& a, const uint4& b, f32x4_t c) { return mfma16_emu(a, b, c, true); }
static inline f32x16_t mfma32_emu(const uint4& a, const uint4& b, f32x16_t c, bool f16) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_a[w][l] = a; wave_b[w][l] = b;
    wave_bar[w]->arrive_and_wait();
    f32x16_t d;
    for (int g = 0; g < 4; ++g)
        for (int x = 0; x < 4; ++x) {
            const int row = 8 * g + 4 * (l >> 5) + x, col = l & 31;
            double s = c[4 * g + x];
            for (int kb = 0; kb < 2; ++kb)
                for (int j = 0; j < 8; ++j) s += (double)elem16(wave_a[w][kb * 32 + row], j, f16) * (double)elem16(wave_b[w][kb * 32 + col], j, f16);
            d[4 * g + x] = (float)s;
        }
    wave_bar[w]->arrive_and_wait();
    return d;
}
// LDS-DMA (global_load_lds_dwordx4): lane l lands at the wave-uniform LDS base + 16 l; performed at once
static inline void __builtin_amdgcn_global_load_lds(const void* g, void* l, int size, int, int) {
    std::memcpy((char*)l + (threadIdx.x & 63) * size, g, size);
}
static inline void wait_vmem_all() {}
// raw buffer addressing (common.h make_rsrc / buf_load16 / buf_store16): base + per-lane byte offset + scalar byte offset, reads
// outside the descriptor's range return zeros, writes outside it are dropped
struct rsrc_t { char* base; uint32_t bytes; };
static inline rsrc_t make_rsrc(const void* base, uint32_t bytes) { return rsrc_t{(char*)base, bytes}; }
static inline uint4 buf_load16(rsrc_t r, int voff, int soff) {
    uint4 v{0, 0, 0, 0};
    const long o = (long)voff + soff;
    if (o >= 0 && o + 16 <= (long)r.bytes) std::memcpy(&v, r.base + o, 16);
    return v;
}
static inline void buf_store16(const uint4& v, rsrc_t r, int voff, int soff) {
    const long o = (long)voff + soff;
    if (o >= 0 && o + 16 <= (long)r.bytes) std::memcpy(r.base + o, &v, 16);
}
// v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 e4m3 operands and unit block scales: lane l holds the 32 bytes k = 32 (l / 32) .. + 31 of
// row (A) / column (B) l % 32; D as the 32x32x16 form
static inline float e4m3_to_f(uint8_t b) {
    const int ex = (b >> 3) & 15, man = b & 7;
    float v = ex == 0 ? std::ldexp((float)man, -9) : (ex == 15 && man == 7 ? NAN : std::ldexp(1.0f + man / 8.0f, ex - 7));
    return (b & 0x80) ? -v : v;
}
static inline f32x16_t mfma32_f8(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1, f32x16_t c) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_a[w][l] = a0; wave_a2[w][l] = a1; wave_b[w][l] = b0; wave_b2[w][l] = b1;
    wave_bar[w]->arrive_and_wait();
    auto byte = [](const uint4& lo, const uint4& hi, int i) { const uint4& q = i < 16 ? lo : hi; return (uint8_t)(((&q.x)[(i & 15) >> 2] >> (8 * (i & 3))) & 255u); };
    f32x16_t d;
    for (int g = 0; g < 4; ++g)
        for (int x = 0; x < 4; ++x) {
            const int row = 8 * g + 4 * (l >> 5) + x, col = l & 31;
            double s = c[4 * g + x];
            for (int kb = 0; kb < 2; ++kb)
                for (int i = 0; i < 32; ++i)
                    s += (double)e4m3_to_f(byte(wave_a[w][kb * 32 + row], wave_a2[w][kb * 32 + row], i)) *
                         (double)e4m3_to_f(byte(wave_b[w][kb * 32 + col], wave_b2[w][kb * 32 + col], i));
            d[4 * g + x] = (float)s;
        }
    wave_bar[w]->arrive_and_wait();
    return d;
}
template <bool F16 = false> static inline f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) { return mfma32_emu(a, b, c, F16); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0; }
static inline unsigned long long __builtin_amdgcn_s_memrealtime() { return 0; }
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_s_barrier();
static inline void atomicAdd(float* p, float v) { std::lock_guard<std::mutex> g(atomic_mutex); *p += v; }
static inline void atomicAdd(int* p, int v) { std::lock_guard<std::mutex> g(atomic_mutex); *p += v; }
#define __expf expf
#define __logf logf
#define __syncthreads() block_bar->arrive_and_wait()
static inline void __builtin_amdgcn_s_barrier() { block_bar->arrive_and_wait(); }
#define MSAM_DEVINL static inline
#define __global__
#define __device__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static
// one workgroup of 256 threads after the other: f() is the kernel call
template <class F> static void launch_grid(int gx, int gy, F f, int threads = 256) {
    gridDim = {gx, gy, 1};
    for (int by = 0; by < gy; ++by) for (int bx = 0; bx < gx; ++bx) {
        std::barrier<> b0(64), b1(64), b2(64), b3(64), b4(64), b5(64), b6(64), b7(64), bb(threads);
        std::barrier<>* wb[8] = {&b0, &b1, &b2, &b3, &b4, &b5, &b6, &b7};
        for (int i = 0; i < 8; ++i) wave_bar[i] = wb[i];
        block_bar = &bb;
        std::vector<std::thread> ts;
        for (int tx = 0; tx < threads; ++tx) ts.emplace_back([=] { threadIdx = {tx, 0, 0}; blockIdx = {bx, by, 0}; f(); });
        for (auto& t : ts) t.join();
    }
}
// (kernels whose workgroups leave early - "if (bid >= tiles) return" before any barrier - are launched with exactly their tiles)
"""


def build(tmpdir, name: str, kernel_source: str, entry_points: str, opt: str = "-O1"):
    """PRELUDE + kernel source + extern "C" entry points -> ctypes library."""
    cpp, so = os.path.join(tmpdir, name + ".cpp"), os.path.join(tmpdir, name + ".so")
    kernel_source = kernel_source.replace('"+v"', '"+r"')            # register-class constraints of inline asm barriers
    with open(cpp, "w") as fh:
        fh.write(PRELUDE + "\n" + kernel_source + "\n" + entry_points)
    subprocess.check_call(["g++", "-std=c++20", opt, "-ffp-contract=off", "-shared", "-fPIC", "-pthread", "-Wno-unknown-pragmas",
                           "-Wno-attributes", "-Wno-psabi", cpp, "-o", so])
    return ctypes.CDLL(so)
