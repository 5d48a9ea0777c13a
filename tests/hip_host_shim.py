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
#include <condition_variable>
#include <cstdio>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
using std::min; using std::max;
struct idx3 { int x, y, z; };
// ---- execution layer: a wave = ONE host thread whose 64 lanes are user-level contexts (switched in a few nanoseconds by ctx_switch
// below); a wave-level synchronisation is a round of the wave's scheduler, __syncthreads a condition-variable barrier between the
// wave threads.  (One host thread per lane - the first form of this shim - spent its time in 64-thread barriers on 8 cores.)
#if !defined(__x86_64__)
#error "tests/hip_host_shim.py: the lane contexts are switched by x86-64 assembly"
#endif
extern "C" void msam_ctx_switch(void** save_sp, void* load_sp);
asm(".text\n.weak msam_ctx_switch\n.type msam_ctx_switch,@function\nmsam_ctx_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size msam_ctx_switch, .-msam_ctx_switch\n");
struct LaneCtx { void* sp; idx3 tid; int wait; bool done; };              // wait: 0 runnable, 1 at a wave sync, 2 at __syncthreads
struct BlockBarrier {                                                         // between the wave threads of a workgroup; waves that
    std::mutex m; std::condition_variable cv; int expected = 0, arrived = 0; unsigned gen = 0;   // have finished the block drop out
    void arrive_and_wait() {
        std::unique_lock<std::mutex> lk(m);
        const unsigned g = gen;
        if (++arrived >= expected) { arrived = 0; ++gen; cv.notify_all(); return; }
        cv.wait(lk, [&] { return gen != g; });
    }
    void drop() { std::lock_guard<std::mutex> lk(m); if (--expected > 0 && arrived >= expected) { arrived = 0; ++gen; cv.notify_all(); } }
};
struct WaveCtx { LaneCtx lane[64]; void* sched_sp; int cur; int index; void (*entry)(void*); void* arg; BlockBarrier* bar; };
static thread_local WaveCtx* WV = nullptr;
static thread_local idx3 blockIdx;
#define threadIdx (WV->lane[WV->cur].tid)
static idx3 gridDim;
static std::mutex atomic_mutex;
static inline void lane_yield(int why) {
    LaneCtx& L = WV->lane[WV->cur];
    L.wait = why;
    msam_ctx_switch(&L.sp, WV->sched_sp);
}
static inline void wave_sync() { lane_yield(1); }
static inline void block_sync() { lane_yield(2); }
static void lane_trampoline() {
    WaveCtx* w = WV;
    w->entry(w->arg);
    w->lane[w->cur].done = true;
    msam_ctx_switch(&w->lane[w->cur].sp, w->sched_sp);                         // never resumed
    std::abort();
}
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
    f32x2_t operator-(float s) const { return {x - s, y - s}; }
    f32x2_t& operator-=(float s) { x -= s; y -= s; return *this; }
};
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
typedef float f32x16_t __attribute__((vector_size(64)));
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct float4 { float x, y, z, w; };
static inline uint2 make_uint2(uint32_t a, uint32_t b) { return uint2{a, b}; }
static inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
struct float2 { float x, y; };
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
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
    wave_sync();
    const long long r = wave_buf[w][l ^ src_lane_xor];
    wave_sync();
    return r;
}
static inline int __shfl_xor(int v, int o) { return (int)wave_xchg(v, o); }
static inline float __shfl_xor(float v, int o) { return __uint_as_float((uint32_t)wave_xchg(__float_as_uint(v), o)); }
static inline unsigned long long __ballot(int pred) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = pred ? 1 : 0;
    wave_sync();
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= (unsigned long long)(wave_buf[w][i] != 0) << i;
    wave_sync();
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
    wave_sync();
    const int g0 = l & ~15, r = l & 15;
    u16 e[4];
    for (int j = 0; j < 4; ++j) {
        const uint2& src = wave_tr[w][g0 + 4 * j + (r >> 2)];
        const uint32_t wd = (r & 2) ? src.y : src.x;
        e[j] = (u16)((r & 1) ? (wd >> 16) : (wd & 0xffffu));
    }
    wave_sync();
    return uint2{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16)};
}
static inline void __builtin_amdgcn_wave_barrier() { wave_sync(); }
#define __builtin_amdgcn_fence(order_, scope_) ((void)0)     /* memory-model fence: the emulation's wave_sync() is already sequentially consistent */
static inline float elem16(const uint4& v, int j, bool f16) {
    const uint32_t wd = (&v.x)[j >> 1];
    const u16 b = (u16)((j & 1) ? (wd >> 16) : (wd & 0xffffu));
    return f16 ? h16_to_f(b) : bf2f(b);
}
static inline f32x4_t mfma16_emu(const uint4& a, const uint4& b, f32x4_t c, bool f16) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_a[w][l] = a; wave_b[w][l] = b;
    wave_sync();
    f32x4_t d;
    for (int i = 0; i < 4; ++i) {
        const int row = (l >> 4) * 4 + i, col = l & 15;
        double s = c[i];
        for (int kb = 0; kb < 4; ++kb)
            for (int j = 0; j < 8; ++j) s += (double)elem16(wave_a[w][kb * 16 + row], j, f16) * (double)elem16(wave_b[w][kb * 16 + col], j, f16);
        d[i] = (float)s;
    }
    wave_sync();
    return d;
}
static inline f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c) { return mfma16_emu(a, b, c, false); }
static inline f32x4_t mfma16h(const uint4& a, const uint4& b, f32x4_t c) { return mfma16_emu(a, b, c, true); }
static inline f32x16_t mfma32_emu(const uint4& a, const uint4& b, f32x16_t c, bool f16) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_a[w][l] = a; wave_b[w][l] = b;
    wave_sync();
    f32x16_t d;
    for (int g = 0; g < 4; ++g)
        for (int x = 0; x < 4; ++x) {
            const int row = 8 * g + 4 * (l >> 5) + x, col = l & 31;
            double s = c[4 * g + x];
            for (int kb = 0; kb < 2; ++kb)
                for (int j = 0; j < 8; ++j) s += (double)elem16(wave_a[w][kb * 32 + row], j, f16) * (double)elem16(wave_b[w][kb * 32 + col], j, f16);
            d[4 * g + x] = (float)s;
        }
    wave_sync();
    return d;
}
// v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate): lane l holds A[row = l % 32][k = l / 32], B[k = l / 32][col = l % 32]; D as the 32x32x16
// form.  The device result is bit for bit a k-ordered fmaf chain (cdna_hip_programming.md section 3), which is what runs here.
static inline f32x16_t __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16_t c, int, int, int) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = (long long)(((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a));
    wave_sync();
    f32x16_t d;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float s = c[r];
        for (int k = 0; k < 2; ++k) {
            const float av = __uint_as_float((uint32_t)(unsigned long long)wave_buf[w][k * 32 + row]);
            const float bv = __uint_as_float((uint32_t)((unsigned long long)wave_buf[w][k * 32 + col] >> 32));
            s = std::fmaf(av, bv, s);
        }
        d[r] = s;
    }
    wave_sync();
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
static inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_t r, void* lds, int size, int voff, int soff, int imm, int) {
    char v[16] = {0};
    const long o = (long)voff + soff + imm;
    if (o >= 0 && o + size <= (long)r.bytes) std::memcpy(v, r.base + o, size);
    std::memcpy((char*)lds + (threadIdx.x & 63) * size, v, size);
}
static inline void buf_store16(const uint4& v, rsrc_t r, int voff, int soff) {
    const long o = (long)voff + soff;
    if (o >= 0 && o + 16 <= (long)r.bytes) std::memcpy(r.base + o, &v, 16);
}
static inline void buf_store8(const uint2& v, rsrc_t r, int voff, int soff) {
    const long o = (long)voff + soff;
    if (o >= 0 && o + 8 <= (long)r.bytes) std::memcpy(r.base + o, &v, 8);
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
    wave_sync();
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
    wave_sync();
    return d;
}
template <bool F16 = false> static inline f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) { return mfma32_emu(a, b, c, F16); }
static inline unsigned long long __builtin_readcyclecounter() { return 0; }
static inline uint8_t f_to_e4m3(float f) {                  // OCP e4m3 (fn): round to nearest even, saturating (v_cvt_pk_fp8_f32)
    uint32_t u; std::memcpy(&u, &f, 4);
    const uint8_t sign = (u >> 24) & 0x80;
    if ((u & 0x7fffffffu) > 0x7f800000u) return sign | 0x7f;
    float a = std::fabs(f);
    if (a >= 448.f) return sign | 0x7e;
    if (a < std::ldexp(1.f, -10)) return sign;                // below half of the smallest subnormal (2^-9)
    int e; std::frexp(a, &e);                                 // a = m 2^e, m in [0.5, 1)
    int ex = e - 1;                                           // a = 1.xxx 2^ex
    if (ex < -6) ex = -6;                                     // subnormal range: fixed exponent
    const float q = std::ldexp(a, 3 - ex);                    // units of 2^(ex - 3): integer part = mantissa incl. the hidden bit
    float r = std::nearbyint(q);                              // current rounding mode: to nearest even
    int m = (int)r;
    int be = ex + 7;
    if (ex == -6 && m < 8) be = 0;                            // subnormal
    else { if (m == 16) { m = 8; ++be; } m -= 8; if (ex == -6 && be == 0) be = 1; }
    if (be > 15 || (be == 15 && m > 6)) return sign | 0x7e;
    return sign | (uint8_t)(be << 3) | (uint8_t)m;
}
static inline int __builtin_amdgcn_cvt_pk_fp8_f32(float a, float b, int old, bool hi) {
    const uint32_t pk = (uint32_t)f_to_e4m3(a) | ((uint32_t)f_to_e4m3(b) << 8);
    const uint32_t o = (uint32_t)old;
    return (int)(hi ? ((o & 0x0000ffffu) | (pk << 16)) : ((o & 0xffff0000u) | pk));
}
static inline int __lane_id() { return threadIdx.x & 63; }
static inline long long wave_read(long long v, int src_lane) {          // value of `v` in lane src_lane of this wave
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    wave_buf[w][l] = v;
    wave_sync();
    const long long r = wave_buf[w][src_lane & 63];
    wave_sync();
    return r;
}
static inline int __shfl(int v, int src) { return (int)wave_read(v, src); }
// v_permlane16_swap / v_permlane32_swap (gfx950): rows of 16 (32) lanes; the ODD rows of the first operand are exchanged with the EVEN
// rows of the second; the builtin returns both registers
struct u32pair_t { uint32_t v[2]; uint32_t operator[](int i) const { return v[i]; } };
static inline u32pair_t permlane_swap_emu(uint32_t a, uint32_t b, int row) {
    const int l = threadIdx.x & 63;
    const bool odd = (l / row) & 1;
    const long long other = wave_read((long long)a | ((long long)b << 32), odd ? l - row : l + row);
    u32pair_t r;
    r.v[0] = odd ? (uint32_t)((unsigned long long)other >> 32) : a;      // odd row of the first operand <- even row of the second
    r.v[1] = odd ? b : (uint32_t)other;                                   // even row of the second operand <- odd row of the first
    return r;
}
static inline u32pair_t __builtin_amdgcn_permlane16_swap(uint32_t a, uint32_t b, bool, bool) { return permlane_swap_emu(a, b, 16); }
static inline u32pair_t __builtin_amdgcn_permlane32_swap(uint32_t a, uint32_t b, bool, bool) { return permlane_swap_emu(a, b, 32); }
static inline float __shfl(float v, int src) { return __uint_as_float((uint32_t)wave_read(__float_as_uint(v), src)); }
static inline int __shfl_up(int v, int d) { const int l = threadIdx.x & 63; const int r = (int)wave_read(v, l >= d ? l - d : l); return r; }
static inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)wave_read(v, lane); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fmaf_rn(float a, float b, float c) { return std::fmaf(a, b, c); }
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31)); }
static inline uint32_t __brev(uint32_t x) { uint32_t r = 0; for (int i = 0; i < 32; ++i) r |= ((x >> i) & 1u) << (31 - i); return r; }
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(uint32_t x) { return __builtin_ffs((int)x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(uint32_t x) { return x ? __builtin_clz(x) : 32; }
static inline unsigned long long __builtin_readcyclecounter_emu() { return 0; }
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
static inline unsigned __builtin_amdgcn_s_getreg(int) { return 0; }
static inline unsigned long long __builtin_amdgcn_s_memrealtime() {      /* the 100 MHz constant clock */
    return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline void __builtin_amdgcn_s_barrier();
static inline float atomicAdd(float* p, float v) { std::lock_guard<std::mutex> g(atomic_mutex); const float o = *p; *p += v; return o; }
static inline int atomicAdd(int* p, int v) { std::lock_guard<std::mutex> g(atomic_mutex); const int o = *p; *p += v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { std::lock_guard<std::mutex> g(atomic_mutex); const unsigned o = *p; *p += v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { std::lock_guard<std::mutex> g(atomic_mutex); const auto o = *p; *p += v; return o; }
static inline int atomicMin(int* p, int v) { std::lock_guard<std::mutex> g(atomic_mutex); const int o = *p; *p = std::min(o, v); return o; }
static inline int atomicMax(int* p, int v) { std::lock_guard<std::mutex> g(atomic_mutex); const int o = *p; *p = std::max(o, v); return o; }
static inline unsigned atomicMin(unsigned* p, unsigned v) { std::lock_guard<std::mutex> g(atomic_mutex); const unsigned o = *p; *p = std::min(o, v); return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { std::lock_guard<std::mutex> g(atomic_mutex); const unsigned o = *p; *p = std::max(o, v); return o; }
static inline int atomicOr(int* p, int v) { std::lock_guard<std::mutex> g(atomic_mutex); const int o = *p; *p |= v; return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { std::lock_guard<std::mutex> g(atomic_mutex); const unsigned o = *p; *p |= v; return o; }
static inline int atomicCAS(int* p, int cmp, int v) { std::lock_guard<std::mutex> g(atomic_mutex); const int o = *p; if (o == cmp) *p = v; return o; }
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long cmp, unsigned long long v) {
    std::lock_guard<std::mutex> g(atomic_mutex); const auto o = *p; if (o == cmp) *p = v; return o;
}
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { std::lock_guard<std::mutex> g(atomic_mutex); const auto o = *p; *p |= v; return o; }
static inline int atomicExch(int* p, int v) { std::lock_guard<std::mutex> g(atomic_mutex); const int o = *p; *p = v; return o; }
#define __expf expf
#define __logf logf
#define __syncthreads() block_sync()
static inline void __builtin_amdgcn_s_barrier() { block_sync(); }
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
    const int nw = (threads + 63) / 64;
    constexpr size_t STACK = 256 * 1024;
    BlockBarrier bar, next;
    next.expected = nw;
    std::vector<std::thread> ts;
    for (int w = 0; w < nw; ++w)
        ts.emplace_back([=, &bar, &next] {
            F fn = f;
            WaveCtx* wv = new WaveCtx();
            char* stacks = (char*)std::malloc(64 * STACK);
            wv->index = w; wv->bar = &bar; wv->arg = &fn;
            wv->entry = [](void* a) { (*(F*)a)(); };
            WV = wv;
            const int lanes = std::min(64, threads - w * 64);
            for (int by = 0; by < gy; ++by)
                for (int bx = 0; bx < gx; ++bx) {
                    blockIdx = {bx, by, 0};
                    if (w == 0) { std::lock_guard<std::mutex> lk(bar.m); bar.expected = nw; bar.arrived = 0; }
                    next.arrive_and_wait();                                    // the block barrier is set up; LDS of the last block is free
                    for (int l = 0; l < 64; ++l) {
                        LaneCtx& L = wv->lane[l];
                        L.tid = {w * 64 + l, 0, 0}; L.wait = 0; L.done = l >= lanes;
                        void** top = (void**)(stacks + (size_t)(l + 1) * STACK);   // 16-byte aligned
                        top[-1] = nullptr;                                      // (return address slot of the trampoline's frame)
                        top[-2] = (void*)&lane_trampoline;
                        for (int i = 3; i <= 8; ++i) top[-i] = nullptr;         // rbp rbx r12 r13 r14 r15
                        L.sp = (void*)(top - 8);
                    }
                    while (true) {
                        int live = 0, w1 = 0, w2 = 0;
                        bool progress = false;
                        for (int l = 0; l < 64; ++l) {
                            LaneCtx& L = wv->lane[l];
                            if (L.done) continue;
                            if (L.wait == 0) { wv->cur = l; msam_ctx_switch(&wv->sched_sp, L.sp); progress = true; }
                            if (L.done) continue;
                            ++live; w1 += L.wait == 1; w2 += L.wait == 2;
                        }
                        if (live == 0) break;
                        if (w1 == live) { for (int l = 0; l < 64; ++l) wv->lane[l].wait = 0; }
                        else if (w2 == live) { bar.arrive_and_wait(); for (int l = 0; l < 64; ++l) wv->lane[l].wait = 0; }
                        else if (!progress) { std::fprintf(stderr, "hip_host_shim: divergent barrier in wave %d (%d live, %d at a wave sync, %d at __syncthreads)\n", w, live, w1, w2); std::abort(); }
                    }
                    bar.drop();                                                 // this wave takes no further part in the block's barriers
                    next.arrive_and_wait();                                    // every wave has left the block: its barrier may be reset
                }
            std::free(stacks);
            delete wv;
        });
    for (auto& t : ts) t.join();
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


# ------------------------------------------------------------------------------------------------------------------------------
# Whole-file builds: a complete csrc/*.hip file - kernels AND its extern "C" entry points - compiled for the host.  The file's
# `#include "common.h"` finds a generated header (the prelude above + the decoder-type helpers + the portable tail of the real common.h
# + a minimal HIP runtime whose hipLaunchKernelGGL runs the workgroups one after the other on host threads), so what is exercised is
# the library's C ABI itself: argument checks, derived launch parameters, kernel, epilogue.
# ------------------------------------------------------------------------------------------------------------------------------

RUNTIME = r"""
// ---- the decoder's 16-bit type (common.h MSAM_DEC_F16 = 1: IEEE fp16)
#define MSAM_DEC_F16 1
static inline f32x4_t mfma16d(const uint4& a, const uint4& b, f32x4_t c) { return mfma16h(a, b, c); }
static inline uint32_t pack2d(float lo, float hi) { return pack2h(lo, hi); }
static inline u16 f2d(float f) { return f2h(f); }
static inline float d2f(u16 h) { return h16_to_f(h); }
#define MSAM_D16_ONE 0x3C00u
#define MSAM_D16 4
// ---- minimal HIP runtime: one stream, launches run to completion before they return
typedef int hipError_t;
enum { hipSuccess = 0 };
typedef void* hipStream_t;
typedef void* hipEvent_t;
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
enum { hipDeviceAttributeMultiprocessorCount = 0, hipFuncAttributeMaxDynamicSharedMemorySize = 0, hipMemcpyDeviceToDevice = 0,
       hipMemcpyDeviceToHost = 1 };
static inline int emu_cus() { const char* v = std::getenv("MSAM_EMU_CUS"); return v ? std::atoi(v) : 2; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "host emulation"; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = emu_cus(); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s_, size_t n, int, hipStream_t) { std::memmove(d, s_, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return hipSuccess; }
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyFromSymbol(void* d, const void* sym, size_t n) { std::memcpy(d, sym, n); return hipSuccess; }
#define hipLaunchKernelGGL(kern_, grid_, block_, shmem_, stream_, ...)                                 \
    do {                                                                                               \
        const dim3 g__ = (grid_), b__ = (block_);                                                      \
        launch_grid((int)g__.x, (int)g__.y, [=] { kern_(__VA_ARGS__); }, (int)b__.x);                  \
    } while (0)
"""


def host_common_h(root: str) -> str:
    """The header a host-compiled csrc file includes instead of csrc/common.h."""
    common = open(os.path.join(root, "micro_sam_amd", "csrc", "common.h")).read()
    g0 = common.index("MSAM_DEVINL float gelu_erf(float x)")
    g1 = common.index("// round-to-nearest-even fp32 -> packed fp16")
    t0 = common.index("MSAM_DEVINL float wave_sum_xor16(float v)")
    tail = common[t0:]
    tail = tail[: tail.rindex("#endif")] if tail.rstrip().endswith("#endif") else tail
    return ("#pragma once\n#include <cstdlib>\n" + PRELUDE.replace("static inline int swz(int row)", "static inline int swz_unused(int row)") +
            "static inline int swz(int row) { return ((row >> 1) & 7) ^ (((row + 4) >> 3) & 1); }\n" + RUNTIME + common[g0:g1] + tail)


def build_file(tmpdir: str, root: str, hip_name: str, replacements=(), extra: str = "", opt: str = "-O1", source: str = None):
    """Compile micro_sam_amd/csrc/<hip_name> as a whole for the host.  ``replacements`` = (old, new) pairs for the few clang-only
    helper definitions of a file (each must occur exactly once); dynamic LDS declarations become static 160 KB arrays.  ``source``:
    the file's text if it is not the one on disk (tools/uf_lab.py's variants of a kernel)."""
    import re
    src = source if source is not None else open(os.path.join(root, "micro_sam_amd", "csrc", hip_name)).read()
    for old, new in replacements:
        assert src.count(old) == 1, (hip_name, old[:60], src.count(old))
        src = src.replace(old, new)
    src = re.sub(r"extern\s+__shared__\s+(__attribute__\(\(aligned\(16\)\)\)\s+)?(\w[\w\s]*?)\s+(\w+)\[\];",
                 lambda m: f"static {m.group(2)} {m.group(3)}[163840 / sizeof({m.group(2)})];", src)
    src = src.replace('"+v"', '"+r"')
    csrc = os.path.join(tmpdir, "pkg", "csrc")
    os.makedirs(csrc, exist_ok=True)
    os.makedirs(os.path.join(tmpdir, "include"), exist_ok=True)
    with open(os.path.join(csrc, "common.h"), "w") as fh:
        fh.write(host_common_h(root))
    with open(os.path.join(tmpdir, "include", "msam_hip.h"), "w") as fh:
        fh.write(open(os.path.join(root, "include", "msam_hip.h")).read())
    cpp = os.path.join(csrc, hip_name.replace(".hip", "_host.cpp"))
    so = os.path.join(tmpdir, hip_name.replace(".hip", "_host.so"))
    with open(cpp, "w") as fh:
        fh.write(src + "\n" + extra)
    subprocess.check_call(["g++", "-std=c++20", opt, "-ffp-contract=off", "-shared", "-fPIC", "-pthread", "-Wno-unknown-pragmas",
                           "-Wno-attributes", "-Wno-psabi", "-Wno-unused-value", cpp, "-o", so])
    return ctypes.CDLL(so)


# the library-wide helpers other files expect from gemm.hip (msam_set_error, msam_check_launch, the profiling marks)
ERROR_STUBS = r"""
#include <string.h>
static char g_emu_err[512];
void msam_set_error(const char* msg) { strncpy(g_emu_err, msg, 511); }
int msam_check_launch(const char*) { return 0; }
void msam_profile_mark2(void*, int, double, double, int) {}
void msam_profile_mark(void*, int, double) {}
extern "C" const char* emu_last_error() { return g_emu_err; }
"""


# ---- the few clang-only helper definitions of the library, file by file: (start marker, end marker, host text) - the text between the
# markers (start inclusive, end exclusive) is replaced
_PKH = """static inline float rh_(float x) { return h16_to_f(f2h(x)); }
template <int EXPM> static inline uint32_t gelu_pk_h(float x0, float x1) {      // packed fp16 arithmetic: every operation rounded to fp16
    float g[2]; const float xs[2] = {x0, x1};
    for (int i = 0; i < 2; ++i) {
        const float x = rh_(xs[i]); const float r = x > 0.f ? x : 0.f; const float t = rh_(std::fmaf(r, 2.f, -x));
        float q = rh_(std::fmaf(t, rh_(-0.0248758f), rh_(-0.49884797f)));
        q = rh_(std::fmaf(q, t, rh_(-1.12922424f))); q = rh_(std::fmaf(q, t, rh_(-1.00353579f)));
        g[i] = rh_(std::fmaf(-t, rh_(std::exp2(q)), r));
    }
    return pack2h(g[0], g[1]);
}
template <int EXPM> static inline void gelu_pk_h2(float x0, float x1, float x2, float x3, uint32_t& g01, uint32_t& g23) {
    g01 = gelu_pk_h<EXPM>(x0, x1); g23 = gelu_pk_h<EXPM>(x2, x3);           // (the device form interleaves the four exponentials: same values)
}
"""
_GW = """struct u32x4_t_gw { const char* base; long bytes; };
static char* gw_lds_base = nullptr;
static inline void gw_dma16(const u32x4_t_gw& r, int voff, int soff, unsigned lds_addr) {
    uint4 v{0, 0, 0, 0};
    const long o = (long)voff + soff;
    if (o >= 0 && o + 16 <= r.bytes) std::memcpy(&v, r.base + o, 16);
    std::memcpy(gw_lds_base + lds_addr + (threadIdx.x & 63) * 16, &v, 16);
}
static inline u32x4_t_gw gw_rsrc(const void* base, long bytes) { return u32x4_t_gw{(const char*)base, bytes}; }

"""
FILE_PATCHES = {
    "upfused.hip": [("typedef _Float16 h16x2_t", "// CEN (round 6): the caller hands over CENTRED first-layer weights", _PKH)],
    "attention.hip": [("typedef short gs16x4_t", "template <int HD, bool F16 = false>\n__global__ __launch_bounds__(256, HD == 64 ? 3 : 2)",
                       "static inline uint2 g_tr16(const unsigned char* p) { return ds_read_tr16_b64_emu(p); }\n\n")],
    "decfold.hip": [("typedef short s16x4_t", "// Q'[p][h*8 + t][c]", "static inline uint2 lds_tr16(const unsigned char* p) { return ds_read_tr16_b64_emu(p); }\n\n")],
    "gemm.hip": [("typedef float f32x16_t __attribute__((ext_vector_type(16)));", "// Epilogue straight from the accumulators of the 128 x 64 wave tile", ""),
                 ("typedef unsigned int u32x4_t_gw", "template <bool F16>\n__global__ __launch_bounds__(256, 2) void gemm2w_kernel", _GW)],
}
TEXT_PATCHES = {
    "strict.hip": [("typedef float f32x16_t __attribute__((ext_vector_type(16)));", "")],
    "gemm.hip": [("#if defined(__HIP_DEVICE_COMPILE__)                  // (address-space-qualified struct copies do not parse in the host pass)", "#if 1"),
                 ("    typedef const __attribute__((address_space(4))) GroupItem* ItemPtr;\n    ItemPtr it = (ItemPtr)__builtin_amdgcn_kernarg_segment_ptr() + blockIdx.y;",
                  "    const GroupItem* it = &g.it[blockIdx.y];          // (the device reads its kernel-argument segment)"),("const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)dyn;", "const unsigned lds0 = 0; gw_lds_base = (char*)dyn;"),
                 ('asm volatile("s_waitcnt vmcnt(6)" ::: "memory")', "(void)0"), ('asm volatile("s_waitcnt vmcnt(0)" ::: "memory")', "(void)0")],
}


def patched_source(root: str, hip_name: str) -> str:
    src = open(os.path.join(root, "micro_sam_amd", "csrc", hip_name)).read()
    for start, end, text in FILE_PATCHES.get(hip_name, []):
        assert src.count(start) == 1, (hip_name, start)
        a = src.index(start)
        b = src.index(end, a)
        src = src[:a] + text + src[b:]
    for old, new in TEXT_PATCHES.get(hip_name, []):
        assert old in src, (hip_name, old)
        src = src.replace(old, new)
    return src


def build_library(tmpdir: str, root: str, files=None, jobs: int = 8):
    """Every csrc/*.hip file (or ``files``) compiled for the host and linked into one shared object with the library's C ABI
    (include/msam_hip.h): the kernels run on host threads behind the same entry points the GPU build exports."""
    import hashlib
    import re
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    csrc_dir = os.path.join(root, "micro_sam_amd", "csrc")
    names = files or sorted(n for n in os.listdir(csrc_dir) if n.endswith(".hip"))
    if files is None:
        # one build per state of the sources and of this shim, shared by the test modules of a run (and by later runs)
        h = hashlib.sha1(open(__file__.replace(".pyc", ".py"), "rb").read())
        for n in sorted(os.listdir(csrc_dir)) + ["../../include/msam_hip.h"]:
            h.update(open(os.path.join(csrc_dir, n), "rb").read())
        tmpdir = os.path.join(tempfile.gettempdir(), "msam_host_lib_" + h.hexdigest()[:16])
        cached = os.path.join(tmpdir, "libmsam_hip_host.so")
        if os.path.exists(cached):
            return ctypes.CDLL(cached)
        os.makedirs(tmpdir, exist_ok=True)
    csrc = os.path.join(tmpdir, "pkg", "csrc")
    os.makedirs(csrc, exist_ok=True)
    os.makedirs(os.path.join(tmpdir, "include"), exist_ok=True)
    with open(os.path.join(csrc, "common.h"), "w") as fh:
        fh.write(host_common_h(root))
    with open(os.path.join(tmpdir, "include", "msam_hip.h"), "w") as fh:
        fh.write(open(os.path.join(root, "include", "msam_hip.h")).read())
    for n in os.listdir(csrc_dir):                                # other headers of csrc (none clang-specific)
        if n.endswith(".h") and n != "common.h":
            with open(os.path.join(csrc, n), "w") as fh:
                fh.write(open(os.path.join(csrc_dir, n)).read())
    objs = []

    def compile_one(name):
        src = patched_source(root, name)
        src = re.sub(r"extern\s+__shared__\s+(__attribute__\(\(aligned\(16\)\)\)\s+)?(\w[\w\s]*?)\s+(\w+)\[\];",
                     lambda m: f"static {m.group(2)} {m.group(3)}[163840 / sizeof({m.group(2)})];", src)
        src = src.replace('"+v"', '"+r"')
        cpp = os.path.join(csrc, name.replace(".hip", "_host.cpp"))
        obj = cpp.replace(".cpp", ".o")
        with open(cpp, "w") as fh:
            fh.write(src)
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-c", "-fPIC", "-pthread", "-Wno-unknown-pragmas",
                               "-Wno-attributes", "-Wno-psabi", "-Wno-unused-value", cpp, "-o", obj])
        return obj

    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(compile_one, names))
    if "gemm.hip" not in names:
        stub = os.path.join(csrc, "stubs.cpp")
        with open(stub, "w") as fh:
            fh.write("#include <cstring>\n" + ERROR_STUBS)
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-c", "-fPIC", stub, "-o", stub.replace(".cpp", ".o")])
        objs.append(stub.replace(".cpp", ".o"))
    so = os.path.join(tmpdir, "libmsam_hip_host.so")
    subprocess.check_call(["g++", "-shared", "-pthread", "-o", so + ".tmp"] + objs)
    os.replace(so + ".tmp", so)                                    # (complete before it becomes visible to a concurrent run)
    return ctypes.CDLL(so)
