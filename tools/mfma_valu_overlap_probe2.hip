// Controls for the question "do the matrix pipe and the vector ALU of one CDNA4 SIMD run at the same time?" (the first
// probe - tools/mfma_valu_overlap_probe.hip - put the MFMA waves first (older), let the compiler pack its "v_fma_f32" into dependent
// v_pk_fma_f32 and never ran one wave per SIMD).  Everything here is inline assembly, so the instruction streams are what the table says:
//   A. two waves per SIMD, one MFMA-only + one VALU-only (plain independent v_fma_f32), in BOTH age orders and with s_setprio on either side;
//   B. ONE wave per SIMD (one 256-thread workgroup per CU, 100 KB of LDS keeps a second one out): v_mfma_f32_32x32x16_bf16 followed by
//      k = 0..8 independent plain v_fma_f32, the guide's experiment (MI355X_MICROARCH.md "one wave per SIMD ... <= 5 fillers per gap");
//   C. the same stream with TWO such waves per SIMD (512-thread workgroup).
// Times: s_memtime of wave 0 of workgroup 0 (shader cycles) per MFMA, and the launch's HIP-event duration.  Counters: run under
// rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES (kernel names carry the variant).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_valu_overlap_probe2.hip -o /tmp/ovl2 && /tmp/ovl2
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define MFMA(acc_) "v_mfma_f32_32x32x16_bf16 %" #acc_ ", %12, %13, %" #acc_ "\n"
// The filler instruction (round 6, second call: WHICH vector instructions overlap with the matrix pipe): -DFILLER=0 plain v_fma_f32 (default),
// 1 v_pk_fma_f16, 3 v_exp_f16, 4 v_cvt_pk_f16_f32, 5 v_pk_max_f16, 6 v_fma_f16, 7 v_exp_f32, 8 v_pk_mul_f16
#ifndef FILLER
#define FILLER 0
#endif
#if FILLER == 0
#define FI(d_) "v_fma_f32 %" #d_ ", %" #d_ ", %14, %15\n"
#elif FILLER == 1
#define FI(d_) "v_pk_fma_f16 %" #d_ ", %" #d_ ", %14, %15\n"
#elif FILLER == 3
#define FI(d_) "v_exp_f16 %" #d_ ", %14\n"
#elif FILLER == 4
#define FI(d_) "v_cvt_pk_f16_f32 %" #d_ ", %14, %15\n"
#elif FILLER == 5
#define FI(d_) "v_pk_max_f16 %" #d_ ", %" #d_ ", %14\n"
#elif FILLER == 6
#define FI(d_) "v_fma_f16 %" #d_ ", %" #d_ ", %14, %15\n"
#elif FILLER == 7
#define FI(d_) "v_exp_f32 %" #d_ ", %14\n"
#elif FILLER == 8
#define FI(d_) "v_pk_mul_f16 %" #d_ ", %" #d_ ", %14\n"
#endif
#define F0 FI(4)
#define F1 FI(5)
#define F2 FI(6)
#define F3 FI(7)
#define F4 FI(8)
#define F5 FI(9)
#define F6 FI(10)
#define F7 FI(11)
#define FILL_0 ""
#define FILL_1 F0
#define FILL_2 F0 F1
#define FILL_3 F0 F1 F2
#define FILL_4 F0 F1 F2 F3
#define FILL_5 F0 F1 F2 F3 F4
#define FILL_6 F0 F1 F2 F3 F4 F5
#define FILL_7 F0 F1 F2 F3 F4 F5 F6
#define FILL_8 F0 F1 F2 F3 F4 F5 F6 F7
#define OPERANDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                 : "v"(bx), "v"(by), "v"(m), "v"(c)

// ---- B / C: one stream per wave, 4 MFMAs (independent accumulators) per iteration, each followed by K fillers
#define STREAM_KERNEL(K_, THREADS_)                                                                                          \
    __global__ __launch_bounds__(THREADS_) void stream_k##K_##_t##THREADS_(int n, unsigned long long* cyc, float* sink) {   \
        extern __shared__ char pad[];                                                                                        \
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};                                                                       \
        s16x8 bx, by;                                                                                                        \
        for (int i = 0; i < 8; ++i) { bx[i] = (short)(0x3f80 + i); by[i] = (short)(0x3f80 + 2 * i); }                        \
        float v0 = threadIdx.x * 1e-4f, v1 = v0 + 1e-4f, v2 = v0 + 2e-4f, v3 = v0 + 3e-4f, v4 = v0 + 4e-4f, v5 = v0 + 5e-4f, \
              v6 = v0 + 6e-4f, v7 = v0 + 7e-4f;                                                                              \
        const float m = 0.999f, c = 1e-3f;                                                                                   \
        if (n < 0) pad[threadIdx.x] = 1;                                                                                     \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                          \
        for (int i = 0; i < n; ++i)                                                                                          \
            asm volatile(MFMA(0) FILL_##K_ MFMA(1) FILL_##K_ MFMA(2) FILL_##K_ MFMA(3) FILL_##K_ OPERANDS);                  \
        asm volatile("s_nop 0" OPERANDS);                                                                                    \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                          \
        float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                                                     \
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];                                                     \
        if (s == 12345.678f) *sink = s;                                                                                      \
        if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;                                                             \
    }
STREAM_KERNEL(0, 256) STREAM_KERNEL(1, 256) STREAM_KERNEL(2, 256) STREAM_KERNEL(3, 256) STREAM_KERNEL(4, 256) STREAM_KERNEL(5, 256)
STREAM_KERNEL(6, 256) STREAM_KERNEL(7, 256) STREAM_KERNEL(8, 256)
STREAM_KERNEL(0, 512) STREAM_KERNEL(2, 512) STREAM_KERNEL(4, 512) STREAM_KERNEL(5, 512) STREAM_KERNEL(8, 512)

// ---- A: two waves per SIMD, one kind each.  ORDER 0: waves 0-3 MFMA (older), 4-7 VALU; ORDER 1: waves 0-3 VALU (older), 4-7 MFMA.
// PRIO: 0 none, 1 s_setprio 1 on the VALU waves, 2 s_setprio 1 on the MFMA waves.
template <int ORDER, int PRIO>
__global__ __launch_bounds__(512) void cross(int run_mfma, int run_valu, int n_mfma, int n_valu, unsigned long long* cyc, float* sink) {
    extern __shared__ char pad[];
    const int w = threadIdx.x >> 6;
    const bool is_mfma = ORDER == 0 ? w < 4 : w >= 4;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    s16x8 bx, by;
    for (int i = 0; i < 8; ++i) { bx[i] = (short)(0x3f80 + i); by[i] = (short)(0x3f80 + 2 * i); }
    float v0 = threadIdx.x * 1e-4f, v1 = v0 + 1e-4f, v2 = v0 + 2e-4f, v3 = v0 + 3e-4f, v4 = v0 + 4e-4f, v5 = v0 + 5e-4f, v6 = v0 + 6e-4f, v7 = v0 + 7e-4f;
    const float m = 0.999f, c = 1e-3f;
    if (n_mfma < 0) pad[threadIdx.x] = 1;
    unsigned long long t0 = 0, t1 = 0;
    if (is_mfma) {
        if (!run_mfma) return;
        if (PRIO == 2) __builtin_amdgcn_s_setprio(1);
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n_mfma; ++i) asm volatile(MFMA(0) MFMA(1) MFMA(2) MFMA(3) OPERANDS);
        asm volatile("s_nop 0" OPERANDS);
        t1 = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) cyc[0] = t1 - t0;
    } else {
        if (!run_valu) return;
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < n_valu; ++i) asm volatile(FILL_8 OPERANDS);
        asm volatile("s_nop 0" OPERANDS);
        t1 = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) cyc[1] = t1 - t0;
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 12345.678f) *sink = s;
}

static unsigned long long* g_cyc; static float* g_sink;
constexpr int LDS_ONE_PER_CU = 100 * 1024;

template <typename K>
static void time_stream(const char* name, K kern, int threads, int n) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ONE_PER_CU);
    float t = 0.f; unsigned long long cyc = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), LDS_ONE_PER_CU, 0, n, g_cyc, g_sink);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&t, a, b);
    }
    (void)hipMemcpy(&cyc, g_cyc, 8, hipMemcpyDeviceToHost);
    const double per_mfma_wave = (double)cyc / (4.0 * n);
    printf("| %s | %.3f | %.1f | %.1f |\n", name, t, per_mfma_wave, per_mfma_wave / (threads / 256));
}

template <int ORDER, int PRIO>
static void time_cross(const char* name, int n_mfma, int n_valu) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipFuncSetAttribute((const void*)cross<ORDER, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ONE_PER_CU);
    float t[3]; unsigned long long cyc[3][2] = {};
    const int cfg[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int c = 0; c < 3; ++c) {
        (void)hipMemset(g_cyc, 0, 16);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL((cross<ORDER, PRIO>), dim3(256), dim3(512), LDS_ONE_PER_CU, 0, cfg[c][0], cfg[c][1], n_mfma, n_valu, g_cyc, g_sink);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&t[c], a, b);
        }
        (void)hipMemcpy(cyc[c], g_cyc, 16, hipMemcpyDeviceToHost);
    }
    printf("| %s | %.3f | %.3f | %.3f | %.2f | %.1f -> %.1f | %.2f -> %.2f |\n", name, t[0], t[1], t[2], t[2] / (t[0] > t[1] ? t[0] : t[1]),
           (double)cyc[0][0] / (4.0 * n_mfma), (double)cyc[2][0] / (4.0 * n_mfma), (double)cyc[1][1] / (8.0 * n_valu), (double)cyc[2][1] / (8.0 * n_valu));
}

int main() {
    (void)hipMalloc(&g_sink, 4); (void)hipMalloc(&g_cyc, 16);
    printf("## B. ONE wave per SIMD: v_mfma_f32_32x32x16_bf16 + k plain independent v_fma_f32 per MFMA (4 MFMAs per iteration, 4000 iterations)\n\n");
    printf("| stream | launch (ms) | cycles per MFMA of a wave (s_memtime) | cycles per MFMA of the SIMD |\n|---|---|---|---|\n");
    const int n = 4000;
    time_stream("k = 0", stream_k0_t256, 256, n); time_stream("k = 1", stream_k1_t256, 256, n); time_stream("k = 2", stream_k2_t256, 256, n);
    time_stream("k = 3", stream_k3_t256, 256, n); time_stream("k = 4", stream_k4_t256, 256, n); time_stream("k = 5", stream_k5_t256, 256, n);
    time_stream("k = 6", stream_k6_t256, 256, n); time_stream("k = 7", stream_k7_t256, 256, n); time_stream("k = 8", stream_k8_t256, 256, n);
    printf("\n## C. the same stream, TWO such waves per SIMD (one 512-thread workgroup per CU)\n\n");
    printf("| stream | launch (ms) | cycles per MFMA of a wave (s_memtime) | cycles per MFMA of the SIMD |\n|---|---|---|---|\n");
    time_stream("k = 0", stream_k0_t512, 512, n); time_stream("k = 2", stream_k2_t512, 512, n); time_stream("k = 4", stream_k4_t512, 512, n);
    time_stream("k = 5", stream_k5_t512, 512, n); time_stream("k = 8", stream_k8_t512, 512, n);
    printf("\n## A. two waves per SIMD, one MFMA-only (4000 x 4 v_mfma_f32_32x32x16_bf16) + one VALU-only (16000 x 8 independent v_fma_f32)\n\n");
    printf("| arrangement | matrix alone (ms) | vector alone (ms) | both (ms) | both / max | cycles per MFMA alone -> both | cycles per v_fma alone -> both |\n|---|---|---|---|---|---|---|\n");
    time_cross<0, 0>("MFMA waves older, no priority", 4000, 16000);
    time_cross<1, 0>("VALU waves older, no priority", 4000, 16000);
    time_cross<0, 1>("MFMA waves older, s_setprio 1 on the VALU waves", 4000, 16000);
    time_cross<1, 2>("VALU waves older, s_setprio 1 on the MFMA waves", 4000, 16000);
    time_cross<0, 2>("MFMA waves older, s_setprio 1 on the MFMA waves", 4000, 16000);
    return 0;
}
