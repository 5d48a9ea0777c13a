// Do the matrix pipe and the vector ALU of one CDNA4 SIMD run at the same time when the instructions come from DIFFERENT waves?
// One workgroup of 8 waves per CU slot (waves 0-3 and 4-7 land on SIMDs 0-3 twice): waves 0-3 issue only MFMAs, waves 4-7 only vector
// instructions.  T(MFMA waves alone), T(vector waves alone), T(both): both ~ max -> the pipes overlap; both ~ sum -> they do not.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_valu_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// kind: 0 = f32 32x32x2 (16 passes), 1 = bf16 32x32x16 (8 passes); valu: 0 = v_fma_f32, 1 = v_exp_f32, 2 = v_pk_fma_f16-like packed f32 (v_pk_fma_f32)
template <int KIND, int VALU>
__global__ __launch_bounds__(512) void probe(int run_mfma, int run_valu, int n_mfma, int n_valu, float* sink) {
    const int w = threadIdx.x >> 6;
    if (w < 4) {
        if (!run_mfma) return;
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
        s16x8 bx, by;
        for (int i = 0; i < 8; ++i) { bx[i] = (short)(0x3f80 + i); by[i] = (short)(0x3f80 + 2 * i); }
        for (int i = 0; i < n_mfma; ++i) {
            if (KIND == 0) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, bx, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, bx, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, by, a3, 0, 0, 0);
            }
        }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
        if (s == 12345.678f) *sink = s;
    } else {
        if (!run_valu) return;
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i) * 1e-4f;
        const float m = 0.999f, c = 1e-3f;
        for (int i = 0; i < n_valu; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (VALU == 0) v[j] = __builtin_fmaf(v[j], m, c);
                else if (VALU == 1) v[j] = __builtin_amdgcn_exp2f(v[j] * m);
                else { v[j] = __builtin_fmaf(v[j], m, c); }
            }
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += v[i];
        if (s == 12345.678f) *sink = s;
    }
}

// the same question inside ONE wave: 4 independent MFMAs followed by NV independent v_fma_f32 per iteration (two waves per SIMD, all alike)
template <int KIND, int NV>
__global__ __launch_bounds__(512) void probe_same_wave(int n, float* sink) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
    s16x8 bx, by;
    for (int i = 0; i < 8; ++i) { bx[i] = (short)(0x3f80 + i); by[i] = (short)(0x3f80 + 2 * i); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i) * 1e-4f;
    const float m = 0.999f, c = 1e-3f;
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, bx, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, bx, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, by, a3, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j & 15] = __builtin_fmaf(v[j & 15], m, c);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r] + v[r];
    if (s == 12345.678f) *sink = s;
}

template <int KIND, int NV>
static float run_same(int n, float* sink) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float t = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((probe_same_wave<KIND, NV>), dim3(256), dim3(512), 0, 0, n, sink);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&t, a, b);
    }
    return t;
}

template <int KIND, int VALU>
static void run(const char* name, int n_mfma, int n_valu, float* sink) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float t[3];
    const int cfg[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int c = 0; c < 3; ++c) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL((probe<KIND, VALU>), dim3(256), dim3(512), 0, 0, cfg[c][0], cfg[c][1], n_mfma, n_valu, sink);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&t[c], a, b);
        }
    }
    printf("| %s | %.3f | %.3f | %.3f | %.2f |\n", name, t[0], t[1], t[2], t[2] / (t[0] > t[1] ? t[0] : t[1]));
}

int main() {
    float* sink; (void)hipMalloc(&sink, 4);
    printf("| matrix waves / vector waves (one of each per SIMD) | matrix alone (ms) | vector alone (ms) | both (ms) | both / max |\n|---|---|---|---|---|\n");
    run<0, 0>("v_mfma_f32_32x32x2_f32 / v_fma_f32", 2000, 16000, sink);
    run<0, 1>("v_mfma_f32_32x32x2_f32 / v_exp_f32", 2000, 4000, sink);
    run<1, 0>("v_mfma_f32_32x32x16_bf16 / v_fma_f32", 4000, 16000, sink);
    run<1, 1>("v_mfma_f32_32x32x16_bf16 / v_exp_f32", 4000, 4000, sink);
    printf("\n| one wave kind, two waves per SIMD: 4 MFMAs + NV v_fma_f32 per iteration (1000 iterations; ms) | NV = 0 | 16 | 32 | 64 | 128 |\n|---|---|---|---|---|---|\n");
    printf("| v_mfma_f32_32x32x2_f32 (4 x 64 cycles per wave and iteration) | %.3f | %.3f | %.3f | %.3f | %.3f |\n", run_same<0, 0>(1000, sink), run_same<0, 16>(1000, sink),
           run_same<0, 32>(1000, sink), run_same<0, 64>(1000, sink), run_same<0, 128>(1000, sink));
    printf("| v_mfma_f32_32x32x16_bf16 (4 x 32 cycles) | %.3f | %.3f | %.3f | %.3f | %.3f |\n", run_same<1, 0>(2000, sink), run_same<1, 16>(2000, sink),
           run_same<1, 32>(2000, sink), run_same<1, 64>(2000, sink), run_same<1, 128>(2000, sink));
    return 0;
}
