// Stand-alone probe (round 6): is a FOUR-wave 256 x 256 tile kernel - one wave per SIMD, 128 x 128 wave tiles, the 256 accumulator registers in the
// accumulation half of the 512-register file, fragment reads of k-sub-step s + 1 under the MFMAs of sub-step s - faster through its k-loop than
// csrc/gemm.hip's gemm256_kernel (eight waves, 128 x 64 wave tiles, ping-pong R / M slots: 0.45 of the bf16 MFMA peak in the k-loop, 0.32 whole-launch)?
// LDS traffic per k-tile and CU halves (each fragment feeds 4 MFMAs instead of 2 / 4): 128 KB of reads instead of 192 KB.
//   hipcc --offload-arch=gfx950 -O3 tools/gemm4w_probe.hip -o /tmp/g4 && /tmp/g4
// C^T = W A^T as in the library (A operand = W fragment, B operand = A fragment): acc[i][j] = n-tile i x m-tile j.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define DEVINL __device__ __forceinline__
DEVINL rsrc_t make_rsrc(const void* base, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000); }
DEVINL uint4 buf_load16(rsrc_t r, int voff, int soff) { return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
DEVINL f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
DEVINL int swz(int row) { return ((row >> 1) & 7) ^ (((row + 4) >> 3) & 1); }          // csrc/common.h: 128-byte rows, 8 chunks

constexpr int G = 256, LDS_BYTES = 2 * 2 * G * 128;                                   // 2 stages x (A, W) x 256 rows x 128 B = 128 KB

template <int STORE, int NOLOAD = 0>      // NOLOAD: the k-loop on a static LDS image (no global loads, no staging writes): what LDS + MFMA + barrier alone reach
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const u16* __restrict__ A, long lda, const u16* __restrict__ W, long ldw, int M, int N, int K,
                                                        float* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) uint4 dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int tiles_n = N / G;
    int nwg = (M / G) * tiles_n, bid = blockIdx.x;
    { int xcd = bid & 7, q = nwg >> 3, r = nwg & 7; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3); }
    const int m0 = (bid / tiles_n) * G, n0 = (bid % tiles_n) * G;
    const int wm = wave >> 1, wn = wave & 1;
    auto stage = [&](int buf, int op) -> uint4* { return dyn + (buf * 2 + op) * (G * 8); };

    // staging map: thread t, pass p covers LDS chunk (row = p * 32 + t / 8, c' = t % 8) <- global chunk c' ^ swz(row)
    const int srow = tid >> 3, scp = tid & 7;
    int aoff[8], woff[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int row = p * 32 + srow, gc = scp ^ swz(row);
        aoff[p] = (int)((long)(m0 + row) * lda * 2 + gc * 16);
        woff[p] = (int)((long)(n0 + row) * ldw * 2 + gc * 16);
    }
    const rsrc_t ra = make_rsrc(A, (uint32_t)((long)M * lda * 2)), rw = make_rsrc(W, (uint32_t)((long)N * ldw * 2));
    uint4 xa[8], xw[8];
    auto load = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 8; ++p) { xa[p] = buf_load16(ra, aoff[p], kt * 128); xw[p] = buf_load16(rw, woff[p], kt * 128); }
    };
    auto commit = [&](int buf) {
        uint4* sa = stage(buf, 0) + srow * 8 + scp; uint4* sw = stage(buf, 1) + srow * 8 + scp;
#pragma unroll
        for (int p = 0; p < 8; ++p) { sa[p * 32 * 8] = xa[p]; sw[p * 32 * 8] = xw[p]; }
    };
    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;

    const int nk = K / 64;
    load(0);
    commit(0);
    load(nk > 1 ? 1 : 0);
    __syncthreads();
    uint4 wf[2][4], af[2][4];
    for (int kt = 0; kt < nk; ++kt) {
        const uint4* la = stage(kt & 1, 0);
        const uint4* lw = stage(kt & 1, 1);
        auto frags = [&](int s4, int set) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rn = wn * 128 + i * 32 + l31, rm = wm * 128 + i * 32 + l31;
                if (NOLOAD == 2) {                               // timing control: 1 KiB-contiguous (certainly conflict-free) fragment reads, wrong values
                    wf[set][i] = lw[(wn * 4 + i) * 256 + s4 * 64 + lane];
                    af[set][i] = la[(wm * 4 + i) * 256 + s4 * 64 + lane];
                } else {
                    wf[set][i] = lw[rn * 8 + ((s4 * 2 + lh) ^ swz(rn))];
                    af[set][i] = la[rm * 8 + ((s4 * 2 + lh) ^ swz(rm))];
                }
            }
        };
        frags(0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (s4 < 3) frags(s4 + 1, (s4 + 1) & 1);
            else if (NOLOAD == 0 && kt + 1 < nk) commit((kt & 1) ^ 1);    // tile kt + 1 (in flight since the end of the previous iteration) -> the buffer every wave left at the last barrier
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma32(wf[s4 & 1][i], af[s4 & 1][j], acc[i][j]);
            // schedule: the sub-step's LDS operations spread under its 16 MFMAs (the compiler otherwise sinks the reads to their first use)
            if (s4 < 3) {
#pragma unroll
                for (int g = 0; g < 8; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
            }
        }
        if (NOLOAD == 0) load(kt + 2 < nk ? kt + 2 : nk - 1);
        __syncthreads();
    }
    if (STORE) {
        // D layout of v_mfma_f32_32x32x16: lane l holds column l & 31 (= m), rows (= n) 8 (x >> 2) + (x & 3) + 4 (l >> 5)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + wm * 128 + j * 32 + l31;
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int n = n0 + wn * 128 + i * 32 + 8 * (x >> 2) + (x & 3) + 4 * lh;
                    C[(long)m * N + n] = acc[i][j][x];
                }
            }
    } else {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int x = 0; x < 16; ++x) s += acc[i][j][x];
        if (s == 12345.678f) C[0] = s;
    }
}

static u16 f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (u16)(u >> 16); }
static float bf2f(u16 h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int shapes[][3] = {{65536, 768, 768}, {65536, 2304, 768}, {65536, 3072, 768}, {65536, 768, 3072}};
    (void)hipFuncSetAttribute((const void*)gemm4w_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)gemm4w_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<u16> hA((size_t)M * K), hW((size_t)N * K);
        srand(1);
        for (auto& v : hA) v = f2bf((rand() % 2001 - 1000) * 1e-3f);
        for (auto& v : hW) v = f2bf((rand() % 2001 - 1000) * 1e-3f);
        u16 *dA, *dW; float* dC;
        (void)hipMalloc(&dA, hA.size() * 2); (void)hipMalloc(&dW, hW.size() * 2); (void)hipMalloc(&dC, (size_t)M * N * 4);
        (void)hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
        const int grid = (M / G) * (N / G);
        hipLaunchKernelGGL(gemm4w_kernel<1>, dim3(grid), dim3(256), LDS_BYTES, 0, dA, (long)K, dW, (long)K, M, N, K, dC);
        (void)hipDeviceSynchronize();
        std::vector<float> hC(1024);
        double maxerr = 0;
        for (int t = 0; t < 200; ++t) {
            const int m = (int)(((long)rand() * 7919) % M), n = rand() % N;
            float c; (void)hipMemcpy(&c, dC + (size_t)m * N + n, 4, hipMemcpyDeviceToHost);
            double ref = 0; for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
            maxerr = fmax(maxerr, fabs(ref - c));
        }
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(gemm4w_kernel<0>, dim3(grid), dim3(256), LDS_BYTES, 0, dA, (long)K, dW, (long)K, M, N, K, dC);
        (void)hipEventRecord(e0);
        const int reps = 20;
        for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(gemm4w_kernel<0>, dim3(grid), dim3(256), LDS_BYTES, 0, dA, (long)K, dW, (long)K, M, N, K, dC);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12;
        (void)hipFuncSetAttribute((const void*)gemm4w_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gemm4w_kernel<0, 1>), dim3(grid), dim3(256), LDS_BYTES, 0, dA, (long)K, dW, (long)K, M, N, K, dC);
        (void)hipEventRecord(e0);
        for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((gemm4w_kernel<0, 1>), dim3(grid), dim3(256), LDS_BYTES, 0, dA, (long)K, dW, (long)K, M, N, K, dC);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms2; (void)hipEventElapsedTime(&ms2, e0, e1); ms2 /= reps;
        printf("   static LDS image (no operand stream): %.3f ms = %.3f of 2500\n", ms2, 2.0 * M * N * K / (ms2 * 1e-3) / 1e12 / 2500.0);
        (void)hipFuncSetAttribute((const void*)gemm4w_kernel<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gemm4w_kernel<0, 2>), dim3(grid), dim3(256), LDS_BYTES, 0, dA, (long)K, dW, (long)K, M, N, K, dC);
        (void)hipEventRecord(e0);
        for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((gemm4w_kernel<0, 2>), dim3(grid), dim3(256), LDS_BYTES, 0, dA, (long)K, dW, (long)K, M, N, K, dC);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms2, e0, e1); ms2 /= reps;
        printf("   static LDS image, 1 KiB-contiguous fragment reads (wrong values): %.3f ms = %.3f of 2500\n", ms2, 2.0 * M * N * K / (ms2 * 1e-3) / 1e12 / 2500.0);
        printf("M %d N %d K %d: k-loop only %.3f ms = %.0f TFLOP/s = %.3f of 2500; max |err| of 200 entries %.3g\n", M, N, K, ms, tf, tf / 2500.0, maxerr);
        (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dC);
    }
    return 0;
}
