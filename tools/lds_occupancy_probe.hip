// How many workgroups of 256 threads are resident per CU as a function of their LDS size?  (MI355X: 160 KB of LDS per CU.)
// 512 workgroups (2 per CU) each wait a fixed time; the launch takes 1x that time if two fit on a CU, 2x if only one does.
//   hipcc --offload-arch=gfx950 -O2 tools/lds_occupancy_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void spin_kernel(long ticks, int* sink) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const long t0 = wall_clock64();                     // 100 MHz constant clock
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (lds[(threadIdx.x + 1) & 255] < 0.f) *sink = 1;
}

int main() {
    int* sink; hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long ticks = 20000;                           // 200 us
    const int sizes[] = {16, 32, 40, 48, 52, 56, 60, 64, 66, 70, 72, 76, 80, 96, 128, 160};
    printf("| LDS per workgroup (KB) | 512 workgroups of 200 us (ms) | 768 workgroups (ms) | resident per CU |\n|---|---|---|---|\n");
    for (int kb : sizes) {
        const size_t bytes = (size_t)kb * 1024;
        if (hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { printf("| %d | refused | | |\n", kb); continue; }
        float ms[2] = {0.f, 0.f};
        for (int v = 0; v < 2; ++v) {
            const int blocks = v ? 768 : 512;
            hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), bytes, 0, ticks, sink);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), bytes, 0, ticks, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms[v], a, b);
        }
        const char* res = ms[1] < 0.3f ? ">= 3" : ms[0] < 0.3f ? "2" : "1";
        printf("| %d | %.3f | %.3f | %s |\n", kb, ms[0], ms[1], res);
    }
    return 0;
}
