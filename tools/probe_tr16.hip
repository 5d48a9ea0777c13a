// Probe of ds_read_b64_tr_b16 (gfx950): prints, for every lane and element, which LDS element index it received
// when lane l supplies the address of element 4*l (linear 8-byte slots).  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    auto v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 512);
    probe<<<1, 64>>>(d);
    uint16_t h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
