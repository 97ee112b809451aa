// gfx950 probe: lane pattern of v_permlane16_swap_b32 (__builtin_amdgcn_permlane16_swap).  Each lane passes vdst = lane,
// src0 = 1000 + lane; prints what every lane holds afterwards.  hipcc --offload-arch=gfx950 -O2 permlane_swap_probe.hip && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x * 2] = r.x;
    out[threadIdx.x * 2 + 1] = r.y;
}
int main() {
    unsigned* d;
    (void)hipMalloc(&d, 512);
    k<<<1, 64>>>(d);
    unsigned h[128];
    (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; i += 4) printf("lane %2d: vdst'=%4u src0'=%4u\n", i, h[2 * i], h[2 * i + 1]);
    return 0;
}
