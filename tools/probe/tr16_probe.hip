// Probe of ds_read_b64_tr_b16 on gfx950: which LDS element lands in (lane, elem)?
// LDS holds element index i at 16-bit slot i.  Test 1: lane l supplies address 8 l (contiguous chunks).
// Test 2: the address pattern conv_bf16.hip uses for a [voxel][16 channel] tile (32-byte rows):
//   lane l supplies base + (4 (l >> 4) + ((l & 15) >> 2)) * 32 + (l & 3) * 8 and expects, in elem j, channel (l & 15) of
//   voxel 4 (l >> 4) + j, i.e. LDS slot (4 (l >> 4) + j) * 16 + (l & 15).
// Build: hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
    const int off = (4 * (l >> 4) + ((l & 15) >> 2)) * 16 + (l & 3) * 4;
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
    for (int j = 0; j < 4; ++j) { out[l * 8 + j] = a[j]; out[l * 8 + 4 + j] = b[j]; }
}
int main() {
    int* d; int h[512];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d: contiguous ->", l);
        for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 8 + j]);
        printf("   tile ->");
        for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 8 + 4 + j]);
        printf("\n");
        for (int j = 0; j < 4; ++j) {
            bad1 += h[l * 8 + j] != (l & 15) + j * 16 + (l >> 4) * 64;
            bad2 += h[l * 8 + 4 + j] != (4 * (l >> 4) + j) * 16 + (l & 15);
        }
    }
    printf("guide formula lds[(l&15) + 16 j + 64 (l>>4)] with contiguous addresses: %s (%d mismatches)\n", bad1 ? "NO" : "yes", bad1);
    printf("tile pattern gives channel (l&15) of voxel 4 (l>>4) + j: %s (%d mismatches)\n", bad2 ? "NO" : "yes", bad2);
    return 0;
}
