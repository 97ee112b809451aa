// Hardware probe (developer tool): semantics of buffer_load ... lds (LDS-DMA) on gfx950 that the conv
// staging relies on: (1) out-of-range lanes write 0.0 into LDS, (2) M0 bases above 64 KiB work,
// (3) the 16-byte form, (4) whether the instruction offset also moves the LDS address.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void __launch_bounds__(64) probe(const float* __restrict__ x, int n, float* __restrict__ out) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 64) smem[i] = 7.0f;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
    // test 1: odd lanes out of range
    int voff = (lane & 1) ? (int)0x80000000 : lane * 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem), 4, voff, 0, 0, 0);
    // test 2: LDS base at 100 KiB (float index 25600), soffset = 256 bytes
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 25600), 4, lane * 4, 256, 0, 0);
    // test 3: 16-byte form at 120 KiB (float index 30720): lane reads x[4*lane .. 4*lane+3]
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 30720), 16, lane * 16, 0, 0, 0);
    // test 4: instruction offset 64 bytes, LDS base float index 1024
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 1024), 4, lane * 4, 0, 64, 0);
    // test 5: partially out-of-range 16-byte lanes (n = 1000 floats: lane 62 -> floats 992..995 ok? use voff near the end)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(smem + 2048), 16, (n - 130) * 4 + lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 128; i += 64) out[i] = smem[i];                 // test 1
    for (int i = lane; i < 128; i += 64) out[128 + i] = smem[25600 + i];   // test 2
    for (int i = lane; i < 320; i += 64) out[256 + i] = smem[30720 + i];   // test 3 (256 + tail)
    for (int i = lane; i < 128; i += 64) out[576 + i] = smem[1024 + i];    // test 4
    for (int i = lane; i < 320; i += 64) out[704 + i] = smem[2048 + i];    // test 5
}

int main() {
    const int n = 1000;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000.0f + i;
    float *dx, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, 1024 * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, dx, n, dout);
    std::vector<float> o(1024);
    hipError_t e = hipMemcpy(o.data(), dout, 1024 * 4, hipMemcpyDeviceToHost);
    printf("status %s\n", hipGetErrorString(e));
    printf("test1 (odd lanes OOB; expect 1000,0,1002,0...): "); for (int i = 0; i < 8; ++i) printf("%g ", o[i]); printf(" | [64..67]: %g %g %g %g\n", o[64], o[65], o[66], o[67]);
    printf("test2 (base 100KiB, soffset 256B; expect 1064,1065..): "); for (int i = 0; i < 4; ++i) printf("%g ", o[128 + i]); printf("... %g | next: %g\n", o[128 + 63], o[128 + 64]);
    printf("test3 (x4 at 120KiB; expect 1000..1255 linear): "); for (int i = 0; i < 6; ++i) printf("%g ", o[256 + i]); printf("... %g %g | next %g\n", o[256 + 254], o[256 + 255], o[256 + 256]);
    printf("test4 (inst offset 64B, lds idx 1024): at[0..3] "); for (int i = 0; i < 4; ++i) printf("%g ", o[576 + i]); printf("| at[16..19] "); for (int i = 16; i < 20; ++i) printf("%g ", o[576 + i]); printf("| at[64..67] "); for (int i = 64; i < 68; ++i) printf("%g ", o[576 + i]); printf("| at [80..83] "); for (int i = 80; i < 84; ++i) printf("%g ", o[576+i]); printf("\n");
    printf("test5 (x4 tail; floats n-130..): "); for (int i = 120; i < 140; ++i) printf("%g ", o[704 + i]); printf("\n");
    return 0;
}
