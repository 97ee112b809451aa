// Hardware probe (developer tool): what v_mfma_f32_16x16x32_bf16 sustains on this chip under a full-chip load as a function of
// the OPERAND DATA (the power management clocks the chip to its budget: MI355X_MICROARCH.md, DVFS give-back) -- the practical
// ceiling the split-fp32 conv kernels (csrc/conv_s3.hip) are priced against.  Operands live in registers (no LDS traffic);
// every wave cycles through NF distinct fragments so that the inputs of consecutive MFMAs differ (toggling).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/bf16_mfma_peak.hip -o tools/probe/bf16_mfma_peak && tools/probe/bf16_mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hash(unsigned h) {
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return h;
}
// MODE 0: zeros; 1: one small constant; 2: random bf16 values in [-1, 1) (full random mantissas);
// 3: the three pieces (h, m, l) of random fp32 values in [-1, 1), fragment f holds piece f % 3 (what the split kernels feed)
__device__ __forceinline__ unsigned short make_bf16(unsigned seed, int mode, int piece) {
    if (mode == 0) return 0;
    if (mode == 1) return 0x3c00;                                  // 2^-7
    const float x = (float)(int)(hash(seed) & 0xffffff) * (1.0f / 8388608.0f) - 1.0f;
    auto rne = [](float v) -> unsigned short { unsigned u = __float_as_uint(v); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
    if (mode == 2) return rne(x);
    const unsigned short h = rne(x);
    const float r1 = x - __uint_as_float((unsigned)h << 16);
    const unsigned short m = rne(r1);
    const float r2 = r1 - __uint_as_float((unsigned)m << 16);
    const unsigned short l = rne(r2);
    return piece == 0 ? h : piece == 1 ? m : l;
}

template <int MODE, int BIG>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters) {
    constexpr int NF = 6;
    u32x4 a[NF], b[NF];
    const unsigned base = (blockIdx.x * 512 + threadIdx.x) * 977u;
    for (int f = 0; f < NF; ++f)
        for (int e = 0; e < 4; ++e) {
            a[f][e] = make_bf16(base + f * 64 + e * 2, MODE, f % 3) | ((unsigned)make_bf16(base + f * 64 + e * 2 + 1, MODE, f % 3) << 16);
            b[f][e] = make_bf16(base + 7777 + f * 64 + e * 2, MODE, (f + 1) % 3) | ((unsigned)make_bf16(base + 7777 + f * 64 + e * 2 + 1, MODE, (f + 1) % 3) << 16);
        }
    const long long t0 = __builtin_readcyclecounter();
    float t = 0.f;
    if (BIG) {
        f32x16 acc[2];
        for (int r = 0; r < 2; ++r) for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[f]), __builtin_bit_cast(bf16x8, b[(f + r) % NF]), acc[r], 0, 0, 0);
        }
        for (int r = 0; r < 2; ++r) for (int j = 0; j < 16; ++j) t += acc[r][j];
    } else {
        f32x4 acc[4];
        for (int r = 0; r < 4; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[f]), __builtin_bit_cast(bf16x8, b[(f + r) % NF]), acc[r], 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) t += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (t == 123.456f) out[0] = t;
}

template <typename K>
static void run(const char* name, K kern, int waves_per_simd, int iters, int big) {
    float* out; long long* cyc; (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    const int blocks = 256 * waves_per_simd / 2;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    float sum = 0.f; const int reps = 4;
    for (int i = 0; i < reps; ++i) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, cyc, iters);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); sum += ms;
    }
    long long c = 0; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per = big ? 2.0 * 32 * 32 * 16 * 2 : 2.0 * 16 * 16 * 32 * 4;       // flops per (f) iteration of a wave
    const double flops = per * 6 * (double)iters * blocks * 8;
    const double ms = sum / reps;
    printf("%-46s waves/SIMD %d: %.3f ms  %7.1f TF  shader clock %.2f GHz (block 0: %lld cycles)\n", name, waves_per_simd, ms, flops / ms / 1e9,
           (double)c / (ms * 1e6), c);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    const int it = 3000;
    for (int w = 2; w <= 4; w += 2) {
        run("16x16x32 bf16, zero operands", k<0, 0>, w, it, 0);
        run("16x16x32 bf16, one constant", k<1, 0>, w, it, 0);
        run("16x16x32 bf16, random bf16 operands", k<2, 0>, w, it, 0);
        run("16x16x32 bf16, (h, m, l) pieces of random fp32", k<3, 0>, w, it, 0);
        run("32x32x16 bf16, random bf16 operands", k<2, 1>, w, it, 1);
        run("32x32x16 bf16, (h, m, l) pieces of random fp32", k<3, 1>, w, it, 1);
    }
    return 0;
}
