// Hardware probe (developer tool) for the two-piece fp16 split of the fp32 convolutions (csrc/conv_s3.hip, NP = 2):
//  (1) does v_mfma_f32_16x16x32_f16 keep SUBNORMAL fp16 inputs (the low piece of a small value is subnormal)?
//  (2) what does the instruction sustain on the (h, l) pieces of random fp32 data, against the bf16 instruction on (h, m, l) pieces
//      (the chip clocks to its power budget: tools/probe/bf16_mfma_peak.hip)?
//  (3) rounding of the conversion the split uses (round to nearest even expected).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/f16_mfma_probe.hip -o tools/probe/f16_mfma_probe && tools/probe/f16_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned hash(unsigned h) {
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ unsigned pack_f16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, f16x2));
}

// (1) A = 2^-20 (fp16 subnormal) everywhere, B = 2^10: D = 32 * 2^-10 = 0.03125 when the input is kept, 0 when it is flushed.
//     second column block: A = 2^-24 (the smallest subnormal), B = 2^12: 32 * 2^-12
__global__ void k_subnormal(float* out) {
    const unsigned short a_bits = 0x0010;          // 2^-20 = 16 * 2^-24
    const unsigned short b_bits = 0x6400;          // 2^10
    const unsigned short a2_bits = 0x0001, b2_bits = 0x6c00;   // 2^-24, 2^12
    u32x4 a, b, a2, b2;
    for (int e = 0; e < 4; ++e) { a[e] = a_bits | (a_bits << 16); b[e] = b_bits | (b_bits << 16); a2[e] = a2_bits | (a2_bits << 16); b2[e] = b2_bits | (b2_bits << 16); }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    const f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a2), __builtin_bit_cast(f16x8, b2), c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = d[0]; out[1] = d2[0]; }
    // (3) conversion: 1 + 2^-11 is a tie between 1 and 1 + 2^-10 (RNE -> 1); 1 + 3 * 2^-11 is a tie -> 1 + 2^-9 (even); RTZ would give 1 and 1 + 2^-10
    if (threadIdx.x == 0) {
        const unsigned p = pack_f16(1.0f + 0x1p-11f, 1.0f + 3 * 0x1p-11f);
        out[2] = (float)(p & 0xffff); out[3] = (float)(p >> 16);
        const unsigned q = pack_f16(0x1p-20f, -0x1.8p-24f);       // subnormal results of the conversion: 2^-20 exact, -1.5 * 2^-24 -> tie -> -2^-23 (even)
        out[4] = (float)(q & 0xffff); out[5] = (float)(q >> 16);
    }
}

// (2) MODE 0: fp16 (h, l) pieces of random fp32 in [-1, 1) scaled by 2^14, 3 products; MODE 1: bf16 (h, m, l) pieces, 6 products
template <int MODE>
__global__ void __launch_bounds__(512) k_rate(float* out, long long* cyc, int iters) {
    constexpr int NF = 6;
    u32x4 a[NF], b[NF];
    const unsigned base = (blockIdx.x * 512 + threadIdx.x) * 977u;
    for (int f = 0; f < NF; ++f)
        for (int e = 0; e < 4; ++e) {
            unsigned w[2][2];
            for (int o = 0; o < 2; ++o)
                for (int side = 0; side < 2; ++side) {
                    const float x = (float)(int)(hash(base + side * 7777 + f * 64 + e * 2 + o) & 0xffffff) * (1.0f / 8388608.0f) - 1.0f;
                    unsigned short v;
                    if (MODE == 0) {
                        const float xs = x * 16384.0f;
                        const _Float16 h = (_Float16)xs;
                        const _Float16 l = (_Float16)(xs - (float)h);
                        const _Float16 pick = ((f + side) & 1) ? l : h;
                        v = __builtin_bit_cast(unsigned short, pick);
                    } else {
                        auto rne = [](float t) -> unsigned short { unsigned u = __float_as_uint(t); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
                        const unsigned short h = rne(x);
                        const float r1 = x - __uint_as_float((unsigned)h << 16);
                        const unsigned short m = rne(r1);
                        const float r2 = r1 - __uint_as_float((unsigned)m << 16);
                        const unsigned short l = rne(r2);
                        const int pc = (f + side) % 3;
                        v = pc == 0 ? h : pc == 1 ? m : l;
                    }
                    w[side][o] = v;
                }
            a[f][e] = w[0][0] | (w[0][1] << 16);
            b[f][e] = w[1][0] | (w[1][1] << 16);
        }
    const long long t0 = __builtin_readcyclecounter();
    f32x4 acc[4];
    for (int r = 0; r < 4; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (MODE == 0) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[f]), __builtin_bit_cast(f16x8, b[(f + r) % NF]), acc[r], 0, 0, 0);
                else acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[f]), __builtin_bit_cast(bf16x8, b[(f + r) % NF]), acc[r], 0, 0, 0);
            }
    }
    float t = 0.f;
    for (int r = 0; r < 4; ++r) t += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (t == 123.456f) out[0] = t;
}

template <typename K>
static void run(const char* name, K kern, int waves_per_simd, int iters) {
    float* out; long long* cyc; (void)hipMalloc(&out, 64); (void)hipMalloc(&cyc, 8);
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
    const double flops = 2.0 * 16 * 16 * 32 * 4 * 6 * (double)iters * blocks * 8;
    const double ms = sum / reps;
    printf("%-52s waves/SIMD %d: %.3f ms  %7.1f TF  shader clock %.2f GHz\n", name, waves_per_simd, ms, flops / ms / 1e9, (double)c / (ms * 1e6));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    float* out; (void)hipMalloc(&out, 64);
    hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, out);
    float h[6]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("subnormal A input 2^-20 x 2^10 x 32: D = %.8g (kept: 0.03125, flushed: 0)\n", h[0]);
    printf("subnormal A input 2^-24 x 2^12 x 32: D = %.8g (kept: 0.0078125, flushed: 0)\n", h[1]);
    printf("cvt f32->f16 pair: 1+2^-11 -> 0x%04x (RNE 0x3c00), 1+3*2^-11 -> 0x%04x (RNE 0x3c02, RTZ 0x3c01)\n", (unsigned)h[2], (unsigned)h[3]);
    printf("cvt to subnormal: 2^-20 -> 0x%04x (0x0010 kept, 0 flushed), -1.5*2^-24 -> 0x%04x (RNE 0x8002)\n", (unsigned)h[4], (unsigned)h[5]);
    const int it = 3000;
    for (int w = 2; w <= 4; w += 2) {
        run("16x16x32 f16, (h, l) pieces of random fp32 * 2^14", k_rate<0>, w, it);
        run("16x16x32 bf16, (h, m, l) pieces of random fp32", k_rate<1>, w, it);
    }
    return 0;
}
