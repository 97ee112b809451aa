// Hardware probe (developer tool): what v_mfma_f32_16x16x4_f32 sustains on this chip under a full-chip load, and what each
// LDS operand fetch costs it -- the practical ceiling the conv kernels are priced against (157.3 TFLOP/s = 2.4 GHz, all CUs).
// Each wave runs stages of 4 independent MFMAs; the operands of stage i+1 are fetched from LDS (NI instructions of WIDTH
// dwords, conflict-free) before stage i's MFMAs are issued, exactly like the conv main loops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// RANDOM: operands are pseudo-random in [-1, 1) instead of a handful of tiny constants -- the data toggling of real
// activations, which is what the power management reacts to
__device__ __forceinline__ float fill_value(int i, int random) {
    unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    return random ? (float)(int)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f : (float)(i & 7) * 1e-3f;
}

template <int NI, int WIDTH, int RANDOM = 0>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    __shared__ float s[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) s[i] = fill_value(i, RANDOM);
    __syncthreads();
    f32x4 acc[4];
    for (int r = 0; r < 4; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int ND = NI * WIDTH;                              // dwords per stage (<= 8)
    float v[2][8];
    for (int i = 0; i < 8; ++i) v[0][i] = v[1][i] = 1e-3f * (lane + i);
    const float* p = s + lane * WIDTH;
    auto fetch = [&](float (&d)[8], int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float* q = p + ((st * NI + i) * 64 * WIDTH & 8191);
            if (WIDTH == 1) d[i] = q[0];
            if (WIDTH == 2) { const f32x2 t = *reinterpret_cast<const f32x2*>(q); d[2 * i] = t.x; d[2 * i + 1] = t.y; }
            if (WIDTH == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(q); d[4 * i] = t.x; d[4 * i + 1] = t.y; d[4 * i + 2] = t.z; d[4 * i + 3] = t.w; }
        }
    };
    if (ND) fetch(v[0], 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (ND) fetch(v[(u + 1) & 1], u + 1 + (it & 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[u & 1][r], v[u & 1][4 + r], acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = 0.f;
    for (int r = 0; r < 4; ++r) t += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
    if (t == 123.456f) out[0] = t;
}

// 32x32x2: 16 passes, same FLOP rate, a quarter of the operand dwords per FLOP
template <int NI>
__global__ void __launch_bounds__(512) k32(float* out, int iters) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    __shared__ float s[16384];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) s[i] = (float)(i & 7) * 1e-3f;
    __syncthreads();
    f32x16 acc[2];
    for (int r = 0; r < 2; ++r) for (int j = 0; j < 16; ++j) acc[r][j] = 0.f;
    float v[2][4];
    for (int i = 0; i < 4; ++i) v[0][i] = v[1][i] = 1e-3f * (lane + i);
    const float* p = s + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < NI; ++i) v[(u + 1) & 1][i] = p[((u + 1 + (it & 1)) * NI + i) * 64 & 8191];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][r], v[u & 1][2 + r], acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float t = 0.f;
    for (int r = 0; r < 2; ++r) for (int j = 0; j < 16; ++j) t += acc[r][j];
    if (t == 123.456f) out[0] = t;
}

template <typename K>
static void run(const char* name, K kern, int waves_per_simd, int iters) {
    float* out; (void)hipMalloc(&out, 4);
    const int blocks = 256 * waves_per_simd / 2;          // 512 threads = 8 waves = 2 per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    float sum = 0.f;
    const int reps = 5;
    for (int i = 0; i < reps; ++i) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        sum += ms;
    }
    const double flops = 2.0 * 16 * 16 * 4 * 64.0 * iters * blocks * 8;
    printf("%-34s waves/SIMD %d: %.3f ms %.1f TF\n", name, waves_per_simd, sum / reps, flops / (sum / reps) / 1e9);
    (void)hipFree(out);
}

int main() {
    const int it = 4000;
    for (int w = 2; w <= 4; w += 2) {
        run("no LDS", k<0, 1>, w, it);
        run("2 x b32 per 4 MFMA", k<2, 1>, w, it);
        run("4 x b32 per 4 MFMA", k<4, 1>, w, it);
        run("8 x b32 per 4 MFMA", k<8, 1>, w, it);
        run("2 x b64 per 4 MFMA (4 dw)", k<2, 2>, w, it);
        run("4 x b64 per 4 MFMA (8 dw)", k<4, 2>, w, it);
        run("1 x b128 per 4 MFMA (4 dw)", k<1, 4>, w, it);
        run("2 x b128 per 4 MFMA (8 dw)", k<2, 4>, w, it);
        run("8 x b32 per 4 MFMA, random data", k<8, 1, 1>, w, it);
        run("2 x b128 per 4 MFMA, random data", k<2, 4, 1>, w, it);
        run("32x32x2: 2 x b32 per 2 MFMA(=4)", k32<2>, w, it);
        run("32x32x2: 4 x b32 per 2 MFMA(=4)", k32<4>, w, it);
    }
    return 0;
}
