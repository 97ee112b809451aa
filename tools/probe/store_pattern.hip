// Hardware probe (developer tool): HBM write rate of the conv epilogue's store patterns.  16 channel planes of V floats
// are written once; a wave-wide store instruction covers either 4 channels x 64 B (the MFMA D layout: lane (kq, n) ->
// channel 4 kq + j, 16 consecutive voxels), 2 channels x 128 B (after a permlane32 swap of two neighbouring tiles), or
// 1 channel x 256 B.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int SEG /* floats per contiguous run: 16, 32, 64 */>
__global__ void __launch_bounds__(512) k(float* __restrict__ y, int V) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 8 + (threadIdx.x >> 6);          // one wave per 64 voxels x 16 channels
    const int nw = V / 64;
    if (wave >= nw) return;
    constexpr int CH = 64 / SEG;                                    // channels per instruction
    const int c_in = lane / SEG, n = lane % SEG;
    for (int i = 0; i < 16; ++i) {                                  // 16 instructions: 64 voxels x 16 channels
        // instruction i covers channels [CH * (i % (16 / CH)) ...] and voxel run (i / (16 / CH))
        const int cgrp = i % (16 / CH), run = i / (16 / CH);
        const int co = cgrp * CH + c_in;
        const size_t v = (size_t)wave * 64 + run * SEG + n;
        y[(size_t)co * V + v] = (float)i;
    }
}

// The conv epilogue's real pattern: a wave owns one depth slice of a tile -- 4 rows x 16 voxels (TW16) or 2 rows x 32 voxels (TW32)
// of 16 channel planes [D][H][W]; one store instruction covers 4 channels x one 16-voxel run.  With TW16 the other half of every
// 128-byte line belongs to the neighbouring tile, i.e. to another block running somewhere else at some other time.
template <int TW>
__global__ void __launch_bounds__(512) kconv(float* __restrict__ y, int D, int H, int W) {
    constexpr int ROWS = 64 / TW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4, n = lane & 15;
    const int nw = W / TW, nh = H / ROWS, V = D * H * W;
    int t = blockIdx.x;
    const int w0 = (t % nw) * TW; t /= nw;
    const int h0 = (t % nh) * ROWS; const int d = (t / nh) * 8 + wave;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < ROWS; ++r)
            for (int half = 0; half < TW / 16; ++half)
                y[(size_t)(kq * 4 + j) * V + ((size_t)d * H + h0 + r) * W + w0 + half * 16 + n] = (float)j;
}
template <int TW>
static void runconv(const char* name, float* y, int D, int H, int W) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = (D / 8) * (H / (64 / TW)) * (W / TW);
    hipLaunchKernelGGL(kconv<TW>, dim3(blocks), dim3(512), 0, 0, y, D, H, W);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kconv<TW>, dim3(blocks), dim3(512), 0, 0, y, D, H, W);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, 16.0 * D * H * W * 4 / ms / 1e6);
}

template <int SEG>
static void run(const char* name, float* y, int V) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = V / 64 / 8;
    hipLaunchKernelGGL(k<SEG>, dim3(blocks), dim3(512), 0, 0, y, V);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<SEG>, dim3(blocks), dim3(512), 0, 0, y, V);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, 16.0 * V * 4 / ms / 1e6);
}

int main() {
    const int V = 160 * 192 * 224;
    float* y; (void)hipMalloc(&y, (size_t)16 * V * 4);
    run<16>("4 channels x 64 B / instr", y, V);
    run<32>("2 channels x 128 B / instr", y, V);
    run<64>("1 channel x 256 B / instr", y, V);
    run<16>("4 channels x 64 B / instr", y, V);
    runconv<16>("conv tile 8x4x16 (64 B runs)", y, 160, 192, 224);
    runconv<32>("conv tile 8x2x32 (128 B runs)", y, 160, 192, 224);
    runconv<16>("conv tile 8x4x16 (64 B runs)", y, 160, 192, 224);
    return 0;
}
