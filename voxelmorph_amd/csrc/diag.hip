// Diagnostics of the fp16-piece convolution engine (csrc/s3_pieces.h): how well does the two-piece representation fit a tensor?
//
// The engine scales every staged tile by ONE power of two that brings the tile's largest magnitude m into [2^14, 2^15) and writes each value
// as h + l (two fp16 numbers).  A value at or above 2^-18 m keeps 22 significand bits; below that the low piece reaches fp16's subnormal
// spacing and the value keeps an ABSOLUTE error of up to 2^-40 m instead.  For a tensor x this kernel accumulates, over the aligned
// 8-channel x 8 x 8 x 16 voxel tiles the forward kernels stage,
//     out[0] += (number of non-zero values of the tile below 2^-18 m) * m^2          out[1] += sum of x^2
//     out[2] += number of non-zero values of the tile below 2^-18 m                    out[3] += number of non-zero values
// sqrt(out[0] * 2^-80 / out[1]) bounds the rel-L2 error of the representation of the TENSOR from that effect (always tiny: the large values carry
// the norm); out[2] / out[3] is the share of values whose own relative precision is degraded -- what matters to a consumer that normalises
// locally, as the windowed NCC does.  voxelmorph_amd.range_report() runs it on every tensor the split kernels read in a training step.
// Not on the hot path: called by the report only.
#include "vxm_common.h"
#include "vxm_device.h"

namespace {

constexpr int RP_TD = 8, RP_TH = 8, RP_TW = 16;

__global__ void __launch_bounds__(256) k_s3_range_probe(const float* __restrict__ x, int C, long long bstride, int blocked, int D, int H, int W,
                                                        double* __restrict__ out) {
    __shared__ float red[4];
    __shared__ double dred[3][4];
    const int ntw = (W + RP_TW - 1) / RP_TW, nth = (H + RP_TH - 1) / RP_TH;
    int t = blockIdx.x;
    const int w0 = (t % ntw) * RP_TW; t /= ntw;
    const int h0 = (t % nth) * RP_TH;
    const int d0 = (t / nth) * RP_TD;
    const int cb = blockIdx.y, b = blockIdx.z;
    const long long V = (long long)D * H * W;
    const float* xb = x + (size_t)b * bstride;
    // 8 channels x 1024 voxels on 256 threads: 32 values per thread, kept in registers for the second pass
    float v[32];
    float m = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int e = threadIdx.x + 256 * i;                      // (channel 8, voxel 1024)
        const int c = e >> 10, vox = e & 1023;
        const int dz = vox >> 7, hy = (vox >> 4) & 7, wx = vox & 15;
        const int d = d0 + dz, h = h0 + hy, w = w0 + wx, ch = cb * 8 + c;
        float val = 0.0f;
        if (d < D && h < H && w < W && ch < C) {
            const long long p = ((long long)d * H + h) * W + w;
            val = blocked ? xb[((size_t)cb * V + p) * 8 + c] : xb[(size_t)ch * V + p];
        }
        v[i] = val;
        m = fmaxf(m, fabsf(val));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float thr = m * 3.814697265625e-06f;                    // 2^-18 m
    double below = 0.0, energy = 0.0, nonzero = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const float a = fabsf(v[i]);
        if (a != 0.0f) nonzero += 1.0;
        if (a != 0.0f && a < thr) below += 1.0;
        energy += (double)v[i] * (double)v[i];
    }
    below = vxm_wave_sum(below);
    energy = vxm_wave_sum(energy);
    nonzero = vxm_wave_sum(nonzero);
    if ((threadIdx.x & 63) == 0) { dred[0][threadIdx.x >> 6] = below; dred[1][threadIdx.x >> 6] = energy; dred[2][threadIdx.x >> 6] = nonzero; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double nb = (dred[0][0] + dred[0][1]) + (dred[0][2] + dred[0][3]);
        const double en = (dred[1][0] + dred[1][1]) + (dred[1][2] + dred[1][3]);
        const double nz = (dred[2][0] + dred[2][1]) + (dred[2][2] + dred[2][3]);
        if (nb != 0.0) { atomicAdd(out, nb * (double)m * (double)m); atomicAdd(out + 2, nb); }
        if (en != 0.0) atomicAdd(out + 1, en);
        if (nz != 0.0) atomicAdd(out + 3, nz);
    }
}

}  // namespace

extern "C" {

int vxm_s3_range_probe(const float* x, int C, int64_t bstride, int blocked, int B, int D, int H, int W, double* out, void* stream) {
    VXM_REQUIRE(x && out, VXM_ERR_NULL_POINTER, "vxm_s3_range_probe: null pointer");
    VXM_REQUIRE(C > 0 && B > 0 && D > 0 && H > 0 && W > 0 && B <= 65535 && (C + 7) / 8 <= 65535 && (!blocked || C % 8 == 0), VXM_ERR_BAD_SHAPE,
                "vxm_s3_range_probe: bad shape B=%d C=%d D=%d H=%d W=%d (a channel-blocked tensor carries multiples of 8 channels)", B, C, D, H, W);
    const long long tiles = (long long)((W + RP_TW - 1) / RP_TW) * ((H + RP_TH - 1) / RP_TH) * ((D + RP_TD - 1) / RP_TD);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_s3_range_probe: too many tiles");
    hipLaunchKernelGGL(k_s3_range_probe, dim3((unsigned)tiles, (C + 7) / 8, B), dim3(256), 0, VXM_STREAM(stream), x, C, (long long)bstride, blocked ? 1 : 0,
                       D, H, W, out);
    return vxm_check_launch("vxm_s3_range_probe");
}

}  // extern "C"
