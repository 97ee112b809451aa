// Pooling / upsampling / activation-gradient glue kernels of the U-Net (HBM-bound, elementwise).
//
// Replaces (paths relative to the reference root):
//   voxelmorph/torch/networks.py:83-84,130   MaxPool3d(2)            (max_pool3d_with_indices)
//   voxelmorph/torch/networks.py:85,137-138  Upsample(2,'nearest') + cat   (backward side; the
//                                            forward side is folded into the conv gather)
//   voxelmorph/torch/networks.py:300,304     LeakyReLU(0.2) backward (leaky_relu_backward)
// and fuses each backward with the leaky_relu_backward of the ConvBlock that produced the tensor.
#include "vxm_common.h"
#include "vxm_device.h"

namespace {

__global__ void __launch_bounds__(256) k_lrelu_bwd(const float* __restrict__ g, long long g_bs, const float* __restrict__ y, long long y_bs,
                                                   float* __restrict__ dz, long long dz_bs, float slope, long long n /* C*V */) {
    const size_t b = blockIdx.y;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        dz[b * dz_bs + i] = g[b * g_bs + i] * vxm_lrelu_grad(y[b * y_bs + i], slope);
}

// one thread per pooled voxel.  CODE (round 6): also one 16-bit word per pooled voxel and channel that holds everything the backward pass reads of
// the 2x2x2 block -- bit k (k = 4 dz + 2 dy + dx): x[k] > 0 (LeakyReLU' of the pooled tensor), bits 8..10: the arg-max the gradient is routed to
// (first maximum in ATen's scan order, a NaN wins) -- so that the consumer of the block's gradient can form it on the fly
// (vxm_conv3d_k3_fewch_bwd_weight_pool) instead of reading one written by vxm_maxpool2_bwd.
template <bool CODE>
__global__ void __launch_bounds__(256) k_maxpool2_fwd(const float* __restrict__ x, long long x_bs, float* __restrict__ y, unsigned short* __restrict__ code,
                                                      int C, int D, int H, int W) {
    const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1;
    const long long V2 = (long long)D2 * H2 * W2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= V2 * C) return;
    const size_t b = blockIdx.y;
    const int c = (int)(i / V2);
    const int q = (int)(i - (long long)c * V2);
    const int w = q % W2, t = q / W2, h = t % H2, d = t / H2;
    const float* p = x + b * x_bs + ((size_t)c * D + 2 * d) * H * W + (size_t)(2 * h) * W + 2 * w;
    float m = p[0];
    unsigned arg = 0u, pos = m > 0.0f ? 1u : 0u;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const float v = p[(size_t)((k >> 2) & 1) * H * W + (size_t)((k >> 1) & 1) * W + (k & 1)];
        if (v > m || v != v) { m = v; arg = (unsigned)k; }   // ATen: (val > maxval) || isnan(val)
        if (CODE) pos |= (v > 0.0f ? 1u : 0u) << k;
    }
    y[b * (size_t)C * V2 + i] = m;
    if (CODE) code[b * (size_t)C * V2 + i] = (unsigned short)(pos | (arg << 8));
}

// one thread per pooled voxel: routes gpool to the first arg-max of its 2x2x2 block, adds the
// skip-branch gradient and applies LeakyReLU' of the pooled tensor (= ConvBlock output).
__global__ void __launch_bounds__(256) k_maxpool2_bwd(const float* __restrict__ x, long long x_bs, const float* __restrict__ gpool,
                                                      const float* __restrict__ gskip, long long gs_bs, float* __restrict__ dz,
                                                      float slope, int C, int D, int H, int W) {
    const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1;
    const long long V2 = (long long)D2 * H2 * W2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= V2 * C) return;
    const size_t b = blockIdx.y;
    const int c = (int)(i / V2);
    const int q = (int)(i - (long long)c * V2);
    const int w = q % W2, t = q / W2, h = t % H2, d = t / H2;
    const size_t off = ((size_t)c * D + 2 * d) * H * W + (size_t)(2 * h) * W + 2 * w;
    const float* p = x + b * x_bs + off;
    float vals[8];
    float m = p[0];
    int arg = 0;
    vals[0] = m;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const float v = p[(size_t)((k >> 2) & 1) * H * W + (size_t)((k >> 1) & 1) * W + (k & 1)];
        vals[k] = v;
        if (v > m || v != v) { m = v; arg = k; }
    }
    const float gp = gpool[b * (size_t)C * V2 + i];
    const size_t V = (size_t)D * H * W;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t o = off + (size_t)((k >> 2) & 1) * H * W + (size_t)((k >> 1) & 1) * W + (k & 1);
        float g = (k == arg) ? gp : 0.0f;
        if (gskip) g += gskip[b * gs_bs + o];
        dz[b * (size_t)C * V + o] = g * vxm_lrelu_grad(vals[k], slope);
    }
}

// The volume may have odd extents (MaxPool floors): voxels outside the pooled region get only the
// skip gradient.  Handled by a second tiny kernel over the uncovered border.
// The same for two W-neighbouring pooled voxels per thread: every row piece is one aligned 16-byte access (W % 4 == 0, 16-byte
// aligned tensors), i.e. 1 KB per wave instruction instead of 64 strided 4-byte words -- the kernel is pure HBM traffic
// (1.37 GB per pair at full resolution).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_maxpool2_bwd_v4(const float* __restrict__ x, long long x_bs, const float* __restrict__ gpool,
                                                         const float* __restrict__ gskip, long long gs_bs, float* __restrict__ dz,
                                                         float slope, int C, int D, int H, int W) {
    const int D2 = D >> 1, H2 = H >> 1, W4 = W >> 2;
    const long long n = (long long)D2 * H2 * W4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * C) return;
    const size_t b = blockIdx.y;
    const int c = (int)(i / n);
    const int q = (int)(i - (long long)c * n);
    const int w4 = q % W4, t = q / W4, h = t % H2, d = t / H2;
    const size_t off = ((size_t)c * D + 2 * d) * H * W + (size_t)(2 * h) * W + 4 * w4;
    const size_t V = (size_t)D * H * W;
    f32x4v xv[4], gs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t o = off + (size_t)(r >> 1) * H * W + (size_t)(r & 1) * W;
        xv[r] = *reinterpret_cast<const f32x4v*>(x + b * x_bs + o);
        gs[r] = gskip ? *reinterpret_cast<const f32x4v*>(gskip + b * gs_bs + o) : (f32x4v){0.f, 0.f, 0.f, 0.f};
    }
    const f32x2v gp = *reinterpret_cast<const f32x2v*>(gpool + b * (size_t)C * (V >> 3) + ((size_t)c * D2 + d) * H2 * (W >> 1) + (size_t)h * (W >> 1) + 2 * w4);
    f32x4v out[4];
#pragma unroll
    for (int half = 0; half < 2; ++half) {              // the two pooled voxels: columns 2 half, 2 half + 1 of the row pieces
        float m = xv[0][2 * half];
        int arg = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k) {                   // ATen scan order (d, h, w): first maximum wins, NaN propagates
            const float v = xv[k >> 1][2 * half + (k & 1)];
            if (v > m || v != v) { m = v; arg = k; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = xv[k >> 1][2 * half + (k & 1)];
            const float g = (k == arg ? gp[half] : 0.0f) + gs[k >> 1][2 * half + (k & 1)];
            out[k >> 1][2 * half + (k & 1)] = g * vxm_lrelu_grad(v, slope);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x4v*>(dz + b * (size_t)C * V + off + (size_t)(r >> 1) * H * W + (size_t)(r & 1) * W) = out[r];
}

__global__ void __launch_bounds__(256) k_maxpool2_bwd_border(const float* __restrict__ x, long long x_bs, const float* __restrict__ gskip,
                                                             long long gs_bs, float* __restrict__ dz, float slope, int C, int D, int H, int W) {
    const long long V = (long long)D * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= V * C) return;
    const size_t b = blockIdx.y;
    const int q = (int)(i % V);
    const int w = q % W, t = q / W, h = t % H, d = t / H;
    if (d < (D & ~1) && h < (H & ~1) && w < (W & ~1)) return;
    const float g = gskip ? gskip[b * gs_bs + i] : 0.0f;
    dz[b * (size_t)C * V + i] = g * vxm_lrelu_grad(x[b * x_bs + i], slope);
}

// one thread per low-res voxel
__global__ void __launch_bounds__(256) k_upsample2_bwd(const float* __restrict__ g, long long g_bs, const float* __restrict__ y,
                                                       float* __restrict__ dz, float slope, int C, int D, int H, int W) {
    const long long V = (long long)D * H * W;     // low-res
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= V * C) return;
    const size_t b = blockIdx.y;
    const int c = (int)(i / V);
    const int q = (int)(i - (long long)c * V);
    const int w = q % W, t = q / W, h = t % H, d = t / H;
    const int H2 = 2 * H, W2 = 2 * W;
    const float* p = g + b * g_bs + ((size_t)c * 2 * D + 2 * d) * H2 * W2 + (size_t)(2 * h) * W2 + 2 * w;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += p[(size_t)((k >> 2) & 1) * H2 * W2 + (size_t)((k >> 1) & 1) * W2 + (k & 1)];
    const float m = y ? vxm_lrelu_grad(y[b * (size_t)C * V + i], slope) : 1.0f;
    dz[b * (size_t)C * V + i] = s * m;
}

// one thread per output voxel of cat([upsample2(x0), x1])
__global__ void __launch_bounds__(256) k_upsample2_cat(const float* __restrict__ x0, int C0, const float* __restrict__ x1, int C1,
                                                       float* __restrict__ out, int D, int H, int W) {
    const long long V = (long long)D * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= V * (C0 + C1)) return;
    const size_t b = blockIdx.y;
    const int c = (int)(i / V);
    const int q = (int)(i - (long long)c * V);
    float v;
    if (c < C0) {
        const int w = q % W, t = q / W, h = t % H, d = t / H;
        const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1;
        v = x0[b * (size_t)C0 * D2 * H2 * W2 + ((size_t)c * D2 + (d >> 1)) * H2 * W2 + (size_t)(h >> 1) * W2 + (w >> 1)];
    } else {
        v = x1[b * (size_t)C1 * V + (size_t)(c - C0) * V + q];
    }
    out[b * (size_t)(C0 + C1) * V + i] = v;
}

// ---- general pooling factors: Unet(max_pool=k or a list) of the reference (networks.py:79-85): MaxPoolNd(k) -- kernel = stride = k, no
// padding, floor -- at :130 and Upsample(scale_factor=k, 'nearest') + cat at :137-138.  Elementwise, one thread per voxel, per-axis factors
// (a 2-D image is a volume of depth 1 with kd = 1).  The U-Net of VxmDense (k = 2) runs the fused kernels above; these serve the op-by-op path.
__global__ void __launch_bounds__(256) k_maxpoolk_fwd(const float* __restrict__ x, float* __restrict__ y, long long n_out, int D, int H, int W, int Do, int Ho,
                                                      int Wo, int kd, int kh, int kw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const int wo = (int)(i % Wo); long long t = i / Wo;
    const int ho = (int)(t % Ho); t /= Ho;
    const int dq = (int)(t % Do); const long long bc = t / Do;
    const float* p = x + ((bc * D + (long long)dq * kd) * H + (long long)ho * kh) * W + (long long)wo * kw;
    float m = p[0];
    for (int a = 0; a < kd; ++a)
        for (int b = 0; b < kh; ++b)
            for (int c = 0; c < kw; ++c) {
                const float v = p[((long long)a * H + b) * W + c];
                m = (v > m || v != v) ? v : m;      // ATen: (val > maxval) || isnan(val)
            }
    y[i] = m;
}
// one thread per INPUT voxel: repeats the forward scan of its window and takes the gradient when it is the arg-max the scan ends on
__global__ void __launch_bounds__(256) k_maxpoolk_bwd(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, long long n_in, int D, int H,
                                                      int W, int Do, int Ho, int Wo, int kd, int kh, int kw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in) return;
    const int w = (int)(i % W); long long t = i / W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D); const long long bc = t / D;
    const int dq = d / kd, ho = h / kh, wo = w / kw;
    float g = 0.0f;
    if (dq < Do && ho < Ho && wo < Wo) {            // (voxels past the last full window belong to no output)
        const float* p = x + ((bc * D + (long long)dq * kd) * H + (long long)ho * kh) * W + (long long)wo * kw;
        const int mine = ((d - dq * kd) * kh + (h - ho * kh)) * kw + (w - wo * kw);
        float m = p[0];
        int arg = 0, idx = 0;
        for (int a = 0; a < kd; ++a)
            for (int b = 0; b < kh; ++b)
                for (int c = 0; c < kw; ++c, ++idx) {
                    const float v = p[((long long)a * H + b) * W + c];
                    if (v > m || v != v) { m = v; arg = idx; }
                }
        if (arg == mine) g = gy[((bc * Do + dq) * Ho + ho) * (long long)Wo + wo];
    }
    gx[i] = g;
}
// y [B][C0 + C1][D][H][W] = cat([nearest upsampling of x0 [B][C0][D/kd][H/kh][W/kw], x1 [B][C1][D][H][W]])
__global__ void __launch_bounds__(256) k_upsamplek_cat(const float* __restrict__ x0, int C0, const float* __restrict__ x1, int C1, float* __restrict__ y,
                                                       long long n, int D, int H, int W, int kd, int kh, int kw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int w = (int)(i % W); long long t = i / W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D); t /= D;
    const int c = (int)(t % (C0 + C1)); const long long b = t / (C0 + C1);
    const int Dl = D / kd, Hl = H / kh, Wl = W / kw;
    y[i] = c < C0 ? x0[(((b * C0 + c) * Dl + d / kd) * Hl + h / kh) * (long long)Wl + w / kw]
                  : x1[(((b * C1 + (c - C0)) * D + d) * H + h) * (long long)W + w];
}
// gx0 [B][C0][D/kd][H/kh][W/kw] = sum of gy[:, :C0] over each voxel's kd x kh x kw children (fixed order)
__global__ void __launch_bounds__(256) k_upsamplek_bwd(const float* __restrict__ gy, int Ctot, int C0, float* __restrict__ gx0, long long n, int D, int H, int W,
                                                       int kd, int kh, int kw) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int Dl = D / kd, Hl = H / kh, Wl = W / kw;
    const int wl = (int)(i % Wl); long long t = i / Wl;
    const int hl = (int)(t % Hl); t /= Hl;
    const int dl = (int)(t % Dl); t /= Dl;
    const int c = (int)(t % C0); const long long b = t / C0;
    const float* p = gy + (((b * Ctot + c) * D + (long long)dl * kd) * H + (long long)hl * kh) * W + (long long)wl * kw;
    float s = 0.0f;
    for (int a = 0; a < kd; ++a)
        for (int bb = 0; bb < kh; ++bb)
            for (int cc = 0; cc < kw; ++cc) s += p[((long long)a * H + bb) * W + cc];
    gx0[i] = s;
}

}  // namespace

extern "C" {

int vxm_lrelu_bwd(const float* g, int64_t g_bstride, const float* y, int64_t y_bstride, float* dz, int64_t dz_bstride, float slope,
                  int B, int C, int64_t V, void* stream) {
    VXM_REQUIRE(g && y && dz, VXM_ERR_NULL_POINTER, "vxm_lrelu_bwd: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && V > 0, VXM_ERR_BAD_SHAPE, "vxm_lrelu_bwd: bad shape");
    const long long n = (long long)C * V;
    const unsigned nb = (unsigned)(vxm_blocks(n, 256) > 65536u ? 65536u : vxm_blocks(n, 256));
    hipLaunchKernelGGL(k_lrelu_bwd, dim3(nb, B), dim3(256), 0, VXM_STREAM(stream), g, (long long)g_bstride, y, (long long)y_bstride, dz,
                       (long long)dz_bstride, slope, n);
    return vxm_check_launch("vxm_lrelu_bwd");
}

int vxm_maxpool2_fwd(const float* x, int64_t x_bstride, float* y, int B, int C, int D, int H, int W, void* stream) {
    VXM_REQUIRE(x && y, VXM_ERR_NULL_POINTER, "vxm_maxpool2_fwd: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && D >= 2 && H >= 2 && W >= 2, VXM_ERR_BAD_SHAPE, "vxm_maxpool2_fwd: bad shape %dx%dx%d", D, H, W);
    const long long n = (long long)C * (D / 2) * (H / 2) * (W / 2);
    hipLaunchKernelGGL(k_maxpool2_fwd<false>, dim3(vxm_blocks(n, 256), B), dim3(256), 0, VXM_STREAM(stream), x, (long long)x_bstride, y, nullptr, C, D, H, W);
    return vxm_check_launch("vxm_maxpool2_fwd");
}

int vxm_maxpool2_fwd_code(const float* x, int64_t x_bstride, float* y, uint16_t* code, int B, int C, int D, int H, int W, void* stream) {
    VXM_REQUIRE(x && y && code, VXM_ERR_NULL_POINTER, "vxm_maxpool2_fwd_code: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && D >= 2 && H >= 2 && W >= 2 && !((D | H | W) & 1), VXM_ERR_BAD_SHAPE,
                "vxm_maxpool2_fwd_code: bad shape %dx%dx%d (even extents: every voxel belongs to a pooled block)", D, H, W);
    const long long n = (long long)C * (D / 2) * (H / 2) * (W / 2);
    hipLaunchKernelGGL(k_maxpool2_fwd<true>, dim3(vxm_blocks(n, 256), B), dim3(256), 0, VXM_STREAM(stream), x, (long long)x_bstride, y, code, C, D, H, W);
    return vxm_check_launch("vxm_maxpool2_fwd_code");
}

int vxm_maxpool2_bwd(const float* x, int64_t x_bstride, const float* gpool, const float* gskip, int64_t gskip_bstride, float* dz,
                     float slope, int B, int C, int D, int H, int W, void* stream) {
    VXM_REQUIRE(x && gpool && dz, VXM_ERR_NULL_POINTER, "vxm_maxpool2_bwd: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && D >= 2 && H >= 2 && W >= 2, VXM_ERR_BAD_SHAPE, "vxm_maxpool2_bwd: bad shape %dx%dx%d", D, H, W);
    const long long n = (long long)C * (D / 2) * (H / 2) * (W / 2);
    auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((W & 3) == 0 && !((D | H) & 1) && a16(x) && a16(gskip) && a16(dz) && (reinterpret_cast<uintptr_t>(gpool) & 7) == 0 && (x_bstride & 3) == 0 &&
        (gskip_bstride & 3) == 0)
        hipLaunchKernelGGL(k_maxpool2_bwd_v4, dim3(vxm_blocks(n / 2, 256), B), dim3(256), 0, VXM_STREAM(stream), x, (long long)x_bstride, gpool, gskip,
                           (long long)gskip_bstride, dz, slope, C, D, H, W);
    else
        hipLaunchKernelGGL(k_maxpool2_bwd, dim3(vxm_blocks(n, 256), B), dim3(256), 0, VXM_STREAM(stream), x, (long long)x_bstride, gpool, gskip,
                           (long long)gskip_bstride, dz, slope, C, D, H, W);
    if ((D | H | W) & 1)
        hipLaunchKernelGGL(k_maxpool2_bwd_border, dim3(vxm_blocks((long long)C * D * H * W, 256), B), dim3(256), 0, VXM_STREAM(stream), x,
                           (long long)x_bstride, gskip, (long long)gskip_bstride, dz, slope, C, D, H, W);
    return vxm_check_launch("vxm_maxpool2_bwd");
}

int vxm_upsample2_bwd(const float* g, int64_t g_bstride, const float* y, float* dz, float slope, int B, int C, int D, int H, int W,
                      void* stream) {
    VXM_REQUIRE(g && dz, VXM_ERR_NULL_POINTER, "vxm_upsample2_bwd: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && D > 0 && H > 0 && W > 0, VXM_ERR_BAD_SHAPE, "vxm_upsample2_bwd: bad shape");
    hipLaunchKernelGGL(k_upsample2_bwd, dim3(vxm_blocks((long long)C * D * H * W, 256), B), dim3(256), 0, VXM_STREAM(stream), g,
                       (long long)g_bstride, y, dz, slope, C, D, H, W);
    return vxm_check_launch("vxm_upsample2_bwd");
}

int vxm_upsample2_cat(const float* x0, int C0, const float* x1, int C1, float* out, int B, int D, int H, int W, void* stream) {
    VXM_REQUIRE(x0 && out && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_upsample2_cat: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C0 > 0 && C1 >= 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, VXM_ERR_BAD_SHAPE, "vxm_upsample2_cat: bad shape");
    hipLaunchKernelGGL(k_upsample2_cat, dim3(vxm_blocks((long long)(C0 + C1) * D * H * W, 256), B), dim3(256), 0, VXM_STREAM(stream), x0, C0,
                       x1, C1, out, D, H, W);
    return vxm_check_launch("vxm_upsample2_cat");
}

int vxm_maxpool3d_k_fwd(const float* x, float* y, int64_t BC, int D, int H, int W, int kd, int kh, int kw, void* stream) {
    VXM_REQUIRE(x && y, VXM_ERR_NULL_POINTER, "vxm_maxpool3d_k_fwd: null pointer");
    VXM_REQUIRE(BC > 0 && kd > 0 && kh > 0 && kw > 0 && D >= kd && H >= kh && W >= kw, VXM_ERR_BAD_SHAPE,
                "vxm_maxpool3d_k_fwd: window %dx%dx%d does not fit %dx%dx%d", kd, kh, kw, D, H, W);
    const long long n = (long long)BC * (D / kd) * (H / kh) * (W / kw);
    hipLaunchKernelGGL(k_maxpoolk_fwd, dim3(vxm_blocks(n, 256)), dim3(256), 0, VXM_STREAM(stream), x, y, n, D, H, W, D / kd, H / kh, W / kw, kd, kh, kw);
    return vxm_check_launch("vxm_maxpool3d_k_fwd");
}

int vxm_maxpool3d_k_bwd(const float* x, const float* gy, float* gx, int64_t BC, int D, int H, int W, int kd, int kh, int kw, void* stream) {
    VXM_REQUIRE(x && gy && gx, VXM_ERR_NULL_POINTER, "vxm_maxpool3d_k_bwd: null pointer");
    VXM_REQUIRE(BC > 0 && kd > 0 && kh > 0 && kw > 0 && D >= kd && H >= kh && W >= kw, VXM_ERR_BAD_SHAPE,
                "vxm_maxpool3d_k_bwd: window %dx%dx%d does not fit %dx%dx%d", kd, kh, kw, D, H, W);
    const long long n = (long long)BC * D * H * W;
    hipLaunchKernelGGL(k_maxpoolk_bwd, dim3(vxm_blocks(n, 256)), dim3(256), 0, VXM_STREAM(stream), x, gy, gx, n, D, H, W, D / kd, H / kh, W / kw, kd, kh, kw);
    return vxm_check_launch("vxm_maxpool3d_k_bwd");
}

int vxm_upsample3d_k_cat(const float* x0, int C0, const float* x1, int C1, float* y, int B, int D, int H, int W, int kd, int kh, int kw, void* stream) {
    VXM_REQUIRE(x0 && y && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_upsample3d_k_cat: null pointer");
    VXM_REQUIRE(B > 0 && C0 > 0 && C1 >= 0 && kd > 0 && kh > 0 && kw > 0 && D > 0 && H > 0 && W > 0 && D % kd == 0 && H % kh == 0 && W % kw == 0,
                VXM_ERR_BAD_SHAPE, "vxm_upsample3d_k_cat: %dx%dx%d is not a multiple of the factors %dx%dx%d", D, H, W, kd, kh, kw);
    const long long n = (long long)B * (C0 + C1) * D * H * W;
    hipLaunchKernelGGL(k_upsamplek_cat, dim3(vxm_blocks(n, 256)), dim3(256), 0, VXM_STREAM(stream), x0, C0, x1, C1, y, n, D, H, W, kd, kh, kw);
    return vxm_check_launch("vxm_upsample3d_k_cat");
}

int vxm_upsample3d_k_bwd(const float* gy, int Ctot, int C0, float* gx0, int B, int D, int H, int W, int kd, int kh, int kw, void* stream) {
    VXM_REQUIRE(gy && gx0, VXM_ERR_NULL_POINTER, "vxm_upsample3d_k_bwd: null pointer");
    VXM_REQUIRE(B > 0 && C0 > 0 && Ctot >= C0 && kd > 0 && kh > 0 && kw > 0 && D > 0 && H > 0 && W > 0 && D % kd == 0 && H % kh == 0 && W % kw == 0,
                VXM_ERR_BAD_SHAPE, "vxm_upsample3d_k_bwd: %dx%dx%d is not a multiple of the factors %dx%dx%d", D, H, W, kd, kh, kw);
    const long long n = (long long)B * C0 * (D / kd) * (H / kh) * (W / kw);
    hipLaunchKernelGGL(k_upsamplek_bwd, dim3(vxm_blocks(n, 256)), dim3(256), 0, VXM_STREAM(stream), gy, Ctot, C0, gx0, n, D, H, W, kd, kh, kw);
    return vxm_check_launch("vxm_upsample3d_k_bwd");
}

}  // extern "C"
