// 2-D (planar) variants of the layer kernels for gfx950: SpatialTransformer / VecInt / ResizeTransform on [B,C,H,W]
// fields with 2-channel flows (channel 0 = row / H displacement, channel 1 = column / W displacement), and the
// MaxPool2d / Upsample(2,'nearest') + cat glue of the 2-D U-Net.
//
// Reference op chains replaced (paths relative to the reference root; the reference is N-D generic):
//   voxelmorph/torch/layers.py:30-48   SpatialTransformer.forward, 2-D: grid_sampler_2d(align_corners=True, zeros)
//   voxelmorph/torch/layers.py:64-68   VecInt.forward
//   voxelmorph/torch/layers.py:85-97   ResizeTransform.forward ('bilinear', align_corners=True)
//   voxelmorph/torch/networks.py:83-85,130,137-138   MaxPool2d(2), Upsample(2,'nearest') + cat
// 2-D slices are at most a few hundred KB: every kernel is one thread per pixel, HBM/L2-bound and launch-dominated;
// the 3x3 convolutions of the 2-D network run on the 3-D MFMA kernels with a depth of one (torch/planar.py).
#include "vxm_common.h"
#include "vxm_device.h"

namespace {

#define VXM_PIXEL_INDEX(H_, W_)                                      \
    const int V = (H_) * (W_);                                       \
    const int p = blockIdx.x * 256 + threadIdx.x;                    \
    if (p >= V) return;                                              \
    const int b = blockIdx.y;                                        \
    const int h = p / (W_), w = p - h * (W_)

// the 4 corners in ATen's order k = 2 dy + dx (nw, ne, sw, se); weights wx * wy as grid_sampler_2d forms them
struct Corners4 {
    int idx[4];
    float w[4];
    bool ok[4];
    float wy[2], wx[2];
};
__device__ __forceinline__ Corners4 corners4(float y, float x, int H, int W) {
    const AxisTaps ay = axis_corners(y, H), ax = axis_corners(x, W);
    Corners4 c;
    c.wy[0] = ay.w0; c.wy[1] = ay.w1; c.wx[0] = ax.w0; c.wx[1] = ax.w1;
    const int iy[2] = {ay.i0, ay.i1}, ix[2] = {ax.i0, ax.i1};
    const bool oy[2] = {ay.ok0, ay.ok1}, ox[2] = {ax.ok0, ax.ok1};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int k = 2 * dy + dx;
            c.idx[k] = iy[dy] * W + ix[dx];
            c.ok[k] = oy[dy] && ox[dx];
            c.w[k] = c.ok[k] ? c.wx[dx] * c.wy[dy] : 0.0f;
        }
    return c;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_warp2d_fwd(const float* __restrict__ src, const float* __restrict__ flow, float* __restrict__ out,
                                                    int C, int H, int W) {
    VXM_PIXEL_INDEX(H, W);
    const float* fl = flow + (size_t)b * 2 * V;
    const float y = vxm_src_coord(h, fl[p], H), x = vxm_src_coord(w, fl[V + p], W);
    const float* s = src + (size_t)b * C * V;
    float* o = out + (size_t)b * C * V + p;
    if (MODE == VXM_INTERP_NEAREST) {
        const float ry = rintf(y), rx = rintf(x);                      // nearbyint: round-half-even
        const bool in = (ry >= 0.0f) & (ry <= (float)(H - 1)) & (rx >= 0.0f) & (rx <= (float)(W - 1));
        const int idx = in ? (int)ry * W + (int)rx : 0;
        for (int c = 0; c < C; ++c) o[(size_t)c * V] = in ? s[(size_t)c * V + idx] : 0.0f;
        return;
    }
    const Corners4 cn = corners4(y, x, H, W);
    for (int c = 0; c < C; ++c) {
        const float* sc = s + (size_t)c * V;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += sc[cn.idx[k]] * cn.w[k];
        o[(size_t)c * V] = acc;
    }
}

// grid_sampler_2d_backward composed with the reference's normalisation chain (voxel units, as the 3-D kernel)
template <int MODE>
__global__ void __launch_bounds__(256) k_warp2d_bwd(const float* __restrict__ src, const float* __restrict__ flow,
                                                    const float* __restrict__ gout, float* __restrict__ gsrc, float* __restrict__ gflow,
                                                    int C, int H, int W) {
    VXM_PIXEL_INDEX(H, W);
    const float* fl = flow + (size_t)b * 2 * V;
    const float y = vxm_src_coord(h, fl[p], H), x = vxm_src_coord(w, fl[V + p], W);
    const float* s = src + (size_t)b * C * V;
    const float* go = gout + (size_t)b * C * V + p;
    float* gs = gsrc ? gsrc + (size_t)b * C * V : nullptr;
    float* gf = gflow ? gflow + (size_t)b * 2 * V + p : nullptr;
    if (MODE == VXM_INTERP_NEAREST) {
        if (gf) { gf[0] = 0.0f; gf[V] = 0.0f; }
        if (gs) {
            const float ry = rintf(y), rx = rintf(x);
            const bool in = (ry >= 0.0f) & (ry <= (float)(H - 1)) & (rx >= 0.0f) & (rx <= (float)(W - 1));
            if (in) {
                const int idx = (int)ry * W + (int)rx;
                for (int c = 0; c < C; ++c) atomicAdd(gs + (size_t)c * V + idx, go[(size_t)c * V]);
            }
        }
        return;
    }
    const Corners4 cn = corners4(y, x, H, W);
    float gy = 0.0f, gx = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float g = go[(size_t)c * V];
        const float* sc = s + (size_t)c * V;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dy = (k >> 1) & 1, dx = k & 1;
            const float vk = cn.ok[k] ? sc[cn.idx[k]] * g : 0.0f;
            gy += (dy ? vk : -vk) * cn.wx[dx];
            gx += (dx ? vk : -vk) * cn.wy[dy];
            if (gs && cn.ok[k]) atomicAdd(gs + (size_t)c * V + cn.idx[k], g * cn.w[k]);
        }
    }
    if (gf) { gf[0] = gy; gf[V] = gx; }
}

// one scaling-and-squaring step: out = v + warp(v, v), v = in * scale (power of two: exact)
__global__ void __launch_bounds__(256) k_vecint2d_step_fwd(const float* __restrict__ in, float scale, float* __restrict__ out, int H, int W) {
    VXM_PIXEL_INDEX(H, W);
    const float* vin = in + (size_t)b * 2 * V;
    const float v0 = vin[p] * scale, v1 = vin[V + p] * scale;
    const Corners4 cn = corners4(vxm_src_coord(h, v0, H), vxm_src_coord(w, v1, W), H, W);
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a0 += (vin[cn.idx[k]] * scale) * cn.w[k];
        a1 += (vin[V + cn.idx[k]] * scale) * cn.w[k];
    }
    float* o = out + (size_t)b * 2 * V + p;
    o[0] = v0 + a0; o[V] = v1 + a1;
}

// backward of one step (see k_vecint_step_bwd in warp.hip for the derivation); gin zeroed by the caller, fp32 atomics
__global__ void __launch_bounds__(256) k_vecint2d_step_bwd(const float* __restrict__ in, float scale, const float* __restrict__ gout,
                                                           float* __restrict__ gin, int H, int W) {
    VXM_PIXEL_INDEX(H, W);
    const float* vin = in + (size_t)b * 2 * V;
    const float* go = gout + (size_t)b * 2 * V + p;
    float* gi = gin + (size_t)b * 2 * V;
    const float v0 = vin[p] * scale, v1 = vin[V + p] * scale;
    const float g0 = go[0], g1 = go[V];
    const Corners4 cn = corners4(vxm_src_coord(h, v0, H), vxm_src_coord(w, v1, W), H, W);
    float gy = g0, gx = g1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = cn.idx[k];
        const int dy = (k >> 1) & 1, dx = k & 1;
        const float sk = cn.ok[k] ? (vin[i] * scale) * g0 + (vin[V + i] * scale) * g1 : 0.0f;
        gy += (dy ? sk : -sk) * cn.wx[dx];
        gx += (dx ? sk : -sk) * cn.wy[dy];
        if (cn.ok[k]) {
            const float wk = cn.w[k] * scale;
            atomicAdd(gi + i, g0 * wk);
            atomicAdd(gi + V + i, g1 * wk);
        }
    }
    atomicAdd(gi + p, gy * scale);
    atomicAdd(gi + V + p, gx * scale);
}

// upsample_bilinear2d(align_corners=True) with the `factor *` rescale before (factor > 1) or after (factor < 1)
__global__ void __launch_bounds__(256) k_resize2d_fwd(const float* __restrict__ x, float* __restrict__ out, int H, int W, int oH, int oW,
                                                      float rh, float rw, float pre, float post) {
    const int oV = oH * oW;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= oV) return;
    const size_t bc = blockIdx.y;
    const int w = p % oW, h = p / oW;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    lin_src(h, rh, H, y0, y1, ly0, ly1);
    lin_src(w, rw, W, x0, x1, lx0, lx1);
    const float* s = x + bc * (size_t)H * W;
    const float v = ly0 * (lx0 * (pre * s[y0 * W + x0]) + lx1 * (pre * s[y0 * W + x1])) +
                    ly1 * (lx0 * (pre * s[y1 * W + x0]) + lx1 * (pre * s[y1 * W + x1]));
    out[bc * (size_t)oV + p] = v * post;
}

// adjoint of the bilinear resize as a scatter (gx zeroed by the caller)
__global__ void __launch_bounds__(256) k_resize2d_bwd(const float* __restrict__ gout, float* __restrict__ gx, int H, int W, int oH, int oW,
                                                      float rh, float rw, float scale) {
    const int oV = oH * oW;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= oV) return;
    const size_t bc = blockIdx.y;
    const int w = p % oW, h = p / oW;
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    lin_src(h, rh, H, y0, y1, ly0, ly1);
    lin_src(w, rw, W, x0, x1, lx0, lx1);
    const float g = gout[bc * (size_t)oV + p] * scale;
    float* o = gx + bc * (size_t)H * W;
    atomicAdd(o + y0 * W + x0, g * (ly0 * lx0)); atomicAdd(o + y0 * W + x1, g * (ly0 * lx1));
    atomicAdd(o + y1 * W + x0, g * (ly1 * lx0)); atomicAdd(o + y1 * W + x1, g * (ly1 * lx1));
}

// MaxPool2d(2): one thread per pooled pixel; ATen's scan order and NaN rule ((val > max) || isnan(val))
__global__ void __launch_bounds__(256) k_maxpool2d_fwd(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W) {
    const int H2 = H >> 1, W2 = W >> 1, V2 = H2 * W2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)V2 * C) return;
    const size_t b = blockIdx.y;
    const int c = (int)(i / V2), q = (int)(i - (long long)c * V2);
    const int w = q % W2, h = q / W2;
    const float* p = x + (b * C + c) * (size_t)H * W + (size_t)(2 * h) * W + 2 * w;
    float m = p[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const float v = p[(k >> 1) * W + (k & 1)];
        m = (v > m || v != v) ? v : m;
    }
    y[b * (size_t)C * V2 + i] = m;
}

// gradient to the first arg-max of each 2x2 block, zero elsewhere (gx zeroed by the caller when H or W is odd)
__global__ void __launch_bounds__(256) k_maxpool2d_bwd(const float* __restrict__ x, const float* __restrict__ gpool, float* __restrict__ gx,
                                                       int C, int H, int W) {
    const int H2 = H >> 1, W2 = W >> 1, V2 = H2 * W2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)V2 * C) return;
    const size_t b = blockIdx.y;
    const int c = (int)(i / V2), q = (int)(i - (long long)c * V2);
    const int w = q % W2, h = q / W2;
    const size_t off = (b * C + c) * (size_t)H * W + (size_t)(2 * h) * W + 2 * w;
    float m = x[off];
    int arg = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const float v = x[off + (k >> 1) * W + (k & 1)];
        if (v > m || v != v) { m = v; arg = k; }
    }
    const float gp = gpool[b * (size_t)C * V2 + i];
#pragma unroll
    for (int k = 0; k < 4; ++k) gx[off + (k >> 1) * W + (k & 1)] = (k == arg) ? gp : 0.0f;
}

// out = cat([upsample_nearest2d(x0, 2), x1], dim=1): one thread per output element
__global__ void __launch_bounds__(256) k_upsample2d_cat(const float* __restrict__ x0, int C0, const float* __restrict__ x1, int C1,
                                                        float* __restrict__ out, int H, int W) {
    const int V = H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)V * (C0 + C1)) return;
    const size_t b = blockIdx.y;
    const int c = (int)(i / V), q = (int)(i - (long long)c * V);
    float v;
    if (c < C0) {
        const int w = q % W, h = q / W, H2 = H >> 1, W2 = W >> 1;
        v = x0[(b * C0 + c) * (size_t)H2 * W2 + (size_t)(h >> 1) * W2 + (w >> 1)];
    } else {
        v = x1[(b * C1 + (c - C0)) * (size_t)V + q];
    }
    out[b * (size_t)(C0 + C1) * V + i] = v;
}

// gradient of the upsampled segment: sum over the 2x2 children of the first C0 channels of g ([B, Ctot, 2H, 2W])
__global__ void __launch_bounds__(256) k_upsample2d_bwd(const float* __restrict__ g, int Ctot, float* __restrict__ gx0, int C0, int H, int W) {
    const int V = H * W;     // low-res
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)V * C0) return;
    const size_t b = blockIdx.y;
    const int c = (int)(i / V), q = (int)(i - (long long)c * V);
    const int w = q % W, h = q / W, W2 = 2 * W;
    const float* p = g + (b * Ctot + c) * (size_t)4 * V + (size_t)(2 * h) * W2 + 2 * w;
    gx0[b * (size_t)C0 * V + i] = (p[0] + p[1]) + (p[W2] + p[W2 + 1]);
}

int check_plane(const char* fn, int B, int C, int H, int W) {
    VXM_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, VXM_ERR_BAD_SHAPE, "%s: bad shape B=%d C=%d H=%d W=%d (2-D images with every extent > 1)", fn, B, C, H, W);
    VXM_REQUIRE((long long)C * H * W < (1ll << 31) && B <= 65535, VXM_ERR_BAD_SHAPE, "%s: per-sample element count must fit int32, B <= 65535", fn);
    return VXM_OK;
}

}  // namespace

extern "C" {

int vxm_warp2d_fwd(const float* src, const float* flow, float* out, int B, int C, int H, int W, int mode, void* stream) {
    if (int e = check_plane("vxm_warp2d_fwd", B, C, H, W)) return e;
    VXM_REQUIRE(src && flow && out, VXM_ERR_NULL_POINTER, "vxm_warp2d_fwd: null pointer");
    VXM_REQUIRE(mode == VXM_INTERP_LINEAR || mode == VXM_INTERP_NEAREST, VXM_ERR_UNSUPPORTED,
                "vxm_warp2d_fwd: mode %d (only 'bilinear' and 'nearest', layers.py:11)", mode);
    const dim3 grid(vxm_blocks((long long)H * W, 256), B);
    if (mode == VXM_INTERP_NEAREST) hipLaunchKernelGGL(k_warp2d_fwd<VXM_INTERP_NEAREST>, grid, dim3(256), 0, VXM_STREAM(stream), src, flow, out, C, H, W);
    else hipLaunchKernelGGL(k_warp2d_fwd<VXM_INTERP_LINEAR>, grid, dim3(256), 0, VXM_STREAM(stream), src, flow, out, C, H, W);
    return vxm_check_launch("vxm_warp2d_fwd");
}

int vxm_warp2d_bwd(const float* src, const float* flow, const float* gout, float* gsrc, float* gflow, int B, int C, int H, int W, int mode,
                   void* stream) {
    if (int e = check_plane("vxm_warp2d_bwd", B, C, H, W)) return e;
    VXM_REQUIRE(src && flow && gout, VXM_ERR_NULL_POINTER, "vxm_warp2d_bwd: null pointer");
    VXM_REQUIRE(mode == VXM_INTERP_LINEAR || mode == VXM_INTERP_NEAREST, VXM_ERR_UNSUPPORTED, "vxm_warp2d_bwd: mode %d", mode);
    if (!gsrc && !gflow) return VXM_OK;
    if (gsrc) (void)hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)B * C * H * W, VXM_STREAM(stream));
    const dim3 grid(vxm_blocks((long long)H * W, 256), B);
    if (mode == VXM_INTERP_NEAREST)
        hipLaunchKernelGGL(k_warp2d_bwd<VXM_INTERP_NEAREST>, grid, dim3(256), 0, VXM_STREAM(stream), src, flow, gout, gsrc, gflow, C, H, W);
    else
        hipLaunchKernelGGL(k_warp2d_bwd<VXM_INTERP_LINEAR>, grid, dim3(256), 0, VXM_STREAM(stream), src, flow, gout, gsrc, gflow, C, H, W);
    return vxm_check_launch("vxm_warp2d_bwd");
}

int vxm_vecint2d_fwd(const float* vec, float* steps, int B, int H, int W, int nsteps, void* stream) {
    if (int e = check_plane("vxm_vecint2d_fwd", B, 2, H, W)) return e;
    VXM_REQUIRE(nsteps >= 1 && nsteps < 31, VXM_ERR_BAD_SHAPE, "vxm_vecint2d_fwd: nsteps should be >= 1, found: %d", nsteps);
    VXM_REQUIRE(vec && steps, VXM_ERR_NULL_POINTER, "vxm_vecint2d_fwd: null pointer");
    const size_t n = (size_t)B * 2 * H * W;
    const dim3 grid(vxm_blocks((long long)H * W, 256), B);
    const float scale = 1.0f / (float)(1u << nsteps);
    for (int k = 0; k < nsteps; ++k)
        hipLaunchKernelGGL(k_vecint2d_step_fwd, grid, dim3(256), 0, VXM_STREAM(stream), k == 0 ? vec : steps + (size_t)(k - 1) * n,
                           k == 0 ? scale : 1.0f, steps + (size_t)k * n, H, W);
    return vxm_check_launch("vxm_vecint2d_fwd");
}

int vxm_vecint2d_bwd(const float* vec, const float* steps, const float* gout, float* gvec, float* work, int B, int H, int W, int nsteps,
                     void* stream) {
    if (int e = check_plane("vxm_vecint2d_bwd", B, 2, H, W)) return e;
    VXM_REQUIRE(nsteps >= 1 && nsteps < 31, VXM_ERR_BAD_SHAPE, "vxm_vecint2d_bwd: nsteps should be >= 1, found: %d", nsteps);
    VXM_REQUIRE(vec && steps && gout && gvec && work, VXM_ERR_NULL_POINTER, "vxm_vecint2d_bwd: null pointer");
    const size_t n = (size_t)B * 2 * H * W;
    const dim3 grid(vxm_blocks((long long)H * W, 256), B);
    const float scale = 1.0f / (float)(1u << nsteps);
    const float* g = gout;
    for (int k = nsteps - 1; k >= 0; --k) {
        float* gn = k == 0 ? gvec : work + (size_t)(k & 1) * n;
        (void)hipMemsetAsync(gn, 0, sizeof(float) * n, VXM_STREAM(stream));
        hipLaunchKernelGGL(k_vecint2d_step_bwd, grid, dim3(256), 0, VXM_STREAM(stream), k == 0 ? vec : steps + (size_t)(k - 1) * n,
                           k == 0 ? scale : 1.0f, g, gn, H, W);
        g = gn;
    }
    return vxm_check_launch("vxm_vecint2d_bwd");
}

int vxm_resize2d_fwd(const float* x, float* out, int B, int C, int H, int W, int oH, int oW, float factor, void* stream) {
    if (int e = check_plane("vxm_resize2d_fwd", B, C, H, W)) return e;
    VXM_REQUIRE(x && out, VXM_ERR_NULL_POINTER, "vxm_resize2d_fwd: null pointer");
    VXM_REQUIRE(oH > 0 && oW > 0 && factor > 0.0f && (long long)B * C <= 65535, VXM_ERR_BAD_SHAPE, "vxm_resize2d_fwd: bad output shape %dx%d / factor %g", oH, oW, factor);
    const float rh = oH > 1 ? (float)(H - 1) / (float)(oH - 1) : 0.0f, rw = oW > 1 ? (float)(W - 1) / (float)(oW - 1) : 0.0f;
    hipLaunchKernelGGL(k_resize2d_fwd, dim3(vxm_blocks((long long)oH * oW, 256), B * C), dim3(256), 0, VXM_STREAM(stream), x, out, H, W, oH, oW,
                       rh, rw, factor > 1.0f ? factor : 1.0f, factor < 1.0f ? factor : 1.0f);
    return vxm_check_launch("vxm_resize2d_fwd");
}

int vxm_resize2d_bwd(const float* gout, float* gx, int B, int C, int H, int W, int oH, int oW, float factor, void* stream) {
    if (int e = check_plane("vxm_resize2d_bwd", B, C, H, W)) return e;
    VXM_REQUIRE(gout && gx, VXM_ERR_NULL_POINTER, "vxm_resize2d_bwd: null pointer");
    VXM_REQUIRE(oH > 0 && oW > 0 && factor > 0.0f && (long long)B * C <= 65535, VXM_ERR_BAD_SHAPE, "vxm_resize2d_bwd: bad output shape %dx%d / factor %g", oH, oW, factor);
    const float rh = oH > 1 ? (float)(H - 1) / (float)(oH - 1) : 0.0f, rw = oW > 1 ? (float)(W - 1) / (float)(oW - 1) : 0.0f;
    (void)hipMemsetAsync(gx, 0, sizeof(float) * (size_t)B * C * H * W, VXM_STREAM(stream));
    hipLaunchKernelGGL(k_resize2d_bwd, dim3(vxm_blocks((long long)oH * oW, 256), B * C), dim3(256), 0, VXM_STREAM(stream), gout, gx, H, W, oH, oW,
                       rh, rw, factor);
    return vxm_check_launch("vxm_resize2d_bwd");
}

int vxm_maxpool2d_fwd(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    if (int e = check_plane("vxm_maxpool2d_fwd", B, C, H, W)) return e;
    VXM_REQUIRE(x && y, VXM_ERR_NULL_POINTER, "vxm_maxpool2d_fwd: null pointer");
    hipLaunchKernelGGL(k_maxpool2d_fwd, dim3(vxm_blocks((long long)C * (H / 2) * (W / 2), 256), B), dim3(256), 0, VXM_STREAM(stream), x, y, C, H, W);
    return vxm_check_launch("vxm_maxpool2d_fwd");
}

int vxm_maxpool2d_bwd(const float* x, const float* gpool, float* gx, int B, int C, int H, int W, void* stream) {
    if (int e = check_plane("vxm_maxpool2d_bwd", B, C, H, W)) return e;
    VXM_REQUIRE(x && gpool && gx, VXM_ERR_NULL_POINTER, "vxm_maxpool2d_bwd: null pointer");
    if ((H | W) & 1) (void)hipMemsetAsync(gx, 0, sizeof(float) * (size_t)B * C * H * W, VXM_STREAM(stream));
    hipLaunchKernelGGL(k_maxpool2d_bwd, dim3(vxm_blocks((long long)C * (H / 2) * (W / 2), 256), B), dim3(256), 0, VXM_STREAM(stream), x, gpool, gx, C, H, W);
    return vxm_check_launch("vxm_maxpool2d_bwd");
}

int vxm_upsample2d_cat(const float* x0, int C0, const float* x1, int C1, float* out, int B, int H, int W, void* stream) {
    VXM_REQUIRE(x0 && out && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_upsample2d_cat: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C0 > 0 && C1 >= 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, VXM_ERR_BAD_SHAPE, "vxm_upsample2d_cat: bad shape");
    hipLaunchKernelGGL(k_upsample2d_cat, dim3(vxm_blocks((long long)(C0 + C1) * H * W, 256), B), dim3(256), 0, VXM_STREAM(stream), x0, C0, x1, C1, out, H, W);
    return vxm_check_launch("vxm_upsample2d_cat");
}

int vxm_upsample2d_bwd(const float* g, int Ctot, float* gx0, int C0, int B, int H, int W, void* stream) {
    VXM_REQUIRE(g && gx0, VXM_ERR_NULL_POINTER, "vxm_upsample2d_bwd: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C0 > 0 && Ctot >= C0 && H > 0 && W > 0, VXM_ERR_BAD_SHAPE, "vxm_upsample2d_bwd: bad shape");
    hipLaunchKernelGGL(k_upsample2d_bwd, dim3(vxm_blocks((long long)C0 * H * W, 256), B), dim3(256), 0, VXM_STREAM(stream), g, Ctot, gx0, C0, H, W);
    return vxm_check_launch("vxm_upsample2d_bwd");
}

}  // extern "C"
