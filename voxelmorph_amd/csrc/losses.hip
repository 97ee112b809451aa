// Loss and optimiser kernels of the VxmDense training step (HBM-bound streaming reductions).
//
// Replaces (paths relative to the reference root):
//   voxelmorph/torch/losses.py:15-67    NCC.loss   (5 dense 9^3 conv3d box sums + ~25 elementwise)
//   voxelmorph/torch/losses.py:75-76    MSE.loss
//   voxelmorph/torch/losses.py:84-90    Dice.loss
//   voxelmorph/torch/losses.py:102-135  Grad.loss
//   scripts/torch/train.py:161,220      torch.optim.Adam.step over 24 tensors -> one flat buffer
// Reductions are accumulated in fp64 (wave shuffle -> LDS -> one atomic per block) so their value
// does not depend on the block schedule beyond 1e-16 relative; the scalar loss is written as fp32.
#include <cmath>
#include "vxm_common.h"
#include "vxm_device.h"

namespace {

__device__ __forceinline__ void block_atomic_add(double v, double* dst, double* red) {
    v = vxm_wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
        atomicAdd(dst, s);
    }
}

// ------------------------------------------------------------------ NCC
// pass 1: products + box sum along W.  out planes q = 0..4 (I, J, I^2, J^2, IJ), layout [5][B][V].
__global__ void __launch_bounds__(256) k_ncc_prod_w(const float* __restrict__ I, const float* __restrict__ J, float* __restrict__ out,
                                                    long long BV, int W, int r) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= BV) return;
    const int w = (int)(p % W);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    const int lo = max(-r, -w), hi = min(r, W - 1 - w);
    for (int k = lo; k <= hi; ++k) {
        const float a = I[p + k], b = J[p + k];
        s0 += a; s1 += b; s2 += a * a; s3 += b * b; s4 += a * b;
    }
    out[p] = s0; out[BV + p] = s1; out[2 * BV + p] = s2; out[3 * BV + p] = s3; out[4 * BV + p] = s4;
}

// generic zero-padded box sum of `nplanes` [D,H,W] planes along one axis (stride/extent given)
__global__ void __launch_bounds__(256) k_box_axis(const float* __restrict__ in, float* __restrict__ out, long long n, int extent,
                                                  long long stride, int r) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int i = (int)((p / stride) % extent);
    const int lo = max(-r, -i), hi = min(r, extent - 1 - i);
    float s = 0.f;
    for (int k = lo; k <= hi; ++k) s += in[p + k * stride];
    out[p] = s;
}

// losses.py:57-65 from the five box sums, in the reference's operation order
__device__ __forceinline__ void ncc_terms(float Is, float Js, float I2s, float J2s, float IJs, float n, float& cross, float& Ivar,
                                          float& Jvar) {
    const float uI = Is / n, uJ = Js / n;
    cross = IJs - uJ * Is - uI * Js + uI * uJ * n;
    Ivar = I2s - 2.0f * uI * Is + uI * uI * n;
    Jvar = J2s - 2.0f * uJ * Js + uJ * uJ * n;
}

// pass 3: box sum along D of the 5 planes, cc, block reduction.
__global__ void __launch_bounds__(256) k_ncc_cc(const float* __restrict__ t2, float* __restrict__ sums, double* __restrict__ acc,
                                                long long BV, int D, long long HW, int r, float n) {
    __shared__ double red[4];
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    double cc = 0.0;
    if (p < BV) {
        const int d = (int)((p / HW) % D);
        const int lo = max(-r, -d), hi = min(r, D - 1 - d);
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = lo; k <= hi; ++k) {
#pragma unroll
            for (int q = 0; q < 5; ++q) s[q] += t2[q * BV + p + k * HW];
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) sums[q * BV + p] = s[q];
        float cross, Ivar, Jvar;
        ncc_terms(s[0], s[1], s[2], s[3], s[4], n, cross, Ivar, Jvar);
        cc = (double)(cross * cross / (Ivar * Jvar + 1e-5f));
    }
    block_atomic_add(cc, acc, red);
}

__global__ void k_finish_mean(const double* __restrict__ acc, float* __restrict__ loss, double scale) {
    if (threadIdx.x == 0 && blockIdx.x == 0) loss[0] = (float)(acc[0] * scale);
}

// backward pass 1: a = dcc/dJs, b = dcc/dJ2s, c = dcc/dIJs at every voxel, box-summed along D.
__global__ void __launch_bounds__(256) k_ncc_abc_d(const float* __restrict__ sums, float* __restrict__ u1, long long BV, int D, long long HW,
                                                   int r, float n) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= BV) return;
    const int d = (int)((p / HW) % D);
    const int lo = max(-r, -d), hi = min(r, D - 1 - d);
    float sa = 0.f, sb = 0.f, sc = 0.f;
    for (int k = lo; k <= hi; ++k) {
        const long long q = p + k * HW;
        const float Is = sums[q], Js = sums[BV + q];
        float cross, Ivar, Jvar;
        ncc_terms(Is, Js, sums[2 * BV + q], sums[3 * BV + q], sums[4 * BV + q], n, cross, Ivar, Jvar);
        const float den = Ivar * Jvar + 1e-5f;
        const float t = cross / den;             // cross/den
        const float t2 = t * t * Ivar;           // cross^2/den^2 * Ivar = -dcc/dJvar
        sa += 2.0f * t * (-Is / n) + t2 * (2.0f * Js / n);
        sb += -t2;
        sc += 2.0f * t;
    }
    u1[p] = sa; u1[BV + p] = sb; u1[2 * BV + p] = sc;
}

// backward pass 3: box sum along W of the three planes and the chain rule onto J:
// dL/dJ = gloss * (-1/N) * [ S(a) + 2 J S(b) + I S(c) ]   (box filter is self-adjoint)
__global__ void __launch_bounds__(256) k_ncc_grad_w(const float* __restrict__ u2, const float* __restrict__ I, const float* __restrict__ J,
                                                    const float* __restrict__ gloss, float* __restrict__ gJ, long long BV, int W, int r) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= BV) return;
    const int w = (int)(p % W);
    const int lo = max(-r, -w), hi = min(r, W - 1 - w);
    float sa = 0.f, sb = 0.f, sc = 0.f;
    for (int k = lo; k <= hi; ++k) { sa += u2[p + k]; sb += u2[BV + p + k]; sc += u2[2 * BV + p + k]; }
    const float scale = -gloss[0] / (float)BV;
    gJ[p] = scale * (sa + 2.0f * J[p] * sb + I[p] * sc);
}

// ---- NCC with ANY window (losses.py:26-36,47-67).  The reference pads EVERY axis by win[0] // 2 whatever the other window sizes are,
// so a window that is not cubic (or an even one) changes the extent of its box sums: O = S + 2 pad - win + 1 per axis; cc and its mean
// live on that shape.  Separable passes whose axis filter may change the extent; tensors are [outer][S][inner] -> [outer][O][inner]
// (planes and batch are part of `outer`, every stage buffer is densely packed at its own size).
//   forward : out[o] = sum_{k=0..w-1} in[o - p + k]   (taps inside [0, S), ascending k as in the cubic passes above)
//   adjoint : the same kernel from O back to S with p' = w - 1 - p
__global__ void __launch_bounds__(256) k_box_axis_g(const float* __restrict__ in, float* __restrict__ out, long long n_out, int S, int O,
                                                    long long inner, int w, int p) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_out) return;
    const long long r = idx % inner, t = idx / inner;
    const int o = (int)(t % O);
    const long long outer = t / O;
    const float* base = in + outer * S * inner + r;
    const int lo = max(0, p - o), hi = min(w - 1, S - 1 - o + p);
    float s = 0.f;
    for (int k = lo; k <= hi; ++k) s += base[(long long)(o - p + k) * inner];
    out[idx] = s;
}

// products + box sum along W: I, J [rows][W] -> out [5][rows][Ow]
__global__ void __launch_bounds__(256) k_ncc_win_prod_w(const float* __restrict__ I, const float* __restrict__ J, float* __restrict__ out,
                                                        long long rows, int W, int Ow, int w, int p) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, n = rows * Ow;
    if (idx >= n) return;
    const int o = (int)(idx % Ow);
    const long long row = idx / Ow;
    const float* Ir = I + row * W;
    const float* Jr = J + row * W;
    const int lo = max(0, p - o), hi = min(w - 1, W - 1 - o + p);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    for (int k = lo; k <= hi; ++k) {
        const float a = Ir[o - p + k], b = Jr[o - p + k];
        s0 += a; s1 += b; s2 += a * a; s3 += b * b; s4 += a * b;
    }
    out[idx] = s0; out[n + idx] = s1; out[2 * n + idx] = s2; out[3 * n + idx] = s3; out[4 * n + idx] = s4;
}

// cc from the five box sums (losses.py:57-65), fp64 block reduction
__global__ void __launch_bounds__(256) k_ncc_win_cc(const float* __restrict__ sums, double* __restrict__ acc, long long N, float n) {
    __shared__ double red[4];
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    double cc = 0.0;
    if (p < N) {
        float cross, Ivar, Jvar;
        ncc_terms(sums[p], sums[N + p], sums[2 * N + p], sums[3 * N + p], sums[4 * N + p], n, cross, Ivar, Jvar);
        cc = (double)(cross * cross / (Ivar * Jvar + 1e-5f));
    }
    block_atomic_add(cc, acc, red);
}

// (a, b, c) = d cc / d(J sum, J^2 sum, IJ sum) on the shape of the box sums (as k_ncc_abc_d)
__global__ void __launch_bounds__(256) k_ncc_win_abc(const float* __restrict__ sums, float* __restrict__ abc, long long N, float n) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= N) return;
    const float Is = sums[q], Js = sums[N + q];
    float cross, Ivar, Jvar;
    ncc_terms(Is, Js, sums[2 * N + q], sums[3 * N + q], sums[4 * N + q], n, cross, Ivar, Jvar);
    const float den = Ivar * Jvar + 1e-5f;
    const float t = cross / den;
    const float t2 = t * t * Ivar;
    abc[q] = 2.0f * t * (-Is / n) + t2 * (2.0f * Js / n);
    abc[N + q] = -t2;
    abc[2 * N + q] = 2.0f * t;
}

// adjoint box filter along W of the three planes [3][rows][Ow] and the chain rule onto J [rows][W]:
// dL/dJ = gloss * (-1/N) * [ S^T(a) + 2 J S^T(b) + I S^T(c) ],  N = number of box sums
__global__ void __launch_bounds__(256) k_ncc_win_grad_w(const float* __restrict__ u, const float* __restrict__ I, const float* __restrict__ J,
                                                        const float* __restrict__ gloss, float* __restrict__ gJ, long long rows, int W, int Ow,
                                                        int w, int p, double N) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * W) return;
    const int i = (int)(idx % W);
    const long long row = idx / W, n = rows * Ow;
    const float* ur = u + row * Ow;
    const int pa = w - 1 - p;
    const int lo = max(0, pa - i), hi = min(w - 1, Ow - 1 - i + pa);
    float sa = 0.f, sb = 0.f, sc = 0.f;
    for (int k = lo; k <= hi; ++k) { const int o = i - pa + k; sa += ur[o]; sb += ur[n + o]; sc += ur[2 * n + o]; }
    const float scale = -gloss[0] / (float)N;
    gJ[idx] = scale * (sa + 2.0f * J[idx] * sb + I[idx] * sc);
}

// ---- fused NCC for windows <= 9: one kernel marches a (8 x 32)-pixel column along D.  Per depth slice the haloed
// products tile goes through LDS (W box sum, then H box sum: direct 2R+1-tap sums in the same order as the generic
// passes above, so the values are identical), the last 2R+1 slices of 2-D sums live in a register shift ring, and
// the 3-D sums of the slice R behind the front give cc, its fp64 block reduction and the three partials
// (a, b, c) = d cc / d(J sum, J^2 sum, IJ sum) that backward box-filters.  HBM traffic: I, J once (+ L2-served halo),
// 3 planes written -- instead of 5 planes x 3 passes.
constexpr int NF_TH = 8, NF_TW = 32, NF_SEG = 40;     // tile, and (at most) slices per block along D
constexpr int NF_PWP = 44;                            // floats per row of the haloed products tile: >= 32 + 2 R (R <= 4) + the 12-float reads below, rows 16-byte aligned

// W box sums of FOUR neighbouring columns of one (plane, row) from three 16-byte LDS reads instead of 4 x WIN 4-byte ones (round 6, last): the
// kernel is bound by vector-instruction issue, and the W pass was 90 of its 417 wave-instructions per slice.  Every sum is
// still t = 0; t += p[c + k], k = 0 .. WIN - 1: the same operations in the same order, the same bits.
template <int WIN>
__device__ __forceinline__ void ncc_wpass4(const float* __restrict__ prow, float* __restrict__ rrow, int cg) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(prow + 4 * cg), b = *reinterpret_cast<const f32x4*>(prow + 4 * cg + 4),
                c = *reinterpret_cast<const f32x4*>(prow + 4 * cg + 8);
    const float v[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < WIN; ++k) t += v[j + k];
        o[j] = t;
    }
    *reinterpret_cast<f32x4*>(rrow + 4 * cg) = o;
}

template <int R>
__global__ void __launch_bounds__(256) k_ncc_fused_fwd(const float* __restrict__ I, const float* __restrict__ J, float* __restrict__ abc,
                                                       double* __restrict__ acc, int D, int H, int W, long long BV, int seg) {
    constexpr int WIN = 2 * R + 1, PH = NF_TH + 2 * R, PW = NF_TW + 2 * R, PWP = NF_PWP;
    __shared__ __attribute__((aligned(16))) float P[5][PH][PWP];
    __shared__ __attribute__((aligned(16))) float Rw[5][PH][NF_TW];
    __shared__ double red[4];
    const int tid = threadIdx.x, wx = tid & 31, hy = tid >> 5;
    const int ntw = (W + NF_TW - 1) / NF_TW;
    // pixel column of this block: block ids go to the XCDs round-robin, so (when the count allows) XCD x takes a contiguous eighth of the
    // columns -- the blocks it runs side by side are neighbours and share their 8-pixel halos in its L2
    const int bcol = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int w0 = (bcol % ntw) * NF_TW, h0 = (bcol / ntw) * NF_TH;
    const int dlo = blockIdx.y * seg, dhi = min(D, dlo + seg);
    const size_t HW = (size_t)H * W, vol = (size_t)blockIdx.z * D * HW;
    const float n = (float)(WIN * WIN * WIN);
    const int gh = h0 + hy, gw = w0 + wx;
    const bool pix_ok = gh < H && gw < W;
    // staging slots of this thread: element idx = tid + 256 t of the haloed PH x PW tile -> its LDS word and its offset inside a slice (-1: padding)
    constexpr int NST = (PH * PW + 255) / 256;
    int st_lds[NST], st_g[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int idx = tid + 256 * t, r = idx / PW, c = idx - r * PW;
        const int y = h0 - R + r, x = w0 - R + c;
        st_lds[t] = idx < PH * PW ? r * PWP + c : -1;
        st_g[t] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? y * W + x : -1;
    }
    float ring[WIN][5];
#pragma unroll
    for (int k = 0; k < WIN; ++k)
#pragma unroll
        for (int q = 0; q < 5; ++q) ring[k][q] = 0.0f;
    double cc_sum = 0.0;
    for (int z = dlo - R; z < dhi + R; ++z) {
        float s2[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (z >= 0 && z < D) {                              // block-uniform; slices outside the volume are zero padding
            const float* Iz = I + vol + (size_t)z * HW;
            const float* Jz = J + vol + (size_t)z * HW;
#pragma unroll
            for (int t = 0; t < NST; ++t) {
                if (st_lds[t] >= 0) {                            // (the slots of a thread and their addresses do not depend on the slice: set up once)
                    float a = 0.0f, b = 0.0f;
                    if (st_g[t] >= 0) { a = Iz[st_g[t]]; b = Jz[st_g[t]]; }
                    float* const p = &P[0][0][0] + st_lds[t];
                    p[0] = a; p[PH * PWP] = b; p[2 * PH * PWP] = a * a; p[3 * PH * PWP] = b * b; p[4 * PH * PWP] = a * b;
                }
            }
            __syncthreads();
            for (int idx = tid; idx < 5 * PH * (NF_TW / 4); idx += 256) ncc_wpass4<WIN>(&P[0][0][0] + (idx >> 3) * PWP, &Rw[0][0][0] + (idx >> 3) * NF_TW, idx & 7);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int k = 0; k < WIN; ++k) s2[q] += Rw[q][hy + k][wx];
        }
#pragma unroll
        for (int k = 0; k + 1 < WIN; ++k)
#pragma unroll
            for (int q = 0; q < 5; ++q) ring[k][q] = ring[k + 1][q];
#pragma unroll
        for (int q = 0; q < 5; ++q) ring[WIN - 1][q] = s2[q];
        const int d = z - R;
        if (d >= dlo && pix_ok) {
            float s[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                float t = 0.0f;
#pragma unroll
                for (int k = 0; k < WIN; ++k) t += ring[k][q];
                s[q] = t;
            }
            float cross, Ivar, Jvar;
            ncc_terms(s[0], s[1], s[2], s[3], s[4], n, cross, Ivar, Jvar);
            const float den = Ivar * Jvar + 1e-5f;
            cc_sum += (double)(cross * cross / den);
            if (abc) {
                const float t = cross / den, t2 = t * t * Ivar;       // as k_ncc_abc_d
                const size_t p = vol + (size_t)d * HW + (size_t)gh * W + gw;
                abc[p] = 2.0f * t * (-s[0] / n) + t2 * (2.0f * s[1] / n);
                abc[BV + p] = -t2;
                abc[2 * BV + p] = 2.0f * t;
            }
        }
    }
    block_atomic_add(cc_sum, acc, red);
}

// backward: 3-D box filter of (a, b, c) by the same march, chain rule onto J fused into the output:
// dL/dJ = gloss * (-1/N) * [ S(a) + 2 J S(b) + I S(c) ]
template <int R>
__global__ void __launch_bounds__(256) k_ncc_fused_bwd(const float* __restrict__ I, const float* __restrict__ J, const float* __restrict__ abc,
                                                       const float* __restrict__ gloss, float* __restrict__ gJ, int D, int H, int W,
                                                       long long BV, int seg) {
    constexpr int WIN = 2 * R + 1, PH = NF_TH + 2 * R, PW = NF_TW + 2 * R, PWP = NF_PWP;
    __shared__ __attribute__((aligned(16))) float P[3][PH][PWP];
    __shared__ __attribute__((aligned(16))) float Rw[3][PH][NF_TW];
    const int tid = threadIdx.x, wx = tid & 31, hy = tid >> 5;
    const int ntw = (W + NF_TW - 1) / NF_TW;
    // pixel column of this block: block ids go to the XCDs round-robin, so (when the count allows) XCD x takes a contiguous eighth of the
    // columns -- the blocks it runs side by side are neighbours and share their 8-pixel halos in its L2
    const int bcol = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int w0 = (bcol % ntw) * NF_TW, h0 = (bcol / ntw) * NF_TH;
    const int dlo = blockIdx.y * seg, dhi = min(D, dlo + seg);
    const size_t HW = (size_t)H * W, vol = (size_t)blockIdx.z * D * HW;
    const int gh = h0 + hy, gw = w0 + wx;
    const bool pix_ok = gh < H && gw < W;
    const float scale = -gloss[0] / (float)BV;
    constexpr int NST = (PH * PW + 255) / 256;                 // staging slots of this thread (as k_ncc_fused_fwd)
    int st_lds[NST], st_g[NST];
#pragma unroll
    for (int t = 0; t < NST; ++t) {
        const int idx = tid + 256 * t, r = idx / PW, c = idx - r * PW;
        const int y = h0 - R + r, x = w0 - R + c;
        st_lds[t] = idx < PH * PW ? r * PWP + c : -1;
        st_g[t] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? y * W + x : -1;
    }
    float ring[WIN][3];
#pragma unroll
    for (int k = 0; k < WIN; ++k)
#pragma unroll
        for (int q = 0; q < 3; ++q) ring[k][q] = 0.0f;
    for (int z = dlo - R; z < dhi + R; ++z) {
        float s2[3] = {0.f, 0.f, 0.f};
        if (z >= 0 && z < D) {
            const size_t zoff = vol + (size_t)z * HW;
#pragma unroll
            for (int t = 0; t < NST; ++t) {
                if (st_lds[t] >= 0) {
                    float* const p = &P[0][0][0] + st_lds[t];
#pragma unroll
                    for (int q = 0; q < 3; ++q) p[q * PH * PWP] = st_g[t] >= 0 ? abc[q * BV + zoff + st_g[t]] : 0.0f;
                }
            }
            __syncthreads();
            for (int idx = tid; idx < 3 * PH * (NF_TW / 4); idx += 256) ncc_wpass4<WIN>(&P[0][0][0] + (idx >> 3) * PWP, &Rw[0][0][0] + (idx >> 3) * NF_TW, idx & 7);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int k = 0; k < WIN; ++k) s2[q] += Rw[q][hy + k][wx];
        }
#pragma unroll
        for (int k = 0; k + 1 < WIN; ++k)
#pragma unroll
            for (int q = 0; q < 3; ++q) ring[k][q] = ring[k + 1][q];
#pragma unroll
        for (int q = 0; q < 3; ++q) ring[WIN - 1][q] = s2[q];
        const int d = z - R;
        if (d >= dlo && pix_ok) {
            float s[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float t = 0.0f;
#pragma unroll
                for (int k = 0; k < WIN; ++k) t += ring[k][q];
                s[q] = t;
            }
            const size_t p = vol + (size_t)d * HW + (size_t)gh * W + gw;
            gJ[p] = scale * (s[0] + 2.0f * J[p] * s[1] + I[p] * s[2]);
        }
    }
}

// ------------------------------------------------------------------ Grad
template <int L2>
__device__ __forceinline__ float pen(float t) { return L2 ? t * t : fabsf(t); }
template <int L2>
__device__ __forceinline__ float dpen(float t) { return L2 ? 2.0f * t : (t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f)); }

// A wave owns whole W rows (c, d, h): the row index is decoded once per row with scalar arithmetic, lanes stride over w -- no
// per-element integer division (three runtime divisions per voxel made the first version compute-bound on the full-resolution
// field of the dense configuration: 115 us for 82 MB).  Row sums are fp32, the running sums fp64.
template <int L2>
__global__ void __launch_bounds__(256) k_gradloss_fwd(const float* __restrict__ y, double* __restrict__ acc, int C, int D, int H, int W) {
    __shared__ double red[4];
    const int HW = H * W, nrow = C * D * H;
    const size_t b = blockIdx.y;
    const float* yb = y + b * (size_t)nrow * W;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double sd = 0.0, sh = 0.0, sw = 0.0;
    for (int r = blockIdx.x * 4 + wave; r < nrow; r += gridDim.x * 4) {           // wave-uniform
        const int h = r % H, d = (r / H) % D;
        const float* row = yb + (size_t)r * W;
        const bool dn = d + 1 < D, hn = h + 1 < H;
        float fd = 0.0f, fh = 0.0f, fw = 0.0f;
        for (int w = lane; w < W; w += 64) {
            const float v = row[w];
            if (dn) fd += pen<L2>(row[w + HW] - v);
            if (hn) fh += pen<L2>(row[w + W] - v);
            if (w + 1 < W) fw += pen<L2>(row[w + 1] - v);
        }
        sd += (double)fd; sh += (double)fh; sw += (double)fw;
    }
    // VXM_GRAD_SLOTS copies of every sum: blocks spread their fp64 atomics over the slots (the L2 serialises same-address
    // atomics), k_gradloss_finish adds the slots up -- which is what lets the grid have thousands of blocks
    const int slot = blockIdx.x % VXM_GRAD_SLOTS;
    block_atomic_add(sd, acc + ((b * 3 + 0) * VXM_GRAD_SLOTS + slot), red);
    block_atomic_add(sh, acc + ((b * 3 + 1) * VXM_GRAD_SLOTS + slot), red);
    block_atomic_add(sw, acc + ((b * 3 + 2) * VXM_GRAD_SLOTS + slot), red);
}

// W % 4 == 0: 16-byte loads, VXM_GRAD_RU consecutive rows per wave with all of their loads issued before the arithmetic (the
// scalar kernel above keeps 4 dword loads per lane in flight: 1.07 TB/s on the 82 MB full-resolution field of the dense
// configuration).  A neighbour that does not exist is replaced by the voxel itself -- its difference is 0 and pen(0) = 0.
#define VXM_GRAD_RU 4
template <int L2>
__global__ void __launch_bounds__(256) k_gradloss_fwd_v4(const float* __restrict__ y, double* __restrict__ acc, int C, int D, int H, int W) {
    __shared__ double red[4];
    const int HW = H * W, nrow = C * D * H, W4 = W >> 2;
    const size_t b = blockIdx.y;
    const float* yb = y + b * (size_t)nrow * W;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double sd = 0.0, sh = 0.0, sw = 0.0;
    for (int r0 = (blockIdx.x * 4 + wave) * VXM_GRAD_RU; r0 < nrow; r0 += gridDim.x * 4 * VXM_GRAD_RU) {           // wave-uniform
        for (int w4 = lane; w4 < W4; w4 += 64) {
            f32x4 v[VXM_GRAD_RU], vd[VXM_GRAD_RU], vh[VXM_GRAD_RU];
            float nx[VXM_GRAD_RU];
#pragma unroll
            for (int u = 0; u < VXM_GRAD_RU; ++u) {
                const int r = min(r0 + u, nrow - 1);
                const int h = r % H, d = (r / H) % D;
                const float* row = yb + (size_t)r * W + w4 * 4;
                v[u] = *reinterpret_cast<const f32x4*>(row);
                vd[u] = *reinterpret_cast<const f32x4*>(row + (d + 1 < D ? HW : 0));
                vh[u] = *reinterpret_cast<const f32x4*>(row + (h + 1 < H ? W : 0));
                nx[u] = row[w4 * 4 + 4 < W ? 4 : 3];
            }
            float fd = 0.0f, fh = 0.0f, fw = 0.0f;
#pragma unroll
            for (int u = 0; u < VXM_GRAD_RU; ++u) {
                if (r0 + u >= nrow) break;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    fd += pen<L2>(vd[u][k] - v[u][k]);
                    fh += pen<L2>(vh[u][k] - v[u][k]);
                    fw += pen<L2>((k < 3 ? v[u][k < 3 ? k + 1 : 3] : nx[u]) - v[u][k]);
                }
            }
            sd += (double)fd; sh += (double)fh; sw += (double)fw;
        }
    }
    const int slot = blockIdx.x % VXM_GRAD_SLOTS;
    block_atomic_add(sd, acc + ((b * 3 + 0) * VXM_GRAD_SLOTS + slot), red);
    block_atomic_add(sh, acc + ((b * 3 + 1) * VXM_GRAD_SLOTS + slot), red);
    block_atomic_add(sw, acc + ((b * 3 + 2) * VXM_GRAD_SLOTS + slot), red);
}

// `axes` = 3 for volumes, 2 for planar images passed with D = 1 (no difference along D exists: that term is left out)
__global__ void k_gradloss_finish(const double* __restrict__ acc, float* __restrict__ loss, int B, int C, int D, int H, int W, double mult,
                                  int axes) {
    if (threadIdx.x || blockIdx.x) return;
    const double nd = (double)C * (D - 1) * H * W, nh = (double)C * D * (H - 1) * W, nw = (double)C * D * H * (W - 1);
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
        double a[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < 3; ++k)
            for (int s = 0; s < VXM_GRAD_SLOTS; ++s) a[k] += acc[(b * 3 + k) * VXM_GRAD_SLOTS + s];
        tot += mult * ((axes == 3 ? a[0] / nd : 0.0) + a[1] / nh + a[2] / nw) / (double)axes;
    }
    loss[0] = (float)(tot / B);
}

template <int L2>
__global__ void __launch_bounds__(256) k_gradloss_bwd(const float* __restrict__ y, const float* __restrict__ gloss, float* __restrict__ gy,
                                                      int B, int C, int D, int H, int W, float mult, int axes) {
    const int HW = H * W, nrow = C * D * H;
    const size_t b = blockIdx.y;
    const float* yb = y + b * (size_t)nrow * W;
    float* gb = gy + b * (size_t)nrow * W;
    const float base = gloss[0] * mult / ((float)axes * (float)B);
    const float kd = axes == 3 ? base / ((float)C * (D - 1) * H * W) : 0.0f, kh = base / ((float)C * D * (H - 1) * W), kw = base / ((float)C * D * H * (W - 1));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int r = blockIdx.x * 4 + wave; r < nrow; r += gridDim.x * 4) {           // one W row per wave and iteration (wave-uniform)
        const int h = r % H, d = (r / H) % D;
        const float* row = yb + (size_t)r * W;
        const bool dp = d > 0, dn = d + 1 < D, hp = h > 0, hn = h + 1 < H;
        for (int w = lane; w < W; w += 64) {
            const float v = row[w];
            float g = 0.f;
            if (dp) g += kd * dpen<L2>(v - row[w - HW]);
            if (dn) g -= kd * dpen<L2>(row[w + HW] - v);
            if (hp) g += kh * dpen<L2>(v - row[w - W]);
            if (hn) g -= kh * dpen<L2>(row[w + W] - v);
            if (w > 0) g += kw * dpen<L2>(v - row[w - 1]);
            if (w + 1 < W) g -= kw * dpen<L2>(row[w + 1] - v);
            gb[(size_t)r * W + w] = g;
        }
    }
}

// W % 4 == 0: two rows per wave, seven loads per row issued together; absent neighbours are replaced by the voxel (dpen(0) = 0)
template <int L2>
__global__ void __launch_bounds__(256) k_gradloss_bwd_v4(const float* __restrict__ y, const float* __restrict__ gloss, float* __restrict__ gy,
                                                         int B, int C, int D, int H, int W, float mult, int axes) {
    const int HW = H * W, nrow = C * D * H, W4 = W >> 2;
    const size_t b = blockIdx.y;
    const float* yb = y + b * (size_t)nrow * W;
    float* gb = gy + b * (size_t)nrow * W;
    const float base = gloss[0] * mult / ((float)axes * (float)B);
    const float kd = axes == 3 ? base / ((float)C * (D - 1) * H * W) : 0.0f, kh = base / ((float)C * D * (H - 1) * W), kw = base / ((float)C * D * H * (W - 1));
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int r0 = (blockIdx.x * 4 + wave) * 2; r0 < nrow; r0 += gridDim.x * 8) {
        for (int w4 = lane; w4 < W4; w4 += 64) {
            f32x4 v[2], dm[2], dq[2], hm[2], hq[2];
            float lf[2], rt[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = min(r0 + u, nrow - 1);
                const int h = r % H, d = (r / H) % D;
                const float* row = yb + (size_t)r * W + w4 * 4;
                v[u] = *reinterpret_cast<const f32x4*>(row);
                dm[u] = *reinterpret_cast<const f32x4*>(row - (d > 0 ? HW : 0));
                dq[u] = *reinterpret_cast<const f32x4*>(row + (d + 1 < D ? HW : 0));
                hm[u] = *reinterpret_cast<const f32x4*>(row - (h > 0 ? W : 0));
                hq[u] = *reinterpret_cast<const f32x4*>(row + (h + 1 < H ? W : 0));
                lf[u] = row[w4 > 0 ? -1 : 0];
                rt[u] = row[w4 * 4 + 4 < W ? 4 : 3];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (r0 + u >= nrow) break;
                f32x4 g;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float c = v[u][k];
                    const float wl = k > 0 ? v[u][k > 0 ? k - 1 : 0] : lf[u], wr = k < 3 ? v[u][k < 3 ? k + 1 : 3] : rt[u];
                    g[k] = kd * (dpen<L2>(c - dm[u][k]) - dpen<L2>(dq[u][k] - c)) + kh * (dpen<L2>(c - hm[u][k]) - dpen<L2>(hq[u][k] - c))
                           + kw * (dpen<L2>(c - wl) - dpen<L2>(wr - c));
                }
                *reinterpret_cast<f32x4*>(gb + (size_t)(r0 + u) * W + w4 * 4) = g;
            }
        }
    }
}

// ------------------------------------------------------------------ MSE
__global__ void __launch_bounds__(256) k_mse_fwd(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ acc, long long n) {
    __shared__ double red[4];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float t = a[i] - b[i];
        s += (double)(t * t);
    }
    block_atomic_add(s, acc, red);
}
__global__ void __launch_bounds__(256) k_mse_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gloss,
                                                 float* __restrict__ ga, float* __restrict__ gb, long long n) {
    const float k = 2.0f * gloss[0] / (float)n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float g = k * (a[i] - b[i]);
        if (ga) ga[i] = g;
        if (gb) gb[i] = -g;
    }
}

// ------------------------------------------------------------------ Dice
__global__ void __launch_bounds__(256) k_dice_fwd(const float* __restrict__ yt, const float* __restrict__ yp, double* __restrict__ acc, long long V) {
    __shared__ double red[4];
    const size_t bc = blockIdx.y;
    const float* t = yt + bc * (size_t)V;
    const float* p = yp + bc * (size_t)V;
    double sp = 0.0, ss = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
        sp += (double)(t[i] * p[i]);
        ss += (double)(t[i] + p[i]);
    }
    block_atomic_add(sp, acc + bc * 2, red);
    block_atomic_add(ss, acc + bc * 2 + 1, red);
}
__global__ void k_dice_finish(const double* __restrict__ acc, float* __restrict__ loss, int BC) {
    if (threadIdx.x || blockIdx.x) return;
    double tot = 0.0;
    for (int i = 0; i < BC; ++i) {
        const float top = 2.0f * (float)acc[2 * i];
        const float bot = fmaxf((float)acc[2 * i + 1], 1e-5f);
        tot += (double)(top / bot);
    }
    loss[0] = (float)(-tot / BC);
}
__global__ void __launch_bounds__(256) k_dice_bwd(const float* __restrict__ yt, const float* __restrict__ yp, const double* __restrict__ acc,
                                                  const float* __restrict__ gloss, float* __restrict__ gyt, float* __restrict__ gyp, int BC,
                                                  long long V) {
    const size_t bc = blockIdx.y;
    const float top = 2.0f * (float)acc[2 * bc];
    const float raw = (float)acc[2 * bc + 1];
    const float bot = fmaxf(raw, 1e-5f);
    const float k = -gloss[0] / (float)BC;
    const float second = raw > 1e-5f ? top / (bot * bot) : 0.0f;     // clamp passes no gradient below min
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < V; i += (long long)gridDim.x * 256) {
        const size_t o = bc * (size_t)V + i;
        if (gyp) gyp[o] = k * (2.0f * yt[o] / bot - second);
        if (gyt) gyt[o] = k * (2.0f * yp[o] / bot - second);
    }
}

// ------------------------------------------------------------------ weighted sum of the loss terms (scripts/torch/train.py:205-212)
// `loss = 0; for n: loss += loss_function(y_true[n], y_pred[n]) * weights[n]` as ONE launch (the products and the running sum in the
// reference's order, fp32) instead of a mul + an add of ATen per term; optionally the terms and the total are added to running sums the
// training script reads back once per epoch (train.py:215 logs them per step with three .item() syncs).  Backward: one launch, g_n = g * w_n.
struct LossTerms { const float* t[VXM_LOSS_TERMS_MAX]; float w[VXM_LOSS_TERMS_MAX]; int n; };
__global__ void k_loss_combine_fwd(const LossTerms lt, float* __restrict__ total, float* __restrict__ running) {
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sum = 0.0f;
    for (int i = 0; i < lt.n; ++i) {
        const float cur = lt.t[i][0] * lt.w[i];
        sum = sum + cur;
        if (running) running[i] += cur;
    }
    total[0] = sum;
    if (running) running[lt.n] += sum;
}
__global__ void k_loss_combine_bwd(const float* __restrict__ gtotal, const LossTerms lt, float* __restrict__ gterms) {
    if (threadIdx.x < lt.n && blockIdx.x == 0) gterms[threadIdx.x] = gtotal[0] * lt.w[threadIdx.x];
}

// out = a + b (the gradient of a tensor with two consumers: autograd's accumulation as a launch of this library -- preint_flow feeds
// both Grad and VecInt, networks.py:262-268)
__global__ void __launch_bounds__(256) k_add2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
    const long long i = 4 * ((long long)blockIdx.x * 256 + threadIdx.x);
    if (i + 3 < n && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + i), y = *reinterpret_cast<const f32x4*>(b + i);
        *reinterpret_cast<f32x4*>(out + i) = x + y;
    } else {
        for (long long k = i; k < n && k < i + 4; ++k) out[k] = a[k] + b[k];
    }
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam defaults: no amsgrad / weight decay)
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                              long long n, float step_size, float beta1, float beta2, float eps, float bc2_sqrt, float gscale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * gscale;
    const float mi = m[i] + (1.0f - beta1) * (gi - m[i]);           // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;       // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

// Capturable variant (vxm_adam_step_dev): the step counter and the two bias-correction factors live in DEVICE memory, so a hipGraph that
// contains the optimiser step replays correctly (a kernel argument would freeze `step` at its capture-time value).  k_adam_tick advances
// the counter and publishes the factors with the arithmetic of vxm_adam_step (double precision, then rounded to float); k_adam_dev is
// k_adam reading them.
struct AdamDevState { long long step; float step_size; float bc2_sqrt; };

__global__ void k_adam_tick(AdamDevState* st, float lr, float beta1, float beta2) {
    const long long t = st->step + 1;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    st->step = t;
    st->step_size = (float)((double)lr / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
}

__global__ void __launch_bounds__(256) k_adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                  long long n, const AdamDevState* __restrict__ st, float beta1, float beta2, float eps, float gscale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float step_size = st->step_size, bc2_sqrt = st->bc2_sqrt;
    const float gi = g[i] * gscale;
    const float mi = m[i] + (1.0f - beta1) * (gi - m[i]);
    const float vi = v[i] * beta2 + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
}

// grid of a reduction kernel: every block ends in fp64 atomics on a handful of addresses, which the L2 serialises
// (8192 blocks x 3 atomics on 3 addresses cost 0.3 ms for a 10 MB field): few, fat blocks.
unsigned reduce_blocks(long long n) {
    const long long nb = (n + 2047) / 2048;
    return (unsigned)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
}

unsigned stream_blocks(long long n) {
    const long long nb = (n + 255) / 256;
    return (unsigned)(nb > 8192 ? 8192 : (nb < 1 ? 1 : nb));
}

// Slices per block of the fused march: 40 (a fifth of them re-marched as warm-up halo for a 9-wide window) when that
// still gives every CU several blocks, 20 when it does not (one 160x192x224 pair is only 168 columns x 4 segments
// = 2.6 blocks per CU at 40).  Every output slice sums the same 2R+1 input slices in the same order whatever the
// segment length, so the result does not depend on it.
int ncc_segment(int B, int D, int H, int W) {
    static const int forced = [] { const char* e = getenv("VXM_NCC_SEG"); return e ? atoi(e) : 0; }();     // developer experiments
    if (forced > 0) return forced;
    const long long cols = (long long)((W + NF_TW - 1) / NF_TW) * ((H + NF_TH - 1) / NF_TH) * B;
    int seg = NF_SEG;
    while (seg > 20 && cols * ((D + seg - 1) / seg) < 1024) seg = (seg + 1) / 2;     // measured at B = 1: 40 -> 0.39 ms, 20 -> 0.30, 14 -> 0.34, 10 -> 0.36
    return seg;
}

// generic separable NCC passes; `sums` receives the five box sums, `work` is 5 (forward) / 6 (backward) planes of scratch
void ncc_generic_fwd(const float* I, const float* J, float* sums, float* work, double* acc, long long BV, int D, int H, int W, int r, float n,
                     hipStream_t s) {
    hipLaunchKernelGGL(k_ncc_prod_w, dim3(vxm_blocks(BV, 256)), dim3(256), 0, s, I, J, sums, BV, W, r);
    hipLaunchKernelGGL(k_box_axis, dim3(vxm_blocks(5 * BV, 256)), dim3(256), 0, s, sums, work, 5 * BV, H, (long long)W, r);
    hipLaunchKernelGGL(k_ncc_cc, dim3(vxm_blocks(BV, 256)), dim3(256), 0, s, work, sums, acc, BV, D, (long long)H * W, r, n);
}
void ncc_generic_bwd(const float* I, const float* J, const float* sums, const float* gloss, float* gJ, float* work, long long BV, int D, int H,
                     int W, int r, float n, hipStream_t s) {
    float* u1 = work;
    float* u2 = work + 3 * BV;
    hipLaunchKernelGGL(k_ncc_abc_d, dim3(vxm_blocks(BV, 256)), dim3(256), 0, s, sums, u1, BV, D, (long long)H * W, r, n);
    hipLaunchKernelGGL(k_box_axis, dim3(vxm_blocks(3 * BV, 256)), dim3(256), 0, s, u1, u2, 3 * BV, H, (long long)W, r);
    hipLaunchKernelGGL(k_ncc_grad_w, dim3(vxm_blocks(BV, 256)), dim3(256), 0, s, u2, I, J, gloss, gJ, BV, W, r);
}

}  // namespace

extern "C" {

/* 1 when vxm_ncc_fwd / vxm_ncc_bwd take the fused march for (B, win): callers size `sums` / `work` from this */
int vxm_ncc_fused(int B, int win) { return (win >= 3 && win <= 9 && B > 0 && B <= 65535) ? 1 : 0; }

int vxm_ncc_fwd(const float* I, const float* J, float* loss, float* sums, float* work, double* acc, int B, int D, int H, int W, int win,
                void* stream) {
    VXM_REQUIRE(I && J && loss && sums && work && acc, VXM_ERR_NULL_POINTER, "vxm_ncc_fwd: null pointer");
    VXM_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && win > 0 && (win & 1), VXM_ERR_BAD_SHAPE, "vxm_ncc_fwd: bad shape / even window %d", win);
    const long long V = (long long)D * H * W, BV = V * B;
    const int r = win / 2;
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(acc, 0, sizeof(double), s);
    if (vxm_ncc_fused(B, win)) {          // fused march; `sums` receives the (a, b, c) planes for backward
        const int seg = ncc_segment(B, D, H, W);
        const dim3 grid(((W + NF_TW - 1) / NF_TW) * ((H + NF_TH - 1) / NF_TH), (D + seg - 1) / seg, B);
        switch (r) {
            case 1: hipLaunchKernelGGL(k_ncc_fused_fwd<1>, grid, dim3(256), 0, s, I, J, sums, acc, D, H, W, BV, seg); break;
            case 2: hipLaunchKernelGGL(k_ncc_fused_fwd<2>, grid, dim3(256), 0, s, I, J, sums, acc, D, H, W, BV, seg); break;
            case 3: hipLaunchKernelGGL(k_ncc_fused_fwd<3>, grid, dim3(256), 0, s, I, J, sums, acc, D, H, W, BV, seg); break;
            default: hipLaunchKernelGGL(k_ncc_fused_fwd<4>, grid, dim3(256), 0, s, I, J, sums, acc, D, H, W, BV, seg); break;
        }
    } else {                                            // generic separable passes; `sums` receives the five box sums
        ncc_generic_fwd(I, J, sums, work, acc, BV, D, H, W, r, (float)win * win * win, s);
    }
    hipLaunchKernelGGL(k_finish_mean, dim3(1), dim3(64), 0, s, acc, loss, -1.0 / (double)BV);
    return vxm_check_launch("vxm_ncc_fwd");
}

int vxm_ncc_bwd(const float* I, const float* J, const float* sums, const float* gloss, float* gJ, float* work, int B, int D, int H, int W,
                int win, void* stream) {
    VXM_REQUIRE(I && J && sums && gloss && gJ && work, VXM_ERR_NULL_POINTER, "vxm_ncc_bwd: null pointer");
    VXM_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && win > 0 && (win & 1), VXM_ERR_BAD_SHAPE, "vxm_ncc_bwd: bad shape / even window %d", win);
    const long long V = (long long)D * H * W, BV = V * B;
    const int r = win / 2;
    hipStream_t s = VXM_STREAM(stream);
    if (vxm_ncc_fused(B, win)) {
        const int seg = ncc_segment(B, D, H, W);
        const dim3 grid(((W + NF_TW - 1) / NF_TW) * ((H + NF_TH - 1) / NF_TH), (D + seg - 1) / seg, B);
        switch (r) {
            case 1: hipLaunchKernelGGL(k_ncc_fused_bwd<1>, grid, dim3(256), 0, s, I, J, sums, gloss, gJ, D, H, W, BV, seg); break;
            case 2: hipLaunchKernelGGL(k_ncc_fused_bwd<2>, grid, dim3(256), 0, s, I, J, sums, gloss, gJ, D, H, W, BV, seg); break;
            case 3: hipLaunchKernelGGL(k_ncc_fused_bwd<3>, grid, dim3(256), 0, s, I, J, sums, gloss, gJ, D, H, W, BV, seg); break;
            default: hipLaunchKernelGGL(k_ncc_fused_bwd<4>, grid, dim3(256), 0, s, I, J, sums, gloss, gJ, D, H, W, BV, seg); break;
        }
        return vxm_check_launch("vxm_ncc_bwd");
    }
    ncc_generic_bwd(I, J, sums, gloss, gJ, work, BV, D, H, W, r, (float)win * win * win, s);
    return vxm_check_launch("vxm_ncc_bwd");
}

/* planar NCC (losses.py:15-67 with ndims = 2: win x win box filter, win_size = win^2): the separable passes with a depth of one */
int vxm_ncc2d_fwd(const float* I, const float* J, float* loss, float* sums, float* work, double* acc, int B, int H, int W, int win, void* stream) {
    VXM_REQUIRE(I && J && loss && sums && work && acc, VXM_ERR_NULL_POINTER, "vxm_ncc2d_fwd: null pointer");
    VXM_REQUIRE(B > 0 && H > 0 && W > 0 && win > 0 && (win & 1), VXM_ERR_BAD_SHAPE, "vxm_ncc2d_fwd: bad shape / even window %d", win);
    const long long BV = (long long)B * H * W;
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(acc, 0, sizeof(double), s);
    ncc_generic_fwd(I, J, sums, work, acc, BV, 1, H, W, win / 2, (float)win * win, s);
    hipLaunchKernelGGL(k_finish_mean, dim3(1), dim3(64), 0, s, acc, loss, -1.0 / (double)BV);
    return vxm_check_launch("vxm_ncc2d_fwd");
}

int vxm_ncc2d_bwd(const float* I, const float* J, const float* sums, const float* gloss, float* gJ, float* work, int B, int H, int W, int win,
                  void* stream) {
    VXM_REQUIRE(I && J && sums && gloss && gJ && work, VXM_ERR_NULL_POINTER, "vxm_ncc2d_bwd: null pointer");
    VXM_REQUIRE(B > 0 && H > 0 && W > 0 && win > 0 && (win & 1), VXM_ERR_BAD_SHAPE, "vxm_ncc2d_bwd: bad shape / even window %d", win);
    ncc_generic_bwd(I, J, sums, gloss, gJ, work, (long long)B * H * W, 1, H, W, win / 2, (float)win * win, VXM_STREAM(stream));
    return vxm_check_launch("vxm_ncc2d_bwd");
}

/* NCC on 1-D signals [B,1,L] (losses.py:15-67 with ndims = 1: conv1d box filter of `win` taps, win_size = win): the separable passes
 * with a depth and a height of one */
int vxm_ncc1d_fwd(const float* I, const float* J, float* loss, float* sums, float* work, double* acc, int B, int L, int win, void* stream) {
    VXM_REQUIRE(I && J && loss && sums && work && acc, VXM_ERR_NULL_POINTER, "vxm_ncc1d_fwd: null pointer");
    VXM_REQUIRE(B > 0 && L > 0 && win > 0 && (win & 1), VXM_ERR_BAD_SHAPE, "vxm_ncc1d_fwd: bad shape / even window %d", win);
    const long long BV = (long long)B * L;
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(acc, 0, sizeof(double), s);
    ncc_generic_fwd(I, J, sums, work, acc, BV, 1, 1, L, win / 2, (float)win, s);
    hipLaunchKernelGGL(k_finish_mean, dim3(1), dim3(64), 0, s, acc, loss, -1.0 / (double)BV);
    return vxm_check_launch("vxm_ncc1d_fwd");
}

int vxm_ncc1d_bwd(const float* I, const float* J, const float* sums, const float* gloss, float* gJ, float* work, int B, int L, int win,
                  void* stream) {
    VXM_REQUIRE(I && J && sums && gloss && gJ && work, VXM_ERR_NULL_POINTER, "vxm_ncc1d_bwd: null pointer");
    VXM_REQUIRE(B > 0 && L > 0 && win > 0 && (win & 1), VXM_ERR_BAD_SHAPE, "vxm_ncc1d_bwd: bad shape / even window %d", win);
    ncc_generic_bwd(I, J, sums, gloss, gJ, work, (long long)B * L, 1, 1, L, win / 2, (float)win, VXM_STREAM(stream));
    return vxm_check_launch("vxm_ncc1d_bwd");
}

/* NCC.loss with any window (losses.py:26-36,47-67): per axis `win` taps and `pad` zeros on both sides; the reference's rule is
 * pad = win[0] // 2 on every axis the tensor has (win 1 / pad 0 on the axes it does not have).  Box sums, cc and the mean live on
 * O = S + 2 pad - win + 1 per axis. */
struct NccWinShape { long long od, oh, ow, n_out, plane; };
static bool ncc_win_shape(int B, int D, int H, int W, int wd, int wh, int ww, int pd, int ph, int pw, NccWinShape& s) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || wd <= 0 || wh <= 0 || ww <= 0 || pd < 0 || ph < 0 || pw < 0) return false;
    s.od = (long long)D + 2 * pd - wd + 1; s.oh = (long long)H + 2 * ph - wh + 1; s.ow = (long long)W + 2 * pw - ww + 1;
    if (s.od <= 0 || s.oh <= 0 || s.ow <= 0) return false;       // the reference's conv raises there too (kernel larger than the padded input)
    const long long st1 = (long long)B * D * H * s.ow, st2 = (long long)B * D * s.oh * s.ow;
    s.n_out = (long long)B * s.od * s.oh * s.ow;
    s.plane = st1 > st2 ? st1 : st2;
    if (s.n_out > s.plane) s.plane = s.n_out;
    return true;
}

int64_t vxm_ncc_win_elems(int B, int D, int H, int W, int wd, int wh, int ww, int pd, int ph, int pw, int64_t* n_out) {
    NccWinShape sh;
    if (!ncc_win_shape(B, D, H, W, wd, wh, ww, pd, ph, pw, sh)) { if (n_out) *n_out = 0; return 0; }
    if (n_out) *n_out = sh.n_out;
    return sh.plane;
}

int vxm_ncc_win_fwd(const float* I, const float* J, float* loss, float* sums, float* work, double* acc, int B, int D, int H, int W,
                    int wd, int wh, int ww, int pd, int ph, int pw, void* stream) {
    VXM_REQUIRE(I && J && loss && sums && work && acc, VXM_ERR_NULL_POINTER, "vxm_ncc_win_fwd: null pointer");
    NccWinShape sh;
    VXM_REQUIRE(ncc_win_shape(B, D, H, W, wd, wh, ww, pd, ph, pw, sh), VXM_ERR_BAD_SHAPE,
                "vxm_ncc_win_fwd: bad shape [%d,%d,%d,%d] window (%d,%d,%d) pad (%d,%d,%d) (a window larger than the padded axis has no box sums)",
                B, D, H, W, wd, wh, ww, pd, ph, pw);
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(acc, 0, sizeof(double), s);
    float* t1 = work;                    // [5][B D H Ow]
    float* t2 = work + 5 * sh.plane;     // [5][B D Oh Ow]
    const long long rows = (long long)B * D * H, n1 = rows * sh.ow, n2 = (long long)B * D * sh.oh * sh.ow;
    hipLaunchKernelGGL(k_ncc_win_prod_w, dim3(vxm_blocks(n1, 256)), dim3(256), 0, s, I, J, t1, rows, W, (int)sh.ow, ww, pw);
    hipLaunchKernelGGL(k_box_axis_g, dim3(vxm_blocks(5 * n2, 256)), dim3(256), 0, s, t1, t2, 5 * n2, H, (int)sh.oh, sh.ow, wh, ph);
    hipLaunchKernelGGL(k_box_axis_g, dim3(vxm_blocks(5 * sh.n_out, 256)), dim3(256), 0, s, t2, sums, 5 * sh.n_out, D, (int)sh.od,
                       sh.oh * sh.ow, wd, pd);
    hipLaunchKernelGGL(k_ncc_win_cc, dim3(vxm_blocks(sh.n_out, 256)), dim3(256), 0, s, sums, acc, sh.n_out, (float)wd * wh * ww);
    hipLaunchKernelGGL(k_finish_mean, dim3(1), dim3(64), 0, s, acc, loss, -1.0 / (double)sh.n_out);
    return vxm_check_launch("vxm_ncc_win_fwd");
}

int vxm_ncc_win_bwd(const float* I, const float* J, const float* sums, const float* gloss, float* gJ, float* work, int B, int D, int H, int W,
                    int wd, int wh, int ww, int pd, int ph, int pw, void* stream) {
    VXM_REQUIRE(I && J && sums && gloss && gJ && work, VXM_ERR_NULL_POINTER, "vxm_ncc_win_bwd: null pointer");
    NccWinShape sh;
    VXM_REQUIRE(ncc_win_shape(B, D, H, W, wd, wh, ww, pd, ph, pw, sh), VXM_ERR_BAD_SHAPE,
                "vxm_ncc_win_bwd: bad shape [%d,%d,%d,%d] window (%d,%d,%d) pad (%d,%d,%d)", B, D, H, W, wd, wh, ww, pd, ph, pw);
    hipStream_t s = VXM_STREAM(stream);
    float* u0 = work;                    // [3][B Od Oh Ow], then [3][B D H Ow]
    float* u1 = work + 3 * sh.plane;     // [3][B D Oh Ow]
    const long long rows = (long long)B * D * H, n1 = rows * sh.ow, n2 = (long long)B * D * sh.oh * sh.ow;
    hipLaunchKernelGGL(k_ncc_win_abc, dim3(vxm_blocks(sh.n_out, 256)), dim3(256), 0, s, sums, u0, sh.n_out, (float)wd * wh * ww);
    hipLaunchKernelGGL(k_box_axis_g, dim3(vxm_blocks(3 * n2, 256)), dim3(256), 0, s, u0, u1, 3 * n2, (int)sh.od, D, sh.oh * sh.ow, wd,
                       wd - 1 - pd);
    hipLaunchKernelGGL(k_box_axis_g, dim3(vxm_blocks(3 * n1, 256)), dim3(256), 0, s, u1, u0, 3 * n1, (int)sh.oh, H, sh.ow, wh, wh - 1 - ph);
    hipLaunchKernelGGL(k_ncc_win_grad_w, dim3(vxm_blocks(rows * W, 256)), dim3(256), 0, s, u0, I, J, gloss, gJ, rows, W, (int)sh.ow, ww, pw,
                       (double)sh.n_out);
    return vxm_check_launch("vxm_ncc_win_bwd");
}

static int gradloss_fwd(const char* fn, const float* y, float* loss, double* acc, int B, int C, int D, int H, int W, int penalty, float mult,
                        int axes, void* stream) {
    VXM_REQUIRE(y && loss && acc, VXM_ERR_NULL_POINTER, "%s: null pointer", fn);
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && (axes == 2 || D > 1) && H > 1 && W > 1, VXM_ERR_BAD_SHAPE, "%s: bad shape", fn);
    VXM_REQUIRE(penalty == VXM_PENALTY_L1 || penalty == VXM_PENALTY_L2, VXM_ERR_UNSUPPORTED, "penalty can only be l1 or l2. Got: %d", penalty);
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(acc, 0, sizeof(double) * 3 * B * VXM_GRAD_SLOTS, s);
    const long long rows4 = ((long long)C * D * H + 3) / 4;
    const dim3 grid((unsigned)(rows4 > 8192 ? 8192 : rows4), B);          // 4 rows per block and pass; atomics spread over the slots
    if (!(W & 3) && !((uintptr_t)y & 15)) {
        const long long rowsb = ((long long)C * D * H + 4 * VXM_GRAD_RU - 1) / (4 * VXM_GRAD_RU);
        const dim3 gridv((unsigned)(rowsb > 1024 ? 1024 : rowsb), B);        // few blocks: the tail is the fp64 atomics (32 per address here)
        if (penalty == VXM_PENALTY_L2) hipLaunchKernelGGL(k_gradloss_fwd_v4<1>, gridv, dim3(256), 0, s, y, acc, C, D, H, W);
        else hipLaunchKernelGGL(k_gradloss_fwd_v4<0>, gridv, dim3(256), 0, s, y, acc, C, D, H, W);
    } else if (penalty == VXM_PENALTY_L2) hipLaunchKernelGGL(k_gradloss_fwd<1>, grid, dim3(256), 0, s, y, acc, C, D, H, W);
    else hipLaunchKernelGGL(k_gradloss_fwd<0>, grid, dim3(256), 0, s, y, acc, C, D, H, W);
    hipLaunchKernelGGL(k_gradloss_finish, dim3(1), dim3(64), 0, s, acc, loss, B, C, D, H, W, (double)mult, axes);
    return vxm_check_launch(fn);
}

static int gradloss_bwd(const char* fn, const float* y, const float* gloss, float* gy, int B, int C, int D, int H, int W, int penalty, float mult,
                        int axes, void* stream) {
    VXM_REQUIRE(y && gloss && gy, VXM_ERR_NULL_POINTER, "%s: null pointer", fn);
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && (axes == 2 || D > 1) && H > 1 && W > 1, VXM_ERR_BAD_SHAPE, "%s: bad shape", fn);
    VXM_REQUIRE(penalty == VXM_PENALTY_L1 || penalty == VXM_PENALTY_L2, VXM_ERR_UNSUPPORTED, "penalty can only be l1 or l2. Got: %d", penalty);
    const long long rows4 = ((long long)C * D * H + 3) / 4;
    const dim3 grid((unsigned)(rows4 > 16384 ? 16384 : rows4), B);        // 4 rows per block and pass
    if (!(W & 3) && !(((uintptr_t)y | (uintptr_t)gy) & 15)) {
        const long long rows8 = ((long long)C * D * H + 7) / 8;
        const dim3 gridv((unsigned)(rows8 > 16384 ? 16384 : rows8), B);
        if (penalty == VXM_PENALTY_L2)
            hipLaunchKernelGGL(k_gradloss_bwd_v4<1>, gridv, dim3(256), 0, VXM_STREAM(stream), y, gloss, gy, B, C, D, H, W, mult, axes);
        else hipLaunchKernelGGL(k_gradloss_bwd_v4<0>, gridv, dim3(256), 0, VXM_STREAM(stream), y, gloss, gy, B, C, D, H, W, mult, axes);
    } else if (penalty == VXM_PENALTY_L2)
        hipLaunchKernelGGL(k_gradloss_bwd<1>, grid, dim3(256), 0, VXM_STREAM(stream), y, gloss, gy, B, C, D, H, W, mult, axes);
    else hipLaunchKernelGGL(k_gradloss_bwd<0>, grid, dim3(256), 0, VXM_STREAM(stream), y, gloss, gy, B, C, D, H, W, mult, axes);
    return vxm_check_launch(fn);
}

int vxm_gradloss_fwd(const float* y, float* loss, double* acc, int B, int C, int D, int H, int W, int penalty, float mult, void* stream) {
    return gradloss_fwd("vxm_gradloss_fwd", y, loss, acc, B, C, D, H, W, penalty, mult, 3, stream);
}
int vxm_gradloss_bwd(const float* y, const float* gloss, float* gy, int B, int C, int D, int H, int W, int penalty, float mult, void* stream) {
    return gradloss_bwd("vxm_gradloss_bwd", y, gloss, gy, B, C, D, H, W, penalty, mult, 3, stream);
}
/* planar Grad (losses.py:102-135 with ndims = 2): the same kernels over [B,C,1,H,W], two axes */
int vxm_gradloss2d_fwd(const float* y, float* loss, double* acc, int B, int C, int H, int W, int penalty, float mult, void* stream) {
    return gradloss_fwd("vxm_gradloss2d_fwd", y, loss, acc, B, C, 1, H, W, penalty, mult, 2, stream);
}
int vxm_gradloss2d_bwd(const float* y, const float* gloss, float* gy, int B, int C, int H, int W, int penalty, float mult, void* stream) {
    return gradloss_bwd("vxm_gradloss2d_bwd", y, gloss, gy, B, C, 1, H, W, penalty, mult, 2, stream);
}

int vxm_mse_fwd(const float* a, const float* b, float* loss, double* acc, int64_t n, void* stream) {
    VXM_REQUIRE(a && b && loss && acc, VXM_ERR_NULL_POINTER, "vxm_mse_fwd: null pointer");
    VXM_REQUIRE(n > 0, VXM_ERR_BAD_SHAPE, "vxm_mse_fwd: empty input");
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(acc, 0, sizeof(double), s);
    hipLaunchKernelGGL(k_mse_fwd, dim3(reduce_blocks(n)), dim3(256), 0, s, a, b, acc, (long long)n);
    hipLaunchKernelGGL(k_finish_mean, dim3(1), dim3(64), 0, s, acc, loss, 1.0 / (double)n);
    return vxm_check_launch("vxm_mse_fwd");
}

int vxm_mse_bwd(const float* a, const float* b, const float* gloss, float* ga, float* gb, int64_t n, void* stream) {
    VXM_REQUIRE(a && b && gloss, VXM_ERR_NULL_POINTER, "vxm_mse_bwd: null pointer");
    VXM_REQUIRE(n > 0, VXM_ERR_BAD_SHAPE, "vxm_mse_bwd: empty input");
    if (!ga && !gb) return VXM_OK;
    hipLaunchKernelGGL(k_mse_bwd, dim3(stream_blocks(n)), dim3(256), 0, VXM_STREAM(stream), a, b, gloss, ga, gb, (long long)n);
    return vxm_check_launch("vxm_mse_bwd");
}

int vxm_dice_fwd(const float* yt, const float* yp, float* loss, double* acc, int B, int C, int64_t V, void* stream) {
    VXM_REQUIRE(yt && yp && loss && acc, VXM_ERR_NULL_POINTER, "vxm_dice_fwd: null pointer");
    VXM_REQUIRE(B > 0 && C > 0 && V > 0 && (long long)B * C <= 65535, VXM_ERR_BAD_SHAPE, "vxm_dice_fwd: bad shape");
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(acc, 0, sizeof(double) * 2 * B * C, s);
    const unsigned nb = stream_blocks(V) > 256 ? 256 : stream_blocks(V);
    hipLaunchKernelGGL(k_dice_fwd, dim3(nb, B * C), dim3(256), 0, s, yt, yp, acc, (long long)V);
    hipLaunchKernelGGL(k_dice_finish, dim3(1), dim3(64), 0, s, acc, loss, B * C);
    return vxm_check_launch("vxm_dice_fwd");
}

int vxm_dice_bwd(const float* yt, const float* yp, const double* acc, const float* gloss, float* gyt, float* gyp, int B, int C, int64_t V,
                 void* stream) {
    VXM_REQUIRE(yt && yp && acc && gloss, VXM_ERR_NULL_POINTER, "vxm_dice_bwd: null pointer");
    VXM_REQUIRE(B > 0 && C > 0 && V > 0 && (long long)B * C <= 65535, VXM_ERR_BAD_SHAPE, "vxm_dice_bwd: bad shape");
    if (!gyt && !gyp) return VXM_OK;
    const unsigned nb = stream_blocks(V) > 256 ? 256 : stream_blocks(V);
    hipLaunchKernelGGL(k_dice_bwd, dim3(nb, B * C), dim3(256), 0, VXM_STREAM(stream), yt, yp, acc, gloss, gyt, gyp, B * C, (long long)V);
    return vxm_check_launch("vxm_dice_bwd");
}

int vxm_loss_combine_fwd(const float* const* terms, const float* weights, int n, float* total, float* running, void* stream) {
    VXM_REQUIRE(terms && weights && total, VXM_ERR_NULL_POINTER, "vxm_loss_combine_fwd: null pointer");
    VXM_REQUIRE(n >= 1 && n <= VXM_LOSS_TERMS_MAX, VXM_ERR_BAD_SHAPE, "vxm_loss_combine_fwd: %d terms (1 .. %d)", n, VXM_LOSS_TERMS_MAX);
    LossTerms lt;
    lt.n = n;
    for (int i = 0; i < VXM_LOSS_TERMS_MAX; ++i) { lt.t[i] = i < n ? terms[i] : nullptr; lt.w[i] = i < n ? weights[i] : 0.0f; }
    for (int i = 0; i < n; ++i) VXM_REQUIRE(lt.t[i], VXM_ERR_NULL_POINTER, "vxm_loss_combine_fwd: term %d is null", i);
    hipLaunchKernelGGL(k_loss_combine_fwd, dim3(1), dim3(64), 0, VXM_STREAM(stream), lt, total, running);
    return vxm_check_launch("vxm_loss_combine_fwd");
}

int vxm_loss_combine_bwd(const float* gtotal, const float* weights, int n, float* gterms, void* stream) {
    VXM_REQUIRE(gtotal && weights && gterms, VXM_ERR_NULL_POINTER, "vxm_loss_combine_bwd: null pointer");
    VXM_REQUIRE(n >= 1 && n <= VXM_LOSS_TERMS_MAX, VXM_ERR_BAD_SHAPE, "vxm_loss_combine_bwd: %d terms (1 .. %d)", n, VXM_LOSS_TERMS_MAX);
    LossTerms lt;
    lt.n = n;
    for (int i = 0; i < VXM_LOSS_TERMS_MAX; ++i) { lt.t[i] = nullptr; lt.w[i] = i < n ? weights[i] : 0.0f; }
    hipLaunchKernelGGL(k_loss_combine_bwd, dim3(1), dim3(64), 0, VXM_STREAM(stream), gtotal, lt, gterms);
    return vxm_check_launch("vxm_loss_combine_bwd");
}

int vxm_add2(const float* a, const float* b, float* out, int64_t n, void* stream) {
    VXM_REQUIRE(a && b && out, VXM_ERR_NULL_POINTER, "vxm_add2: null pointer");
    VXM_REQUIRE(n > 0, VXM_ERR_BAD_SHAPE, "vxm_add2: n=%lld", (long long)n);
    hipLaunchKernelGGL(k_add2, dim3(vxm_blocks((n + 3) / 4, 256)), dim3(256), 0, VXM_STREAM(stream), a, b, out, (long long)n);
    return vxm_check_launch("vxm_add2");
}

int vxm_fill_zero(void* p, size_t bytes, void* stream) {
    VXM_REQUIRE(p || bytes == 0, VXM_ERR_NULL_POINTER, "vxm_fill_zero: null pointer");
    if (bytes == 0) return VXM_OK;
    const hipError_t e = hipMemsetAsync(p, 0, bytes, VXM_STREAM(stream));
    if (e != hipSuccess) return vxm_fail(VXM_ERR_HIP, "vxm_fill_zero: %s", hipGetErrorString(e));
    return VXM_OK;
}

int vxm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step,
                  float gscale, void* stream) {
    VXM_REQUIRE(p && g && m && v, VXM_ERR_NULL_POINTER, "vxm_adam_step: null pointer");
    VXM_REQUIRE(n > 0 && step >= 1, VXM_ERR_BAD_SHAPE, "vxm_adam_step: n=%lld step=%d", (long long)n, step);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(k_adam, dim3(vxm_blocks(n, 256)), dim3(256), 0, VXM_STREAM(stream), p, g, m, v, (long long)n, (float)(lr / bc1), beta1,
                       beta2, eps, (float)sqrt(bc2), gscale);
    return vxm_check_launch("vxm_adam_step");
}

int vxm_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, void* state,
                      float gscale, void* stream) {
    VXM_REQUIRE(p && g && m && v && state, VXM_ERR_NULL_POINTER, "vxm_adam_step_dev: null pointer");
    VXM_REQUIRE(n > 0 && (reinterpret_cast<uintptr_t>(state) & 7) == 0, VXM_ERR_BAD_SHAPE, "vxm_adam_step_dev: n=%lld / state alignment", (long long)n);
    AdamDevState* st = static_cast<AdamDevState*>(state);
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, VXM_STREAM(stream), st, lr, beta1, beta2);
    hipLaunchKernelGGL(k_adam_dev, dim3(vxm_blocks(n, 256)), dim3(256), 0, VXM_STREAM(stream), p, g, m, v, (long long)n, st, beta1, beta2, eps, gscale);
    return vxm_check_launch("vxm_adam_step_dev");
}

}  // extern "C"
