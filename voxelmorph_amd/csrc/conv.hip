// 3x3x3 convolution of the VxmDense U-Net as fp32 MFMA implicit GEMM on gfx950.
//
// Replaces (paths relative to the reference root):
//   voxelmorph/torch/networks.py:299-305  ConvBlock = Conv3d(k3,s1,p1) + LeakyReLU(0.2)
//   voxelmorph/torch/networks.py:211,257   flow Conv3d (no activation)
//   voxelmorph/torch/networks.py:137-138   Upsample(2,'nearest') + cat  (folded into the gather)
//   and their autograd twins (convolution_backward, leaky_relu_backward, ...).
//
// Formulation (forward and backward-data share one kernel; backward-data is the forward operator
// with the flipped/transposed weights):
//   D[co, voxel] = sum_{tap, ci} Wt[co, (tap,ci)] * X[(tap,ci), voxel]
//   -> v_mfma_f32_16x16x4_f32 with M = 16 output channels, N = 16 voxels (one W-row segment),
//      K = 4 input channels of one tap per instruction; exact fp32 (an fmaf chain).
// Block = 4 waves, output tile 4(D) x 4(H) x 16(W) voxels; wave w owns the 4 W-rows of depth
// slice w and all (<=32) output channels: 4 x NCT accumulators of 4 VGPRs.
// LDS per input-channel chunk (CK = 8 channels): the haloed input tile [CK][6][6][20] (plane
// stride 720 = 16 mod 32 -> the two k-groups of a half-wave read disjoint banks) and the packed
// weights of the chunk for all 27 taps in MFMA A-fragment order (one ds_read_b32 per fragment,
// lane-linear, conflict free).
//
// Backward-weight: gW[co,(ci,tap)] = sum_voxels dZ[co,v] * X[ci, v+tap]: M = 16 output channels,
// N = 16 (ci,tap) pairs, K = 4 voxels per instruction.  9 waves per block, each owning 3 N-tiles
// (for a 16-channel chunk: the 3 kw taps of one (kd,kh)); blocks are persistent over voxel tiles,
// keep their partial gW in registers and write it once; a second kernel reduces the per-block
// partials in a fixed order (deterministic).
#include "vxm_common.h"
#include "vxm_device.h"

namespace {

constexpr int TD = 4, TH = 4, TW = 16;          // output tile
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;   // haloed input tile (648 voxels)
constexpr int HVOX = HD * HH * HW;

struct ConvIn {                // virtual concat of two channel segments (see include/vxm_hip.h)
    const float* x0; const float* x1;
    long long bs0, bs1;
    int C0, C1, up0;
};

// Value of virtual input channel c of sample b at voxel (d,h,w); zero outside the volume
// (padding=1) and for channel padding.
__device__ __forceinline__ float load_in(const ConvIn& in, int b, int c, int d, int h, int w, int D, int H, int W) {
    if ((unsigned)d >= (unsigned)D || (unsigned)h >= (unsigned)H || (unsigned)w >= (unsigned)W) return 0.0f;
    if (c < in.C0) {
        if (in.up0) {
            const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1;
            return in.x0[(size_t)b * in.bs0 + ((size_t)c * D2 + (d >> 1)) * H2 * W2 + (size_t)(h >> 1) * W2 + (w >> 1)];
        }
        return in.x0[(size_t)b * in.bs0 + ((size_t)c * D + d) * H * W + (size_t)h * W + w];
    }
    c -= in.C0;
    if (c < in.C1) return in.x1[(size_t)b * in.bs1 + ((size_t)c * D + d) * H * W + (size_t)h * W + w];
    return 0.0f;
}

__device__ __forceinline__ void tile_origin(int tile, int D, int H, int W, int& b, int& d0, int& h0, int& w0) {
    const int nw = (W + TW - 1) / TW, nh = (H + TH - 1) / TH, nd = (D + TD - 1) / TD;
    const int tw = tile % nw; int t = tile / nw;
    const int th = t % nh; t /= nh;
    const int td = t % nd; b = t / nd;
    d0 = td * TD; h0 = th * TH; w0 = tw * TW;
}

// ------------------------------------------------------------------------------------------
// forward / backward-data kernel
// ------------------------------------------------------------------------------------------
constexpr int FWD_TWP = 20;
constexpr int FWD_PS = HD * HH * FWD_TWP;        // 720

template <int CK, int NCT>
__global__ void __launch_bounds__(256) k_conv3d_k3(ConvIn in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                   float* __restrict__ y, long long y_bs, int Cout, float act_slope,
                                                   const float* __restrict__ mask, long long mask_bs, float mask_slope,
                                                   int D, int H, int W, int Q) {
    VXM_DYN_SMEM(float, smem);
    float* Xs = smem;                               // [CK][FWD_PS]
    float* Ws = smem + CK * FWD_PS;                 // [27][CK/4][NCT][64]
    constexpr int KS = CK / 4;
    constexpr int WCHUNK = 27 * KS * NCT * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, n = lane & 15;
    int b, d0, h0, w0;
    tile_origin(blockIdx.x, D, H, W, b, d0, h0, w0);
    const int g = blockIdx.y;                       // output-channel group of 16*NCT
    const int Cin = in.C0 + in.C1;

    f32x4 acc[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ct][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int bbase = kq * FWD_PS + (wave * HH) * FWD_TWP + n;

    for (int q = 0; q < Q; ++q) {
        // ---- stage the haloed input tile of channels [q*CK, q*CK+CK)
        for (int i = tid; i < CK * HVOX; i += 256) {
            const int c = i / HVOX, rem = i - c * HVOX;
            const int dz = rem / (HH * HW), r2 = rem - dz * (HH * HW);
            const int hy = r2 / HW, wx = r2 - hy * HW;
            const int cg = q * CK + c;
            const float v = cg < Cin ? load_in(in, b, cg, d0 + dz - 1, h0 + hy - 1, w0 + wx - 1, D, H, W) : 0.0f;
            Xs[c * FWD_PS + (dz * HH + hy) * FWD_TWP + wx] = v;
        }
        // ---- stage the packed weights of this (group, chunk)
        {
            const float4* src = reinterpret_cast<const float4*>(wp + ((size_t)g * Q + q) * WCHUNK);
            float4* dst = reinterpret_cast<float4*>(Ws);
            for (int i = tid; i < WCHUNK / 4; i += 256) dst[i] = src[i];
        }
        __syncthreads();
        // ---- 27 taps x KS k-steps x (NCT x 4) MFMAs
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                float a[NCT], bv[4];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) a[ct] = Ws[((t * KS + s) * NCT + ct) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = Xs[bbase + s * 4 * FWD_PS + (kd * HH + r + kh) * FWD_TWP + kw];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[ct][r] = vxm_mfma16(a[ct], bv[r], acc[ct][r]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: bias + LeakyReLU (+ fused leaky_relu_backward mask), NCDHW store
    const int d = d0 + wave, w = w0 + n;
    if (d >= D || w >= W) return;
    const size_t V = (size_t)D * H * W;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = (g * NCT + ct) * 16 + kq * 4 + j;
            if (co >= Cout) continue;
            const float bz = bias ? bias[co] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = h0 + r;
                if (h >= H) continue;
                const size_t off = (size_t)co * V + ((size_t)d * H + h) * W + w;
                float v = acc[ct][r][j] + bz;
                v = v > 0.0f ? v : v * act_slope;
                if (mask) v *= vxm_lrelu_grad(mask[(size_t)b * mask_bs + off], mask_slope);
                y[(size_t)b * y_bs + off] = v;
            }
        }
    }
}

struct ConvCfg { int CK, NCT, Q, G; size_t elems; };
ConvCfg conv_cfg(int Cin, int Cout) {
    ConvCfg c;
    c.CK = Cin <= 4 ? 4 : 8;
    c.NCT = Cout <= 16 ? 1 : 2;
    c.Q = (Cin + c.CK - 1) / c.CK;
    c.G = (Cout + 16 * c.NCT - 1) / (16 * c.NCT);
    c.elems = (size_t)c.G * c.Q * 27 * (c.CK / 4) * c.NCT * 64;
    return c;
}

// w: [Cw_out][Cw_in][27] (reference layout).  Packed operator has Cin_p inputs / Cout_p outputs.
__global__ void __launch_bounds__(256) k_pack_weights(const float* __restrict__ w, float* __restrict__ wp, int Cw_in, int Cw_out,
                                                      int flip, int Cin_p, int Cout_p, int CK, int NCT, int Q, size_t elems) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    const int KS = CK / 4;
    size_t r = i;
    const int lane = r % 64; r /= 64;
    const int ct = r % NCT; r /= NCT;
    const int s = r % KS; r /= KS;
    const int t = r % 27; r /= 27;
    const int q = r % Q; const int g = (int)(r / Q);
    const int co = (g * NCT + ct) * 16 + (lane & 15);
    const int ci = q * CK + 4 * s + (lane >> 4);
    float v = 0.0f;
    if (co < Cout_p && ci < Cin_p) v = flip ? w[((size_t)ci * Cw_in + co) * 27 + (26 - t)] : w[((size_t)co * Cw_in + ci) * 27 + t];
    wp[i] = v;
}

// ------------------------------------------------------------------------------------------
// backward-weight kernel
// ------------------------------------------------------------------------------------------
constexpr int BW_PSX = 674;       // haloed plane (648) padded so that stride = 2 mod 32
constexpr int BW_PZ = 258;        // dZ plane (256 voxels) padded likewise
constexpr int BW_THREADS = 576;   // 9 waves
constexpr int BW_CKI = 16;        // input channels per chunk

template <int NCT>
__global__ void __launch_bounds__(BW_THREADS) k_conv3d_k3_bwd_weight(ConvIn in, const float* __restrict__ dz, long long dz_bs, int Cout,
                                                                    float* __restrict__ part, int B, int D, int H, int W) {
    VXM_DYN_SMEM(float, smem);
    float* Xs = smem;                          // [16][BW_PSX]
    float* Zs = smem + BW_CKI * BW_PSX;        // [16*NCT][BW_PZ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, n = lane & 15;
    const int Cin = in.C0 + in.C1;
    const int c0 = blockIdx.y * BW_CKI;
    const int ckc = min(BW_CKI, Cin - c0);
    const int nent = 27 * ckc, ntile = (nent + 15) / 16;
    const int cog = blockIdx.z * 16 * NCT;

    // N-tile j of this wave (slot i): entries e = j*16 + n  ->  (tap = e / ckc, channel = e % ckc)
    int boff[3];
    bool tile_ok[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = wave + 9 * i;
        tile_ok[i] = j < ntile;
        const int e = j * 16 + n;
        int off = 0;
        if (e < nent) {
            const int t = e / ckc, cl = e - t * ckc;
            off = cl * BW_PSX + ((t / 9) * HH + (t / 3) % 3) * HW + t % 3;
        }
        boff[i] = off;
    }
    f32x4 acc[3][NCT];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nw = (W + TW - 1) / TW, nh = (H + TH - 1) / TH, nd = (D + TD - 1) / TD;
    const int ntiles = B * nd * nh * nw;
    const size_t V = (size_t)D * H * W;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b, d0, h0, w0;
        tile_origin(tile, D, H, W, b, d0, h0, w0);
        for (int i = tid; i < ckc * HVOX; i += BW_THREADS) {
            const int c = i / HVOX, rem = i - c * HVOX;
            const int zz = rem / (HH * HW), r2 = rem - zz * (HH * HW);
            const int hy = r2 / HW, wx = r2 - hy * HW;
            Xs[c * BW_PSX + rem] = load_in(in, b, c0 + c, d0 + zz - 1, h0 + hy - 1, w0 + wx - 1, D, H, W);
        }
        for (int i = tid; i < 16 * NCT * 256; i += BW_THREADS) {
            const int co = i >> 8, v = i & 255;
            const int d = d0 + (v >> 6), h = h0 + ((v >> 4) & 3), w = w0 + (v & 15);
            float val = 0.0f;
            if (cog + co < Cout && d < D && h < H && w < W)
                val = dz[(size_t)b * dz_bs + (size_t)(cog + co) * V + ((size_t)d * H + h) * W + w];
            Zs[co * BW_PZ + v] = val;
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < 64; ++s) {
            // voxels 4s..4s+3 of the tile: row = s>>2 -> (dz, hy) = (row>>2, row&3), wx = 4*(s&3) + kq
            const int row = s >> 2;
            const int xbase = ((row >> 2) * HH + (row & 3)) * HW + 4 * (s & 3) + kq;
            float a[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) a[ct] = Zs[(ct * 16 + n) * BW_PZ + 4 * s + kq];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (!tile_ok[i]) continue;         // wave-uniform
                const float bv = Xs[boff[i] + xbase];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = vxm_mfma16(a[ct], bv, acc[i][ct]);
            }
        }
        __syncthreads();
    }

    // partial gW of this block: part[blockIdx.x][co][ci][tap]
    float* out = part + (size_t)blockIdx.x * Cout * Cin * 27;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (!tile_ok[i]) continue;
        const int e = (wave + 9 * i) * 16 + n;
        if (e >= nent) continue;
        const int t = e / ckc, ci = c0 + (e - t * ckc);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = cog + ct * 16 + kq * 4 + j;
                if (co < Cout) out[((size_t)co * Cin + ci) * 27 + t] = acc[i][ct][j];
            }
    }
}

__global__ void __launch_bounds__(256) k_reduce_partials(const float* __restrict__ part, float* __restrict__ gw, int nparts, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    gw[i] = s;
}

// bias gradient: gb[co] = sum_{b,v} dz[b,co,v]; one block per (co, slice), fp64 atomics into acc.
__global__ void __launch_bounds__(256) k_bias_grad(const float* __restrict__ dz, long long dz_bs, double* __restrict__ acc, int B, size_t V) {
    const int co = blockIdx.x;
    double s = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* p = dz + (size_t)b * dz_bs + (size_t)co * V;
        for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < V; i += (size_t)gridDim.y * 256) s += (double)p[i];
    }
    s = vxm_wave_sum(s);
    __shared__ double red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc + co, red[0] + red[1] + red[2] + red[3]);
}
__global__ void k_bias_finish(const double* __restrict__ acc, float* __restrict__ gb, int Cout) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < Cout) gb[i] = (float)acc[i];
}

int bw_nbx(int Cin, int Cout, int B, int D, int H, int W, int& Qc, int& G, int& NCT) {
    NCT = Cout <= 16 ? 1 : 2;
    Qc = (Cin + BW_CKI - 1) / BW_CKI;
    G = (Cout + 16 * NCT - 1) / (16 * NCT);
    const long long tiles = (long long)B * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    long long nbx = 1024 / ((long long)Qc * G);
    if (nbx < 1) nbx = 1;
    if (nbx > tiles) nbx = tiles;
    return (int)nbx;
}

int check_conv(const char* fn, int C0, int C1, int x0_up, int Cout, int B, int D, int H, int W) {
    VXM_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, VXM_ERR_BAD_SHAPE,
                "%s: bad shape B=%d C0=%d C1=%d Cout=%d D=%d H=%d W=%d", fn, B, C0, C1, Cout, D, H, W);
    VXM_REQUIRE(!x0_up || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), VXM_ERR_BAD_SHAPE,
                "%s: upsampled segment needs even extents, got %dx%dx%d", fn, D, H, W);
    VXM_REQUIRE((long long)(C0 + C1 > Cout ? C0 + C1 : Cout) * D * H * W < (1ll << 40), VXM_ERR_BAD_SHAPE, "%s: volume too large", fn);
    return VXM_OK;
}

}  // namespace

extern "C" {

size_t vxm_conv3d_k3_packed_elems(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return conv_cfg(Cin, Cout).elems;
}

int vxm_conv3d_k3_pack_weights(const float* w, float* wpacked, int Cin, int Cout, int transpose_flip, void* stream) {
    VXM_REQUIRE(w && wpacked, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_pack_weights: null pointer");
    VXM_REQUIRE(Cin > 0 && Cout > 0, VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_pack_weights: Cin=%d Cout=%d", Cin, Cout);
    const int cin_p = transpose_flip ? Cout : Cin, cout_p = transpose_flip ? Cin : Cout;
    const ConvCfg c = conv_cfg(cin_p, cout_p);
    hipLaunchKernelGGL(k_pack_weights, dim3(vxm_blocks((long long)c.elems, 256)), dim3(256), 0, VXM_STREAM(stream), w, wpacked,
                       Cin, Cout, transpose_flip, cin_p, cout_p, c.CK, c.NCT, c.Q, c.elems);
    return vxm_check_launch("vxm_conv3d_k3_pack_weights");
}

int vxm_conv3d_k3_fwd(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                      const float* wpacked, const float* bias, float* y, int64_t y_bstride, int Cout, float act_slope,
                      const float* mask_src, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W, void* stream) {
    if (int e = check_conv("vxm_conv3d_k3_fwd", C0, C1, x0_up, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(x0 && wpacked && y && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_fwd: null pointer");
    const ConvCfg c = conv_cfg(C0 + C1, Cout);
    ConvIn in{x0, x1, (long long)x0_bstride, (long long)x1_bstride, C0, C1, x0_up};
    const long long tiles = (long long)B * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_fwd: too many tiles");
    const dim3 grid((unsigned)tiles, c.G);
    const size_t lds = sizeof(float) * ((size_t)c.CK * FWD_PS + 27 * (c.CK / 4) * c.NCT * 64);
#define LAUNCH(CK_, NCT_) hipLaunchKernelGGL((k_conv3d_k3<CK_, NCT_>), grid, dim3(256), lds, VXM_STREAM(stream), in, wpacked, bias, y, \
        (long long)y_bstride, Cout, act_slope, mask_src, (long long)mask_bstride, mask_slope, D, H, W, c.Q)
    if (c.CK == 4 && c.NCT == 1) LAUNCH(4, 1);
    else if (c.CK == 4) LAUNCH(4, 2);
    else if (c.NCT == 1) LAUNCH(8, 1);
    else LAUNCH(8, 2);
#undef LAUNCH
    return vxm_check_launch("vxm_conv3d_k3_fwd");
}

size_t vxm_conv3d_k3_bwd_weight_workspace_bytes(int Cin, int Cout, int B, int D, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    int Qc, G, NCT;
    const int nbx = bw_nbx(Cin, Cout, B, D, H, W, Qc, G, NCT);
    return 256 + sizeof(double) * (size_t)((Cout + 31) / 32 * 32) + sizeof(float) * (size_t)nbx * Cout * Cin * 27;
}

int vxm_conv3d_k3_bwd_weight(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                             const float* dz, int64_t dz_bstride, int Cout, float* gw, float* gb, void* workspace,
                             size_t workspace_bytes, int B, int D, int H, int W, void* stream) {
    if (int e = check_conv("vxm_conv3d_k3_bwd_weight", C0, C1, x0_up, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(x0 && dz && gw && workspace && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_bwd_weight: null pointer");
    const int Cin = C0 + C1;
    VXM_REQUIRE(workspace_bytes >= vxm_conv3d_k3_bwd_weight_workspace_bytes(Cin, Cout, B, D, H, W), VXM_ERR_WORKSPACE,
                "vxm_conv3d_k3_bwd_weight: workspace too small (%zu bytes)", workspace_bytes);
    int Qc, G, NCT;
    const int nbx = bw_nbx(Cin, Cout, B, D, H, W, Qc, G, NCT);
    // workspace: [Cout doubles (bias accumulators), padded][partials]
    uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255;
    double* bacc = reinterpret_cast<double*>(base);
    float* part = reinterpret_cast<float*>(base + sizeof(double) * (size_t)((Cout + 31) / 32 * 32));
    ConvIn in{x0, x1, (long long)x0_bstride, (long long)x1_bstride, C0, C1, x0_up};
    const dim3 grid(nbx, Qc, G);
    const size_t lds = sizeof(float) * ((size_t)BW_CKI * BW_PSX + 16 * NCT * BW_PZ);
    // 74 KB of dynamic LDS (> the 64 KB default cap): opt in once per kernel
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_bwd_weight<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_bwd_weight<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        lds_opt_in = true;
    }
    if (NCT == 1)
        hipLaunchKernelGGL(k_conv3d_k3_bwd_weight<1>, grid, dim3(BW_THREADS), lds, VXM_STREAM(stream), in, dz, (long long)dz_bstride, Cout, part, B, D, H, W);
    else
        hipLaunchKernelGGL(k_conv3d_k3_bwd_weight<2>, grid, dim3(BW_THREADS), lds, VXM_STREAM(stream), in, dz, (long long)dz_bstride, Cout, part, B, D, H, W);
    const int n = Cout * Cin * 27;
    hipLaunchKernelGGL(k_reduce_partials, dim3(vxm_blocks(n, 256)), dim3(256), 0, VXM_STREAM(stream), part, gw, nbx, n);
    if (gb) {
        (void)hipMemsetAsync(bacc, 0, sizeof(double) * Cout, VXM_STREAM(stream));
        const size_t V = (size_t)D * H * W;
        const unsigned ny = (unsigned)(V / 16384 > 0 ? (V / 16384 > 256 ? 256 : V / 16384) : 1);
        hipLaunchKernelGGL(k_bias_grad, dim3(Cout, ny), dim3(256), 0, VXM_STREAM(stream), dz, (long long)dz_bstride, bacc, B, V);
        hipLaunchKernelGGL(k_bias_finish, dim3((Cout + 63) / 64), dim3(64), 0, VXM_STREAM(stream), bacc, gb, Cout);
    }
    return vxm_check_launch("vxm_conv3d_k3_bwd_weight");
}

}  // extern "C"
