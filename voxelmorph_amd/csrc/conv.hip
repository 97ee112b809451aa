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
#include <cstdlib>
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

__device__ __forceinline__ void tile_origin(int tile, int D, int H, int W, int& b, int& d0, int& h0, int& w0) {
    const int nw = (W + TW - 1) / TW, nh = (H + TH - 1) / TH, nd = (D + TD - 1) / TD;
    const int tw = tile % nw; int t = tile / nw;
    const int th = t % nh; t /= nh;
    const int td = t % nd; b = t / nd;
    d0 = td * TD; h0 = th * TH; w0 = tw * TW;
}

// Row-slab gather used by both conv kernels: one wave-instruction fetches 3 haloed rows (18 floats
// each, lanes 54..63 idle) of one (channel, depth) slab, so the channel / depth part of the address
// is wave-uniform (SALU) and the row / column part is a per-lane constant of the tile.
struct SlabLane {
    int rr, wx;        // row inside the 3-row group, column inside the haloed row
    int gh0, gw;       // global row of row-group 0 and global column of this lane (may be -1 / >= extent)
    bool act, wok;     // lane carries data; column inside the volume
};

__device__ __forceinline__ SlabLane make_slab_lane(int lane, int h0, int w0, int W) {
    SlabLane L;
    L.rr = lane / HW; L.wx = lane - L.rr * HW;
    L.act = lane < 3 * HW;
    L.gw = w0 + L.wx - 1; L.gh0 = h0 + L.rr - 1;
    L.wok = L.act && (unsigned)L.gw < (unsigned)W;
    return L;
}

// Value of virtual input channel cg at depth d for this lane's (row, column).  b, cg, d, hb are
// wave-uniform: the 64-bit base is SALU math (s_cselect, no branches), the lane contributes a 32-bit
// offset.  The load is UNCONDITIONAL on a clamped in-bounds address and the padding zeros are applied
// by a select afterwards: branch-free, so the unrolled loads of one chunk issue back-to-back.
__device__ __forceinline__ float slab_load(const float* x0, const float* x1, long long bs0, long long bs1, int C0, int C1, int up0,
                                           const SlabLane& L, int b, int cg, int d, int hb, int D, int H, int W) {
    const bool uok = (unsigned)d < (unsigned)D && cg < C0 + C1;       // uniform validity
    const int cgc = min(cg, C0 + C1 - 1), dc = min(max(d, 0), D - 1);
    const bool s0 = cgc < C0;
    const int sh = (s0 && up0) ? 1 : 0;                               // x2 nearest upsampling of segment 0
    const float* p = s0 ? x0 + (size_t)b * bs0 : x1 + (size_t)b * bs1;
    const int cc = s0 ? cgc : cgc - C0;
    const int Ds = D >> sh, Hs = H >> sh, Ws = W >> sh;
    const float* base = p + ((size_t)cc * Ds + (dc >> sh)) * Hs * Ws;
    const int gh = L.gh0 + 3 * hb;
    const bool ok = L.wok && (unsigned)gh < (unsigned)H;
    // 32-bit BYTE offset (planes are < 4 GB): lets the load use the SGPR-base + 32-bit-VGPR-offset form
    const unsigned boff = ok ? (unsigned)((gh >> sh) * Ws + (L.gw >> sh)) << 2 : 0u;
    const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff);
    return (ok && uok) ? v : 0.0f;
}

// ------------------------------------------------------------------------------------------
// buffer-descriptor helpers.  Tiles are zero padded through the descriptor: a lane whose offset is beyond
// num_records loads 0.0 (and an LDS-DMA lane writes 0.0 -- probed on gfx950, tools/probe/ldsdma_probe.hip).
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int VXM_OOB = (int)0x80000000;            // voffset beyond any num_records -> the lane loads 0.0

__device__ __forceinline__ __amdgpu_buffer_rsrc_t vxm_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void vxm_lds_dma4(__amdgpu_buffer_rsrc_t r, float* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 4, voff, soff, 0, 0);      // lane l -> LDS base + 4 l
}
__device__ __forceinline__ void vxm_lds_dma16(__amdgpu_buffer_rsrc_t r, float* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, soff, 0, 0);     // lane l -> LDS base + 16 l
}


// ------------------------------------------------------------------------------------------
// forward / backward-data kernel
// ------------------------------------------------------------------------------------------
constexpr int FWD_TWP = 20;
constexpr int BV_RS_FWD = 20;          // row stride of the wide-load tile layouts (interior at columns 2..17)
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

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> SALU address math
    const int kq = lane >> 4, n = lane & 15;
    int b, d0, h0, w0;
    tile_origin(blockIdx.x, D, H, W, b, d0, h0, w0);
    const int g = blockIdx.y;                       // output-channel group of 16*NCT
    const float* const ix0 = in.x0; const float* const ix1 = in.x1;
    const long long ibs0 = in.bs0, ibs1 = in.bs1;
    const int iC0 = in.C0, iC1 = in.C1, iup0 = in.up0;

    f32x4 acc[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ct][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int bbase = kq * FWD_PS + (wave * HH) * FWD_TWP + n;

    // ---- staging: global -> registers (prefetched under the previous chunk's MFMAs) -> LDS
    constexpr int XIT = CK * 3;                         // 3-row slab groups per wave per chunk
    constexpr int WIT = (WCHUNK / 4 + 255) / 256;       // float4 weight pieces per thread per chunk
    const SlabLane L = make_slab_lane(lane, h0, w0, W);
    const int lds_lane = L.rr * FWD_TWP + L.wx;
    float xv[XIT];
    f32x4 wv[WIT];

    auto prefetch = [&](int q) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int c = wave * (CK / 4) + it / 12, dz = (it % 12) >> 1, hb = it & 1;   // channel / depth: wave-uniform
            xv[it] = slab_load(ix0, ix1, ibs0, ibs1, iC0, iC1, iup0, L, b, q * CK + c, d0 + dz - 1, hb, D, H, W);
        }
        const f32x4* src = reinterpret_cast<const f32x4*>(wp + ((size_t)g * Q + q) * WCHUNK);
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + it * 256;
            wv[it] = src[min(i, WCHUNK / 4 - 1)];      // clamped: the LDS store below is predicated
        }
    };

    prefetch(0);
    for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int c = wave * (CK / 4) + it / 12, dz = (it % 12) >> 1, hb = it & 1;
            if (L.act) Xs[c * FWD_PS + (dz * HH + hb * 3) * FWD_TWP + lds_lane] = xv[it];
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + it * 256;
            if (i < WCHUNK / 4) reinterpret_cast<f32x4*>(Ws)[i] = wv[it];
        }
        __syncthreads();
        if (q + 1 < Q) prefetch(q + 1);                 // loads stay in flight across the MFMA phase
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

    // ---- epilogue: bias + LeakyReLU (+ fused leaky_relu_backward mask), NCDHW store.
    // All bias / mask loads are issued first (clamped addresses, no branches), then the math + stores.
    const int d = d0 + wave, w = w0 + n;
    const size_t V = (size_t)D * H * W;
    const bool vox_ok = d < D && w < W;
    const size_t vox_off = ((size_t)min(d, D - 1) * H) * W + min(w, W - 1);
    float bz[NCT][4], mk[NCT][4][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bz[ct][j] = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) mk[ct][j][r] = 1.0f;
        }
    if (bias) {                                   // hoisted uniform tests: the loads inside issue back-to-back
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) bz[ct][j] = bias[min((g * NCT + ct) * 16 + kq * 4 + j, Cout - 1)];
    }
    if (mask) {
        const float* mb = mask + (size_t)b * mask_bs + vox_off;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = min((g * NCT + ct) * 16 + kq * 4 + j, Cout - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) mk[ct][j][r] = mb[(size_t)co * V + (size_t)min(h0 + r, H - 1) * W];
            }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mk[ct][j][r] = vxm_lrelu_grad(mk[ct][j][r], mask_slope);
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = (g * NCT + ct) * 16 + kq * 4 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = h0 + r;
                float v = acc[ct][r][j] + bz[ct][j];
                v = (v > 0.0f ? v : v * act_slope) * mk[ct][j][r];
                if (vox_ok && h < H && co < Cout)
                    y[(size_t)b * y_bs + (size_t)co * V + ((size_t)d * H + h) * W + w] = v;
            }
        }
}


// ------------------------------------------------------------------------------------------
// forward / backward-data kernel, 8-wave version for the large layers (W % 4 == 0, 16-byte aligned tensors)
// ------------------------------------------------------------------------------------------
// Same implicit GEMM as k_conv3d_k3, restructured so that the per-chunk staging costs almost nothing:
//  * block = 8 waves, output tile 8(D) x 4(H) x 16(W) voxels, wave w = depth slice w; two blocks per CU;
//  * wave w stages channel w of every 8-channel chunk: its per-lane source offsets are computed ONCE per block
//    (they depend on the tile only; the channel moves the wave-uniform soffset), the zero padding comes from the
//    buffer descriptor, interior rows are dwordx4 loads (dwordx2 + duplicate through the x2-upsampled segment);
//  * the loads of chunk q+1 (X plane pieces + packed weights) are in flight in registers under the MFMAs of chunk q;
//  * the MFMA loop is fully unrolled, branch-free, with the operands of step s+1 requested before the MFMAs of s.
// X chunk in LDS: [8][10][6][20], interior columns at 2..17, halo columns at 1 / 18; plane stride 1200 = 16 mod 32.
constexpr int T8_TD = 8;
constexpr int T8_PS = (T8_TD + 2) * HH * BV_RS_FWD;   // 1200
constexpr int T8_THREADS = 512;

template <int NCT>
__global__ void __launch_bounds__(T8_THREADS, 4) k_conv3d_k3_t8(ConvIn in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                              float* __restrict__ y, long long y_bs, int Cout, float act_slope,
                                                              const float* __restrict__ mask, long long mask_bs, float mask_slope,
                                                              int B, int D, int H, int W, int Q) {
    VXM_DYN_SMEM(float, smem);
    constexpr int CK = 8, KS = 2, RS = BV_RS_FWD;
    constexpr int WCHUNK = 27 * KS * NCT * 64;
    constexpr int WIT = (WCHUNK / 4 + T8_THREADS - 1) / T8_THREADS;
    float* const Xs = smem;                         // [CK][T8_PS]
    float* const Ws = smem + CK * T8_PS;            // [27][KS][NCT][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n = lane & 15;
    const float* const ix0 = in.x0; const float* const ix1 = in.x1;
    const int iC0 = in.C0, iC1 = in.C1, iup0 = in.up0;

    // tile of this block: block b runs on XCD b % 8 (observed; speed only), XCD x takes the contiguous tile range
    // [nt x / 8, nt (x+1) / 8) so that concurrently running neighbours share halo lines and weights in one L2
    const int nw = (W + TW - 1) / TW, nh = (H + TH - 1) / TH, nd = (D + T8_TD - 1) / T8_TD;
    const int ntiles = B * nd * nh * nw;
    int tile = blockIdx.x;
    if (ntiles >= 64) {
        const int x = tile & 7, j = tile >> 3;
        const int lo = (int)((long long)ntiles * x / 8), hi = (int)((long long)ntiles * (x + 1) / 8);
        tile = lo + j;
        if (tile >= hi) return;                    // grid is rounded up to 8 x ceil(nt / 8)
    } else if (tile >= ntiles) {
        return;
    }
    const int tw = tile % nw; int tq = tile / nw;
    const int th = tq % nh; tq /= nh;
    const int td = tq % nd; const int b = tq / nd;
    const int d0 = td * T8_TD, h0 = th * TH, w0 = tw * TW;
    const int g = blockIdx.y;                       // output-channel group of 16*NCT

    const int V = D * H * W;
    const int Hs = H >> 1, Ws2 = W >> 1;
    const int V0 = iup0 ? (D >> 1) * Hs * Ws2 : V;
    const __amdgpu_buffer_rsrc_t r0 = vxm_rsrc(ix0 + (size_t)b * in.bs0, (unsigned)iC0 * (unsigned)V0 * 4u);
    const __amdgpu_buffer_rsrc_t r1 = vxm_rsrc(iC1 ? ix1 + (size_t)b * in.bs1 : ix0, (unsigned)iC1 * (unsigned)V * 4u);

    // staging roles of this lane: interior slot 64 j + lane (< 240) -> row 16 j + (lane >> 2) of the [10][6] row grid,
    // columns 4 (lane & 3)..+3; halo slot 64 j + lane (< 120) -> row 32 j + (lane >> 1), side lane & 1.
    const int lq = lane & 3, lr4 = lane >> 2, lr2 = lane >> 1, hside = lane & 1;
    const int ibase = lr4 * RS + 2 + 4 * lq;
    const int hbase = lr2 * RS + (hside ? 18 : 1);
    int vi[4], vh[2];              // byte offsets inside a plane; for the upsampled segment: of the half-resolution source
    int viu[4], vhu[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rr = 16 * j + lr4;
        const int gd = d0 - 1 + rr / HH, gh = h0 - 1 + rr % HH, gw = w0 + 4 * lq;
        const bool ok = rr < (T8_TD + 2) * HH && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw < W;
        vi[j] = ok ? ((gd * H + gh) * W + gw) << 2 : VXM_OOB;
        viu[j] = ok ? (((gd >> 1) * Hs + (gh >> 1)) * Ws2 + (gw >> 1)) << 2 : VXM_OOB;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rr = 32 * j + lr2;
        const int gd = d0 - 1 + rr / HH, gh = h0 - 1 + rr % HH, gw = hside ? w0 + TW : w0 - 1;
        const bool ok = rr < (T8_TD + 2) * HH && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        vh[j] = ok ? ((gd * H + gh) * W + gw) << 2 : VXM_OOB;
        vhu[j] = ok ? (((gd >> 1) * Hs + (gh >> 1)) * Ws2 + (gw >> 1)) << 2 : VXM_OOB;
    }

    f32x4 acc[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ct][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 xi[4];
    float xh[2];
    f32x4 wv[WIT];
    auto load_chunk = [&](int q) __attribute__((always_inline)) {
        const int cg = q * CK + wave;               // channel staged by this wave (wave-uniform)
        if (cg < iC0 + iC1) {
            if (cg < iC0 && iup0) {
                const int soff = cg * V0 * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r0, viu[j], soff, 0));
                    xi[j] = (f32x4){t.x, t.x, t.y, t.y};
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) xh[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, vhu[j], soff, 0));
            } else {
                const bool s0 = cg < iC0;
                const __amdgpu_buffer_rsrc_t r = s0 ? r0 : r1;
                const int soff = (s0 ? cg * V0 : (cg - iC0) * V) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) xi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vi[j], soff, 0));
#pragma unroll
                for (int j = 0; j < 2; ++j) xh[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vh[j], soff, 0));
            }
        } else {                                    // channel padding of the last chunk
#pragma unroll
            for (int j = 0; j < 4; ++j) xi[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            xh[0] = xh[1] = 0.0f;
        }
        const __amdgpu_buffer_rsrc_t rw = vxm_rsrc(wp + ((size_t)g * Q + q) * WCHUNK, WCHUNK * 4u);
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            wv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tid + T8_THREADS * it) * 16, 0, 0));
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
        float* dst = Xs + wave * T8_PS;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < 3 || lane < 48) {               // 240 interior slots
                *reinterpret_cast<f32x2*>(dst + ibase + 16 * j * RS) = (f32x2){xi[j].x, xi[j].y};
                *reinterpret_cast<f32x2*>(dst + ibase + 16 * j * RS + 2) = (f32x2){xi[j].z, xi[j].w};
            }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (j < 1 || lane < 56) dst[hbase + 32 * j * RS] = xh[j];       // 120 halo slots
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + T8_THREADS * it;
            if (i < WCHUNK / 4) reinterpret_cast<f32x4*>(Ws)[i] = wv[it];
        }
    };

    const int bbase = kq * T8_PS + wave * HH * RS + n + 1;      // + 4 s planes, + (kd, r + kh) rows, + kw
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int q = 0; q < Q; ++q) {
        if (q + 1 < Q) load_chunk(q + 1);           // in flight under the MFMAs below
        // ---- 27 taps x KS k-steps x (NCT x 4) MFMAs, operands double-buffered in registers
        float a[2][NCT], bv[2][4];
        auto fetch = [&](int st, float (&af)[NCT], float (&bf)[4]) __attribute__((always_inline)) {
            const int t = st / KS, s = st % KS;
            const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) af[ct] = Ws[((t * KS + s) * NCT + ct) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) bf[r] = Xs[bbase + s * 4 * T8_PS + (kd * HH + r + kh) * RS + kw];
        };
        fetch(0, a[0], bv[0]);
#pragma unroll
        for (int st = 0; st < 27 * KS; ++st) {
            if (st + 1 < 27 * KS) fetch(st + 1, a[(st + 1) & 1], bv[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ct][r] = vxm_mfma16(a[st & 1][ct], bv[st & 1][r], acc[ct][r]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q + 1 < Q) {
            __syncthreads();                        // every wave is done reading chunk q
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue: bias + LeakyReLU (+ fused leaky_relu_backward mask), NCDHW store
    const int d = d0 + wave, w = w0 + n;
    const bool vox_ok = d < D && w < W;
    const size_t vox_off = ((size_t)min(d, D - 1) * H) * W + min(w, W - 1);
    float bz[NCT][4], mk[NCT][4][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bz[ct][j] = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) mk[ct][j][r] = 1.0f;
        }
    if (bias) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) bz[ct][j] = bias[min((g * NCT + ct) * 16 + kq * 4 + j, Cout - 1)];
    }
    if (mask) {
        const float* mb = mask + (size_t)b * mask_bs + vox_off;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = min((g * NCT + ct) * 16 + kq * 4 + j, Cout - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) mk[ct][j][r] = mb[(size_t)co * V + (size_t)min(h0 + r, H - 1) * W];
            }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mk[ct][j][r] = vxm_lrelu_grad(mk[ct][j][r], mask_slope);
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = (g * NCT + ct) * 16 + kq * 4 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = h0 + r;
                float v = acc[ct][r][j] + bz[ct][j];
                v = (v > 0.0f ? v : v * act_slope) * mk[ct][j][r];
                if (vox_ok && h < H && co < Cout)
                    y[(size_t)b * y_bs + (size_t)co * V + ((size_t)d * H + h) * W + w] = v;
            }
        }
}

// ------------------------------------------------------------------------------------------
// forward kernel for cat([upsample2(x0), x1]) inputs with the upsampled segment at LOW resolution cost
// ------------------------------------------------------------------------------------------
// A 3x3x3 conv over a nearest-x2-upsampled tensor touches, per axis, only 2 distinct low-resolution inputs: for an
// output coordinate o = 2m + p the taps {-1,0,+1} land on low-res indices (m-1, m, m) for p = 0 and (m, m, m+1) for
// p = 1.  Summing the weights that share an input ("collapsed" 2x2x2 kernels, one per output parity (pd,ph,pw)):
//     y[co, o] += sum_{ci in seg 0} sum_{j in {0,1}^3} Wc[p(o)][co,ci,j] * x0[ci, m(o) + j - 1 + p(o)]
// does the upsampled segment with 8 instead of 27 MACs per (ci, co, voxel) -- 3.4x fewer MFMAs for 2/3 of the
// input channels of the largest layer -- and reads x0 at its own resolution.  Border behaviour is unchanged: low-res
// index -1 / M is exactly the zero padding of the upsampled volume.  (Weights are pre-summed, so the result differs
// from the tap-by-tap sum by fp32 rounding only: within the conv tolerance of DESIGN.md §2.)
// An MFMA's 16 voxel columns must share the parity class (they share the A = weight fragment): with the 8x4x16 tile of
// k_conv3d_k3_t8 (wave = depth slice -> pd), a wave's four N-tiles are (ph, pw) in {0,1}^2, lane n holding voxel
// (row ph + 2 (n >> 3), w = 2 (n & 7) + pw).  Segment 1 (the skip connection, full resolution) runs the regular 27-tap
// loop on the same accumulators; its LDS rows are de-interleaved by column parity ([even cols | odd cols]) so that the
// stride-2 voxel pattern reads consecutive words.
constexpr int TU_RS = 24;                               // seg-1 row: [even block 12 | odd block 12]
constexpr int TU_PS1 = (T8_TD + 2) * HH * TU_RS + 8;    // 1448 = 8 mod 32
constexpr int TU_PSL = 6 * 4 * 16 + 8;                  // low-res plane [6][4][16] -> 392 = 8 mod 32
constexpr int TU_CK0 = 8;                               // segment-0 channels per chunk (two k-steps)

template <int NCT> constexpr int tu_wc_floats() { return 8 * 8 * 2 * NCT * 64; }      // collapsed weights of a seg-0 chunk [par][j][ks][ct][64]
template <int NCT> constexpr int tu_w1_floats() { return 27 * 2 * NCT * 64; }         // packed weights of a seg-1 chunk
template <int NCT> constexpr int tu_s0_floats() { return TU_CK0 * TU_PSL + tu_wc_floats<NCT>(); }   // seg-0 stage: [weights][X]
template <int NCT> constexpr int tu_lds_floats() {      // the larger of the two stage kinds (single buffered; chunk q+1 waits in registers)
    return (8 * TU_PS1 + tu_w1_floats<NCT>()) > tu_s0_floats<NCT>() ? (8 * TU_PS1 + tu_w1_floats<NCT>()) : tu_s0_floats<NCT>();
}

template <int NCT>
__global__ void __launch_bounds__(T8_THREADS, 4) k_conv3d_k3_t8u(const float* __restrict__ x0, long long bs0, int C0, const float* __restrict__ x1,
                                                               long long bs1, int C1, const float* __restrict__ wp, const float* __restrict__ bias,
                                                               float* __restrict__ y, long long y_bs, int Cout, float act_slope,
                                                               int B, int D, int H, int W) {
    VXM_DYN_SMEM(float, smem);
    constexpr int WIT = (tu_wc_floats<NCT>() / 4 + T8_THREADS - 1) / T8_THREADS;       // >= the seg-1 chunk's pieces as well
    static_assert(tu_w1_floats<NCT>() <= tu_wc_floats<NCT>(), "weight staging registers sized by the seg-0 chunk");
    static_assert(NCT == 1, "the 2-tile instance does not fit 128 VGPRs (see up_nct)");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n = lane & 15;
    const int Q0 = (C0 + TU_CK0 - 1) / TU_CK0, Q1 = (C1 + 7) / 8, Q = Q0 + Q1;

    const int nw = (W + TW - 1) / TW, nh = (H + TH - 1) / TH, nd = (D + T8_TD - 1) / T8_TD;
    const int ntiles = B * nd * nh * nw;
    int tile = blockIdx.x;
    if (ntiles >= 64) {                              // XCD-contiguous tile numbering, as k_conv3d_k3_t8
        const int x = tile & 7, j = tile >> 3;
        const int lo = (int)((long long)ntiles * x / 8), hi = (int)((long long)ntiles * (x + 1) / 8);
        tile = lo + j;
        if (tile >= hi) return;
    } else if (tile >= ntiles) {
        return;
    }
    const int tw = tile % nw; int tq = tile / nw;
    const int th = tq % nh; tq /= nh;
    const int td = tq % nd; const int b = tq / nd;
    const int d0 = td * T8_TD, h0 = th * TH, w0 = tw * TW;
    const int g = blockIdx.y;

    const int V = D * H * W;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1, Vl = Dl * Hl * Wl;
    const __amdgpu_buffer_rsrc_t r0 = vxm_rsrc(x0 + (size_t)b * bs0, (unsigned)C0 * (unsigned)Vl * 4u);
    const __amdgpu_buffer_rsrc_t r1 = vxm_rsrc(C1 ? x1 + (size_t)b * bs1 : x0, (unsigned)C1 * (unsigned)V * 4u);

    // ---- staging roles.  Segment 1: as k_conv3d_k3_t8 (wave w stages channel w of the chunk).  The per-lane offsets are
    // recomputed per chunk (a few dozen VALU against >= 128 MFMAs) instead of living in 10 VGPRs: the kernel sits at the
    // 128-register limit of 4 waves / SIMD.
    const int lq = lane & 3, lr4 = lane >> 2, lr2 = lane >> 1, hside = lane & 1;
    auto off_interior = [&](int j) __attribute__((always_inline)) -> int {
        const int rr = 16 * j + lr4;
        const int gd = d0 - 1 + rr / HH, gh = h0 - 1 + rr % HH, gw = w0 + 4 * lq;
        const bool ok = rr < (T8_TD + 2) * HH && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw < W;
        return ok ? ((gd * H + gh) * W + gw) << 2 : VXM_OOB;
    };
    auto off_halo = [&](int j) __attribute__((always_inline)) -> int {
        const int rr = 32 * j + lr2;
        const int gd = d0 - 1 + rr / HH, gh = h0 - 1 + rr % HH, gw = hside ? w0 + TW : w0 - 1;
        const bool ok = rr < (T8_TD + 2) * HH && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        return ok ? ((gd * H + gh) * W + gw) << 2 : VXM_OOB;
    };
    // LDS positions of a seg-1 row piece: columns c0..c0+3 (c0 = 4 lq) -> (col + 1) = 4 lq + 1..4: the odd ones go to the odd
    // block at [12 + 2 lq, +1], the even ones to the even block (shifted by one word) at [2 lq + 2, +1]: two aligned b64 stores
    const int s1_odd = lr4 * TU_RS + 12 + 2 * lq, s1_even = lr4 * TU_RS + 2 * lq + 2;
    const int s1_halo = lr2 * TU_RS + (hside ? 20 : 1);          // col 16 -> odd index 8; col -1 -> even index 0 (+1 shift)
    // Segment 0 (low resolution): slot = tid + 512 k (< 1920) -> channel c = slot / 240, (dl, hl, wl) of the [6][4][10] region
    auto low_slot = [&](int k, int& voff, int& lds) __attribute__((always_inline)) {
        const int slot = tid + T8_THREADS * k;
        const int c = slot / 240, r = slot - c * 240, dl = r / 40, hl = (r % 40) / 10, wl = r % 10;
        const int gd = (d0 >> 1) - 1 + dl, gh = (h0 >> 1) - 1 + hl, gw = (w0 >> 1) - 1 + wl;
        const bool ok = slot < 1920 && (unsigned)gd < (unsigned)Dl && (unsigned)gh < (unsigned)Hl && (unsigned)gw < (unsigned)Wl;
        voff = ok ? (c * Vl + (gd * Hl + gh) * Wl + gw) << 2 : VXM_OOB;        // channel of the chunk in the per-lane offset
        lds = slot < 1920 ? c * TU_PSL + (dl * 4 + hl) * 16 + wl : -1;
    };

    f32x4 acc[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ct][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 xi[4];               // seg-1 interior pieces; seg-0: xi[0] carries the four low-resolution words of this thread
    float xh[2];
    f32x4 wv[WIT];
    auto wchunk_base = [&](int q) __attribute__((always_inline)) -> size_t {       // packed layout: [g][Q0 collapsed chunks][Q1 regular chunks]
        const size_t per_g = (size_t)Q0 * tu_wc_floats<NCT>() + (size_t)Q1 * tu_w1_floats<NCT>();
        return (size_t)g * per_g + (q < Q0 ? (size_t)q * tu_wc_floats<NCT>() : (size_t)Q0 * tu_wc_floats<NCT>() + (size_t)(q - Q0) * tu_w1_floats<NCT>());
    };
    auto load_chunk = [&](int q) __attribute__((always_inline)) {
        if (q < Q0) {                                 // low-resolution chunk: 4 channels x [6][4][10]
            const int soff = q * TU_CK0 * Vl * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int vo, ld;
                low_slot(k, vo, ld);
                // channels beyond C0 in the last chunk: offset beyond num_records -> 0.0 (descriptor covers C0 planes)
                xi[0][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, vo, soff, 0));
            }
        } else {
            const int cg = (q - Q0) * 8 + wave;       // channel of segment 1 staged by this wave
            if (cg < C1) {
                const int soff = cg * V * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) xi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r1, off_interior(j), soff, 0));
#pragma unroll
                for (int j = 0; j < 2; ++j) xh[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r1, off_halo(j), soff, 0));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) xi[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                xh[0] = xh[1] = 0.0f;
            }
        }
        const unsigned wbytes = (q < Q0 ? tu_wc_floats<NCT>() : tu_w1_floats<NCT>()) * 4u;
        const __amdgpu_buffer_rsrc_t rw = vxm_rsrc(wp + wchunk_base(q), wbytes);
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            wv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tid + T8_THREADS * it) * 16, 0, 0));
    };
    auto store_chunk = [&](int q) __attribute__((always_inline)) {
        float* Ws;
        int wcount;
        if (q < Q0) {
            float* const X0 = smem + tu_wc_floats<NCT>();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int vo, ld;
                low_slot(k, vo, ld);
                if (ld >= 0) X0[ld] = xi[0][k];
            }
            Ws = smem;
            wcount = tu_wc_floats<NCT>() / 4;
        } else {
            float* dst = smem + tu_w1_floats<NCT>() + wave * TU_PS1;      // seg-1 stage: [weights][X] (all operand offsets < 64 KB)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < 3 || lane < 48) {
                    *reinterpret_cast<f32x2*>(dst + 16 * j * TU_RS + s1_odd) = (f32x2){xi[j].x, xi[j].z};
                    *reinterpret_cast<f32x2*>(dst + 16 * j * TU_RS + s1_even) = (f32x2){xi[j].y, xi[j].w};
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (j < 1 || lane < 56) dst[32 * j * TU_RS + s1_halo] = xh[j];
            Ws = smem;
            wcount = tu_w1_floats<NCT>() / 4;
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + T8_THREADS * it;
            if (i < wcount) reinterpret_cast<f32x4*>(Ws)[i] = wv[it];
        }
    };

    // lane parts of the B-operand addresses
    const int pd = wave & 1;
    const int b1base = kq * TU_PS1 + (n & 7) + 2 * TU_RS * (n >> 3) + wave * HH * TU_RS;
    const int b0base = kq * TU_PSL + (n & 7) + 16 * (n >> 3) + ((wave >> 1) + pd) * 64;
    const int a0base = pd * 4 * 8 * 2 * NCT * 64 + lane;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int q = 0; q < Q; ++q) {
        if (q + 1 < Q) load_chunk(q + 1);             // in flight (registers) under the MFMAs of chunk q
        if (q < Q0) {
            // ---- collapsed taps: 8 j x 2 k-steps x 4 N-tiles (ph, pw)
            const float* Wc = smem;
            const float* X0 = smem + tu_wc_floats<NCT>();
            float a[2][NCT], bb[2];
            auto fetch = [&](int st, float (&af)[NCT], float& bf) __attribute__((always_inline)) {
                const int j = st >> 3, ks = (st >> 2) & 1, nt = st & 3, ph = nt >> 1, pw = nt & 1;
                const int jd = j >> 2, jh = (j >> 1) & 1, jw = j & 1;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) af[ct] = Wc[a0base + ((((ph * 2 + pw) * 8 + j) * 2 + ks) * NCT + ct) * 64];
                bf = X0[b0base + ks * 4 * TU_PSL + jd * 64 + (jh + ph) * 16 + jw + pw];
            };
            fetch(0, a[0], bb[0]);
#pragma unroll
            for (int st = 0; st < 64; ++st) {
                if (st + 1 < 64) fetch(st + 1, a[(st + 1) & 1], bb[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct][st & 3] = vxm_mfma16(a[st & 1][ct], bb[st & 1], acc[ct][st & 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // ---- regular 27 taps x 2 k-steps on the de-interleaved rows
            const float* Ws = smem;
            const float* X1 = smem + tu_w1_floats<NCT>();
            float a[2][NCT], bv[2][4];
            auto fetch = [&](int st, float (&af)[NCT], float (&bf)[4]) __attribute__((always_inline)) {
                const int t = st >> 1, s2 = st & 1;
                const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) af[ct] = Ws[((t * 2 + s2) * NCT + ct) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int ph = nt >> 1, pw = nt & 1;
                    const int e = (pw + kw) & 1, sh = (pw + kw) >> 1;
                    bf[nt] = X1[b1base + s2 * 4 * TU_PS1 + (kd * HH + ph + kh) * TU_RS + (e ? 12 + sh : 1 + sh)];
                }
            };
            fetch(0, a[0], bv[0]);
#pragma unroll
            for (int st = 0; st < 54; ++st) {
                if (st + 1 < 54) fetch(st + 1, a[(st + 1) & 1], bv[(st + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[ct][nt] = vxm_mfma16(a[st & 1][ct], bv[st & 1][nt], acc[ct][nt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (q + 1 < Q) {
            __syncthreads();                          // every wave is done reading chunk q
            store_chunk(q + 1);
            __syncthreads();
        }
    }

    // ---- epilogue: bias + LeakyReLU; lane n of N-tile (ph, pw) is voxel (h0 + ph + 2 (n >> 3), w0 + 2 (n & 7) + pw)
    const int d = d0 + wave;
    if (d < D) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = (g * NCT + ct) * 16 + kq * 4 + j;
                if (co >= Cout) continue;
                const float bz = bias ? bias[co] : 0.0f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int h = h0 + (nt >> 1) + 2 * (n >> 3), w = w0 + 2 * (n & 7) + (nt & 1);
                    float v = acc[ct][nt][j] + bz;
                    v = v > 0.0f ? v : v * act_slope;
                    if (h < H && w < W) y[(size_t)b * y_bs + (size_t)co * V + ((size_t)d * H + h) * W + w] = v;
                }
            }
    }
}

// packed weights of k_conv3d_k3_t8u: per output group g, Q0 collapsed chunks [par 8][j 8][ks 2][NCT][64] (segment-0 channel
// 8 q + 4 ks + (lane >> 4)), then Q1 regular chunks [27][2][NCT][64] (segment-1 channel 8 q + 4 s + (lane >> 4)); co = (g NCT + ct) 16 + (lane & 15).
// Per axis the collapsed tap j of parity p sums the kernel taps {p=0: j=0 -> {0}, j=1 -> {1,2};  p=1: j=0 -> {0,1}, j=1 -> {2}}.
__global__ void __launch_bounds__(256) k_pack_weights_up(const float* __restrict__ w, float* __restrict__ wp, int C0, int C1, int Cout, int NCT,
                                                         size_t elems) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    const int Cin = C0 + C1, Q0 = (C0 + TU_CK0 - 1) / TU_CK0, Q1 = (C1 + 7) / 8;
    const size_t wc = (size_t)8 * 8 * 2 * NCT * 64, w1 = (size_t)27 * 2 * NCT * 64, per_g = Q0 * wc + Q1 * w1;
    const int g = (int)(i / per_g);
    size_t r = i - (size_t)g * per_g;
    float v = 0.0f;
    if (r < Q0 * wc) {
        const int q = (int)(r / wc); r -= (size_t)q * wc;
        const int lane = r % 64; r /= 64;
        const int ct = r % NCT; r /= NCT;
        const int ks = r % 2; r /= 2;
        const int j = r % 8; const int par = (int)(r / 8);
        const int co = (g * NCT + ct) * 16 + (lane & 15), ci = q * TU_CK0 + 4 * ks + (lane >> 4);
        if (co < Cout && ci < C0) {
            const int p[3] = {(par >> 2) & 1, (par >> 1) & 1, par & 1}, jj[3] = {(j >> 2) & 1, (j >> 1) & 1, j & 1};
            int lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                if (p[a] == 0) { lo[a] = jj[a] ? 1 : 0; hi[a] = jj[a] ? 2 : 0; }
                else { lo[a] = jj[a] ? 2 : 0; hi[a] = jj[a] ? 2 : 1; }
            }
            const float* wk = w + ((size_t)co * Cin + ci) * 27;
            for (int kd = lo[0]; kd <= hi[0]; ++kd)
                for (int kh = lo[1]; kh <= hi[1]; ++kh)
                    for (int kw = lo[2]; kw <= hi[2]; ++kw) v += wk[(kd * 3 + kh) * 3 + kw];
        }
    } else {
        r -= Q0 * wc;
        const int q = (int)(r / w1); r -= (size_t)q * w1;
        const int lane = r % 64; r /= 64;
        const int ct = r % NCT; r /= NCT;
        const int s2 = r % 2; const int t = (int)(r / 2);
        const int co = (g * NCT + ct) * 16 + (lane & 15), c1 = q * 8 + 4 * s2 + (lane >> 4);
        if (co < Cout && c1 < C1) v = w[((size_t)co * Cin + C0 + c1) * 27 + t];
    }
    wp[i] = v;
}

// ------------------------------------------------------------------------------------------
// backward-data of the upsampled segment, straight to the LOW-resolution gradient
// ------------------------------------------------------------------------------------------
// d L / d x0[ci, m] = sum_co sum_{delta in {-1,0,1,2}^3} Wt[delta][ci,co] * dZ[co, 2 m + delta], with, per axis,
// Wt(-1) = w[2], Wt(0) = w[1] + w[2], Wt(1) = w[0] + w[1], Wt(2) = w[0]  (the adjoint of the collapsed forward above):
// a stride-2, 4x4x4-tap conv from the full-resolution dZ to the half-resolution input gradient.  It replaces the
// full-resolution backward-data of those channels (27 taps at 8x the voxels) PLUS upsample_nearest3d_backward (the
// 2x2x2 child sum is inside the taps), and the LeakyReLU' of the decoder block is applied in the epilogue.
// Implicit GEMM: M = 16 NCT input channels, N = 16 low-res voxels of a W row, K = 4 output channels of one tap.
// Block = 8 waves, low-res tile 2(D) x 4(H) x 16(W) (wave = one row); chunk = 4 output channels: dZ region
// [4][6][10][2 x 20] (rows de-interleaved by column parity, so the stride-2 columns of a tap are consecutive words)
// + the 64 taps' weights.
constexpr int DL_RS = 40, DL_ROWS = 6 * 10, DL_PS = DL_ROWS * DL_RS + 16;         // plane 2416 = 16 mod 32 (the two channel groups of a read hit disjoint banks)
template <int NCT> constexpr int dl_w_floats() { return 64 * NCT * 64; }
template <int NCT> constexpr int dl_lds_floats() { return dl_w_floats<NCT>() + 4 * DL_PS; }

template <int NCT>
__global__ void __launch_bounds__(T8_THREADS, 4) k_conv3d_k3_dlow(const float* __restrict__ dz, long long dz_bs, int Cout, const float* __restrict__ wp,
                                                                float* __restrict__ gx, long long gx_bs, int C0, const float* __restrict__ mask,
                                                                long long mask_bs, float mask_slope, int B, int D, int H, int W) {
    VXM_DYN_SMEM(float, smem);
    constexpr int WIT = dl_w_floats<NCT>() / 4 / T8_THREADS;           // 2 NCT float4 per thread
    float* const Ws = smem;                                           // [64 taps][NCT][64]
    float* const Zs = smem + dl_w_floats<NCT>();                      // [4][DL_PS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n = lane & 15;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1, Vl = Dl * Hl * Wl, V = D * H * W;
    const int nw = (Wl + 15) / 16, nh = (Hl + 3) / 4, nd = (Dl + 1) / 2;
    int t = blockIdx.x;
    const int tw = t % nw; t /= nw;
    const int th = t % nh; t /= nh;
    const int td = t % nd; const int b = t / nd;
    const int d0 = td * 2, h0 = th * 4, w0 = tw * 16;                 // low-res tile origin
    const int g = blockIdx.y;
    const int Q = (Cout + 3) / 4;
    const __amdgpu_buffer_rsrc_t rz = vxm_rsrc(dz + (size_t)b * dz_bs, (unsigned)Cout * (unsigned)V * 4u);

    // staging roles: dZ region rows r = (dz 0..5, hy 0..9) <-> full-res (2 d0 - 1 + dz, 2 h0 - 1 + hy), columns 2 w0 - 1 .. 2 w0 + 32.
    // interior slot = tid + 512 k (< 1920): channel c = slot / 480, row = (slot % 480) / 8, columns 4 q..4 q+3 (q = slot & 7)
    auto zslot = [&](int k, int& voff, int& lds_odd, int& lds_even) __attribute__((always_inline)) {
        const int slot = tid + T8_THREADS * k;
        const int c = slot / 480, r = (slot % 480) >> 3, q = slot & 7;
        const int gd = 2 * d0 - 1 + r / 10, gh = 2 * h0 - 1 + r % 10, gw = 2 * w0 + 4 * q;
        const bool ok = slot < 1920 && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw < W;
        voff = ok ? (c * V + (gd * H + gh) * W + gw) << 2 : VXM_OOB;
        // (col + 1) = 4 q + 1..4: odd ones -> odd block [20 + 2 q, +1], even ones -> even block (shifted one word) [2 q + 2, +1]
        lds_odd = slot < 1920 ? c * DL_PS + r * DL_RS + 20 + 2 * q : -1;
        lds_even = c * DL_PS + r * DL_RS + 2 * q + 2;
    };
    auto hslot = [&](int& voff, int& lds) __attribute__((always_inline)) {        // 480 halo elements: one per thread
        const int c = tid / 120, r = (tid % 120) >> 1, side = tid & 1;
        const int gd = 2 * d0 - 1 + r / 10, gh = 2 * h0 - 1 + r % 10, gw = side ? 2 * w0 + 32 : 2 * w0 - 1;
        const bool ok = tid < 480 && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        voff = ok ? (c * V + (gd * H + gh) * W + gw) << 2 : VXM_OOB;
        lds = tid < 480 ? c * DL_PS + r * DL_RS + (side ? 20 + 16 : 1) : -1;      // col 32 -> odd index 16; col -1 -> even index 0 (+1)
    };

    f32x4 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 xi[4];
    float xh;
    f32x4 wv[WIT];
    auto load_chunk = [&](int q) __attribute__((always_inline)) {
        const int soff = q * 4 * V * 4;          // channels beyond Cout in the last chunk fall outside num_records -> 0
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int vo, lo, le;
            zslot(k, vo, lo, le);
            xi[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, vo, soff, 0));
        }
        int vo, ld;
        hslot(vo, ld);
        xh = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rz, vo, soff, 0));
        const __amdgpu_buffer_rsrc_t rw = vxm_rsrc(wp + ((size_t)g * Q + q) * dl_w_floats<NCT>(), dl_w_floats<NCT>() * 4u);
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            wv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tid + T8_THREADS * it) * 16, 0, 0));
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int vo, lo, le;
            zslot(k, vo, lo, le);
            if (lo >= 0) {
                *reinterpret_cast<f32x2*>(Zs + lo) = (f32x2){xi[k].x, xi[k].z};
                *reinterpret_cast<f32x2*>(Zs + le) = (f32x2){xi[k].y, xi[k].w};
            }
        }
        int vo, ld;
        hslot(vo, ld);
        if (ld >= 0) Zs[ld] = xh;
#pragma unroll
        for (int it = 0; it < WIT; ++it) reinterpret_cast<f32x4*>(Ws)[tid + T8_THREADS * it] = wv[it];
    };

    // B operand: lane (kq = output channel of the chunk, n = low-res column): Zs[kq][row(2 dl + dd, 2 hl + dh)][parity block][n + shift]
    const int dl = wave >> 2, hl = wave & 3;
    const int bbase = kq * DL_PS + ((2 * dl) * 10 + 2 * hl) * DL_RS + n;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int q = 0; q < Q; ++q) {
        if (q + 1 < Q) load_chunk(q + 1);
        float a[2][NCT], bb[2];
        auto fetch = [&](int tap, float (&af)[NCT], float& bf) __attribute__((always_inline)) {
            const int ed = tap >> 4, eh = (tap >> 2) & 3, ew = tap & 3;       // delta + 1 in {0,1,2,3} per axis
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) af[ct] = Ws[(tap * NCT + ct) * 64 + lane];
            // column 2 n + (ew - 1) -> (col + 1) = 2 n + ew: parity ew & 1, index n + (ew >> 1) (+1 word shift in the even block)
            bf = Zs[bbase + (ed * 10 + eh) * DL_RS + ((ew & 1) ? 20 + (ew >> 1) : 1 + (ew >> 1))];
        };
        fetch(0, a[0], bb[0]);
#pragma unroll
        for (int tap = 0; tap < 64; ++tap) {
            if (tap + 1 < 64) fetch(tap + 1, a[(tap + 1) & 1], bb[(tap + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = vxm_mfma16(a[tap & 1][ct], bb[tap & 1], acc[ct]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q + 1 < Q) {
            __syncthreads();
            store_chunk();
            __syncthreads();
        }
    }
    // ---- epilogue: x LeakyReLU'(mask) (the decoder block's activation), NCDHW store at low resolution
    const int od = d0 + dl, oh = h0 + hl, ow = w0 + n;
    if (od < Dl && oh < Hl && ow < Wl) {
        const size_t vox = ((size_t)od * Hl + oh) * Wl + ow;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ci = (g * NCT + ct) * 16 + kq * 4 + j;
                if (ci >= C0) continue;
                float v = acc[ct][j];
                if (mask) v *= vxm_lrelu_grad(mask[(size_t)b * mask_bs + (size_t)ci * Vl + vox], mask_slope);
                gx[(size_t)b * gx_bs + (size_t)ci * Vl + vox] = v;
            }
    }
}

// packed weights of k_conv3d_k3_dlow: [g][q][tap 64][NCT][64], lane -> (ci = (g NCT + ct) 16 + (lane & 15), co = 4 q + (lane >> 4));
// tap = (ed, eh, ew), e = delta + 1; per axis the kernel taps summed: e=0 -> {2}, e=1 -> {1,2}, e=2 -> {0,1}, e=3 -> {0}.
__global__ void __launch_bounds__(256) k_pack_weights_dlow(const float* __restrict__ w, float* __restrict__ wp, int C0, int Cin, int Cout, int NCT,
                                                           size_t elems) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    const int Q = (Cout + 3) / 4;
    size_t r = i;
    const int lane = r % 64; r /= 64;
    const int ct = r % NCT; r /= NCT;
    const int tap = r % 64; r /= 64;
    const int q = r % Q; const int g = (int)(r / Q);
    const int ci = (g * NCT + ct) * 16 + (lane & 15), co = 4 * q + (lane >> 4);
    float v = 0.0f;
    if (ci < C0 && co < Cout) {
        const int e[3] = {tap >> 4, (tap >> 2) & 3, tap & 3};
        int lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = e[a] == 0 ? 2 : (e[a] == 1 ? 1 : 0);
            hi[a] = e[a] == 0 ? 2 : (e[a] == 1 ? 2 : (e[a] == 2 ? 1 : 0));
        }
        const float* wk = w + ((size_t)co * Cin + ci) * 27;
        for (int kd = lo[0]; kd <= hi[0]; ++kd)
            for (int kh = lo[1]; kh <= hi[1]; ++kh)
                for (int kw = lo[2]; kw <= hi[2]; ++kw) v += wk[(kd * 3 + kh) * 3 + kw];
    }
    wp[i] = v;
}

// ------------------------------------------------------------------------------------------
// forward conv with <= 4 output channels (the 16 -> 3 flow conv, networks.py:211,257)
// ------------------------------------------------------------------------------------------
// On the MFMA path 3 output channels occupy 3 of 16 rows.  Here the contraction runs on the vector ALUs instead:
// a thread owns 4 consecutive W voxels x CO output channels (4 CO accumulators), a block a 4 x 8 x 32 tile; per
// (ci, kd, kh) it reads its 6 input columns as one ds_read_b128 + one ds_read_b64, takes the 3 x CO weights from
// scalar loads (wave-uniform addresses in the reference layout: SGPR operands, no pack launch, no LDS traffic), and
// issues 12 CO FMAs.  Input chunks of 4 channels go through LDS ([4][6][10][36], halo columns at 0 / 33
// so that the 6-column reads are 16-byte aligned), zero padding from the buffer descriptor.
constexpr int FO_TD = 4, FO_TH = 8, FO_TW = 32, FO_CK = 4;
constexpr int FO_RS = 36, FO_ROWS = (FO_TD + 2) * (FO_TH + 2), FO_PS = FO_ROWS * FO_RS;   // 60 rows, plane 2160 floats

template <int CO>
__global__ void __launch_bounds__(256) k_conv3d_k3_fewout(const float* __restrict__ x, long long x_bs, int Cin, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, long long y_bs, float act_slope,
                                                          int D, int H, int W) {
    VXM_DYN_SMEM(float, smem);
    float* const Xs = smem;                                  // [FO_CK][FO_PS]
    const int tid = threadIdx.x, tx = tid & 7, ty = (tid >> 3) & 7, tz = tid >> 6;
    const int ntw = (W + FO_TW - 1) / FO_TW, nth = (H + FO_TH - 1) / FO_TH;
    int t = blockIdx.x;
    const int w0 = (t % ntw) * FO_TW; t /= ntw;
    const int h0 = (t % nth) * FO_TH;
    const int d0 = (t / nth) * FO_TD;
    const int b = blockIdx.y;
    const int V = D * H * W;
    const __amdgpu_buffer_rsrc_t rx = vxm_rsrc(x + (size_t)b * x_bs, (unsigned)Cin * (unsigned)V * 4u);
    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[co][j] = 0.0f;

    const int Q = (Cin + FO_CK - 1) / FO_CK;
    for (int q = 0; q < Q; ++q) {
        // ---- stage chunk q: 4 planes x 60 rows x (8 interior float4 + 2 halo columns)
        for (int slot = tid; slot < FO_CK * FO_ROWS * 8; slot += 256) {
            const int c = slot / (FO_ROWS * 8), rr = (slot / 8) % FO_ROWS, g4 = slot & 7;
            const int cg = q * FO_CK + c;
            const int gd = d0 - 1 + rr / (FO_TH + 2), gh = h0 - 1 + rr % (FO_TH + 2), gw = w0 + 4 * g4;
            const bool ok = cg < Cin && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw < W;
            const int vo = ok ? (cg * V + (gd * H + gh) * W + gw) << 2 : VXM_OOB;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0));
            float* dst = Xs + c * FO_PS + rr * FO_RS + 1 + 4 * g4;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        for (int slot = tid; slot < FO_CK * FO_ROWS * 2; slot += 256) {
            const int c = slot / (FO_ROWS * 2), rr = (slot >> 1) % FO_ROWS, side = slot & 1;
            const int cg = q * FO_CK + c;
            const int gd = d0 - 1 + rr / (FO_TH + 2), gh = h0 - 1 + rr % (FO_TH + 2), gw = side ? w0 + FO_TW : w0 - 1;
            const bool ok = cg < Cin && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const int vo = ok ? (cg * V + (gd * H + gh) * W + gw) << 2 : VXM_OOB;
            Xs[c * FO_PS + rr * FO_RS + (side ? FO_TW + 1 : 0)] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, vo, 0, 0));
        }
        __syncthreads();
        // ---- 4 channels x 9 (kd, kh) x [6 inputs, 3 x CO weights] -> 12 CO FMAs
#pragma unroll
        for (int c = 0; c < FO_CK; ++c) {
            const int ci = q * FO_CK + c;
            if (ci < Cin) {
#pragma unroll
                for (int kk = 0; kk < 9; ++kk) {
                    const int kd = kk / 3, kh = kk % 3;
                    const float* row = Xs + c * FO_PS + ((tz + kd) * (FO_TH + 2) + ty + kh) * FO_RS + 4 * tx;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(row);
                    const f32x2 e = *reinterpret_cast<const f32x2*>(row + 4);
                    const float in[6] = {a.x, a.y, a.z, a.w, e.x, e.y};
                    // the 3 x CO weights of (ci, kd, kh): wave-uniform addresses -> scalar loads, SGPR operands of the FMAs
                    float wc[CO][3];
#pragma unroll
                    for (int co = 0; co < CO; ++co)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) wc[co][kw] = w[((size_t)co * Cin + ci) * 27 + kk * 3 + kw];
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int co = 0; co < CO; ++co)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[co][j] = fmaf(in[j + kw], wc[co][kw], acc[co][j]);
                }
            }
        }
        __syncthreads();
    }
    const int d = d0 + tz, h = h0 + ty, wq = w0 + 4 * tx;
    if (d < D && h < H && wq < W) {                          // W % 4 == 0: the 4 voxels are in or out together
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float bz = bias ? bias[co] : 0.0f;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[co][j] + bz;
                o[j] = v > 0.0f ? v : v * act_slope;
            }
            *reinterpret_cast<f32x4*>(y + (size_t)b * y_bs + (size_t)co * V + ((size_t)d * H + h) * W + wq) = o;
        }
    }
}

// VXM_CONV_GENERIC=1 routes every conv launch through the generic kernels (any W / alignment; LDS-DMA backward-
// weight), so that the parity tests can exercise them on shapes the wide-load kernels would otherwise take.
bool bw_force_generic() {
    static const bool f = [] { const char* e = getenv("VXM_CONV_GENERIC"); return e && e[0] == '1'; }();
    return f;
}
// The 8-wave forward kernel is used from this many 8x4x16 tiles up (below, its 512-voxel tiles leave CUs idle);
// VXM_CONV_WIDE_MIN_TILES overrides the threshold so that the parity tests can run it on small volumes.
long long wide_min_tiles() {
    static const long long v = [] { const char* e = getenv("VXM_CONV_WIDE_MIN_TILES"); return e ? atoll(e) : 1024ll; }();
    return v;
}

struct ConvCfg { int CK, NCT, Q, G; size_t elems; };
static bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
ConvCfg conv_cfg(int Cin, int Cout) {
    ConvCfg c;
    c.CK = Cin <= 4 ? 4 : 8;
    // 16-channel output tiles per block: 3 when that wastes fewer MFMA rows than 2 (e.g. Cout = 48: 3 x 16, not 2 x 32)
    c.NCT = Cout <= 16 ? 1 : (Cout <= 32 ? 2 : ((Cout + 47) / 48 * 48 < (Cout + 31) / 32 * 32 ? 3 : 2));
    c.Q = (Cin + c.CK - 1) / c.CK;
    c.G = (Cout + 16 * c.NCT - 1) / (16 * c.NCT);
    c.elems = (size_t)c.G * c.Q * 27 * (c.CK / 4) * c.NCT * 64;
    return c;
}

bool fwd_wide_ok(const ConvCfg& c, const float* x0, int64_t bs0, const float* x1, int C1, int64_t bs1, const float* wpacked,
                 int B, int D, int H, int W) {
    const long long tiles8 = (long long)B * ((D + T8_TD - 1) / T8_TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return c.CK == 8 && c.NCT <= 2 && (W & 3) == 0 && al16(x0) && (C1 == 0 || al16(x1)) && (bs0 & 3) == 0 && (bs1 & 3) == 0 &&
           al16(wpacked) && tiles8 >= wide_min_tiles() && tiles8 < (1ll << 30) && !bw_force_generic();
}
bool bwd_weight_wide_ok(const float* x0, int64_t bs0, const float* x1, int C1, int64_t bs1, const float* dz, int64_t dz_bs, int W) {
    return (W & 3) == 0 && al16(x0) && (C1 == 0 || al16(x1)) && al16(dz) && (bs0 & 3) == 0 && (bs1 & 3) == 0 && (dz_bs & 3) == 0 &&
           !bw_force_generic();
}

// Few output channels (the 16 -> 3 flow conv): M = co would use 3 of 16 MFMA rows.  The product is computed with the
// roles swapped instead, gW[co,ci,tap] = sum_u X[ci,u] dZ[co,u - tap]: X becomes the (halo-free) A operand with M = ci,
// the zero-padded dZ the shifted B operand with N = (tap, co) = 81 entries -> 6 N-tiles instead of 27: the same kernel
// called with (x, dz) exchanged; the result comes out as [ci][co][26 - tap] and the reducer writes it back in place.
bool bwd_weight_swap_ok(int C0, int C1, int x0_up, int Cout, bool vec) {
    return vec && Cout <= 4 && C1 == 0 && !x0_up && C0 >= 8;
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
// backward-weight kernels
// ------------------------------------------------------------------------------------------
// gW[co,(ci,tap)] = sum_voxels dZ[co,v] * X[ci, v+tap]:  M = 16 output channels (A operand, dZ), N = 16
// (tap, ci) entries (B operand, shifted X), K = 4 voxels per v_mfma_f32_16x16x4_f32.  A block owns one
// 16-input-channel chunk x one 16*NCT output-channel group, keeps its partial gW in registers while it walks
// voxel tiles (4x4x16), and writes it once; k_reduce_partials sums the per-block partials in a fixed order.
// Tiles are zero padded through the buffer descriptor: a lane whose offset is beyond num_records loads 0.0
// (and an LDS-DMA lane writes 0.0 -- probed on gfx950, tools/probe/ldsdma_probe.hip).
constexpr int BW_WAVES = 16;      // one 1024-thread block per CU: 4 waves per SIMD share the MFMA pipe
constexpr int BW_THREADS = 64 * BW_WAVES;
constexpr int BW_SLOTS = 2;       // N-tiles per wave: 27 taps over 16 waves = 11 x 2 + 5 x 1 -> 7,7,7,6 per SIMD
constexpr int BW_CKI = 16;        // input channels per chunk (wave w stages channel w)
constexpr int BW_PZ = 260;        // dZ plane (256 voxels) stride of the LDS-DMA kernel: 16-byte aligned planes for the dwordx4
                                  // form (A-operand reads of channels n and n+8 share a bank: 2-way on the A reads only)

// The k-steps S0..S1-1 (4 voxels each) of one 4x4x16 tile for a wave that owns S N-tiles: per step NCT
// A-fragments (dZ) + S B-fragments (shifted X) from LDS feed S x NCT MFMAs.  Fully unrolled (every LDS offset an
// immediate), branch-free, operands of step s+1 requested before the MFMAs of step s (register double buffer).
// X plane layout [6][6][RS]; voxels 4s..4s+3: row = s>>2 -> (dz, hy) = (row>>2, row&3), wx = 4 (s&3) + kq (kq in boff).
// BIAS: the wave also sums its dZ fragments (VALU adds beside the MFMAs): lane (co = n, kq) collects the voxels
// 4 s + kq of output channel co -> the bias gradient sum_v dZ[co, v] without another pass over dZ.
template <int NCT, int S, int S0, int S1, int RS, int PZ, bool BIAS>
__device__ __forceinline__ void bw_ksteps(const float* __restrict__ Xb, const float* __restrict__ Zb, const int (&boff)[BW_SLOTS], int aoff,
                                          f32x4 (&acc)[BW_SLOTS][NCT], float (&bsum)[NCT]) {
    float a[2][NCT], bv[2][S];
    auto fetch = [&](int s, float (&af)[NCT], float (&bf)[S]) __attribute__((always_inline)) {
        const int row = s >> 2;
        const int xbase = ((row >> 2) * HH + (row & 3)) * RS + 4 * (s & 3);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) af[ct] = Zb[aoff + ct * 16 * PZ + 4 * s];
#pragma unroll
        for (int i = 0; i < S; ++i) bf[i] = Xb[boff[i] + xbase];
    };
    fetch(S0, a[S0 & 1], bv[S0 & 1]);
#pragma unroll
    for (int s = S0; s < S1; ++s) {
        if (s + 1 < S1) fetch(s + 1, a[(s + 1) & 1], bv[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch above the MFMAs
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = vxm_mfma16(a[s & 1][ct], bv[s & 1][i], acc[i][ct]);
        if (BIAS) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) bsum[ct] += a[s & 1][ct];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// uniform dispatch on the wave's N-tile count OUTSIDE the k-loop (a per-slot test inside it splits every MFMA
// group into its own basic block and serialises ds_read -> wait -> MFMA)
template <int NCT, int S0, int S1, int RS, int PZ>
__device__ __forceinline__ void bw_ksteps_n(int nslots, bool bias, const float* Xb, const float* Zb, const int (&boff)[BW_SLOTS], int aoff,
                                            f32x4 (&acc)[BW_SLOTS][NCT], float (&bsum)[NCT]) {
    if (bias) {             // wave 0 of the chunk-0 blocks (it always owns at least one N-tile)
        if (nslots == 2) bw_ksteps<NCT, 2, S0, S1, RS, PZ, true>(Xb, Zb, boff, aoff, acc, bsum);
        else bw_ksteps<NCT, 1, S0, S1, RS, PZ, true>(Xb, Zb, boff, aoff, acc, bsum);
        return;
    }
    switch (nslots) {
        case 2: bw_ksteps<NCT, 2, S0, S1, RS, PZ, false>(Xb, Zb, boff, aoff, acc, bsum); break;
        case 1: bw_ksteps<NCT, 1, S0, S1, RS, PZ, false>(Xb, Zb, boff, aoff, acc, bsum); break;
        default: break;
    }
}

// 1-D grid of T blocks (one per CU): block b -> combo = b % (Qc G) (input-channel chunk x output-channel group), the
// idx = b / (Qc G)-th of the cnt blocks of that combo, which walks the idx-th of cnt CONTIGUOUS ranges of the tile
// list (consecutive tiles of a block share halo lines through its own L1/L2; cnt differs by at most one between
// combos, so 256 CUs stay busy when Qc G does not divide 256).
struct BwBlock { int idx, c0, ckc, nent, ntile, cog, lo, hi; };
__device__ __forceinline__ BwBlock bw_block(int Cin, int NCT, int B, int D, int H, int W, int Qc, int G) {
    BwBlock k;
    const int cb = Qc * G, T = gridDim.x;
    const int combo = blockIdx.x % cb;
    k.idx = blockIdx.x / cb;
    const int cnt = (T - combo + cb - 1) / cb;
    k.c0 = (combo % Qc) * BW_CKI;
    k.ckc = min(BW_CKI, Cin - k.c0);
    k.nent = 27 * k.ckc;
    k.ntile = (k.nent + 15) / 16;
    k.cog = (combo / Qc) * 16 * NCT;
    const int ntiles = B * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    k.lo = (int)((long long)ntiles * k.idx / cnt);
    k.hi = (int)((long long)ntiles * (k.idx + 1) / cnt);
    return k;
}
// partial gW of this block: part[idx][co][ci][tap] (the combos of one idx tile the array)
// slot layout: [Cout][Cin][27] weight-gradient partial followed by [Cout] bias-gradient partial
template <int NCT>
__device__ __forceinline__ void bw_write_partial(const BwBlock& k, float* __restrict__ part, int Cout, int Cin, int wave, int lane, int nslots,
                                                 const f32x4 (&acc)[BW_SLOTS][NCT], bool bias, float (&bsum)[NCT]) {
    const int kq = lane >> 4, n = lane & 15;
    float* out = part + (size_t)k.idx * ((size_t)Cout * Cin * 27 + Cout);
    if (bias) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float t = bsum[ct];
            t += __shfl_xor(t, 16, 64);
            t += __shfl_xor(t, 32, 64);
            const int co = k.cog + ct * 16 + n;
            if (kq == 0 && co < Cout) out[(size_t)Cout * Cin * 27 + co] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < BW_SLOTS; ++i) {
        if (i >= nslots) continue;
        const int e = (wave + BW_WAVES * i) * 16 + n;
        if (e >= k.nent) continue;
        const int t = e / k.ckc, ci = k.c0 + (e - t * k.ckc);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = k.cog + ct * 16 + kq * 4 + j;
                if (co < Cout) out[((size_t)co * Cin + ci) * 27 + t] = acc[i][ct][j];
            }
    }
}

// ---- fast path (W % 4 == 0, 16-byte aligned tensors): register-staged tiles from wide buffer loads ------------
// X plane in LDS: [6][6][20] with the 16 interior columns at 2..17 (8-byte aligned -> ds_write_b64), the halo
// columns at 1 and 18; plane stride 738 = 2 mod 32 (conflict-free B-operand reads).  Per tile a wave issues 3
// dwordx4 (interior rows of its channel; dwordx2 + duplicate for the x2-upsampled segment), 2 dword (halo
// columns) and NCT dwordx4 (dZ) buffer loads up front, runs the 64 k-steps of the CURRENT tile out of LDS while
// they are in flight, then writes them into the OTHER LDS tile buffer; one barrier per tile.  2 x 80.5 KB of LDS.
constexpr int BV_RS = 20;
constexpr int BV_PSX = 738;
constexpr int BV_PZ = 258;        // dZ plane stride = 2 mod 32: conflict-free A-operand reads (planes 8-byte aligned: ds_write_b64)
template <int NCT> constexpr int bv_lds_floats() { return BW_CKI * BV_PSX + 16 * NCT * BV_PZ; }

template <int NCT>
__global__ void __launch_bounds__(BW_THREADS) k_conv3d_k3_bwd_weight_vec(ConvIn in, const float* __restrict__ dz, long long dz_bs, int Cout, int want_bias,
                                                                           float* __restrict__ part, int B, int D, int H, int W,
                                                                           int Qc, int G) {
    VXM_DYN_SMEM(float, smem);
    constexpr int BUF = bv_lds_floats<NCT>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int kq = lane >> 4, n = lane & 15;
    const float* const ix0 = in.x0; const float* const ix1 = in.x1;
    const long long ibs0 = in.bs0, ibs1 = in.bs1;
    const int iC0 = in.C0, iC1 = in.C1, iup0 = in.up0;
    const int Cin = iC0 + iC1;
    const BwBlock k = bw_block(Cin, NCT, B, D, H, W, Qc, G);
    const int HWp = H * W, V = D * HWp;
    const int Hs = H >> 1, Ws = W >> 1;
    const int V0 = iup0 ? (D >> 1) * Hs * Ws : V;     // plane size of segment 0

    // N-tile j of this wave (slot i): entries e = j*16 + n  ->  (tap = e / ckc, channel = e % ckc)
    int boff[BW_SLOTS];
    const int nslots = wave < k.ntile ? (k.ntile - wave + BW_WAVES - 1) / BW_WAVES : 0;     // wave-uniform
#pragma unroll
    for (int i = 0; i < BW_SLOTS; ++i) {
        const int e = (wave + BW_WAVES * i) * 16 + n;
        int off = 2;
        if (e < k.nent) {
            const int t = e / k.ckc, cl = e - t * k.ckc;
            off = cl * BV_PSX + ((t / 9) * HH + (t / 3) % 3) * BV_RS + t % 3 + 1;       // column = wx + kw - 1 + 2
        }
        boff[i] = off + kq;                       // + voxel k of the MFMA B operand
    }
    const int aoff = n * BV_PZ + kq;              // MFMA A operand: dZ[co = n][voxel 4s + kq]
    f32x4 acc[BW_SLOTS][NCT];
#pragma unroll
    for (int i = 0; i < BW_SLOTS; ++i)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool bias = want_bias && k.c0 == 0 && wave == 0;       // wave-uniform: one wave per output-channel group
    float bsum[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bsum[ct] = 0.0f;

    // staging roles of this lane (tile independent).  Interior: slot 64 j + lane (< 144) of a plane -> row
    // 16 j + (lane >> 2), columns 4 (lane & 3)..+3;  halo: slot 64 j + lane (< 72) -> row 32 j + (lane >> 1), side
    // lane & 1;  dZ: lane -> (row = lane >> 2, columns 4 (lane & 3)..+3).  LDS offsets = lane base + immediate.
    const int lq = lane & 3, lr4 = lane >> 2, lr2 = lane >> 1, hside = lane & 1;
    const int ibase = lr4 * BV_RS + 2 + 4 * lq;          // + 16 j rows
    const int hbase = lr2 * BV_RS + (hside ? 18 : 1);    // + 32 j rows
    auto ivalid = [&](int j) __attribute__((always_inline)) { return j < 2 || lane < 16; };
    auto hvalid = [&](int j) __attribute__((always_inline)) { return j < 1 || lane < 8; };

    f32x4 xi[3];             // interior pieces of this wave's channel
    float xh[2];             // halo pieces
    f32x4 zv[NCT];           // dZ planes

    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        int sb, sd0, sh0, sw0;
        tile_origin(tile, D, H, W, sb, sd0, sh0, sw0);
        const __amdgpu_buffer_rsrc_t r0 = vxm_rsrc(ix0 + (size_t)sb * ibs0, (unsigned)iC0 * (unsigned)V0 * 4u);
        const __amdgpu_buffer_rsrc_t r1 = vxm_rsrc(iC1 ? ix1 + (size_t)sb * ibs1 : ix0, (unsigned)iC1 * (unsigned)V * 4u);
        const __amdgpu_buffer_rsrc_t rz = vxm_rsrc(dz + (size_t)sb * dz_bs, (unsigned)Cout * (unsigned)V * 4u);
        // per-lane byte offsets inside a plane (full-res source, and the x2-upsampled source of segment 0)
        int vi[3], viu[3], vh[2], vhu[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int rr = 16 * j + lr4;
            const int gd = sd0 - 1 + rr / HH, gh = sh0 - 1 + rr % HH, gw = sw0 + 4 * lq;
            const bool ok = ivalid(j) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw < W;
            vi[j] = ok ? ((gd * H + gh) * W + gw) << 2 : VXM_OOB;
            viu[j] = ok ? (((gd >> 1) * Hs + (gh >> 1)) * Ws + (gw >> 1)) << 2 : VXM_OOB;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rr = 32 * j + lr2;
            const int gd = sd0 - 1 + rr / HH, gh = sh0 - 1 + rr % HH, gw = hside ? sw0 + TW : sw0 - 1;
            const bool ok = hvalid(j) && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            vh[j] = ok ? ((gd * H + gh) * W + gw) << 2 : VXM_OOB;
            vhu[j] = ok ? (((gd >> 1) * Hs + (gh >> 1)) * Ws + (gw >> 1)) << 2 : VXM_OOB;
        }
        {                                                       // this wave stages channel `wave` of the chunk
            const int cl = wave, cg = k.c0 + cl;
            if (cl < k.ckc) {                                   // wave-uniform
                if (cg < iC0 && iup0) {                         // x2 nearest upsampling: 2 source floats -> 4 columns
                    const int soff = cg * V0 * 4;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r0, viu[j], soff, 0));
                        xi[j] = (f32x4){t.x, t.x, t.y, t.y};
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) xh[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, vhu[j], soff, 0));
                } else {
                    const bool s0 = cg < iC0;
                    const __amdgpu_buffer_rsrc_t r = s0 ? r0 : r1;
                    const int soff = (s0 ? cg * V0 : (cg - iC0) * V) * 4;
#pragma unroll
                    for (int j = 0; j < 3; ++j) xi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vi[j], soff, 0));
#pragma unroll
                    for (int j = 0; j < 2; ++j) xh[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vh[j], soff, 0));
                }
            }
        }
        // dZ: planes co = NCT wave + i; lane -> (row = (dd, hy), 4 floats at wx = 4 lq)
        const int zd = sd0 + (lr4 >> 2), zh = sh0 + (lr4 & 3), zw = sw0 + 4 * lq;
        const int zvo = (zd < D && zh < H && zw < W) ? ((zd * H + zh) * W + zw) << 2 : VXM_OOB;
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
            const int co = NCT * wave + i;
            const bool uok = k.cog + co < Cout;                 // wave-uniform
            zv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, uok ? zvo : VXM_OOB, uok ? (k.cog + co) * V * 4 : 0, 0));
        }
    };
    auto store_tile = [&](float* Xn, float* Zn) __attribute__((always_inline)) {
        if (wave < k.ckc) {
            float* dst = Xn + wave * BV_PSX;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (ivalid(j)) {
                    *reinterpret_cast<f32x2*>(dst + ibase + 16 * j * BV_RS) = (f32x2){xi[j].x, xi[j].y};
                    *reinterpret_cast<f32x2*>(dst + ibase + 16 * j * BV_RS + 2) = (f32x2){xi[j].z, xi[j].w};
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (hvalid(j)) dst[hbase + 32 * j * BV_RS] = xh[j];
        }
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
            float* zp = Zn + (NCT * wave + i) * BV_PZ + 4 * lane;
            *reinterpret_cast<f32x2*>(zp) = (f32x2){zv[i].x, zv[i].y};
            *reinterpret_cast<f32x2*>(zp + 2) = (f32x2){zv[i].z, zv[i].w};
        }
    };

    int tile = k.lo;
    if (tile < k.hi) {
        load_tile(tile);
        store_tile(smem, smem + BW_CKI * BV_PSX);
    }
    __syncthreads();
    for (int iter = 0; tile < k.hi; ++tile, ++iter) {
        const bool more = tile + 1 < k.hi;
        if (more) load_tile(tile + 1);                 // in flight under the MFMAs below
        const float* Xb = smem + (iter & 1) * BUF;
        bw_ksteps_n<NCT, 0, 64, BV_RS, BV_PZ>(nslots, bias, Xb, Xb + BW_CKI * BV_PSX, boff, aoff, acc, bsum);
        float* Xn = smem + ((iter + 1) & 1) * BUF;     // last read before the previous barrier
        if (more) store_tile(Xn, Xn + BW_CKI * BV_PSX);
        __syncthreads();
    }
    bw_write_partial<NCT>(k, part, Cout, Cin, wave, lane, nslots, acc, bias, bsum);
}

// ---- generic path (any W / alignment): LDS-DMA staging -------------------------------------------------------
// buffer_load_dword ... lds: a wave-instruction writes 64 consecutive LDS dwords from 64 arbitrary global
// addresses, so a haloed X plane [6][6][18] (648 floats, lane-linear) is 11 wave-instructions whose per-lane
// offsets depend on the TILE only, and a dZ plane is 4 (1 dwordx4 when W % 4 == 0).  No staging VGPRs, but each
// LDS-DMA instruction costs the CU ~200 cycles (measured), which is why the wide-load path above is the default.
// One 16-wave block per CU (2 x 78 KB LDS tile buffers), loads of tile t+1 issued before the k-steps of tile t.
constexpr int BW_XJ = 11;         // wave-loads per haloed X plane (648 floats)
constexpr int BW_PSX = 706;       // X plane stride: >= 64*BW_XJ and = 2 mod 32 (conflict-free B-operand reads)
template <int NCT> constexpr int bw_buf_floats() { return BW_CKI * BW_PSX + 16 * NCT * BW_PZ; }

template <int NCT>
__global__ void __launch_bounds__(BW_THREADS) k_conv3d_k3_bwd_weight_dma(ConvIn in, const float* __restrict__ dz, long long dz_bs, int Cout, int want_bias,
                                                                        float* __restrict__ part, int B, int D, int H, int W,
                                                                        int Qc, int G) {
    VXM_DYN_SMEM(float, smem);
    constexpr int BUF = bw_buf_floats<NCT>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int kq = lane >> 4, n = lane & 15;
    const float* const ix0 = in.x0; const float* const ix1 = in.x1;
    const long long ibs0 = in.bs0, ibs1 = in.bs1;
    const int iC0 = in.C0, iC1 = in.C1, iup0 = in.up0;
    const int Cin = iC0 + iC1;
    const BwBlock k = bw_block(Cin, NCT, B, D, H, W, Qc, G);
    const int HWp = H * W, V = D * HWp;
    const int Hs = H >> 1, Ws = W >> 1;
    const int V0 = iup0 ? (D >> 1) * Hs * Ws : V;     // plane size of segment 0

    int boff[BW_SLOTS];
    const int nslots = wave < k.ntile ? (k.ntile - wave + BW_WAVES - 1) / BW_WAVES : 0;     // wave-uniform
#pragma unroll
    for (int i = 0; i < BW_SLOTS; ++i) {
        const int e = (wave + BW_WAVES * i) * 16 + n;
        int off = 0;
        if (e < k.nent) {
            const int t = e / k.ckc, cl = e - t * k.ckc;
            off = cl * BW_PSX + ((t / 9) * HH + (t / 3) % 3) * HW + t % 3;
        }
        boff[i] = off + kq;
    }
    const int aoff = n * BW_PZ + kq;
    f32x4 acc[BW_SLOTS][NCT];
#pragma unroll
    for (int i = 0; i < BW_SLOTS; ++i)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool bias = want_bias && k.c0 == 0 && wave == 0;       // wave-uniform: one wave per output-channel group
    float bsum[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bsum[ct] = 0.0f;

    // lane constants of the staging pattern: element e = 64 j + lane of the haloed plane -> (dz, hy, wx)
    int pk[BW_XJ];
#pragma unroll
    for (int j = 0; j < BW_XJ; ++j) {
        const int e = 64 * j + lane;
        const int pdz = e / (HH * HW), r = e - pdz * (HH * HW), phy = r / HW, pwx = r - phy * HW;
        pk[j] = e < HVOX ? (pdz | (phy << 8) | (pwx << 16)) : -1;
    }
    const int zr = lane >> 4, zx = lane & 15;     // dword dZ slab [4 rows][16]: one wave-load per (co, depth)

    auto stage = [&](int tile, float* Xn, float* Zn) __attribute__((always_inline)) {
        int sb, sd0, sh0, sw0;
        tile_origin(tile, D, H, W, sb, sd0, sh0, sw0);
        int vo[BW_XJ], vu[BW_XJ];
#pragma unroll
        for (int j = 0; j < BW_XJ; ++j) {
            const int gd = sd0 - 1 + (pk[j] & 0xff), gh = sh0 - 1 + ((pk[j] >> 8) & 0xff), gw = sw0 - 1 + ((pk[j] >> 16) & 0xff);
            const bool ok = pk[j] >= 0 && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            vo[j] = ok ? ((gd * H + gh) * W + gw) << 2 : VXM_OOB;
            vu[j] = ok ? (((gd >> 1) * Hs + (gh >> 1)) * Ws + (gw >> 1)) << 2 : VXM_OOB;
        }
        const __amdgpu_buffer_rsrc_t r0 = vxm_rsrc(ix0 + (size_t)sb * ibs0, (unsigned)iC0 * (unsigned)V0 * 4u);
        const __amdgpu_buffer_rsrc_t r1 = vxm_rsrc(iC1 ? ix1 + (size_t)sb * ibs1 : ix0, (unsigned)iC1 * (unsigned)V * 4u);
        {                                                       // this wave stages channel `wave` of the chunk
            const int cl = wave, cg = k.c0 + cl;
            if (cl < k.ckc) {                                   // wave-uniform
                float* dst = Xn + cl * BW_PSX;
                if (cg < iC0) {
                    const int soff = cg * V0 * 4;
                    if (iup0) {
#pragma unroll
                        for (int j = 0; j < BW_XJ; ++j) vxm_lds_dma4(r0, dst + 64 * j, vu[j], soff);
                    } else {
#pragma unroll
                        for (int j = 0; j < BW_XJ; ++j) vxm_lds_dma4(r0, dst + 64 * j, vo[j], soff);
                    }
                } else {
                    const int soff = (cg - iC0) * V * 4;
#pragma unroll
                    for (int j = 0; j < BW_XJ; ++j) vxm_lds_dma4(r1, dst + 64 * j, vo[j], soff);
                }
            }
        }
        // dZ: planes co = NCT wave + i
        const __amdgpu_buffer_rsrc_t rz = vxm_rsrc(dz + (size_t)sb * dz_bs, (unsigned)Cout * (unsigned)V * 4u);
        if ((W & 3) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0 && (dz_bs & 3) == 0) {
            // one dwordx4 wave-load per plane: lane -> (row = lane >> 2 -> (dd, hy) = (row >> 2, row & 3), wx = 4 (lane & 3))
            const int zrow = lane >> 2, zd = sd0 + (zrow >> 2), zh = sh0 + (zrow & 3), zw = sw0 + 4 * (lane & 3);
            const int zvo = (zd < D && zh < H && zw < W) ? ((zd * H + zh) * W + zw) << 2 : VXM_OOB;
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                const int co = NCT * wave + i;
                const bool uok = k.cog + co < Cout;                           // wave-uniform
                vxm_lds_dma16(rz, Zn + co * BW_PZ, uok ? zvo : VXM_OOB, uok ? (k.cog + co) * V * 4 : 0);
            }
        } else {
            const int zh = sh0 + zr, zw = sw0 + zx;
            const int zvo = (zh < H && zw < W) ? (zh * W + zw) << 2 : VXM_OOB;
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                const int co = NCT * wave + i;
#pragma unroll
                for (int dd = 0; dd < TD; ++dd) {
                    const bool uok = k.cog + co < Cout && sd0 + dd < D;       // wave-uniform
                    const int soff = uok ? ((k.cog + co) * D + sd0 + dd) * HWp * 4 : 0;
                    vxm_lds_dma4(rz, Zn + co * BW_PZ + 64 * dd, uok ? zvo : VXM_OOB, soff);
                }
            }
        }
    };

    int tile = k.lo;
    if (tile < k.hi) stage(tile, smem, smem + BW_CKI * BW_PSX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int iter = 0; tile < k.hi; ++tile, ++iter) {
        float* Xn = smem + ((iter + 1) & 1) * BUF;
        const float* Xb = smem + (iter & 1) * BUF;
        const float* Zb = Xb + BW_CKI * BW_PSX;
        if (tile + 1 < k.hi) stage(tile + 1, Xn, Xn + BW_CKI * BW_PSX);      // in flight under the MFMAs below
        bw_ksteps_n<NCT, 0, 64, HW, BW_PZ>(nslots, bias, Xb, Zb, boff, aoff, acc, bsum);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    bw_write_partial<NCT>(k, part, Cout, Cin, wave, lane, nslots, acc, bias, bsum);
}

// ---- backward-weight of the upsampled segment in collapsed form ----------------------------------------------------
// gWc[p][j][co][ci] = sum_{o of parity p} dZ[co,o] x0[ci, (o >> 1) + j - 1 + p]  (p, j in {0,1}^3: 8 x 8 low-resolution taps
// instead of 27 full-resolution ones: 8 instead of 27 MACs per channel pair and voxel), and afterwards
// gW[co,ci,(kd,kh,kw)] = sum_p gWc[p][(j_pd(kd), j_ph(kh), j_pw(kw))]  with  j_0(k) = (k >= 1), j_1(k) = (k == 2).
// Same block plan as k_conv3d_k3_bwd_weight_vec (16 waves, contiguous tile ranges, double-buffered LDS tile, one barrier
// per tile).  Wave w owns parity p = w >> 1 and the taps j = 4 (w & 1) .. +3: its k-steps are the 8 groups of 4 voxels
// of parity p in the 4x4x16 tile (d = pd + 2 dd, h = ph + 2 hh, w = pw + 2 (4 wh + k)), so all 16 waves work on disjoint
// voxels of the same staged tile.  dZ rows are stored de-interleaved by column parity ([8 even | 8 odd]), x0 as its
// [4][4][10] low-resolution neighbourhood.
constexpr int BU_PZ = 258;                    // dZ plane [4][4][16] -> 2 mod 32
constexpr int BU_PSL = 4 * 4 * 12 + 2;        // x0 plane [4][4][12] -> 194 = 2 mod 32
template <int NCT> constexpr int bu_buf_floats() { return 16 * NCT * BU_PZ + BW_CKI * BU_PSL; }

template <int NCT>
__global__ void __launch_bounds__(BW_THREADS) k_conv3d_k3_bwd_weight_up(const float* __restrict__ x0, long long bs0, int C0, const float* __restrict__ dz,
                                                                       long long dz_bs, int Cout, float* __restrict__ part, int B, int D, int H,
                                                                       int W, int Qc, int G) {
    VXM_DYN_SMEM(float, smem);
    constexpr int BUF = bu_buf_floats<NCT>();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n = lane & 15;
    const BwBlock k = bw_block(C0, NCT, B, D, H, W, Qc, G);
    const int V = D * H * W;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1, Vl = Dl * Hl * Wl;
    const int par = wave >> 1, pd = par >> 2, ph = (par >> 1) & 1, pw = par & 1, jbase = 4 * (wave & 1);

    f32x4 acc[4][NCT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: wave w loads dZ planes NCT w + i (one dwordx4 per lane: row = lane >> 2, columns 4 (lane & 3)..+3) and the
    // low-resolution plane of channel w (160 elements: 3 dwords per lane)
    const int lq = lane & 3, lr4 = lane >> 2;
    f32x4 zv[NCT];
    float xl[3];
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        int sb, sd0, sh0, sw0;
        tile_origin(tile, D, H, W, sb, sd0, sh0, sw0);
        const __amdgpu_buffer_rsrc_t rz = vxm_rsrc(dz + (size_t)sb * dz_bs, (unsigned)Cout * (unsigned)V * 4u);
        const __amdgpu_buffer_rsrc_t rx = vxm_rsrc(x0 + (size_t)sb * bs0, (unsigned)C0 * (unsigned)Vl * 4u);
        const int zd = sd0 + (lr4 >> 2), zh = sh0 + (lr4 & 3), zw = sw0 + 4 * lq;
        const int zvo = (zd < D && zh < H && zw < W) ? ((zd * H + zh) * W + zw) << 2 : VXM_OOB;
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
            const int co = NCT * wave + i;
            const bool uok = k.cog + co < Cout;
            zv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, uok ? zvo : VXM_OOB, uok ? (k.cog + co) * V * 4 : 0, 0));
        }
        const int cg = k.c0 + wave;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int e = 64 * t + lane, dl = e / 40, hl = (e % 40) / 10, wl = e % 10;
            const int gd = (sd0 >> 1) - 1 + dl, gh = (sh0 >> 1) - 1 + hl, gw = (sw0 >> 1) - 1 + wl;
            const bool ok = e < 160 && wave < k.ckc && (unsigned)gd < (unsigned)Dl && (unsigned)gh < (unsigned)Hl && (unsigned)gw < (unsigned)Wl;
            xl[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, ok ? ((gd * Hl + gh) * Wl + gw) << 2 : VXM_OOB, ok ? cg * Vl * 4 : 0, 0));
        }
    };
    auto store_tile = [&](float* Zn, float* Xn) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NCT; ++i) {            // row [8 even | 8 odd]: columns 4q, 4q+2 -> even[2q, 2q+1]; 4q+1, 4q+3 -> odd[2q, 2q+1]
            float* zp = Zn + (NCT * wave + i) * BU_PZ + lr4 * 16 + 2 * lq;
            *reinterpret_cast<f32x2*>(zp) = (f32x2){zv[i].x, zv[i].z};
            *reinterpret_cast<f32x2*>(zp + 8) = (f32x2){zv[i].y, zv[i].w};
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int e = 64 * t + lane, dl = e / 40, hl = (e % 40) / 10, wl = e % 10;
            if (e < 160) Xn[wave * BU_PSL + (dl * 4 + hl) * 12 + wl] = xl[t];
        }
    };

    // operand addresses: A = dZ[co = n (+16 ct)][d = pd + 2 dd][h = ph + 2 hh][parity block pw][4 wh + kq]
    //                    B = x0[ci = n][dl = dd + jd + pd][hl = hh + jh + ph][wl = 4 wh + kq + jw + pw]
    const int abase = n * BU_PZ + (pd * 4 + ph) * 16 + pw * 8 + kq;
    const int bbase = n * BU_PSL + (pd * 4 + ph) * 12 + pw + kq;
    int tile = k.lo;
    if (tile < k.hi) {
        load_tile(tile);
        store_tile(smem, smem + 16 * NCT * BU_PZ);
    }
    __syncthreads();
    for (int iter = 0; tile < k.hi; ++tile, ++iter) {
        const bool more = tile + 1 < k.hi;
        if (more) load_tile(tile + 1);
        const float* Zb = smem + (iter & 1) * BUF;
        const float* Xb = Zb + 16 * NCT * BU_PZ;
        float a[2][NCT], bv[2][4];
        auto fetch = [&](int s, float (&af)[NCT], float (&bf)[4]) __attribute__((always_inline)) {
            const int dd = s >> 2, hh = (s >> 1) & 1, wh = s & 1;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) af[ct] = Zb[abase + ct * 16 * BU_PZ + (dd * 8 + hh * 2) * 16 + 4 * wh];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // j = jbase + i (jbase in {0,4}: jd = wave & 1 is runtime-uniform, (jh, jw) = (i >> 1, i & 1) compile-time)
                bf[i] = Xb[bbase + ((dd + (jbase >> 2)) * 4 + hh + (i >> 1)) * 12 + 4 * wh + (i & 1)];
            }
        };
        fetch(0, a[0], bv[0]);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) fetch(s + 1, a[(s + 1) & 1], bv[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = vxm_mfma16(a[s & 1][ct], bv[s & 1][i], acc[i][ct]);
            __builtin_amdgcn_sched_barrier(0);
        }
        float* Zn = smem + ((iter + 1) & 1) * BUF;
        if (more) store_tile(Zn, Zn + 16 * NCT * BU_PZ);
        __syncthreads();
    }
    // partial: part[idx][co][ci (C0)][p*8 + j]
    float* out = part + (size_t)k.idx * ((size_t)Cout * C0 * 64);
    const int ci = k.c0 + n;
    if (n < k.ckc) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = k.cog + ct * 16 + kq * 4 + j;
                    if (co < Cout) out[((size_t)co * C0 + ci) * 64 + par * 8 + jbase + i] = acc[i][ct][j];
                }
    }
}

// Reduction of the collapsed partials in two coalesced steps: (1) red[e] = sum_p part[p][e] over the blocks of element e's combo
// (e = (co, ci, par*8 + j); 64 elements x 4 partial-slices per block like k_reduce_partials), (2) gw[co][ci][tap] = sum over
// the 8 parities of the collapsed entry that contains `tap`.
__global__ void __launch_bounds__(256) k_reduce_partials_up_sum(const float* __restrict__ part, float* __restrict__ red, int C0, int Cout, int T,
                                                                int Qc, int G, int cog_size) {
    __shared__ float sm[4][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + x, n = Cout * C0 * 64;
    float s0 = 0.0f, s1 = 0.0f;
    if (e < n) {
        const int co = e / (C0 * 64), ci = (e >> 6) % C0;
        const int cb = Qc * G, combo = ci / BW_CKI + Qc * (co / cog_size);
        const int nparts = (T - combo + cb - 1) / cb;
        int p = y;
        for (; p + 4 < nparts; p += 8) {
            s0 += part[(size_t)p * n + e];
            s1 += part[(size_t)(p + 4) * n + e];
        }
        for (; p < nparts; p += 4) s0 += part[(size_t)p * n + e];
    }
    sm[y][x] = s0 + s1;
    __syncthreads();
    if (y == 0 && e < n) red[e] = (sm[0][x] + sm[1][x]) + (sm[2][x] + sm[3][x]);
}
__global__ void __launch_bounds__(256) k_reduce_partials_up_map(const float* __restrict__ red, float* __restrict__ gw, int C0, int Cout, int gw_cin) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (co, ci, tap)
    if (i >= Cout * C0 * 27) return;
    const int tap = i % 27, ci = (i / 27) % C0, co = i / (27 * C0);
    const int k3[3] = {tap / 9, (tap / 3) % 3, tap % 3};
    const float* r = red + ((size_t)co * C0 + ci) * 64;
    float s = 0.0f;
#pragma unroll
    for (int par = 0; par < 8; ++par) {
        const int p3[3] = {(par >> 2) & 1, (par >> 1) & 1, par & 1};
        int j = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) j = j * 2 + (p3[a] ? (k3[a] == 2) : (k3[a] >= 1));
        s += r[par * 8 + j];
    }
    gw[((size_t)co * gw_cin + ci) * 27 + tap] = s;
}

// gw[i] = sum_p part[p][i] (and gb[co] = sum_p part[p][n + co]) in a fixed order (deterministic): 64 outputs x 4
// partial-slices per block, 4 independent accumulators per thread so that the (latency-bound) loads overlap.
// Element i = (co, ci, tap) belongs to combo (ci / 16, co / cog_size), which has cnt = ceil((T - combo) / cb) slots;
// the bias partials live in the chunk-0 combos.
// swapflip: the partials are those of the role-swapped product (see vxm_conv3d_k3_bwd_weight): element (co' = ci, ci' = co, t)
// goes to gw[co][ci][26 - t].
__global__ void __launch_bounds__(256) k_reduce_partials(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb,
                                                         int n, int Cin, int Cout, int T, int Qc, int G, int cog_size, int swapflip,
                                                         int gw_cin, int ci_off) {
    __shared__ float red[4][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + x;
    const int ntot = n + (gb ? Cout : 0);
    const size_t stride = (size_t)n + Cout;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (i < ntot) {
        const int co = i < n ? i / (Cin * 27) : i - n, ci = i < n ? (i / 27) % Cin : 0;
        const int cb = Qc * G, combo = ci / BW_CKI + Qc * (co / cog_size);
        const int nparts = (T - combo + cb - 1) / cb;
        int p = y;
        for (; p + 12 < nparts; p += 16) {
            s0 += part[(size_t)p * stride + i];
            s1 += part[(size_t)(p + 4) * stride + i];
            s2 += part[(size_t)(p + 8) * stride + i];
            s3 += part[(size_t)(p + 12) * stride + i];
        }
        for (; p < nparts; p += 4) s0 += part[(size_t)p * stride + i];
    }
    red[y][x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (y == 0 && i < ntot) {
        const float t = (red[0][x] + red[1][x]) + (red[2][x] + red[3][x]);
        if (i >= n) gb[i - n] = t;
        else if (!swapflip) {                 // gw may be a channel sub-range [ci_off, ci_off + Cin) of a [Cout][gw_cin][27] array
            const int co = i / (Cin * 27), r = i - co * (Cin * 27);
            gw[((size_t)co * gw_cin + ci_off) * 27 + r] = t;
        }
        else {
            const int cop = i / (Cin * 27), cip = (i / 27) % Cin, tap = i % 27;        // Cin = inner extent of the partial = original Cout
            gw[((size_t)cip * Cout + cop) * 27 + (26 - tap)] = t;                       // Cout = outer extent = original Cin
        }
    }
}

// bias gradient of the role-swapped path: gb[co] = sum_{b,v} dz[b,co,v] in two deterministic stages
constexpr int CS_SLICES = 256;
__global__ void __launch_bounds__(256) k_channel_sum_partial(const float* __restrict__ dz, long long dz_bs, float* __restrict__ ws, int B, size_t V) {
    __shared__ float red[4];
    const int co = blockIdx.x;
    float s = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float* p = dz + (size_t)b * dz_bs + (size_t)co * V;
        for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < V; i += (size_t)CS_SLICES * 256) s += p[i];
    }
    s = vxm_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) ws[co * CS_SLICES + blockIdx.y] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(64) k_channel_sum_finish(const float* __restrict__ ws, float* __restrict__ gb) {
    float s = 0.0f;
    for (int i = threadIdx.x; i < CS_SLICES; i += 64) s += ws[blockIdx.x * CS_SLICES + i];
    s = vxm_wave_sum(s);
    if (threadIdx.x == 0) gb[blockIdx.x] = s;
}

struct BwPlan { int NCT, Qc, G, T, nparts; };
BwPlan bw_plan(int Cin, int Cout, int B, int D, int H, int W) {
    BwPlan p;
    p.NCT = Cout <= 16 ? 1 : 2;
    p.Qc = (Cin + BW_CKI - 1) / BW_CKI;
    p.G = (Cout + 16 * p.NCT - 1) / (16 * p.NCT);
    const long long tiles = (long long)B * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    const int cb = p.Qc * p.G;
    long long T = tiles * cb < 256 ? tiles * cb : 256;       // one resident 16-wave block per CU
    if (T < cb) T = cb;                                       // every combo needs a block
    p.T = (int)T;
    p.nparts = (p.T + cb - 1) / cb;
    return p;
}

int check_conv(const char* fn, int C0, int C1, int x0_up, int Cout, int B, int D, int H, int W) {
    VXM_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, VXM_ERR_BAD_SHAPE,
                "%s: bad shape B=%d C0=%d C1=%d Cout=%d D=%d H=%d W=%d", fn, B, C0, C1, Cout, D, H, W);
    VXM_REQUIRE(!x0_up || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), VXM_ERR_BAD_SHAPE,
                "%s: upsampled segment needs even extents, got %dx%dx%d", fn, D, H, W);
    VXM_REQUIRE((long long)(C0 + C1 > Cout ? C0 + C1 : Cout) * D * H * W < (1ll << 29), VXM_ERR_BAD_SHAPE,
                "%s: a tensor of one sample must stay below 2 GiB (32-bit byte offsets in the buffer descriptors)", fn);
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
    // large layers: the 8-wave wide-load kernel (needs 4-float groups that neither straddle row ends nor break alignment)
    {
        const long long tiles8 = (long long)B * ((D + T8_TD - 1) / T8_TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
        const bool vec = fwd_wide_ok(c, x0, x0_bstride, x1, C1, x1_bstride, wpacked, B, D, H, W);
        if (vec) {
            static bool opt_in = false;
            if (!opt_in) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_t8<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
                opt_in = true;
            }
            const dim3 grid8((unsigned)((tiles8 + 7) / 8 * 8), c.G);
            const size_t lds8 = sizeof(float) * ((size_t)8 * T8_PS + 27 * 2 * c.NCT * 64);
#define LAUNCH8(NCT_) hipLaunchKernelGGL((k_conv3d_k3_t8<NCT_>), grid8, dim3(T8_THREADS), lds8, VXM_STREAM(stream), in, wpacked, bias, y, \
        (long long)y_bstride, Cout, act_slope, mask_src, (long long)mask_bstride, mask_slope, B, D, H, W, c.Q)
            if (c.NCT == 1) LAUNCH8(1);
            else LAUNCH8(2);
#undef LAUNCH8
            return vxm_check_launch("vxm_conv3d_k3_fwd");
        }
    }
    const long long tiles = (long long)B * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_fwd: too many tiles");
    const dim3 grid((unsigned)tiles, c.G);
    const size_t lds = sizeof(float) * ((size_t)c.CK * FWD_PS + 27 * (c.CK / 4) * c.NCT * 64);
#define LAUNCH(CK_, NCT_) hipLaunchKernelGGL((k_conv3d_k3<CK_, NCT_>), grid, dim3(256), lds, VXM_STREAM(stream), in, wpacked, bias, y, \
        (long long)y_bstride, Cout, act_slope, mask_src, (long long)mask_bstride, mask_slope, D, H, W, c.Q)
    if (lds > 64 * 1024) {           // <8,3>: opt in to > 64 KB of dynamic LDS once
        static bool opt_in = false;
        if (!opt_in) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3<8, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            opt_in = true;
        }
    }
    if (c.CK == 4 && c.NCT == 1) LAUNCH(4, 1);
    else if (c.CK == 4 && c.NCT == 2) LAUNCH(4, 2);
    else if (c.CK == 4) LAUNCH(4, 3);
    else if (c.NCT == 1) LAUNCH(8, 1);
    else if (c.NCT == 2) LAUNCH(8, 2);
    else LAUNCH(8, 3);
#undef LAUNCH
    return vxm_check_launch("vxm_conv3d_k3_fwd");
}

int vxm_conv3d_k3_fwd_variant(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride,
                              const float* wpacked, int Cout, int B, int D, int H, int W) {
    if (C0 <= 0 || C1 < 0 || Cout <= 0) return -1;
    const ConvCfg c = conv_cfg(C0 + C1, Cout);
    return (fwd_wide_ok(c, x0, x0_bstride, x1, C1, x1_bstride, wpacked, B, D, H, W) ? 100 : 0) + 10 * c.CK + c.NCT;
}

int vxm_conv3d_k3_bwd_weight_variant(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                                     const float* dz, int64_t dz_bstride, int Cout, int D, int H, int W) {
    const bool vec = bwd_weight_wide_ok(x0, x0_bstride, x1, C1, x1_bstride, dz, dz_bstride, W);
    const int nct = Cout <= 16 ? 1 : 2;                 // of the unswapped plan
    if (x0_up && vec && (D & 1) == 0 && (H & 1) == 0) return 20 + nct;      // collapsed upsampled segment (+ regular skip segment)
    return (vec ? 10 : 0) + nct;
}

int vxm_conv3d_k3_up_ok(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, float* y,
                        int Cout, int B, int D, int H, int W) {
    if (C0 <= 0 || C1 < 0 || Cout <= 0 || Cout > 32 && ((Cout + 47) / 48 * 48 < (Cout + 31) / 32 * 32)) return 0;
    const long long tiles8 = (long long)B * ((D + T8_TD - 1) / T8_TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    return (W & 3) == 0 && (D & 1) == 0 && (H & 1) == 0 && (C1 == 0 || al16(x1)) && (x1_bstride & 3) == 0 && x0 && y &&
           (long long)C0 * (D / 2) * (H / 2) * (W / 2) < (1ll << 29) && tiles8 >= wide_min_tiles() && tiles8 < (1ll << 30) &&
           !bw_force_generic() && (x0_bstride >= 0);
}

// 16 output channels per block: the 2-tile instance needs ~150 VGPRs (both MFMA loops + two kinds of staging registers)
// and spills at the 128 that two 8-wave blocks per CU allow; two 1-tile blocks re-read X but do not spill.
static int up_nct(int Cout) { (void)Cout; return 1; }

size_t vxm_conv3d_k3_up_packed_elems(int C0, int C1, int Cout) {
    if (C0 <= 0 || C1 < 0 || Cout <= 0) return 0;
    const int NCT = up_nct(Cout), G = (Cout + 16 * NCT - 1) / (16 * NCT);
    const size_t Q0 = (C0 + TU_CK0 - 1) / TU_CK0, Q1 = (C1 + 7) / 8;
    return (size_t)G * (Q0 * 8 * 8 * 2 * NCT * 64 + Q1 * 27 * 2 * NCT * 64);
}

int vxm_conv3d_k3_up_pack_weights(const float* w, float* wpacked, int C0, int C1, int Cout, void* stream) {
    VXM_REQUIRE(w && wpacked, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_up_pack_weights: null pointer");
    VXM_REQUIRE(C0 > 0 && C1 >= 0 && Cout > 0, VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_up_pack_weights: C0=%d C1=%d Cout=%d", C0, C1, Cout);
    const size_t elems = vxm_conv3d_k3_up_packed_elems(C0, C1, Cout);
    hipLaunchKernelGGL(k_pack_weights_up, dim3(vxm_blocks((long long)elems, 256)), dim3(256), 0, VXM_STREAM(stream), w, wpacked, C0, C1, Cout,
                       up_nct(Cout), elems);
    return vxm_check_launch("vxm_conv3d_k3_up_pack_weights");
}

int vxm_conv3d_k3_up_fwd(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* wpacked,
                         const float* bias, float* y, int64_t y_bstride, int Cout, float act_slope, int B, int D, int H, int W, void* stream) {
    if (int e = check_conv("vxm_conv3d_k3_up_fwd", C0, C1, 1, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(x0 && wpacked && y && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_up_fwd: null pointer");
    VXM_REQUIRE(vxm_conv3d_k3_up_ok(x0, C0, x0_bstride, x1, C1, x1_bstride, y, Cout, B, D, H, W) && al16(wpacked), VXM_ERR_UNSUPPORTED,
                "vxm_conv3d_k3_up_fwd: operands do not qualify (see vxm_conv3d_k3_up_ok); use vxm_conv3d_k3_fwd with x0_up = 1");
    const int NCT = up_nct(Cout), G = (Cout + 16 * NCT - 1) / (16 * NCT);
    const long long tiles8 = (long long)B * ((D + T8_TD - 1) / T8_TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    static bool opt_in = false;
    if (!opt_in) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_t8u<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        opt_in = true;
    }
    const dim3 grid((unsigned)((tiles8 + 7) / 8 * 8), G);
#define TU_LAUNCH(NCT_) hipLaunchKernelGGL((k_conv3d_k3_t8u<NCT_>), grid, dim3(T8_THREADS), sizeof(float) * (size_t)tu_lds_floats<NCT_>(), \
        VXM_STREAM(stream), x0, (long long)x0_bstride, C0, x1, (long long)x1_bstride, C1, wpacked, bias, y, (long long)y_bstride, Cout, act_slope, B, D, H, W)
    (void)NCT;
    TU_LAUNCH(1);
#undef TU_LAUNCH
    return vxm_check_launch("vxm_conv3d_k3_up_fwd");
}

static int dlow_nct(int C0) { return C0 <= 16 ? 1 : 2; }

int vxm_conv3d_k3_up_bwd_low_ok(const float* dz, int64_t dz_bstride, int C0, int Cout, int B, int D, int H, int W) {
    const long long tiles = (long long)B * ((D / 2 + 1) / 2) * ((H / 2 + 3) / 4) * ((W / 2 + 15) / 16);
    return C0 > 0 && Cout > 0 && (W & 3) == 0 && (D & 1) == 0 && (H & 1) == 0 && al16(dz) && (dz_bstride & 3) == 0 &&
           (long long)Cout * D * H * W < (1ll << 29) && tiles >= (wide_min_tiles() + 3) / 4 && tiles < (1ll << 30) && !bw_force_generic();
}

size_t vxm_conv3d_k3_up_bwd_low_packed_elems(int C0, int Cout) {
    if (C0 <= 0 || Cout <= 0) return 0;
    const int NCT = dlow_nct(C0), G = (C0 + 16 * NCT - 1) / (16 * NCT), Q = (Cout + 3) / 4;
    return (size_t)G * Q * 64 * NCT * 64;
}

int vxm_conv3d_k3_up_bwd_low(const float* dz, int64_t dz_bstride, int Cout, const float* w, int C0, int Cin, float* wpacked, float* gx_low,
                             int64_t gx_bstride, const float* mask_low, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W,
                             void* stream) {
    if (int e = check_conv("vxm_conv3d_k3_up_bwd_low", C0, Cin - C0, 1, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(dz && w && wpacked && gx_low, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_up_bwd_low: null pointer");
    VXM_REQUIRE(C0 <= Cin && vxm_conv3d_k3_up_bwd_low_ok(dz, dz_bstride, C0, Cout, B, D, H, W) && al16(wpacked), VXM_ERR_UNSUPPORTED,
                "vxm_conv3d_k3_up_bwd_low: operands do not qualify (see vxm_conv3d_k3_up_bwd_low_ok)");
    const int NCT = dlow_nct(C0), G = (C0 + 16 * NCT - 1) / (16 * NCT);
    const size_t elems = vxm_conv3d_k3_up_bwd_low_packed_elems(C0, Cout);
    hipLaunchKernelGGL(k_pack_weights_dlow, dim3(vxm_blocks((long long)elems, 256)), dim3(256), 0, VXM_STREAM(stream), w, wpacked, C0, Cin, Cout, NCT, elems);
    static bool opt_in = false;
    if (!opt_in) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_dlow<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_dlow<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        opt_in = true;
    }
    const long long tiles = (long long)B * ((D / 2 + 1) / 2) * ((H / 2 + 3) / 4) * ((W / 2 + 15) / 16);
    const dim3 grid((unsigned)tiles, G);
#define DL_LAUNCH(NCT_) hipLaunchKernelGGL((k_conv3d_k3_dlow<NCT_>), grid, dim3(T8_THREADS), sizeof(float) * (size_t)dl_lds_floats<NCT_>(), VXM_STREAM(stream), \
        dz, (long long)dz_bstride, Cout, wpacked, gx_low, (long long)gx_bstride, C0, mask_low, (long long)mask_bstride, mask_slope, B, D, H, W)
    if (NCT == 1) DL_LAUNCH(1);
    else DL_LAUNCH(2);
#undef DL_LAUNCH
    return vxm_check_launch("vxm_conv3d_k3_up_bwd_low");
}

int vxm_conv3d_k3_fewout_ok(const float* x, int64_t x_bstride, float* y, int64_t y_bstride, int Cin, int Cout, int W) {
    return Cout >= 1 && Cout <= 4 && Cin >= 1 && (W & 3) == 0 && al16(x) && al16(y) && (x_bstride & 3) == 0 && (y_bstride & 3) == 0 &&
           !bw_force_generic();
}

int vxm_conv3d_k3_fewout_fwd(const float* x, int Cin, int64_t x_bstride, const float* w, const float* bias, float* y, int64_t y_bstride,
                             int Cout, float act_slope, int B, int D, int H, int W, void* stream) {
    if (int e = check_conv("vxm_conv3d_k3_fewout_fwd", Cin, 0, 0, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(x && w && y, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_fewout_fwd: null pointer");
    VXM_REQUIRE(vxm_conv3d_k3_fewout_ok(x, x_bstride, y, y_bstride, Cin, Cout, W), VXM_ERR_UNSUPPORTED,
                "vxm_conv3d_k3_fewout_fwd: needs Cout <= 4, W %% 4 == 0 and 16-byte aligned tensors (use vxm_conv3d_k3_fwd)");
    VXM_REQUIRE(B <= 65535 && Cin <= 512, VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_fewout_fwd: B <= 65535, Cin <= 512");
    const long long tiles = (long long)((W + FO_TW - 1) / FO_TW) * ((H + FO_TH - 1) / FO_TH) * ((D + FO_TD - 1) / FO_TD);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_fewout_fwd: too many tiles");
    const size_t lds = sizeof(float) * (size_t)FO_CK * FO_PS;
    const dim3 grid((unsigned)tiles, B);
#define FO_LAUNCH(CO_) hipLaunchKernelGGL(k_conv3d_k3_fewout<CO_>, grid, dim3(256), lds, VXM_STREAM(stream), x, (long long)x_bstride, Cin, w, bias, y, \
        (long long)y_bstride, act_slope, D, H, W)
    switch (Cout) {
        case 1: FO_LAUNCH(1); break;
        case 2: FO_LAUNCH(2); break;
        case 3: FO_LAUNCH(3); break;
        default: FO_LAUNCH(4); break;
    }
#undef FO_LAUNCH
    return vxm_check_launch("vxm_conv3d_k3_fewout_fwd");
}

size_t vxm_conv3d_k3_bwd_weight_workspace_bytes(int Cin, int Cout, int B, int D, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const BwPlan p = bw_plan(Cin, Cout, B, D, H, W);
    size_t need = sizeof(float) * (size_t)p.nparts * ((size_t)Cout * Cin * 27 + Cout);
    if (Cout <= 4) {                                   // role-swapped product (+ the channel-sum scratch of its bias gradient)
        const BwPlan q = bw_plan(Cout, Cin, B, D, H, W);
        const size_t alt = sizeof(float) * ((size_t)q.nparts * ((size_t)Cout * Cin * 27 + Cin) + (size_t)Cout * CS_SLICES);
        if (alt > need) need = alt;
    }
    {                                                  // collapsed product of an upsampled segment: 64 instead of 27 entries per (co, ci)
        // nparts(C0) * C0 <= (256 / (Qc G) + 1) * 16 Qc <= 4096 / G + Cin + 16 for any split C0 <= Cin
        const size_t alt = sizeof(float) * ((size_t)Cout * 64 * (4096 / (size_t)p.G + Cin + 16) + (size_t)Cout * CS_SLICES + (size_t)Cout * Cin * 64);
        if (alt > need) need = alt;
    }
    return 256 + need;
}

int vxm_conv3d_k3_bwd_weight(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                             const float* dz, int64_t dz_bstride, int Cout, float* gw, float* gb, void* workspace,
                             size_t workspace_bytes, int B, int D, int H, int W, void* stream) {
    if (int e = check_conv("vxm_conv3d_k3_bwd_weight", C0, C1, x0_up, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(x0 && dz && gw && workspace && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_bwd_weight: null pointer");
    const int Cin = C0 + C1;
    VXM_REQUIRE(workspace_bytes >= vxm_conv3d_k3_bwd_weight_workspace_bytes(Cin, Cout, B, D, H, W), VXM_ERR_WORKSPACE,
                "vxm_conv3d_k3_bwd_weight: workspace too small (%zu bytes)", workspace_bytes);
    // wide-load path: rows of 4-float groups must not straddle row ends and must be 16-byte aligned in memory
    const bool vec = bwd_weight_wide_ok(x0, x0_bstride, x1, C1, x1_bstride, dz, dz_bstride, W);
    const bool swap = bwd_weight_swap_ok(C0, C1, x0_up, Cout, vec);
    // workspace: per-block partials [nparts][Cout*Cin*27 + (Cout | Cin)] (+ channel-sum scratch when swapped)
    uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255;
    float* part = reinterpret_cast<float*>(base);
    // up to 161 KB of dynamic LDS (> the 64 KB default cap): opt in once per kernel
    static bool lds_opt_in = false;
    if (!lds_opt_in) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_bwd_weight_vec<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_bwd_weight_vec<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_bwd_weight_dma<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_bwd_weight_dma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        lds_opt_in = true;
    }
    const int n = Cout * Cin * 27;
    if (x0_up && vec && (D & 1) == 0 && (H & 1) == 0 && (long long)C0 * (D / 2) * (H / 2) * (W / 2) < (1ll << 29)) {
        // upsampled segment: collapsed product (8 x 8 low-resolution taps), then the skip segment alone through the regular kernel
        static bool up_opt_in = false;
        if (!up_opt_in) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_bwd_weight_up<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_bwd_weight_up<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            up_opt_in = true;
        }
        const BwPlan u = bw_plan(C0, Cout, B, D, H, W);
        if (u.NCT == 1)
            hipLaunchKernelGGL(k_conv3d_k3_bwd_weight_up<1>, dim3(u.T), dim3(BW_THREADS), sizeof(float) * 2 * (size_t)bu_buf_floats<1>(), VXM_STREAM(stream),
                               x0, (long long)x0_bstride, C0, dz, (long long)dz_bstride, Cout, part, B, D, H, W, u.Qc, u.G);
        else
            hipLaunchKernelGGL(k_conv3d_k3_bwd_weight_up<2>, dim3(u.T), dim3(BW_THREADS), sizeof(float) * 2 * (size_t)bu_buf_floats<2>(), VXM_STREAM(stream),
                               x0, (long long)x0_bstride, C0, dz, (long long)dz_bstride, Cout, part, B, D, H, W, u.Qc, u.G);
        float* red = part + (size_t)u.nparts * (size_t)Cout * C0 * 64 + (size_t)Cout * CS_SLICES;      // behind the partials and the channel-sum scratch
        hipLaunchKernelGGL(k_reduce_partials_up_sum, dim3(vxm_blocks((long long)Cout * C0 * 64, 64)), dim3(256), 0, VXM_STREAM(stream), part, red, C0, Cout,
                           u.T, u.Qc, u.G, 16 * u.NCT);
        hipLaunchKernelGGL(k_reduce_partials_up_map, dim3(vxm_blocks((long long)Cout * C0 * 27, 256)), dim3(256), 0, VXM_STREAM(stream), red, gw, C0, Cout, Cin);
        if (C1 > 0) {
            const BwPlan s1 = bw_plan(C1, Cout, B, D, H, W);
            ConvIn sin{x1, nullptr, (long long)x1_bstride, 0, C1, 0, 0};
            const int n1 = Cout * C1 * 27;
            if (s1.NCT == 1)
                hipLaunchKernelGGL(k_conv3d_k3_bwd_weight_vec<1>, dim3(s1.T), dim3(BW_THREADS), sizeof(float) * 2 * (size_t)bv_lds_floats<1>(), VXM_STREAM(stream),
                                   sin, dz, (long long)dz_bstride, Cout, gb ? 1 : 0, part, B, D, H, W, s1.Qc, s1.G);
            else
                hipLaunchKernelGGL(k_conv3d_k3_bwd_weight_vec<2>, dim3(s1.T), dim3(BW_THREADS), sizeof(float) * 2 * (size_t)bv_lds_floats<2>(), VXM_STREAM(stream),
                                   sin, dz, (long long)dz_bstride, Cout, gb ? 1 : 0, part, B, D, H, W, s1.Qc, s1.G);
            hipLaunchKernelGGL(k_reduce_partials, dim3(vxm_blocks(n1 + (gb ? Cout : 0), 64)), dim3(256), 0, VXM_STREAM(stream), part, gw, gb, n1, C1, Cout,
                               s1.T, s1.Qc, s1.G, 16 * s1.NCT, 0, Cin, C0);
        } else if (gb) {
            float* cs = part + (size_t)u.nparts * (size_t)Cout * C0 * 64;
            hipLaunchKernelGGL(k_channel_sum_partial, dim3(Cout, CS_SLICES), dim3(256), 0, VXM_STREAM(stream), dz, (long long)dz_bstride, cs, B, (size_t)D * H * W);
            hipLaunchKernelGGL(k_channel_sum_finish, dim3(Cout), dim3(64), 0, VXM_STREAM(stream), cs, gb);
        }
        return vxm_check_launch("vxm_conv3d_k3_bwd_weight");
    }
#define BW_LAUNCH(KERNEL, LDSF, IN_, DZ_, DZBS_, CO_, BIAS_, P_) hipLaunchKernelGGL(KERNEL, dim3((P_).T), dim3(BW_THREADS), sizeof(float) * (size_t)(LDSF), \
        VXM_STREAM(stream), IN_, DZ_, (long long)(DZBS_), CO_, BIAS_, part, B, D, H, W, (P_).Qc, (P_).G)
    if (swap) {
        const BwPlan q = bw_plan(Cout, Cin, B, D, H, W);          // "input" = dz (Cout channels), "output gradient" = x (Cin channels)
        ConvIn sin{dz, nullptr, (long long)dz_bstride, 0, Cout, 0, 0};
        if (q.NCT == 1) BW_LAUNCH(k_conv3d_k3_bwd_weight_vec<1>, 2 * bv_lds_floats<1>(), sin, x0, x0_bstride, Cin, 0, q);
        else BW_LAUNCH(k_conv3d_k3_bwd_weight_vec<2>, 2 * bv_lds_floats<2>(), sin, x0, x0_bstride, Cin, 0, q);
        hipLaunchKernelGGL(k_reduce_partials, dim3(vxm_blocks(n, 64)), dim3(256), 0, VXM_STREAM(stream), part, gw, (float*)nullptr, n, Cout, Cin,
                           q.T, q.Qc, q.G, 16 * q.NCT, 1, Cin, 0);
        if (gb) {
            float* cs = part + (size_t)q.nparts * ((size_t)n + Cin);
            hipLaunchKernelGGL(k_channel_sum_partial, dim3(Cout, CS_SLICES), dim3(256), 0, VXM_STREAM(stream), dz, (long long)dz_bstride, cs, B,
                               (size_t)D * H * W);
            hipLaunchKernelGGL(k_channel_sum_finish, dim3(Cout), dim3(64), 0, VXM_STREAM(stream), cs, gb);
        }
        return vxm_check_launch("vxm_conv3d_k3_bwd_weight");
    }
    const BwPlan p = bw_plan(Cin, Cout, B, D, H, W);
    ConvIn in{x0, x1, (long long)x0_bstride, (long long)x1_bstride, C0, C1, x0_up};
    if (vec) {
        if (p.NCT == 1) BW_LAUNCH(k_conv3d_k3_bwd_weight_vec<1>, 2 * bv_lds_floats<1>(), in, dz, dz_bstride, Cout, gb ? 1 : 0, p);
        else BW_LAUNCH(k_conv3d_k3_bwd_weight_vec<2>, 2 * bv_lds_floats<2>(), in, dz, dz_bstride, Cout, gb ? 1 : 0, p);
    } else {
        if (p.NCT == 1) BW_LAUNCH(k_conv3d_k3_bwd_weight_dma<1>, 2 * bw_buf_floats<1>(), in, dz, dz_bstride, Cout, gb ? 1 : 0, p);
        else BW_LAUNCH(k_conv3d_k3_bwd_weight_dma<2>, 2 * bw_buf_floats<2>(), in, dz, dz_bstride, Cout, gb ? 1 : 0, p);
    }
#undef BW_LAUNCH
    // the bias gradient rides along: its per-block partials sit behind the weight partials of every slot
    hipLaunchKernelGGL(k_reduce_partials, dim3(vxm_blocks(n + (gb ? Cout : 0), 64)), dim3(256), 0, VXM_STREAM(stream), part, gw, gb, n, Cin, Cout,
                       p.T, p.Qc, p.G, 16 * p.NCT, 0, Cin, 0);
    return vxm_check_launch("vxm_conv3d_k3_bwd_weight");
}

}  // extern "C"
