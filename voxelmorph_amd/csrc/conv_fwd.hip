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
// (backward-weight: conv_bwd_weight.hip)
#include "conv_common.h"

namespace {

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

    // ---- epilogue: bias + LeakyReLU (+ fused leaky_relu_backward mask), NCDHW store (branch-free buffer stores)
    const int d = d0 + wave, w = w0 + n;
    const int V = D * H * W;
    float bz[NCT][4];
    conv_load_bias<NCT>(bz, bias, Cout, g, kq);
    conv_epilogue_store<NCT, 4>(acc, y + (size_t)b * y_bs, bz, mask ? mask + (size_t)b * mask_bs : nullptr, act_slope, mask_slope, Cout, g, kq,
                                d < D && w < W, (d * H + h0) * W + w, h0, H, W, V);
}

// ------------------------------------------------------------------------------------------
// forward / backward-data kernel for SMALL volumes (the U-Net levels at 1/8 and 1/16 resolution: 20x24x28, 10x12x14)
// ------------------------------------------------------------------------------------------
// k_conv3d_k3 covers such a level with <= 60 blocks of four waves, and every wave walks all Cin x 27 taps of its depth slice: 432 Q
// dependent-issue MFMAs of 32 cycles on 240 of the chip's 1024 SIMDs -- 38 - 75 us per launch for 0.7 - 1.5 GFLOP (round 6 dispatch trace),
// six launches on the step's critical path.  Here the SAME implicit GEMM (same packed operator, same fragments) is cut four ways more:
//  * output tile 2(D) x 4(H) x 16(W) voxels x ONE 16-channel output tile (blockIdx.y), so a 32-channel layer at 20x24x28 is 240 blocks;
//  * the input channels are split over the NW waves of a block (wave v multiplies the 8-channel chunks q = v, v + NW, ...): each wave
//    stages its own chunk into its own LDS region (no block barrier while multiplying) and keeps the chunk's 54 weight fragments in
//    registers, straight from the packed operator in L2;
//  * the NW partial tiles meet in LDS and are added in wave order 0 .. NW - 1 (deterministic), wave v finishing 8 / NW of the tile's rows.
// The sum over the input channels is therefore associated differently from k_conv3d_k3 (per-wave fp32 chains, then NW - 1 adds):
// same accuracy class (tests/test_gpu_parity.py gates both against fp64), not the same bits.
constexpr int SM_TD = 2, SM_HD = SM_TD + 2;
constexpr int SM_PS = 496;                      // channel plane [SM_HD][6][20] = 480 floats, padded to 16 mod 32 (as FWD_PS)
constexpr int SM_XW = 8 * SM_PS;                // LDS floats of one wave: its haloed 8-channel chunk, later its partial tile (8 x 64 x 4 floats)

template <int NW>
__global__ void __launch_bounds__(64 * NW) k_conv3d_k3_sm(ConvIn in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                         float* __restrict__ y, long long y_bs, int Cout, float act_slope,
                                                         const float* __restrict__ mask, long long mask_bs, float mask_slope,
                                                         int D, int H, int W, int Q, int PNCT) {
    VXM_DYN_SMEM(float, smem);
    constexpr int PER = 8 / NW;                     // rows of the tile a wave finishes
    static_assert(NW == 2 || NW == 4 || NW == 8, "a wave finishes rows of ONE depth slice");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n = lane & 15;
    const int nw = (W + TW - 1) / TW, nh = (H + TH - 1) / TH, nd = (D + SM_TD - 1) / SM_TD;
    int t = blockIdx.x;
    const int tw = t % nw; t /= nw;
    const int th = t % nh; t /= nh;
    const int td = t % nd; const int b = t / nd;
    const int d0 = td * SM_TD, h0 = th * TH, w0 = tw * TW;
    const int gc = blockIdx.y, g = gc / PNCT, ct = gc - g * PNCT;       // 16-channel output tile gc = tile ct of the operator's group g
    float* const Xs = smem + wave * SM_XW;

    f32x4 acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging addresses: the lane part (row, column of a 3-row slab: byte offset inside a depth plane, full-resolution and through the x2
    // upsampling gather; beyond the volume -> out of the descriptor's range -> 0.0) is computed once, channel / depth are scalar offsets
    const SlabLane L = make_slab_lane(lane, h0, w0, W);
    const int lds_lane = L.rr * FWD_TWP + L.wx;
    const int bbase = kq * SM_PS + n;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    int voffF[2], voffU[2];
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int gh = L.gh0 + 3 * hb;
        const bool ok = L.wok && (unsigned)gh < (unsigned)H;
        voffF[hb] = ok ? (gh * W + L.gw) << 2 : VXM_OOB;
        voffU[hb] = ok ? ((gh >> 1) * Wl + (L.gw >> 1)) << 2 : VXM_OOB;
    }
    const int Vfull = D * H * W, V0 = in.up0 ? Dl * Hl * Wl : Vfull;
    const __amdgpu_buffer_rsrc_t r0 = vxm_rsrc(in.x0 + (size_t)b * in.bs0, (unsigned)in.C0 * (unsigned)V0 * 4u);
    const __amdgpu_buffer_rsrc_t r1 = vxm_rsrc(in.C1 ? in.x1 + (size_t)b * in.bs1 : in.x0, (unsigned)in.C1 * (unsigned)Vfull * 4u);
    const int wchunk = 27 * 2 * PNCT * 64;
    for (int q = wave; q < Q; q += NW) {
        float a[54];
        const __amdgpu_buffer_rsrc_t rw = vxm_rsrc(wp + ((size_t)g * Q + q) * wchunk, (unsigned)wchunk * 4u);
#pragma unroll
        for (int i = 0; i < 54; ++i) a[i] = vxm_bload(rw, (ct * 64 + lane) << 2, (i * PNCT) << 8);
        constexpr int XIT = 8 * SM_HD * 2;
        float xv[XIT];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int cg = q * 8 + c;                                   // wave-uniform
            const bool s0 = cg < in.C0, up = s0 && in.up0;
            const __amdgpu_buffer_rsrc_t r = s0 ? r0 : r1;
            const int cc = s0 ? cg : cg - in.C0;
            const int Ds = up ? Dl : D, PV = up ? Hl * Wl : H * W;
#pragma unroll
            for (int dz = 0; dz < SM_HD; ++dz) {
                const int d = d0 + dz - 1;
                const bool ok = cg < in.C0 + in.C1 && (unsigned)d < (unsigned)D;
                const int soff = ok ? ((cc * Ds + (up ? d >> 1 : d)) * PV) << 2 : 0;
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
                    xv[(c * SM_HD + dz) * 2 + hb] = vxm_bload(r, ok ? (up ? voffU[hb] : voffF[hb]) : VXM_OOB, soff);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // (second chunk of a wave: its reads of the first are done)
        if (L.act) {
#pragma unroll
            for (int it = 0; it < XIT; ++it) {
                const int c = it / (SM_HD * 2), dz = (it >> 1) % SM_HD, hb = it & 1;
                Xs[c * SM_PS + (dz * HH + hb * 3) * FWD_TWP + lds_lane] = xv[it];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");          // the region is this wave's own: no block barrier
#pragma unroll
        for (int tp = 0; tp < 27; ++tp) {
            const int kd = tp / 9, kh = (tp / 3) % 3, kw = tp % 3;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float bv[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) bv[m] = Xs[bbase + s * 4 * SM_PS + (((m >> 2) + kd) * HH + (m & 3) + kh) * FWD_TWP + kw];
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[m] = vxm_mfma16(a[tp * 2 + s], bv[m], acc[m]);
            }
        }
    }

    // ---- the NW partial tiles meet in LDS (each wave's own region, its chunk is read), added in wave order
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int m = 0; m < 8; ++m) reinterpret_cast<f32x4*>(Xs)[m * 64 + lane] = acc[m];
    __syncthreads();
    f32x4 fin[1][PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int m = wave * PER + p;
        f32x4 sum = reinterpret_cast<const f32x4*>(smem)[m * 64 + lane];
#pragma unroll
        for (int v = 1; v < NW; ++v) sum += reinterpret_cast<const f32x4*>(smem + v * SM_XW)[m * 64 + lane];
        fin[0][p] = sum;
    }
    // ---- epilogue of rows [row0, row0 + PER) of depth slice dd: bias + LeakyReLU (+ fused leaky_relu_backward mask), NCDHW store
    const int dd = (wave * PER) >> 2, row0 = (wave * PER) & 3;
    const int d = d0 + dd, w = w0 + n;
    const int V = Vfull;
    float bz[1][4];
    conv_load_bias<1>(bz, bias, Cout, gc, kq);
    conv_epilogue_store<1, PER>(fin, y + (size_t)b * y_bs, bz, mask ? mask + (size_t)b * mask_bs : nullptr, act_slope, mask_slope, Cout, gc, kq,
                                d < D && w < W, (d * H + h0 + row0) * W + w, h0 + row0, H, W, V);
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
constexpr int T8_THREADS = 512;

// ROWS = H rows per wave (tile 8 x ROWS x 16).  4 is used throughout: 6 rows with a single output tile (324 instead of 216 MFMAs
// per wave and staged chunk) measured 4 % slower, 8 rows need more than the 128 VGPRs that two blocks per CU allow.
// TWX = tile width: 16 (4 rows x 16 voxels per wave) or 32 (2 rows x 32 voxels: the output then leaves in 128-byte runs --
// with 16-wide tiles the two halves of every 128-byte line are written by different blocks at different times, which the
// memory system takes at 3.6 instead of 5.0 TB/s, tools/probe/store_pattern.hip).  ROWS = tile rows, MT = ROWS TWX / 16 MFMA
// N-tiles per wave and output-channel tile.
template <int NCT, int ROWS, int TWX = 16>
__global__ void __launch_bounds__(T8_THREADS, 4) k_conv3d_k3_t8(ConvIn in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                              float* __restrict__ y, long long y_bs, int Cout, float act_slope,
                                                              const float* __restrict__ mask, long long mask_bs, float mask_slope,
                                                              int B, int D, int H, int W, int Q) {
    VXM_DYN_SMEM(float, smem);
    constexpr int CK = 8, KS = 2, RS = TWX + 4;           // rows: halo column at 1, interior at 2 .. TWX+1, halo at TWX+2
    constexpr int HALVES = TWX / 16, MT = ROWS * HALVES, GPR = TWX / 4, RPL = 64 / GPR;      // dwordx4 groups per row, rows per wave-load
    constexpr int HR = ROWS + 2, NROW = (T8_TD + 2) * HR;                        // haloed rows per plane
    constexpr int PS = NROW * RS + ((NROW * RS) % 32 == 16 ? 0 : 16);            // plane stride = 16 mod 32 (1200 / 1616 / 2000)
    constexpr int NI = (NROW * GPR + 63) / 64, NHL = (NROW * 2 + 63) / 64;        // interior / halo wave-loads per plane
    constexpr int WCHUNK = 27 * KS * NCT * 64;
    constexpr int WIT = (WCHUNK / 4 + T8_THREADS - 1) / T8_THREADS;
    float* const Xs = smem;                         // [CK][PS]
    float* const Ws = smem + CK * PS;               // [27][KS][NCT][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n = lane & 15;
    const float* const ix0 = in.x0; const float* const ix1 = in.x1;
    const int iC0 = in.C0, iC1 = in.C1, iup0 = in.up0;

    // tile of this block: block b runs on XCD b % 8 (observed; speed only), XCD x takes the contiguous tile range
    // [nt x / 8, nt (x+1) / 8) so that concurrently running neighbours share halo lines and weights in one L2
    const int nw = (W + TWX - 1) / TWX, nh = (H + ROWS - 1) / ROWS, nd = (D + T8_TD - 1) / T8_TD;
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
    const int d0 = td * T8_TD, h0 = th * ROWS, w0 = tw * TWX;
    const int g = blockIdx.y;                       // output-channel group of 16*NCT

    const int V = D * H * W;
    const int Hs = H >> 1, Ws2 = W >> 1;
    const int V0 = iup0 ? (D >> 1) * Hs * Ws2 : V;
    const __amdgpu_buffer_rsrc_t r0 = vxm_rsrc(ix0 + (size_t)b * in.bs0, (unsigned)iC0 * (unsigned)V0 * 4u);
    const __amdgpu_buffer_rsrc_t r1 = vxm_rsrc(iC1 ? ix1 + (size_t)b * in.bs1 : ix0, (unsigned)iC1 * (unsigned)V * 4u);

    // staging roles of this lane: interior slot 64 j + lane (< 240) -> row 16 j + (lane >> 2) of the [10][6] row grid,
    // columns 4 (lane & 3)..+3; halo slot 64 j + lane (< 120) -> row 32 j + (lane >> 1), side lane & 1.
    const int lq = lane % GPR, lr4 = lane / GPR, lr2 = lane >> 1, hside = lane & 1;
    const int ibase = lr4 * RS + 2 + 4 * lq;
    const int hbase = lr2 * RS + (hside ? TWX + 2 : 1);
    // per-lane byte offsets inside a plane (for the upsampled segment: of the half-resolution source); recomputed per chunk
    // -- a few dozen VALU against 432 MFMAs -- rather than kept in 2 (NI + NHL) registers
    auto off_i = [&](int j, bool up) __attribute__((always_inline)) -> int {
        const int rr = RPL * j + lr4;
        const int gd = d0 - 1 + rr / HR, gh = h0 - 1 + rr % HR, gw = w0 + 4 * lq;
        const bool ok = rr < NROW && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw < W;
        return !ok ? VXM_OOB : (up ? (((gd >> 1) * Hs + (gh >> 1)) * Ws2 + (gw >> 1)) << 2 : ((gd * H + gh) * W + gw) << 2);
    };
    auto off_h = [&](int j, bool up) __attribute__((always_inline)) -> int {
        const int rr = 32 * j + lr2;
        const int gd = d0 - 1 + rr / HR, gh = h0 - 1 + rr % HR, gw = hside ? w0 + TWX : w0 - 1;
        const bool ok = rr < NROW && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        return !ok ? VXM_OOB : (up ? (((gd >> 1) * Hs + (gh >> 1)) * Ws2 + (gw >> 1)) << 2 : ((gd * H + gh) * W + gw) << 2);
    };

    f32x4 acc[NCT][MT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < MT; ++r) acc[ct][r] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 xi[NI];
    float xh[NHL];
    f32x4 wv[WIT];
    auto load_chunk = [&](int q) __attribute__((always_inline)) {
        const int cg = q * CK + wave;               // channel staged by this wave (wave-uniform)
        if (cg < iC0 + iC1) {
            if (cg < iC0 && iup0) {
                const int soff = cg * V0 * 4;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const f32x2 t = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r0, off_i(j, true), soff, 0));
                    xi[j] = (f32x4){t.x, t.x, t.y, t.y};
                }
#pragma unroll
                for (int j = 0; j < NHL; ++j) xh[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r0, off_h(j, true), soff, 0));
            } else {
                const bool s0 = cg < iC0;
                const __amdgpu_buffer_rsrc_t r = s0 ? r0 : r1;
                const int soff = (s0 ? cg * V0 : (cg - iC0) * V) * 4;
#pragma unroll
                for (int j = 0; j < NI; ++j) xi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off_i(j, false), soff, 0));
#pragma unroll
                for (int j = 0; j < NHL; ++j) xh[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off_h(j, false), soff, 0));
            }
        } else {                                    // channel padding of the last chunk
#pragma unroll
            for (int j = 0; j < NI; ++j) xi[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NHL; ++j) xh[j] = 0.0f;
        }
        const __amdgpu_buffer_rsrc_t rw = vxm_rsrc(wp + ((size_t)g * Q + q) * WCHUNK, WCHUNK * 4u);
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            wv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tid + T8_THREADS * it) * 16, 0, 0));
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
        float* dst = Xs + wave * PS;
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (64 * j + lane < NROW * GPR) {       // interior slots
                *reinterpret_cast<f32x2*>(dst + ibase + RPL * j * RS) = (f32x2){xi[j].x, xi[j].y};
                *reinterpret_cast<f32x2*>(dst + ibase + RPL * j * RS + 2) = (f32x2){xi[j].z, xi[j].w};
            }
#pragma unroll
        for (int j = 0; j < NHL; ++j)
            if (64 * j + lane < NROW * 2) dst[hbase + 32 * j * RS] = xh[j];     // halo slots
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + T8_THREADS * it;
            if (i < WCHUNK / 4) reinterpret_cast<f32x4*>(Ws)[i] = wv[it];
        }
    };

    const int bbase = kq * PS + wave * HR * RS + n + 1;      // + 4 s planes, + (kd, r + kh) rows, + kw
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int q = 0; q < Q; ++q) {
        if (q + 1 < Q) load_chunk(q + 1);           // in flight under the MFMAs below
        // ---- 27 taps x KS k-steps x (NCT x 4) MFMAs, operands double-buffered in registers
        float a[2][NCT], bv[2][MT];
        auto fetch = [&](int st, float (&af)[NCT], float (&bf)[MT]) __attribute__((always_inline)) {
            const int t = st / KS, s = st % KS;
            const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) af[ct] = Ws[((t * KS + s) * NCT + ct) * 64 + lane];
#pragma unroll
            for (int r = 0; r < MT; ++r) bf[r] = Xs[bbase + s * 4 * PS + (kd * HR + r / HALVES + kh) * RS + (r % HALVES) * 16 + kw];
        };
        fetch(0, a[0], bv[0]);
#pragma unroll
        for (int st = 0; st < 27 * KS; ++st) {
            if (st + 1 < 27 * KS) fetch(st + 1, a[(st + 1) & 1], bv[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);       // keep the prefetch above the MFMAs (unpinned, the scheduler hoists every read)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < MT; ++r) acc[ct][r] = vxm_mfma16(a[st & 1][ct], bv[st & 1][r], acc[ct][r]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q + 1 < Q) {
            __syncthreads();                        // every wave is done reading chunk q
            store_chunk();
            __syncthreads();
        }
    }

    // ---- epilogue: bias + LeakyReLU (+ fused leaky_relu_backward mask), NCDHW store (branch-free buffer stores)
    const int d = d0 + wave, w = w0 + n;
    float bz[NCT][4];
    conv_load_bias<NCT>(bz, bias, Cout, g, kq);
    conv_epilogue_store<NCT, ROWS, HALVES>(acc, y + (size_t)b * y_bs, bz, mask ? mask + (size_t)b * mask_bs : nullptr, act_slope, mask_slope, Cout, g, kq,
                                   d < D && w < W, (d * H + h0) * W + w, h0, H, W, V);
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

    // ---- epilogue: bias + LeakyReLU; lane n of N-tile (ph, pw) is voxel (h0 + ph + 2 (n >> 3), w0 + 2 (n & 7) + pw): the two
    // pw tiles of a lane are neighbours along W and leave as ONE 8-byte buffer store (8 lanes = a 64-byte run); H and W
    // are even and the tile origin is even, so both rows / both columns of a lane are inside the volume or neither is.
    {
        const int d = d0 + wave, h = h0 + 2 * (n >> 3), w = w0 + 2 * (n & 7);
        const __amdgpu_buffer_rsrc_t ry = vxm_rsrc(y + (size_t)b * y_bs, (unsigned)Cout * (unsigned)V * 4u);
        const int cbase = g * NCT * 16 + kq * 4;
        const int nvalid = (d < D && h < H && w < W) ? Cout - cbase : 0;         // existing channel slots of this lane (explicit: see conv_epilogue_store)
        const int navail = Cout - g * NCT * 16;
        const int voff = (cbase * V + (d * H + h) * W + w) << 2;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float bz = bias ? bias[min((g * NCT + ct) * 16 + kq * 4 + j, Cout - 1)] : 0.0f;
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    float v0 = acc[ct][2 * ph][j] + bz, v1 = acc[ct][2 * ph + 1][j] + bz;
                    v0 = v0 > 0.0f ? v0 : v0 * act_slope;
                    v1 = v1 > 0.0f ? v1 : v1 * act_slope;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2){v0, v1}), ry, ct * 16 + j < nvalid ? voff : VXM_OOB,
                                                          (min(ct * 16 + j, navail - 1) * V + ph * W) << 2, 0);
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

// BLK (round 6): x is CHANNEL-BLOCKED [Cin / 8][voxel][8] (include/vxm_hip.h VXM_S3_IN0_BLOCKED; Cin % 8 == 0) -- the output of the last ConvBlock
// of the fused U-Net when the flow conv reads it.  A staging slot is then one voxel of the haloed tile (6 x 10 x 34 = 2040 slots, 8 per thread):
// an EVEN chunk fetches the voxel's 32 bytes with two 16-byte loads, writes channels 0 .. 3 into the planar LDS planes (four ds_write_b32, lanes
// on consecutive columns) and KEEPS channels 4 .. 7 in registers; the ODD chunk writes those and issues no load.  Sector requests per 8 channels
// and tile: 1020 (every byte of every sector used) against 1920 for the planar rows (2 interior + 2 halo sectors per row and channel).
template <int CO, bool BLK = false>
__global__ void __launch_bounds__(256) k_conv3d_k3_fewout(const float* __restrict__ x, long long x_bs, int Cin, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, long long y_bs, float act_slope,
                                                          int D, int H, int W) {
    VXM_DYN_SMEM(float, smem);
    float* const Xs = smem;                                  // [FO_CK][FO_PS]
    const int tid = threadIdx.x, tx = tid & 7, ty = (tid >> 3) & 7, tz = tid >> 6;
    const int ntw = (W + FO_TW - 1) / FO_TW, nth = (H + FO_TH - 1) / FO_TH;
    // block -> tile: block ids go to the 8 XCDs round-robin, so XCD x takes the tile range [nt x / 8, nt (x + 1) / 8) in block order -- the
    // blocks an XCD runs side by side are then spatial neighbours and find each other's halo sectors in its L2 (round 4: the counters showed
    // 1.72 GB fetched for a 440 MB input with tile = block id).  The grid is rounded up to a multiple of 8; surplus blocks leave at once.
    int t;
    {
        const int nt = ntw * nth * ((D + FO_TD - 1) / FO_TD), x = blockIdx.x & 7;
        t = (int)((long long)nt * x / 8) + (int)(blockIdx.x >> 3);
        if (t >= (int)((long long)nt * (x + 1) / 8)) return;
    }
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
    [[maybe_unused]] f32x4 held[BLK ? (FO_ROWS * (FO_TW + 2) + 255) / 256 : 1];      // channels 4 .. 7 of this thread's voxels, between an even chunk and the odd one

    const int Q = (Cin + FO_CK - 1) / FO_CK;
    for (int q = 0; q < Q; ++q) {
        if constexpr (BLK) {
            constexpr int NSL = FO_ROWS * (FO_TW + 2), NJ = (NSL + 255) / 256;
            if ((q & 1) == 0) {                              // block-uniform
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int slot = tid + 256 * j;
                    const int rr = slot / (FO_TW + 2), col = slot - rr * (FO_TW + 2);
                    const int gd = d0 - 1 + rr / (FO_TH + 2), gh = h0 - 1 + rr % (FO_TH + 2), gw = w0 - 1 + col;
                    const bool ok = slot < NSL && 4 * q < Cin && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
                    const int vo = ok ? ((q >> 1) * V + (gd * H + gh) * W + gw) << 5 : VXM_OOB;
                    const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0));
                    held[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 16, 0));
                    if (slot < NSL) {
                        float* dst = Xs + rr * FO_RS + col;
                        dst[0] = lo.x; dst[FO_PS] = lo.y; dst[2 * FO_PS] = lo.z; dst[3 * FO_PS] = lo.w;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int slot = tid + 256 * j;
                    const int rr = slot / (FO_TW + 2), col = slot - rr * (FO_TW + 2);
                    if (slot < NSL) {
                        float* dst = Xs + rr * FO_RS + col;
                        dst[0] = held[j].x; dst[FO_PS] = held[j].y; dst[2 * FO_PS] = held[j].z; dst[3 * FO_PS] = held[j].w;
                    }
                }
            }
        } else {
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

// ------------------------------------------------------------------------------------------
// forward kernel for layers with at most 4 input channels (first encoder block: 2 -> 16; backward-data of the flow conv:
// 3 -> 16): the reduction index K = 27 Cin is packed DENSELY into the k-steps of the MFMA (k = tap * Cin + ci), 14 / 21
// k-steps for Cin = 2 / 3 instead of the 27 that channel chunks padded to 4 cost.  A lane's (tap, ci) changes from
// k-step to k-step, so every lane keeps its own LDS offset per k-step in registers (the weights of a 16-channel
// output group are loop constants of the block and sit in LDS).  These layers are HBM-bound on the output (16
// channels written per voxel), and a tile is only 4 NS MFMAs per wave: blocks are persistent and the planes of the next
// tile are fetched under the MFMAs of the current one into the second of two LDS buffers.
// ------------------------------------------------------------------------------------------
constexpr int kpack_steps(int cin) { return (27 * cin + 3) / 4; }
constexpr int KP_ROWS = 4;          // tile rows of the 16-wide instance (plane stride 1200 = 16 mod 32: the two channels of a half-wave, Cin = 2, sit 16 banks apart)

// TWX: tile 8 x 4 x 16 or 8 x 2 x 32 (output in 128-byte runs when W % 32 == 0), as k_conv3d_k3_t8.
template <int CIN, int TWX = 16>
__global__ void __launch_bounds__(T8_THREADS, 4) k_conv3d_k3_kpack(ConvIn in, const float* __restrict__ wk, const float* __restrict__ bias,
                                                                 float* __restrict__ y, long long y_bs, int Cout, float act_slope,
                                                                 const float* __restrict__ mask, long long mask_bs, float mask_slope,
                                                                 int B, int D, int H, int W, int lay) {
    constexpr int HALVES = TWX / 16, ROWS = 4 / HALVES, MT = ROWS * HALVES, RS = TWX + 4, HR = ROWS + 2, NROW = (T8_TD + 2) * HR;
    constexpr int PS = NROW * RS + ((NROW * RS) % 32 == 16 ? 0 : 16), NS = kpack_steps(CIN), GPR = TWX / 4, RPL = 64 / GPR;
    constexpr int NI = (NROW * GPR + 63) / 64, NHL = (NROW * 2 + 63) / 64;
    __shared__ __attribute__((aligned(16))) float Xs2[2 * CIN * PS];      // double-buffered planes: one barrier per tile
    __shared__ float Wl[NS * 64];                                         // A fragments of this block's 16 output channels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, n = lane & 15;
    const float* const ix0 = in.x0; const float* const ix1 = in.x1;
    const int iC0 = in.C0, iC1 = in.C1;

    const int nw = (W + TWX - 1) / TWX, nh = (H + ROWS - 1) / ROWS, nd = (D + T8_TD - 1) / T8_TD;
    const int ntiles = B * nd * nh * nw;
    int tile, tile_end, tile_step;
    if (ntiles >= 64) {                             // XCD x = blockIdx.x % 8 walks its contiguous eighth of the tiles
        const int x = blockIdx.x & 7;
        tile_step = gridDim.x >> 3;
        tile = (int)((long long)ntiles * x / 8) + (blockIdx.x >> 3);
        tile_end = (int)((long long)ntiles * (x + 1) / 8);
    } else {
        tile = blockIdx.x; tile_end = ntiles; tile_step = gridDim.x;
    }
    if (tile >= tile_end) return;
    struct Org { int b, d0, h0, w0; };
    auto decode = [&](int t) __attribute__((always_inline)) -> Org {
        const int tw = t % nw; int tq = t / nw;
        const int th = tq % nh; tq /= nh;
        return Org{tq / nd, (tq % nd) * T8_TD, th * ROWS, tw * TWX};
    };
    Org cur = decode(tile);
    const int g = blockIdx.y;                       // output-channel group of 16
    const int V = D * H * W;

    // per-lane k-step table: LDS offset of (ci, tap) for this lane's k = 4 s + kq; the weights of the group go to LDS once
    int offs[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * s + kq;
        const bool valid = k < 27 * CIN;            // the padding k-values of the last step carry zero weights
        const int tap = valid ? k / CIN : 0, ci = valid ? k % CIN : 0;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        offs[s] = ci * PS + (kd * HR + kh) * RS + kw + wave * HR * RS + n + 1;
    }
    for (int i = tid; i < NS * 64; i += T8_THREADS) Wl[i] = wk[(size_t)g * NS * 64 + i];
    float bz[1][4];                                 // once per block, not per tile
    conv_load_bias<1>(bz, bias, Cout, g, kq);

    auto opaque = [](int v) __attribute__((always_inline)) -> int { asm volatile("" : "+v"(v)); return v; };
    f32x4 xi[NI];
    float xh[NHL];
    auto load_x = [&](const Org& o) __attribute__((always_inline)) {
        if (wave >= CIN) return;                    // wave c stages input channel c
        const int ln = opaque(lane);                // (keeps the address arithmetic inside the tile loop: registers)
        const bool s0 = wave < iC0;
        const float* base = s0 ? ix0 + (size_t)o.b * in.bs0 + (size_t)wave * V : ix1 + (size_t)o.b * in.bs1 + (size_t)(wave - iC0) * V;
        const __amdgpu_buffer_rsrc_t r = vxm_rsrc(base, (unsigned)V * 4u);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int rr = RPL * j + ln / GPR;
            const int gd = o.d0 - 1 + rr / HR, gh = o.h0 - 1 + rr % HR, gw = o.w0 + 4 * (ln % GPR);
            const bool ok = rr < NROW && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && gw < W;
            xi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, ok ? ((gd * H + gh) * W + gw) << 2 : VXM_OOB, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < NHL; ++j) {
            const int rr = 32 * j + (ln >> 1);
            const int gd = o.d0 - 1 + rr / HR, gh = o.h0 - 1 + rr % HR, gw = (ln & 1) ? o.w0 + TWX : o.w0 - 1;
            const bool ok = rr < NROW && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            xh[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, ok ? ((gd * H + gh) * W + gw) << 2 : VXM_OOB, 0, 0));
        }
    };
    auto store_x = [&](int buf) __attribute__((always_inline)) {
        if (wave >= CIN) return;
        float* dst = Xs2 + buf * (CIN * PS) + wave * PS;
        const int ibase = (lane / GPR) * RS + 2 + 4 * (lane % GPR), hbase = (lane >> 1) * RS + ((lane & 1) ? TWX + 2 : 1);
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (64 * j + lane < NROW * GPR) {
                *reinterpret_cast<f32x2*>(dst + ibase + RPL * j * RS) = (f32x2){xi[j].x, xi[j].y};
                *reinterpret_cast<f32x2*>(dst + ibase + RPL * j * RS + 2) = (f32x2){xi[j].z, xi[j].w};
            }
#pragma unroll
        for (int j = 0; j < NHL; ++j)
            if (64 * j + lane < NROW * 2) dst[hbase + 32 * j * RS] = xh[j];
    };

    f32x4 acc[MT];
#pragma unroll
    for (int r = 0; r < MT; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_x(cur);
    store_x(0);
    __syncthreads();
    int buf = 0;
    for (;;) {
        const int tile_next = tile + tile_step;
        const bool has_next = tile_next < tile_end;
        if (has_next) load_x(decode(tile_next));    // in flight under the MFMAs below
        const float* Xs = Xs2 + buf * (CIN * PS);
        float bv[2][MT], wa[2];
        wa[0] = Wl[lane];
#pragma unroll
        for (int r = 0; r < MT; ++r) bv[0][r] = Xs[offs[0] + (r / HALVES) * RS + (r % HALVES) * 16];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) {
                wa[(s + 1) & 1] = Wl[(s + 1) * 64 + lane];
#pragma unroll
                for (int r = 0; r < MT; ++r) bv[(s + 1) & 1][r] = Xs[offs[s + 1] + (r / HALVES) * RS + (r % HALVES) * 16];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < MT; ++r) acc[r] = vxm_mfma16(wa[s & 1], bv[s & 1][r], acc[r]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // The next tile's planes go to the other LDS buffer BEFORE this tile's output stores are issued: vmcnt counts loads
        // and stores alike, so waiting for the prefetch after the stores would wait for the stores' write acknowledgements.
        if (has_next) store_x(buf ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue: bias + LeakyReLU (+ fused leaky_relu_backward mask), NCDHW store (branch-free buffer stores)
        {
            const int ln = opaque(lane), ekq = ln >> 4, en = ln & 15;
            const int d = cur.d0 + wave, w = cur.w0 + en;
            f32x4 (&acc1)[1][MT] = *reinterpret_cast<f32x4 (*)[1][MT]>(&acc);
            if (lay & VXM_S3_OUT_BLOCKED)            // (round 6) channel-blocked y [Cout / 8][voxel][8]; the mask in the same layout, or a sign tensor (VXM_S3_MASK_SIGNS)
                conv_epilogue_store_blocked<1, ROWS, HALVES>(acc1, y + (size_t)cur.b * y_bs, bz,
                                                             mask ? ((lay & VXM_S3_MASK_SIGNS) ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(mask) + (size_t)cur.b * mask_bs)
                                                                                               : mask + (size_t)cur.b * mask_bs) : nullptr,
                                                             act_slope, mask_slope, Cout, g, ekq, d < D && w < W, (d * H + cur.h0) * W + w, cur.h0, H, W, V, lay);
            else
            conv_epilogue_store<1, ROWS, HALVES>(acc1, y + (size_t)cur.b * y_bs, bz, mask ? mask + (size_t)cur.b * mask_bs : nullptr, act_slope, mask_slope,
                                         Cout, g, ekq, d < D && w < W, (d * H + cur.h0) * W + w, cur.h0, H, W, V);
#pragma unroll
            for (int r = 0; r < MT; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (!has_next) break;
        __syncthreads();                            // the other buffer is complete, and every wave is done reading this one
        buf ^= 1;
        tile = tile_next;
        cur = decode(tile);
    }
}

// w: [Cw_out][Cw_in][27] (reference layout) -> [G][NS][64]: lane (kq, n) of k-step s holds the weight of output channel
// 16 g + n for k = 4 s + kq = tap * Cin_p + ci (zero beyond 27 Cin_p); flip = the adjoint operator (backward-data)
__global__ void __launch_bounds__(256) k_pack_weights_kpack(const float* __restrict__ w, float* __restrict__ wk, int Cw_in, int ci_lo, int flip, int Cin_p,
                                                            int Cout_p, int NS, size_t elems) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    const int lane = (int)(i % 64);
    const int s = (int)((i / 64) % NS), g = (int)(i / 64 / NS);
    const int co = g * 16 + (lane & 15), k = 4 * s + (lane >> 4);
    float v = 0.0f;
    if (co < Cout_p && k < 27 * Cin_p) {
        const int tap = k / Cin_p, ci = k % Cin_p;
        v = flip ? w[((size_t)ci * Cw_in + ci_lo + co) * 27 + (26 - tap)] : w[((size_t)co * Cw_in + ci_lo + ci) * 27 + tap];
    }
    wk[i] = v;
}

struct ConvCfg { int CK, NCT, Q, G; size_t elems; };
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
// layers with at most 4 input channels carry a second copy of their weights in the dense-K order of k_conv3d_k3_kpack
// VXM_T8_TILE32=0 keeps the 16-wide tiles everywhere (developer A/B switch)
bool t8_tile32() {
    static const bool on = [] { const char* e = getenv("VXM_T8_TILE32"); return !(e && e[0] == '0'); }();
    return on;
}
size_t kpack_elems(int Cin, int Cout) { return Cin <= 4 ? (size_t)((Cout + 15) / 16) * kpack_steps(Cin) * 64 : 0; }
int kpack_blocks() {
    static const int n = [] {
        const char* e = getenv("VXM_KPACK_BLOCKS");     // developer experiments
        const int v = e ? atoi(e) : 1024;
        return v >= 8 ? v / 8 * 8 : 1024;
    }();
    return n;
}
bool kpack_ok(int C0, int C1, int x0_up, const float* x0, int64_t bs0, const float* x1, int64_t bs1, const float* wpacked, int B, int D, int H,
              int W) {
    const long long tiles8 = (long long)B * ((D + T8_TD - 1) / T8_TD) * ((H + KP_ROWS - 1) / KP_ROWS) * ((W + TW - 1) / TW);
    return C0 + C1 <= 4 && !x0_up && (W & 3) == 0 && al16(x0) && (C1 == 0 || al16(x1)) && (bs0 & 3) == 0 && (bs1 & 3) == 0 &&
           (((long long)D * H * W) & 3) == 0 && al16(wpacked) && tiles8 >= wide_min_tiles() && tiles8 < (1ll << 30) && !bw_force_generic();
}

// k_conv3d_k3_sm replaces k_conv3d_k3 where that kernel's grid (4 x 4 x 16 tiles x output-channel groups) has at most this many blocks, i.e.
// leaves more than half of the 256 CUs without a block; VXM_CONV_SMALL_MAX_BLOCKS overrides (0: never -- the tests' A/B switch)
long long small_max_blocks() {
    static const long long v = [] { const char* e = getenv("VXM_CONV_SMALL_MAX_BLOCKS"); return e ? atoll(e) : 128ll; }();
    return v;
}

bool small_ok(const ConvCfg& c, long long tiles) { return c.CK == 8 && c.Q >= 4 && tiles * c.G <= small_max_blocks() && !bw_force_generic(); }

// rows per wave of the 8-wave kernel
int fwd_wide_rows(const ConvCfg& c) { (void)c; return 4; }     // 6 rows with one output tile measured 4 % slower than 4
bool fwd_wide_ok(const ConvCfg& c, const float* x0, int64_t bs0, const float* x1, int C1, int64_t bs1, const float* wpacked,
                 int B, int D, int H, int W) {
    const int rows = fwd_wide_rows(c);
    const long long tiles8 = (long long)B * ((D + T8_TD - 1) / T8_TD) * ((H + rows - 1) / rows) * ((W + TW - 1) / TW);
    return c.CK == 8 && c.NCT <= 2 && (W & 3) == 0 && al16(x0) && (C1 == 0 || al16(x1)) && (bs0 & 3) == 0 && (bs1 & 3) == 0 &&
           al16(wpacked) && tiles8 >= wide_min_tiles() && tiles8 < (1ll << 30) && !bw_force_generic();
}

// w: [Cw_out][Cw_in][27] (reference layout).  Packed operator has Cin_p inputs / Cout_p outputs.
__global__ void __launch_bounds__(256) k_pack_weights(const float* __restrict__ w, float* __restrict__ wp, int Cw_in, int ci_lo,
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
    if (co < Cout_p && ci < Cin_p) v = flip ? w[((size_t)ci * Cw_in + ci_lo + co) * 27 + (26 - t)] : w[((size_t)co * Cw_in + ci_lo + ci) * 27 + t];
    wp[i] = v;
}

}  // namespace

extern "C" {

size_t vxm_conv3d_k3_packed_elems(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return conv_cfg(Cin, Cout).elems + kpack_elems(Cin, Cout);
}

int vxm_conv3d_k3_pack_weights_range(const float* w, float* wpacked, int Cw_in, int Cout, int ci_lo, int ci_n, int transpose_flip, void* stream) {
    VXM_REQUIRE(w && wpacked, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_pack_weights: null pointer");
    VXM_REQUIRE(Cw_in > 0 && Cout > 0 && ci_lo >= 0 && ci_n > 0 && ci_lo + ci_n <= Cw_in, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_pack_weights: channel range [%d, %d) of %d, Cout=%d", ci_lo, ci_lo + ci_n, Cw_in, Cout);
    const int cin_p = transpose_flip ? Cout : ci_n, cout_p = transpose_flip ? ci_n : Cout;
    const ConvCfg c = conv_cfg(cin_p, cout_p);
    hipLaunchKernelGGL(k_pack_weights, dim3(vxm_blocks((long long)c.elems, 256)), dim3(256), 0, VXM_STREAM(stream), w, wpacked,
                       Cw_in, ci_lo, transpose_flip, cin_p, cout_p, c.CK, c.NCT, c.Q, c.elems);
    if (const size_t ke = kpack_elems(cin_p, cout_p))
        hipLaunchKernelGGL(k_pack_weights_kpack, dim3(vxm_blocks((long long)ke, 256)), dim3(256), 0, VXM_STREAM(stream), w, wpacked + c.elems,
                           Cw_in, ci_lo, transpose_flip, cin_p, cout_p, kpack_steps(cin_p), ke);
    return vxm_check_launch("vxm_conv3d_k3_pack_weights");
}

int vxm_conv3d_k3_pack_weights(const float* w, float* wpacked, int Cin, int Cout, int transpose_flip, void* stream) {
    return vxm_conv3d_k3_pack_weights_range(w, wpacked, Cin, Cout, 0, Cin, transpose_flip, stream);
}

int vxm_conv3d_k3_fwd(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                      const float* wpacked, const float* bias, float* y, int64_t y_bstride, int Cout, float act_slope,
                      const float* mask_src, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W, void* stream) {
    return vxm_conv3d_k3_fwd_layout(x0, C0, x0_bstride, x0_up, x1, C1, x1_bstride, wpacked, bias, y, y_bstride, Cout, act_slope, mask_src, mask_bstride,
                                    mask_slope, B, D, H, W, 0, stream);
}

int vxm_conv3d_k3_fwd_layout_ok(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* wpacked, int Cout,
                                int B, int D, int H, int W) {
    return (C0 > 0 && C1 >= 0 && Cout > 0 && Cout % 8 == 0 && kpack_ok(C0, C1, 0, x0, x0_bstride, x1, x1_bstride, wpacked, B, D, H, W)) ? 1 : 0;
}

int vxm_conv3d_k3_fwd_layout(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                             const float* wpacked, const float* bias, float* y, int64_t y_bstride, int Cout, float act_slope,
                             const float* mask_src, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W, int layout, void* stream) {
    if (int e = check_conv("vxm_conv3d_k3_fwd", C0, C1, x0_up, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(x0 && wpacked && y && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_fwd: null pointer");
    VXM_REQUIRE(layout == 0 || ((layout & ~(VXM_S3_OUT_BLOCKED | VXM_S3_MASK_SIGNS)) == 0 && (layout & VXM_S3_OUT_BLOCKED) && !x0_up &&
                                (!(layout & VXM_S3_MASK_SIGNS) || mask_src) &&
                                vxm_conv3d_k3_fwd_layout_ok(x0, C0, x0_bstride, x1, C1, x1_bstride, wpacked, Cout, B, D, H, W)),
                VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_fwd_layout: layout flags 0x%x need VXM_S3_OUT_BLOCKED (+ VXM_S3_MASK_SIGNS with a sign tensor), Cout %% 8 == 0 "
                "and a launch the few-input-channel kernel takes (vxm_conv3d_k3_fwd_layout_ok); %d + %d -> %d channels", layout, C0, C1, Cout);
    const ConvCfg c = conv_cfg(C0 + C1, Cout);
    ConvIn in{x0, x1, (long long)x0_bstride, (long long)x1_bstride, C0, C1, x0_up};
    if (kpack_ok(C0, C1, x0_up, x0, x0_bstride, x1, x1_bstride, wpacked, B, D, H, W)) {      // few input channels: dense-K MFMA kernel
        const bool wide32 = (W & 31) == 0 && t8_tile32();
        const int trows = wide32 ? 2 : KP_ROWS, tw = wide32 ? 32 : TW;
        const long long tiles8 = (long long)B * ((D + T8_TD - 1) / T8_TD) * ((H + trows - 1) / trows) * ((W + tw - 1) / tw);
        const long long nb = tiles8 < (long long)kpack_blocks() ? (tiles8 + 7) / 8 * 8 : (long long)kpack_blocks();
        const dim3 gridk((unsigned)nb, (Cout + 15) / 16);
        const float* wk = wpacked + c.elems;
#define LAUNCHK(...) hipLaunchKernelGGL((k_conv3d_k3_kpack<__VA_ARGS__>), gridk, dim3(T8_THREADS), 0, VXM_STREAM(stream), in, wk, bias, y, (long long)y_bstride, \
        Cout, act_slope, mask_src, (long long)mask_bstride, mask_slope, B, D, H, W, layout)
        switch ((C0 + C1) * 2 + (wide32 ? 1 : 0)) {
            case 2: LAUNCHK(1); break;
            case 3: LAUNCHK(1, 32); break;
            case 4: LAUNCHK(2); break;
            case 5: LAUNCHK(2, 32); break;
            case 6: LAUNCHK(3); break;
            case 7: LAUNCHK(3, 32); break;
            case 8: LAUNCHK(4); break;
            default: LAUNCHK(4, 32); break;
        }
#undef LAUNCHK
        return vxm_check_launch("vxm_conv3d_k3_fwd");
    }
    // large layers: the 8-wave wide-load kernel (needs 4-float groups that neither straddle row ends nor break alignment)
    {
        const int rows = fwd_wide_rows(c);
        const long long tiles8 = (long long)B * ((D + T8_TD - 1) / T8_TD) * ((H + rows - 1) / rows) * ((W + TW - 1) / TW);
        const bool vec = fwd_wide_ok(c, x0, x0_bstride, x1, C1, x1_bstride, wpacked, B, D, H, W);
        if (vec) {
            static bool opt_in = false;
            if (!opt_in) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_t8<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_t8<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_t8<2, 2, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_t8<1, 2, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
                opt_in = true;
            }
            // 32-voxel-wide tiles (8 x 2 x 32: the output leaves in 128-byte runs) when W allows, 16-wide (8 x 4 x 16) otherwise
            const bool wide32 = (W & 31) == 0 && t8_tile32();
            const int trows = wide32 ? 2 : rows, tw = wide32 ? 32 : TW;
            const long long ntile = (long long)B * ((D + T8_TD - 1) / T8_TD) * ((H + trows - 1) / trows) * ((W + tw - 1) / tw);
            const dim3 grid8((unsigned)((ntile + 7) / 8 * 8), c.G);
            const int plane8 = (T8_TD + 2) * (trows + 2) * (tw + 4), ps8 = plane8 + (plane8 % 32 == 16 ? 0 : 16);
            const size_t lds8 = sizeof(float) * ((size_t)8 * ps8 + 27 * 2 * c.NCT * 64);
#define LAUNCH8(...) hipLaunchKernelGGL((k_conv3d_k3_t8<__VA_ARGS__>), grid8, dim3(T8_THREADS), lds8, VXM_STREAM(stream), in, wpacked, bias, y, \
        (long long)y_bstride, Cout, act_slope, mask_src, (long long)mask_bstride, mask_slope, B, D, H, W, c.Q)
            if (wide32) {
                if (c.NCT == 1) LAUNCH8(1, 2, 32);
                else LAUNCH8(2, 2, 32);
            } else {
                if (c.NCT == 1) LAUNCH8(1, 4);
                else LAUNCH8(2, 4);
            }
#undef LAUNCH8
            return vxm_check_launch("vxm_conv3d_k3_fwd");
        }
    }
    const long long tiles = (long long)B * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_fwd: too many tiles");
    // small volumes: input channels split over the waves of a block, 2 x 4 x 16 tiles, one 16-channel output tile per block
    if (small_ok(c, tiles)) {
        const long long tsm = (long long)B * ((D + SM_TD - 1) / SM_TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
        const dim3 gsm((unsigned)tsm, (Cout + 15) / 16);
        static bool opt_in_sm = false;
        if (!opt_in_sm) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3d_k3_sm<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * SM_XW * 4);
            opt_in_sm = true;
        }
#define LAUNCHS(NW_) hipLaunchKernelGGL((k_conv3d_k3_sm<NW_>), gsm, dim3(64 * NW_), sizeof(float) * NW_ * SM_XW, VXM_STREAM(stream), in, wpacked, bias, y, \
        (long long)y_bstride, Cout, act_slope, mask_src, (long long)mask_bstride, mask_slope, D, H, W, c.Q, c.NCT)
        if (c.Q >= 8) LAUNCHS(8);
        else LAUNCHS(4);
#undef LAUNCHS
        return vxm_check_launch("vxm_conv3d_k3_fwd");
    }
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
    if (kpack_ok(C0, C1, 0, x0, x0_bstride, x1, x1_bstride, wpacked, B, D, H, W)) return 200 + C0 + C1;
    if (fwd_wide_ok(c, x0, x0_bstride, x1, C1, x1_bstride, wpacked, B, D, H, W)) return 100 + 10 * c.CK + c.NCT;
    const long long tiles = (long long)B * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    if (small_ok(c, tiles)) return 300 + (c.Q >= 8 ? 8 : 4);
    return 10 * c.CK + c.NCT;
}

int vxm_conv3d_k3_up_ok(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, float* y,
                        int Cout, int B, int D, int H, int W) {
    if (C0 <= 0 || C1 < 0 || Cout <= 0) return 0;
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
           (long long)Cout * D * H * W < (1ll << 29) && tiles >= dlow_min_tiles() && tiles < (1ll << 30) && !bw_force_generic();
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
    return vxm_conv3d_k3_fewout_fwd_layout(x, Cin, x_bstride, w, bias, y, y_bstride, Cout, act_slope, B, D, H, W, 0, stream);
}

int vxm_conv3d_k3_fewout_fwd_layout(const float* x, int Cin, int64_t x_bstride, const float* w, const float* bias, float* y, int64_t y_bstride,
                                    int Cout, float act_slope, int B, int D, int H, int W, int layout, void* stream) {
    if (int e = check_conv("vxm_conv3d_k3_fewout_fwd", Cin, 0, 0, Cout, B, D, H, W)) return e;
    VXM_REQUIRE((layout & ~VXM_S3_IN0_BLOCKED) == 0 && (layout == 0 || Cin % 8 == 0), VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_fewout_fwd_layout: layout flags 0x%x (VXM_S3_IN0_BLOCKED with Cin %% 8 == 0 only; Cin = %d)", layout, Cin);
    VXM_REQUIRE(x && w && y, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_fewout_fwd: null pointer");
    VXM_REQUIRE(vxm_conv3d_k3_fewout_ok(x, x_bstride, y, y_bstride, Cin, Cout, W), VXM_ERR_UNSUPPORTED,
                "vxm_conv3d_k3_fewout_fwd: needs Cout <= 4, W %% 4 == 0 and 16-byte aligned tensors (use vxm_conv3d_k3_fwd)");
    VXM_REQUIRE(B <= 65535 && Cin <= 512, VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_fewout_fwd: B <= 65535, Cin <= 512");
    const long long tiles = (long long)((W + FO_TW - 1) / FO_TW) * ((H + FO_TH - 1) / FO_TH) * ((D + FO_TD - 1) / FO_TD);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_fewout_fwd: too many tiles");
    const size_t lds = sizeof(float) * (size_t)FO_CK * FO_PS;
    const dim3 grid((unsigned)(8 * ((tiles + 7) / 8)), B);          // XCD-contiguous tile ranges: see the kernel
#define FO_LAUNCH(...) hipLaunchKernelGGL((k_conv3d_k3_fewout<__VA_ARGS__>), grid, dim3(256), lds, VXM_STREAM(stream), x, (long long)x_bstride, Cin, w, bias, y, \
        (long long)y_bstride, act_slope, D, H, W)
    switch (Cout + (layout ? 4 : 0)) {
        case 1: FO_LAUNCH(1); break;
        case 2: FO_LAUNCH(2); break;
        case 3: FO_LAUNCH(3); break;
        case 4: FO_LAUNCH(4); break;
        case 5: FO_LAUNCH(1, true); break;
        case 6: FO_LAUNCH(2, true); break;
        case 7: FO_LAUNCH(3, true); break;
        default: FO_LAUNCH(4, true); break;
    }
#undef FO_LAUNCH
    return vxm_check_launch("vxm_conv3d_k3_fewout_fwd");
}

}  // extern "C"
