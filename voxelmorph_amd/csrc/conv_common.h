// Shared pieces of the conv kernels (conv_fwd.hip, conv_bwd_weight.hip): tile geometry, the virtual-concat input
// descriptor, buffer-descriptor helpers, kernel-selection switches and argument checks.
#ifndef VXM_CONV_COMMON_H
#define VXM_CONV_COMMON_H
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
__device__ __forceinline__ void vxm_lds_dma4(__amdgpu_buffer_rsrc_t r, float* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 4, voff, soff, 0, 0);      // lane l -> LDS base + 4 l
}
__device__ __forceinline__ void vxm_lds_dma16(__amdgpu_buffer_rsrc_t r, float* lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, soff, 0, 0);     // lane l -> LDS base + 16 l
}



// VXM_CONV_GENERIC=1 routes every conv launch through the generic kernels (any W / alignment; LDS-DMA backward-
// weight), so that the parity tests can exercise them on shapes the wide-load kernels would otherwise take.
bool bw_force_generic() {
    static const bool f = [] { const char* e = getenv("VXM_CONV_GENERIC"); return e && e[0] == '1'; }();
    return f;
}
// The 8-wave forward kernel is used from this many 8x4x16 tiles up (below, its 512-voxel tiles leave CUs idle);
// VXM_CONV_WIDE_MIN_TILES overrides the threshold so that the parity tests can run it on small volumes.
// Smallest tile count for which the 8-wave kernels replace the generic 4-wave kernel (measured on the default U-Net: the
// 40x48x56 level, 210 tiles, is still faster on them; below that the generic kernel's smaller tiles win), and the same for the
// low-resolution backward-data kernel (counted in its own tiles; it only pays from the 80x96x112 level up).
// VXM_CONV_WIDE_MIN_TILES overrides both (the tests force every kernel onto small volumes with it).
[[maybe_unused]] long long wide_min_tiles() {
    static const long long v = [] { const char* e = getenv("VXM_CONV_WIDE_MIN_TILES"); return e ? atoll(e) : 128ll; }();
    return v;
}
[[maybe_unused]] long long dlow_min_tiles() {
    static const long long v = [] { const char* e = getenv("VXM_CONV_WIDE_MIN_TILES"); return e ? (atoll(e) + 3) / 4 : 256ll; }();
    return v;
}

static bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

// bias of the 4 output channels a lane holds per 16-channel tile (MFMA D layout)
template <int NCT>
__device__ __forceinline__ void conv_load_bias(float (&bz)[NCT][4], const float* __restrict__ bias, int Cout, int g, int kq) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) bz[ct][j] = bias ? bias[min((g * NCT + ct) * 16 + kq * 4 + j, Cout - 1)] : 0.0f;
}

// ---- branch-free epilogue of the MFMA kernels (D layout: lane (kq, n) holds, in acc[ct][r][j], output channel
// 16 (g NCT + ct) + 4 kq + j at voxel (d, h0 + r, w) with w = w0 + n): bias + LeakyReLU (+ the fused leaky_relu_backward
// mask), NCDHW store.  Buffer stores / loads through a descriptor whose range is the tensor of the sample: the lane part
// of the address is ONE 32-bit offset per tile, channel and row steps are wave-uniform scalar offsets, and a lane whose
// voxel is outside the volume or whose channel is >= Cout gets an out-of-range offset, which the hardware drops -- no
// per-store branch, no 64-bit address arithmetic.  (The channel test is explicit, one v_cndmask per store, and the scalar
// offset is clamped into the tensor: the range check of a raw buffer compares the per-lane offset with num_records - soffset,
// which must not be relied on once the scalar offset alone exceeds the range.)  Rows beyond H are skipped by a wave-uniform test.
// HALVES = 16-voxel MFMA tiles side by side along W (acc index m = row * HALVES + half; needs W % (16 HALVES) == 0).
template <int NCT, int ROWS, int HALVES = 1, int AUX = 0 /* cache policy bits of the stores (2: non-temporal) */>
__device__ __forceinline__ void conv_epilogue_store(f32x4 (&acc)[NCT][ROWS * HALVES], float* __restrict__ yb /* y + b * y_bs */, const float (&bz)[NCT][4],
                                                    const float* __restrict__ maskb /* mask + b * mask_bs or null */, float act_slope,
                                                    float mask_slope, int Cout, int g, int kq, bool vox_ok, int vox /* (d H + h0) W + w */,
                                                    int h0, int H, int W, int V) {
    const __amdgpu_buffer_rsrc_t ry = vxm_rsrc(yb, (unsigned)Cout * (unsigned)V * 4u);
    const __amdgpu_buffer_rsrc_t rm = vxm_rsrc(maskb ? maskb : yb, (unsigned)Cout * (unsigned)V * 4u);
    const int cbase = g * NCT * 16 + kq * 4;       // first of this lane's channels; slot (ct, j) is channel cbase + 16 ct + j
    const int nvalid = vox_ok ? Cout - cbase : 0;   // channel slots of this lane that exist (<= 0: none)
    const int navail = Cout - g * NCT * 16;         // channel slots of the whole group (wave-uniform, >= 1)
    const int voff = (cbase * V + vox) << 2;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        constexpr int MT = ROWS * HALVES;
        float mk[4][MT];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < MT; ++r) mk[j][r] = 1.0f;
        if (maskb) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < MT; ++r)
                    mk[j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rm, ct * 16 + j < nvalid ? voff : VXM_OOB,
                                                   (min(ct * 16 + j, navail - 1) * V + (r / HALVES) * W + (r % HALVES) * 16) << 2, 0));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < MT; ++r) mk[j][r] = vxm_lrelu_grad(mk[j][r], mask_slope);
        }
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
            if (h0 + row < H) {                      // wave-uniform
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int half = 0; half < HALVES; ++half) {      // the halves of a row back to back: one 128-byte run per channel
                        const int r = row * HALVES + half;
                        float v = acc[ct][r][j] + bz[ct][j];
                        v = (v > 0.0f ? v : v * act_slope) * mk[j][r];
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, ct * 16 + j < nvalid ? voff : VXM_OOB,
                                                              (min(ct * 16 + j, navail - 1) * V + row * W + half * 16) << 2, AUX);
                    }
            }
        }
    }
}

// ---- the same epilogue for a CHANNEL-BLOCKED output tensor [Cout / 8][voxel][8] (and mask, same layout): the 4 channels a lane holds per
// 16-channel tile are 16 contiguous bytes of a voxel's 32-byte group -- one 16-byte store (and mask load) per row, 512 contiguous bytes
// per pair of lane groups.  Cout % 8 == 0 (so a lane's 4 channels exist or not together).  Same arithmetic as conv_epilogue_store.
// Round 6, SIGN tensors (include/vxm_hip.h VXM_S3_MASK_SIGNS / VXM_S3_OUT_SIGNS): LeakyReLU' needs one bit of the activation it is taken at, and
// the fp32 mask was 35 % of what rem1's backward-data launch moves (0.12 of its 0.68 ms, tools/mask_ab.py).  A sign tensor is [Cout / 4][voxel]
// bytes, bit j of byte (q, v) = (y[4 q + j][v] > 0): exactly the four channels a lane holds, so a forward epilogue writes one byte per row
// beside its 16-byte store (lay & OUT_SIGNS: `maskb` is that OUTPUT) and a backward-data epilogue reads one (lay & MASK_SIGNS).  Same products.
// HALVES = 16-voxel MFMA tiles side by side along W (acc index m = row * HALVES + half), as in conv_epilogue_store.
template <int NCT, int ROWS, int HALVES = 1>
__device__ __forceinline__ void conv_epilogue_store_blocked(f32x4 (&acc)[NCT][ROWS * HALVES], float* __restrict__ yb, const float (&bz)[NCT][4],
                                                            const float* __restrict__ maskb, float act_slope, float mask_slope, int Cout, int g, int kq,
                                                            bool vox_ok, int vox, int h0, int H, int W, int V, int lay = 0) {
    const bool sg_out = maskb != nullptr && (lay & VXM_S3_OUT_SIGNS) != 0, sg_in = maskb != nullptr && (lay & VXM_S3_MASK_SIGNS) != 0;      // wave-uniform
    const __amdgpu_buffer_rsrc_t ry = vxm_rsrc(yb, (unsigned)Cout * (unsigned)V * 4u);
    const __amdgpu_buffer_rsrc_t rm = vxm_rsrc(maskb ? maskb : yb, (sg_out || sg_in) ? (unsigned)(Cout >> 2) * (unsigned)V : (unsigned)Cout * (unsigned)V * 4u);
    const int cbase = g * NCT * 16 + kq * 4;
    const int voff = (((cbase >> 3) * V + vox) << 5) + ((kq & 1) << 4);
    const int boff = (cbase >> 2) * V + vox;                     // byte of this lane's four channels in a sign tensor
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const bool ok = vox_ok && cbase + 16 * ct < Cout;
        const int soff_ct = ((g * NCT + ct) * 16 < Cout ? 2 * ct : 0) * V;      // wave-uniform, kept inside the tensor (see conv_epilogue_store); a tile beyond Cout is dropped by `ok`
        constexpr int MT = ROWS * HALVES;
        f32x4 mk[MT];
#pragma unroll
        for (int r = 0; r < MT; ++r) mk[r] = (f32x4){1.0f, 1.0f, 1.0f, 1.0f};
        if (sg_in) {
            unsigned sb[MT];
#pragma unroll
            for (int r = 0; r < MT; ++r) sb[r] = __builtin_amdgcn_raw_buffer_load_b8(rm, ok ? boff : VXM_OOB, 2 * soff_ct + (r / HALVES) * W + (r % HALVES) * 16, 0);
#pragma unroll
            for (int r = 0; r < MT; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) mk[r][j] = ((sb[r] >> j) & 1u) ? 1.0f : mask_slope;
        } else if (maskb && !sg_out) {
#pragma unroll
            for (int r = 0; r < MT; ++r)
                mk[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, ok ? voff : VXM_OOB, (soff_ct + (r / HALVES) * W + (r % HALVES) * 16) << 5, 0));
#pragma unroll
            for (int r = 0; r < MT; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) mk[r][j] = vxm_lrelu_grad(mk[r][j], mask_slope);
        }
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
            if (h0 + row < H) {                      // wave-uniform
#pragma unroll
                for (int half = 0; half < HALVES; ++half) {
                    const int r = row * HALVES + half, vrel = row * W + half * 16;
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acc[ct][r][j] + bz[ct][j];
                        o[j] = (v > 0.0f ? v : v * act_slope) * mk[r][j];
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry, ok ? voff : VXM_OOB, (soff_ct + vrel) << 5, 0);
                    if (sg_out) {
                        const unsigned nib = (o[0] > 0.0f ? 1u : 0u) | (o[1] > 0.0f ? 2u : 0u) | (o[2] > 0.0f ? 4u : 0u) | (o[3] > 0.0f ? 8u : 0u);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)nib, rm, ok ? boff : VXM_OOB, 2 * soff_ct + vrel, 0);
                    }
                }
            }
        }
    }
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
#endif
