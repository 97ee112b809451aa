// SpatialTransformer / VecInt / ResizeTransform kernels for gfx950 (HBM-bound, coalesced along W).
//
// Reference op chains replaced (paths relative to the reference root):
//   voxelmorph/torch/layers.py:30-48  SpatialTransformer.forward  (16 ATen launches -> 1 kernel)
//   voxelmorph/torch/layers.py:64-68  VecInt.forward              (1 + 7*17 launches -> 7 kernels)
//   voxelmorph/torch/layers.py:85-97  ResizeTransform.forward     (2 launches -> 1 kernel)
// Layout: NCDHW fp32; one thread per output voxel, the 64 lanes of a wave walk W so the flow /
// output planes are read and written as 256-byte rows; the 8-corner gathers hit L1/L2 because
// neighbouring lanes sample neighbouring source voxels for the smooth fields of this path.
#include "vxm_common.h"
#include "vxm_device.h"
#include <type_traits>

namespace {

// the 8 corners in ATen's order k = 4 dz + 2 dy + dx: plane offsets, trilinear weights (wz wy) wx, validity
struct Corners8 {
    int idx[8];
    float w[8];
    bool ok[8];
    float wz[2], wy[2], wx[2];
};
__device__ __forceinline__ Corners8 corners8(float z, float y, float x, int D, int H, int W) {
    const AxisTaps az = axis_corners(z, D), ay = axis_corners(y, H), ax = axis_corners(x, W);
    Corners8 c;
    c.wz[0] = az.w0; c.wz[1] = az.w1; c.wy[0] = ay.w0; c.wy[1] = ay.w1; c.wx[0] = ax.w0; c.wx[1] = ax.w1;
    const int iz[2] = {az.i0, az.i1}, iy[2] = {ay.i0, ay.i1}, ix[2] = {ax.i0, ax.i1};
    const bool oz[2] = {az.ok0, az.ok1}, oy[2] = {ay.ok0, ay.ok1}, ox[2] = {ax.ok0, ax.ok1};
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int row = (iz[dz] * H + iy[dy]) * W;
            const float wzy = c.wz[dz] * c.wy[dy];
            const bool ozy = oz[dz] && oy[dy];
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int k = 4 * dz + 2 * dy + dx;
                c.idx[k] = row + ix[dx];
                c.ok[k] = ozy && ox[dx];
                c.w[k] = c.ok[k] ? wzy * c.wx[dx] : 0.0f;
            }
        }
    return c;
}

// Thread -> voxel without a per-voxel div/mod chain: grid = (ceil(H W / 256), D, B); one division by W per thread.
#define VXM_VOXEL_INDEX(D_, H_, W_)                                  \
    const int HW_ = (H_) * (W_), V = (D_) * HW_;                     \
    const int p2_ = blockIdx.x * 256 + threadIdx.x;                  \
    if (p2_ >= HW_) return;                                          \
    const int d = blockIdx.y, b = blockIdx.z;                        \
    const int h = p2_ / (W_), w = p2_ - h * (W_);                    \
    const int p = d * HW_ + p2_

// Addressing: every plane of a sample is reached through one buffer descriptor per tensor with 32-bit byte offsets (lane part
// 4 p or 4 idx, channel part in the wave-uniform scalar offset) -- the first version spent a quarter of its instructions on
// 64-bit pointer arithmetic for the 8 gathers (V < 2^29 voxels per plane is checked on the host).
template <int MODE>
__global__ void __launch_bounds__(256) k_warp3d_fwd(const float* __restrict__ src, const float* __restrict__ flow,
                                                    float* __restrict__ out, int C, int D, int H, int W) {
    VXM_VOXEL_INDEX(D, H, W);
    const __amdgpu_buffer_rsrc_t rf = vxm_rsrc(flow + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
    const int p4 = p << 2, V4 = V << 2;
    const float z = vxm_src_coord(d, vxm_bload(rf, p4, 0), D), y = vxm_src_coord(h, vxm_bload(rf, p4, V4), H),
                x = vxm_src_coord(w, vxm_bload(rf, p4, 2 * V4), W);
    const float* s = src + (size_t)b * C * V;
    float* o = out + (size_t)b * C * V;
    if (MODE == VXM_INTERP_NEAREST) {
        const float rz = rintf(z), ry = rintf(y), rx = rintf(x);       // nearbyint: round-half-even
        const bool in = (rz >= 0.0f) & (rz <= (float)(D - 1)) & (ry >= 0.0f) & (ry <= (float)(H - 1)) &
                        (rx >= 0.0f) & (rx <= (float)(W - 1));
        const int idx4 = in ? (((int)rz * H + (int)ry) * W + (int)rx) << 2 : VXM_OOB;      // outside: the load returns 0.0
        for (int c = 0; c < C; ++c) {
            const __amdgpu_buffer_rsrc_t rs = vxm_rsrc(s + (size_t)c * V, (unsigned)V4), ro = vxm_rsrc(o + (size_t)c * V, (unsigned)V4);
            vxm_bstore(vxm_bload(rs, idx4, 0), ro, p4, 0);
        }
        return;
    }
    const Corners8 cn = corners8(z, y, x, D, H, W);
    for (int c = 0; c < C; ++c) {
        const __amdgpu_buffer_rsrc_t rs = vxm_rsrc(s + (size_t)c * V, (unsigned)V4), ro = vxm_rsrc(o + (size_t)c * V, (unsigned)V4);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = vxm_bload(rs, cn.idx[k] << 2, 0);       // 8 independent gathers in flight
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k] * cn.w[k];
        vxm_bstore(acc, ro, p4, 0);
    }
}

// grid_sampler_3d_backward composed with the reference's normalisation chain: in voxel units
// d out / d flow_a = sum_corners sign_a * prod_{b != a} w_b * src[corner]  (in-bounds corners only).
template <int MODE>
__global__ void __launch_bounds__(256) k_warp3d_bwd(const float* __restrict__ src, const float* __restrict__ flow,
                                                    const float* __restrict__ gout, float* __restrict__ gsrc,
                                                    float* __restrict__ gflow, int C, int D, int H, int W) {
    VXM_VOXEL_INDEX(D, H, W);
    const __amdgpu_buffer_rsrc_t rf = vxm_rsrc(flow + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
    const int p4 = p << 2, V4 = V << 2;
    const float z = vxm_src_coord(d, vxm_bload(rf, p4, 0), D), y = vxm_src_coord(h, vxm_bload(rf, p4, V4), H),
                x = vxm_src_coord(w, vxm_bload(rf, p4, 2 * V4), W);
    const float* s = src + (size_t)b * C * V;
    const float* go = gout + (size_t)b * C * V;
    float* gs = gsrc ? gsrc + (size_t)b * C * V : nullptr;
    const __amdgpu_buffer_rsrc_t rg = vxm_rsrc(gflow ? gflow + (size_t)b * 3 * V : flow, gflow ? 3u * (unsigned)V * 4u : 0u);   // no gflow: every store dropped
    if (MODE == VXM_INTERP_NEAREST) {
        vxm_bstore(0.0f, rg, p4, 0); vxm_bstore(0.0f, rg, p4, V4); vxm_bstore(0.0f, rg, p4, 2 * V4);
        if (gs) {
            const float rz = rintf(z), ry = rintf(y), rx = rintf(x);
            const bool in = (rz >= 0.0f) & (rz <= (float)(D - 1)) & (ry >= 0.0f) & (ry <= (float)(H - 1)) &
                            (rx >= 0.0f) & (rx <= (float)(W - 1));
            if (in) {
                const int idx = ((int)rz * H + (int)ry) * W + (int)rx;
                for (int c = 0; c < C; ++c) atomicAdd(gs + (size_t)c * V + idx, go[(size_t)c * V + p]);
            }
        }
        return;
    }
    const Corners8 cn = corners8(z, y, x, D, H, W);
    float gz = 0.0f, gy = 0.0f, gx = 0.0f;
    for (int c = 0; c < C; ++c) {
        const __amdgpu_buffer_rsrc_t rs = vxm_rsrc(s + (size_t)c * V, (unsigned)V4), rgo = vxm_rsrc(go + (size_t)c * V, (unsigned)V4);
        const float g = vxm_bload(rgo, p4, 0);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = vxm_bload(rs, cn.idx[k] << 2, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int dz = (k >> 2) & 1, dy = (k >> 1) & 1, dx = k & 1;
            const float vk = cn.ok[k] ? v[k] * g : 0.0f;
            gz += (dz ? vk : -vk) * (cn.wy[dy] * cn.wx[dx]);
            gy += (dy ? vk : -vk) * (cn.wz[dz] * cn.wx[dx]);
            gx += (dx ? vk : -vk) * (cn.wz[dz] * cn.wy[dy]);
            if (gs && cn.ok[k]) atomicAdd(gs + (size_t)c * V + cn.idx[k], g * cn.w[k]);
        }
    }
    vxm_bstore(gz, rg, p4, 0); vxm_bstore(gy, rg, p4, V4); vxm_bstore(gx, rg, p4, 2 * V4);
}

// One scaling-and-squaring step: out = v + warp(v, v), v = in * scale (scale is a power of two:
// the product is exact, so folding it here equals the reference's separate `vec * self.scale`).
__global__ void __launch_bounds__(256) k_vecint_step_fwd(const float* __restrict__ in, float scale,
                                                         float* __restrict__ out, int D, int H, int W) {
    VXM_VOXEL_INDEX(D, H, W);
    const __amdgpu_buffer_rsrc_t ri = vxm_rsrc(in + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
    const __amdgpu_buffer_rsrc_t ro = vxm_rsrc(out + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
    const int p4 = p << 2, V4 = V << 2;
    const float v0 = vxm_bload(ri, p4, 0) * scale, v1 = vxm_bload(ri, p4, V4) * scale, v2 = vxm_bload(ri, p4, 2 * V4) * scale;
    const Corners8 cn = corners8(vxm_src_coord(d, v0, D), vxm_src_coord(h, v1, H), vxm_src_coord(w, v2, W), D, H, W);
    float s0[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {                                     // 24 independent gathers in flight
        const int i4 = cn.idx[k] << 2;
        s0[k] = vxm_bload(ri, i4, 0); s1[k] = vxm_bload(ri, i4, V4); s2[k] = vxm_bload(ri, i4, 2 * V4);
    }
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a0 += (s0[k] * scale) * cn.w[k];
        a1 += (s1[k] * scale) * cn.w[k];
        a2 += (s2[k] * scale) * cn.w[k];
    }
    vxm_bstore(v0 + a0, ro, p4, 0); vxm_bstore(v1 + a1, ro, p4, V4); vxm_bstore(v2 + a2, ro, p4, 2 * V4);
}

// backward of one step.  With v = in*scale, out_c(p) = v_c(p) + sum_k w_k(v(p)) v_c(q_k):
//   d/d v_c(p)   : g_c(p)                                   (identity)
//                  + sum_c' g_c'(p) d(sample_c')/d flow_c     (v is also the flow)
//   d/d v_c(q_k) : g_c(p) w_k                                 (v is also the sampled source; scatter)
// (The first implementation scattered the last term with 27 fp32 atomics per voxel into a zeroed buffer; it was bound by the L2
// atomic rate, 0.66 ms per pair, and is gone: see the gather below.)
// ---- the same step backward as a GATHER (no atomics in the regime of scaling and squaring) ---------------------------------
// Voxel p scatters to the 8 corners of x'(p) = p + v(p).  When every |x'_a(p) - p_a| < 1 ("near": floor(x'_a) - p_a in {-1, 0}),
// those corners lie in the 3x3x3 neighbourhood of p, so a target q only ever receives from the 27 voxels p = q + o, o in {-1,0,1}^3:
//   gin_c(q) = scale * [ local_c(q) + sum_o near(q+o) * w(q; x'(q+o)) * g_c(q+o) ],   w = product over axes of
//   (1 - fr_a) if q_a == floor(x'_a), fr_a if q_a == floor(x'_a) + 1, else 0          (fr = x' - floor(x'), the forward's weights)
// A block stages, for its 4x8x32 tile grown by one voxel, the per-voxel record {fr_z, fr_y, fr_x, g_0, g_1, g_2, code} in LDS
// (code: near flag + the three floor offsets) -- 3 IEEE divisions per staged voxel instead of per (voxel, neighbour) -- and every
// thread sums its 27 candidates in a fixed order (deterministic).  Voxels that are not "near" (displacement >= 1 voxel: rare
// in VecInt, whose steps are v / 2^k) are counted and handled by a second launch that adds only their contributions with
// atomics; when the count is zero that launch exits at once.  gin needs no zeroing.
constexpr int VG_TD = 4, VG_TH = 8, VG_TW = 32;
constexpr int VG_LD = VG_TD + 2, VG_LH = VG_TH + 2, VG_LW = VG_TW + 2, VG_LN = VG_LD * VG_LH * VG_LW;      // 2040 staged voxels

__device__ __forceinline__ void vg_axis(float xp, int p, int S, float& fr, int& frel, bool& near) {
    const float f = floorf(xp);
    fr = xp - f;
    const float rel = f - (float)p;                       // exact for the magnitudes that matter; huge values fail the test below
    near = rel == -1.0f || rel == 0.0f;
    frel = rel == -1.0f ? 1 : 0;                          // 1: floor = p - 1
    (void)S;
}

// One thread per voxel of the 4 x 8 x 32 tile (1024-thread blocks, two per CU).  A block makes ONE trip to memory (the first gather made
// four: v -> near test -> g -> barrier -> own v, g -> 24 corner gathers): a thread loads v and g of its own voxel and of one halo voxel
// of the tile grown by one (1016 halo voxels on 1024 threads) in a single batch of 12 loads, and everything after the barrier is served
// from LDS.  That took the backward chain of the headline pair from 235 to 212 us; what bounds the step now is the vector ALU: 816 VALU
// instructions per thread (ISA count; 27 candidates x ~14, three IEEE divisions per staged voxel for the reference's normalise /
// un-normalise round trip, the 8-corner derivative) = 13440 waves x 816 x 4 cycles over 1024 SIMDs = 21 us of a 27 us step.
//   * the staged record of a sender is SIX floats: per axis one value E_a = fr_a with the floor code in its SIGN BIT (negative:
//     floor(x'_a) = p_a - 1; -0.0 is a valid value), and its gradient with the "near" test folded in (a sender that is not near stages
//     g = 0: its scatter belongs to the atomic pass) -- one ds_read_b128 + one ds_read_b64 per candidate.  The sender's weight towards
//     the target at offset o is rebuilt per candidate from E_a with the offset known at compile time: o = +1: floor code 1 ? 1 - fr : 0;
//     o = 0: code ? fr : 1 - fr;  o = -1: code ? 0 : fr  (~10 vector instructions per candidate; the first gather unpacked a code word
//     and compared it against every offset, ~24 per candidate, and a version with all nine weights precomputed in 48-byte records
//     needed 96 KB of LDS, one block per CU, and was SLOWER, 39 vs 30 us per step);
//   * v of the grown tile (3 planes of floats): the 8 corners of x'(q) of a near voxel q lie in its 3x3x3 neighbourhood, so the
//     derivative through the sampling position reads them from LDS (a voxel that is not near gathers them from global memory).
// 36 bytes per staged voxel, 73 KB per block.
typedef float vg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float vg_weight(float e, int o) {
    const bool r = (__float_as_uint(e) >> 31) != 0u;              // floor code: 1 <=> floor(x') = p - 1
    const float a = fabsf(e);
    return o > 0 ? (r ? 1.0f - a : 0.0f) : (o == 0 ? (r ? a : 1.0f - a) : (r ? 0.0f : a));
}
constexpr int VG_NHALO = VG_LN - VG_TD * VG_TH * VG_TW;          // 1016 voxels of the grown tile outside the tile
static_assert(VG_NHALO <= VG_TD * VG_TH * VG_TW, "one halo voxel per thread");
// halo voxel j -> (lz, ly, lx) of the grown tile: the two z faces, then the y faces of the inner planes, then their x edges
__device__ __forceinline__ void vg_halo_cell(int j, int& lz, int& ly, int& lx) {
    constexpr int FZ = VG_LH * VG_LW, NZ = 2 * FZ, NY = VG_TD * 2 * VG_LW;
    if (j < NZ) {
        const int s = j / FZ, r = j - s * FZ;
        lz = s ? VG_LD - 1 : 0; ly = r / VG_LW; lx = r - ly * VG_LW;
    } else if (j < NZ + NY) {
        const int r = j - NZ, z = r / (2 * VG_LW), r2 = r - z * (2 * VG_LW), s = r2 / VG_LW;
        lz = 1 + z; ly = s ? VG_LH - 1 : 0; lx = r2 - s * VG_LW;
    } else {
        const int r = j - NZ - NY, z = r / (2 * VG_TH), r2 = r - z * (2 * VG_TH);
        lz = 1 + z; ly = 1 + (r2 >> 1); lx = (r2 & 1) ? VG_LW - 1 : 0;
    }
}
struct VgStaged { float e[3]; bool near; int frel[3]; };
// fr / floor code / near test of a voxel with step vector v at (pz, py, px)
__device__ __forceinline__ VgStaged vg_stage(float v0, float v1, float v2, int pz, int py, int px, int D, int H, int W) {
    VgStaged s;
    float fz, fy, fx;
    bool nz, ny, nx;
    vg_axis(vxm_src_coord(pz, v0, D), pz, D, fz, s.frel[0], nz);
    vg_axis(vxm_src_coord(py, v1, H), py, H, fy, s.frel[1], ny);
    vg_axis(vxm_src_coord(px, v2, W), px, W, fx, s.frel[2], nx);
    s.near = nz && ny && nx;
    s.e[0] = s.frel[0] ? -fz : fz; s.e[1] = s.frel[1] ? -fy : fy; s.e[2] = s.frel[2] ? -fx : fx;      // fr >= 0: the sign bit is free (-0.0 when fr == 0)
    return s;
}
__global__ void __launch_bounds__(1024) k_vecint_step_bwd_gather(const float* __restrict__ in, float scale, const float* __restrict__ gout,
                                                                 float* __restrict__ gin, unsigned* __restrict__ far_count, unsigned* __restrict__ tile_dm,
                                                                 int D, int H, int W) {
    __shared__ f32x4 recA[VG_LN];                                 // {E_z, E_y, E_x, g_0 (0 when not near)}
    __shared__ vg_f32x2 recB[VG_LN];                              // {g_1, g_2}
    __shared__ float vL[3][VG_LN];                                // v = in * scale
    __shared__ unsigned farw[16][3];                              // per wave: far senders, their largest displacement / |g| (float bits)
    const int tid = threadIdx.x, tx = tid & 31, ty = (tid >> 5) & 7, dd = tid >> 8;
    const int ntw = (W + VG_TW - 1) / VG_TW, nth = (H + VG_TH - 1) / VG_TH;
    int t = blockIdx.x;
    const int w0 = (t % ntw) * VG_TW; t /= ntw;
    const int h0 = (t % nth) * VG_TH;
    const int d0 = (t / nth) * VG_TD;
    const int b = blockIdx.y;
    const int HW = H * W, V = D * HW;
    // three planes per tensor behind one buffer descriptor each: 32-bit byte offsets, channel step in the scalar offset; a voxel outside
    // the volume gets an out-of-range offset and loads zeros
    const __amdgpu_buffer_rsrc_t rv = vxm_rsrc(in + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
    const __amdgpu_buffer_rsrc_t rgo = vxm_rsrc(gout + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
    const __amdgpu_buffer_rsrc_t rgi = vxm_rsrc(gin + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
    const int V4 = V << 2;
    // own voxel q and one halo voxel: 12 loads in one batch
    const int h = h0 + ty, w = w0 + tx, d = d0 + dd;
    const bool own_in = h < H && w < W && d < D;
    const int p4 = (d * HW + h * W + w) << 2;
    const int lq = ((dd + 1) * VG_LH + ty + 1) * VG_LW + tx + 1;
    int hz = 0, hy = 0, hx = 0;
    if (tid < VG_NHALO) vg_halo_cell(tid, hz, hy, hx);
    const int qz = d0 - 1 + hz, qy = h0 - 1 + hy, qx = w0 - 1 + hx;
    const bool halo_in = tid < VG_NHALO && (unsigned)qz < (unsigned)D && (unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W;
    const int q4 = (qz * HW + qy * W + qx) << 2;
    const int lh = (hz * VG_LH + hy) * VG_LW + hx;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, u0 = 0.f, u1 = 0.f, u2 = 0.f, k0 = 0.f, k1 = 0.f, k2 = 0.f;
    if (own_in) {
        v0 = vxm_bload(rv, p4, 0); v1 = vxm_bload(rv, p4, V4); v2 = vxm_bload(rv, p4, 2 * V4);
        g0 = vxm_bload(rgo, p4, 0); g1 = vxm_bload(rgo, p4, V4); g2 = vxm_bload(rgo, p4, 2 * V4);      // the voxel's own upstream gradient
    }
    if (halo_in) {
        u0 = vxm_bload(rv, q4, 0); u1 = vxm_bload(rv, q4, V4); u2 = vxm_bload(rv, q4, 2 * V4);
        k0 = vxm_bload(rgo, q4, 0); k1 = vxm_bload(rgo, q4, V4); k2 = vxm_bload(rgo, q4, 2 * V4);
    }
    v0 *= scale; v1 *= scale; v2 *= scale; u0 *= scale; u1 *= scale; u2 *= scale;
    const VgStaged so = vg_stage(v0, v1, v2, d, h, w, D, H, W);
    {
        const bool send = own_in && so.near;                      // a sender that is not near (or not in the volume) stages g = 0
        recA[lq] = (f32x4){so.e[0], so.e[1], so.e[2], send ? g0 : 0.0f};
        recB[lq] = (vg_f32x2){send ? g1 : 0.0f, send ? g2 : 0.0f};
        vL[0][lq] = v0; vL[1][lq] = v1; vL[2][lq] = v2;
    }
    // the senders the gather leaves out (displacement >= 1 voxel): their number, largest displacement and largest |g| of the block ->
    // far_count[0..2] (one set of atomics per BLOCK: in the regime of a trained network most voxels of the last steps are such senders)
    {
        const bool isfar = own_in && !so.near;
        const float dsp = isfar ? fmaxf(fmaxf(fabsf(vxm_src_coord(d, v0, D) - (float)d), fabsf(vxm_src_coord(h, v1, H) - (float)h)),
                                        fabsf(vxm_src_coord(w, v2, W) - (float)w)) : 0.0f;
        const float gmx = isfar ? fmaxf(fmaxf(fabsf(g0), fabsf(g1)), fabsf(g2)) : 0.0f;
        const unsigned long long bal = __ballot(isfar);
        float wd = dsp, wg = gmx;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { wd = fmaxf(wd, __shfl_xor(wd, o, 64)); wg = fmaxf(wg, __shfl_xor(wg, o, 64)); }
        if ((tid & 63) == 0) { farw[tid >> 6][0] = (unsigned)__popcll(bal); farw[tid >> 6][1] = __float_as_uint(wd); farw[tid >> 6][2] = __float_as_uint(wg); }
    }
    if (tid < VG_NHALO) {
        const VgStaged sh = vg_stage(u0, u1, u2, qz, qy, qx, D, H, W);
        const bool send = halo_in && sh.near;
        recA[lh] = (f32x4){sh.e[0], sh.e[1], sh.e[2], send ? k0 : 0.0f};
        recB[lh] = (vg_f32x2){send ? k1 : 0.0f, send ? k2 : 0.0f};
        vL[0][lh] = u0; vL[1][lh] = u1; vL[2][lh] = u2;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned cnt = 0, dm = 0, gm = 0;
        for (int i = 0; i < 16; ++i) { cnt += farw[i][0]; dm = max(dm, farw[i][1]); gm = max(gm, farw[i][2]); }       // (non-negative floats order as their bits)
        if (cnt) { atomicAdd(far_count, cnt); atomicMax(far_count + 1, dm); atomicMax(far_count + 2, gm); }
        // the largest displacement among THIS tile's far senders (float bits, 0: none): the far pass grows an output tile by what the sender
        // tiles around it need, not by the batch's maximum (round 6).  Every block writes its entry on every step: no zeroing.
        if (tile_dm) tile_dm[(size_t)b * gridDim.x + blockIdx.x] = cnt ? dm : 0u;
    }
    if (!own_in) return;
    // ---- local part: identity + derivative through the sampling position (the 8 corners of x'(q) sampled from v itself)
    const Corners8 cn = corners8(vxm_src_coord(d, v0, D), vxm_src_coord(h, v1, H), vxm_src_coord(w, v2, W), D, H, W);
    float gz = g0, gy = g1, gx = g2;
    const int lc = lq - (so.frel[0] * VG_LH + so.frel[1]) * VG_LW - so.frel[2];        // corner 0 in the grown tile (near voxels)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dz = (k >> 2) & 1, dy = (k >> 1) & 1, dx = k & 1;
        float c0, c1, c2;
        if (so.near) {
            const int i = lc + (dz * VG_LH + dy) * VG_LW + dx;
            c0 = vL[0][i]; c1 = vL[1][i]; c2 = vL[2][i];
        } else {
            const int i4 = cn.idx[k] << 2;
            c0 = vxm_bload(rv, i4, 0) * scale; c1 = vxm_bload(rv, i4, V4) * scale; c2 = vxm_bload(rv, i4, 2 * V4) * scale;
        }
        const float sk = cn.ok[k] ? c0 * g0 + c1 * g1 + c2 * g2 : 0.0f;
        gz += (dz ? sk : -sk) * (cn.wy[dy] * cn.wx[dx]);
        gy += (dy ? sk : -sk) * (cn.wz[dz] * cn.wx[dx]);
        gx += (dx ? sk : -sk) * (cn.wz[dz] * cn.wy[dy]);
    }
    // ---- gathered part: the 27 possible senders, fixed order (deterministic)
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
    for (int oz = -1; oz <= 1; ++oz)
#pragma unroll
        for (int oy = -1; oy <= 1; ++oy)
#pragma unroll
            for (int ox = -1; ox <= 1; ++ox) {
                const int i = lq + (oz * VG_LH + oy) * VG_LW + ox;
                const f32x4 ra = recA[i];
                const vg_f32x2 rb = recB[i];
                // sender p = q + o: q is corner 0 of x'(p) along axis a if o_a == frel_a, corner 1 if o_a == frel_a - 1
                const float wk = (vg_weight(ra.x, oz) * vg_weight(ra.y, oy)) * vg_weight(ra.z, ox);
                a0 += ra.w * wk; a1 += rb.x * wk; a2 += rb.y * wk;
            }
    vxm_bstore((gz + a0) * scale, rgi, p4, 0);
    vxm_bstore((gy + a1) * scale, rgi, p4, V4);
    vxm_bstore((gx + a2) * scale, rgi, p4, 2 * V4);
}

// second launch of the step: the scatter contributions of the voxels the gather skipped (displacement >= 1 voxel), by atomics
__device__ __forceinline__ void vecint_far_voxel(const float* __restrict__ in, float scale, const float* __restrict__ gout, float* __restrict__ gin,
                                                      int b, int p, int D, int H, int W) {
    const int HWf = H * W, V = D * HWf;
    const int d = p / HWf, r = p - d * HWf, h = r / W, w = r - h * W;
    const float* vin = in + (size_t)b * 3 * V;
    float* gi = gin + (size_t)b * 3 * V;
    const float v0 = vin[p] * scale, v1 = vin[V + p] * scale, v2 = vin[2 * (size_t)V + p] * scale;
    const float xz = vxm_src_coord(d, v0, D), xy = vxm_src_coord(h, v1, H), xx = vxm_src_coord(w, v2, W);
    float fr;
    int rr;
    bool nz, ny, nx;
    vg_axis(xz, d, D, fr, rr, nz); vg_axis(xy, h, H, fr, rr, ny); vg_axis(xx, w, W, fr, rr, nx);
    if (nz && ny && nx) return;
    const float* go = gout + (size_t)b * 3 * V + p;
    const float g0 = go[0], g1 = go[V], g2 = go[2 * (size_t)V];
    const Corners8 cn = corners8(xz, xy, xx, D, H, W);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!cn.ok[k]) continue;
        const int i = cn.idx[k];
        const float wk = cn.w[k] * scale;
        atomicAdd(gi + i, g0 * wk);
        atomicAdd(gi + V + i, g1 * wk);
        atomicAdd(gi + 2 * (size_t)V + i, g2 * wk);
    }
}
// Far senders whose displacement exceeds what the tile pass below covers (> VF_RMAX voxels per step: not a registration field any more)
// keep the atomic scatter (order-dependent sums); it runs inside the same launch as the tile pass (k_vecint_step_bwd_far_tiles), one
// thread per voxel of the block's tile.
constexpr int VF_RMAX = 24;
// The scatter of the far senders, DETERMINISTIC (round 4; rounds 1-3 added them to gin with global float atomics, whose order -- and
// with it the sum -- changes from run to run).  One block per 4 x 8 x 32 OUTPUT tile walks the tile grown by R = the step's largest
// displacement (known on the device: far_count[1]), recomputes the near test of every candidate sender, and adds the corner
// contributions of the far ones that land inside ITS tile to 64-bit FIXED-POINT accumulators in LDS (ds_add_u64: integer addition is
// associative, so the order of the LDS atomics does not matter).  The scale of the fixed point comes from the step's largest |g|
// (far_count[2]): 2^46 <= max |g| 2^k < 2^47, i.e. 46 significant bits below the largest contribution and 16 bits of headroom above.
// Every output voxel is then read-modify-written by exactly one thread.  Cost ~ (grown tile / tile) x the forward step's arithmetic.
__global__ void __launch_bounds__(1024) k_vecint_step_bwd_far_tiles(const float* __restrict__ in, float scale, const float* __restrict__ gout,
                                                                    float* __restrict__ gin, const unsigned* __restrict__ far_count,
                                                                    const unsigned* __restrict__ tile_dm, int B, int D, int H, int W) {
    if (far_count[0] == 0) return;                                // block-uniform: the regime of small steps (the launch is a small persistent grid)
    const float dm = __uint_as_float(far_count[1]);
    __shared__ unsigned long long acc[3][VG_TD * VG_TH * VG_TW];
    __shared__ unsigned pmask[3][VG_TD * VG_TH * VG_TW / 32];      // targets that received a non-finite contribution: THEY become NaN (grid_sample's
                                                                  // backward poisons the 8 corners of the sender, not the tile; ADVICE rounds 4 / 5)
    __shared__ unsigned rtile;                                    // this output tile's radius (float bits)
    const int tid = threadIdx.x;
    const int tx = tid & 31, ty = (tid >> 5) & 7, dd = tid >> 8;
    int Eg = (int)(far_count[2] >> 23) & 255;
    Eg = Eg < 1 ? 1 : Eg;                                         // fixed point: value * 2^(173 - Eg)
    const bool finite_g = Eg < 255;                               // the step's largest |g| is finite (NaN never wins the maxima: checked per contribution below)
    const int ntw = (W + VG_TW - 1) / VG_TW, nth = (H + VG_TH - 1) / VG_TH, ntd = (D + VG_TD - 1) / VG_TD;
    const int HW = H * W, V = D * HW, V4 = V << 2;
    const int ntile = ntw * nth * ntd;
    const bool atomic_pass = !(dm < (float)VF_RMAX) || !finite_g;  // beyond the tile pass (or an Inf among the gradients): the atomic scatter, which propagates NaN / Inf
    for (int job = blockIdx.x; job < ntile * B; job += gridDim.x) {
        const int b = job / ntile;
        int t = job - b * ntile;
        const int w0 = (t % ntw) * VG_TW; t /= ntw;
        const int h0 = (t % nth) * VG_TH;
        const int d0 = (t / nth) * VG_TD;
        if (atomic_pass) {
            const int h = h0 + ty, w = w0 + tx, d = d0 + dd;
            if (h < H && w < W && d < D) vecint_far_voxel(in, scale, gout, gin, b, d * HW + h * W + w, D, H, W);
            continue;
        }
        const __amdgpu_buffer_rsrc_t rv = vxm_rsrc(in + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
        const __amdgpu_buffer_rsrc_t rgo = vxm_rsrc(gout + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
        __syncthreads();                                          // the previous tile of this block is written back
        acc[0][tid] = 0ull; acc[1][tid] = 0ull; acc[2][tid] = 0ull;
        if (tid < 3 * (VG_TD * VG_TH * VG_TW / 32)) (&pmask[0][0])[tid] = 0u;
        if (tid == 0) rtile = tile_dm ? 0u : far_count[1];
        __syncthreads();
        if (tile_dm) {
            // Radius of THIS tile: the largest displacement among the sender tiles that can reach it -- a sender tile S with largest
            // displacement d_S reaches the voxels within ceil(d_S) of its box.  Candidates: the tiles within the step's global maximum.
            // (One 20-voxel outlier used to make EVERY tile walk (4 + 42) x (8 + 42) x (32 + 42) voxels, ~200 x its volume.)
            const int Rg = (int)dm + 1;
            const int tz0 = d0 / VG_TD, ty0 = h0 / VG_TH, tx0 = w0 / VG_TW;
            const int nz = (Rg + VG_TD - 1) / VG_TD, ny = (Rg + VG_TH - 1) / VG_TH, nx = (Rg + VG_TW - 1) / VG_TW;
            const int cz = 2 * nz + 1, cy = 2 * ny + 1, cx = 2 * nx + 1;
            unsigned best = 0u;
            for (int i = tid; i < cz * cy * cx; i += 1024) {
                const int oz = i / (cy * cx) - nz, r2 = i % (cy * cx), oy = r2 / cx - ny, ox = r2 % cx - nx;
                const int sz = tz0 + oz, sy = ty0 + oy, sx = tx0 + ox;
                if ((unsigned)sz >= (unsigned)ntd || (unsigned)sy >= (unsigned)nth || (unsigned)sx >= (unsigned)ntw) continue;
                const unsigned dS = tile_dm[(size_t)b * ntile + (sz * nth + sy) * ntw + sx];
                if (dS == 0u) continue;
                const int reach = (int)__uint_as_float(dS) + 1;
                // gap between the boxes along each axis (0 when they touch or overlap)
                const int gz = max(0, (abs(oz) - 1) * VG_TD + 1), gy = max(0, (abs(oy) - 1) * VG_TH + 1), gx = max(0, (abs(ox) - 1) * VG_TW + 1);
                if ((oz == 0 || gz <= reach) && (oy == 0 || gy <= reach) && (ox == 0 || gx <= reach)) best = max(best, dS);
            }
            if (best) atomicMax(&rtile, best);
            __syncthreads();
            if (rtile == 0u) continue;                            // no far sender reaches this tile (block-uniform)
        }
        const int R = (int)__uint_as_float(rtile) + 1;            // corners of a sender lie within ceil(displacement) of it
        const int gy = VG_TH + 2 * R, gx = VG_TW + 2 * R, ng = (VG_TD + 2 * R) * gy * gx;
        for (int i = tid; i < ng; i += 1024) {
            const int lz = i / (gy * gx), r2 = i - lz * gy * gx, ly = r2 / gx, lx = r2 - ly * gx;
            const int pz = d0 - R + lz, py = h0 - R + ly, px = w0 - R + lx;
            if ((unsigned)pz >= (unsigned)D || (unsigned)py >= (unsigned)H || (unsigned)px >= (unsigned)W) continue;
            const int p4 = (pz * HW + py * W + px) << 2;
            const float v0 = vxm_bload(rv, p4, 0) * scale, v1 = vxm_bload(rv, p4, V4) * scale, v2 = vxm_bload(rv, p4, 2 * V4) * scale;
            const float xz = vxm_src_coord(pz, v0, D), xy = vxm_src_coord(py, v1, H), xx = vxm_src_coord(px, v2, W);
            float fr;
            int rr;
            bool nz, ny, nx;
            vg_axis(xz, pz, D, fr, rr, nz); vg_axis(xy, py, H, fr, rr, ny); vg_axis(xx, px, W, fr, rr, nx);
            if (nz && ny && nx) continue;                             // a near sender: the gather has it
            const AxisTaps az = axis_corners(xz, D), ay = axis_corners(xy, H), ax = axis_corners(xx, W);
            const int iz[2] = {az.i0, az.i1}, iy[2] = {ay.i0, ay.i1}, ix[2] = {ax.i0, ax.i1};
            const float wz[2] = {az.w0, az.w1}, wy[2] = {ay.w0, ay.w1}, wx[2] = {ax.w0, ax.w1};
            const bool oz[2] = {az.ok0 && (unsigned)(az.i0 - d0) < (unsigned)VG_TD, az.ok1 && (unsigned)(az.i1 - d0) < (unsigned)VG_TD};
            const bool oy[2] = {ay.ok0 && (unsigned)(ay.i0 - h0) < (unsigned)VG_TH, ay.ok1 && (unsigned)(ay.i1 - h0) < (unsigned)VG_TH};
            const bool ox[2] = {ax.ok0 && (unsigned)(ax.i0 - w0) < (unsigned)VG_TW, ax.ok1 && (unsigned)(ax.i1 - w0) < (unsigned)VG_TW};
            if (!((oz[0] || oz[1]) && (oy[0] || oy[1]) && (ox[0] || ox[1]))) continue;
            const float g[3] = {vxm_bload(rgo, p4, 0), vxm_bload(rgo, p4, V4), vxm_bload(rgo, p4, 2 * V4)};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dz = (k >> 2) & 1, dy = (k >> 1) & 1, dx = k & 1;
                if (!(oz[dz] && oy[dy] && ox[dx])) continue;
                const int li = ((iz[dz] - d0) * VG_TH + (iy[dy] - h0)) * VG_TW + (ix[dx] - w0);
                const float wk = (wz[dz] * wy[dy]) * wx[dx];           // the forward's weight of this corner
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const unsigned u = __float_as_uint(g[c] * wk);
                    const int ev = (int)(u >> 23) & 255;
                    if (ev == 255) { atomicOr(&pmask[c][li >> 5], 1u << (li & 31)); continue; }      // NaN (fmaxf / atomicMax drop it from far_count[2]) or Inf: fixed point cannot carry it
                    int sh = ev - Eg + 23;                            // <= 24: |g wk| <= the step's largest |g| (+ one rounding)
                    if (ev == 0 || sh <= -24) continue;
                    sh = sh > 24 ? 24 : sh;
                    const unsigned long long m = (unsigned long long)((u & 0x7fffffu) | 0x800000u);
                    const unsigned long long mag = sh >= 0 ? m << sh : m >> (-sh);
                    atomicAdd(&acc[c][li], (u >> 31) ? (0ull - mag) : mag);      // two's complement: signed sums wrap correctly
                }
            }
        }
        __syncthreads();
        const int h = h0 + ty, w = w0 + tx, d = d0 + dd;
        if (h >= H || w >= W || d >= D) continue;
        float* gi = gin + (size_t)b * 3 * V + (size_t)d * HW + h * W + w;
        // 2^(Eg - 173) as a double (exponent field 1023 + Eg - 173)
        const double unit = __longlong_as_double((long long)(1023 + Eg - 173) << 52);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const long long sacc = (long long)acc[c][tid];
            const bool bad = (pmask[c][tid >> 5] >> (tid & 31)) & 1u;
            if (bad) gi[(size_t)c * V] = __uint_as_float(0x7fc00000u);        // a diverged step surfaces in dL/dvec, at the voxels the sender targets
            else if (sacc != 0) gi[(size_t)c * V] += (float)((double)sacc * unit) * scale;
        }
    }
}

// adjoint of the trilinear resize as a scatter (gx zeroed by the caller); only used for very large upsampling factors
__global__ void __launch_bounds__(256) k_resize3d_bwd(const float* __restrict__ gout, float* __restrict__ gx, int D, int H,
                                                      int W, int oD, int oH, int oW, float rd, float rh, float rw,
                                                      float scale) {
    const int oV = oD * oH * oW;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= oV) return;
    const size_t bc = blockIdx.y;
    const int w = p % oW, t = p / oW, h = t % oH, d = t / oH;
    int z0, z1, y0, y1, x0, x1;
    float lz0, lz1, ly0, ly1, lx0, lx1;
    lin_src(d, rd, D, z0, z1, lz0, lz1);
    lin_src(h, rh, H, y0, y1, ly0, ly1);
    lin_src(w, rw, W, x0, x1, lx0, lx1);
    const float g = gout[bc * (size_t)oV + p] * scale;
    float* o = gx + bc * (size_t)D * H * W;
#define ADD(zz, yy, xx, wt) atomicAdd(o + ((size_t)(zz) * H + (yy)) * W + (xx), g * (wt))
    ADD(z0, y0, x0, lz0 * ly0 * lx0); ADD(z0, y0, x1, lz0 * ly0 * lx1);
    ADD(z0, y1, x0, lz0 * ly1 * lx0); ADD(z0, y1, x1, lz0 * ly1 * lx1);
    ADD(z1, y0, x0, lz1 * ly0 * lx0); ADD(z1, y0, x1, lz1 * ly0 * lx1);
    ADD(z1, y1, x0, lz1 * ly1 * lx0); ADD(z1, y1, x1, lz1 * ly1 * lx1);
#undef ADD
}

// Gather form of the adjoint (deterministic, no atomics): input voxel i receives, per axis, the
// outputs o whose source index pair (i0, i1) contains i.  Those satisfy ratio*o in [i-1, i+1), i.e. at
// most floor(2/ratio)+1 consecutive outputs; the KC candidates starting one below floor((i-1)/ratio)
// are a superset and each is re-evaluated with the forward's exact index/lambda arithmetic.
constexpr int RS_KC = 7;
__device__ __forceinline__ void axis_taps(int i, float ratio, int n_in, int n_out, int& olo, float (&w)[RS_KC]) {
    olo = ratio > 0.0f ? max(0, (int)floorf((float)(i - 1) / ratio) - 1) : 0;
#pragma unroll
    for (int k = 0; k < RS_KC; ++k) {
        const int o = olo + k;
        int i0, i1;
        float l0, l1;
        lin_src(min(o, n_out - 1), ratio, n_in, i0, i1, l0, l1);
        const float wt = (i0 == i ? l0 : 0.0f) + (i1 == i ? l1 : 0.0f);
        w[k] = o < n_out ? wt : 0.0f;
    }
}

// ---- tiled forms: a block owns a 4 x 8 x 32 (D x H x W) tile, its 44 per-axis interpolation records are computed
// once (44 lin_src / axis_taps evaluations instead of 3 per thread and depth) and staged in LDS, and the
// thread's w / h never need a div / mod.  Same arithmetic and summation order as the per-voxel kernels above.
constexpr int RT_W = 32, RT_H = 8, RT_D = 4, RT_N = RT_W + RT_H + RT_D;

// the trilinear combination of upsample_trilinear3d in ONE place (resize forward and the fused resize + warp kernels evaluate the same expression)
// Roundings pinned (explicit fused multiply-adds, no contraction left to the compiler): a kernel that evaluates the in-plane part once per
// staged plane and reuses it for several output depths (k_warp3d_up_*) gets the bits of one that evaluates the whole expression per voxel.
__device__ __forceinline__ float rt_bilerp(float ly0, float ly1, float lx0, float lx1, float a00, float a01, float a10, float a11) {
#pragma clang fp contract(off)
    const float x0 = __builtin_fmaf(lx0, a00, lx1 * a01), x1 = __builtin_fmaf(lx0, a10, lx1 * a11);
    return __builtin_fmaf(ly0, x0, ly1 * x1);
}
__device__ __forceinline__ float rt_zlerp(float lz0, float lz1, float y0, float y1) {
#pragma clang fp contract(off)
    return __builtin_fmaf(lz0, y0, lz1 * y1);
}
__device__ __forceinline__ float rt_trilerp(float lz0, float lz1, float ly0, float ly1, float lx0, float lx1, float a000, float a001, float a010,
                                            float a011, float a100, float a101, float a110, float a111) {
    return rt_zlerp(lz0, lz1, rt_bilerp(ly0, ly1, lx0, lx1, a000, a001, a010, a011), rt_bilerp(ly0, ly1, lx0, lx1, a100, a101, a110, a111));
}

__device__ __forceinline__ void rt_tile(int nW, int nH, int& d0, int& h0, int& w0) {
    const int tw = (nW + RT_W - 1) / RT_W, th = (nH + RT_H - 1) / RT_H;
    int t = blockIdx.x;
    w0 = (t % tw) * RT_W; t /= tw;
    h0 = (t % th) * RT_H;
    d0 = (t / th) * RT_D;
}

__global__ void __launch_bounds__(256) k_resize3d_fwd_tiled(const float* __restrict__ x, float* __restrict__ out, int D, int H, int W,
                                                            int oD, int oH, int oW, float rd, float rh, float rw, float pre, float post) {
    __shared__ int si0[RT_N], si1[RT_N];
    __shared__ float sl0[RT_N], sl1[RT_N];
    int d0, h0, w0;
    rt_tile(oW, oH, d0, h0, w0);
    const int tid = threadIdx.x;
    if (tid < RT_N) {
        const int ax = tid < RT_W ? 2 : (tid < RT_W + RT_H ? 1 : 0);
        const int dst = ax == 2 ? w0 + tid : (ax == 1 ? h0 + tid - RT_W : d0 + tid - RT_W - RT_H);
        const int n_out = ax == 2 ? oW : (ax == 1 ? oH : oD), n_in = ax == 2 ? W : (ax == 1 ? H : D);
        int i0, i1;
        float l0, l1;
        lin_src(min(dst, n_out - 1), ax == 2 ? rw : (ax == 1 ? rh : rd), n_in, i0, i1, l0, l1);
        si0[tid] = i0; si1[tid] = i1; sl0[tid] = l0; sl1[tid] = l1;
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
    const int w = w0 + tx, h = h0 + ty;
    if (w >= oW || h >= oH) return;
    const size_t bc = blockIdx.y;
    const float* s = x + bc * (size_t)D * H * W;
    const int x0 = si0[tx], x1 = si1[tx], y0 = si0[RT_W + ty], y1 = si1[RT_W + ty];
    const float lx0 = sl0[tx], lx1 = sl1[tx], ly0 = sl0[RT_W + ty], ly1 = sl1[RT_W + ty];
#pragma unroll
    for (int dd = 0; dd < RT_D; ++dd) {
        const int d = d0 + dd;
        if (d >= oD) break;
        const int z0 = si0[RT_W + RT_H + dd], z1 = si1[RT_W + RT_H + dd];
        const float lz0 = sl0[RT_W + RT_H + dd], lz1 = sl1[RT_W + RT_H + dd];
#define AT(zz, yy, xx) (pre * s[((size_t)(zz) * H + (yy)) * W + (xx)])
        const float v = rt_trilerp(lz0, lz1, ly0, ly1, lx0, lx1, AT(z0, y0, x0), AT(z0, y0, x1), AT(z0, y1, x0), AT(z0, y1, x1),
                                   AT(z1, y0, x0), AT(z1, y0, x1), AT(z1, y1, x0), AT(z1, y1, x1));
#undef AT
        out[bc * (size_t)oD * oH * oW + ((size_t)d * oH + h) * oW + w] = post * v;
    }
}

__global__ void __launch_bounds__(256) k_resize3d_bwd_gather_tiled(const float* __restrict__ gout, float* __restrict__ gx, int D, int H,
                                                                   int W, int oD, int oH, int oW, float rd, float rh, float rw,
                                                                   float scale) {
    __shared__ int solo[RT_N], sklo[RT_N], skhi[RT_N];
    __shared__ float sw[RT_N][RS_KC + 1];
    int d0, h0, w0;
    rt_tile(W, H, d0, h0, w0);
    const int tid = threadIdx.x;
    if (tid < RT_N) {
        const int ax = tid < RT_W ? 2 : (tid < RT_W + RT_H ? 1 : 0);
        const int i = ax == 2 ? w0 + tid : (ax == 1 ? h0 + tid - RT_W : d0 + tid - RT_W - RT_H);
        const int n_in = ax == 2 ? W : (ax == 1 ? H : D), n_out = ax == 2 ? oW : (ax == 1 ? oH : oD);
        int olo;
        float wt[RS_KC];
        axis_taps(min(i, n_in - 1), ax == 2 ? rw : (ax == 1 ? rh : rd), n_in, n_out, olo, wt);
        int klo = RS_KC, khi = 0;
#pragma unroll
        for (int k = 0; k < RS_KC; ++k) {
            sw[tid][k] = wt[k];
            if (wt[k] != 0.0f) { klo = min(klo, k); khi = k + 1; }
        }
        solo[tid] = olo; sklo[tid] = klo; skhi[tid] = khi;
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
    const int w = w0 + tx, h = h0 + ty;
    if (w >= W || h >= H) return;
    const size_t bc = blockIdx.y;
    const float* g = gout + bc * (size_t)oD * oH * oW;
    const int ow0 = solo[tx], cl = sklo[tx], ch = skhi[tx];
    const int oh0 = solo[RT_W + ty], el = sklo[RT_W + ty], eh = skhi[RT_W + ty];
    for (int dd = 0; dd < RT_D; ++dd) {
        const int d = d0 + dd;
        if (d >= D) break;
        const int od0 = solo[RT_W + RT_H + dd], al = sklo[RT_W + RT_H + dd], ah = skhi[RT_W + RT_H + dd];
        float acc = 0.0f;
        for (int a = al; a < ah; ++a) {
            const float wd = sw[RT_W + RT_H + dd][a];
            for (int e = el; e < eh; ++e) {
                const float wdh = wd * sw[RT_W + ty][e];
                const float* row = g + ((size_t)(od0 + a) * oH + (oh0 + e)) * oW + ow0;
                for (int c = cl; c < ch; ++c) acc += wdh * sw[tx][c] * row[c];
            }
        }
        gx[bc * (size_t)D * H * W + ((size_t)d * H + h) * W + w] = acc * scale;
    }
}

// ---- adjoint of a DOWN-sampling resize (every ratio >= 1: an input voxel is named by at most two outputs per axis, and by none when the
// ratio skips it): the gather above with its loops resolved -- two taps per axis from the same axis_taps() records, eight unconditional
// loads per voxel (clamped addresses, weight zero where a tap does not exist), same a -> e -> c summation order.  The general kernel ran
// its three data-dependent loops at 47 us for the 10 -> 83 MB of the headline field (k_resize3d_bwd_gather_tiled, 2.0 TB/s).
__global__ void __launch_bounds__(256) k_resize3d_bwd_down(const float* __restrict__ gout, float* __restrict__ gx, int D, int H, int W, int oD, int oH,
                                                           int oW, float rd, float rh, float rw, float scale) {
    // 16 depths per block (the 44 + 12 axis records cost as much as four voxels of a thread: amortised over sixteen)
    constexpr int RD_D = 16, RD_N = RT_W + RT_H + RD_D;
    __shared__ int so[RD_N];
    __shared__ float s0[RD_N], s1[RD_N];
    const int tw_ = (W + RT_W - 1) / RT_W, th_ = (H + RT_H - 1) / RT_H;
    int t_ = blockIdx.x;
    const int w0 = (t_ % tw_) * RT_W; t_ /= tw_;
    const int h0 = (t_ % th_) * RT_H;
    const int d0 = (t_ / th_) * RD_D;
    const int tid = threadIdx.x;
    if (tid < RD_N) {
        const int ax = tid < RT_W ? 2 : (tid < RT_W + RT_H ? 1 : 0);
        const int i = ax == 2 ? w0 + tid : (ax == 1 ? h0 + tid - RT_W : d0 + tid - RT_W - RT_H);
        const int n_in = ax == 2 ? W : (ax == 1 ? H : D), n_out = ax == 2 ? oW : (ax == 1 ? oH : oD);
        int olo;
        float wt[RS_KC];
        axis_taps(min(i, n_in - 1), ax == 2 ? rw : (ax == 1 ? rh : rd), n_in, n_out, olo, wt);
        int klo = 0;
        while (klo < RS_KC - 2 && wt[klo] == 0.0f) ++klo;                      // first tap that counts; at most one more follows (ratio >= 1)
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int j = 0; j < RS_KC; ++j) { a = j == klo ? wt[j] : a; b = j == klo + 1 ? wt[j] : b; }
        so[tid] = min(olo + klo, n_out - 1); s0[tid] = a; s1[tid] = olo + klo + 1 < n_out ? b : 0.0f;
    }
    __syncthreads();
    const int tx = tid & 31, ty = tid >> 5;
    const int w = w0 + tx, h = h0 + ty;
    if (w >= W || h >= H) return;
    const size_t bc = blockIdx.y;
    const __amdgpu_buffer_rsrc_t rg = vxm_rsrc(gout + bc * (size_t)oD * oH * oW, (unsigned)oD * (unsigned)oH * (unsigned)oW * 4u);
    const int ow0 = so[tx], ow1 = min(ow0 + 1, oW - 1), oh0 = so[RT_W + ty], oh1 = min(oh0 + 1, oH - 1);
    const float wc[2] = {s0[tx], s1[tx]}, we[2] = {s0[RT_W + ty], s1[RT_W + ty]};
    const int col[2] = {ow0, ow1}, row[2] = {oh0 * oW, oh1 * oW};
    float* o = gx + bc * (size_t)D * H * W;
#pragma unroll 4
    for (int dd = 0; dd < RD_D; ++dd) {
        const int d = d0 + dd;
        if (d >= D) break;
        const int od0 = so[RT_W + RT_H + dd], od1 = min(od0 + 1, oD - 1);
        const float wa[2] = {s0[RT_W + RT_H + dd], s1[RT_W + RT_H + dd]};
        const int pl[2] = {od0 * oH * oW, od1 * oH * oW};
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = vxm_bload(rg, (pl[k >> 2] + row[(k >> 1) & 1] + col[k & 1]) << 2, 0);
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += (wa[k >> 2] * we[(k >> 1) & 1]) * wc[k & 1] * g[k];
        o[((size_t)d * H + h) * W + w] = acc * scale;
    }
}

// ---- adjoint of an UP-sampling resize (ratio < 1: every input voxel receives from 4-5 outputs per axis, 64-125 terms in the
// gather above) as three separable 1-D passes through LDS: the forward weight is a product of per-axis weights, so
//   gx[d,h,w] = sum_od wD[d][od] sum_oh wH[h][oh] sum_ow wW[w][ow] g[od,oh,ow].
// A block owns a 4 x 8 x 16 tile of inputs, loads the (<= 14 x 22 x 38) block of output gradients its taps can touch once,
// and contracts W, then H, then D (7 candidate taps per axis from axis_taps(), zeros included): ~5x fewer FMAs, each gradient
// read once per tile instead of once per receiving voxel.  Deterministic.  A tile whose tap span does not fit the LDS block
// (ratios below ~0.47) takes the direct gather.
constexpr int RB_D = 4, RB_H = 8, RB_W = 16, RB_N = RB_W + RB_H + RB_D;
constexpr int RB_SD = 14, RB_SH = 22, RB_SW = 38, RB_THREADS = 1024;

__global__ void __launch_bounds__(RB_THREADS) k_resize3d_bwd_sep(const float* __restrict__ gout, float* __restrict__ gx, int D, int H, int W, int oD,
                                                          int oH, int oW, float rd, float rh, float rw, float scale) {
    VXM_DYN_SMEM(float, smem);                          // 66.5 KB: beyond the static limit
    float* const b0 = smem;                             // [RB_SD][RB_SH][RB_SW] output gradients; later [RB_SD][RB_H][RB_W]
    float* const b1 = b0 + RB_SD * RB_SH * RB_SW;       // [RB_SD][RB_SH][RB_W]
    float (*const sw)[RS_KC + 1] = reinterpret_cast<float (*)[RS_KC + 1]>(b1 + RB_SD * RB_SH * RB_W);
    int* const solo = reinterpret_cast<int*>(b1 + RB_SD * RB_SH * RB_W + RB_N * (RS_KC + 1));
    const int tw = (W + RB_W - 1) / RB_W, th = (H + RB_H - 1) / RB_H;
    int t = blockIdx.x;
    const int w0 = (t % tw) * RB_W; t /= tw;
    const int h0 = (t % th) * RB_H;
    const int d0 = (t / th) * RB_D;
    const int tid = threadIdx.x;
    if (tid < RB_N) {
        const int ax = tid < RB_W ? 2 : (tid < RB_W + RB_H ? 1 : 0);
        const int i = ax == 2 ? w0 + tid : (ax == 1 ? h0 + tid - RB_W : d0 + tid - RB_W - RB_H);
        const int n_in = ax == 2 ? W : (ax == 1 ? H : D), n_out = ax == 2 ? oW : (ax == 1 ? oH : oD);
        int olo;
        float wt[RS_KC];
        axis_taps(min(i, n_in - 1), ax == 2 ? rw : (ax == 1 ? rh : rd), n_in, n_out, olo, wt);
#pragma unroll
        for (int k = 0; k < RS_KC; ++k) sw[tid][k] = i < n_in ? wt[k] : 0.0f;
        solo[tid] = olo;
    }
    __syncthreads();
    const int bw = solo[0], bh = solo[RB_W], bd = solo[RB_W + RB_H];           // olo is non-decreasing along an axis
    const bool fits = solo[RB_W - 1] + RS_KC - bw <= RB_SW && solo[RB_W + RB_H - 1] + RS_KC - bh <= RB_SH && solo[RB_N - 1] + RS_KC - bd <= RB_SD;
    const size_t bc = blockIdx.y;
    const float* g = gout + bc * (size_t)oD * oH * oW;
    float* o = gx + bc * (size_t)D * H * W;
    if (!fits) {                                            // block-uniform: direct gather for this tile
        for (int idx = tid; idx < RB_D * RB_H * RB_W; idx += RB_THREADS) {
            const int tx = idx % RB_W, ty = (idx / RB_W) % RB_H, dd = idx / (RB_W * RB_H);
            const int w = w0 + tx, h = h0 + ty, d = d0 + dd;
            if (w >= W || h >= H || d >= D) continue;
            float acc = 0.0f;
            for (int a = 0; a < RS_KC; ++a)
                for (int e = 0; e < RS_KC; ++e)
                    for (int c = 0; c < RS_KC; ++c) {
                        const float wt = sw[RB_W + RB_H + dd][a] * sw[RB_W + ty][e] * sw[tx][c];
                        if (wt != 0.0f)
                            acc += wt * g[((size_t)(solo[RB_W + RB_H + dd] + a) * oH + (solo[RB_W + ty] + e)) * oW + solo[tx] + c];
                    }
            o[((size_t)d * H + h) * W + w] = acc * scale;
        }
        return;
    }
    // rows of RB_SW gradients: thread -> (row slot, column)
    {
        constexpr int NROW = RB_SD * RB_SH, RPP = RB_THREADS / 64, NL = (NROW + RPP - 1) / RPP;   // 16 rows per pass: 64 lanes cover one 38-wide row
        const int c = tid & 63, rsub = tid >> 6;
        const __amdgpu_buffer_rsrc_t rg = vxm_rsrc(g, (unsigned)oD * (unsigned)oH * (unsigned)oW * 4u);
        const bool cok = c < RB_SW && bw + c < oW;
        float v[NL];                                                   // every load of the tile in flight before the first LDS write
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int row = RPP * u + rsub;                            // wave-uniform
            const int a = row / RB_SH, e = row - a * RB_SH;
            const bool ok = cok && row < NROW && bd + a < oD && bh + e < oH;
            v[u] = vxm_bload(rg, ok ? (((bd + a) * oH + bh + e) * oW + bw + c) << 2 : VXM_OOB, 0);
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int row = RPP * u + rsub;
            if (c < RB_SW && row < NROW) b0[row * RB_SW + c] = v[u];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < RB_SD * RB_SH * RB_W; idx += RB_THREADS) {            // contract W
        const int tx = idx % RB_W, row = idx / RB_W;
        const float* src = b0 + row * RB_SW + (solo[tx] - bw);
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < RS_KC; ++c) acc += sw[tx][c] * src[c];
        b1[idx] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < RB_SD * RB_H * RB_W; idx += RB_THREADS) {             // contract H -> b0 reused as [RB_SD][RB_H][RB_W]
        const int tx = idx % RB_W, ty = (idx / RB_W) % RB_H, a = idx / (RB_W * RB_H);
        const float* src = b1 + (a * RB_SH + (solo[RB_W + ty] - bh)) * RB_W + tx;
        float acc = 0.0f;
#pragma unroll
        for (int e = 0; e < RS_KC; ++e) acc += sw[RB_W + ty][e] * src[e * RB_W];
        b0[idx] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < RB_D * RB_H * RB_W; idx += RB_THREADS) {              // contract D
        const int tx = idx % RB_W, ty = (idx / RB_W) % RB_H, dd = idx / (RB_W * RB_H);
        const int w = w0 + tx, h = h0 + ty, d = d0 + dd;
        if (w >= W || h >= H || d >= D) continue;
        const float* src = b0 + ((solo[RB_W + RB_H + dd] - bd) * RB_H + ty) * RB_W + tx;
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < RS_KC; ++a) acc += sw[RB_W + RB_H + dd][a] * src[a * RB_H * RB_W];
        o[((size_t)d * H + h) * W + w] = acc * scale;
    }
}

// ---- `fullsize` fused into the final SpatialTransformer (networks.py:275-280: pos_flow = fullsize(integrate(v)); y = transformer(source, pos_flow)).
// The warp reads flow[p] only at its own voxel, so the upsampled displacement is computed in registers from the half-resolution field (10 MB:
// resident in the L2 / memory-side cache) and the full-resolution pos_flow (82.6 MB per pair) is never written or read.  A block owns a
// 4 x 8 x 32 tile of OUTPUT voxels: its 44 per-axis interpolation records (k_resize3d_fwd_tiled's) and the (<= 4 x 6 x 18 x 3) block of the
// low-resolution field its taps touch are staged in LDS once; a thread then evaluates rt_trilerp -- the resize kernel's expression --
// and the warp of k_warp3d_fwd (same coordinate arithmetic: bit-identical to the two-kernel path).  Ratios up to 0.51 (factor >= 2).
constexpr int WU_LZ = 4, WU_LY = 6, WU_LX = 18, WU_LN = WU_LZ * WU_LY * WU_LX;
struct WuRec { int z0, z1, y0, y1, x0, x1; float lz0, lz1, ly0, ly1, lx0, lx1; };

// stage the records and the low-resolution block of the tile at (d0, h0, w0); returns through LDS.  fl[c][.] holds pre * flow
__device__ __forceinline__ void wu_stage(const float* __restrict__ flo, int lD, int lH, int lW, int oD, int oH, int oW, float rd, float rh, float rw,
                                         float pre, int d0, int h0, int w0, int* si0, int* si1, float* sl0, float* sl1, float (*fl)[WU_LN]) {
    const int tid = threadIdx.x;
    if (tid < RT_N) {
        const int ax = tid < RT_W ? 2 : (tid < RT_W + RT_H ? 1 : 0);
        const int dst = ax == 2 ? w0 + tid : (ax == 1 ? h0 + tid - RT_W : d0 + tid - RT_W - RT_H);
        const int n_out = ax == 2 ? oW : (ax == 1 ? oH : oD), n_in = ax == 2 ? lW : (ax == 1 ? lH : lD);
        int i0, i1;
        float l0, l1;
        lin_src(min(dst, n_out - 1), ax == 2 ? rw : (ax == 1 ? rh : rd), n_in, i0, i1, l0, l1);
        si0[tid] = i0; si1[tid] = i1; sl0[tid] = l0; sl1[tid] = l1;
    }
    __syncthreads();
    const int xlo = si0[0], ylo = si0[RT_W], zlo = si0[RT_W + RT_H];
    const int lHW = lH * lW, lV = lD * lHW;
    const __amdgpu_buffer_rsrc_t rf = vxm_rsrc(flo, 3u * (unsigned)lV * 4u);
    for (int i = tid; i < 3 * WU_LN; i += 256) {
        const int c = i / WU_LN, r = i - c * WU_LN;
        const int z = r / (WU_LY * WU_LX), r2 = r - z * (WU_LY * WU_LX), y = r2 / WU_LX, x = r2 - y * WU_LX;
        const int gz = min(zlo + z, lD - 1), gy = min(ylo + y, lH - 1), gx = min(xlo + x, lW - 1);
        fl[c][r] = pre * vxm_bload(rf, (gz * lHW + gy * lW + gx) << 2, (c * lV) << 2);
    }
    __syncthreads();
}
// in-plane part of the upsampled displacement for this thread's (h, w): one bilinear value per channel and staged low-resolution plane
struct WuPlanes { float y[3][WU_LZ]; };
__device__ __forceinline__ WuPlanes wu_planes(const int* si0, const int* si1, const float* sl0, const float* sl1, const float (*fl)[WU_LN], int tx, int ty) {
    const int xlo = si0[0], ylo = si0[RT_W];
    const int x0 = si0[tx] - xlo, x1 = si1[tx] - xlo, y0 = si0[RT_W + ty] - ylo, y1 = si1[RT_W + ty] - ylo;
    const float lx0 = sl0[tx], lx1 = sl1[tx], ly0 = sl0[RT_W + ty], ly1 = sl1[RT_W + ty];
    WuPlanes p;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int z = 0; z < WU_LZ; ++z) {
            const float* q = fl[c] + z * (WU_LY * WU_LX);
            p.y[c][z] = rt_bilerp(ly0, ly1, lx0, lx1, q[y0 * WU_LX + x0], q[y0 * WU_LX + x1], q[y1 * WU_LX + x0], q[y1 * WU_LX + x1]);
        }
    return p;
}
// upsampled displacement (3 channels) of the output voxel at depth record dd: the depth interpolation of the plane values
__device__ __forceinline__ void wu_flow(const int* si0, const int* si1, const float* sl0, const float* sl1, const WuPlanes& p, int dd, float post,
                                        float& f0, float& f1, float& f2) {
    const int zlo = si0[RT_W + RT_H];
    const int z0 = si0[RT_W + RT_H + dd] - zlo, z1 = si1[RT_W + RT_H + dd] - zlo;
    const float lz0 = sl0[RT_W + RT_H + dd], lz1 = sl1[RT_W + RT_H + dd];
    float f[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = p.y[c][0], b = p.y[c][0];
#pragma unroll
        for (int z = 1; z < WU_LZ; ++z) { a = z0 == z ? p.y[c][z] : a; b = z1 == z ? p.y[c][z] : b; }      // (register select: no dynamic indexing)
        f[c] = post * rt_zlerp(lz0, lz1, a, b);
    }
    f0 = f[0]; f1 = f[1]; f2 = f[2];
}

template <int MODE>
__global__ void __launch_bounds__(256) k_warp3d_up_fwd(const float* __restrict__ src, const float* __restrict__ flo, float* __restrict__ out,
                                                       float* __restrict__ pos, int C, int D, int H, int W, int lD, int lH, int lW, float rd, float rh,
                                                       float rw, float pre, float post) {
    __shared__ int si0[RT_N], si1[RT_N];
    __shared__ float sl0[RT_N], sl1[RT_N];
    __shared__ float fl[3][WU_LN];
    int d0, h0, w0;
    rt_tile(W, H, d0, h0, w0);
    const int b = blockIdx.y, tid = threadIdx.x;
    const int V = D * H * W, V4 = V << 2;
    wu_stage(flo + (size_t)b * 3 * lD * lH * lW, lD, lH, lW, D, H, W, rd, rh, rw, pre, d0, h0, w0, si0, si1, sl0, sl1, fl);
    const int tx = tid & 31, ty = tid >> 5;
    const int w = w0 + tx, h = h0 + ty;
    if (w >= W || h >= H) return;
    const float* s = src + (size_t)b * C * V;
    float* o = out + (size_t)b * C * V;
    const __amdgpu_buffer_rsrc_t rp = vxm_rsrc(pos ? pos + (size_t)b * 3 * V : src, pos ? 3u * (unsigned)V * 4u : 0u);     // no pos: every store dropped
    const WuPlanes pl = wu_planes(si0, si1, sl0, sl1, fl, tx, ty);
#pragma unroll
    for (int dd = 0; dd < RT_D; ++dd) {
        const int d = d0 + dd;
        if (d >= D) break;
        float f0, f1, f2;
        wu_flow(si0, si1, sl0, sl1, pl, dd, post, f0, f1, f2);
        const int p4 = ((d * H + h) * W + w) << 2;
        vxm_bstore(f0, rp, p4, 0); vxm_bstore(f1, rp, p4, V4); vxm_bstore(f2, rp, p4, 2 * V4);
        const float z = vxm_src_coord(d, f0, D), y = vxm_src_coord(h, f1, H), x = vxm_src_coord(w, f2, W);
        if (MODE == VXM_INTERP_NEAREST) {
            const float rz = rintf(z), ry = rintf(y), rx = rintf(x);
            const bool in = (rz >= 0.0f) & (rz <= (float)(D - 1)) & (ry >= 0.0f) & (ry <= (float)(H - 1)) & (rx >= 0.0f) & (rx <= (float)(W - 1));
            const int idx4 = in ? (((int)rz * H + (int)ry) * W + (int)rx) << 2 : VXM_OOB;
            for (int c = 0; c < C; ++c) {
                const __amdgpu_buffer_rsrc_t rs = vxm_rsrc(s + (size_t)c * V, (unsigned)V4), ro = vxm_rsrc(o + (size_t)c * V, (unsigned)V4);
                vxm_bstore(vxm_bload(rs, idx4, 0), ro, p4, 0);
            }
            continue;
        }
        const Corners8 cn = corners8(z, y, x, D, H, W);
        for (int c = 0; c < C; ++c) {
            const __amdgpu_buffer_rsrc_t rs = vxm_rsrc(s + (size_t)c * V, (unsigned)V4), ro = vxm_rsrc(o + (size_t)c * V, (unsigned)V4);
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = vxm_bload(rs, cn.idx[k] << 2, 0);
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k] * cn.w[k];
            vxm_bstore(acc, ro, p4, 0);
        }
    }
}

// backward of the fused kernel w.r.t. the DISPLACEMENT at full resolution: gpos [B,3,D,H,W] = d out / d pos_flow (k_warp3d_bwd's sum with
// the displacement recomputed from the low-resolution field); vxm_resize3d_bwd then carries it onto the low-resolution grid.  (The two
// are not one kernel: a block that contracts gpos for a tile of low-resolution voxels needs it on the tile grown by the tap span, 1.46 x
// the voxels for the shapes of this path -- recomputing a vector-ALU-bound sum costs more than the 165 MB round trip it would save.)
__global__ void __launch_bounds__(256) k_warp3d_up_bwd(const float* __restrict__ src, const float* __restrict__ flo, const float* __restrict__ gout,
                                                       float* __restrict__ gpos, int C, int D, int H, int W, int lD, int lH, int lW, float rd, float rh,
                                                       float rw, float pre, float post) {
    __shared__ int si0[RT_N], si1[RT_N];
    __shared__ float sl0[RT_N], sl1[RT_N];
    __shared__ float fl[3][WU_LN];
    int d0, h0, w0;
    rt_tile(W, H, d0, h0, w0);
    const int b = blockIdx.y, tid = threadIdx.x;
    const int V = D * H * W, V4 = V << 2;
    wu_stage(flo + (size_t)b * 3 * lD * lH * lW, lD, lH, lW, D, H, W, rd, rh, rw, pre, d0, h0, w0, si0, si1, sl0, sl1, fl);
    const int tx = tid & 31, ty = tid >> 5;
    const int w = w0 + tx, h = h0 + ty;
    if (w >= W || h >= H) return;
    const float* s = src + (size_t)b * C * V;
    const float* go = gout + (size_t)b * C * V;
    const __amdgpu_buffer_rsrc_t rg = vxm_rsrc(gpos + (size_t)b * 3 * V, 3u * (unsigned)V * 4u);
    const WuPlanes pl = wu_planes(si0, si1, sl0, sl1, fl, tx, ty);
#pragma unroll
    for (int dd = 0; dd < RT_D; ++dd) {
        const int d = d0 + dd;
        if (d >= D) break;
        float f0, f1, f2;
        wu_flow(si0, si1, sl0, sl1, pl, dd, post, f0, f1, f2);
        const int p4 = ((d * H + h) * W + w) << 2;
        const Corners8 cn = corners8(vxm_src_coord(d, f0, D), vxm_src_coord(h, f1, H), vxm_src_coord(w, f2, W), D, H, W);
        float gz = 0.0f, gy = 0.0f, gx = 0.0f;
        for (int c = 0; c < C; ++c) {
            const __amdgpu_buffer_rsrc_t rs = vxm_rsrc(s + (size_t)c * V, (unsigned)V4), rgo = vxm_rsrc(go + (size_t)c * V, (unsigned)V4);
            const float g = vxm_bload(rgo, p4, 0);
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = vxm_bload(rs, cn.idx[k] << 2, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dz = (k >> 2) & 1, dy = (k >> 1) & 1, dx = k & 1;
                const float vk = cn.ok[k] ? v[k] * g : 0.0f;
                gz += (dz ? vk : -vk) * (cn.wy[dy] * cn.wx[dx]);
                gy += (dy ? vk : -vk) * (cn.wz[dz] * cn.wx[dx]);
                gx += (dx ? vk : -vk) * (cn.wz[dz] * cn.wy[dy]);
            }
        }
        vxm_bstore(gz, rg, p4, 0); vxm_bstore(gy, rg, p4, V4); vxm_bstore(gx, rg, p4, 2 * V4);
    }
}

// ---- adjoint of an UP-sampling resize, marching form (round 6; k_resize3d_bwd_sep ran at 0.8 TB/s: 1024-thread blocks, three block-wide
// contractions of a 14 x 22 x 38 block of gradients per 4 x 8 x 16 tile, one tile in flight per CU).  A block of 256 threads owns an
// 8 x 32 (H x W) tile of INPUT voxels -- one per thread -- over a segment of input depths and marches through the output planes that reach
// it: plane od contributes to the two input depths lin_src(od) names, so a thread keeps two running sums and retires one whenever the
// lower depth advances.  Per plane: the (<= 21 x 75) gradients the tile's taps touch go to LDS (requested one plane ahead), are contracted
// along W (5 taps), then along H (5 taps), and the thread adds its value to its sums -- each gradient is read once per block, the depth
// overlap between segments is 3 - 4 planes in ~ 40.  Same weights (axis_taps: the forward's index / lambda arithmetic) and the same
// W -> H -> D, ascending-tap summation order as the separable kernel; deterministic.  Ratios in [0.45, 0.75).
constexpr int RM_H = 8, RM_W = 32, RM_N = RM_W + RM_H, RM_T = 5, RM_SH = 21, RM_SW = 75, RM_NL = (RM_SH * RM_SW + 255) / 256;
__global__ void __launch_bounds__(256) k_resize3d_bwd_march(const float* __restrict__ gout, float* __restrict__ gx, int D, int H, int W, int oD, int oH,
                                                            int oW, float rd, float rh, float rw, float scale, int seg_len) {
    __shared__ float G[RM_SH * RM_SW];
    __shared__ float T[RM_SH * RM_W];
    __shared__ float sw[RM_N][RM_T + 1];
    __shared__ int so[RM_N];
    const int tw = (W + RM_W - 1) / RM_W, th = (H + RM_H - 1) / RM_H;
    int t = blockIdx.x;
    const int w0 = (t % tw) * RM_W; t /= tw;
    const int h0 = (t % th) * RM_H;
    const int dlo = (t / th) * seg_len, dhi = min(D, dlo + seg_len);
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    if (tid < RM_N) {
        const int ax = tid < RM_W ? 2 : 1;
        const int i = ax == 2 ? w0 + tid : h0 + tid - RM_W;
        const int n_in = ax == 2 ? W : H, n_out = ax == 2 ? oW : oH;
        int olo;
        float wt[RS_KC];
        axis_taps(min(i, n_in - 1), ax == 2 ? rw : rh, n_in, n_out, olo, wt);
        int klo = 0;
        while (klo < RS_KC - RM_T && wt[klo] == 0.0f) ++klo;                   // first tap that counts; at most RM_T of them (ratio >= 0.4)
#pragma unroll
        for (int k = 0; k < RM_T; ++k) {
            float v = 0.0f;
#pragma unroll
            for (int j = 0; j < RS_KC; ++j) v = (j == klo + k) ? wt[j] : v;
            sw[tid][k] = i < n_in ? v : 0.0f;
        }
        so[tid] = olo + klo;
    }
    __syncthreads();
    const int bw = so[0], bh = so[RM_W];
    const int nw = min(RM_SW, oW - bw), nh = min(RM_SH, oH - bh);              // the block of output gradients the tile can touch (clipped)
    const size_t bc = blockIdx.y;
    const __amdgpu_buffer_rsrc_t rg = vxm_rsrc(gout + bc * (size_t)oD * oH * oW, (unsigned)oD * (unsigned)oH * (unsigned)oW * 4u);
    // output planes that reach the input depths [dlo, dhi): those whose lower source depth is in [dlo - 1, dhi - 1]
    int od0 = rd > 0.0f ? max(0, (int)floorf((float)(dlo - 1) / rd) - 1) : 0;
    {
        int i0, i1; float l0, l1;
        for (;; ++od0) { lin_src(min(od0, oD - 1), rd, D, i0, i1, l0, l1); if (i1 >= dlo || od0 >= oD - 1) break; }
    }
    // THREE planes in flight per thread (the body of a plane is ~500 cycles, the latency of its loads several times that): register sets
    // rotate through an unrolled-by-three plane loop
    constexpr int RM_PF = 3;
    float v[RM_PF][RM_NL];
    int voff[RM_NL];
#pragma unroll
    for (int u = 0; u < RM_NL; ++u) {
        const int i = tid + 256 * u, r = i / RM_SW, c = i - r * RM_SW;
        voff[u] = (r < nh && c < nw && i < RM_SH * RM_SW) ? ((bh + r) * oW + bw + c) << 2 : VXM_OOB;
    }
    const int plane4 = (oH * oW) << 2;
    auto request = [&](auto set_, int od) __attribute__((always_inline)) {
        constexpr int S = decltype(set_)::value;
        const int gate = od < oD ? 0 : VXM_OOB;                                 // wave-uniform: past the last plane nothing is read
#pragma unroll
        for (int u = 0; u < RM_NL; ++u) v[S][u] = vxm_bload(rg, voff[u] | gate, od < oD ? od * plane4 : 0);
    };
    const int ch = so[RM_W + ty] - bh;
    float wH[RM_T];
#pragma unroll
    for (int k = 0; k < RM_T; ++k) wH[k] = sw[RM_W + ty][k];
    float* o = gx + bc * (size_t)D * H * W;
    const bool own = h0 + ty < H && w0 + tx < W;
    int dcur = dlo - 1;                                                        // depth accA belongs to; accB: dcur + 1
    float accA = 0.0f, accB = 0.0f;
    bool done = false;
    auto plane = [&](auto set_, int od) __attribute__((always_inline)) {
        constexpr int S = decltype(set_)::value;
        if (done || od >= oD) { done = true; return; }
        int i0, i1;
        float l0, l1;
        lin_src(od, rd, D, i0, i1, l0, l1);
        if (i0 >= dhi) { done = true; return; }                                 // block-uniform
#pragma unroll
        for (int u = 0; u < RM_NL; ++u) {
            const int i = tid + 256 * u;
            if (i < RM_SH * RM_SW) G[i] = v[S][u];
        }
        request(set_, od + RM_PF);
        __syncthreads();
        for (int i = tid; i < RM_SH * RM_W; i += 256) {                         // contract W: row r, input column i & 31
            const int r = i >> 5, x = i & 31;
            const int c0 = so[x] - bw;                                          // (taps past the clipped block have weight zero: clamped, not read out of bounds)
            const float* g = G + r * RM_SW;
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < RM_T; ++k) a += sw[x][k] * g[min(c0 + k, RM_SW - 1)];
            T[i] = a;
        }
        __syncthreads();
        float val = 0.0f;
#pragma unroll
        for (int k = 0; k < RM_T; ++k) val += wH[k] * T[min(ch + k, RM_SH - 1) * RM_W + tx];
        while (dcur < i0) {                                                     // the lower source depth advanced: accA is complete
            if (own && dcur >= dlo) o[((size_t)dcur * H + h0 + ty) * W + w0 + tx] = accA * scale;
            accA = accB; accB = 0.0f; ++dcur;
        }
        // dcur == i0 here (i0 >= dlo - 1 by the choice of od0)
        if (i1 == i0) accA += (l0 + l1) * val;
        else { accA += l0 * val; accB += l1 * val; }
    };
    using J0 = std::integral_constant<int, 0>;
    using J1 = std::integral_constant<int, 1>;
    using J2 = std::integral_constant<int, 2>;
    request(J0{}, od0); request(J1{}, od0 + 1); request(J2{}, od0 + 2);
    for (int od = od0; !done; od += RM_PF) {
        plane(J0{}, od);
        plane(J1{}, od + 1);
        plane(J2{}, od + 2);
    }
    for (; dcur < dhi; ++dcur) {                                                // retire what is left
        if (own && dcur >= dlo) o[((size_t)dcur * H + h0 + ty) * W + w0 + tx] = accA * scale;
        accA = accB; accB = 0.0f;
    }
}

int check_vol(const char* fn, int B, int C, int D, int H, int W) {
    VXM_REQUIRE(B > 0 && C > 0 && D > 1 && H > 1 && W > 1, VXM_ERR_BAD_SHAPE,
                "%s: bad shape B=%d C=%d D=%d H=%d W=%d (3-D volumes with every extent > 1)", fn, B, C, D, H, W);
    VXM_REQUIRE((long long)C * D * H * W < (1ll << 31) && (long long)D * H * W < (1ll << 28) && B <= 65535 && D <= 65535, VXM_ERR_BAD_SHAPE,
                "%s: per-sample element count must fit int32 (32-bit byte offsets within 3 planes: fewer than 2^28 voxels), B and D <= 65535", fn);
    return VXM_OK;
}

}  // namespace

extern "C" {

int vxm_warp3d_fwd(const float* src, const float* flow, float* out, int B, int C, int D, int H, int W, int mode,
                   void* stream) {
    if (int e = check_vol("vxm_warp3d_fwd", B, C, D, H, W)) return e;
    VXM_REQUIRE(src && flow && out, VXM_ERR_NULL_POINTER, "vxm_warp3d_fwd: null pointer");
    VXM_REQUIRE(mode == VXM_INTERP_LINEAR || mode == VXM_INTERP_NEAREST, VXM_ERR_UNSUPPORTED,
                "vxm_warp3d_fwd: mode %d (only 'bilinear' and 'nearest', layers.py:11)", mode);
    const dim3 grid(vxm_blocks((long long)H * W, 256), D, B);
    if (mode == VXM_INTERP_NEAREST)
        hipLaunchKernelGGL(k_warp3d_fwd<VXM_INTERP_NEAREST>, grid, dim3(256), 0, VXM_STREAM(stream), src, flow, out, C, D, H, W);
    else
        hipLaunchKernelGGL(k_warp3d_fwd<VXM_INTERP_LINEAR>, grid, dim3(256), 0, VXM_STREAM(stream), src, flow, out, C, D, H, W);
    return vxm_check_launch("vxm_warp3d_fwd");
}

int vxm_warp3d_bwd(const float* src, const float* flow, const float* gout, float* gsrc, float* gflow, int B, int C,
                   int D, int H, int W, int mode, void* stream) {
    if (int e = check_vol("vxm_warp3d_bwd", B, C, D, H, W)) return e;
    VXM_REQUIRE(src && flow && gout, VXM_ERR_NULL_POINTER, "vxm_warp3d_bwd: null pointer");
    VXM_REQUIRE(mode == VXM_INTERP_LINEAR || mode == VXM_INTERP_NEAREST, VXM_ERR_UNSUPPORTED, "vxm_warp3d_bwd: mode %d", mode);
    if (!gsrc && !gflow) return VXM_OK;
    const size_t V = (size_t)D * H * W;
    if (gsrc) hipMemsetAsync(gsrc, 0, sizeof(float) * B * C * V, VXM_STREAM(stream));
    const dim3 grid(vxm_blocks((long long)H * W, 256), D, B);
    if (mode == VXM_INTERP_NEAREST)
        hipLaunchKernelGGL(k_warp3d_bwd<VXM_INTERP_NEAREST>, grid, dim3(256), 0, VXM_STREAM(stream), src, flow, gout, gsrc, gflow, C, D, H, W);
    else
        hipLaunchKernelGGL(k_warp3d_bwd<VXM_INTERP_LINEAR>, grid, dim3(256), 0, VXM_STREAM(stream), src, flow, gout, gsrc, gflow, C, D, H, W);
    return vxm_check_launch("vxm_warp3d_bwd");
}

static int warp_up_args(const char* fn, int B, int C, int D, int H, int W, int lD, int lH, int lW, float factor, float& rd, float& rh, float& rw) {
    if (int e = check_vol(fn, B, C, D, H, W)) return e;
    if (int e = check_vol(fn, B, 3, lD, lH, lW)) return e;
    VXM_REQUIRE(factor > 0.0f, VXM_ERR_BAD_SHAPE, "%s: factor %g", fn, factor);
    rd = (float)(lD - 1) / (float)(D - 1); rh = (float)(lH - 1) / (float)(H - 1); rw = (float)(lW - 1) / (float)(W - 1);
    VXM_REQUIRE(rd <= 0.51f && rh <= 0.51f && rw <= 0.51f, VXM_ERR_UNSUPPORTED,
                "%s: the fused kernel stages a 4 x 6 x 18 block of the low-resolution field per 4 x 8 x 32 output tile: it takes fields at most about "
                "half as fine as the image per axis (%dx%dx%d -> %dx%dx%d); resize and warp separately otherwise (vxm_warp3d_up_ok)", fn, lD, lH, lW, D, H, W);
    return VXM_OK;
}

int vxm_warp3d_up_ok(int D, int H, int W, int lD, int lH, int lW) {
    if (D < 2 || H < 2 || W < 2 || lD < 2 || lH < 2 || lW < 2) return 0;
    return (float)(lD - 1) / (float)(D - 1) <= 0.51f && (float)(lH - 1) / (float)(H - 1) <= 0.51f && (float)(lW - 1) / (float)(W - 1) <= 0.51f;
}

int vxm_warp3d_up_fwd(const float* src, const float* flow_lo, float* out, float* pos_flow, int B, int C, int D, int H, int W, int lD, int lH, int lW,
                      float factor, int mode, void* stream) {
    float rd, rh, rw;
    if (int e = warp_up_args("vxm_warp3d_up_fwd", B, C, D, H, W, lD, lH, lW, factor, rd, rh, rw)) return e;
    VXM_REQUIRE(src && flow_lo && out, VXM_ERR_NULL_POINTER, "vxm_warp3d_up_fwd: null pointer");
    VXM_REQUIRE(mode == VXM_INTERP_LINEAR || mode == VXM_INTERP_NEAREST, VXM_ERR_UNSUPPORTED, "vxm_warp3d_up_fwd: mode %d", mode);
    const float pre = factor > 1.0f ? factor : 1.0f, post = factor < 1.0f ? factor : 1.0f;
    const long long tiles = (long long)((W + RT_W - 1) / RT_W) * ((H + RT_H - 1) / RT_H) * ((D + RT_D - 1) / RT_D);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_warp3d_up_fwd: too many tiles");
    if (mode == VXM_INTERP_NEAREST)
        hipLaunchKernelGGL(k_warp3d_up_fwd<VXM_INTERP_NEAREST>, dim3((unsigned)tiles, B), dim3(256), 0, VXM_STREAM(stream), src, flow_lo, out, pos_flow,
                           C, D, H, W, lD, lH, lW, rd, rh, rw, pre, post);
    else
        hipLaunchKernelGGL(k_warp3d_up_fwd<VXM_INTERP_LINEAR>, dim3((unsigned)tiles, B), dim3(256), 0, VXM_STREAM(stream), src, flow_lo, out, pos_flow,
                           C, D, H, W, lD, lH, lW, rd, rh, rw, pre, post);
    return vxm_check_launch("vxm_warp3d_up_fwd");
}

int vxm_warp3d_up_bwd(const float* src, const float* flow_lo, const float* gout, float* gflow_lo, float* work, size_t work_bytes, int B, int C, int D,
                      int H, int W, int lD, int lH, int lW, float factor, int mode, void* stream) {
    float rd, rh, rw;
    if (int e = warp_up_args("vxm_warp3d_up_bwd", B, C, D, H, W, lD, lH, lW, factor, rd, rh, rw)) return e;
    VXM_REQUIRE(src && flow_lo && gout && gflow_lo && work, VXM_ERR_NULL_POINTER, "vxm_warp3d_up_bwd: null pointer");
    VXM_REQUIRE(mode == VXM_INTERP_LINEAR || mode == VXM_INTERP_NEAREST, VXM_ERR_UNSUPPORTED, "vxm_warp3d_up_bwd: mode %d", mode);
    const size_t nfull = (size_t)B * 3 * D * H * W;
    VXM_REQUIRE(work_bytes >= nfull * sizeof(float), VXM_ERR_WORKSPACE, "vxm_warp3d_up_bwd: work holds %zu bytes, needs %zu (B x 3 x D x H x W floats)",
                work_bytes, nfull * sizeof(float));
    if (mode == VXM_INTERP_NEAREST) {            // nearest sampling has no gradient w.r.t. the displacement (k_warp3d_bwd writes zeros)
        (void)hipMemsetAsync(gflow_lo, 0, sizeof(float) * (size_t)B * 3 * lD * lH * lW, VXM_STREAM(stream));
        return vxm_check_launch("vxm_warp3d_up_bwd");
    }
    const float pre = factor > 1.0f ? factor : 1.0f, post = factor < 1.0f ? factor : 1.0f;
    const long long tiles = (long long)((W + RT_W - 1) / RT_W) * ((H + RT_H - 1) / RT_H) * ((D + RT_D - 1) / RT_D);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_warp3d_up_bwd: too many tiles");
    hipLaunchKernelGGL(k_warp3d_up_bwd, dim3((unsigned)tiles, B), dim3(256), 0, VXM_STREAM(stream), src, flow_lo, gout, work, C, D, H, W, lD, lH, lW,
                       rd, rh, rw, pre, post);
    if (int e = vxm_check_launch("vxm_warp3d_up_bwd")) return e;
    return vxm_resize3d_bwd(work, gflow_lo, B, 3, lD, lH, lW, D, H, W, factor, stream);
}

int vxm_vecint_fwd(const float* vec, float* steps, int B, int D, int H, int W, int nsteps, void* stream) {
    if (int e = check_vol("vxm_vecint_fwd", B, 3, D, H, W)) return e;
    VXM_REQUIRE(nsteps >= 1 && nsteps < 31, VXM_ERR_BAD_SHAPE, "vxm_vecint_fwd: nsteps should be >= 1, found: %d", nsteps);
    VXM_REQUIRE(vec && steps, VXM_ERR_NULL_POINTER, "vxm_vecint_fwd: null pointer");
    const size_t n = (size_t)B * 3 * D * H * W;
    const dim3 grid(vxm_blocks((long long)H * W, 256), D, B);
    const float scale = 1.0f / (float)(1u << nsteps);
    for (int k = 0; k < nsteps; ++k) {
        const float* in = k == 0 ? vec : steps + (size_t)(k - 1) * n;
        hipLaunchKernelGGL(k_vecint_step_fwd, grid, dim3(256), 0, VXM_STREAM(stream), in, k == 0 ? scale : 1.0f,
                           steps + (size_t)k * n, D, H, W);
    }
    return vxm_check_launch("vxm_vecint_fwd");
}

int vxm_vecint_bwd_ws(const float* vec, const float* steps, const float* gout, float* gvec, float* work, size_t work_bytes, int B, int D,
                      int H, int W, int nsteps, void* stream) {
    if (int e = check_vol("vxm_vecint_bwd", B, 3, D, H, W)) return e;
    VXM_REQUIRE(nsteps >= 1 && nsteps < 31, VXM_ERR_BAD_SHAPE, "vxm_vecint_bwd: nsteps should be >= 1, found: %d", nsteps);
    VXM_REQUIRE(vec && steps && gout && gvec && work, VXM_ERR_NULL_POINTER, "vxm_vecint_bwd: null pointer");
    const size_t n = (size_t)B * 3 * D * H * W;
    VXM_REQUIRE(work_bytes >= (2 * n + VXM_VECINT_WORK_EXTRA) * sizeof(float), VXM_ERR_WORKSPACE,
                "vxm_vecint_bwd: work holds %zu bytes, needs at least %zu (2 x B x 3 x D x H x W + VXM_VECINT_WORK_EXTRA floats; "
                "vxm_workspace_bytes(VXM_WS_VECINT_BWD, ...) also covers the per-tile displacement records)", work_bytes,
                (2 * n + VXM_VECINT_WORK_EXTRA) * sizeof(float));
    const long long tiles = (long long)((W + VG_TW - 1) / VG_TW) * ((H + VG_TH - 1) / VG_TH) * ((D + VG_TD - 1) / VG_TD);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_vecint_bwd: too many tiles");
    const dim3 grid_t((unsigned)tiles, B);
    const unsigned far_blocks = (unsigned)(tiles * B < 512 ? tiles * B : 512);
    const float scale = 1.0f / (float)(1u << nsteps);
    // per-step statistics of the senders the gather leaves to the far pass live behind the two gradient buffers: {count, largest
    // displacement, largest |g|} (float bits) per step
    unsigned* far = reinterpret_cast<unsigned*>(work + 2 * n);
    (void)hipMemsetAsync(far, 0, sizeof(unsigned) * VXM_VECINT_WORK_EXTRA, VXM_STREAM(stream));
    // per-tile largest displacement of each step's far senders: [nsteps][B][tiles] words behind the statistics, when the caller's scratch holds
    // them (vxm_workspace_bytes(VXM_WS_VECINT_BWD) asks for it; the ABI-0.4 size falls back to the step's global maximum as the radius)
    const size_t rec_words = (size_t)nsteps * B * (size_t)tiles;
    unsigned* const tdm = work_bytes >= (2 * n + VXM_VECINT_WORK_EXTRA + rec_words) * sizeof(float) ? far + VXM_VECINT_WORK_EXTRA : nullptr;
    const float* g = gout;
    for (int k = nsteps - 1; k >= 0; --k) {
        const float* in = k == 0 ? vec : steps + (size_t)(k - 1) * n;
        float* gn = k == 0 ? gvec : work + (size_t)(k & 1) * n;
        const float sc = k == 0 ? scale : 1.0f;
        unsigned* const tdm_k = tdm ? tdm + (size_t)k * B * (size_t)tiles : nullptr;
        hipLaunchKernelGGL(k_vecint_step_bwd_gather, grid_t, dim3(1024), 0, VXM_STREAM(stream), in, sc, g, gn, far + 4 * k, tdm_k, D, H, W);
        // ONE far launch per step (rounds 3-4: two, 4.8 us each when they had nothing to do): a persistent grid of at most two blocks per CU
        // that exits at once when the gather counted no far sender, walks the output tiles otherwise
        hipLaunchKernelGGL(k_vecint_step_bwd_far_tiles, dim3(far_blocks), dim3(1024), 0, VXM_STREAM(stream), in, sc, g, gn, far + 4 * k, tdm_k, B, D, H, W);
        g = gn;
    }
    return vxm_check_launch("vxm_vecint_bwd");
}

int vxm_vecint_bwd(const float* vec, const float* steps, const float* gout, float* gvec, float* work, int B, int D,
                   int H, int W, int nsteps, void* stream) {
    return vxm_vecint_bwd_ws(vec, steps, gout, gvec, work, B > 0 && D > 0 && H > 0 && W > 0 ? ((size_t)2 * B * 3 * D * H * W + VXM_VECINT_WORK_EXTRA) * sizeof(float) : 0,
                             B, D, H, W, nsteps, stream);
}

static int resize_args(const char* fn, int B, int C, int D, int H, int W, int oD, int oH, int oW, float factor) {
    if (int e = check_vol(fn, B, C, D, H, W)) return e;
    VXM_REQUIRE(oD > 0 && oH > 0 && oW > 0 && factor > 0.0f, VXM_ERR_BAD_SHAPE, "%s: bad output shape %dx%dx%d / factor %g", fn, oD, oH, oW, factor);
    VXM_REQUIRE((long long)B * C <= 65535, VXM_ERR_BAD_SHAPE, "%s: B*C must be <= 65535", fn);
    return VXM_OK;
}

int vxm_resize3d_fwd(const float* x, float* out, int B, int C, int D, int H, int W, int oD, int oH, int oW, float factor,
                     void* stream) {
    if (int e = resize_args("vxm_resize3d_fwd", B, C, D, H, W, oD, oH, oW, factor)) return e;
    VXM_REQUIRE(x && out, VXM_ERR_NULL_POINTER, "vxm_resize3d_fwd: null pointer");
    const float rd = oD > 1 ? (float)(D - 1) / (float)(oD - 1) : 0.0f, rh = oH > 1 ? (float)(H - 1) / (float)(oH - 1) : 0.0f,
                rw = oW > 1 ? (float)(W - 1) / (float)(oW - 1) : 0.0f;
    const float pre = factor > 1.0f ? factor : 1.0f, post = factor < 1.0f ? factor : 1.0f;
    const long long tiles = (long long)((oW + RT_W - 1) / RT_W) * ((oH + RT_H - 1) / RT_H) * ((oD + RT_D - 1) / RT_D);
    VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_resize3d_fwd: too many tiles");
    hipLaunchKernelGGL(k_resize3d_fwd_tiled, dim3((unsigned)tiles, B * C), dim3(256), 0, VXM_STREAM(stream),
                       x, out, D, H, W, oD, oH, oW, rd, rh, rw, pre, post);
    return vxm_check_launch("vxm_resize3d_fwd");
}

int vxm_resize3d_bwd(const float* gout, float* gx, int B, int C, int D, int H, int W, int oD, int oH, int oW, float factor,
                     void* stream) {
    if (int e = resize_args("vxm_resize3d_bwd", B, C, D, H, W, oD, oH, oW, factor)) return e;
    VXM_REQUIRE(gout && gx, VXM_ERR_NULL_POINTER, "vxm_resize3d_bwd: null pointer");
    const float rd = oD > 1 ? (float)(D - 1) / (float)(oD - 1) : 0.0f, rh = oH > 1 ? (float)(H - 1) / (float)(oH - 1) : 0.0f,
                rw = oW > 1 ? (float)(W - 1) / (float)(oW - 1) : 0.0f;
    const float rmin = fminf(oD > 1 ? rd : 1.0f, fminf(oH > 1 ? rh : 1.0f, oW > 1 ? rw : 1.0f));
    const float rmax = fmaxf(rd, fmaxf(rh, rw));
    static const bool use_march = [] { const char* e = getenv("VXM_RESIZE_BWD"); return !(e && e[0] == 's'); }();      // VXM_RESIZE_BWD=sep: the round-3 kernel (A/B)
    if (use_march && rmin >= 0.47f && rmax < 0.75f) {      // up-sampling by ~2: one thread per input voxel marching through the output planes
        const int tiles2 = ((W + RM_W - 1) / RM_W) * ((H + RM_H - 1) / RM_H);
        int nseg = (1024 + tiles2 * B * C - 1) / (tiles2 * B * C);          // ~1000 blocks; a segment re-reads 3 - 4 output planes of its neighbour
        nseg = nseg < 1 ? 1 : nseg;
        int seg_len = (D + nseg - 1) / nseg;
        seg_len = seg_len < 4 ? (D < 4 ? D : 4) : seg_len;
        nseg = (D + seg_len - 1) / seg_len;
        hipLaunchKernelGGL(k_resize3d_bwd_march, dim3((unsigned)(tiles2 * nseg), B * C), dim3(256), 0, VXM_STREAM(stream),
                           gout, gx, D, H, W, oD, oH, oW, rd, rh, rw, factor, seg_len);
    } else if (rmin > 0.4f && rmax < 0.75f) {      // separable passes through LDS
        const long long tiles = (long long)((W + RB_W - 1) / RB_W) * ((H + RB_H - 1) / RB_H) * ((D + RB_D - 1) / RB_D);
        VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_resize3d_bwd: too many tiles");
        constexpr int lds = (RB_SD * RB_SH * RB_SW + RB_SD * RB_SH * RB_W + RB_N * (RS_KC + 1) + RB_N) * 4;
        static const bool attr = [] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_resize3d_bwd_sep), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            return true;
        }();
        (void)attr;
        hipLaunchKernelGGL(k_resize3d_bwd_sep, dim3((unsigned)tiles, B * C), dim3(RB_THREADS), lds, VXM_STREAM(stream),
                           gout, gx, D, H, W, oD, oH, oW, rd, rh, rw, factor);
    } else if (use_march && rmin >= 1.0f) {      // down-sampling: at most two outputs per axis name an input voxel
        const long long tiles = (long long)((W + RT_W - 1) / RT_W) * ((H + RT_H - 1) / RT_H) * ((D + 15) / 16);
        VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_resize3d_bwd: too many tiles");
        hipLaunchKernelGGL(k_resize3d_bwd_down, dim3((unsigned)tiles, B * C), dim3(256), 0, VXM_STREAM(stream),
                           gout, gx, D, H, W, oD, oH, oW, rd, rh, rw, factor);
    } else if (rmin > 0.4f) {       // floor(2/ratio) + 3 <= RS_KC: every contributing output is among the candidates
        const long long tiles = (long long)((W + RT_W - 1) / RT_W) * ((H + RT_H - 1) / RT_H) * ((D + RT_D - 1) / RT_D);
        VXM_REQUIRE(tiles < (1ll << 31), VXM_ERR_BAD_SHAPE, "vxm_resize3d_bwd: too many tiles");
        hipLaunchKernelGGL(k_resize3d_bwd_gather_tiled, dim3((unsigned)tiles, B * C), dim3(256), 0, VXM_STREAM(stream),
                           gout, gx, D, H, W, oD, oH, oW, rd, rh, rw, factor);
    } else {                 // very large upsampling factors: scatter with atomics
        (void)hipMemsetAsync(gx, 0, sizeof(float) * (size_t)B * C * D * H * W, VXM_STREAM(stream));
        hipLaunchKernelGGL(k_resize3d_bwd, dim3(vxm_blocks((long long)oD * oH * oW, 256), B * C), dim3(256), 0, VXM_STREAM(stream),
                           gout, gx, D, H, W, oD, oH, oW, rd, rh, rw, factor);
    }
    return vxm_check_launch("vxm_resize3d_bwd");
}

}  // extern "C"
