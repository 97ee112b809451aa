// Backward-weight (and bias) of the 3x3x3 convolutions of the VxmDense U-Net on gfx950 -- convolution_backward w.r.t.
// weight / bias of voxelmorph/torch/networks.py:299-305, 211 -- as fp32 MFMA implicit GEMMs (see conv_fwd.hip for the
// forward / backward-data side and DESIGN.md section 4.1 for the measurements behind the design).
#include "conv_common.h"
#include "s3_pieces.h"

namespace {

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

// K-split variant for blocks with at most 8 N-tiles (2 input channels: 4 tiles; the role-swapped 3-channel flow conv: 6):
// instead of leaving most of the 16 waves without an N-tile, the 64 k-steps of a voxel tile are split into `ksplit` ranges
// and wave w takes N-tile w % (16 / ksplit) over range w / (16 / ksplit); the ranges' partial sums meet in LDS at the end
// (bw_ksplit_reduce).  LEN = 64 / ksplit k-steps; the range offset enters through boff / aoff (it is linear in the range).
template <int NCT, int RS, int PZ>
__device__ __forceinline__ void bw_ksteps_split(int len, bool active, bool bias, const float* Xb, const float* Zb, const int (&boff)[BW_SLOTS],
                                                int aoff, f32x4 (&acc)[BW_SLOTS][NCT], float (&bsum)[NCT]) {
    if (!active) return;
    if (bias) {
        switch (len) {
            case 32: bw_ksteps<NCT, 1, 0, 32, RS, PZ, true>(Xb, Zb, boff, aoff, acc, bsum); break;
            case 16: bw_ksteps<NCT, 1, 0, 16, RS, PZ, true>(Xb, Zb, boff, aoff, acc, bsum); break;
            default: bw_ksteps<NCT, 1, 0, 8, RS, PZ, true>(Xb, Zb, boff, aoff, acc, bsum); break;
        }
        return;
    }
    switch (len) {
        case 32: bw_ksteps<NCT, 1, 0, 32, RS, PZ, false>(Xb, Zb, boff, aoff, acc, bsum); break;
        case 16: bw_ksteps<NCT, 1, 0, 16, RS, PZ, false>(Xb, Zb, boff, aoff, acc, bsum); break;
        default: bw_ksteps<NCT, 1, 0, 8, RS, PZ, false>(Xb, Zb, boff, aoff, acc, bsum); break;
    }
}
// sum the k-ranges of every N-tile into the range-0 wave (fixed order: deterministic); scratch = the dead tile buffers
template <int NCT>
__device__ __forceinline__ void bw_ksplit_reduce(float* scratch, int wave, int lane, int tp, int ksplit, f32x4 (&acc)[BW_SLOTS][NCT], float (&bsum)[NCT]) {
    __syncthreads();                                  // every wave is done with the tile buffers
    float* mine = scratch + wave * (NCT * 5 * 64);
    if (wave >= tp) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
            for (int j = 0; j < 4; ++j) mine[(ct * 5 + j) * 64 + lane] = acc[0][ct][j];
            mine[(ct * 5 + 4) * 64 + lane] = bsum[ct];
        }
    }
    __syncthreads();
    if (wave < tp) {
        for (int kp = 1; kp < ksplit; ++kp) {
            const float* o = scratch + (wave + kp * tp) * (NCT * 5 * 64);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0][ct][j] += o[(ct * 5 + j) * 64 + lane];
                bsum[ct] += o[(ct * 5 + 4) * 64 + lane];
            }
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

    // N-tile j of this wave (slot i): entries e = j*16 + n  ->  (tap = e / ckc, channel = e % ckc).  With at most 8 N-tiles
    // the k-steps are split instead (see bw_ksteps_split): tp N-tiles x ksplit ranges of klen k-steps.
    const int tp = k.ntile <= 2 ? 2 : (k.ntile <= 4 ? 4 : (k.ntile <= 8 ? 8 : BW_WAVES));
    const int ksplit = BW_WAVES / tp, klen = 64 / ksplit;
    const int wtile = ksplit > 1 ? wave % tp : wave, kpart = ksplit > 1 ? wave / tp : 0;
    const int rowbase = (klen * kpart) >> 2;          // first (dz, hy) row of this wave's k-range
    int boff[BW_SLOTS];
    const int nslots = ksplit > 1 ? (wtile < k.ntile ? 1 : 0) : (wave < k.ntile ? (k.ntile - wave + BW_WAVES - 1) / BW_WAVES : 0);     // wave-uniform
#pragma unroll
    for (int i = 0; i < BW_SLOTS; ++i) {
        const int e = (wtile + BW_WAVES * i) * 16 + n;
        int off = 2;
        if (e < k.nent) {
            const int t = e / k.ckc, cl = e - t * k.ckc;
            off = cl * BV_PSX + ((t / 9) * HH + (t / 3) % 3) * BV_RS + t % 3 + 1;       // column = wx + kw - 1 + 2
        }
        boff[i] = off + kq + ((rowbase >> 2) * HH + (rowbase & 3)) * BV_RS;              // + voxel k of the MFMA B operand (+ k-range)
    }
    const int aoff = n * BV_PZ + kq + 4 * klen * kpart;      // MFMA A operand: dZ[co = n][voxel 4s + kq]
    f32x4 acc[BW_SLOTS][NCT];
#pragma unroll
    for (int i = 0; i < BW_SLOTS; ++i)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool bias = want_bias && k.c0 == 0 && wtile == 0;      // wave-uniform: the wave(s) of N-tile 0 of each output-channel group
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
        if (ksplit > 1) bw_ksteps_split<NCT, BV_RS, BV_PZ>(klen, nslots > 0, bias, Xb, Xb + BW_CKI * BV_PSX, boff, aoff, acc, bsum);
        else bw_ksteps_n<NCT, 0, 64, BV_RS, BV_PZ>(nslots, bias, Xb, Xb + BW_CKI * BV_PSX, boff, aoff, acc, bsum);
        float* Xn = smem + ((iter + 1) & 1) * BUF;     // last read before the previous barrier
        if (more) store_tile(Xn, Xn + BW_CKI * BV_PSX);
        __syncthreads();
    }
    if (ksplit > 1) {
        bw_ksplit_reduce<NCT>(smem, wave, lane, tp, ksplit, acc, bsum);
        if (kpart > 0) return;
    }
    bw_write_partial<NCT>(k, part, Cout, Cin, wtile, lane, nslots, acc, bias && kpart == 0, bsum);
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
        // 8 independent loads in flight per thread (the partials of one element are 256 KB apart: latency, not bandwidth)
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int p = y;
        for (; p + 28 < nparts; p += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += part[(size_t)(p + 4 * u) * n + e];
        }
        for (; p < nparts; p += 4) s[0] += part[(size_t)p * n + e];
        s0 = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
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
__global__ void __launch_bounds__(1024) k_reduce_partials(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb,
                                                          int n, int Cin, int Cout, int T, int Qc, int G, int cog_size, int swapflip,
                                                          int gw_cin, int ci_off) {
    // 64 outputs x 16 slices of the partials per block, 8 loads in flight per thread (the partials of one element are a whole
    // partial-array apart: latency, not bandwidth); slices combined in a fixed tree (deterministic)
    __shared__ float red[16][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + x;
    const int ntot = n + (gb ? Cout : 0);
    const size_t stride = (size_t)n + Cout;
    float acc = 0.0f;
    if (i < ntot) {
        const int co = i < n ? i / (Cin * 27) : i - n, ci = i < n ? (i / 27) % Cin : 0;
        const int cb = Qc * G, combo = ci / BW_CKI + Qc * (co / cog_size);
        const int nparts = (T - combo + cb - 1) / cb;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int p = y;
        for (; p + 16 * 7 < nparts; p += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += part[(size_t)(p + 16 * u) * stride + i];
        }
        for (; p < nparts; p += 16) s[0] += part[(size_t)p * stride + i];
        acc = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
    red[y][x] = acc;
    __syncthreads();
    if (y == 0 && i < ntot) {
        const float t = (((red[0][x] + red[1][x]) + (red[2][x] + red[3][x])) + ((red[4][x] + red[5][x]) + (red[6][x] + red[7][x]))) +
                        (((red[8][x] + red[9][x]) + (red[10][x] + red[11][x])) + ((red[12][x] + red[13][x]) + (red[14][x] + red[15][x])));
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
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;          // four independent chains: four loads in flight per thread
    constexpr size_t STEP = (size_t)CS_SLICES * 256;
    for (int b = 0; b < B; ++b) {
        const float* p = dz + (size_t)b * dz_bs + (size_t)co * V;
        size_t i = (size_t)blockIdx.y * 256 + threadIdx.x;
        for (; i + 3 * STEP < V; i += 4 * STEP) { s0 += p[i]; s1 += p[i + STEP]; s2 += p[i + 2 * STEP]; s3 += p[i + 3 * STEP]; }
        for (; i < V; i += STEP) s0 += p[i];
    }
    float s = (s0 + s1) + (s2 + s3);
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

// ------------------------------------------------------------------------------------------
// backward-weight of the layers with 1-3 channels on one side (first block 2 -> 16, flow conv 16 -> 3)
// ------------------------------------------------------------------------------------------
// 2 % of the conv FLOPs, but in the general kernel above (16 waves per 4 x 4 x 16 tile, ~300 instructions of per-tile address
// arithmetic against a few dozen MFMAs) they took 0.75 ms per step at 40 TFLOP/s.  Here: R[m][(c, tap)] = sum_v P[m][v] S[c][v + off(tap)]
// with P the 16-channel tensor (A operand, M = 16, halo-free), S the tensor with cs <= 3 channels -- its 27 cs shifted copies are the
// N columns (NT = ceil(27 cs / 16) N-tiles), K = 4 voxels of a W row per v_mfma_f32_16x16x4_f32 (exact fp32).
//   first block (x has 2 channels: virtual concat of two 1-channel tensors):  P = dz, S = x, off(tap) = tap - 1,     gw[co = m][ci = c][tap]
//   flow conv (dz has 3 channels):                                            P = x,  S = dz, off(tap) = 1 - tap,   gw[co = c][ci = m][tap]
// plus, for the first case, the bias gradient as one more column against a plane of ones.  Persistent blocks of 4 waves walk
// 1 x 8 x 64 voxel tiles: the 16 P planes of a tile (dwordx4 loads, 32 KB) and the haloed S planes sit in LDS (plane stride 514 = 2 mod 32:
// the A reads of a half-wave hit 32 distinct banks), the next tile is in flight in registers, a wave owns two rows and walks them in 16
// K-steps of NT MFMAs with immediate offsets; per-block partials, reduced in a fixed order by k_fewch_reduce (deterministic).
constexpr int FC_TH = 8, FC_TW = 64, FC_THREADS = 256;
constexpr int FC_PSTRIDE = FC_TH * FC_TW + 2;                 // floats per P plane in LDS
constexpr int FC_SW = FC_TW + 4, FC_SPLANE = 3 * (FC_TH + 2) * FC_SW;     // haloed S rows hold w0 - 1 .. w0 + 64 (+ padding); planes of one channel
constexpr int fc_lds_floats(int cs) { return 16 * FC_PSTRIDE + (cs + 1) * FC_SPLANE; }      // + one plane set of ones (bias column)

struct FcIn {
    const float* P; long long p_bs;            // [B][16][D][H][W]
    const float* S0; long long s0_bs; int cs0;  // small operand: virtual concat of S0 (cs0 channels) and S1 (cs - cs0 channels)
    const float* S1; long long s1_bs;
    // k_fewch_bwd_weight_h<.., POOL>: P is the skip-branch gradient and the operand is formed from it on the fly (see the kernel)
    const float* gpool = nullptr; const unsigned short* code = nullptr; float slope = 1.0f;
};

template <int NT>
__global__ void __launch_bounds__(FC_THREADS, 2) k_fewch_bwd_weight(FcIn in, int cs, int flip, int with_ones, float* __restrict__ part, int B, int D,
                                                                 int H, int W) {
    VXM_DYN_SMEM(float, smem);
    float* const Ps = smem;                                  // [16][FC_PSTRIDE]: [row][w]
    float* const Ss = smem + 16 * FC_PSTRIDE;                // [cs + 1][3][FC_TH + 2][FC_SW], the last "channel" is all ones
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, idx = lane & 15;
    const int nw = (W + FC_TW - 1) / FC_TW, nh = (H + FC_TH - 1) / FC_TH;
    const int ntiles = B * D * nh * nw;
    const int V = D * H * W, HW = H * W;

    // column n = c * 27 + tap of N-tile nt -> LDS offset of S[c] shifted by the tap (haloed coordinates 0..2 per axis); columns
    // beyond 27 cs: the ones plane at tap (1, 1, 1) for column 27 cs (bias gradient), any valid address otherwise (dropped by the reducer)
    int sbase[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + idx;
        int c = n / 27, tap = n - c * 27;
        if (n >= 27 * cs) { c = cs; tap = 13; }
        int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        if (flip && n < 27 * cs) { kd = 2 - kd; kh = 2 - kh; kw = 2 - kw; }
        sbase[nt] = c * FC_SPLANE + (kd * (FC_TH + 2) + kh) * FC_SW + kw + kq;
    }
    for (int i = tid; i < FC_SPLANE; i += FC_THREADS) Ss[cs * FC_SPLANE + i] = 1.0f;      // written once: no tile touches it

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging roles, fixed per thread.  P: 16 planes x 8 rows x 16 float4: thread -> (plane tid >> 7, row, float4), 8 loads two planes apart
    // (the plane step is a wave-uniform scalar offset); S: one haloed plane is 10 x 66 floats = 3 slots per thread, the same three slots
    // for every (channel, depth plane) of the region, which again only moves the scalar offset.
    constexpr int NPV = 8, NSK = 3, NSV = 3 * 3 * NSK;          // cs <= 3 channels x 3 depth planes x 3 slots
    f32x4 pv[NPV];
    float sv[NSV];
    const int pm0 = tid >> 7, prow = (tid >> 4) & 7, pw4 = tid & 15;
    const int pdst = pm0 * FC_PSTRIDE + prow * FC_TW + 4 * pw4;
    int spr[NSK], spw[NSK];
#pragma unroll
    for (int k = 0; k < NSK; ++k) {
        const int e = tid + FC_THREADS * k;
        spr[k] = e / (FC_TW + 2); spw[k] = e - spr[k] * (FC_TW + 2);          // spr >= FC_TH + 2: no slot
    }
    auto tile_coords = [&](int t, int& b, int& d, int& h0, int& w0) __attribute__((always_inline)) {
        const int tw = t % nw; int q = t / nw;
        const int th = q % nh; q /= nh;
        d = q % D; b = q / D;
        h0 = th * FC_TH; w0 = tw * FC_TW;
    };
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        int b, d, h0, w0;
        tile_coords(t < ntiles ? t : 0, b, d, h0, w0);
        int dead = t < ntiles ? 0 : VXM_OOB;                     // past the last tile: every lane out of range (branch-free)
        asm volatile("" : "+v"(dead));
        const __amdgpu_buffer_rsrc_t rp = vxm_rsrc(in.P + (size_t)b * in.p_bs, 16u * (unsigned)V * 4u);
        const int gh = h0 + prow, gw = w0 + 4 * pw4;
        const int pvoff = ((gh < H && gw < W) ? (pm0 * V + d * HW + gh * W + gw) << 2 : VXM_OOB) | dead;   // W % 4 == 0: a float4 is inside or outside the row
#pragma unroll
        for (int j = 0; j < NPV; ++j) pv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, pvoff, (2 * j * V) << 2, 0));
        const __amdgpu_buffer_rsrc_t r0 = vxm_rsrc(in.S0 + (size_t)b * in.s0_bs, (unsigned)in.cs0 * (unsigned)V * 4u);
        const __amdgpu_buffer_rsrc_t r1 = vxm_rsrc(cs > in.cs0 ? in.S1 + (size_t)b * in.s1_bs : in.S0, (unsigned)(cs > in.cs0 ? cs - in.cs0 : in.cs0) * (unsigned)V * 4u);
        int svoff[NSK];
#pragma unroll
        for (int k = 0; k < NSK; ++k) {
            const int sh = h0 - 1 + spr[k], sw = w0 - 1 + spw[k];
            const bool ok = spr[k] < FC_TH + 2 && (unsigned)sh < (unsigned)H && (unsigned)sw < (unsigned)W;
            svoff[k] = (ok ? (sh * W + sw) << 2 : VXM_OOB) | dead;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int pd = 0; pd < 3; ++pd) {
                const int gd = d - 1 + pd;                       // wave-uniform
                const bool first = c < in.cs0;
                const __amdgpu_buffer_rsrc_t r = first ? r0 : r1;
                const int cc = first ? c : c - in.cs0;
                int bad = (c < cs && (unsigned)gd < (unsigned)D) ? 0 : VXM_OOB;
                asm volatile("" : "+v"(bad));
                const int soff = (c < cs && (unsigned)gd < (unsigned)D) ? (cc * V + gd * HW) << 2 : 0;
#pragma unroll
                for (int k = 0; k < NSK; ++k) sv[(c * 3 + pd) * NSK + k] = vxm_bload(r, svoff[k] | bad, soff);
            }
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPV; ++j) {
            float* dst = Ps + pdst + 2 * j * FC_PSTRIDE;         // 8-byte aligned (plane stride even)
            *reinterpret_cast<f32x2*>(dst) = (f32x2){pv[j].x, pv[j].y};
            *reinterpret_cast<f32x2*>(dst + 2) = (f32x2){pv[j].z, pv[j].w};
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int pd = 0; pd < 3; ++pd)
#pragma unroll
                for (int k = 0; k < NSK; ++k)
                    if (c < cs && spr[k] < FC_TH + 2) Ss[c * FC_SPLANE + (pd * (FC_TH + 2) + spr[k]) * FC_SW + spw[k]] = sv[(c * 3 + pd) * NSK + k];
    };

    int t = blockIdx.x;
    load_tile(t);
    store_tile();
    __syncthreads();
    for (; t < ntiles; t += gridDim.x) {
        load_tile(t + gridDim.x);                                // unconditional (past the end every lane is out of range)
        const int abase = idx * FC_PSTRIDE + kq;
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr;
#pragma unroll 4                                                 // (fully unrolled, the compiler hoists all 16 x (NT + 1) LDS reads: 256 VGPRs and spills)
            for (int j = 0; j < FC_TW / 4; ++j) {
                const float a = Ps[abase + row * FC_TW + 4 * j];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = vxm_mfma16(a, Ss[sbase[nt] + row * FC_SW + 4 * j], acc[nt]);
            }
        }
        __syncthreads();                                        // every wave is done reading this tile
        if (t + (int)gridDim.x < ntiles) store_tile();
        __syncthreads();
    }
    (void)with_ones;
    // ---- partials: part[block][16 m][NT * 16 columns]; the four waves of a block are summed through LDS in a fixed order
    float* const red = smem;                                   // [4][16][NT * 16]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * kq + r) * (NT * 16) + nt * 16 + idx] = acc[nt][r];
    __syncthreads();
    for (int i = tid; i < 16 * NT * 16; i += FC_THREADS)
        part[(size_t)blockIdx.x * (16 * NT * 16) + i] = (red[i] + red[16 * NT * 16 + i]) + (red[2 * 16 * NT * 16 + i] + red[3 * 16 * NT * 16 + i]);
}

// ---- the same product on the 16-bit matrix pipe (round 6): fp16 pieces of both operands, three piece products (the "f16x2" scheme of
// conv_s3.hip / s3_pieces.h: x s = h + l up to 2^-22, one power-of-two scale per staged tile and operand, per-tile MFMA chains folded into
// fp32 totals by the vector ALU).  The fp32-MFMA kernel above spends 32 cycles per K = 4 voxels (62 % of its time with the matrix pipe busy:
// 0.19 + 0.26 ms per step for 0.5 GB of operands each, and the first layer's launch is the last thing the step waits for); with K = 32
// voxels per 16-cycle MFMA the three products cost a fifth of that.
//   Tile 1 x 8 x 32 voxels, 4 waves, a wave owns two rows = two K-steps; three blocks per CU (31 KB of LDS each) hide each other's loads.
// P (A operand): [piece][16 planes][8 rows x 32 + 8] halves -- lane (m, kq) reads the 8 consecutive voxels 8 kq .. 8 kq + 7 of its row with
// one aligned ds_read_b128.  S (B operand): column n = (c, tap) wants 8 consecutive voxels SHIFTED by the tap: kd and kh move the plane /
// row, kw = 0 / 1 / 2 halves would misalign the 16-byte read -- so a lane reads the aligned 12 halves 8 kq .. 8 kq + 11 of the haloed row
// (b128 + b64) and funnel-shifts them by ITS kw in registers (v_alignbit with a per-lane shift: 2 vector instructions per dword).
// (A first version kept three kw-shifted copies of S in LDS: 108 two-byte LDS writes per thread and tile, 77 KB, one block per CU -- slower
// than the fp32 kernel.)  Partials in the fp32 kernel's format (k_fewch_reduce is shared); the bias gradient (first layer) from MFMAs
// against a register of ones.
constexpr int FH_TH = 8, FH_TW = 32, FH_WAVES = 4, FH_THREADS = 64 * FH_WAVES;
constexpr int FH_PST = FH_TH * FH_TW + 8;                     // halves per P plane (528 bytes: the 16 planes of a fragment read spread over the banks)
constexpr int FH_PPIECE = 16 * FH_PST * 2;                    // bytes of one piece of P
constexpr int FH_SRW = FH_TW + 8, FH_SROWS = FH_TH + 2;       // halves per haloed S row (34 used); haloed rows
constexpr int FH_SPLANE = FH_SROWS * FH_SRW * 2;              // bytes of one (c, plane) block
constexpr int FH_SPIECE = 3 * 3 * FH_SPLANE;                  // bytes of one piece of S (cs <= 3)
constexpr int FH_TAB = 64;                                    // floats: wave maxima of P [0..3] and S [4..7]
constexpr int FH_STAGE = 2 * FH_PPIECE + 2 * FH_SPIECE + FH_TAB * 4;
constexpr int FH_LDS = FH_STAGE > FH_WAVES * 16 * 96 * 4 ? FH_STAGE : FH_WAVES * 16 * 96 * 4;
static_assert(FH_LDS <= 52 * 1024, "three blocks of k_fewch_bwd_weight_h per CU");

// PBLK (round 6, late): P is CHANNEL-BLOCKED [2][voxel][8] (VXM_S3_IN0_BLOCKED: the flow conv's x when the fused U-Net keeps its last activation
// blocked).  A thread then fetches the 16 channels of ONE voxel (four 16-byte loads, as many as before); W-neighbour lanes swap halves of
// their channels (DPP quad_perm), so that the even lane owns channels 0 .. 7 and the odd lane channels 8 .. 15 of the voxel PAIR and each writes
// the same packed fp16 words to the same LDS places as the planar staging -- every later instruction and every bit of the result is unchanged.
// POOL (round 6, last): the 16-channel operand is the gradient at the first ConvBlock's pre-activation, which vxm_maxpool2_bwd used to write
// (0.44 GB at full resolution, 0.24 ms) for this kernel alone to read: dz[c][p] = (gskip[c][p] + (p is the arg-max of its 2x2x2 block ?
// gpool[c][p >> 1] : 0)) * LeakyReLU'(y[c][p]).  in.P is gskip; a thread's four W-neighbours are two pooled blocks, whose routed gradients (one
// 8-byte load) and 16-bit codes (vxm_maxpool2_fwd_code: sign bits of the block's eight activations + arg-max; one 4-byte load) arrive with the
// tile and are folded into pv[] before the maximum is taken -- the same three operations in the same order as k_maxpool2_bwd_v4, hence the same
// bits in every later instruction.  Even D, H, W; W % 4 == 0.
template <int NT, bool PBLK = false, bool POOL = false>
__global__ void __launch_bounds__(FH_THREADS, 3) k_fewch_bwd_weight_h(FcIn in, int cs, int flip, int want_bias, float* __restrict__ part, int B, int D,
                                                                     int H, int W) {
    static_assert(!(PBLK && POOL), "the pooled-gradient operand is planar");
    using P2 = S3P<2>;
    VXM_DYN_SMEM(char, smem);
    char* const Ps = smem;                                     // [2 pieces][16][FH_PST] halves
    char* const Ss = smem + 2 * FH_PPIECE;                     // [2 pieces][3 c][3 planes][FH_SROWS][FH_SRW] halves
    float* const Tab = reinterpret_cast<float*>(smem + 2 * FH_PPIECE + 2 * FH_SPIECE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, idx = lane & 15;
    const int nw = (W + FH_TW - 1) / FH_TW, nh = (H + FH_TH - 1) / FH_TH;
    const int ntiles = B * D * nh * nw;
    const int V = D * H * W, HW = H * W;

    // column n = c * 27 + tap of N-tile nt -> byte offset of (c, plane kd, row kh) inside a piece of S, and the lane's kw as a funnel shift:
    // halves 8 kq + kw .. + 7 of the row.  Columns beyond 27 cs read column 0's data (finite; the reducer drops them -- column 27 cs is
    // overwritten with the bias sums below)
    int sbase[NT];
    unsigned ksh[NT];                                          // bit 0..4: shift of v_alignbit (0 or 16), bit 8: start one dword on (kw = 2)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + idx;
        int c = n / 27, tap = n - c * 27;
        if (n >= 27 * cs) { c = 0; tap = 13; }
        int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        if (flip && n < 27 * cs) { kd = 2 - kd; kh = 2 - kh; kw = 2 - kw; }
        sbase[nt] = ((c * 3 + kd) * FH_SROWS + kh) * (FH_SRW * 2) + kq * 16;
        ksh[nt] = (kw == 1 ? 16u : 0u) | (kw == 2 ? 256u : 0u);
    }
    f32x4 tot[NT], totb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) tot[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const u32x4 ones = {P2::ONES, P2::ONES, P2::ONES, P2::ONES};

    // staging roles.  P: 16 planes x 8 rows x 8 float4 = 1024 slots, four per thread (planes pm0 + 4 j: the plane step is a scalar offset);
    // S: a haloed plane is 10 rows x 17 PAIRS of W neighbours = 170 slots, one per thread (8-byte loads: W is even and tiles start at
    // multiples of 32, so a pair from the odd column w0 - 1 + 2 q is inside the volume or outside it per element), the same slot for each
    // of the (up to) 9 (channel, depth plane) pairs
    constexpr int NPV = 4;
    f32x4 pv[NPV];
    f32x2 sv[9];
    [[maybe_unused]] f32x2 gpv[POOL ? NPV : 1];                      // POOL: routed gradients / codes of the two pooled blocks under pv[j]
    [[maybe_unused]] unsigned cdv[POOL ? NPV : 1];
    [[maybe_unused]] unsigned kbase = 0u;                            // position of this thread's voxels inside their block: 4 (d & 1) + 2 (h & 1)
    const int pm0 = tid >> 6, prow = (tid >> 3) & 7, pw4 = tid & 7;
    const int spr = tid / 17, spq = tid - spr * 17;                          // spr >= FH_SROWS: no slot
    auto tile_coords = [&](int t, int& b, int& d, int& h0, int& w0) __attribute__((always_inline)) {
        const int tw = t % nw; int q = t / nw;
        const int th = q % nh; q /= nh;
        d = q % D; b = q / D;
        h0 = th * FH_TH; w0 = tw * FH_TW;
    };
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        int b, d, h0, w0;
        tile_coords(t < ntiles ? t : 0, b, d, h0, w0);
        int dead = t < ntiles ? 0 : VXM_OOB;                     // past the last tile: every lane out of range (branch-free)
        asm volatile("" : "+v"(dead));
        const __amdgpu_buffer_rsrc_t rp = vxm_rsrc(in.P + (size_t)b * in.p_bs, 16u * (unsigned)V * 4u);
        if constexpr (PBLK) {                                   // thread = voxel (row tid >> 5, column tid & 31): pv[j] = its channels 4 j .. 4 j + 3
            const int gh = h0 + (tid >> 5), gw = w0 + (tid & 31);
            const int pvoff = ((gh < H && gw < W) ? (d * HW + gh * W + gw) << 5 : VXM_OOB) | dead;
#pragma unroll
            for (int j = 0; j < NPV; ++j) pv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, pvoff, ((j >> 1) * V << 5) + ((j & 1) << 4), 0));
        } else {
        const int gh = h0 + prow, gw = w0 + 4 * pw4;
        const int pvoff = ((gh < H && gw < W) ? (pm0 * V + d * HW + gh * W + gw) << 2 : VXM_OOB) | dead;   // W % 4 == 0: a float4 is inside or outside the row
#pragma unroll
        for (int j = 0; j < NPV; ++j) pv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp, pvoff, (4 * j * V) << 2, 0));
        if constexpr (POOL) {
            const int V2 = V >> 3, W2 = W >> 1;
            const __amdgpu_buffer_rsrc_t rg = vxm_rsrc(in.gpool + (size_t)b * 16 * V2, 16u * (unsigned)V2 * 4u);
            const __amdgpu_buffer_rsrc_t rc = vxm_rsrc(reinterpret_cast<const float*>(in.code + (size_t)b * 16 * V2), 16u * (unsigned)V2 * 2u);
            const int cell = pm0 * V2 + ((d >> 1) * (H >> 1) + (gh >> 1)) * W2 + (gw >> 1);       // the first of the two blocks (gw % 4 == 0: even)
            const bool in_vol = gh < H && gw < W;
            const int goff = (in_vol ? cell << 2 : VXM_OOB) | dead, coff = (in_vol ? cell << 1 : VXM_OOB) | dead;
#pragma unroll
            for (int j = 0; j < NPV; ++j) {
                gpv[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rg, goff, (4 * j * V2) << 2, 0));
                cdv[j] = __builtin_amdgcn_raw_buffer_load_b32(rc, coff, (4 * j * V2) << 1, 0);
            }
            kbase = (unsigned)(((d & 1) << 2) | ((gh & 1) << 1));
        }
        }
        const __amdgpu_buffer_rsrc_t r0 = vxm_rsrc(in.S0 + (size_t)b * in.s0_bs, (unsigned)in.cs0 * (unsigned)V * 4u);
        const __amdgpu_buffer_rsrc_t r1 = vxm_rsrc(cs > in.cs0 ? in.S1 + (size_t)b * in.s1_bs : in.S0, (unsigned)(cs > in.cs0 ? cs - in.cs0 : in.cs0) * (unsigned)V * 4u);
        // the pair (w0 - 1 + 2 q, w0 + 2 q): the first element of pair 0 of the first tile column (w = -1) and the second of pair 16 of the
        // last (w = W) are outside the row -- those pairs are fetched as single dwords into the half that exists
        const int sh = h0 - 1 + spr, sw = w0 - 1 + 2 * spq;
        const bool rok = spr < FH_SROWS && (unsigned)sh < (unsigned)H;
        const bool ok0 = rok && (unsigned)sw < (unsigned)W, ok1 = rok && (unsigned)(sw + 1) < (unsigned)W;
        const int svoff = ((ok0 && ok1) ? (sh * W + sw) << 2 : VXM_OOB) | dead;
        const int svone = ((ok0 != ok1) ? (sh * W + (ok0 ? sw : sw + 1)) << 2 : VXM_OOB) | dead;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int pd = 0; pd < 3; ++pd) {
                const int gd = d - 1 + pd;                       // wave-uniform
                const bool first = c < in.cs0;
                const __amdgpu_buffer_rsrc_t r = first ? r0 : r1;
                const int cc = first ? c : c - in.cs0;
                int bad = (c < cs && (unsigned)gd < (unsigned)D) ? 0 : VXM_OOB;
                asm volatile("" : "+v"(bad));
                const int soff = (c < cs && (unsigned)gd < (unsigned)D) ? (cc * V + gd * HW) << 2 : 0;
                f32x2 v2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, svoff | bad, soff, 0));
                const float one = vxm_bload(r, svone | bad, soff);         // (volume-edge pairs only: out of range for every other lane)
                if (ok0 != ok1) { v2.x = ok0 ? one : 0.0f; v2.y = ok0 ? 0.0f : one; }
                sv[c * 3 + pd] = v2;
            }
    };
    // POOL: pv[] = gskip -> the gradient at the pre-activation (k_maxpool2_bwd_v4's arithmetic: (routed or 0) + skip, times LeakyReLU')
    auto fold_pool = [&]() __attribute__((always_inline)) {
        if constexpr (POOL) {
#pragma unroll
            for (int j = 0; j < NPV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned cd = (cdv[j] >> (16 * (e >> 1))) & 0xffffu, k = kbase | (unsigned)(e & 1);
                    const float g = (((cd >> 8) & 7u) == k ? gpv[j][e >> 1] : 0.0f) + pv[j][e];
                    pv[j][e] = g * (((cd >> k) & 1u) ? 1.0f : in.slope);
                }
        }
    };
    // largest magnitudes this wave loaded for the tile in flight -> Tab (P: [wave], S: [4 + wave])
    auto publish_max = [&]() __attribute__((always_inline)) {
        fold_pool();
        const float mp = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NPV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) f(pv[j][e]);
        });
        const float ms = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 9; ++i) { f(sv[i].x); f(sv[i].y); }
        });
        if (lane == 0) { Tab[wave] = mp; Tab[4 + wave] = ms; }
    };
    float sP = 1.0f, sS = 1.0f, inv_next = 1.0f, invP_next = 1.0f;      // scales of the tile in flight; the inverse product / P inverse of it
    auto take_scales = [&]() __attribute__((always_inline)) {
        const f32x4* const t4 = reinterpret_cast<const f32x4*>(Tab);
        const f32x4 a = t4[0], c2 = t4[1];
        const float mp = fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w));
        const float ms = fmaxf(fmaxf(c2.x, c2.y), fmaxf(c2.z, c2.w));
        float iP, iS;
        s3_scale_of(mp, sP, iP);
        s3_scale_of(ms, sS, iS);
        inv_next = iP * iS; invP_next = iP;
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
        if constexpr (PBLK) {
            const bool odd = (tid & 1) != 0;
            char* const dst0 = Ps + ((odd ? 8 : 0) * FH_PST + (tid >> 5) * FH_TW + (tid & 30)) * 2;     // channel 0 / 8 of the voxel pair (w & ~1, w | 1)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float mine_lo = pv[i >> 2][i & 3], mine_hi = pv[2 + (i >> 2)][i & 3];             // this voxel's channels i and 8 + i
                const float give = odd ? mine_lo : mine_hi;
                const float got = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(give), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                unsigned h_, l_;
                s3_split2_f16(odd ? got : mine_lo, odd ? mine_hi : got, sP, h_, l_);                    // (voxel w & ~1, voxel w | 1) of channel i + (odd ? 8 : 0)
                *reinterpret_cast<unsigned*>(dst0 + i * FH_PST * 2) = h_;
                *reinterpret_cast<unsigned*>(dst0 + i * FH_PST * 2 + FH_PPIECE) = l_;
            }
        } else {
#pragma unroll
        for (int j = 0; j < NPV; ++j) {
            unsigned h0_, l0_, h1_, l1_;
            s3_split2_f16(pv[j].x, pv[j].y, sP, h0_, l0_);
            s3_split2_f16(pv[j].z, pv[j].w, sP, h1_, l1_);
            char* const dst = Ps + ((pm0 + 4 * j) * FH_PST + prow * FH_TW + 4 * pw4) * 2;
            *reinterpret_cast<u32x2*>(dst) = (u32x2){h0_, h1_};
            *reinterpret_cast<u32x2*>(dst + FH_PPIECE) = (u32x2){l0_, l1_};
        }
        }
        if (spr < FH_SROWS) {
            char* const dst0 = Ss + spr * (FH_SRW * 2) + spq * 4;              // haloed columns 2 q, 2 q + 1: one aligned word
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int pd = 0; pd < 3; ++pd) {
                    if (c < cs) {
                        unsigned hh, ll;
                        s3_split2_f16(sv[c * 3 + pd].x, sv[c * 3 + pd].y, sS, hh, ll);
                        *reinterpret_cast<unsigned*>(dst0 + (c * 3 + pd) * FH_SPLANE) = hh;
                        *reinterpret_cast<unsigned*>(dst0 + (c * 3 + pd) * FH_SPLANE + FH_SPIECE) = ll;
                    }
                }
        }
    };
    // the 8 halves kw .. kw + 7 of the 12 a lane read (e[0..5]: b128 + b64), kw per lane: dword i = alignbit(e[i + 1], kw == 2 ? e[i + 1] : e[i], 16 (kw & 1))
    auto shifted = [&](u32x4 lo, u32x2 hi, unsigned k) __attribute__((always_inline)) {
        const unsigned e[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
        const bool two = (k & 256u) != 0u;
        const unsigned sh = k & 31u;
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_alignbit(e[i + 1], two ? e[i + 1] : e[i], sh);
        return o;
    };

    // pad halves of the S rows (34 .. 39) are read by kq = 3 (halves 24 .. 35): zero them once -- 0 x anything finite, and never rewritten
    for (int i = tid; i < 2 * FH_SPIECE / 4; i += FH_THREADS) reinterpret_cast<unsigned*>(Ss)[i] = 0u;
    __syncthreads();
    int t = blockIdx.x;
    load_tile(t);
    publish_max();
    __syncthreads();
    take_scales();
    store_tile();
    float inv_cur = inv_next, invP_cur = invP_next;
    __syncthreads();
    for (; t < ntiles; t += gridDim.x) {
        load_tile(t + gridDim.x);                                // unconditional (past the end every lane is out of range)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NT], accb = {0.f, 0.f, 0.f, 0.f};
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {                         // this wave's two rows = two K-steps of 32 voxels
            const int row = 2 * wave + rr;
            u32x4 af[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) af[p] = *reinterpret_cast<const u32x4*>(Ps + p * FH_PPIECE + (idx * FH_PST + row * FH_TW + 8 * kq) * 2);
            if (want_bias) {
#pragma unroll
                for (int p = 0; p < 2; ++p) accb = P2::mfma(af[p], ones, accb);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                u32x4 bf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const char* const q = Ss + p * FH_SPIECE + sbase[nt] + row * (FH_SRW * 2);
                    bf[p] = shifted(*reinterpret_cast<const u32x4*>(q), *reinterpret_cast<const u32x2*>(q + 16), ksh[nt]);
                }
#pragma unroll
                for (int tp = 0; tp < P2::NPROD; ++tp) acc[nt] = P2::mfma(af[P2::PA[tp]], bf[P2::PB[tp]], (rr == 0 && tp == 0) ? zero4 : acc[nt]);
            }
        }
        // the chain of this tile carries the scales of its P and S tiles: undone here (exact powers of two), folded into the fp32 totals
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) tot[nt][j] = __builtin_fmaf(acc[nt][j], inv_cur, tot[nt][j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) totb[j] = __builtin_fmaf(accb[j], invP_cur, totb[j]);
        __builtin_amdgcn_sched_barrier(0);
        publish_max();                                           // (waits for the loads of the next tile: they had the MFMA phase, and two other blocks, to arrive)
        __syncthreads();                                        // every wave is done reading this tile; the maxima of the next are visible
        if (t + (int)gridDim.x < ntiles) { take_scales(); store_tile(); }
        inv_cur = inv_next; invP_cur = invP_next;
        __syncthreads();
    }
    // ---- partials: part[block][16 m][NT * 16 columns]; the four waves of a block are summed through LDS in a fixed order.  Column 27 cs carries
    // the bias sums (every column of the ones product holds sum_v P[m][v]; S has x 1 there: only the P scale was applied)
    float* const red = reinterpret_cast<float*>(smem);         // [4][16][NT * 16]
    const int nb = 27 * cs;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool bias_col = want_bias && nt * 16 + idx == nb;
            red[(wave * 16 + 4 * kq + r) * (NT * 16) + nt * 16 + idx] = bias_col ? totb[r] : tot[nt][r];
        }
    __syncthreads();
    constexpr int NE = 16 * NT * 16;
    for (int i = tid; i < NE; i += FH_THREADS)
        part[(size_t)blockIdx.x * NE + i] = (red[i] + red[NE + i]) + (red[2 * NE + i] + red[3 * NE + i]);
}

// gw / gb from the per-block partials (fixed order: 16 slices of the blocks, then a tree)
__global__ void __launch_bounds__(256) k_fewch_reduce(const float* __restrict__ part, int nblocks, int ncols, int cs, int flip, int Cw_in,
                                                      float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float sm[16][17];
    const int e = blockIdx.x * 16 + (threadIdx.x & 15), y = threadIdx.x >> 4;       // element (m, col) of the 16 x ncols result; slice y of 16
    float s = 0.0f;
    if (e < 16 * ncols)
        for (int k = y; k < nblocks; k += 16) s += part[(size_t)k * (16 * ncols) + e];
    sm[y][threadIdx.x & 15] = s;
    __syncthreads();
    if (y == 0 && e < 16 * ncols) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sm[u][threadIdx.x & 15];
        const float sum = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) + (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        const int m = e / ncols, n = e - m * ncols;
        if (n < 27 * cs) {
            const int c = n / 27, tap = n - c * 27;
            if (!flip) gw[((size_t)m * Cw_in + c) * 27 + tap] = sum;            // P = dz: m = co, c = ci
            else gw[((size_t)c * Cw_in + m) * 27 + tap] = sum;                  // P = x:  m = ci, c = co
        } else if (n == 27 * cs && gb != nullptr && !flip) {
            gb[m] = sum;                                                        // ones column: sum_v dz[co][v]
        }
    }
}

bool fewch_enabled() {
    static const bool on = [] { const char* e = getenv("VXM_FEWCH"); return !(e && e[0] == '0'); }();
    return on;
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

}  // namespace

extern "C" {

int vxm_conv3d_k3_bwd_weight_variant(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                                     const float* dz, int64_t dz_bstride, int Cout, int D, int H, int W) {
    const bool vec = bwd_weight_wide_ok(x0, x0_bstride, x1, C1, x1_bstride, dz, dz_bstride, W);
    const int nct = Cout <= 16 ? 1 : 2;                 // of the unswapped plan
    if (vec && fewch_enabled() && !x0_up && ((Cout == 16 && C0 + C1 <= 3) || (Cout <= 3 && C0 == 16 && C1 == 0))) return 30 + nct;    // k_fewch_bwd_weight
    if (x0_up && vec && (D & 1) == 0 && (H & 1) == 0) return 20 + nct;      // collapsed upsampled segment (+ regular skip segment)
    return (vec ? 10 : 0) + nct;
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
    {                                                  // few-channel kernels: up to 768 block partials of 16 x 96 (+ channel-sum scratch)
        const size_t alt = sizeof(float) * ((size_t)768 * 16 * 96 + (size_t)Cout * CS_SLICES);
        if ((Cin <= 3 || Cout <= 3) && alt > need) need = alt;
    }
    return 256 + need;
}

// few-channel layers on the fp16-piece scheme (k_fewch_bwd_weight_h): same operands, workspace and reducer as the fp32-MFMA few-channel path
static int fewch_h_launch(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* dz, int64_t dz_bstride,
                          int Cout, float* gw, float* gb, float* part, int B, int D, int H, int W, void* stream, bool p_blocked = false,
                          const float* gpool = nullptr, const uint16_t* code = nullptr, float slope = 1.0f) {
    const int Cin = C0 + C1;
    const bool few_in = Cout == 16 && Cin <= 3;
    const int cs = few_in ? Cin : Cout;
    const int NT = (27 * cs + (few_in && gb ? 1 : 0) + 15) / 16;
    FcIn fin;
    if (few_in) fin = FcIn{dz, (long long)dz_bstride, x0, (long long)x0_bstride, C0, x1, (long long)x1_bstride};
    else fin = FcIn{x0, (long long)x0_bstride, dz, (long long)dz_bstride, Cout, nullptr, 0};
    fin.gpool = gpool; fin.code = code; fin.slope = slope;
    const long long ntiles = (long long)B * D * ((H + FH_TH - 1) / FH_TH) * ((W + FH_TW - 1) / FH_TW);
    const int nblk = (int)(ntiles < 768 ? ntiles : 768);              // three 4-wave blocks per CU (31 KB of LDS each; the workspace holds 512 x 16 x 96 partials)
    static bool opt_in = false;
    if (!opt_in) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight_h<2>), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight_h<4>), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight_h<6>), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight_h<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight_h<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight_h<6, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight_h<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight_h<4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight_h<6, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FH_LDS);
        opt_in = true;
    }
#define FH_LAUNCH(...) hipLaunchKernelGGL((k_fewch_bwd_weight_h<__VA_ARGS__>), dim3(nblk), dim3(FH_THREADS), FH_LDS, VXM_STREAM(stream), fin, cs, few_in ? 0 : 1, \
                                          few_in && gb ? 1 : 0, part, B, D, H, W)
    if (p_blocked) { if (NT <= 2) FH_LAUNCH(2, true); else if (NT <= 4) FH_LAUNCH(4, true); else FH_LAUNCH(6, true); }
    else if (code) { if (NT <= 2) FH_LAUNCH(2, false, true); else if (NT <= 4) FH_LAUNCH(4, false, true); else FH_LAUNCH(6, false, true); }
    else if (NT <= 2) FH_LAUNCH(2); else if (NT <= 4) FH_LAUNCH(4); else FH_LAUNCH(6);
#undef FH_LAUNCH
    const int ncols = (NT <= 2 ? 2 : (NT <= 4 ? 4 : 6)) * 16;
    hipLaunchKernelGGL(k_fewch_reduce, dim3((16 * ncols + 15) / 16), dim3(256), 0, VXM_STREAM(stream), part, nblk, ncols, cs, few_in ? 0 : 1, Cin, gw,
                       few_in ? gb : nullptr);
    if (!few_in && gb) {
        float* cs_ws = part + (size_t)nblk * 16 * ncols;
        hipLaunchKernelGGL(k_channel_sum_partial, dim3(Cout, CS_SLICES), dim3(256), 0, VXM_STREAM(stream), dz, (long long)dz_bstride, cs_ws, B, (size_t)D * H * W);
        hipLaunchKernelGGL(k_channel_sum_finish, dim3(Cout), dim3(64), 0, VXM_STREAM(stream), cs_ws, gb);
    }
    return vxm_check_launch("vxm_conv3d_k3_fewch_bwd_weight");
}

static int bwd_weight_impl(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                           const float* dz, int64_t dz_bstride, int Cout, float* gw, float* gb, void* workspace,
                           size_t workspace_bytes, int B, int D, int H, int W, void* stream, bool seg0_only) {
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
    // layers with 1-3 channels on one side: the dedicated persistent kernel (k_fewch_bwd_weight)
    const bool few_in = vec && fewch_enabled() && !x0_up && Cout == 16 && Cin <= 3 && !seg0_only;           // first block: P = dz, S = x
    const bool few_out = vec && fewch_enabled() && !x0_up && Cout <= 3 && C0 == 16 && C1 == 0 && !seg0_only;  // flow conv: P = x, S = dz
    if (few_in || few_out) {
        const int cs = few_in ? Cin : Cout;
        const int NT = (27 * cs + (few_in && gb ? 1 : 0) + 15) / 16;
        FcIn fin;
        if (few_in) fin = FcIn{dz, (long long)dz_bstride, x0, (long long)x0_bstride, C0, x1, (long long)x1_bstride};
        else fin = FcIn{x0, (long long)x0_bstride, dz, (long long)dz_bstride, Cout, nullptr, 0};
        const long long ntiles = (long long)B * D * ((H + FC_TH - 1) / FC_TH) * ((W + FC_TW - 1) / FC_TW);
        const int nblk = (int)(ntiles < 512 ? ntiles : 512);
        const size_t lds = sizeof(float) * (size_t)fc_lds_floats(cs);
        static bool fc_opt_in = false;
        if (!fc_opt_in) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fewch_bwd_weight<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            fc_opt_in = true;
        }
#define FC_LAUNCH(NT_) hipLaunchKernelGGL(k_fewch_bwd_weight<NT_>, dim3(nblk), dim3(FC_THREADS), lds, VXM_STREAM(stream), fin, cs, few_out ? 1 : 0, 1, part, B, D, H, W)
        if (NT <= 2) FC_LAUNCH(2); else if (NT <= 4) FC_LAUNCH(4); else FC_LAUNCH(6);
#undef FC_LAUNCH
        const int ncols = (NT <= 2 ? 2 : (NT <= 4 ? 4 : 6)) * 16;
        hipLaunchKernelGGL(k_fewch_reduce, dim3((16 * ncols + 15) / 16), dim3(256), 0, VXM_STREAM(stream), part, nblk, ncols, cs, few_out ? 1 : 0, Cin, gw,
                           few_in ? gb : nullptr);
        if (few_out && gb) {
            float* cs_ws = part + (size_t)nblk * 16 * ncols;
            hipLaunchKernelGGL(k_channel_sum_partial, dim3(Cout, CS_SLICES), dim3(256), 0, VXM_STREAM(stream), dz, (long long)dz_bstride, cs_ws, B, (size_t)D * H * W);
            hipLaunchKernelGGL(k_channel_sum_finish, dim3(Cout), dim3(64), 0, VXM_STREAM(stream), cs_ws, gb);
        }
        return vxm_check_launch("vxm_conv3d_k3_bwd_weight");
    }
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
        if (seg0_only) return vxm_check_launch("vxm_conv3d_k3_bwd_weight_up_segment");      // the caller owns the skip segment and the bias
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
            hipLaunchKernelGGL(k_reduce_partials, dim3(vxm_blocks(n1 + (gb ? Cout : 0), 64)), dim3(1024), 0, VXM_STREAM(stream), part, gw, gb, n1, C1, Cout,
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
        hipLaunchKernelGGL(k_reduce_partials, dim3(vxm_blocks(n, 64)), dim3(1024), 0, VXM_STREAM(stream), part, gw, (float*)nullptr, n, Cout, Cin,
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
    hipLaunchKernelGGL(k_reduce_partials, dim3(vxm_blocks(n + (gb ? Cout : 0), 64)), dim3(1024), 0, VXM_STREAM(stream), part, gw, gb, n, Cin, Cout,
                       p.T, p.Qc, p.G, 16 * p.NCT, 0, Cin, 0);
    return vxm_check_launch("vxm_conv3d_k3_bwd_weight");
}

int vxm_conv3d_k3_bwd_weight(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                             const float* dz, int64_t dz_bstride, int Cout, float* gw, float* gb, void* workspace,
                             size_t workspace_bytes, int B, int D, int H, int W, void* stream) {
    return bwd_weight_impl(x0, C0, x0_bstride, x0_up, x1, C1, x1_bstride, dz, dz_bstride, Cout, gw, gb, workspace, workspace_bytes, B, D, H, W, stream,
                           false);
}

int vxm_conv3d_k3_bwd_weight_up_segment(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* dz,
                                        int64_t dz_bstride, int Cout, float* gw, void* workspace, size_t workspace_bytes, int B, int D, int H,
                                        int W, void* stream) {
    VXM_REQUIRE(vxm_conv3d_k3_bwd_weight_variant(x0, C0, x0_bstride, 1, x1, C1, x1_bstride, dz, dz_bstride, Cout, D, H, W) / 10 == 2 &&
                    (long long)C0 * (D / 2) * (H / 2) * (W / 2) < (1ll << 29),
                VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_bwd_weight_up_segment: operands do not qualify for the collapsed kernel (use vxm_conv3d_k3_bwd_weight)");
    return bwd_weight_impl(x0, C0, x0_bstride, 1, x1, C1, x1_bstride, dz, dz_bstride, Cout, gw, nullptr, workspace, workspace_bytes, B, D, H, W, stream,
                           true);
}

/* few-channel layers (first block 2 -> 16, flow conv 16 -> 3) on the fp16-piece scheme: include/vxm_hip.h */
int vxm_conv3d_k3_fewch_bwd_weight_ok(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* dz,
                                      int64_t dz_bstride, int Cout, int pieces, int W) {
    if (pieces != 2 || !fewch_enabled()) return 0;
    if (!bwd_weight_wide_ok(x0, x0_bstride, x1, C1, x1_bstride, dz, dz_bstride, W)) return 0;
    return (Cout == 16 && C0 + C1 <= 3 && C0 >= 1) || (Cout <= 3 && Cout >= 1 && C0 == 16 && C1 == 0);
}

int vxm_conv3d_k3_fewch_bwd_weight(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* dz,
                                   int64_t dz_bstride, int Cout, float* gw, float* gb, void* workspace, size_t workspace_bytes, int B, int D, int H,
                                   int W, int pieces_and_layout, void* stream) {
    const int pieces = pieces_and_layout & 0xff, lay = pieces_and_layout & ~0xff;
    VXM_REQUIRE(lay == 0 || (lay == VXM_S3_IN0_BLOCKED && C0 == 16 && C1 == 0 && Cout <= 3), VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_fewch_bwd_weight: layout flags 0x%x (VXM_S3_IN0_BLOCKED for the 16-channel x0 of a 16 -> 1..3 layer only)", lay);
    if (int e = check_conv("vxm_conv3d_k3_fewch_bwd_weight", C0, C1, 0, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(x0 && dz && gw && workspace && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_fewch_bwd_weight: null pointer");
    VXM_REQUIRE(vxm_conv3d_k3_fewch_bwd_weight_ok(x0, C0, x0_bstride, x1, C1, x1_bstride, dz, dz_bstride, Cout, pieces, W), VXM_ERR_UNSUPPORTED,
                "vxm_conv3d_k3_fewch_bwd_weight: 1-3 channels against 16, W %% 4 == 0, 16-byte aligned tensors, pieces = 2 (got C0=%d C1=%d Cout=%d "
                "pieces=%d W=%d)", C0, C1, Cout, pieces, W);
    VXM_REQUIRE(workspace_bytes >= vxm_conv3d_k3_bwd_weight_workspace_bytes(C0 + C1, Cout, B, D, H, W), VXM_ERR_WORKSPACE,
                "vxm_conv3d_k3_fewch_bwd_weight: workspace too small (%zu bytes; vxm_conv3d_k3_bwd_weight_workspace_bytes)", workspace_bytes);
    const uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255;
    return fewch_h_launch(x0, C0, x0_bstride, x1, C1, x1_bstride, dz, dz_bstride, Cout, gw, gb, reinterpret_cast<float*>(base), B, D, H, W, stream, lay != 0);
}

int vxm_conv3d_k3_fewch_bwd_weight_pool(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* gskip,
                                        int64_t gskip_bstride, const float* gpool, const uint16_t* code, float slope, float* gw, float* gb, void* workspace,
                                        size_t workspace_bytes, int B, int D, int H, int W, int pieces, void* stream) {
    if (int e = check_conv("vxm_conv3d_k3_fewch_bwd_weight_pool", C0, C1, 0, 16, B, D, H, W)) return e;
    VXM_REQUIRE(x0 && gskip && gpool && code && gw && workspace && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_fewch_bwd_weight_pool: null pointer");
    VXM_REQUIRE(!((D | H | W) & 1) && (reinterpret_cast<uintptr_t>(gpool) & 7) == 0 && (reinterpret_cast<uintptr_t>(code) & 3) == 0, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_fewch_bwd_weight_pool: even extents, gpool 8-byte and code 4-byte aligned (got %dx%dx%d)", D, H, W);
    VXM_REQUIRE(vxm_conv3d_k3_fewch_bwd_weight_ok(x0, C0, x0_bstride, x1, C1, x1_bstride, gskip, gskip_bstride, 16, pieces, W), VXM_ERR_UNSUPPORTED,
                "vxm_conv3d_k3_fewch_bwd_weight_pool: 1-3 input channels against 16, W %% 4 == 0, 16-byte aligned tensors, pieces = 2 (got C0=%d C1=%d "
                "pieces=%d W=%d)", C0, C1, pieces, W);
    VXM_REQUIRE(workspace_bytes >= vxm_conv3d_k3_bwd_weight_workspace_bytes(C0 + C1, 16, B, D, H, W), VXM_ERR_WORKSPACE,
                "vxm_conv3d_k3_fewch_bwd_weight_pool: workspace too small (%zu bytes; vxm_conv3d_k3_bwd_weight_workspace_bytes)", workspace_bytes);
    const uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255;
    return fewch_h_launch(x0, C0, x0_bstride, x1, C1, x1_bstride, gskip, gskip_bstride, 16, gw, gb, reinterpret_cast<float*>(base), B, D, H, W, stream, false,
                          gpool, code, slope);
}

}  // extern "C"
