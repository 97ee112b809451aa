// fp32 convolutions of the VxmDense U-Net on the bf16 matrix pipe: the three-way bf16 split ("bf16x3").
//
// Replaces, for the plain (not collapsed-upsample) full-resolution layers, the fp32-MFMA kernels of conv_fwd.hip:
//   voxelmorph/torch/networks.py:299-305  ConvBlock = Conv3d(k3,s1,p1) + LeakyReLU(0.2)   (forward)
//   and its autograd twin convolution_backward w.r.t. the input (+ leaky_relu_backward of the previous block).
//
// Arithmetic.  Every fp32 operand is written as the exact sum of three bf16 pieces, x = h + m + l (+ a remainder below 2^-24 |x|):
//     h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)          (the two subtractions are exact in fp32)
// and a product x w is accumulated in fp32 from the six piece products whose magnitude is above 2^-24 |x w|:
//     x w  ~=  h_x h_w + h_x m_w + m_x h_w + h_x l_w + l_x h_w + m_x m_w           (dropped: m l, l m, l l  <=  2^-25 |x w|)
// Each piece product is exact in the MFMA (8 x 8 significand bits), the accumulation is the fp32 accumulation of
// v_mfma_f32_16x16x32_bf16.  The result is an fp32 convolution to within a few fp32 ulps of the exact dot product (measured
// against an fp64 evaluation by the same gate as the fp32-MFMA kernels: rel-L2 <= 1e-5, tests/test_gpu_s3.py), on a pipe that does
// 8192 MACs in 16 cycles instead of 1024 in 32: six instructions replace sixteen.
//
// Layout.  HBM tensors stay planar fp32 NCDHW (the C ABI and every other kernel are unchanged); a block splits its haloed
// input tile while staging it: LDS holds, per piece, channel-blocked bf16 [8-channel block][hd][hr][hw][8] -- the 8 channels of
// a voxel are one 16-byte word, one lane's share of a K = 32 MFMA operand.  Weights are split when they are packed.
//
// Implicit GEMM (as conv_bf16.hip): M = 16 output channels, N = 16 voxels of a W row, K = 32 = four units of 8 input channels,
// unit = (kd, kw, 8-channel block); the kh tap is walked by SLIDING over the haloed rows: the three B pieces of haloed row r
// serve output rows r, r-1, r-2 with the weight fragments of kh = 0, 1, 2 (3 LDS reads feed up to 18 NCT MFMAs).  With 8-channel
// chunks eight of the nine (kd, kw) units fill two sliding K-steps and the ninth is multiplied in a "row-tap" step whose four
// lane groups hold its three kh taps: 27 of 28 K slots carry data (s3_unit below).
// Block = 8 waves, output tile 8(D) x ROWS(H) x 16(W), wave = depth slice; chunk = CB 8-channel blocks of one segment of the
// virtual concat [x0 (optionally through a nearest x2 upsampling gather) | x1].
#include "conv_common.h"
#include "s3_pieces.h"

#ifdef VXM_S3_EXP
#define S3_DBG(dbg, bit) (((dbg) & (bit)) != 0)
#else
#define S3_DBG(dbg, bit) false
#endif

namespace {

constexpr int S3_TD = 8, S3_THREADS = 512, S3_HWV = 18;
// launch flag VXM_S3_REVERSE_TILES (include/vxm_hip.h; travels inside `lay`): walk the tiles from the END of the tensor.  A layer that
// reads what the previous launch has just written finds the last-written part of it in the memory-side cache (256 MB on MI355X, the tensors
// are 440 - 880 MB) when it starts where its producer stopped; in the same direction the cache holds the wrong end.
constexpr int S3_REVERSE_TILES = VXM_S3_REVERSE_TILES;

// Which unit (kd, kw, 8-channel block) lane group kg multiplies in K-step s.  A ds_read_b128 is served in four groups of 16 lanes
// that pair lanes of kg 0 with lanes of kg 1 (and kg 2 with kg 3; MI355X_MICROARCH.md, LDS): a group is conflict-free only when
// the 16-byte words of the two lane groups are congruent mod 16.  Words of one (kd, cb) plane are contiguous along W, so a pair
// must share kw and differ in the depth plane / channel block only, and plane / block strides are padded to multiples of 16 words
// (the first layout, units in (kd, kw) order with unpadded strides, cost 8 instead of 4 LDS cycles per read).
//   CB = 2: pairs are the two channel blocks of one (kd, kw): unit u = 4 s + kg = 2 (3 kd + kw) + cb, 18 units in 5 steps.
//   CB = 1: 9 units do not fill K-steps of 4 (the first version used three sliding steps, 9 of 12 slots: a quarter of all MFMAs
//   multiplied zeros, on a pipe that is power-limited -- see DESIGN.md).  Now 27 of 28 slots: two SLIDING steps
//       s0: (0,0) (1,0) | (0,1) (1,1)      s1: (0,2) (1,2) | (2,0) (2,1)      (the last pair is 1 word apart: 6 instead of 4 LDS cycles)
//   and the remaining unit (kd, kw) = (2, 2) in one ROW-TAP step per output row: lane group kg multiplies tap kh = kg of haloed row
//   r + kg (kg = 3: zero weights, kg 2's address) -- 7 instead of 9 MFMA sets per output row and 8-channel chunk.
// mask tensor of sample b: fp32 elements, or -- lay & (VXM_S3_MASK_SIGNS | VXM_S3_OUT_SIGNS) -- the BYTES of a sign tensor (batch stride in bytes)
__device__ __forceinline__ const float* s3_mask_of_sample(const float* mask, int b, long long mask_bs, int lay) {
    if (mask == nullptr) return nullptr;
    if (lay & (VXM_S3_MASK_SIGNS | VXM_S3_OUT_SIGNS)) return reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(mask) + (size_t)b * mask_bs);
    return mask + (size_t)b * mask_bs;
}
struct S3Unit { int kd, kw, cb, valid; };
__host__ __device__ constexpr S3Unit s3_unit(int CB, int s, int kg) {
    if (CB == 2) {
        const int u = 4 * s + kg, v = u < 18, p = (v ? u : u - 2) >> 1;
        return S3Unit{p / 3, p % 3, u & 1, v};
    }
    if (s == 0) return S3Unit{kg & 1, kg >> 1, 0, 1};
    return kg < 2 ? S3Unit{kg, 2, 0, 1} : S3Unit{2, kg - 2, 0, 1};       // s == 1 (CB = 1 has two sliding steps)
}

template <int NCT, int ROWS, int CB, int NP>
struct S3Cfg {
    static constexpr int HR = ROWS + 2, PLANE_USED = HR * S3_HWV;
    static constexpr int PLANE = (PLANE_USED + 15) / 16 * 16, SLOTS = (S3_TD + 2) * PLANE;    // 16-byte words of one (piece, block); strides = 0 mod 16
    static constexpr int NSLOT = CB * (S3_TD + 2) * PLANE_USED;                               // haloed voxels x blocks to stage
    static constexpr int NS = CB == 2 ? 5 : 2;                                                // sliding K-steps of a chunk
    static constexpr int RT = CB == 2 ? 0 : 1;                                                // row-tap steps (CB = 1: unit (2, 2))
    static constexpr int XW = NP * CB * SLOTS;                                                // [piece][cb][slot]
    static constexpr int WCH = (NS * 3 + RT) * NP * NCT * 64;                                 // [s][kh][piece][ct][lane], then [piece][ct][lane] of the row-tap step
    static constexpr int LDS_BYTES = (XW + WCH) * 16 + (NP == 2 ? 64 : 0);                    // NP = 2: + the eight wave maxima of the chunk in flight
    static constexpr int NI = (NSLOT + S3_THREADS - 1) / S3_THREADS;                          // staging slots per thread
    static constexpr int WIT = (WCH + S3_THREADS - 1) / S3_THREADS;
    // two blocks per CU when the LDS allows it and the accumulators fit 128 registers (NP = 2 carries a second, per-chunk accumulator set)
    static constexpr int MIN_WAVES = (LDS_BYTES <= 80 * 1024 && !(NP == 2 && NCT == 2)) ? 4 : 2;
};

// wp: [G][Q]{[NS][kh 3][piece NP][NCT][64 lanes], [RT][piece NP][NCT][64 lanes]} 16-byte words (k_s3_pack_weights), NP = 2: followed by one
// trailer word {1 / weight scale, weight scale, -, -}.  Q0 chunks cover segment 0.
// RUN (NP = 2 only): the tile's accumulators take the MFMA chains directly and carry a RUNNING power-of-two scale (as conv_s3u.hip) instead of
// a per-chunk accumulator set folded by the vector ALU: 4 NCT ROWS registers fewer -- what lets the 8-row tile keep two blocks per CU.
// BLK (NP = 2, CB = 1, no upsampling gather): the input tensor is CHANNEL-BLOCKED [C / 8][voxel][8] -- the 8 channels of a staging slot are 32
// contiguous bytes, a haloed row of the tile 576 contiguous bytes instead of 8 x 72 (the vector L1's sector requests bound
// these kernels).  lay & VXM_S3_OUT_BLOCKED: output (and mask) tensor in the same layout.  Same values, same arithmetic, same results.
template <int NCT, int ROWS, int CB, int NP, bool RUN = false, bool BLK = false>
__global__ void __launch_bounds__(S3_THREADS, (S3Cfg<NCT, ROWS, CB, NP>::MIN_WAVES))
k_s3_conv(ConvIn in, const u32x4* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, long long y_bs, int Cout,
          float act_slope, const float* __restrict__ mask, long long mask_bs, float mask_slope, int B, int D, int H, int W, int Q0, int Q, int lay, int dbg) {
    using C = S3Cfg<NCT, ROWS, CB, NP>;
    using P = S3P<NP>;
    VXM_DYN_SMEM(u32x4, smem);
    constexpr int HR = C::HR, PLANE = C::PLANE, PUSED = C::PLANE_USED, SLOTS = C::SLOTS, NSLOT = C::NSLOT, NS = C::NS, NI = C::NI, WCH = C::WCH, WIT = C::WIT;
    u32x4* const Xs = smem;                 // [NP][CB][SLOTS]
    u32x4* const Ws = smem + C::XW;         // [NS][3][NP][NCT][64]
    float* const Wm = reinterpret_cast<float*>(smem + C::XW + WCH);      // NP = 2: largest magnitude each wave loaded for the chunk in flight
    const int tid = threadIdx.x, tid_ = tid, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, n = lane & 15;

    // tiles of this block: block b runs on XCD b % 8 (observed; speed only), XCD x takes a contiguous tile range and its blocks walk it
    // with the stride of the blocks per XCD (one tile per block when the grid covers every tile)
    const int nw = (W + 15) / 16, nh = (H + ROWS - 1) / ROWS, nd = (D + S3_TD - 1) / S3_TD;
    const int ntiles = B * nd * nh * nw;
    int t_lo, t_hi, t_step;
    if (ntiles >= 64) {
        const int x = blockIdx.x & 7;
        t_lo = (int)((long long)ntiles * x / 8) + (int)(blockIdx.x >> 3); t_hi = (int)((long long)ntiles * (x + 1) / 8); t_step = (int)(gridDim.x >> 3);
    } else {
        t_lo = blockIdx.x; t_hi = ntiles; t_step = gridDim.x;
    }
    const int g = blockIdx.y;
    const int V = D * H * W;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const int V0 = in.up0 ? Dl * Hl * Wl : V;

    // The STAGING tile (whose chunks are fetched and written to LDS): origin, batch descriptors and the staging roles of this thread.
    // slot i = tid + 512 j = (cb, hd, hh, hw) of the haloed tile, one register per slot: (hd, hh, hw) packed as hd << 10 | hh << 5 | hw,
    // -1 where there is no slot or the voxel is outside the volume (padding); global offsets and the LDS word are rebuilt from it per chunk
    // (a few VALU).  It runs ahead of the tile being computed by one chunk: under the MFMAs of a tile's LAST chunk, chunk 0 of the block's
    // NEXT tile is in flight, so only a block's first tile waits for load -> split -> LDS -> barrier before its first MFMA.
    int d0 = 0, h0 = 0, w0 = 0, bt = 0;
    __amdgpu_buffer_rsrc_t r0, r1;
    int spos[NI];
    auto set_tile = [&](int tile) __attribute__((always_inline)) {
        int tid = tid_;
        asm volatile("" : "+v"(tid));
        const bool live = tile < t_hi;                        // past the block's last tile: every slot is padding, nothing is fetched
        const int tl = (lay & S3_REVERSE_TILES) ? ntiles - 1 - (live ? tile : t_lo) : (live ? tile : t_lo);
        int tw = tl % nw; int tq = tl / nw;
        int th = tq % nh; tq /= nh;
        int td = tq % nd;
        bt = tq / nd;
        if (S3_DBG(dbg, 64)) {                                  // depth fastest
            td = tl % nd; tq = tl / nd; tw = tq % nw; tq /= nw; th = tq % nh; bt = tq / nh;
        }
        if (S3_DBG(dbg, 128)) {                                 // groups of (2 d, 2 h, every w); needs even nd, nh
            const int gi = tl / (4 * nw), r = tl - gi * 4 * nw, ngh = nh >> 1, ngd = nd >> 1;
            tw = r % nw; th = (gi % ngh) * 2 + ((r / nw) & 1); td = ((gi / ngh) % ngd) * 2 + ((r / nw) >> 1); bt = gi / (ngh * ngd);
        }
        d0 = td * S3_TD; h0 = th * ROWS; w0 = tw * 16;
        r0 = vxm_rsrc(in.x0 + (size_t)bt * in.bs0, (unsigned)in.C0 * (unsigned)V0 * 4u);
        r1 = vxm_rsrc(in.C1 ? in.x1 + (size_t)bt * in.bs1 : in.x0, (unsigned)in.C1 * (unsigned)V * 4u);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = tid + S3_THREADS * j;
            const int cb = i / ((S3_TD + 2) * PUSED), rem = i - cb * (S3_TD + 2) * PUSED;
            const int hd = rem / PUSED, r2 = rem - hd * PUSED, hh = r2 / S3_HWV, hw = r2 - hh * S3_HWV;
            const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
            bool ok = live && i < NSLOT && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            if (S3_DBG(dbg, 1)) ok = false;                                                     // no global reads at all
            if (S3_DBG(dbg, 16) && (hw == 0 || hw == 17)) ok = false;                           // no W halo: every row is one aligned 64-byte sector
            if (S3_DBG(dbg, 32) && (hd == 0 || hd == S3_TD + 1 || hh == 0 || hh == ROWS + 1)) ok = false;      // no D / H halo
            spos[j] = ok ? (hd << 10 | hh << 5 | hw) : -1;
        }
    };

    // per-lane LDS word offset of the unit this lane group reads in K-step s (s3_unit; zero-weight units read their partner's word)
    int xoff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const S3Unit u0 = s3_unit(CB, s, 0), u1 = s3_unit(CB, s, 1), u2 = s3_unit(CB, s, 2), u3 = s3_unit(CB, s, 3);
        const S3Unit u = kg == 0 ? u0 : kg == 1 ? u1 : kg == 2 ? u2 : u3;
        xoff[s] = u.cb * SLOTS + (wave + u.kd) * PLANE + u.kw + n;
    }
    // row-tap step (CB = 1): lane group kg reads tap (kd, kh, kw) = (2, kg, 2) of output row r at haloed row r + kg; kg = 3 mirrors kg 2
    const int xoff_rt = (wave + 2) * PLANE + (kg < 3 ? kg : 2) * S3_HWV + 2 + n;

    float xr[NI][8];                                         // chunk q + 1 in flight under the MFMAs of chunk q
    int voffs[NI];                                           // its per-lane offsets: kept live across the MFMA phase (see keep_offsets)
    auto load_chunk = [&](int q) __attribute__((always_inline)) {
        int tid = tid_;
        asm volatile("" : "+v"(tid));
        const bool s0 = q < Q0;                               // wave-uniform
        const bool up = s0 && in.up0;
        const __amdgpu_buffer_rsrc_t r = s0 ? r0 : r1;
        const int Cseg = s0 ? in.C0 : in.C1, cbg = (s0 ? q : q - Q0) * CB, Vs = up ? V0 : V;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int cb = (tid + S3_THREADS * j) / ((S3_TD + 2) * PUSED);
            const int gd = d0 - 1 + (spos[j] >> 10), gh = h0 - 1 + ((spos[j] >> 5) & 31), gw = w0 - 1 + (spos[j] & 31);
            const int sv = up ? ((gd >> 1) * Hl + (gh >> 1)) * Wl + (gw >> 1) : (gd * H + gh) * W + gw;
            const bool ok = spos[j] >= 0 && (cbg + cb) * 8 < Cseg && q < Q;  // segments carry multiples of 8 channels; q == Q: nothing to fetch
            voffs[j] = ok ? ((cbg + cb) * 8 * Vs + sv) << 2 : VXM_OOB;
            if constexpr (BLK) {
                voffs[j] = ok ? ((cbg + cb) * Vs + sv) << 5 : VXM_OOB;
                const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffs[j], 0, 0));
                const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffs[j], 16, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) { xr[j][e] = lo[e]; xr[j][4 + e] = hi[e]; }
                continue;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[j][e] = vxm_bload(r, voffs[j], (e * Vs) << 2);
        }
    };
    // The address VGPR of a buffer load that is still in flight must not be reused: the compiler otherwise hands it to the first
    // ds_read of the MFMA phase and protects that write with `s_waitcnt vmcnt(0)` -- i.e. waits for the whole prefetch right after
    // issuing it (seen in the ISA).  Keeping the offsets formally live until the MFMA phase is over costs NI registers.
    auto keep_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(voffs[j]));
    };
    // NP = 2: the largest magnitude of the chunk in flight.  A wave publishes the maximum over what ITS lanes loaded right after its
    // MFMA loop (the loads have had the whole phase to land), i.e. before the barrier that ends the phase anyway: no extra barrier.
    auto publish_max = [&]() __attribute__((always_inline)) {
        if constexpr (NP == 2) {
            const float m = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) f(xr[j][e]);
            });
            if (lane == 0) Wm[wave] = m;
        }
    };
    float inv_cur = 1.0f;                                    // NP = 2: 1 / scale of the chunk that is in LDS
    int E_run = 15;                                          // RUN: exponent field the running scale of the tile is built from
    float ratio = 1.0f;                                      // RUN: what the accumulators are multiplied by before the MFMAs of the chunk just stored
    auto store_chunk = [&](int q, bool first = false) __attribute__((always_inline)) {
        // (an opaque copy of the thread index: the slot arithmetic below is loop-invariant, and hoisted out of the tile loop it would
        // occupy registers across the MFMA phases -- 24 spilled VGPRs in the 128-register instance)
        int tid = tid_;
        asm volatile("" : "+v"(tid));
        // the chunk's packed weights: requested first, written to LDS after the split arithmetic below has covered their latency
        u32x4 wv[WIT];
        const __amdgpu_buffer_rsrc_t rw = vxm_rsrc(reinterpret_cast<const float*>(wp + ((size_t)g * Q + q) * WCH), WCH * 16u);
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            wv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tid + S3_THREADS * it) * 16, 0, 0));
        float sc = 1.0f;
        if constexpr (NP == 2) {
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(Wm), m1 = *reinterpret_cast<const f32x4*>(Wm + 4);
            const float mx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
            if constexpr (RUN) {
                int E = (int)(__float_as_uint(mx) >> 23) & 255;
                E = E < 15 ? 15 : E;
                const int E_new = first ? E : (E > E_run ? E : E_run), dE = E_new - E_run;
                ratio = (first || dE == 0) ? 1.0f : (dE > 126 ? 0.0f : __uint_as_float((unsigned)(127 - dE) << 23));
                E_run = E_new;
                sc = __uint_as_float((unsigned)(268 - E_run) << 23);
                inv_cur = __uint_as_float((unsigned)(E_run - 14) << 23);
            } else {
                s3_scale_of(mx, sc, inv_cur);
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            unsigned pk[NP][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (NP == 3) s3_split2(xr[j][2 * e], xr[j][2 * e + 1], pk[0][e], pk[1][e], pk[2][e]);
                else s3_split2_f16(xr[j][2 * e], xr[j][2 * e + 1], sc, pk[0][e], pk[1][e]);
            }
            const int i = tid + S3_THREADS * j;
            if (i < NSLOT) {                                   // (padding slots are written too: zeros from the out-of-range loads)
                const int cb = i / ((S3_TD + 2) * PUSED), rem = i - cb * (S3_TD + 2) * PUSED;
                const int hd = rem / PUSED, lw = cb * SLOTS + hd * PLANE + (rem - hd * PUSED);
#pragma unroll
                for (int p = 0; p < NP; ++p) Xs[p * CB * SLOTS + lw] = (u32x4){pk[p][0], pk[p][1], pk[p][2], pk[p][3]};
            }
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + S3_THREADS * it;
            if (i < WCH) Ws[i] = wv[it];
        }
    };

    set_tile(t_lo);
    load_chunk(0);
    if constexpr (NP == 2) { publish_max(); __syncthreads(); }
    store_chunk(0, true);
    for (int tile = t_lo; tile < t_hi; tile += t_step) {
    const int cd0 = d0, ch0 = h0, cw0 = w0, cbt = bt;           // the tile being computed (the staging tile moves on under its last chunk)
    __syncthreads();                                            // chunk 0 of this tile is in LDS
    f32x4 acc[NCT][ROWS];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) acc[ct][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float inv_fin = 1.0f;
    for (int q = 0; q < Q; ++q) {
        const bool last = q + 1 == Q;                           // wave-uniform
        if (last) set_tile(tile + t_step);
        // unconditional (past the block's last tile every lane is out of range and nothing is fetched): a branch around the prefetch makes
        // the compiler wait for it at the join, right after it was issued (s_waitcnt vmcnt(0) in front of the first MFMA, seen in the ISA)
        load_chunk(last ? 0 : q + 1);
        // NP = 3: the MFMA chains run into the tile's accumulators.  NP = 2: a chain lives for this chunk (its operands carry this chunk's
        // scale) and is folded into the tile's accumulators below.
        f32x4 accq[NCT][ROWS];                                  // (NP = 3 / RUN: unused, eliminated)
        if constexpr (NP == 2 && !RUN) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < ROWS; ++r) accq[ct][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (NP == 2 && RUN) {
            if (ratio != 1.0f) {                                // wave-uniform: this chunk raised the tile's running maximum
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) acc[ct][r] *= ratio;
            }
        }
        f32x4 (&A_)[NCT][ROWS] = s3_sel<(NP == 2 && !RUN)>(accq, acc);
        // ---- NS K-steps x (ROWS + 2) haloed rows: NP B pieces per row, up to 3 kh x NCT x NPROD piece products per read set.
        // The B pieces of row hr + 1 are requested before the MFMAs of row hr (register double buffer; sched_barrier pins the order:
        // unpinned, the compiler sinks every ds_read to right before its first use and the wave stalls on LDS latency once per row).
        if (!S3_DBG(dbg, 2)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            u32x4 a[3][NP][NCT], bf[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[0][p] = Xs[p * CB * SLOTS + xoff[s]];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) a[kh][p][ct] = Ws[(((s * 3 + kh) * NP + p) * NCT + ct) * 64 + lane];
#pragma unroll
            for (int hr = 0; hr < HR; ++hr) {
                if (hr + 1 < HR) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) bf[(hr + 1) & 1][p] = Xs[p * CB * SLOTS + xoff[s] + (hr + 1) * S3_HWV];
                }
                __builtin_amdgcn_sched_barrier(0);
                // consecutive MFMAs go to different accumulators (rows hr, hr-1, hr-2)
#pragma unroll
                for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int row = hr - kh;
                        if (row >= 0 && row < ROWS) {
#pragma unroll
                            for (int ct = 0; ct < NCT; ++ct) A_[ct][row] = P::mfma(a[kh][P::PA[t]][ct], bf[hr & 1][P::PB[t]], A_[ct][row]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (C::RT) {
            u32x4 a[NP][NCT], bf[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[0][p] = Xs[p * CB * SLOTS + xoff_rt];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) a[p][ct] = Ws[((NS * 3 * NP + p) * NCT + ct) * 64 + lane];
#pragma unroll
            for (int row = 0; row < ROWS; ++row) {
                if (row + 1 < ROWS) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) bf[(row + 1) & 1][p] = Xs[p * CB * SLOTS + xoff_rt + (row + 1) * S3_HWV];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) A_[ct][row] = P::mfma(a[P::PA[t]][ct], bf[row & 1][P::PB[t]], A_[ct][row]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }
        keep_offsets();
        if constexpr (NP == 2) {
            if constexpr (!RUN) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < ROWS; ++r)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ct][r][j] = __builtin_fmaf(accq[ct][r][j], inv_cur, acc[ct][r][j]);
            } else if (last) {
                inv_fin = inv_cur;                  // the scale the finished accumulators carry (store_chunk below moves on to the next tile)
            }
            publish_max();                          // of the chunk in flight (q + 1, or chunk 0 of the next tile)
        }
        __syncthreads();                            // every wave is done reading chunk q
        if (!last) {
            store_chunk(q + 1);
            __syncthreads();
        } else if (tile + t_step < t_hi) {
            store_chunk(0, true);                   // chunk 0 of the next tile; the barrier at the top of the tile loop publishes it
        }
    }

    // ---- epilogue: the D layout of the bf16 MFMA is the fp32 one's (lane (kg, n): channels 4 kg + j of voxel column n):
    // bias + LeakyReLU (+ fused leaky_relu_backward mask), planar fp32 store, shared with the fp32-MFMA kernels
    if constexpr (NP == 2) {
        float inv_w = __uint_as_float(__builtin_amdgcn_readfirstlane((int)wp[(size_t)gridDim.y * Q * WCH].x));      // trailer of the packed operator
        if constexpr (RUN) inv_w *= inv_fin;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[ct][r] *= inv_w;
    }
    const int d = cd0 + wave, w = cw0 + n;
    float bz[NCT][4];
    conv_load_bias<NCT>(bz, bias, Cout, g, kg);
    if (lay & VXM_S3_OUT_BLOCKED) {
        if constexpr (NP == 2 && CB == 1)
            conv_epilogue_store_blocked<NCT, ROWS>(acc, y + (size_t)cbt * y_bs, bz, s3_mask_of_sample(mask, cbt, mask_bs, lay), act_slope, mask_slope, Cout, g,
                                                   kg, d < D && w < W, (d * H + ch0) * W + w, ch0, H, W, V, lay);
    } else if (S3_DBG(dbg, 8))
        conv_epilogue_store<NCT, ROWS, 1, 2>(acc, y + (size_t)cbt * y_bs, bz, mask ? mask + (size_t)cbt * mask_bs : nullptr, act_slope, mask_slope, Cout, g, kg,
                                             d < D && w < W, (d * H + ch0) * W + w, ch0, H, W, V);
    else
    conv_epilogue_store<NCT, ROWS, 1>(acc, y + (size_t)cbt * y_bs, bz, mask ? mask + (size_t)cbt * mask_bs : nullptr, act_slope, mask_slope, Cout, g, kg,
                                      d < D && w < W && !S3_DBG(dbg, 4), (d * H + ch0) * W + w, ch0, H, W, V);
    }
}

// ------------------------------------------------------------------------------------------
// k_s3p_conv: the same convolution (fp16 pieces, 8-channel chunks, 8 x 8 x 16 tile, running scale) with PRODUCER and CONSUMER waves.
// ------------------------------------------------------------------------------------------
// Why.  k_s3_conv alternates, inside every block, "multiply chunk q" and "wait for chunk q + 1, split it, write it to LDS"; its time on
// the full-resolution 16 -> 16 layer is the SUM of its parts (profiles/r04y_conv_kernel_experiments.txt: 0.44 ms = reads 0.16 + MFMA 0.05 +
// stores 0.08 + split / weights / barriers 0.15) although two blocks share a CU.  Here one block of 16 waves owns the CU: waves 8 .. 15
// only fetch, split and write (chunk k + 2 in flight from HBM while chunk k + 1 is split into the second LDS buffer), waves 0 .. 7 only
// multiply chunk k out of the first buffer and store finished tiles; the split arithmetic runs beside the MFMAs of other waves of the same
// SIMD and there is ONE barrier per chunk.  Measured (same file): -4 .. -5 % on the forward launches, and -- because the 32 blocks of an
// XCD now walk neighbouring tiles in step -- memory-side reads of 558 MB instead of 844 MB per launch for 440 MB of input (TCC_EA0_RDREQ,
// all 128-byte requests) and writes of 448 instead of 543 MB.  What bounds both kernels is the vector L1: ~23 M 64-byte requests per
// launch at ~510 cycles average L2 latency with the TCP stalled on pending data half of the time, i.e. ~12 GB/s per CU; a haloed 72-byte
// row of a planar tensor costs three sector requests (with the addresses of a channel-blocked tensor the same kernel runs 20 - 28 % faster).
// The chunk stream of a block: c = (tile i of the block, chunk q of the tile), c = 0 .. nphase - 1.  Phase k (k = -1 .. nphase - 1):
//   producers: request the weights of chunk k + 1 and the raw fp32 of chunk k + 2 (register set k & 1), split chunk k + 1 (set (k + 1) & 1,
//              scale from the maxima published in phase k - 1) into LDS buffer (k + 1) & 1, wait for chunk k + 2, publish its wave maxima
//   consumers: multiply chunk k out of buffer k & 1; after the barrier, if it was the tile's last chunk: epilogue
// LDS: 2 x (pieces 61,440 + weights 14,336) + tables = 151,680 bytes.  Same packed operator, same epilogue, same results as k_s3_conv<1,8,1,2,true>.
constexpr int S3P_THREADS = 1024, S3P_PT = 512;                  // threads of the block / of its producer half
template <int NCT> struct S3PCfg {
    using C = S3Cfg<NCT, 8, 1, 2>;
    static constexpr int BUF = C::XW + C::WCH;                   // 16-byte words of one buffer: the chunk's pieces, then its weights
    static constexpr int NI = (C::NSLOT + S3P_PT - 1) / S3P_PT, WIT = (C::WCH + S3P_PT - 1) / S3P_PT;
    static constexpr int LDS_BYTES = 2 * BUF * 16 + 128;         // + [2][8] wave maxima, [2]{ratio, 1 / scale}
};

template <int NCT, bool BLK = false>
__global__ void __launch_bounds__(S3P_THREADS)
k_s3p_conv(ConvIn in, const u32x4* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y, long long y_bs, int Cout,
           float act_slope, const float* __restrict__ mask, long long mask_bs, float mask_slope, int B, int D, int H, int W, int Q0, int Q, int lay, int dbg) {
    using PC = S3PCfg<NCT>;
    using C = typename PC::C;
    using P = S3P<2>;
    VXM_DYN_SMEM(u32x4, smem);
    constexpr int ROWS = 8, HR = C::HR, PLANE = C::PLANE, PUSED = C::PLANE_USED, SLOTS = C::SLOTS, NSLOT = C::NSLOT, NS = C::NS, WCH = C::WCH;
    constexpr int BUF = PC::BUF, NI = PC::NI, WIT = PC::WIT;
    float* const Tab = reinterpret_cast<float*>(smem + 2 * BUF);        // [0..15]: wave maxima [parity][producer wave]; [16..19]: {ratio, 1 / scale}[parity]
    const int tid_ = threadIdx.x, lane = tid_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);

    // the tiles of this block (as k_s3_conv: XCD x owns a contiguous tile range, its blocks take the tiles round-robin)
    const int nw = (W + 15) / 16, nh = (H + ROWS - 1) / ROWS, nd = (D + S3_TD - 1) / S3_TD;
    const int ntiles = B * nd * nh * nw;
    int t_lo, t_hi, t_step;
    if (ntiles >= 64) {
        const int x = blockIdx.x & 7;
        t_lo = (int)((long long)ntiles * x / 8) + (int)(blockIdx.x >> 3); t_hi = (int)((long long)ntiles * (x + 1) / 8); t_step = (int)(gridDim.x >> 3);
    } else {
        t_lo = blockIdx.x; t_hi = ntiles; t_step = gridDim.x;
    }
    const int n_my = t_lo < t_hi ? (t_hi - t_lo + t_step - 1) / t_step : 0;
    const int nphase = n_my * Q;
    const int g = blockIdx.y;
    const int V = D * H * W;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const int V0 = in.up0 ? Dl * Hl * Wl : V;
    auto tile_origin = [&](int tl_, int& bt, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
        const int tl = (lay & S3_REVERSE_TILES) ? ntiles - 1 - tl_ : tl_;       // walk the tensor from its end (see S3_REVERSE_TILES)
        const int tw = tl % nw; int tq = tl / nw;
        const int th = tq % nh; tq /= nh;
        const int td = tq % nd;
        bt = tq / nd; d0 = td * S3_TD; h0 = th * ROWS; w0 = tw * 16;
    };

    if (wave >= 8) {
        // =========================== producers ===========================
        const int ptid_ = tid_ - S3P_PT, pw = wave - 8;
        int d0 = 0, h0 = 0, w0 = 0;
        __amdgpu_buffer_rsrc_t r0, r1;
        int spos[NI];
        auto set_tile = [&](int tile) __attribute__((always_inline)) {
            int ptid = ptid_;
            asm volatile("" : "+v"(ptid));
            const bool live = tile < t_hi;                        // past the block's last tile: every slot is padding, nothing is fetched
            int bt;
            tile_origin(live ? tile : t_lo, bt, d0, h0, w0);
            r0 = vxm_rsrc(in.x0 + (size_t)bt * in.bs0, (unsigned)in.C0 * (unsigned)V0 * 4u);
            r1 = vxm_rsrc(in.C1 ? in.x1 + (size_t)bt * in.bs1 : in.x0, (unsigned)in.C1 * (unsigned)V * 4u);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int i = ptid + S3P_PT * j;
                const int hd = i / PUSED, r2 = i - hd * PUSED, hh = r2 / S3_HWV, hw = r2 - hh * S3_HWV;
                const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
                const bool ok = live && i < NSLOT && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
                spos[j] = ok ? (hd << 10 | hh << 5 | hw) : -1;
            }
        };
        float xr[2][NI][8];                                       // raw fp32 of two chunks: set = chunk parity
        int voffs[2][NI];
        auto load_chunk = [&](auto set_, int q) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value;
            const bool s0 = q < Q0;                               // wave-uniform
            const bool up = s0 && in.up0;
            const __amdgpu_buffer_rsrc_t r = s0 ? r0 : r1;
            const int Cseg = s0 ? in.C0 : in.C1, cbg = s0 ? q : q - Q0, Vs = up ? V0 : V;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int gd = d0 - 1 + (spos[j] >> 10), gh = h0 - 1 + ((spos[j] >> 5) & 31), gw = w0 - 1 + (spos[j] & 31);
                const int sv = up ? ((gd >> 1) * Hl + (gh >> 1)) * Wl + (gw >> 1) : (gd * H + gh) * W + gw;
                const bool ok = spos[j] >= 0 && cbg * 8 < Cseg && !S3_DBG(dbg, 1);  // segments carry multiples of 8 channels
                voffs[S][j] = ok ? (cbg * 8 * Vs + sv) << 2 : VXM_OOB;
                if constexpr (BLK) {                               // channel-blocked input [C / 8][voxel][8] (see k_s3_conv)
                    voffs[S][j] = ok ? (cbg * Vs + sv) << 5 : VXM_OOB;
                    const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffs[S][j], 0, 0));
                    const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffs[S][j], 16, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { xr[S][j][e] = lo[e]; xr[S][j][4 + e] = hi[e]; }
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) xr[S][j][e] = vxm_bload(r, voffs[S][j], (e * Vs) << 2);
            }
        };
        auto publish_max = [&](auto set_) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value;
            const float m = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) f(xr[S][j][e]);
            });
            if (lane == 0) Tab[8 * S + pw] = m;
        };
        int E_run = 15;
        // split chunk (set S) of operator chunk q into buffer S: weights first (requested by `wv` BEFORE the raw loads of this phase,
        // so that waiting for them leaves those loads in flight), then pieces
        auto split_chunk = [&](auto set_, const u32x4 (&wv)[WIT], bool first) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value;
            int ptid = ptid_;
            asm volatile("" : "+v"(ptid));
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(Tab + 8 * S), m1 = *reinterpret_cast<const f32x4*>(Tab + 8 * S + 4);
            const float mx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
            int E = (int)(__float_as_uint(mx) >> 23) & 255;
            E = E < 15 ? 15 : E;
            const int E_new = first ? E : (E > E_run ? E : E_run), dE = E_new - E_run;
            const float ratio = (first || dE == 0) ? 1.0f : (dE > 126 ? 0.0f : __uint_as_float((unsigned)(127 - dE) << 23));
            E_run = E_new;
            const float sc = __uint_as_float((unsigned)(268 - E_run) << 23), inv = __uint_as_float((unsigned)(E_run - 14) << 23);
            if (ptid == 0) { Tab[16 + 2 * S] = ratio; Tab[17 + 2 * S] = inv; }
            u32x4* const Xs = smem + S * BUF;
            u32x4* const Ws = Xs + C::XW;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                unsigned pk[2][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) s3_split2_f16(xr[S][j][2 * e], xr[S][j][2 * e + 1], sc, pk[0][e], pk[1][e]);
                const int i = ptid + S3P_PT * j;
                if (i < NSLOT) {                                   // (padding slots are written too: zeros from the out-of-range loads)
                    const int hd = i / PUSED, lw = hd * PLANE + (i - hd * PUSED);
#pragma unroll
                    for (int p = 0; p < 2; ++p) Xs[p * SLOTS + lw] = (u32x4){pk[p][0], pk[p][1], pk[p][2], pk[p][3]};
                }
            }
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int i = ptid + S3P_PT * it;
                if (i < WCH) Ws[i] = wv[it];
            }
        };
        auto load_weights = [&](u32x4 (&wv)[WIT], int q) __attribute__((always_inline)) {
            int ptid = ptid_;
            asm volatile("" : "+v"(ptid));
            const __amdgpu_buffer_rsrc_t rw = vxm_rsrc(reinterpret_cast<const float*>(wp + ((size_t)g * Q + q) * WCH), WCH * 16u);
#pragma unroll
            for (int it = 0; it < WIT; ++it)
                wv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (ptid + S3P_PT * it) * 16, 0, 0));
        };

        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        int ltile = t_lo, lq = 0;                                 // the next chunk to request
        int sq = 0;                                               // operator chunk of the next chunk to split
        auto advance_load = [&]() __attribute__((always_inline)) {
            if (++lq == Q) { lq = 0; ltile += t_step; set_tile(ltile); }
        };
        set_tile(ltile);
        load_chunk(I0{}, lq);                                     // chunk 0
        advance_load();
        publish_max(I0{});
        __syncthreads();
        // phase k: PAR = (k + 1) & 1 = parity of the chunk that is split; the chunk requested (k + 2) has parity PAR ^ 1
        auto phase = [&](auto par_, int k) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_)::value;
            using SS = std::integral_constant<int, PAR>;
            using SL = std::integral_constant<int, PAR ^ 1>;
            u32x4 wv[WIT];
            const bool do_split = k + 1 < nphase;                 // block-uniform
            load_weights(wv, do_split ? sq : 0);
            load_chunk(SL{}, lq);                                 // past the block's last tile every lane is out of range: nothing is fetched
            advance_load();
            if (do_split) {
                split_chunk(SS{}, wv, sq == 0);
                if (++sq == Q) sq = 0;
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(voffs[PAR ^ 1][j]));      // (see keep_offsets in k_s3_conv)
            publish_max(SL{});
            __syncthreads();
        };
        for (int k = -1; k < nphase; k += 2) {
            phase(I0{}, k);                                       // k odd-from-minus-one: chunk k + 1 is even
            if (k + 1 < nphase) phase(I1{}, k + 1);
        }
        return;
    }

    // =========================== consumers ===========================
    const int kg = lane >> 4, n = lane & 15;
    int xoff[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const S3Unit u0 = s3_unit(1, s, 0), u1 = s3_unit(1, s, 1), u2 = s3_unit(1, s, 2), u3 = s3_unit(1, s, 3);
        const S3Unit u = kg == 0 ? u0 : kg == 1 ? u1 : kg == 2 ? u2 : u3;
        xoff[s] = (wave + u.kd) * PLANE + u.kw + n;
    }
    const int xoff_rt = (wave + 2) * PLANE + (kg < 3 ? kg : 2) * S3_HWV + 2 + n;
    const float inv_w = __uint_as_float(__builtin_amdgcn_readfirstlane((int)wp[(size_t)gridDim.y * Q * WCH].x));      // trailer of the packed operator
    float bz[NCT][4];
    conv_load_bias<NCT>(bz, bias, Cout, g, kg);
    __syncthreads();                                            // (producers: the maxima of chunk 0)
    __syncthreads();                                            // (producers: phase -1, chunk 0 is in buffer 0)
    int k = 0;
    for (int tile = t_lo; tile < t_hi; tile += t_step) {
        int cbt, cd0, ch0, cw0;
        tile_origin(tile, cbt, cd0, ch0, cw0);
        f32x4 acc[NCT][ROWS];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[ct][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float inv_fin = 1.0f;
        for (int q = 0; q < Q; ++q, ++k) {
            const int par = k & 1;
            const u32x4* const Xs = smem + par * BUF;
            const u32x4* const Ws = Xs + C::XW;
            const float ratio = __uint_as_float(__builtin_amdgcn_readfirstlane((int)__float_as_uint(Tab[16 + 2 * par])));
            inv_fin = __uint_as_float(__builtin_amdgcn_readfirstlane((int)__float_as_uint(Tab[17 + 2 * par])));
            if (ratio != 1.0f) {                                // wave-uniform: this chunk raised the tile's running maximum
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < ROWS; ++r) acc[ct][r] *= ratio;
            }
            if (!S3_DBG(dbg, 2)) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                u32x4 a[3][2][NCT], bf[2][2];
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[0][p] = Xs[p * SLOTS + xoff[s]];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) a[kh][p][ct] = Ws[(((s * 3 + kh) * 2 + p) * NCT + ct) * 64 + lane];
#pragma unroll
                for (int hr = 0; hr < HR; ++hr) {
                    if (hr + 1 < HR) {
#pragma unroll
                        for (int p = 0; p < 2; ++p) bf[(hr + 1) & 1][p] = Xs[p * SLOTS + xoff[s] + (hr + 1) * S3_HWV];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh) {
                            const int row = hr - kh;
                            if (row >= 0 && row < ROWS) {
#pragma unroll
                                for (int ct = 0; ct < NCT; ++ct) acc[ct][row] = P::mfma(a[kh][P::PA[t]][ct], bf[hr & 1][P::PB[t]], acc[ct][row]);
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            {
                u32x4 a[2][NCT], bf[2][2];
#pragma unroll
                for (int p = 0; p < 2; ++p) bf[0][p] = Xs[p * SLOTS + xoff_rt];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) a[p][ct] = Ws[((NS * 3 * 2 + p) * NCT + ct) * 64 + lane];
#pragma unroll
                for (int row = 0; row < ROWS; ++row) {
                    if (row + 1 < ROWS) {
#pragma unroll
                        for (int p = 0; p < 2; ++p) bf[(row + 1) & 1][p] = Xs[p * SLOTS + xoff_rt + (row + 1) * S3_HWV];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) acc[ct][row] = P::mfma(a[P::PA[t]][ct], bf[row & 1][P::PB[t]], acc[ct][row]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
            __syncthreads();                                    // chunk k is read; chunk k + 1 is in the other buffer
        }
        // ---- epilogue (as k_s3_conv): undo the scales, bias + LeakyReLU (+ fused leaky_relu_backward mask), planar fp32 store
        const float fin = inv_w * inv_fin;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < ROWS; ++r) acc[ct][r] *= fin;
        const int d = cd0 + wave, w = cw0 + n;
        if (lay & VXM_S3_OUT_BLOCKED) {
            conv_epilogue_store_blocked<NCT, ROWS>(acc, y + (size_t)cbt * y_bs, bz, s3_mask_of_sample(mask, cbt, mask_bs, lay), act_slope, mask_slope, Cout, g,
                                                   kg, d < D && w < W, (d * H + ch0) * W + w, ch0, H, W, V, lay);
        } else
        conv_epilogue_store<NCT, ROWS, 1>(acc, y + (size_t)cbt * y_bs, bz, mask ? mask + (size_t)cbt * mask_bs : nullptr, act_slope, mask_slope, Cout, g, kg,
                                          d < D && w < W && !S3_DBG(dbg, 4), (d * H + ch0) * W + w, ch0, H, W, V);
    }
}

// w: [Cw_out][Cw_in][27] fp32 (reference layout) -> [G][Q]{[NS][kh][piece][NCT][64 lanes], [RT][piece][NCT][64 lanes]} x 8 bf16 / fp16: lane (kg, m) of
// sliding step s / row tap kh / piece p holds, for output channel 16 (g NCT + ct) + m, piece p of the weights of unit s3_unit(CB, s, kg);
// lane (kg, m) of the row-tap step those of tap (kd, kh, kw) = (2, kg, 2) (kg = 3: zeros).
// Operator: y[o] = sum_i Wop[o][i][tap] x[i] over the virtual input channels i (segment 0: [0, seg0), segment 1: the rest);
// chunk q < Q0 holds segment-0 channels 8 CB q + 8 cb + e, chunk q >= Q0 segment-1 channels seg0 + 8 CB (q - Q0) + 8 cb + e.
// forward: Wop[o][i][t] = w[o][ci_lo + i][t]; flip (backward-data onto input channels [ci_lo, ci_lo + OutC)): Wop[o][i][t] = w[i][ci_lo + o][26 - t].
// NP = 2: the values are multiplied by the operator's scale first (k_s3_wmax wrote {1 / scale, scale} into the trailer word `words`).
struct S3PackJob {
    const float* w; u32x4* wp;
    int Cw_in, Cw_out, ci_lo, ci_n, flip, InC, seg0, OutC, NCT, CB, Q0, Q, NP;
    unsigned first_block, words;
};
__device__ __forceinline__ void s3_pack_word(const S3PackJob& jb, size_t i) {
    const int NS = jb.CB == 2 ? 5 : 2, RT = jb.CB == 2 ? 0 : 1, NP = jb.NP;
    const int per_chunk = (NS * 3 + RT) * NP * jb.NCT * 64;
    size_t r = i;
    const int wi = (int)(r % per_chunk); r /= per_chunk;       // word inside the chunk
    const int q = r % jb.Q; const int g = (int)(r / jb.Q);
    const int lane = wi % 64;
    int t2 = wi / 64;
    const int ct = t2 % jb.NCT; t2 /= jb.NCT;                   // t2: (s, kh, piece) of a sliding step, or NS * 3 * NP + piece of the row-tap step
    const int kg = lane >> 4, m = lane & 15;
    int p, tap, cb;
    bool valid;
    if (t2 < NS * 3 * NP) {
        p = t2 % NP;
        const int kh = (t2 / NP) % 3, st = t2 / (3 * NP);
        const S3Unit un = s3_unit(jb.CB, st, kg);
        valid = un.valid != 0; cb = un.cb; tap = un.kd * 9 + kh * 3 + un.kw;
    } else {
        p = t2 - NS * 3 * NP;
        valid = kg < 3; cb = 0; tap = 2 * 9 + kg * 3 + 2;       // (kd, kh, kw) = (2, kg, 2)
    }
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int o = (g * jb.NCT + ct) * 16 + m;
    if (valid && o < jb.OutC) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool s0 = q < jb.Q0;
            const int cl = (s0 ? q : q - jb.Q0) * 8 * jb.CB + cb * 8 + e;          // channel inside its segment
            const int ci = s0 ? cl : jb.seg0 + cl;
            if (cl < (s0 ? jb.seg0 : jb.InC - jb.seg0))
                v[e] = jb.flip ? jb.w[((size_t)ci * jb.Cw_in + jb.ci_lo + o) * 27 + (26 - tap)] : jb.w[((size_t)o * jb.Cw_in + jb.ci_lo + ci) * 27 + tap];
        }
    }
    unsigned pk[3][4];
    if (NP == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s3_split2(v[2 * e], v[2 * e + 1], pk[0][e], pk[1][e], pk[2][e]);
    } else {
        const float sc = __uint_as_float(jb.wp[jb.words].y);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s3_split2_f16(v[2 * e], v[2 * e + 1], sc, pk[0][e], pk[1][e]); pk[2][e] = 0u; }
    }
    jb.wp[i] = (u32x4){pk[p][0], pk[p][1], pk[p][2], pk[p][3]};
}

// Every stale operator of a step in one launch; the job table travels in the kernel arguments, a block finds its job by a scan.
#define S3_PACK_JOBS 40
struct S3PackBatch {
    S3PackJob job[S3_PACK_JOBS];
    int n;
};
__global__ void __launch_bounds__(256) k_s3_pack_weights(const S3PackBatch batch) {
    int j = 0;
    while (j + 1 < batch.n && blockIdx.x >= batch.job[j + 1].first_block) ++j;          // block-uniform
    const S3PackJob& jb = batch.job[j];
    const size_t i = (size_t)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    if (i < jb.words) s3_pack_word(jb, i);
}
// NP = 2: the scale of each operator of the batch -- one block per job takes the largest magnitude of the weights the operator reads
// (w[a][ci_lo + b][*], a < Cw_out, b < ci_n: the same set for the forward operator and its adjoint) and writes {1 / s, s} to the trailer.
__global__ void __launch_bounds__(1024) k_s3_wmax(const S3PackBatch batch) {
    __shared__ float red[16];
    const S3PackJob& jb = batch.job[blockIdx.x];
    if (jb.NP != 2) return;                                                             // block-uniform
    // one block per operator: 1024 threads, four independent loads in flight per thread (the first version -- 256 threads, one dependent
    // load per iteration -- took 57 us for the 41 k weights of rem0 and sat on the step's critical path twice)
    const int n = jb.Cw_out * jb.ci_n * 27, row = jb.ci_n * 27;
    float mm[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto at = [&](int i) __attribute__((always_inline)) {
        if (i >= n) return 0.0f;
        const int a = i / row, r = i - a * row;
        return s3_finite_mag(jb.w[((size_t)a * jb.Cw_in + jb.ci_lo) * 27 + r]);
    };
    for (int i = threadIdx.x; i < n; i += 8192) {                     // eight independent loads in flight per thread (round 6: the launch sits on the main chain)
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = at(i + 1024 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) mm[u] = fmaxf(mm[u], a[u]);
    }
    const float m = s3_wave_max(fmaxf(fmaxf(fmaxf(mm[0], mm[1]), fmaxf(mm[2], mm[3])), fmaxf(fmaxf(mm[4], mm[5]), fmaxf(mm[6], mm[7]))));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, red[i]);
        float s, inv;
        s3_scale_of(mx, s, inv);
        jb.wp[jb.words] = (u32x4){__float_as_uint(inv), __float_as_uint(s), 0u, 0u};
    }
}

// ------------------------------------------------------------------------------------------
// backward-weight on the bf16 pipe: gW[co][ci][tap] = sum_v dZ[co][v] X[ci][v + tap - 1],  gb[co] = sum_v dZ[co][v]
// ------------------------------------------------------------------------------------------
// Replaces convolution_backward w.r.t. weight and bias (autograd twin of networks.py:299) for plain full-resolution tensors.
// The contraction runs over VOXELS: M = 16 output channels (A = dZ^T), N = 16 input channels (B = X), K = 32 voxels of a W row,
// so no K slot is ever padding.  Both operands are split into three bf16 pieces while they are staged from planar fp32 into
// [voxel][16 channel] LDS tiles (32-byte rows), and read with the transposing LDS read of gfx950 (ds_read_b64_tr_b16: a lane
// receives 4 consecutive voxels of one channel; lane pattern probed in tools/probe/tr16_probe.hip) -- the structure of
// k_bf16_conv_bwd_weight (conv_bf16.hip), with six piece products per (A, B) fragment pair.
// Block = 12 waves = (row half) x (depth slice of a 2 x 4 x 32 voxel tile) x (kd) for ONE 16-output-channel tile and ONE
// 16-input-channel chunk; a wave keeps the 9 (kh, kw) taps of its kd in 36 accumulator VGPRs over ALL its tiles and slides over
// the 4 haloed X rows of its two output rows (three kw-shifted B fragment sets per row serve kh = 0, 1, 2; the dZ fragment sets
// of its two rows stay in registers).  (The first version had 6 waves, four rows each: 1.5 waves per SIMD, two SIMDs with twice
// the work of the others -- matrix pipe 31 % busy, measured.)  One persistent block per CU walks down the depth of a (b, th, tw) column: consecutive tiles share two of their
// four haloed X planes in a 6-slot LDS ring, the raw fp32 loads of tile t + 1 are in flight in registers under the MFMAs of tile
// t.  LDS: 3 pieces x (6 planes x 6 x 34 voxels + 2 x 4 x 32 voxels) x 32 B = 142,080 B (three pieces leave no room for a second
// dZ buffer: two barriers per tile).  Partials per block, summed in a fixed order by k_s3_reduce_partials (deterministic).
constexpr int SW_WAVES = 12, SW_THREADS = 64 * SW_WAVES, SW_NSL = 1;     // waves = (row half, depth slice, kd); a block combines its four (depth slice, row half) sums and writes ONE partial slice
constexpr int SW_TD = 2, SW_TH = 4, SW_TW = 32, SW_XW = SW_TW + 2, SW_HR = SW_TH + 2, SW_RING = 6;
constexpr int SW_PLANE = SW_HR * SW_XW * 32;                  // bytes of one haloed X plane of one piece: [hh][hw][16 ci]
constexpr int SW_XPIECE = SW_RING * SW_PLANE;
constexpr int SW_ZPIECE = SW_TD * SW_TH * SW_TW * 32;         // dZ tile of one piece: [ds][row][w][16 co]
constexpr int SW_EPI_BYTES = (SW_WAVES * 9 + 4) * 256 * 4;    // the epilogue's combine buffer (reuses the staging space)
// NP = 3: single dZ buffer (three pieces leave no room for a second).  NP = 2: the dZ tile is double-buffered, and a table of the
// inverse scales of the six ring planes and the two dZ buffers + the wave maxima of the tile in flight follow the tiles.
template <int NP> struct SwCfg {
    static constexpr int ZB = NP == 2 ? 2 : 1;
    static constexpr int TILE_BYTES = NP * (SW_XPIECE + ZB * SW_ZPIECE);
    static constexpr int TAB_BYTES = NP == 2 ? 256 : 0;       // floats: [0..5] ring planes, [6..7] dZ buffers, [8..43] wave maxima (three slots of 12)
    static constexpr int LDS_BYTES = (TILE_BYTES + TAB_BYTES > SW_EPI_BYTES ? TILE_BYTES + TAB_BYTES : SW_EPI_BYTES);
};

struct SwTasks { int ncol, nseg, seg_len, nd, nh, nw; };       // tasks = (column (b, th, tw), depth segment), seg_len tiles each

// x: [B][C][D][H][W] fp32 (batch stride x_bs), dz: [B][Cdz][D][H][W] fp32; grid = NBLK x NCOMBO, combo = (16-channel chunk q of x, 16-channel tile of dz)
template <int NP, bool PIPE2 = true>
__global__ void __launch_bounds__(SW_THREADS, 3) k_s3_bwd_weight(const float* __restrict__ x, long long x_bs, int C, const float* __restrict__ dz,
                                                              long long dz_bs, int Cdz, float* __restrict__ part, int D, int H, int W, int NBLK,
                                                              int NCO, SwTasks tk, int task_rr, int lay, int dbg) {
    using P = S3P<NP>;
    using CF = SwCfg<NP>;
    VXM_DYN_SMEM(char, smem);
    char* const Xs = smem;                                       // [NP pieces][6 ring planes][SW_PLANE]
    char* const Zs = smem + NP * SW_XPIECE;                      // [ZB buffers][NP pieces][SW_ZPIECE]
    float* const Tab = reinterpret_cast<float*>(smem + CF::TILE_BYTES);      // NP = 2: inverse scales + wave maxima
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rh = wave / 6, w6 = wave - 6 * rh, ds = w6 / 3, kd = w6 - 3 * ds;      // rows 2 rh, 2 rh + 1 of depth slice ds, taps kd
    // block -> (task range bx, combo).  The combos of one bx read the same X / dZ tiles at the same pace: their workgroup ids are
    // 8 apart so that they share one XCD's L2 (as k_bf16_conv_bwd_weight)
    const int NCOMBO = gridDim.x / NBLK, xmain = NBLK & ~7;
    int bx, combo;
    if ((int)blockIdx.x < xmain * NCOMBO) {
        const int j = blockIdx.x >> 3;
        combo = j % NCOMBO;
        bx = (j / NCOMBO) * 8 + (blockIdx.x & 7);
    } else {
        const int r = blockIdx.x - xmain * NCOMBO;
        bx = xmain + r / NCOMBO;
        combo = r % NCOMBO;
    }
    const int q = combo / NCO, cot = combo - q * NCO;
    const int ntask = tk.ncol * tk.nseg;
    // Which tasks this block walks.  Tasks are numbered depth-segment-major, columns (b, th, tw) consecutive, so task ids that are one
    // apart are W-NEIGHBOURS at the same depth.  XCD x (= bx & 7: block ids map to XCDs round-robin) owns a contiguous task range and
    // its blocks take the tasks of that range round-robin: at any moment the blocks of an XCD work on adjacent columns at the same depth,
    // so the halo columns of a tile (w0 - 1 and w0 + 32: a 4-byte voxel each, a whole 64-byte sector from HBM) are hits in the XCD's L2
    // when the neighbour already fetched that sector.  (Rounds 3 / early 4 gave every block a contiguous range of tasks: the counters
    // showed 2.0 GB fetched per launch for 0.78 GB of operands.)  VXM_S3_BW_TASKS=range restores that order for A/B.
    int k_lo, k_hi, k_step;
    if ((NBLK & 7) == 0 && task_rr) {
        const int x = bx & 7;
        k_lo = (int)((long long)ntask * x / 8) + (bx >> 3); k_hi = (int)((long long)ntask * (x + 1) / 8); k_step = NBLK >> 3;
    } else {
        k_lo = (int)((long long)ntask * bx / NBLK); k_hi = (int)((long long)ntask * (bx + 1) / NBLK); k_step = 1;
    }
    const int V = D * H * W, HW = H * W;

    f32x4 tot[3][3], totb = {0.f, 0.f, 0.f, 0.f};               // running totals (vector-ALU sums of the per-tile MFMA chains)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) tot[kh][kw] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lp = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;     // lane pattern of the transposing read
    const u32x4 ones = {P::ONES, P::ONES, P::ONES, P::ONES};    // 1.0 x 8: B operand of the bias sum

    // Staging roles of a thread, fixed for the kernel.  A slot is (PAIR of W-neighbouring voxels, 8-channel half): 8 planar fp32 dwordx2 loads
    // (one per channel), split into 2 x NP 16-byte words.  Pairs start at even w (W is even, tiles start at multiples of 32), so a pair is inside
    // the volume or outside it as a whole: the haloed row w0 - 1 .. w0 + 32 is covered by the 18 pairs from w0 - 2, whose first and last voxel
    // are dropped at the LDS write.  A tile stages 2 haloed X planes (2 x 6 rows x 18 pairs x 2 halves = 432 slots) and the dZ tile (2 x 4
    // rows x 16 pairs x 2 = 256 slots) on 768 threads in ONE round: waves 0 .. 6 stage X, waves 7 .. 10 dZ (wave-uniform roles: one
    // descriptor per wave).  What bounded the first versions was the NUMBER of vector-memory instructions -- one dword per lane and channel,
    // 192 wave-loads per tile: with the loads removed the kernel ran 1.40 -> 0.95 ms (rem1), with split + LDS writes removed only
    // 1.51 -> 1.40 -- hence pairs: half the instructions for the same bytes.
    constexpr int SW_XPAIRS = SW_XW / 2 + 1, NXS = SW_HR * SW_XPAIRS * 2, NZS = SW_TD * SW_TH * (SW_TW / 2) * 2;    // 18, 216 per plane, 256
    static_assert(SW_XW % 2 == 0 && 2 * NXS <= 7 * 64 && NZS == 4 * 64 && SW_WAVES >= 11, "slot plan");
    const int s_role = wave <= 6 ? 1 : (wave <= 10 ? 2 : 0);                             // 1 = X pair, 2 = dZ pair, 0 = none (wave-uniform)
    const int s_i = s_role == 1 ? tid : tid - 7 * 64;                                    // X: slot of the plane pair (>= 2 NXS: idle lane); dZ: slot of the tile
    const int x_pl = s_i >= NXS ? 1 : 0, x_r = s_i - x_pl * NXS;                         // X role: plane of the pair, slot inside the plane
    const int z_ds = wave >= 9 ? 1 : 0;                                                  // dZ role: depth slice of the tile (wave-uniform)
    // raw loads in flight: first / second voxel of the pair, 8 channels.  NP = 3: one set (the tile after the one being multiplied).
    // NP = 2: TWO sets -- the scale of a tile is the block's maximum over it, which must be published one barrier before the tile is
    // split; so a tile is requested two phases ahead and split one phase ahead (set = tile parity), and a phase needs ONE barrier.
    constexpr int NSET = NP == 2 ? 2 : 1;
    float ra[NSET][8], rb[NSET][8];
    unsigned ka[NP][4], kb[NP][4];
    int off0 = VXM_OOB, ldst = 0;                                // per task: byte offset inside a depth slice; LDS byte offset of the first voxel's word inside
                                                                 // a plane / the dZ tile, | 1: drop the first voxel, | 2: drop the second
    int vk[NSET];                                                // the offsets the in-flight loads were issued with (kept live until the MFMA phase is over)

    // The task loop is instantiated per STAGING ROLE of the wave (XR: stages X / nothing, else the dZ tile) and per layout of the tensor that
    // role stages (BLK), so that role and layout are compile-time inside it: a planar and a channel-blocked operand need different load
    // instructions, and a run-time choice between them would put branches around loads that must stay in flight across the MFMA phase.
    // Every instance executes the same barriers; a wave runs exactly one of them for the whole kernel.
    auto run_tasks = [&](auto xr_, auto bl_) __attribute__((always_inline)) {
        constexpr bool XR = decltype(xr_)::value, BLK = decltype(bl_)::value;
        for (int task = k_lo; task < k_hi; task += k_step) {
            const int seg = task_rr ? task / tk.ncol : task % tk.nseg, col = task_rr ? task - seg * tk.ncol : task / tk.nseg;      // (depth-segment-major when round-robin)
            const int tw = col % tk.nw; int cq = col / tk.nw;
            const int th = cq % tk.nh; const int b = cq / tk.nh;
            const int td0 = seg * tk.seg_len, ntile = min(tk.seg_len, tk.nd - td0);
            const int dbase = td0 * SW_TD, h0 = th * SW_TH, w0 = tw * SW_TW;
            // ONE descriptor per wave, chosen by its (wave-uniform) staging role here, on scalars: selecting between two descriptors inside the
            // load lambda made the compiler keep both in scratch memory and pick one through a pointer + a waterfall loop (seen in the ISA)
            constexpr bool xrole = XR;                             // (= s_role != 2: the whole task loop is instantiated per staging role, see below)
            // blk: the tensor this wave stages (x: lay & VXM_S3_IN0_BLOCKED, dz: & VXM_S3_IN1_BLOCKED) is channel-blocked [C / 8][voxel][8].  A slot
            // (pair of W neighbours, 8 channels) is then 64 contiguous bytes fetched by four 16-byte loads instead of eight 8-byte loads from
            // eight planes, and a haloed X row is one run of 1152 bytes instead of sixteen of 144 (19 instead of 32 sector requests).
            constexpr bool blk = BLK;
            constexpr int lsh = blk ? 5 : 2;                           // byte offset of a voxel inside its plane / 8-channel block = voxel << lsh
            const __amdgpu_buffer_rsrc_t rd = vxm_rsrc(xrole ? x + (size_t)b * x_bs : dz + (size_t)b * dz_bs, (unsigned)(xrole ? C : Cdz) * (unsigned)V * 4u);
            if (s_role == 1) {
                const int cb = x_r & 1, pr = x_r >> 1, hh = pr / SW_XPAIRS, pp = pr - hh * SW_XPAIRS;
                const int gh = h0 - 1 + hh, gw = w0 - 2 + 2 * pp;
                const bool live = s_i < 2 * NXS && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W && q * 16 + cb * 8 < C;
                off0 = live ? (blk ? ((q * 2 + cb) * V + gh * W + gw) << 5 : ((q * 16 + cb * 8) * V + gh * W + gw) << 2) : VXM_OOB;
                ldst = ((hh * SW_XW + 2 * pp - 1) * 32 + cb * 16) | (pp == 0 ? 1 : 0) | (pp == SW_XPAIRS - 1 ? 2 : 0);
            } else if (s_role == 2) {
                const int cb = s_i & 1, r2 = (s_i >> 1) & (SW_TH * SW_TW / 2 - 1), zh = r2 / (SW_TW / 2), zw = 2 * (r2 - zh * (SW_TW / 2));
                const bool live = h0 + zh < H && w0 + zw < W && cot * 16 + cb * 8 < Cdz;
                off0 = live ? (blk ? ((cot * 2 + cb) * V + (z_ds * H + h0 + zh) * W + w0 + zw) << 5 : ((cot * 16 + cb * 8) * V + (z_ds * H + h0 + zh) * W + w0 + zw) << 2)
                            : VXM_OOB;
                ldst = ((z_ds * SW_TH + zh) * SW_TW + zw) * 32 + cb * 16;
            }

            // planes p0, p0 + 1 of this task (plane p = global depth dbase - 1 + p) and the dZ tile tz (tz < 0: none) -> registers.  Branch-free:
            // a plane outside the volume ORs the out-of-range bit into the lane offsets (selects on wave-uniform conditions would become
            // branches around the loads, and the joins behind them make the compiler wait for the prefetch at the START of the MFMA phase)
            auto load_tile = [&](auto set_, int p0, int tz) __attribute__((always_inline)) {
                constexpr int S = decltype(set_)::value;
                const int gd0 = dbase - 1 + p0, gd1 = gd0 + 1;           // wave-uniform
                int o0 = (unsigned)gd0 < (unsigned)D ? (gd0 * HW) << lsh : 0, f0 = (unsigned)gd0 < (unsigned)D ? 0 : VXM_OOB;
                int o1 = (unsigned)gd1 < (unsigned)D ? (gd1 * HW) << lsh : 0, f1 = (unsigned)gd1 < (unsigned)D ? 0 : VXM_OOB;
                const int gz = dbase + (tz < 0 ? 0 : tz) * SW_TD;         // first depth slice of the dZ tile (always < D)
                int fz = tz < 0 ? VXM_OOB : 0, fz1 = (tz < 0 || gz + 1 >= D) ? VXM_OOB : 0;
                asm volatile("" : "+v"(f0), "+v"(f1), "+v"(fz), "+v"(fz1));
                vk[S] = xrole ? ((off0 + (x_pl ? o1 : o0)) | (x_pl ? f1 : f0)) : (off0 | (z_ds ? fz1 : fz));
                const int sb = xrole ? 0 : (gz * HW) << lsh;
                if (S3_DBG(dbg, 1) && xrole) vk[S] = VXM_OOB;        // timing experiments: no X reads / no dZ reads
                if (S3_DBG(dbg, 4) && !xrole) vk[S] = VXM_OOB;
                if constexpr (blk) {
    #pragma unroll
                    for (int k = 0; k < 2; ++k) {                    // first voxel: channels 4 k .. 4 k + 3; second voxel 32 bytes on
                        const f32x4 ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vk[S], sb + 16 * k, 0));
                        const f32x4 tb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vk[S], sb + 32 + 16 * k, 0));
    #pragma unroll
                        for (int e = 0; e < 4; ++e) { ra[S][4 * k + e] = ta[e]; rb[S][4 * k + e] = tb[e]; }
                    }
                } else {
    #pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const f32x2 t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rd, vk[S], sb + ((e * V) << 2), 0));
                        ra[S][e] = t2.x; rb[S][e] = t2.y;
                    }
                }
            };
            // NP = 2: the largest magnitude this wave loaded for a tile -> Tab[8 + 12 slot + wave]; after the next barrier every thread
            // takes the maximum over the X waves (0 .. 6) resp. the dZ waves (7 .. 10) and derives the scale of the plane pair / the dZ tile
            auto publish_max = [&](auto set_, int slot) __attribute__((always_inline)) {
                constexpr int S = decltype(set_)::value;
                if constexpr (NP == 2) {
                    const float m = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
    #pragma unroll
                        for (int e = 0; e < 8; ++e) { f(ra[S][e]); f(rb[S][e]); }
                    });
                    if (lane == 0) Tab[8 + 12 * slot + wave] = m;
                }
            };
            float sc_role = 1.0f;                                    // NP = 2: scale of what this thread staged (X pair of planes, or the dZ tile)
            auto take_scales = [&](int slot, int p0, int zbuf) __attribute__((always_inline)) {
                if constexpr (NP == 2) {
                    const f32x4* const t4 = reinterpret_cast<const f32x4*>(Tab + 8 + 12 * slot);
                    const f32x4 m0 = t4[0], m1 = t4[1], m2 = t4[2];
                    const float mxx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), m1.z));
                    const float mxz = fmaxf(fmaxf(m1.w, m2.x), fmaxf(m2.y, m2.z));
                    float sx, ix, sz, iz;
                    s3_scale_of(mxx, sx, ix);
                    s3_scale_of(mxz, sz, iz);
                    sc_role = s_role == 2 ? sz : sx;
                    if (tid == 0) { Tab[p0 % SW_RING] = ix; Tab[(p0 + 1) % SW_RING] = ix; }
                    if (tid == 64 && zbuf >= 0) Tab[6 + zbuf] = iz;
                }
            };
            // split (registers only) and write (LDS) are separate steps: a wave splits the tile it prefetched right after ITS OWN MFMA loop
            auto split_tile = [&](auto set_) __attribute__((always_inline)) {
                constexpr int S = decltype(set_)::value;
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (NP == 3) {
                        s3_split2(ra[S][2 * e], ra[S][2 * e + 1], ka[0][e], ka[1][e], ka[2][e]);
                        s3_split2(rb[S][2 * e], rb[S][2 * e + 1], kb[0][e], kb[1][e], kb[2][e]);
                    } else {
                        s3_split2_f16(ra[S][2 * e], ra[S][2 * e + 1], sc_role, ka[0][e], ka[1][e]);
                        s3_split2_f16(rb[S][2 * e], rb[S][2 * e + 1], sc_role, kb[0][e], kb[1][e]);
                    }
                }
            };
            // X planes go to ring slots the current tile does not read, so their writes need no barrier: they are issued right after the wave's own
            // MFMA loop and overlap the MFMAs of the waves still computing (waves 0 .. 6 stage X and are the first to finish: the matrix pipe of
            // a SIMD serves its oldest wave first).  The dZ tile is single-buffered and is written between the two barriers.  Measured with
            // s_memtime stamps: the phase between the barriers 1620 -> 430 cycles of ~11.7 k per tile, 2.50 -> 2.44 M cycles per launch on rem1.
            // (NP = 2: the dZ tile is double-buffered too, so both writes go to free space and the phase has ONE barrier; the block's scale the
            // split needs was published one phase earlier -- see the raw sets above.)
            auto write_x = [&](int p0) __attribute__((always_inline)) {
                if (s_role == 1) {
                    if (s_i < 2 * NXS) {
                        char* const d = Xs + ((p0 + x_pl) % SW_RING) * SW_PLANE + (ldst & ~3);
                        if (!(ldst & 1)) {
    #pragma unroll
                            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(d + p * SW_XPIECE) = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                        }
                        if (!(ldst & 2)) {
    #pragma unroll
                            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(d + p * SW_XPIECE + 32) = (u32x4){kb[p][0], kb[p][1], kb[p][2], kb[p][3]};
                        }
                    }
                }
            };
            auto write_z = [&](int zbuf) __attribute__((always_inline)) {
                if (s_role == 2) {
    #pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        *reinterpret_cast<u32x4*>(Zs + (zbuf * NP + p) * SW_ZPIECE + ldst) = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                        *reinterpret_cast<u32x4*>(Zs + (zbuf * NP + p) * SW_ZPIECE + ldst + 32) = (u32x4){kb[p][0], kb[p][1], kb[p][2], kb[p][3]};
                    }
                }
            };

            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, NP == 2 ? 1 : 0>;
            // the MFMA phase of tile t: this wave's chains, folded into the running totals
            auto mfma_tile = [&](int t) __attribute__((always_inline)) {
                if (S3_DBG(dbg, 2)) return;
                const char* const xp = Xs + ((2 * t + ds + kd) % SW_RING) * SW_PLANE;
                const int zcur = NP == 2 ? (t & 1) : 0;              // dZ buffer of this tile
                float unscale_x = 1.0f, unscale_z = 1.0f;
                if constexpr (NP == 2) { unscale_x = Tab[(2 * t + ds + kd) % SW_RING]; unscale_z = Tab[6 + zcur]; }
                // The fp32 accumulation of the bf16 MFMA TRUNCATES (tools/bw_accuracy.py: a chain of ~10^4 MFMAs into one accumulator drifts,
                // 1.5e-5 against 4e-6 for the fp32 MFMA on cancelling sums).  So an MFMA chain lives for ONE tile -- it starts from zero and is
                // at most 12 links long -- and is then added to the running totals by the vector ALU (round to nearest, unbiased).
                u32x4 az[2][NP];                                     // dZ fragment sets (NP pieces) of this wave's two output rows
                f32x4 accb = {0.f, 0.f, 0.f, 0.f};
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                __builtin_amdgcn_s_setprio(3);
    #pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
    #pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const int zb = (zcur * NP + p) * SW_ZPIECE + ((ds * SW_TH + 2 * rh + hl) * SW_TW) * 32 + lp;
                        const u32x2 lo = s3_tr_read(Zs, zb), hi = s3_tr_read(Zs, zb + 16 * 32);
                        az[hl][p] = (u32x4){lo.x, lo.y, hi.x, hi.y};
                    }
                    if (kd == 0) {                                   // wave-uniform: bias gradient = sum of the pieces against ones
    #pragma unroll
                        for (int p = 0; p < NP; ++p) accb = P::mfma(az[hl][p], ones, accb);
                    }
                }
                // kw outermost: the three kh chains of one kw are alive at a time (12 accumulator registers instead of 36 -- the second raw set of
                // the NP = 2 pipeline needs the room); every (haloed row, kw) B fragment set is still read exactly once
    #pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    // a wave's priority falls as it advances through the tile: the three waves of a SIMD then progress evenly instead of oldest
                    // first and share the matrix pipe to the end of the phase -- a lone last wave cannot keep it busy (s_memtime stamps, round 3:
                    // the waves finished at 3.5 k, 5.7 k and 7.7 k cycles of a tile whose MFMAs need 5.5 k; -2 .. -3 % on the three big launches)
                    if (kw == 1) __builtin_amdgcn_s_setprio(2); else if (kw == 2) __builtin_amdgcn_s_setprio(1);
                    f32x4 acc[3];
    #pragma unroll
                    for (int hl = 0; hl < 4; ++hl) {                 // haloed rows 2 rh + hl serve output rows 2 rh + hl - kh
                        const int hr = 2 * rh + hl;                  // wave-uniform
                        u32x4 bxf[NP];
    #pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const int xb = p * SW_XPIECE + (hr * SW_XW + kw) * 32 + lp;
                            const u32x2 lo = s3_tr_read(xp, xb), hi = s3_tr_read(xp, xb + 16 * 32);
                            bxf[p] = (u32x4){lo.x, lo.y, hi.x, hi.y};
                        }
    #pragma unroll
                        for (int tp = 0; tp < P::NPROD; ++tp)
    #pragma unroll
                            for (int kh = 0; kh < 3; ++kh) {
                                const int rl = hl - kh;              // local output row
                                if (rl >= 0 && rl < 2)               // the first link of a tile's chain (output row 0, first product) starts from zero
                                    acc[kh] = P::mfma(az[rl][P::PA[tp]], bxf[P::PB[tp]], (rl == 0 && tp == 0) ? zero4 : acc[kh]);
                            }
                    }
                    if constexpr (NP == 3) {
    #pragma unroll
                        for (int kh = 0; kh < 3; ++kh) tot[kh][kw] += acc[kh];
                    } else {                                         // the chain carries the scales of its X plane and its dZ tile: undone here (exact)
    #pragma unroll
                        for (int kh = 0; kh < 3; ++kh)
    #pragma unroll
                            for (int j = 0; j < 4; ++j) tot[kh][kw][j] = __builtin_fmaf(acc[kh][j] * unscale_x, unscale_z, tot[kh][kw][j]);
                    }
                }
                if constexpr (NP == 3) {
                    totb += accb;
                } else {
    #pragma unroll
                    for (int j = 0; j < 4; ++j) totb[j] = __builtin_fmaf(accb[j], unscale_z, totb[j]);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            };

            __syncthreads();                                        // every wave is done with the previous task
            if constexpr (NP == 3) {
                load_tile(I0{}, 0, 0);
                split_tile(I0{});
                write_x(0);
                write_z(0);
                load_tile(I0{}, 2, -1);
                split_tile(I0{});
                write_x(2);
                __syncthreads();
                for (int t = 0; t < ntile; ++t) {
                    const bool more = t + 1 < ntile;                 // wave-uniform
                    // the raw loads of tile t + 1 (2 X planes + the dZ tile) are in flight under the MFMAs of tile t; unconditional: past
                    // the last tile the planes lie beyond this task's depth range and are simply not stored
                    load_tile(I0{}, 2 * t + 4, more ? t + 1 : t);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_tile(t);
                    // the address registers of the loads in flight stay live until here: reused earlier, the compiler guards every reuse
                    // with s_waitcnt vmcnt(..), i.e. waits for the prefetch at the start of the MFMA phase (seen in the ISA)
                    asm volatile("" ::"v"(vk[0]));
                    split_tile(I0{});                                // before the barrier: overlaps the other waves' MFMAs
                    if (more) write_x(2 * t + 4);
                    __syncthreads();                                 // every wave is done reading tile t (X ring slots of t - 1 and the dZ tile are free)
                    if (more) write_z(0);
                    __syncthreads();
                }
            } else if constexpr (!PIPE2) {
                // ---- NP = 2, the first version (kept for same-box A/B, VXM_S3_BW_PIPE=1): one tile ahead, the maxima published before a first
                // barrier, split + writes between it and a second one
                load_tile(I0{}, 0, 0);
                publish_max(I0{}, 0);
                __syncthreads();
                take_scales(0, 0, 0);
                split_tile(I0{});
                write_x(0);
                write_z(0);
                load_tile(I0{}, 2, -1);
                publish_max(I0{}, 2);
                __syncthreads();
                take_scales(2, 2, -1);
                split_tile(I0{});
                write_x(2);
                __syncthreads();
                for (int t = 0; t < ntile; ++t) {
                    const bool more = t + 1 < ntile;
                    load_tile(I0{}, 2 * t + 4, more ? t + 1 : t);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_tile(t);
                    asm volatile("" ::"v"(vk[0]));
                    publish_max(I0{}, 0);
                    __syncthreads();
                    if (more) {
                        take_scales(0, 2 * t + 4, (t & 1) ^ 1);
                        split_tile(I0{});
                        write_x(2 * t + 4);
                        write_z((t & 1) ^ 1);
                    }
                    __syncthreads();
                }
            } else {
                // ---- NP = 2.  Tile tau (tau >= 1) = planes 2 tau + 2, 2 tau + 3 and dZ tile tau; its raw loads live in set tau & 1, its maxima in
                // table slot tau & 1 (tile 0's second plane pair: slot 2).  Phase t: request tile t + 2, multiply tile t, split + write tile
                // t + 1 (scale from the maxima published during phase t - 1), publish the maxima of tile t + 2, ONE barrier.
                load_tile(I0{}, 0, 0);
                load_tile(I1{}, 2, -1);
                publish_max(I0{}, 0);
                publish_max(I1{}, 2);
                __syncthreads();
                take_scales(0, 0, 0);
                split_tile(I0{});
                write_x(0);
                write_z(0);
                take_scales(2, 2, -1);
                split_tile(I1{});
                write_x(2);
                load_tile(I1{}, 4, ntile > 1 ? 1 : -1);              // tile 1
                publish_max(I1{}, 1);
                __syncthreads();
                auto phase = [&](auto par_, int t) __attribute__((always_inline)) {
                    constexpr int PAR = decltype(par_)::value;       // = t & 1: tile t + 2 goes to set PAR, tile t + 1 sits in set PAR ^ 1
                    using SN = std::integral_constant<int, PAR>;
                    using SC = std::integral_constant<int, PAR ^ 1>;
                    load_tile(SN{}, 2 * t + 6, t + 2 < ntile ? t + 2 : -1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_tile(t);
                    asm volatile("" ::"v"(vk[PAR]));
                    if (t + 1 < ntile) {                              // wave-uniform
                        take_scales((t + 1) & 1, 2 * t + 4, (t + 1) & 1);
                        split_tile(SC{});
                        write_x(2 * t + 4);
                        write_z((t + 1) & 1);
                    }
                    publish_max(SN{}, t & 1);
                    __syncthreads();                                  // tile t is read, tile t + 1 is written, the maxima of tile t + 2 are published
                };
                for (int t = 0; t < ntile; t += 2) {
                    phase(I0{}, t);
                    if (t + 1 < ntile) phase(I1{}, t + 1);
                }
            }
        }
    };
    if constexpr (NP == 2) {
        if (s_role != 2) { if (lay & VXM_S3_IN0_BLOCKED) run_tasks(std::true_type{}, std::true_type{}); else run_tasks(std::true_type{}, std::false_type{}); }
        else             { if (lay & VXM_S3_IN1_BLOCKED) run_tasks(std::false_type{}, std::true_type{}); else run_tasks(std::false_type{}, std::false_type{}); }
    } else {
        if (s_role != 2) run_tasks(std::true_type{}, std::false_type{}); else run_tasks(std::false_type{}, std::false_type{});
    }

    // ---- partials: part[bx][q][tap 0..27][co 16 NCO][ci 16].  The four waves (row half, depth slice) that share a kd hold sums of the same
    // taps: they are combined through LDS in a fixed order first (the staging buffers are free now), so a block writes 28 x 16 x 16
    // floats instead of four times that -- the partials were 29 MB per launch on rem1, read back by the reducer.
    // D layout: lane (kg, n) holds co = 4 kg + r, ci = n
    const int Q = NCOMBO / NCO;
    float* const Ls = reinterpret_cast<float*>(smem);            // [wave 12][tap 9][co 16][ci 16] + [slice 4][co 16][ci 16] (bias sums)
    __syncthreads();
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ls[((wave * 9 + kh * 3 + kw) * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = tot[kh][kw][r];
    if (kd == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Ls[((SW_WAVES * 9 + 2 * ds + rh) * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = totb[r];
    }
    __syncthreads();
    float* const pp = part + (((size_t)bx * Q + q) * 28) * (16 * NCO) * 16 + cot * 256;
    for (int e = tid; e < 28 * 256; e += SW_THREADS) {
        const int tap = e >> 8, i = e & 255;
        float v;
        if (tap < 27) {
            const int kd_ = tap / 9, t9 = tap - 9 * kd_;         // waves (rh, ds, kd_) = rh 6 + ds 3 + kd_, summed in slice order 2 ds + rh
            const float* const l = Ls + (kd_ * 9 + t9) * 256 + i;
            v = ((l[0] + l[6 * 9 * 256]) + (l[3 * 9 * 256] + l[9 * 9 * 256]));
        } else {
            const float* const l = Ls + SW_WAVES * 9 * 256 + i;
            v = ((l[0] + l[256]) + (l[512] + l[768]));
        }
        pp[(size_t)tap * (16 * NCO) * 16 + i] = v;
    }
}

// gw[co][ci_off + ci][tap] (row length gw_cin) = sum over the (block, depth slice) partials in a fixed order; gb[co] from tap slot 27
// of chunk 0 (every chunk carries the same bias sum).  Block = 64 consecutive outputs x 16 slices, 8 loads in flight per thread,
// slices combined through LDS in a fixed tree (deterministic) -- the reducer of conv_bf16.hip with a channel sub-range destination.
// swap (k_s3_bww_pc<true>): the partials hold [dz tile q][mirrored tap][ci (16 NCO rows)][co 16] -- Q counts dz tiles, NCO x chunks.
__global__ void __launch_bounds__(1024) k_s3_reduce_partials(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb, int C,
                                                             int Cout, int gw_cin, int ci_off, int Q, int NCO, int NBLK, int swap) {
    __shared__ float sm[16][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const int CoP = 16 * NCO, per_q = 28 * CoP * 16, n = Q * per_q;
    const int e = blockIdx.x * 64 + x;
    float tot = 0.0f;
    if (e < n) {
        const int q = e / per_q, r = e - q * per_q;
        const size_t stride_blk = (size_t)Q * SW_NSL * per_q;
        const float* p = part + (size_t)q * SW_NSL * per_q + r;      // + blk * stride_blk + slice * per_q
        const int K = SW_NSL * NBLK;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = y;
        for (; k + 16 * 7 < K; k += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + 16 * u;
                s[u] += p[(size_t)(kk / SW_NSL) * stride_blk + (size_t)(kk % SW_NSL) * per_q];
            }
        }
        for (; k < K; k += 16) s[0] += p[(size_t)(k / SW_NSL) * stride_blk + (size_t)(k % SW_NSL) * per_q];
        tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
    sm[y][x] = tot;
    __syncthreads();
    if (y == 0 && e < n) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sm[u][x];
        const float sum = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
                          (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        const int q = e / per_q, r = e - q * per_q;
        const int tap = r / (CoP * 16), row = (r / 16) % CoP, col = q * 16 + (r & 15);
        const int co = swap ? col : row, ci = swap ? row : col;
        if (tap < 27) {
            if (co < Cout && ci < C) gw[((size_t)co * gw_cin + ci_off + ci) * 27 + (swap ? 26 - tap : tap)] = sum;
        } else if (gb != nullptr && (swap ? row == 0 : (q == 0 && (r & 15) == 0)) && co < Cout) {
            gb[co] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_s3_bww_pc (round 6): the same contraction with PRODUCER and CONSUMER waves and TWO 16-channel tiles of the un-haloed operand per block
// ------------------------------------------------------------------------------------------
// k_s3_bwd_weight gives every (16 x-channel chunk, 16 dz-channel tile) combo its own block: rem0's skip segment (x 16, dz 32) stages its
// planar x twice, rem1 (x 32, dz 16) its dz twice, and a staged tile feeds 648 MFMAs -- 324 executed FLOP per staged byte, the machine's
// balance against HBM, with every staging thread holding two raw sets (168 registers: no room for a second accumulator set).  Here
//   * a block stages ONE 16-channel chunk of the HALOED operand (the 6-plane ring of k_s3_bwd_weight) and TWO 16-channel tiles of the
//     PLAIN operand, and multiplies each against the haloed chunk: 1296 MFMAs per staged tile, 2 / 3 of the staging per MFMA;
//   * 12 consumer waves = (plain tile, depth slice, kd) only multiply -- all four rows of their slice, so a B fragment set (haloed row, kw)
//     read once from LDS feeds nine MFMAs instead of up to nine of a two-row wave's six; 4 producer waves only fetch, split and write.
//     The staging registers leave the multiplying waves (96 live registers: 36 totals, 32 dZ fragments, chains), and a producer wave is
//     alone with its 64 raw registers: 16 waves at 128 registers, one block per CU;
//   * a producer wave stages a whole scale unit by itself -- one haloed plane (16 channels x 6 x 34 voxels: waves 12, 13), or the two
//     depth slices of one plain tile (waves 14, 15) -- so the unit's largest magnitude is a wave reduction (DPP) and the power-of-two
//     scale needs no exchange between waves: load, maximum, split, write in one phase, and the loads of the tile after next are issued
//     before the phase's barrier (the raw registers are free as soon as the pieces are written), in flight under the next phase.
// The contraction is symmetric in its operands: gW[co][ci][tap] = sum_v dz[co][v] x[ci][v + tap - 1] = sum_u x[ci][u] dz[co][u - (tap - 1)],
// so either tensor can be the haloed one.  SWAP = false: haloed = x, plain = dz (Cdz a multiple of 32: rem0's skip segment, enc1, the
// 32 -> 32 layers); SWAP = true: haloed = dz, plain = x (C a multiple of 32, Cdz not: rem1), the partials then hold [ci][co] at the
// MIRRORED tap (26 - tap) and the bias sum comes from the interior of the haloed operand; k_s3_reduce_partials maps both.
// fp16 pieces only.  Partials: part[bx][haloed chunk][tap 0..27][plain channel 16 NPL][haloed channel 16], NPL = 16-channel tiles of plain.
constexpr int PW_CONS = 12, PW_PROD = 4, PW_THREADS = 64 * (PW_CONS + PW_PROD);
constexpr int PW_XBYTES = 2 * SW_XPIECE;                       // two pieces of the 6-plane haloed ring
constexpr int PW_ZBYTES = 2 * 2 * 2 * SW_ZPIECE;                // [buffer 2][plain tile 2][piece 2]
constexpr int PW_TAB_BYTES = 256;                              // floats: [0..5] inverse scales of the ring planes, [6 + 4 buf + 2 pt + ds] of the plain units
constexpr int PW_EPI_BYTES = (PW_CONS * 9 + 4) * 256 * 4;
constexpr int PW_LDS_BYTES = PW_XBYTES + PW_ZBYTES + PW_TAB_BYTES > PW_EPI_BYTES ? PW_XBYTES + PW_ZBYTES + PW_TAB_BYTES : PW_EPI_BYTES;
static_assert(PW_LDS_BYTES <= 160 * 1024, "k_s3_bww_pc: LDS");

// NPT = 1 (a 16 x 16 layer: rem2): ONE plain tile, consumer wave = (row half, depth slice, kd) with two rows each as in k_s3_bwd_weight;
// producer wave 14 stages the plain tile's depth slice 0, wave 15 its slice 1.
template <bool SWAP, int NPT = 2>
__global__ void __launch_bounds__(PW_THREADS) k_s3_bww_pc(const float* __restrict__ hal, long long hal_bs, int Chal, const float* __restrict__ pla,
                                                          long long pla_bs, int Cpla, float* __restrict__ part, int D, int H, int W, int NBLK,
                                                          int NPL2, SwTasks tk, int task_rr, int hal_blocked, int pla_blocked, int dbg) {
    using P = S3P<2>;
    static_assert(NPT == 2 || !SWAP, "one plain tile: haloed = x only");
    constexpr int RW = NPT == 2 ? SW_TH : 2, NHL = RW + 2;      // output rows / haloed rows of a consumer wave
    VXM_DYN_SMEM(char, smem);
    char* const Xs = smem;                                       // [2 pieces][6 ring planes][SW_PLANE]
    char* const Zs = smem + PW_XBYTES;                           // [2 buffers][2 plain tiles][2 pieces][SW_ZPIECE]
    float* const Tab = reinterpret_cast<float*>(smem + PW_XBYTES + PW_ZBYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NCOMBO = gridDim.x / NBLK, xmain = NBLK & ~7;
    int bx, combo;                                               // block -> (task range, combo) as k_s3_bwd_weight: the combos of one bx share an XCD
    if ((int)blockIdx.x < xmain * NCOMBO) {
        const int j = blockIdx.x >> 3;
        combo = j % NCOMBO;
        bx = (j / NCOMBO) * 8 + (blockIdx.x & 7);
    } else {
        const int r = blockIdx.x - xmain * NCOMBO;
        bx = xmain + r / NCOMBO;
        combo = r % NCOMBO;
    }
    const int qh = combo / NPL2, pp2 = combo - qh * NPL2;        // 16-channel chunk of the haloed operand, 32-channel pair (NPT = 1: 16-channel tile) of the plain one
    const int ntask = tk.ncol * tk.nseg;
    int k_lo, k_hi, k_step;
    if ((NBLK & 7) == 0 && task_rr) {
        const int x = bx & 7;
        k_lo = (int)((long long)ntask * x / 8) + (bx >> 3); k_hi = (int)((long long)ntask * (x + 1) / 8); k_step = NBLK >> 3;
    } else {
        k_lo = (int)((long long)ntask * bx / NBLK); k_hi = (int)((long long)ntask * (bx + 1) / NBLK); k_step = 1;
    }
    const int V = D * H * W, HW = H * W;
    auto task_geom = [&](int task, int& b, int& h0, int& w0, int& dbase, int& ntile) __attribute__((always_inline)) {
        const int seg = task_rr ? task / tk.ncol : task % tk.nseg, col = task_rr ? task - seg * tk.ncol : task / tk.nseg;
        const int tw = col % tk.nw; int cq = col / tk.nw;
        const int th = cq % tk.nh; b = cq / tk.nh;
        const int td0 = seg * tk.seg_len;
        ntile = min(tk.seg_len, tk.nd - td0);
        dbase = td0 * SW_TD; h0 = th * SW_TH; w0 = tw * SW_TW;
    };

    // ---- consumers: wave cw = (plain tile pt, depth slice ds, kd); the 9 (kh, kw) taps of kd over the four rows of slice ds
    const int cw = wave, c6 = cw / 6, w6 = cw - 6 * c6, ds = w6 / 3, kd = w6 - 3 * ds;
    const int pt = NPT == 2 ? c6 : 0, rb0 = NPT == 2 ? 0 : 2 * c6;      // plain tile / first output row of this wave
    f32x4 tot[3][3], totb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) tot[kh][kw] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (wave < PW_CONS) {
        const int lp = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;     // lane pattern of the transposing read
        const u32x4 ones = {P::ONES, P::ONES, P::ONES, P::ONES};
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        for (int task = k_lo; task < k_hi; task += k_step) {
            int b, h0, w0, dbase, ntile;
            task_geom(task, b, h0, w0, dbase, ntile);
            __syncthreads();                                     // the first two plane pairs and plain tile 0 are written
            for (int t = 0; t < ntile; ++t) {
                if (S3_DBG(dbg, 2)) { __syncthreads(); continue; }            // timing experiment: no multiply phase
                const int ring = (2 * t + ds + kd) % SW_RING, zcur = t & 1;
                const char* const xp = Xs + ring * SW_PLANE;
                const float unscale_h = Tab[ring], unscale_p = Tab[6 + 4 * zcur + 2 * pt + ds];
                u32x4 az[RW][2];                                 // plain fragments (two pieces) of the wave's rows
                f32x4 accb = zero4;
                __builtin_amdgcn_s_setprio(2);
#pragma unroll
                for (int r = 0; r < RW; ++r) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int zb = ((zcur * NPT + pt) * 2 + p) * SW_ZPIECE + ((ds * SW_TH + rb0 + r) * SW_TW) * 32 + lp;
                        const u32x2 lo = s3_tr_read(Zs, zb), hi = s3_tr_read(Zs, zb + 16 * 32);
                        az[r][p] = (u32x4){lo.x, lo.y, hi.x, hi.y};
                    }
                    if (!SWAP && kd == 0) {                      // bias gradient = sum of the plain operand (dz) against ones
#pragma unroll
                        for (int p = 0; p < 2; ++p) accb = P::mfma(az[r][p], ones, accb);
                    }
                }
                // the B fragment set (haloed row, kw) of step i + 1 is requested before the MFMAs of step i (two sets in registers)
                u32x4 bq[2][2];
                auto read_b = [&](auto i_) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_)::value, kw = i / NHL, hl = i - kw * NHL;
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const int xb = p * SW_XPIECE + ((rb0 + hl) * SW_XW + kw) * 32 + lp;
                        const u32x2 lo = s3_tr_read(xp, xb), hi = s3_tr_read(xp, xb + 16 * 32);
                        bq[i & 1][p] = (u32x4){lo.x, lo.y, hi.x, hi.y};
                    }
                };
                read_b(std::integral_constant<int, 0>{});
                f32x4 acc[3];
                auto step = [&](auto i_) __attribute__((always_inline)) {
                    constexpr int i = decltype(i_)::value, kw = i / NHL, hl = i - kw * NHL;       // haloed row rb0 + hl serves output rows rb0 + hl - kh
                    if constexpr (hl == 0 && kw == 1) __builtin_amdgcn_s_setprio(1);
                    if constexpr (hl == 0 && kw == 2) __builtin_amdgcn_s_setprio(0);
                    if constexpr (i + 1 < 3 * NHL) read_b(std::integral_constant<int, i + 1>{});
#pragma unroll
                    for (int tp = 0; tp < P::NPROD; ++tp)
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh) {
                            const int rl = hl - kh;
                            if (rl >= 0 && rl < RW)
                                acc[kh] = P::mfma(az[rl][P::PA[tp]], bq[i & 1][P::PB[tp]], (rl == 0 && tp == 0) ? zero4 : acc[kh]);
                        }
                    if constexpr (SWAP && kw == 1 && hl >= 1 && hl <= SW_TH) {    // bias gradient = sum of the haloed operand (dz) over its interior
                        if (kd == 1 && pt == 0) {
#pragma unroll
                            for (int p = 0; p < 2; ++p) accb = P::mfma(ones, bq[i & 1][p], accb);
                        }
                    }
                    if constexpr (hl == NHL - 1) {
#pragma unroll
                        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                            for (int j = 0; j < 4; ++j) tot[kh][kw][j] = __builtin_fmaf(acc[kh][j] * unscale_h, unscale_p, tot[kh][kw][j]);
                    }
                };
                s3_static_for(step, std::make_integer_sequence<int, 3 * NHL>{});
                const float ub = SWAP ? unscale_h : unscale_p;
#pragma unroll
                for (int j = 0; j < 4; ++j) totb[j] = __builtin_fmaf(accb[j], ub, totb[j]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                                 // tile t is read, tile t + 1 is written
            }
        }
    } else {
        // ---- producers: wave 12 + pw.  pw 0 / 1: haloed plane p0 + pw of a plane pair (216 slots = (pair of W neighbours, 8-channel half)
        // over 6 rows x 18 pairs, four rounds of 64 lanes); pw 2 / 3: plain tile pw - 2, rounds 0, 1 = depth slice 0, rounds 2, 3 = slice 1.
        const int pw = wave - PW_CONS;
        // the producers' vector work is small and everything waits for it at the phase's barrier: above the consumers (at their priority or
        // below it the multiply phase starved it, and split + multiply took longer than one after the other: 0.53 ms against 0.40 + 0.21 - 0.12)
        __builtin_amdgcn_s_setprio(3);
        auto produce = [&](auto hr_, auto bl_) __attribute__((always_inline)) {
            constexpr bool HR = decltype(hr_)::value, BLK = decltype(bl_)::value;
            constexpr int lsh = BLK ? 5 : 2;
            constexpr int SW_XPAIRS = SW_XW / 2 + 1, NXS = SW_HR * SW_XPAIRS * 2;      // 18 pairs, 216 slots per plane
            const int ppt = NPT == 2 ? pw - 2 : 0;                   // plain role: which of the two tiles (NPT = 1: wave pw stages depth slice pw - 2)
            float ra[4][8], rb[4][8];
            int off0[4], ldst[4], vk[4];
            for (int task = k_lo; task < k_hi; task += k_step) {
                int b, h0, w0, dbase, ntile;
                task_geom(task, b, h0, w0, dbase, ntile);
                const __amdgpu_buffer_rsrc_t rd = vxm_rsrc(HR ? hal + (size_t)b * hal_bs : pla + (size_t)b * pla_bs, (unsigned)(HR ? Chal : Cpla) * (unsigned)V * 4u);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    if constexpr (HR) {
                        const int si = rr * 64 + lane, cb = si & 1, pr = si >> 1, hh = pr / SW_XPAIRS, pp = pr - hh * SW_XPAIRS;
                        const int gh = h0 - 1 + hh, gw = w0 - 2 + 2 * pp;
                        const bool live = si < NXS && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W && qh * 16 + cb * 8 < Chal;
                        off0[rr] = live ? (BLK ? ((qh * 2 + cb) * V + gh * W + gw) << 5 : ((qh * 16 + cb * 8) * V + gh * W + gw) << 2) : VXM_OOB;
                        ldst[rr] = si < NXS ? (((hh * SW_XW + 2 * pp - 1) * 32 + cb * 16) | (pp == 0 ? 1 : 0) | (pp == SW_XPAIRS - 1 ? 2 : 0)) : 3;      // 3: idle lane, neither voxel stored
                    } else {
                        const int zds = NPT == 2 ? rr >> 1 : pw - 2, si = (rr & 1) * 64 + lane, cb = si & 1, r2 = si >> 1, zh = r2 / (SW_TW / 2), zw = 2 * (r2 - zh * (SW_TW / 2));
                        const int ch = pp2 * (16 * NPT) + ppt * 16 + cb * 8;
                        const bool live = h0 + zh < H && w0 + zw < W && ch < Cpla && (NPT == 2 || rr < 2);
                        off0[rr] = live ? (BLK ? ((ch >> 3) * V + (zds * H + h0 + zh) * W + w0 + zw) << 5 : (ch * V + (zds * H + h0 + zh) * W + w0 + zw) << 2) : VXM_OOB;
                        ldst[rr] = ((zds * SW_TH + zh) * SW_TW + zw) * 32 + cb * 16;
                    }
                }
                // unit u of a task: haloed planes 2 u, 2 u + 1 (plane p = depth dbase - 1 + p) and plain tile u - 1 (u = 0: none).
                // load_round: the raw loads of round rr of unit u (past the task's last unit, or outside the volume: nothing fetched, zeros)
                auto load_round = [&](auto rr_, int u) __attribute__((always_inline)) {
                    constexpr int rr = decltype(rr_)::value;
                    int sb = 0;
                    if constexpr (HR) {
                        const int gd = dbase - 1 + 2 * u + pw;                      // wave-uniform
                        const bool in = (unsigned)gd < (unsigned)D && u <= ntile;
                        int o = in ? (gd * HW) << lsh : 0, f = in ? 0 : VXM_OOB;
                        asm volatile("" : "+v"(f));
                        vk[rr] = (off0[rr] + o) | f;
                    } else {
                        const int tz = u - 1, gz = dbase + (tz < 0 ? 0 : tz) * SW_TD;    // first depth slice of the plain tile
                        int f = (tz < 0 || tz >= ntile || gz + (NPT == 2 ? rr >> 1 : pw - 2) >= D) ? VXM_OOB : 0;
                        asm volatile("" : "+v"(f));
                        vk[rr] = off0[rr] | f;
                        sb = (gz * HW) << lsh;
                    }
                    if (S3_DBG(dbg, 1)) vk[rr] = VXM_OOB;      // timing experiment: nothing fetched
                    if constexpr (BLK) {
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const f32x4 ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vk[rr], sb + 16 * k, 0));
                            const f32x4 tb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vk[rr], sb + 32 + 16 * k, 0));
#pragma unroll
                            for (int e = 0; e < 4; ++e) { ra[rr][4 * k + e] = ta[e]; rb[rr][4 * k + e] = tb[e]; }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const f32x2 t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rd, vk[rr], sb + ((e * V) << 2), 0));
                            ra[rr][e] = t2.x; rb[rr][e] = t2.y;
                        }
                    }
                };
                // scale of rounds RLO .. RLO + NR - 1 (one scale unit): wave maximum -> power of two; its inverse goes to the table
                auto unit_scale = [&](auto rlo_, auto nr_, float* tab_slot) __attribute__((always_inline)) -> float {
                    constexpr int RLO = decltype(rlo_)::value, NR = decltype(nr_)::value;
                    const float m = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
                        for (int rr = RLO; rr < RLO + NR; ++rr)
#pragma unroll
                            for (int e = 0; e < 8; ++e) { f(ra[rr][e]); f(rb[rr][e]); }
                    });
                    float sc, inv;
                    s3_scale_of(m, sc, inv);
                    if (lane == 0) *tab_slot = inv;
                    return sc;
                };
                // split round rr with scale sc and write its pieces
                auto split_round = [&](auto rr_, float sc, char* dst) __attribute__((always_inline)) {
                    constexpr int rr = decltype(rr_)::value;
                    unsigned ka[2][4], kb[2][4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s3_split2_f16(ra[rr][2 * e], ra[rr][2 * e + 1], sc, ka[0][e], ka[1][e]);
                        s3_split2_f16(rb[rr][2 * e], rb[rr][2 * e + 1], sc, kb[0][e], kb[1][e]);
                    }
                    if constexpr (HR) {
                        char* const d = dst + (ldst[rr] & ~3);
                        if (!(ldst[rr] & 1)) {
#pragma unroll
                            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(d + p * SW_XPIECE) = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                        }
                        if (!(ldst[rr] & 2)) {
#pragma unroll
                            for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(d + p * SW_XPIECE + 32) = (u32x4){kb[p][0], kb[p][1], kb[p][2], kb[p][3]};
                        }
                    } else {
#pragma unroll
                        for (int p = 0; p < 2; ++p) {
                            *reinterpret_cast<u32x4*>(dst + p * SW_ZPIECE + ldst[rr]) = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                            *reinterpret_cast<u32x4*>(dst + p * SW_ZPIECE + ldst[rr] + 32) = (u32x4){kb[p][0], kb[p][1], kb[p][2], kb[p][3]};
                        }
                    }
                };
                using I0 = std::integral_constant<int, 0>;
                using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>;
                using I3 = std::integral_constant<int, 3>;
                using I4 = std::integral_constant<int, 4>;
                // finish unit u (in registers) and request unit un: the raw registers of a round are free as soon as it is split, so its next
                // loads go out at once -- a round is in flight for nearly a whole phase, and the block's requests are spread over the phase
                // instead of leaving in one burst behind the last write (first version: 0.80 ms where loads, multiply and split alone take
                // 0.32 / 0.40 / 0.24: they ran one after the other)
                auto advance = [&](int u, bool fin, int un) __attribute__((always_inline)) {
                    if (S3_DBG(dbg, 4)) fin = false;             // timing experiment: no split, no LDS writes
                    if constexpr (HR) {
                        const int slot = (2 * u + pw) % SW_RING;
                        char* const dst = Xs + slot * SW_PLANE;
                        float sc = 1.0f;
                        if (fin) sc = unit_scale(I0{}, I4{}, Tab + slot);
                        if (fin) split_round(I0{}, sc, dst);
                        load_round(I0{}, un);
                        if (fin) split_round(I1{}, sc, dst);
                        load_round(I1{}, un);
                        if (fin) split_round(I2{}, sc, dst);
                        load_round(I2{}, un);
                        if (fin) split_round(I3{}, sc, dst);
                        load_round(I3{}, un);
                    } else {
                        fin = fin && u >= 1;                     // unit 0 carries no plain tile
                        const int zbuf = (u - 1) & 1;
                        char* const dst = Zs + ((zbuf * NPT + ppt) * 2) * SW_ZPIECE;
                        float sc = 1.0f;
                        if (fin) sc = unit_scale(I0{}, I2{}, Tab + 6 + 4 * zbuf + 2 * ppt + (NPT == 2 ? 0 : pw - 2));
                        if (fin) split_round(I0{}, sc, dst);
                        load_round(I0{}, un);
                        if (fin) split_round(I1{}, sc, dst);
                        load_round(I1{}, un);
                        if constexpr (NPT == 2) {
                            if (fin) sc = unit_scale(I2{}, I2{}, Tab + 6 + 4 * zbuf + 2 * ppt + 1);
                            if (fin) split_round(I2{}, sc, dst);
                            load_round(I2{}, un);
                            if (fin) split_round(I3{}, sc, dst);
                            load_round(I3{}, un);
                        }
                    }
                };
                advance(0, false, 0);
                advance(0, true, 1);
                advance(1, true, 2);
                __syncthreads();
                for (int t = 0; t < ntile; ++t) {
                    advance(t + 2, t + 1 < ntile, t + 3);        // tile t + 1 = unit t + 2; past the task's last unit nothing is fetched
                    asm volatile("" ::"v"(vk[0]), "v"(vk[1]), "v"(vk[2]), "v"(vk[3]));
                    __syncthreads();
                }
            }
        };
        if (pw < 2) { if (hal_blocked) produce(std::true_type{}, std::true_type{}); else produce(std::true_type{}, std::false_type{}); }
        else        { if (pla_blocked) produce(std::false_type{}, std::true_type{}); else produce(std::false_type{}, std::false_type{}); }
    }

    // ---- partials: the two depth-slice waves of a (plain tile, kd) are summed through LDS in a fixed order
    float* const Ls = reinterpret_cast<float*>(smem);            // [wave 12][tap 9][plain 16][haloed 16] + [4 bias slots][16][16]
    __syncthreads();
    if (wave < PW_CONS) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ls[((cw * 9 + kh * 3 + kw) * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = tot[kh][kw][r];
        if (SWAP ? (kd == 1 && pt == 0) : (kd == 0)) {
            const int slot = SWAP ? ds : 2 * c6 + ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) Ls[((PW_CONS * 9 + slot) * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = totb[r];
        }
    }
    __syncthreads();
    const int NPL = NPT * NPL2, QH = NCOMBO / NPL2;
    float* const po = part + (((size_t)bx * QH + qh) * 28) * (16 * NPL) * 16 + (size_t)(NPT * pp2) * 256;
    for (int e = tid; e < NPT * 28 * 256; e += PW_THREADS) {
        const int pt_ = e / (28 * 256), r_ = e - pt_ * (28 * 256), tap = r_ >> 8, i = r_ & 255;
        float v;
        if (NPT == 1) {                                      // the four (row half, depth slice) waves of a kd, in slice order 2 ds + rh as k_s3_bwd_weight
            if (tap < 27) {
                const int kd_ = tap / 9, t9 = tap - 9 * kd_;
                const float* const l = Ls + (kd_ * 9 + t9) * 256 + i;
                v = ((l[0] + l[6 * 9 * 256]) + (l[3 * 9 * 256] + l[9 * 9 * 256]));
            } else {
                const float* const l = Ls + PW_CONS * 9 * 256 + i;     // bias slots 2 rh + ds
                v = ((l[0] + l[512]) + (l[256] + l[768]));
            }
        } else if (tap < 27) {
            const int kd_ = tap / 9, t9 = tap - 9 * kd_;
            const float* const l = Ls + ((pt_ * 6 + kd_) * 9 + t9) * 256 + i;
            v = l[0] + l[3 * 9 * 256];
        } else if (SWAP) {
            const float* const l = Ls + PW_CONS * 9 * 256 + i;
            v = pt_ == 0 ? l[0] + l[256] : 0.0f;
        } else {
            const float* const l = Ls + (PW_CONS * 9 + 2 * pt_) * 256 + i;
            v = l[0] + l[256];
        }
        po[(size_t)tap * (16 * NPL) * 16 + pt_ * 256 + i] = v;
    }
}

int sw_cus() {
    static const int cus = [] {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }();
    return cus;
}
// one persistent block per CU over all combos; columns are cut into depth segments until there are ~4 tasks per block
SwTasks sw_tasks(int ncombo, int B, int D, int H, int W, int& NBLK) {
    SwTasks tk;
    tk.nd = (D + SW_TD - 1) / SW_TD; tk.nh = (H + SW_TH - 1) / SW_TH; tk.nw = (W + SW_TW - 1) / SW_TW;
    tk.ncol = B * tk.nh * tk.nw;
    const int nb = sw_cus() / ncombo > 0 ? sw_cus() / ncombo : 1;
    int nseg = (4 * nb + tk.ncol - 1) / tk.ncol;
    if (nseg > tk.nd) nseg = tk.nd;
    if (nseg < 1) nseg = 1;
    tk.seg_len = (tk.nd + nseg - 1) / nseg;
    tk.nseg = (tk.nd + tk.seg_len - 1) / tk.seg_len;
    const long long ntask = (long long)tk.ncol * tk.nseg;
    NBLK = (int)(nb < ntask ? nb : ntask);
    return tk;
}

// which launches k_s3_bww_pc takes (fp16 pieces): 1 = haloed x, two plain dz tiles (Cout a multiple of 32), 2 = swapped (C a multiple of 32),
// 3 = haloed x, one plain dz tile, 0 = none.
// VXM_S3_BW_PC=0 keeps every launch on k_s3_bwd_weight (same-box A/B).
int sw_pc_mode(int C, int Cout, int pieces) {
    static const bool on = [] { const char* e = getenv("VXM_S3_BW_PC"); return !(e && e[0] == '0'); }();
    if (!on || pieces != 2 || C % 16 || Cout % 16) return 0;
    if (Cout % 32 == 0) return 1;
    if (C % 32 == 0) return 2;
    return 3;                                                    // one plain tile per block (16 x 16, 16 x 48 ...): the producer / consumer structure alone
}

// ---- host side -------------------------------------------------------------------------------------------------------------
// Kernel instance of an operator with OutC output channels.  VXM_S3_CB=2 stages 16-channel chunks (5 K-steps, 90 % of the K slots
// used, one block per CU) instead of 8-channel ones (3 K-steps, 75 %, two blocks per CU); VXM_S3_NCT=1 runs 32-channel operators as
// two 16-channel groups.  The packed layout depends on both, so they are read once per process.
struct S3Variant { int NCT, CB; };
S3Variant s3_variant(int OutC) {
    static const int cb = [] { const char* e = getenv("VXM_S3_CB"); return e && e[0] == '2' ? 2 : 1; }();
    static const int nct_max = [] { const char* e = getenv("VXM_S3_NCT"); return e && e[0] == '1' ? 1 : 2; }();
    S3Variant v;
    v.CB = cb;
    v.NCT = (OutC > 16 && nct_max == 2 && cb == 1) ? 2 : 1;      // (NCT 2, CB 2) does not fit the LDS
    return v;
}
long long s3_min_tiles() {
    // 128 tiles of 8 x 4 x 16: the two finest levels and the 40 x 48 x 56 level of the headline shape (240 tiles; measured in the step with
    // the level on the split kernels: 13.05 -> 12.81 ms, same box -- its backward-weight launches gain most); 20 x 24 x 28 (36 tiles) stays
    static const long long v = [] { const char* e = getenv("VXM_S3_MIN_TILES"); return e ? atoll(e) : 128ll; }();
    return v;
}
int s3_chunks(int C, int CB) { return (C + 8 * CB - 1) / (8 * CB); }
// 16-byte words of the packed operator proper (NP = 2: one trailer word {1 / scale, scale} follows them)
size_t s3_packed_words(int seg0, int seg1, int OutC, int NP) {
    const S3Variant v = s3_variant(OutC);
    const int Q = s3_chunks(seg0, v.CB) + s3_chunks(seg1, v.CB), G = (OutC + 16 * v.NCT - 1) / (16 * v.NCT);
    const int NS = v.CB == 2 ? 5 : 2, RT = v.CB == 2 ? 0 : 1;
    return (size_t)G * Q * (NS * 3 + RT) * NP * v.NCT * 64;
}
bool s3_pieces_ok(int np) { return np == 2 || np == 3; }

// Developer experiments (timing only, results wrong): compiled in only with -DVXM_S3_EXP (tools/build_exp.sh), selected per launch by VXM_S3_DBG
int s3_dbg() {
#ifdef VXM_S3_EXP
    const char* e = getenv("VXM_S3_DBG");
    return e ? atoi(e) : 0;
#else
    return 0;
#endif
}

template <int NCT, int ROWS, int CB, int NP, bool RUN = false, bool BLK = false>
void s3_launch(const ConvIn& in, const void* wp, const float* bias, float* y, long long y_bs, int Cout, float slope, const float* mask,
               long long mask_bs, float mask_slope, int B, int D, int H, int W, hipStream_t s, int lay = 0) {
    using C = S3Cfg<NCT, ROWS, CB, NP>;
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_conv<NCT, ROWS, CB, NP, RUN, BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (getenv("VXM_S3_DEBUG")) {             // developer switch: what the runtime says about co-residency
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_s3_conv<NCT, ROWS, CB, NP, RUN, BLK>, S3_THREADS, C::LDS_BYTES);
            fprintf(stderr, "k_s3_conv<%d,%d,%d,%d>: %d bytes of LDS, %d block(s) per CU\n", NCT, ROWS, CB, NP, C::LDS_BYTES, nb);
        }
        return true;
    }();
    (void)attr;
    const int Q0 = s3_chunks(in.C0, CB), Q = Q0 + s3_chunks(in.C1, CB);
    const long long ntiles = (long long)B * ((D + S3_TD - 1) / S3_TD) * ((H + ROWS - 1) / ROWS) * ((W + 15) / 16);       // (of THIS instance's tiles)
    unsigned gx = ntiles >= 64 ? (unsigned)(8 * ((ntiles + 7) / 8)) : (unsigned)ntiles;
    const int G = (Cout + 16 * NCT - 1) / (16 * NCT);
    // A grid of at most n blocks per CU whose blocks walk their XCD's tile range (VXM_S3_PERSIST=n: n per CU, 0: one tile per block).
    // Measured on the full-resolution layers: 16 -> 16 operators (two chunks per tile, i.e. short blocks) 0.80 -> 0.74 and 0.745 -> 0.69 ms,
    // the up-sampling gather 2.64 -> 2.48 ms, the 32 <-> 16 operators unchanged (2, 4 and 8 blocks per CU within 1 % of each other).
    // What a tile still pays is its first chunk -- load, split, LDS, barrier before the first MFMA: with that stage skipped altogether
    // (wrong results, timing only) the same operators ran 7 - 21 % faster, which bounds what a cross-tile prefetch could gain.
    // The one-block-per-CU instances (32-channel operators) run best with exactly one block per CU -- every block then walks several
    // tiles even at half resolution and stages ahead across them (same-box A/B, 20 iterations x 2: enc1 forward 0.143 -> 0.130 ms, rem1
    // backward-data 1.22 -> 1.175, dec3 0.448 -> 0.439) --, the two-blocks-per-CU instances with 8 (1 per CU: +12 %, 2 .. 8 within 2 %).
    static const int persist = [] { const char* e = getenv("VXM_S3_PERSIST"); return e ? atoi(e) : (C::MIN_WAVES == 2 ? 1 : 8); }();      // < 0: that many blocks in all (tests)
    if (persist != 0 && ntiles >= 64) {
        const unsigned want = persist > 0 ? (unsigned)(sw_cus() * persist / G) : (unsigned)(-persist);
        const unsigned cap = 8 * ((want + 7) / 8);
        if (cap < gx) gx = cap;
    }
    hipLaunchKernelGGL((k_s3_conv<NCT, ROWS, CB, NP, RUN, BLK>), dim3(gx, G), dim3(S3_THREADS), C::LDS_BYTES, s, in, static_cast<const u32x4*>(wp), bias, y, y_bs,
                       Cout, slope, mask, mask_bs, mask_slope, B, D, H, W, Q0, Q, lay, s3_dbg());
}

// VXM_S3_ROWS8_MIN_BLOCKS: smallest grid (8 x 8 x 16 tiles x 32-channel output groups) for which a 32-channel operator keeps the 8-row instance
// (default 200: measured at 120 blocks -- 4 rows ahead, 33 against 46 us -- and at 240 -- 8 rows ahead, 47 against 53).
// The choice depends on the SHAPE only, never on the layout flags: a channel-blocked launch and the planar launch of the same shape run the
// same instance (tile geometry decides the per-tile scales, i.e. the bits), which is what the layout tests assert.
bool s3_rows8_fills_chip(int Cout, int B, int D, int H, int W) {
    static const long long rows8_min = [] { const char* e = getenv("VXM_S3_ROWS8_MIN_BLOCKS"); return e ? atoll(e) : 200ll; }();
    const long long nb8 = (long long)B * ((D + S3_TD - 1) / S3_TD) * ((H + 7) / 8) * ((W + 15) / 16) * ((Cout + 31) / 32);
    return nb8 >= rows8_min;
}

bool s3_use_pc(int B, int D, int H, int W, bool has_mask) {
    static const int pc = [] { const char* e = getenv("VXM_S3_PC"); return e ? atoi(e) : -1; }();
    const long long nt8 = (long long)B * ((D + 7) / 8) * ((H + 7) / 8) * ((W + 15) / 16);
    return pc == 1 || (pc < 0 && nt8 >= 2048 && !has_mask);
}

// k_s3p_conv: one block of 16 waves per CU, every block walks its XCD's tile range (VXM_S3P_BLOCKS=n: n blocks in all, tests / A/B)
template <int NCT, bool BLK = false>
void s3p_launch(const ConvIn& in, const void* wp, const float* bias, float* y, long long y_bs, int Cout, float slope, const float* mask,
                long long mask_bs, float mask_slope, int B, int D, int H, int W, hipStream_t s, int lay = 0) {
    using PC = S3PCfg<NCT>;
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3p_conv<NCT, BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, PC::LDS_BYTES);
        return true;
    }();
    (void)attr;
    const int Q0 = s3_chunks(in.C0, 1), Q = Q0 + s3_chunks(in.C1, 1);
    const long long ntiles = (long long)B * ((D + S3_TD - 1) / S3_TD) * ((H + 7) / 8) * ((W + 15) / 16);
    unsigned gx = ntiles >= 64 ? (unsigned)(8 * ((ntiles + 7) / 8)) : (unsigned)ntiles;
    const int G = (Cout + 16 * NCT - 1) / (16 * NCT);
    static const int nblk = [] { const char* e = getenv("VXM_S3P_BLOCKS"); return e ? atoi(e) : 0; }();
    if (ntiles >= 64) {
        const unsigned want = nblk > 0 ? (unsigned)nblk : (unsigned)(sw_cus() / G);
        const unsigned cap = 8 * ((want + 7) / 8);
        if (cap < gx) gx = cap;
    }
    hipLaunchKernelGGL((k_s3p_conv<NCT, BLK>), dim3(gx, G), dim3(S3P_THREADS), PC::LDS_BYTES, s, in, static_cast<const u32x4*>(wp), bias, y, y_bs,
                       Cout, slope, mask, mask_bs, mask_slope, B, D, H, W, Q0, Q, lay, s3_dbg());
}

}  // namespace

extern "C" {

int vxm_conv3d_k3_s3_ok(int C0, int C1, int Cout, int B, int D, int H, int W) {
    if (C0 <= 0 || C1 < 0 || Cout < 8 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (C0 % 8 || C1 % 8) return 0;
    if (s3_variant(Cout).CB == 2 && C1 > 0 && C0 % 16) return 0;          // a chunk lies inside one segment
    if ((long long)(C0 + C1 > Cout ? C0 + C1 : Cout) * D * H * W >= (1ll << 29)) return 0;
    const long long ntiles = (long long)B * ((D + S3_TD - 1) / S3_TD) * ((H + 3) / 4) * ((W + 15) / 16);
    return ntiles >= s3_min_tiles() ? 1 : 0;
}

int vxm_conv3d_k3_s3_variant(int Cout) {
    const S3Variant v = s3_variant(Cout);
    return 10 * v.NCT + v.CB;
}

/* 1 when vxm_conv3d_k3_s3_fwd accepts the layout flags for a launch of this shape: the 8-row instances of the fp16 scheme (the tensors of the
 * plain full-resolution layers), one segment, no upsampling gather, channel counts in multiples of 8 */
int vxm_conv3d_k3_s3_layout_ok(int C0, int C1, int x0_up, int Cout, int H, int pieces) {
    static const bool rows8 = [] { const char* e = getenv("VXM_S3_ROWS"); return !(e && e[0] == '4'); }();
    return (pieces == 2 && rows8 && H >= 8 && s3_variant(Cout).CB == 1 && C1 == 0 && !x0_up && C0 > 0 && C0 % 8 == 0 && Cout % 8 == 0) ? 1 : 0;
}

/* 1 when vxm_conv3d_k3_s3_fwd will run the producer / consumer kernel k_s3p_conv for this launch (profiling labels) */
int vxm_conv3d_k3_s3_producer_consumer(int Cout, int pieces, int has_mask, int B, int D, int H, int W) {
    static const bool rows8 = [] { const char* e = getenv("VXM_S3_ROWS"); return !(e && e[0] == '4'); }();
    const S3Variant v = s3_variant(Cout);
    return (pieces == 2 && rows8 && H >= 8 && v.CB == 1 && v.NCT == 1 && s3_use_pc(B, D, H, W, has_mask != 0)) ? 1 : 0;
}

/* rows of the output tile the launch of vxm_conv3d_k3_s3_fwd will use (profiling labels): 8 on the fp16 scheme with 8-channel chunks, else 4 */
int vxm_conv3d_k3_s3_tile_rows(int Cout, int pieces, int H) {
    static const bool rows8 = [] { const char* e = getenv("VXM_S3_ROWS"); return !(e && e[0] == '4'); }();
    return (pieces == 2 && rows8 && H >= 8 && s3_variant(Cout).CB == 1) ? 8 : 4;
}
/* the same for a launch of this shape on planar tensors: a 32-channel operator whose 8 x 8 x 16 tiles would leave CUs without a block takes 4 rows */
int vxm_conv3d_k3_s3_tile_rows_at(int Cout, int pieces, int B, int D, int H, int W) {
    const int r = vxm_conv3d_k3_s3_tile_rows(Cout, pieces, H);
    return (r == 8 && s3_variant(Cout).NCT == 2 && !s3_rows8_fills_chip(Cout, B, D, H, W)) ? 4 : r;
}

size_t vxm_conv3d_k3_s3_packed_bytes(int seg0, int seg1, int OutC, int pieces) {
    if (seg0 <= 0 || seg1 < 0 || OutC <= 0 || !s3_pieces_ok(pieces)) return 0;
    return (s3_packed_words(seg0, seg1, OutC, pieces) + (pieces == 2 ? 1 : 0)) * 16;
}

int vxm_conv3d_k3_s3_pack_weights_batch(const VxmS3PackJob* jobs, int n_jobs, void* stream) {
    VXM_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3_pack_weights_batch: null job table");
    for (int j = 0; j < n_jobs; ++j) {
        const VxmS3PackJob& a = jobs[j];
        VXM_REQUIRE(a.w && a.wpacked, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3_pack_weights_batch: job %d: null pointer", j);
        VXM_REQUIRE(a.Cw_in > 0 && a.Cw_out > 0 && a.ci_lo >= 0 && a.ci_n > 0 && a.ci_lo + a.ci_n <= a.Cw_in, VXM_ERR_BAD_SHAPE,
                    "vxm_conv3d_k3_s3_pack_weights_batch: job %d: channel range [%d, %d) of %d", j, a.ci_lo, a.ci_lo + a.ci_n, a.Cw_in);
        const int InC = a.transpose_flip ? a.Cw_out : a.ci_n;
        VXM_REQUIRE(a.seg0 > 0 && a.seg0 <= InC && (reinterpret_cast<uintptr_t>(a.wpacked) & 15) == 0, VXM_ERR_BAD_SHAPE,
                    "vxm_conv3d_k3_s3_pack_weights_batch: job %d: segment split %d of %d input channels / alignment", j, a.seg0, InC);
        VXM_REQUIRE(s3_pieces_ok(a.pieces), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3_pack_weights_batch: job %d: pieces = %d (3: bf16, 2: fp16)", j, a.pieces);
    }
    for (int j0 = 0; j0 < n_jobs; j0 += S3_PACK_JOBS) {
        S3PackBatch batch;
        batch.n = n_jobs - j0 < S3_PACK_JOBS ? n_jobs - j0 : S3_PACK_JOBS;
        unsigned blocks = 0;
        bool any2 = false;
        for (int j = 0; j < batch.n; ++j) {
            const VxmS3PackJob& a = jobs[j0 + j];
            const int InC = a.transpose_flip ? a.Cw_out : a.ci_n, OutC = a.transpose_flip ? a.ci_n : a.Cw_out;
            const S3Variant v = s3_variant(OutC);
            const int Q0 = s3_chunks(a.seg0, v.CB), Q = Q0 + s3_chunks(InC - a.seg0, v.CB);
            const size_t words = s3_packed_words(a.seg0, InC - a.seg0, OutC, a.pieces);
            batch.job[j] = {a.w, static_cast<u32x4*>(a.wpacked), a.Cw_in, a.Cw_out, a.ci_lo, a.ci_n, a.transpose_flip ? 1 : 0, InC, a.seg0, OutC,
                            v.NCT, v.CB, Q0, Q, a.pieces, blocks, (unsigned)words};
            blocks += (unsigned)((words + 255) / 256);
            any2 = any2 || a.pieces == 2;
        }
        if (any2) hipLaunchKernelGGL(k_s3_wmax, dim3(batch.n), dim3(1024), 0, VXM_STREAM(stream), batch);      // the operators' scales, read by the pack below
        hipLaunchKernelGGL(k_s3_pack_weights, dim3(blocks), dim3(256), 0, VXM_STREAM(stream), batch);
    }
    return vxm_check_launch("vxm_conv3d_k3_s3_pack_weights_batch");
}

int vxm_conv3d_k3_s3_fwd(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride, const void* wpacked,
                         const float* bias, float* y, int64_t y_bstride, int Cout, float leaky_slope, const float* mask, int64_t mask_bstride,
                         float mask_slope, int B, int D, int H, int W, int pieces_and_layout, void* stream) {
    const int pieces = pieces_and_layout & 0xff, lay = pieces_and_layout & ~0xff;
    VXM_REQUIRE(x0 && wpacked && y && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3_fwd: null pointer");
    VXM_REQUIRE(s3_pieces_ok(pieces), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3_fwd: pieces = %d (3: bf16, 2: fp16)", pieces);
    const int sg = lay & (VXM_S3_MASK_SIGNS | VXM_S3_OUT_SIGNS);         // `mask` is a sign tensor (read by a backward-data epilogue / written by a forward one)
    VXM_REQUIRE(sg == 0 || (sg != (VXM_S3_MASK_SIGNS | VXM_S3_OUT_SIGNS) && mask && (lay & VXM_S3_OUT_BLOCKED) && pieces == 2 && Cout % 8 == 0), VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3_fwd: sign-tensor flags 0x%x need one of the two, a sign tensor in `mask_src`, a channel-blocked output and the fp16 pieces", sg);
    const int lay_ = lay & ~VXM_S3_REVERSE_TILES & ~sg;                  // (scheduling hint / mask format, not layouts: taken off before the layout checks)
    VXM_REQUIRE(lay_ == 0 || ((lay_ & ~(VXM_S3_IN0_BLOCKED | VXM_S3_OUT_BLOCKED)) == 0 && vxm_conv3d_k3_s3_layout_ok(C0, C1, x0_up, Cout, H, pieces)), VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3_fwd: layout flags 0x%x are not available for this launch (%d + %d -> %d channels, upsampled %d, H = %d, pieces %d)", lay, C0, C1,
                Cout, x0_up, H, pieces);
    if (int e = check_conv("vxm_conv3d_k3_s3_fwd", C0, C1, x0_up, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(C0 % 8 == 0 && C1 % 8 == 0, VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3_fwd: segments carry multiples of 8 channels, got %d + %d", C0, C1);
    const S3Variant v = s3_variant(Cout);
    VXM_REQUIRE(!(v.CB == 2 && C1 > 0 && C0 % 16), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3_fwd: 16-channel chunks need C0 %% 16 == 0 beside a second segment");
    VXM_REQUIRE((reinterpret_cast<uintptr_t>(wpacked) & 15) == 0, VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3_fwd: packed weights must be 16-byte aligned");
    const ConvIn in = {x0, x1, (long long)x0_bstride, (long long)x1_bstride, C0, C1, x0_up ? 1 : 0};
    hipStream_t s = VXM_STREAM(stream);
    // (32-channel operators with two blocks per CU were measured twice and not kept: on 8 x 2 x 16 tiles -- 81 KB of LDS -- 32 spilled registers
    // at the 128-VGPR limit of four waves per SIMD, 2.56 instead of 2.02 ms per step on these launches; as two passes of the 16-channel
    // instance over one staged chunk -- weights of one pass in LDS at a time -- 41 spilled registers, rem1 backward-data 1.49 instead of 1.39 ms)
#define S3_GO(NCT, CB)                                                                                                                    \
    do {                                                                                                                                 \
        if (pieces == 2) s3_launch<NCT, 4, CB, 2>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s); \
        else s3_launch<NCT, 4, CB, 3>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s);            \
    } while (0)
    // 16-channel operators on the fp16 scheme: 8 x 8 x 16 tiles (halo 1.76 instead of 2.11 staged voxels per output voxel) with the running
    // scale; VXM_S3_ROWS=4 keeps the 8 x 4 x 16 tiles (A/B)
    static const bool rows8 = [] { const char* e = getenv("VXM_S3_ROWS"); return !(e && e[0] == '4'); }();
    const bool blk_in = (lay & VXM_S3_IN0_BLOCKED) != 0;
    // 32-channel operators whose 8 x 8 x 16 tiles x output-channel groups do not give every CU a block (the 40x48x56 level: 120 tiles for 256
    // CUs) take the 8 x 4 x 16 instance of the same kernel: enc2 forward 46 -> 33 us, its backward-data 44 -> 31, dec2 forward 76 -> 52 (round 6,
    // rocprof); with two output-channel groups (240 blocks) the 8-row instance stays ahead (47 against 54 us).
    if (v.NCT == 2 && pieces == 2 && rows8 && H >= 8 && v.CB == 1) {
        if (!s3_rows8_fills_chip(Cout, B, D, H, W)) {
            if (blk_in) s3_launch<2, 4, 1, 2, true, true>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s, lay);
            else s3_launch<2, 4, 1, 2, true>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s, lay);
        }
        else if (blk_in) s3_launch<2, 8, 1, 2, true, true>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s, lay);
        else s3_launch<2, 8, 1, 2, true>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s, lay);
    }
    else if (v.NCT == 2) S3_GO(2, 1);
    else if (v.CB == 2) S3_GO(1, 2);
    else if (pieces == 2 && rows8 && H >= 8) {
        // producer / consumer waves (k_s3p_conv) for forward launches from 2048 tiles up (same-box A/B at 160x192x224: 16 -> 16 0.498 -> 0.472 ms,
        // 32 -> 16 0.848 -> 0.809; backward-data launches, whose epilogue waits for the mask it reads, 0.493 -> 0.500 and 0.813 -> 0.836: not routed;
        // requesting the mask under the MFMAs of the tile's last chunk cost the forward launches their gain and did not help these).
        // VXM_S3_PC=0: the alternating kernel everywhere, =1: k_s3p_conv on every eligible launch (tests)
        if (s3_use_pc(B, D, H, W, mask != nullptr && sg == 0)) {
            if (blk_in) s3p_launch<1, true>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s, lay);
            else s3p_launch<1>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s, lay);
        } else {
            if (blk_in) s3_launch<1, 8, 1, 2, true, true>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s, lay);
            else s3_launch<1, 8, 1, 2, true>(in, wpacked, bias, y, y_bstride, Cout, leaky_slope, mask, mask_bstride, mask_slope, B, D, H, W, s, lay);
        }
    }
    else S3_GO(1, 1);
#undef S3_GO
    return vxm_check_launch("vxm_conv3d_k3_s3_fwd");
}

int vxm_conv3d_k3_s3_bwd_weight_ok(int C, int Cout, int B, int D, int H, int W) {
    if (C <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (C % 16 || Cout % 16 || Cout < 16 || W % 2) return 0;      // W even: the staging loads voxel pairs
    if ((C / 16) * (Cout / 16) > sw_cus()) return 0;
    if ((long long)(C > Cout ? C : Cout) * D * H * W >= (1ll << 29)) return 0;
    const long long ntiles = (long long)B * ((D + S3_TD - 1) / S3_TD) * ((H + 3) / 4) * ((W + 15) / 16);
    return ntiles >= s3_min_tiles() ? 1 : 0;
}

/* which kernel vxm_conv3d_k3_s3_bwd_weight launches for this shape (for profiles and bench regions): 0 = k_s3_bwd_weight<pieces>,
 * 1 = k_s3_bww_pc<false, 2>, 2 = k_s3_bww_pc<true, 2>, 3 = k_s3_bww_pc<false, 1> */
int vxm_conv3d_k3_s3_bwd_weight_kernel(int C, int Cout, int pieces) { return sw_pc_mode(C, Cout, pieces & 0xff); }

size_t vxm_conv3d_k3_s3_bwd_weight_workspace_bytes(int C, int Cout, int B, int D, int H, int W) {
    if (C <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const int Q = (C + 15) / 16, NCO = (Cout + 15) / 16;
    int NBLK = 1;
    (void)sw_tasks(Q * NCO, B, D, H, W, NBLK);
    size_t need = (size_t)NBLK * Q * SW_NSL * 28 * (16 * NCO) * 16 * sizeof(float);
    const int mode = sw_pc_mode(C, Cout, 2);                     // k_s3_bww_pc runs fewer combos, hence more blocks per combo: the larger of the two
    if (mode) {
        (void)sw_tasks(mode == 3 ? Q * NCO : Q * NCO / 2, B, D, H, W, NBLK);
        const size_t need2 = (size_t)NBLK * Q * 28 * (16 * NCO) * 16 * sizeof(float);
        if (need2 > need) need = need2;
    }
    return need;
}

int vxm_conv3d_k3_s3_bwd_weight(const float* x, int C, int64_t x_bstride, const float* dz, int64_t dz_bstride, int Cout, float* gw, int gw_cin,
                                int ci_off, float* gb, void* work, size_t work_bytes, int B, int D, int H, int W, int pieces_and_layout, void* stream) {
    const int pieces = pieces_and_layout & 0xff, phase = pieces_and_layout & (VXM_S3_BW_CONTRACT_ONLY | VXM_S3_BW_REDUCE_ONLY);
    const int lay = pieces_and_layout & ~0xff & ~(VXM_S3_BW_CONTRACT_ONLY | VXM_S3_BW_REDUCE_ONLY);
    VXM_REQUIRE(phase != (VXM_S3_BW_CONTRACT_ONLY | VXM_S3_BW_REDUCE_ONLY), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3_bwd_weight: both phase flags set");
    VXM_REQUIRE(x && dz && gw && work, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3_bwd_weight: null pointer");
    VXM_REQUIRE(s3_pieces_ok(pieces), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3_bwd_weight: pieces = %d (3: bf16, 2: fp16)", pieces);
    VXM_REQUIRE(lay == 0 || (pieces == 2 && (lay & ~(VXM_S3_IN0_BLOCKED | VXM_S3_IN1_BLOCKED)) == 0), VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3_bwd_weight: layout flags 0x%x (x: IN0, dz: IN1; fp16 scheme only)", lay);
    if (int e = check_conv("vxm_conv3d_k3_s3_bwd_weight", C, 0, 0, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(C % 16 == 0 && Cout % 16 == 0 && ci_off >= 0 && ci_off + C <= gw_cin && W % 2 == 0, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3_bwd_weight: %d input / %d output channels (multiples of 16), destination channels [%d, %d) of %d, W = %d (even)", C,
                Cout, ci_off, ci_off + C, gw_cin, W);
    const int Q = C / 16, NCO = Cout / 16;
    VXM_REQUIRE(Q * NCO <= sw_cus(), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3_bwd_weight: %d x %d channel tiles exceed the compute units", Q, NCO);
    const int pc = sw_pc_mode(C, Cout, pieces);                  // 1 / 2: k_s3_bww_pc (two plain tiles per staged haloed chunk), haloed = x / dz
    int NBLK = 1;
    const SwTasks tk = sw_tasks(pc == 1 || pc == 2 ? Q * NCO / 2 : Q * NCO, B, D, H, W, NBLK);
    VXM_REQUIRE(work_bytes >= (size_t)NBLK * Q * SW_NSL * 28 * (16 * NCO) * 16 * sizeof(float), VXM_ERR_WORKSPACE,
                "vxm_conv3d_k3_s3_bwd_weight: workspace too small");
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_bwd_weight<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SwCfg<3>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_bwd_weight<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SwCfg<2>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_bwd_weight<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SwCfg<2>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_bww_pc<false>), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_bww_pc<true>), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3_bww_pc<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, PW_LDS_BYTES);
        return true;
    }();
    (void)attr;
    hipStream_t s = VXM_STREAM(stream);
    float* part = static_cast<float*>(work);
    const char* pe = getenv("VXM_S3_BW_PIPE");                  // developer A/B switch: 1 = the two-barrier pipeline of the first fp16 version
    const char* te = getenv("VXM_S3_BW_TASKS");                 // developer A/B switch: range = a contiguous task range per block (rounds 3 / early 4)
    // (round-robin order: -8 .. -16 % at 160x192x224, +3 % at 80x96x112 -- same-box A/B, profiles/r04r_bw_task_order.txt)
    const int task_rr = ((te && te[0] == 'r' && te[1] == 'a') || (long long)D * H * W < (1ll << 21)) ? 0 : 1;
    if (phase == VXM_S3_BW_REDUCE_ONLY) {}
    else if (pc == 1)
        hipLaunchKernelGGL((k_s3_bww_pc<false>), dim3(NBLK * Q * NCO / 2), dim3(PW_THREADS), PW_LDS_BYTES, s, x, (long long)x_bstride, C, dz, (long long)dz_bstride,
                           Cout, part, D, H, W, NBLK, NCO / 2, tk, task_rr, lay & VXM_S3_IN0_BLOCKED ? 1 : 0, lay & VXM_S3_IN1_BLOCKED ? 1 : 0, s3_dbg());
    else if (pc == 2)
        hipLaunchKernelGGL((k_s3_bww_pc<true>), dim3(NBLK * Q * NCO / 2), dim3(PW_THREADS), PW_LDS_BYTES, s, dz, (long long)dz_bstride, Cout, x, (long long)x_bstride,
                           C, part, D, H, W, NBLK, Q / 2, tk, task_rr, lay & VXM_S3_IN1_BLOCKED ? 1 : 0, lay & VXM_S3_IN0_BLOCKED ? 1 : 0, s3_dbg());
    else if (pc == 3)
        hipLaunchKernelGGL((k_s3_bww_pc<false, 1>), dim3(NBLK * Q * NCO), dim3(PW_THREADS), PW_LDS_BYTES, s, x, (long long)x_bstride, C, dz, (long long)dz_bstride,
                           Cout, part, D, H, W, NBLK, NCO, tk, task_rr, lay & VXM_S3_IN0_BLOCKED ? 1 : 0, lay & VXM_S3_IN1_BLOCKED ? 1 : 0, s3_dbg());
    else if (pieces == 2 && pe && pe[0] == '1')
        hipLaunchKernelGGL((k_s3_bwd_weight<2, false>), dim3(NBLK * Q * NCO), dim3(SW_THREADS), SwCfg<2>::LDS_BYTES, s, x, (long long)x_bstride, C, dz,
                           (long long)dz_bstride, Cout, part, D, H, W, NBLK, NCO, tk, task_rr, lay, s3_dbg());
    else if (pieces == 2)
        hipLaunchKernelGGL((k_s3_bwd_weight<2, true>), dim3(NBLK * Q * NCO), dim3(SW_THREADS), SwCfg<2>::LDS_BYTES, s, x, (long long)x_bstride, C, dz,
                           (long long)dz_bstride, Cout, part, D, H, W, NBLK, NCO, tk, task_rr, lay, s3_dbg());
    else
        hipLaunchKernelGGL((k_s3_bwd_weight<3, true>), dim3(NBLK * Q * NCO), dim3(SW_THREADS), SwCfg<3>::LDS_BYTES, s, x, (long long)x_bstride, C, dz,
                           (long long)dz_bstride, Cout, part, D, H, W, NBLK, NCO, tk, task_rr, lay, s3_dbg());
    const int n = 16 * NCO * 16 * Q * 28;
    if (phase != VXM_S3_BW_CONTRACT_ONLY) {
        if (pc == 2) hipLaunchKernelGGL(k_s3_reduce_partials, dim3(vxm_blocks(n, 64)), dim3(1024), 0, s, part, gw, gb, C, Cout, gw_cin, ci_off, NCO, Q, NBLK, 1);
        else hipLaunchKernelGGL(k_s3_reduce_partials, dim3(vxm_blocks(n, 64)), dim3(1024), 0, s, part, gw, gb, C, Cout, gw_cin, ci_off, Q, NCO, NBLK, 0);
    }
    return vxm_check_launch("vxm_conv3d_k3_s3_bwd_weight");
}

}  // extern "C"
