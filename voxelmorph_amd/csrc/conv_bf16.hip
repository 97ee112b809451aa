// bf16 activations / fp32 accumulate path of the U-Net (BASELINE.json configs[1]; SURVEY.md §7 step 4, §8a "bf16 mode").
//
// Replaces, for `ConvBlock` / the flow conv (voxelmorph/torch/networks.py:290-305, 211,257), `MaxPool3d(2)` (:83-84,130) and
// `Upsample(2,'nearest')` + `cat` (:85,137-138) what torch.autocast(bfloat16) would run through MIOpen, with:
//   * activations and activation gradients in HBM as CHANNEL-BLOCKED bf16, [B][C/8][D][H][W][8]: the 8 channels of a voxel are
//     one 16-byte word, which is exactly one lane's share of a v_mfma_f32_16x16x32_bf16 operand (8 consecutive K values);
//   * master weights, biases, parameter gradients, losses, coordinates and Adam in fp32; weights are rounded to bf16 when
//     they are packed into MFMA fragment order, accumulation is fp32, outputs are rounded once (RNE) in the epilogue.
//
// Forward / backward-data: implicit GEMM, M = 16 output channels, N = 16 voxels of a W row, K = 32 = four "units" of 8 input
// channels; a unit is (kd, kw, 8-channel block) and the kh tap is walked by SLIDING over the haloed rows: the B fragment of
// haloed row r (one ds_read_b128) serves output rows r, r-1, r-2 with the weight fragments of kh = 0, 1, 2, so one LDS read
// feeds up to 3 NCT MFMAs.  Inputs are staged in 16-channel chunks (18 units -> 5 K-steps, 90 % of the K slots used).
// Backward-weight contracts over voxels: both operands are read from [voxel][16 channel] LDS tiles with the transposing
// LDS read of gfx950 (ds_read_b64_tr_b16: a lane receives 4 consecutive VOXELS of one channel), K = 32 voxels of a W row.
#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 bf_mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned bf_pack2(float lo, float hi) {            // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){lo, hi}, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

struct BfIn {                  // virtual concat of two blocked tensors (channel counts in 8-channel blocks, both even)
    const void* x0; const void* x1;
    int CB0, CB1, up0;
    int rev;                   // walk the output tiles from the end of the tensor (scheduling hint: see VXM_S3_REVERSE_TILES in include/vxm_hip.h)
};

constexpr int BF_TD = 8, BF_THREADS = 512, BF_HWV = 18, BF_STEPS = 5;
constexpr int bf_cbs(int rows) { return ((BF_TD + 2) * (rows + 2) * BF_HWV * 16 + 255) / 256 * 256; }      // LDS bytes of one 8-channel block of a chunk
constexpr int bf_wchunk(int nct) { return BF_STEPS * 3 * nct * 64; }                                       // 16-byte words of a chunk's weights
constexpr int bf_lds_bytes(int nct, int rows) { return 2 * bf_cbs(rows) + bf_wchunk(nct) * 16; }

// OUT: 0 = blocked bf16 (bias + LeakyReLU, optional LeakyReLU' mask of the previous block = fused leaky_relu_backward),
//      1 = planar fp32 [B][Cout<=4][D][H][W] (the flow head: bias only)
//      2 = blocked bf16 at HALF resolution: the sum over each voxel's 2 x 2 x 2 children (adjoint of nearest-x2 upsampling; no bias /
//          activation; optional LeakyReLU' mask of the low-resolution block) -- backward-data onto an upsampled segment
// Round 6 (late): PERSISTENT blocks with the next chunk in flight.  Rounds 2 - 5 ran one tile per block and, per 16-channel chunk,
// load -> LDS -> barrier -> MFMA -> barrier with nothing in flight during the multiplies (most full-resolution launches have one or two
// chunks per tile, so there was nothing to pipeline inside a tile either).  Now a block walks its XCD's tile range (as the fp32 split kernels,
// conv_s3.hip) and the chunk stream (tile, q) is software-pipelined: the 16-byte words of chunk c + 1 -- and its weights -- are requested into
// registers before the MFMAs of chunk c and written to LDS after them (16-channel operators: two blocks per CU, one LDS buffer each).  The
// 32-channel operators are persistent too but fetch in the store phase (PF below; DB, a double-buffered tile with one block per CU, is kept as a
// compile-time switch with its measurement).  Same tiles, same packed operators, same arithmetic in the same order: results are bit-identical to
// the round-2 kernel.  Measured at 160x192x224 (dense_bf16 step): k_bf16_conv<1,8,0> 1.33 -> 1.11 ms per step.
constexpr bool bf_db(int nct) { (void)nct; return false; }      // (see the header comment of k_bf16_conv: the double-buffered form measured slower)
constexpr int bf_lds_bytes_p(int nct, int rows) { return (bf_db(nct) ? 2 : 1) * bf_lds_bytes(nct, rows); }

template <int NCT, int ROWS, int OUT>
__global__ void __launch_bounds__(BF_THREADS, (bf_db(NCT) ? 2 : 4)) k_bf16_conv(BfIn in, const u32x4* __restrict__ wp, const float* __restrict__ bias,
                                                          void* __restrict__ y, int Cout, float act_slope, const void* __restrict__ mask,
                                                          float mask_slope, int B, int D, int H, int W, int Q) {
    VXM_DYN_SMEM(u32x4, smem);
    constexpr bool DB = bf_db(NCT);
    constexpr int HR = ROWS + 2, CBS = bf_cbs(ROWS) / 16, PLANE = HR * BF_HWV;     // in 16-byte words
    constexpr int NSLOT = 2 * (BF_TD + 2) * PLANE, NI = (NSLOT + BF_THREADS - 1) / BF_THREADS;
    constexpr int WCH = bf_wchunk(NCT), WIT = (WCH + BF_THREADS - 1) / BF_THREADS;
    constexpr bool PF = NCT == 1;           // the next chunk in flight under the MFMAs: the 16-channel instances.  The 32-channel ones have no registers
                                            // for it at two blocks per CU (154 needed, 128 there: the spilled prefetch waited for its loads at once), and
                                            // with one block per CU and a double-buffered tile they measured 7 % slower than two unpipelined blocks
                                            // that cover each other's loads: they keep that form (fetch in the store phase), persistent
    constexpr bool WPF = PF;                // (the weights of a one-chunk operator stay in LDS for every tile of the block)
    constexpr int BUF = 2 * CBS + WCH;      // 16-byte words of one buffer: [2][CBS] activations ([cb][hd][hr][hw]), then [5][3][NCT][64] weights
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, n = lane & 15;

    // tiles of this block: block b runs on XCD b % 8 (observed; speed only); XCD x owns a contiguous tile range, its blocks take the tiles round-robin
    const int nw = (W + 15) / 16, nh = (H + ROWS - 1) / ROWS, nd = (D + BF_TD - 1) / BF_TD;
    const int ntiles = B * nd * nh * nw;
    int t_lo, t_hi, t_step;
    if (ntiles >= 64) {
        const int x = blockIdx.x & 7;
        t_lo = (int)((long long)ntiles * x / 8) + (int)(blockIdx.x >> 3); t_hi = (int)((long long)ntiles * (x + 1) / 8); t_step = (int)(gridDim.x >> 3);
    } else {
        t_lo = blockIdx.x; t_hi = ntiles; t_step = gridDim.x;
    }
    if (t_lo >= t_hi) return;
    const int nchunks = ((t_hi - t_lo + t_step - 1) / t_step) * Q;
    const int g = blockIdx.y;
    const int V = D * H * W;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const int V0 = in.up0 ? Dl * Hl * Wl : V;
    struct Org { int b, d0, h0, w0; };
    auto origin = [&](int t) __attribute__((always_inline)) -> Org {
        if (in.rev) t = ntiles - 1 - t;
        const int tw = t % nw; int tq = t / nw;
        const int th = tq % nh; tq /= nh;
        return Org{tq / nd, (tq % nd) * BF_TD, th * ROWS, tw * 16};
    };

    // per-lane LDS word offset of the unit this lane group reads in K-step s: unit u = 4 s + kg = 2 (3 kd + kw) + cb
    int xoff[BF_STEPS];
#pragma unroll
    for (int s = 0; s < BF_STEPS; ++s) {
        const int u = 4 * s + kg, uu = u < 18 ? u : 0;       // units 18, 19 carry zero weights: any valid address
        const int kdkw = uu >> 1, cb = uu & 1, kd = kdkw / 3, kw = kdkw - 3 * kd;
        xoff[s] = cb * CBS + (wave + kd) * PLANE + kw + n;
    }

    // ---- the chunk in flight: 2 blocks x haloed tile, 16 bytes per slot (zero padding through the buffer descriptor), and its weights
    u32x4 xr[NI];
    [[maybe_unused]] u32x4 wr[WPF ? WIT : 1];
    int voffs[NI];
    auto load_chunk = [&](int t, int q, bool in_store_phase) __attribute__((always_inline)) {
        if (PF == in_store_phase) return;                     // (compile-time after inlining)
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));                        // (keeps the slot arithmetic inside the loop: registers)
        const Org o = origin(t);
        const bool s0 = 2 * q < in.CB0;                       // block-uniform
        const bool up = s0 && in.up0;
        const int cbg = s0 ? 2 * q : 2 * q - in.CB0, Vs = up ? V0 : V;
        const __amdgpu_buffer_rsrc_t r = s0 ? bf_rsrc(static_cast<const char*>(in.x0) + (size_t)o.b * in.CB0 * V0 * 16, (unsigned)in.CB0 * (unsigned)V0 * 16u)
                                            : bf_rsrc(static_cast<const char*>(in.x1) + (size_t)o.b * in.CB1 * V * 16, (unsigned)in.CB1 * (unsigned)V * 16u);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = tid_ + BF_THREADS * j;
            const int cb = i / ((BF_TD + 2) * PLANE), rem = i - cb * (BF_TD + 2) * PLANE;
            const int hd = rem / PLANE, r2 = rem - hd * PLANE, hh = r2 / BF_HWV, hw = r2 - hh * BF_HWV;
            const int gd = o.d0 - 1 + hd, gh = o.h0 - 1 + hh, gw = o.w0 - 1 + hw;
            const bool ok = i < NSLOT && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            const int vox = up ? ((gd >> 1) * Hl + (gh >> 1)) * Wl + (gw >> 1) : (gd * H + gh) * W + gw;
            voffs[j] = ok ? ((cbg + cb) * Vs + vox) << 4 : VXM_OOB;
            xr[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffs[j], 0, 0));
        }
        if constexpr (WPF) {
            const __amdgpu_buffer_rsrc_t rw = bf_rsrc(wp + ((size_t)g * Q + q) * WCH, WCH * 16u);
#pragma unroll
            for (int it = 0; it < WIT; ++it) wr[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tid_ + BF_THREADS * it) * 16, 0, 0));
        }
    };
    auto store_chunk = [&](int buf, int t, int q, bool first) __attribute__((always_inline)) {
        load_chunk(t, q, true);
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));
        u32x4* const Xd = smem + buf * BUF;
        if constexpr (!WPF) {
            if (first || Q > 1) {                             // block-uniform; a one-chunk operator stays in LDS for every tile of the block
                const __amdgpu_buffer_rsrc_t rw = bf_rsrc(wp + ((size_t)g * Q + q) * WCH, WCH * 16u);
#pragma unroll
                for (int it = 0; it < WIT; ++it) {
                    const int i = tid_ + BF_THREADS * it;
                    if (i < WCH) Xd[2 * CBS + i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, i * 16, 0, 0));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = tid_ + BF_THREADS * j;
            if (i < NSLOT) {
                const int cb = i / ((BF_TD + 2) * PLANE), rem = i - cb * (BF_TD + 2) * PLANE;
                Xd[cb * CBS + rem] = xr[j];
            }
        }
        if constexpr (WPF) {
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int i = tid_ + BF_THREADS * it;
                if (i < WCH) Xd[2 * CBS + i] = wr[it];
            }
        }
    };

    f32x4 acc[NCT][ROWS];
    int tile = t_lo, q = 0;
    load_chunk(tile, 0, false);
    store_chunk(0, tile, 0, true);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const bool has_next = c + 1 < nchunks;                // block-uniform
        const bool last_q = q + 1 == Q;
        const int tile_n = last_q ? tile + t_step : tile, q_n = last_q ? 0 : q + 1;
        if (has_next) load_chunk(tile_n, q_n, false);         // in flight under the MFMAs below
        const int buf = DB ? (c & 1) : 0;
        const u32x4* const Xs = smem + buf * BUF;
        const u32x4* const Ws = Xs + 2 * CBS;
        if (q == 0) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < ROWS; ++r) acc[ct][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        // ---- 5 K-steps x (ROWS + 2) haloed rows: one B read per row, up to 3 NCT MFMAs per read
#pragma unroll
        for (int s = 0; s < BF_STEPS; ++s) {
            u32x4 a[3][NCT];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) a[kh][ct] = Ws[((s * 3 + kh) * NCT + ct) * 64 + lane];
#pragma unroll
            for (int hr = 0; hr < HR; ++hr) {
                const u32x4 bf = Xs[xoff[s] + hr * BF_HWV];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int row = hr - kh;
                    if (row >= 0 && row < ROWS) {
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) acc[ct][row] = bf_mfma(a[kh][ct], bf, acc[ct][row]);
                    }
                }
            }
        }
        // (the address VGPR of a buffer load that is still in flight must not be handed to the LDS reads above: see keep_offsets in conv_s3.hip)
        if constexpr (PF) {
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(voffs[j]));
        }

        if (last_q) {
        // ---- epilogue.  D layout: lane (kg, n) holds output channels 16 ct + 4 kg + j (j = 0..3) of voxel (d, h0 + row, w0 + n).
        const Org o = origin(tile);
        const int b = o.b, d0 = o.d0, h0 = o.h0, w0 = o.w0;
        const int d = d0 + wave, w = w0 + n;
        const bool vok = d < D && w < W;
        if constexpr (OUT == 1) {
            float* const yp = static_cast<float*>(y) + (size_t)b * Cout * V;
            if (kg == 0 && vok) {
#pragma unroll
                for (int row = 0; row < ROWS; ++row)
                    if (h0 + row < H) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < Cout) {
                                float v = acc[0][row][j] + (bias ? bias[j] : 0.0f);
                                v = v > 0.0f ? v : v * act_slope;
                                yp[(size_t)j * V + (d * H + h0 + row) * W + w] = v;
                            }
                    }
            }
        } else if constexpr (OUT == 2) {
            // the adjoint of nearest-x2 upsampling fused in: every voxel's result is rounded to bf16 (the value the unfused pair of
            // kernels stored), the 2 x 2 x 2 children are added in fp32 -- h pairs in registers, w pairs across lanes n ^ 1, d pairs
            // across waves through LDS -- and the sum, times LeakyReLU'(mask) of the LOW-resolution block, is stored at low resolution.
            __syncthreads();                                        // every wave is done reading this buffer
            f32x4* const ex = reinterpret_cast<f32x4*>(smem + buf * BUF);         // [4 odd waves][NCT][ROWS / 2][64]
            f32x4 sum[NCT][ROWS / 2];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int hp = 0; hp < ROWS / 2; ++hp) {
                    const f32x4 a0 = acc[ct][2 * hp], a1 = acc[ct][2 * hp + 1];
                    const unsigned p0 = bf_pack2(a0[0], a0[1]), p1 = bf_pack2(a0[2], a0[3]), p2 = bf_pack2(a1[0], a1[1]), p3 = bf_pack2(a1[2], a1[3]);
                    f32x4 t = {bf_lo(p0) + bf_lo(p2), bf_hi(p0) + bf_hi(p2), bf_lo(p1) + bf_lo(p3), bf_hi(p1) + bf_hi(p3)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[j] += __shfl_xor(t[j], 1);
                    sum[ct][hp] = t;
                    if (wave & 1) ex[(((wave >> 1) * NCT + ct) * (ROWS / 2) + hp) * 64 + lane] = t;
                }
            __syncthreads();
            if (!((wave & 1) || (n & 1))) {
                const int Vl = Dl * Hl * Wl, CBo = Cout >> 3;
                const int dl = (d0 + wave) >> 1, wl = (w0 + n) >> 1;
                const __amdgpu_buffer_rsrc_t ry = bf_rsrc(static_cast<char*>(y) + (size_t)b * CBo * Vl * 16, (unsigned)CBo * (unsigned)Vl * 16u);
                const __amdgpu_buffer_rsrc_t rm = bf_rsrc(mask ? static_cast<const char*>(mask) + (size_t)b * CBo * Vl * 16 : y, (unsigned)CBo * (unsigned)Vl * 16u);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int co = (g * NCT + ct) * 16 + 4 * kg, pb = co >> 3;
                    const bool cok = dl < Dl && wl < Wl && pb < CBo;
#pragma unroll
                    for (int hp = 0; hp < ROWS / 2; ++hp) {
                        const int hl = (h0 >> 1) + hp;
                        if (hl < Hl) {                                  // wave-uniform
                            const f32x4 ov = ex[(((wave >> 1) * NCT + ct) * (ROWS / 2) + hp) * 64 + lane];
                            float v[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = sum[ct][hp][j] + ov[j];
                            const int voff = cok ? (((pb * Dl + dl) * Hl + hl) * Wl + wl) * 16 + (kg & 1) * 8 : VXM_OOB;
                            if (mask) {
                                const u32x2 m = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rm, voff, 0, 0));
                                v[0] *= vxm_lrelu_grad(bf_lo(m.x), mask_slope); v[1] *= vxm_lrelu_grad(bf_hi(m.x), mask_slope);
                                v[2] *= vxm_lrelu_grad(bf_lo(m.y), mask_slope); v[3] *= vxm_lrelu_grad(bf_hi(m.y), mask_slope);
                            }
                            const u32x2 st = {bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3])};
                            __builtin_amdgcn_raw_buffer_store_b64(st, ry, voff, 0, 0);
                        }
                    }
                }
            }
        } else {
            const int CBo = Cout >> 3;
            const __amdgpu_buffer_rsrc_t ry = bf_rsrc(static_cast<char*>(y) + (size_t)b * CBo * V * 16, (unsigned)CBo * (unsigned)V * 16u);
            const __amdgpu_buffer_rsrc_t rm = bf_rsrc(mask ? static_cast<const char*>(mask) + (size_t)b * CBo * V * 16 : y, (unsigned)CBo * (unsigned)V * 16u);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int co = (g * NCT + ct) * 16 + 4 * kg;           // first of this lane's 4 channels
                const int pb = co >> 3;                                // its 8-channel block
                const bool cok = vok && pb < CBo;
                float bz[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bz[j] = (bias && co + j < Cout) ? bias[co + j] : 0.0f;
                // rows in pairs: the lane groups kg and kg ^ 1 hold the two halves of an 8-channel block; v_permlane16_swap (gfx950; lane
                // pattern probed in tools/probe/permlane_swap_probe.hip) hands the even group the other half of row 2 rp and the odd group
                // the other half of row 2 rp + 1, so that every lane stores ONE full 16-byte word (half the store instructions: the
                // epilogue was 8 % store issue + 12 % store traffic of these launches, measured with the stores dropped)
#pragma unroll
                for (int rp = 0; rp < ROWS / 2; ++rp) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int row = 2 * rp + e;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = acc[ct][row][j] + bz[j];
                            v[j] = v[j] > 0.0f ? v[j] : v[j] * act_slope;
                        }
                        if (mask) {
                            const int moff = (cok && h0 + row < H) ? (((pb * D + d) * H + h0 + row) * W + w) * 16 + (kg & 1) * 8 : VXM_OOB;
                            const u32x2 m = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rm, moff, 0, 0));
                            v[0] *= vxm_lrelu_grad(bf_lo(m.x), mask_slope); v[1] *= vxm_lrelu_grad(bf_hi(m.x), mask_slope);
                            v[2] *= vxm_lrelu_grad(bf_lo(m.y), mask_slope); v[3] *= vxm_lrelu_grad(bf_hi(m.y), mask_slope);
                        }
                        pk[e][0] = bf_pack2(v[0], v[1]);
                        pk[e][1] = bf_pack2(v[2], v[3]);
                    }
                    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                    const int hrow = h0 + 2 * rp + (kg & 1);
                    const int voff = (cok && hrow < H) ? (((pb * D + d) * H + hrow) * W + w) * 16 : VXM_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){s0.x, s1.x, s0.y, s1.y}, ry, voff, 0, 0);
                }
            }
        }
        }
        if (!DB) __syncthreads();                               // every wave is done reading the one buffer
        if (has_next) store_chunk(DB ? ((c + 1) & 1) : 0, tile_n, q_n, false);      // (waits for the requests issued in front of the MFMAs)
        __syncthreads();
        tile = tile_n; q = q_n;
    }
}

// w: [Cw_out][Cw_in][27] fp32 (reference layout) -> bf16 [G][Q][5][3][NCT][64 lanes][8]: lane (kg, m) of K-step s / row tap kh holds,
// for output channel 16 (g NCT + ct) + m, the 8 input channels of unit u = 4 s + kg (kd, kw, block) of chunk q.
// Operator: y[o] = sum_i Wop[o][i][tap] x[i];  forward Wop[o][i][t] = w[o][ci_lo + i][t] (InC = ci_n inputs, OutC = Cw_out);
// flip (backward-data onto the input-channel range [ci_lo, ci_lo + ci_n)): Wop[o][i][t] = w[i][ci_lo + o][26 - t]
// (InC = Cw_out inputs, OutC = ci_n).
__device__ __forceinline__ void bf_pack_word(const float* __restrict__ w, u32x4* __restrict__ wp, size_t i, int Cw_in, int ci_lo, int flip, int InC,
                                             int OutC, int NCT, int Q) {
    size_t r = i;
    const int lane = r % 64; r /= 64;
    const int ct = r % NCT; r /= NCT;
    const int kh = r % 3; r /= 3;
    const int s = r % BF_STEPS; r /= BF_STEPS;
    const int q = r % Q; const int g = (int)(r / Q);
    const int kg = lane >> 4, m = lane & 15, u = 4 * s + kg;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int o = (g * NCT + ct) * 16 + m;
    if (u < 18 && o < OutC) {
        const int kdkw = u >> 1, cb = u & 1, kd = kdkw / 3, kw = kdkw % 3, tap = kd * 9 + kh * 3 + kw;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = q * 16 + cb * 8 + e;
            if (ci < InC) v[e] = flip ? w[((size_t)ci * Cw_in + ci_lo + o) * 27 + (26 - tap)] : w[((size_t)o * Cw_in + ci_lo + ci) * 27 + tap];
        }
    }
    wp[i] = (u32x4){bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3]), bf_pack2(v[4], v[5]), bf_pack2(v[6], v[7])};
}

__global__ void __launch_bounds__(256) k_bf16_pack_weights(const float* __restrict__ w, u32x4* __restrict__ wp, int Cw_in, int ci_lo, int flip,
                                                           int InC, int OutC, int NCT, int Q, size_t words) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < words) bf_pack_word(w, wp, i, Cw_in, ci_lo, flip, InC, OutC, NCT, Q);
}

// All the operators of a network in ONE launch (a training step re-packs ~27 operators after every optimiser step: 27 launches of
// ~4 us each were 2 % of the bf16 step).  The job table travels in the kernel arguments; a block finds its job by a scan.
#define BF_PACK_JOBS 48
struct BfPackJob {
    const float* w;
    u32x4* wp;
    int Cw_in, ci_lo, flip, InC, OutC, NCT, Q;
    unsigned first_block, words;
};
struct BfPackBatch {
    BfPackJob job[BF_PACK_JOBS];
    int n;
};
__global__ void __launch_bounds__(256) k_bf16_pack_weights_batch(const BfPackBatch batch) {
    int j = 0;
    while (j + 1 < batch.n && blockIdx.x >= batch.job[j + 1].first_block) ++j;          // block-uniform
    const BfPackJob& jb = batch.job[j];
    const size_t i = (size_t)(blockIdx.x - jb.first_block) * 256 + threadIdx.x;
    if (i < jb.words) bf_pack_word(jb.w, jb.wp, i, jb.Cw_in, jb.ci_lo, jb.flip, jb.InC, jb.OutC, jb.NCT, jb.Q);
}

// planar fp32 [B][C0 (+ C1)][V] -> blocked bf16 [B][CB][V][8], channels beyond C0 + C1 zero.  One thread per (block, voxel).
__global__ void __launch_bounds__(256) k_bf16_to_blocked(const float* __restrict__ x0, int C0, long long bs0, const float* __restrict__ x1, int C1,
                                                         long long bs1, u32x4* __restrict__ out, int CB, long long V) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= V * CB) return;
    const size_t b = blockIdx.y;
    const int cb = (int)(i / V);
    const long long vox = i - (long long)cb * V;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cb * 8 + e;
        v[e] = c < C0 ? x0[b * bs0 + (size_t)c * V + vox] : (c < C0 + C1 ? x1[b * bs1 + (size_t)(c - C0) * V + vox] : 0.0f);
    }
    out[b * (size_t)CB * V + i] = (u32x4){bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3]), bf_pack2(v[4], v[5]), bf_pack2(v[6], v[7])};
}

// blocked bf16 -> planar fp32 [B][C][V] (the first C channels)
__global__ void __launch_bounds__(256) k_bf16_from_blocked(const u32x4* __restrict__ x, int CB, float* __restrict__ out, int C, long long V) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nb = (C + 7) / 8;
    if (i >= V * nb) return;
    const size_t b = blockIdx.y;
    const int cb = (int)(i / V);
    const long long vox = i - (long long)cb * V;
    const u32x4 u = x[b * (size_t)CB * V + i];
    const float v[8] = {bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y), bf_lo(u.z), bf_hi(u.z), bf_lo(u.w), bf_hi(u.w)};
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (cb * 8 + e < C) out[b * (size_t)C * V + (size_t)(cb * 8 + e) * V + vox] = v[e];
}

__device__ __forceinline__ void bf_unpack8(u32x4 u, float (&v)[8]) {
    v[0] = bf_lo(u.x); v[1] = bf_hi(u.x); v[2] = bf_lo(u.y); v[3] = bf_hi(u.y);
    v[4] = bf_lo(u.z); v[5] = bf_hi(u.z); v[6] = bf_lo(u.w); v[7] = bf_hi(u.w);
}
__device__ __forceinline__ u32x4 bf_pack8(const float (&v)[8]) {
    return (u32x4){bf_pack2(v[0], v[1]), bf_pack2(v[2], v[3]), bf_pack2(v[4], v[5]), bf_pack2(v[6], v[7])};
}

// MaxPool3d(2) on blocked tensors: one thread per pooled (block, voxel), 8 channels at a time
__global__ void __launch_bounds__(256) k_bf16_maxpool2_fwd(const u32x4* __restrict__ x, u32x4* __restrict__ y, int CB, int D, int H, int W) {
    const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1;
    const long long V2 = (long long)D2 * H2 * W2, V = (long long)D * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= V2 * CB) return;
    const size_t b = blockIdx.y;
    const int cb = (int)(i / V2);
    const int q = (int)(i - (long long)cb * V2);
    const int w = q % W2, t = q / W2, h = t % H2, d = t / H2;
    const u32x4* p = x + b * (size_t)CB * V + ((size_t)cb * D + 2 * d) * H * W + (size_t)(2 * h) * W + 2 * w;
    float m[8];
    bf_unpack8(p[0], m);
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        float v[8];
        bf_unpack8(p[(size_t)((k >> 2) & 1) * H * W + (size_t)((k >> 1) & 1) * W + (k & 1)], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (v[e] > m[e] || v[e] != v[e]) ? v[e] : m[e];       // ATen: (val > maxval) || isnan(val)
    }
    y[b * (size_t)CB * V2 + i] = bf_pack8(m);
}

// backward of MaxPool3d(2) fused with the skip-branch gradient add and LeakyReLU' of the pooled ConvBlock output:
// dz[child] = ((child is the FIRST arg-max of its 2x2x2 block ? gpool : 0) + gskip[child]) * LeakyReLU'(x[child])
__global__ void __launch_bounds__(256) k_bf16_maxpool2_bwd(const u32x4* __restrict__ x, const u32x4* __restrict__ gpool, const u32x4* __restrict__ gskip,
                                                           u32x4* __restrict__ dz, float slope, int CB, int D, int H, int W) {
    const int D2 = D >> 1, H2 = H >> 1, W2 = W >> 1;
    const long long V2 = (long long)D2 * H2 * W2, V = (long long)D * H * W;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= V2 * CB) return;
    const size_t b = blockIdx.y;
    const int cb = (int)(i / V2);
    const int q = (int)(i - (long long)cb * V2);
    const int w = q % W2, t = q / W2, h = t % H2, d = t / H2;
    const size_t base = b * (size_t)CB * V + ((size_t)cb * D + 2 * d) * H * W + (size_t)(2 * h) * W + 2 * w;
    float xv[8][8], m[8], gp[8];
    int arg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bf_unpack8(x[base + (size_t)((k >> 2) & 1) * H * W + (size_t)((k >> 1) & 1) * W + (k & 1)], xv[k]);
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = xv[0][e]; arg[e] = 0; }
#pragma unroll
    for (int k = 1; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (xv[k][e] > m[e] || xv[k][e] != xv[k][e]) { m[e] = xv[k][e]; arg[e] = k; }
    bf_unpack8(gpool[b * (size_t)CB * V2 + i], gp);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t o = base + (size_t)((k >> 2) & 1) * H * W + (size_t)((k >> 1) & 1) * W + (k & 1);
        float gs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, r[8];
        if (gskip) bf_unpack8(gskip[o], gs);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = ((arg[e] == k ? gp[e] : 0.0f) + gs[e]) * vxm_lrelu_grad(xv[k][e], slope);
        dz[o] = bf_pack8(r);
    }
}

// backward of Upsample(2,'nearest') fused with LeakyReLU' of the decoder block: dz[q] = (sum of the 8 children of g) * LeakyReLU'(y[q])
__global__ void __launch_bounds__(256) k_bf16_upsample2_bwd(const u32x4* __restrict__ g, const u32x4* __restrict__ y, u32x4* __restrict__ dz, float slope,
                                                            int CB, int Dl, int Hl, int Wl) {
    const long long Vl = (long long)Dl * Hl * Wl;
    const int H = 2 * Hl, W = 2 * Wl;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= Vl * CB) return;
    const size_t b = blockIdx.y;
    const int cb = (int)(i / Vl);
    const int q = (int)(i - (long long)cb * Vl);
    const int w = q % Wl, t = q / Wl, h = t % Hl, d = t / Hl;
    const size_t base = b * (size_t)CB * Vl * 8 + ((size_t)cb * 2 * Dl + 2 * d) * H * W + (size_t)(2 * h) * W + 2 * w;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v[8];
        bf_unpack8(g[base + (size_t)((k >> 2) & 1) * H * W + (size_t)((k >> 1) & 1) * W + (k & 1)], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
    if (y) {
        float yv[8];
        bf_unpack8(y[b * (size_t)CB * Vl + i], yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] *= vxm_lrelu_grad(yv[e], slope);
    }
    dz[b * (size_t)CB * Vl + i] = bf_pack8(s);
}

// dz = g * LeakyReLU'(y), elementwise on blocked tensors (n 16-byte words)
__global__ void __launch_bounds__(256) k_bf16_lrelu_bwd(const u32x4* __restrict__ g, const u32x4* __restrict__ y, u32x4* __restrict__ dz, float slope,
                                                        long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float gv[8], yv[8];
        bf_unpack8(g[i], gv);
        bf_unpack8(y[i], yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[e] *= vxm_lrelu_grad(yv[e], slope);
        dz[i] = bf_pack8(gv);
    }
}

// ------------------------------------------------------------------------------------------
// backward-weight: gW[co][ci][tap] = sum_v dZ[co][v] X[ci][v + tap - 1],  gb[co] = sum_v dZ[co][v]
// ------------------------------------------------------------------------------------------
// M = 16 output channels (A = dZ^T), N = 16 input channels (B = X), K = 32 voxels of a W row.  Block = 6 waves = (depth slice
// of a 2 x 8 x 32 voxel tile) x (kd); a wave keeps the 9 (kh, kw) taps x NCO output-channel tiles of its kd in registers
// (36 NCO accumulator VGPRs) over ALL its tiles and slides over the 10 haloed X rows: the three kw-shifted B fragments of a
// row serve kh = 0, 1, 2 with the A fragments of output rows r, r-1, r-2.
// HBM-bound (X and dZ are streamed once per 16-input-channel chunk), so the kernel is built around the stream:
//  * one persistent block per CU walks DOWN THE DEPTH of a (b, th, tw) column: consecutive tiles share two of their four
//    haloed X planes, which stay in a 6-slot LDS ring -- a tile fetches 2 new planes (x 1.33 halo amplification from H / W
//    only, instead of x 2.66);
//  * the loads of tile t+1 (2 X planes + the dZ tile, 7-10 dwordx4 per lane) are in flight in registers under the MFMAs of
//    tile t and written to LDS afterwards (ring slots / the other dZ buffer): one barrier per tile;
//  * a block owns one chunk and a contiguous range of (column, depth segment) tasks; its accumulators are written once, as
//    partials that k_bf16_reduce_partials sums in a fixed order (deterministic).
constexpr int BWB_WAVES = 6;                                                       // per output-channel tile: (depth slice, kd)
constexpr int bwb_threads(int nco) { return 64 * BWB_WAVES * nco; }
constexpr int BWB_TD = 2, BWB_TH = 8, BWB_TW = 32;
constexpr int BWB_XW = BWB_TW + 2;                                                 // haloed row, voxels
constexpr int BWB_PLANE = (BWB_TH + 2) * BWB_XW * 32;                              // one haloed X plane: [hh][hw][16 ci], bytes
constexpr int BWB_RING = 6;
constexpr int BWB_ZBYTES = BWB_TD * BWB_TH * BWB_TW * 32;                          // dZ tile per output-channel tile: [ds][row][w][16 co]
constexpr int bwb_lds_bytes(int nco) { return BWB_RING * BWB_PLANE + 2 * nco * BWB_ZBYTES; }

// transposing read of a [voxel][16 channel] bf16 tile (32-byte rows): the lane pattern supplies, per 16-lane group, the
// 16 8-byte pieces of 4 consecutive voxels; lane (n, kg) receives channel n of voxels 4 kg .. 4 kg + 3 (tools/probe/tr16_probe.hip)
__device__ __forceinline__ u32x2 bf_tr_read(const char* lds_base, int byte_off) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(__attribute__((address_space(3))) void*)(lds_base + byte_off));
    return __builtin_bit_cast(u32x2, v);
}

struct BwbTasks { int ncol, nseg, seg_len, nd, nh, nw; };       // tasks = (column (b, th, tw), depth segment), seg_len tiles each

template <int NCO>
__global__ void __launch_bounds__(bwb_threads(NCO), 1) k_bf16_conv_bwd_weight(BfIn in, const void* __restrict__ dz, float* __restrict__ part,
                                                                         int D, int H, int W, int NBLK, BwbTasks tk) {
    VXM_DYN_SMEM(char, smem);
    char* const Xs = smem;                                       // ring of 6 haloed planes
    char* const Zs = smem + BWB_RING * BWB_PLANE;                // [2 buffers][NCO][ZBYTES]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BWB_THREADS = bwb_threads(NCO);
    const int cow = wave / BWB_WAVES, w6 = wave - BWB_WAVES * cow;    // this wave's output-channel tile, (depth slice, kd)
    const int ds = w6 / 3, kd = w6 - 3 * ds;
    // block -> (task range x, 16-channel chunk q of the virtual concat).  The Q blocks of one x read the SAME dz tiles at the same pace:
    // they are given workgroup ids 8 apart so that they run side by side on ONE XCD (workgroup id % 8) and share those tiles through
    // its L2 (as (x, q) = (blockIdx.x, blockIdx.y) with 85 x 3 blocks the three readers sat on three XCDs and dz came from HBM three
    // times: 1.87 GB fetched per launch of the 48 -> 32 layer, measured, for 0.72 GB of operands).
    const int Q = gridDim.x / NBLK, xmain = NBLK & ~7;
    int bx, q;
    if ((int)blockIdx.x < xmain * Q) {
        const int j = blockIdx.x >> 3;
        q = j % Q;
        bx = (j / Q) * 8 + (blockIdx.x & 7);
    } else {
        const int r = blockIdx.x - xmain * Q;
        bx = xmain + r / Q;
        q = r % Q;
    }
    const int ntask = tk.ncol * tk.nseg;
    // Which tasks this block walks (round 4, as k_s3_bwd_weight): tasks numbered depth-segment-major, columns consecutive; XCD x (= bx & 7)
    // owns a contiguous task range and its blocks take the tasks of that range round-robin, so the blocks an XCD runs side by side work on
    // W-neighbouring columns at the same depth and share the halo sectors in its L2.  Large volumes only (measured on the fp32 twin).
    const bool task_rr = (NBLK & 7) == 0 && (long long)D * H * W >= (1ll << 21);
    int k_lo, k_hi, k_step;
    if (task_rr) {
        const int x = bx & 7;
        k_lo = (int)((long long)ntask * x / 8) + (bx >> 3); k_hi = (int)((long long)ntask * (x + 1) / 8); k_step = NBLK >> 3;
    } else {
        k_lo = (int)((long long)ntask * bx / NBLK); k_hi = (int)((long long)ntask * (bx + 1) / NBLK); k_step = 1;
    }

    const int V = D * H * W;
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const bool s0 = 2 * q < in.CB0;
    const bool up = s0 && in.up0;
    const int Vs = up ? Dl * Hl * Wl : V;
    const int CBs = s0 ? in.CB0 : in.CB1, cbg = s0 ? 2 * q : 2 * q - in.CB0;
    constexpr int CBz = 2 * NCO;

    f32x4 acc[3][3], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh][kw] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lp = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;     // lane pattern of the transposing read
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};       // bf16 1.0 x 8: B operand of the bias sum

    // staging roles of a thread, fixed for the whole kernel: slot i = tid + THREADS j of one haloed X plane ([hh][hw][cb], 16 bytes)
    // and of the dZ tile ([co][zd][zh][zw][cb]).  Per (column) task the lane offsets are rebuilt once -- H / W validity does not
    // change down a column -- and per tile only the wave-uniform depth offset moves: a few VALU per load instead of ~40.
    constexpr int NXP = (BWB_TH + 2) * BWB_XW * 2, NXI = (NXP + BWB_THREADS - 1) / BWB_THREADS;             // one plane
    constexpr int NZ = BWB_TD * BWB_TH * BWB_TW * 2, NZI = (NZ * NCO + BWB_THREADS - 1) / BWB_THREADS;
    u32x4 xv[2][NXI], zv[NZI];
    int xoff[NXI], zoff[NZI];                                    // byte offsets inside a depth slice of the tensor (VXM_OOB: padding)
    const int HWs = up ? Hl * Wl : H * W;                        // voxels per depth slice of the X source

    for (int task = k_lo; task < k_hi; task += k_step) {
        const int seg = task_rr ? task / tk.ncol : task % tk.nseg, col = task_rr ? task - seg * tk.ncol : task / tk.nseg;
        const int tw = col % tk.nw; int cq = col / tk.nw;
        const int th = cq % tk.nh; const int b = cq / tk.nh;
        const int td0 = seg * tk.seg_len, ntile = min(tk.seg_len, tk.nd - td0);
        const int dbase = td0 * BWB_TD, h0 = th * BWB_TH, w0 = tw * BWB_TW;
        const __amdgpu_buffer_rsrc_t rx = bf_rsrc(static_cast<const char*>(s0 ? in.x0 : in.x1) + (size_t)b * CBs * Vs * 16, (unsigned)CBs * (unsigned)Vs * 16u);
        const __amdgpu_buffer_rsrc_t rz = bf_rsrc(static_cast<const char*>(dz) + (size_t)b * CBz * V * 16, (unsigned)CBz * (unsigned)V * 16u);
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int i = tid + BWB_THREADS * j;
            const int cb = i & 1, v = i >> 1, hh = v / BWB_XW, hw = v - hh * BWB_XW;
            const int gh = h0 - 1 + hh, gw = w0 - 1 + hw;
            const bool ok = i < NXP && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            xoff[j] = !ok ? VXM_OOB : ((cbg + cb) * Vs + (up ? (gh >> 1) * Wl + (gw >> 1) : gh * W + gw)) << 4;
        }
#pragma unroll
        for (int j = 0; j < NZI; ++j) {
            const int i = tid + BWB_THREADS * j;
            const int co = i / NZ, r = i - co * NZ;
            const int cb = r & 1, v = r >> 1;
            const int zd = v / (BWB_TH * BWB_TW), r2 = v - zd * BWB_TH * BWB_TW, zh = r2 / BWB_TW, zw = r2 - zh * BWB_TW;
            const bool ok = i < NZ * NCO && h0 + zh < H && w0 + zw < W;
            // depth validity: the tile's second slice (zd = 1) may lie beyond D -- folded in per tile through `zlast`
            zoff[j] = !ok ? VXM_OOB : (((2 * co + cb) * V + (zd * H + h0 + zh) * W + w0 + zw) << 4) | zd;      // bit 0: depth slice of the slot
        }

        // planes p0, p0 + 1 of this task (plane p = global depth dbase - 1 + p) -> registers / -> ring slots p % 6
        auto load_planes = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const int gd = dbase - 1 + p0 + pl;                 // wave-uniform
                const bool dok = (unsigned)gd < (unsigned)D;
                const int soff = dok ? ((up ? gd >> 1 : gd) * HWs) << 4 : 0;
#pragma unroll
                for (int j = 0; j < NXI; ++j)
                    xv[pl][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, dok ? xoff[j] : VXM_OOB, soff, 0));
            }
        };
        auto store_planes = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                char* const dst = Xs + ((p0 + pl) % BWB_RING) * BWB_PLANE;
#pragma unroll
                for (int j = 0; j < NXI; ++j)
                    if (tid + BWB_THREADS * j < NXP) *reinterpret_cast<u32x4*>(dst + (tid + BWB_THREADS * j) * 16) = xv[pl][j];
            }
        };
        auto load_z = [&](int t) __attribute__((always_inline)) {
            const int gd = dbase + t * BWB_TD;                      // first depth slice of the tile (always < D)
            const bool zlast = gd + 1 < D;
            const int soff = (gd * H * W) << 4;
#pragma unroll
            for (int j = 0; j < NZI; ++j)
                zv[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, ((zoff[j] & 1) && !zlast) ? VXM_OOB : (zoff[j] & ~15), soff, 0));
        };
        auto store_z = [&](int t) __attribute__((always_inline)) {
            char* const zb = Zs + (t & 1) * NCO * BWB_ZBYTES;
#pragma unroll
            for (int j = 0; j < NZI; ++j)
                if (tid + BWB_THREADS * j < NZ * NCO) *reinterpret_cast<u32x4*>(zb + (tid + BWB_THREADS * j) * 16) = zv[j];
        };

        __syncthreads();                                        // every wave is done with the previous task
        load_planes(0);
        load_z(0);
        store_planes(0);
        load_planes(2);
        store_z(0);
        store_planes(2);
        __syncthreads();

        for (int t = 0; t < ntile; ++t) {
            const bool more = t + 1 < ntile;                     // wave-uniform
            const char* const xp = Xs + ((2 * t + ds + kd) % BWB_RING) * BWB_PLANE;
            const char* const zp = Zs + ((t & 1) * NCO + cow) * BWB_ZBYTES;
            // the loads of tile t + 1 (2 X planes + the dZ tile) are in flight under the MFMAs of tile t
            if (more) { load_planes(2 * t + 4); load_z(t + 1); }
            __builtin_amdgcn_sched_barrier(0);
            // haloed rows of X plane 2 t + ds + kd: per row three kw-shifted B fragments; the A fragments (dZ rows hr, hr - 1,
            // hr - 2 of this wave's output-channel tile) are re-read per use -- LDS has the headroom
#pragma unroll
            for (int hr = 0; hr < BWB_TH + 2; ++hr) {
                u32x4 bx[3];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int xb = (hr * BWB_XW + kw) * 32 + lp;
                    const u32x2 lo = bf_tr_read(xp, xb), hi = bf_tr_read(xp, xb + 16 * 32);
                    bx[kw] = (u32x4){lo.x, lo.y, hi.x, hi.y};
                }
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int row = hr - kh;
                    if (row >= 0 && row < BWB_TH) {
                        const int zb = ((ds * BWB_TH + row) * BWB_TW) * 32 + lp;
                        const u32x2 lo = bf_tr_read(zp, zb), hi = bf_tr_read(zp, zb + 16 * 32);
                        const u32x4 az = {lo.x, lo.y, hi.x, hi.y};
                        if (kh == 0 && kd == 0) accb = bf_mfma(az, ones, accb);
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) acc[kh][kw] = bf_mfma(az, bx[kw], acc[kh][kw]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) { store_planes(2 * t + 4); store_z(t + 1); }   // ring slots / dZ buffer last read by tile t - 1
            __syncthreads();
        }
    }

    // ---- partials: part[blk][q][ds][tap 0..27][co 16 NCO][ci 16]; D layout: lane (kg, n) holds co = 4 kg + r, ci = n
    float* const pp = part + ((((size_t)bx * Q + q) * BWB_TD + ds) * 28) * (16 * NCO) * 16;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                pp[((size_t)(kd * 9 + kh * 3 + kw) * (16 * NCO) + cow * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[kh][kw][r];
    if (kd == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[((size_t)27 * (16 * NCO) + cow * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = accb[r];
    }
}

// gw[co][ci][tap] = sum over the (block, depth slice) partials in a fixed order; gb[co] from tap slot 27 of chunk 0.
// Block = 64 consecutive outputs (ordered like a partial: chunk, tap, co, ci -> coalesced 256-byte reads) x 16 slices of the
// 2 NBLK partials, 8 loads in flight per thread, slices combined through LDS in a fixed tree (deterministic).
__global__ void __launch_bounds__(1024) k_bf16_reduce_partials(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb,
                                                               int Cin_w, int Cout_w, int Q, int NCO, int NBLK) {
    __shared__ float sm[16][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const int CoP = 16 * NCO, per_q = 28 * CoP * 16, n = Q * per_q;
    const int e = blockIdx.x * 64 + x;
    float tot = 0.0f;
    if (e < n) {
        const int q = e / per_q, r = e - q * per_q;
        const size_t stride_blk = (size_t)Q * BWB_TD * per_q;
        const float* p = part + (size_t)q * BWB_TD * per_q + r;       // + blk * stride_blk + ds * per_q
        const int K = 2 * NBLK;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = y;
        for (; k + 16 * 7 < K; k += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + 16 * u;
                s[u] += p[(size_t)(kk >> 1) * stride_blk + (size_t)(kk & 1) * per_q];
            }
        }
        for (; k < K; k += 16) s[0] += p[(size_t)(k >> 1) * stride_blk + (size_t)(k & 1) * per_q];
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
        const int tap = r / (CoP * 16), co = (r / 16) % CoP, ci = q * 16 + (r & 15);
        if (tap < 27) {
            if (co < Cout_w && ci < Cin_w) gw[((size_t)co * Cin_w + ci) * 27 + tap] = sum;
        } else if (gb != nullptr && q == 0 && (r & 15) == 0 && co < Cout_w) {
            gb[co] = sum;
        }
    }
}

int bf_check(const char* fn, int C0, int C1, int up0, int Cout, int B, int D, int H, int W) {
    VXM_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, VXM_ERR_BAD_SHAPE, "%s: bad shape", fn);
    VXM_REQUIRE(C0 % 16 == 0 && C1 % 16 == 0, VXM_ERR_BAD_SHAPE,
                "%s: blocked bf16 tensors carry multiples of 16 channels (two 8-channel blocks), got %d + %d", fn, C0, C1);
    VXM_REQUIRE(!up0 || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), VXM_ERR_BAD_SHAPE, "%s: upsampled segment needs even extents", fn);
    const int cmax = C0 > C1 ? (C0 > Cout ? C0 : Cout) : (C1 > Cout ? C1 : Cout);
    VXM_REQUIRE((long long)cmax * D * H * W * 2 < (1ll << 31), VXM_ERR_BAD_SHAPE,
                "%s: a tensor of one sample must stay below 2 GiB (32-bit byte offsets in the buffer descriptors)", fn);
    return VXM_OK;
}
int bf_nct(int OutC) { return OutC <= 16 ? 1 : 2; }
bool bf_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int bwb_cus() {
    static const int cus = [] {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }();
    return cus;
}

template <int NCT, int ROWS, int OUT>
void bf_launch_conv(const BfIn& in, const void* wp, const float* bias, void* y, int Cout, float slope, const void* mask, float mask_slope,
                    int B, int D, int H, int W, hipStream_t s) {
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bf16_conv<NCT, ROWS, OUT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  bf_lds_bytes_p(NCT, ROWS));
        return true;
    }();
    (void)attr;
    const int Q = (in.CB0 + in.CB1) / 2;
    const long long ntiles = (long long)B * ((D + BF_TD - 1) / BF_TD) * ((H + ROWS - 1) / ROWS) * ((W + 15) / 16);
    unsigned gx = ntiles >= 64 ? (unsigned)(8 * ((ntiles + 7) / 8)) : (unsigned)ntiles;
    const int G = (Cout + 16 * NCT - 1) / (16 * NCT);
    // persistent blocks: as many as fit the chip at once (two per CU for the 16-channel instances, one for the double-buffered 32-channel ones),
    // each walking its XCD's tile range; VXM_BF16_PERSIST=n: n blocks per CU, 0: one tile per block, < 0: that many blocks in all (tests)
    static const int persist = [] { const char* e = getenv("VXM_BF16_PERSIST"); return e ? atoi(e) : (bf_db(NCT) ? 1 : 2); }();
    if (persist != 0 && ntiles >= 64) {
        const unsigned want = persist > 0 ? (unsigned)(bwb_cus() * persist / G) : (unsigned)(-persist);
        const unsigned cap = 8 * ((want + 7) / 8);
        if (cap < gx) gx = cap;
    }
    hipLaunchKernelGGL((k_bf16_conv<NCT, ROWS, OUT>), dim3(gx, G), dim3(BF_THREADS), bf_lds_bytes_p(NCT, ROWS), s, in,
                       static_cast<const u32x4*>(wp), bias, y, Cout, slope, mask, mask_slope, B, D, H, W, Q);
}

// one persistent block per CU over all chunks; columns are cut into depth segments until there are ~4 tasks per block
// (a segment start re-fetches two planes: 1 / seg_len overhead)
BwbTasks bwb_tasks(int Q, int B, int D, int H, int W, int& NBLK) {
    BwbTasks tk;
    tk.nd = (D + BWB_TD - 1) / BWB_TD; tk.nh = (H + BWB_TH - 1) / BWB_TH; tk.nw = (W + BWB_TW - 1) / BWB_TW;
    tk.ncol = B * tk.nh * tk.nw;
    const int nb = bwb_cus() / Q > 0 ? bwb_cus() / Q : 1;
    int nseg = (4 * nb + tk.ncol - 1) / tk.ncol;
    if (nseg > tk.nd) nseg = tk.nd;
    if (nseg < 1) nseg = 1;
    tk.seg_len = (tk.nd + nseg - 1) / nseg;
    tk.nseg = (tk.nd + tk.seg_len - 1) / tk.seg_len;
    const long long ntask = (long long)tk.ncol * tk.nseg;
    NBLK = (int)(nb < ntask ? nb : ntask);
    return tk;
}

}  // namespace

extern "C" {

int vxm_bf16_to_blocked(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, void* out, int Cblk,
                        int B, int64_t V, void* stream) {
    VXM_REQUIRE(x0 && out && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_bf16_to_blocked: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && V > 0 && C0 > 0 && C1 >= 0 && Cblk % 8 == 0 && Cblk >= C0 + C1 && bf_al16(out), VXM_ERR_BAD_SHAPE,
                "vxm_bf16_to_blocked: bad shape (C0=%d C1=%d blocked channels=%d)", C0, C1, Cblk);
    const int CB = Cblk / 8;
    hipLaunchKernelGGL(k_bf16_to_blocked, dim3(vxm_blocks(V * CB, 256), B), dim3(256), 0, VXM_STREAM(stream), x0, C0, (long long)x0_bstride, x1, C1,
                       (long long)x1_bstride, static_cast<u32x4*>(out), CB, (long long)V);
    return vxm_check_launch("vxm_bf16_to_blocked");
}

int vxm_bf16_from_blocked(const void* x, int Cblk, float* out, int C, int B, int64_t V, void* stream) {
    VXM_REQUIRE(x && out, VXM_ERR_NULL_POINTER, "vxm_bf16_from_blocked: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && V > 0 && C > 0 && Cblk % 8 == 0 && C <= Cblk && bf_al16(x), VXM_ERR_BAD_SHAPE, "vxm_bf16_from_blocked: bad shape");
    hipLaunchKernelGGL(k_bf16_from_blocked, dim3(vxm_blocks(V * ((C + 7) / 8), 256), B), dim3(256), 0, VXM_STREAM(stream),
                       static_cast<const u32x4*>(x), Cblk / 8, out, C, (long long)V);
    return vxm_check_launch("vxm_bf16_from_blocked");
}

size_t vxm_bf16_conv_packed_bytes(int InC, int OutC) {
    if (InC <= 0 || OutC <= 0) return 0;
    const int NCT = bf_nct(OutC), Q = (InC + 15) / 16, G = (OutC + 16 * NCT - 1) / (16 * NCT);
    return (size_t)G * Q * bf_wchunk(NCT) * 16;
}

int vxm_bf16_conv_pack_weights(const float* w, int Cw_in, int Cw_out, int ci_lo, int ci_n, int transpose_flip, void* wpacked, void* stream) {
    VXM_REQUIRE(w && wpacked, VXM_ERR_NULL_POINTER, "vxm_bf16_conv_pack_weights: null pointer");
    VXM_REQUIRE(Cw_in > 0 && Cw_out > 0 && ci_lo >= 0 && ci_n > 0 && ci_lo + ci_n <= Cw_in && bf_al16(wpacked), VXM_ERR_BAD_SHAPE,
                "vxm_bf16_conv_pack_weights: channel range [%d, %d) of %d", ci_lo, ci_lo + ci_n, Cw_in);
    const int InC = transpose_flip ? Cw_out : ci_n, OutC = transpose_flip ? ci_n : Cw_out;
    const int NCT = bf_nct(OutC), Q = (InC + 15) / 16;
    const size_t words = vxm_bf16_conv_packed_bytes(InC, OutC) / 16;
    hipLaunchKernelGGL(k_bf16_pack_weights, dim3(vxm_blocks((long long)words, 256)), dim3(256), 0, VXM_STREAM(stream), w, static_cast<u32x4*>(wpacked),
                       Cw_in, ci_lo, transpose_flip, InC, OutC, NCT, Q, words);
    return vxm_check_launch("vxm_bf16_conv_pack_weights");
}

int vxm_bf16_conv_pack_weights_batch(const VxmBf16PackJob* jobs, int n_jobs, void* stream) {
    VXM_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0), VXM_ERR_NULL_POINTER, "vxm_bf16_conv_pack_weights_batch: null job table");
    for (int j = 0; j < n_jobs; ++j) {
        const VxmBf16PackJob& a = jobs[j];
        VXM_REQUIRE(a.w && a.wpacked, VXM_ERR_NULL_POINTER, "vxm_bf16_conv_pack_weights_batch: job %d: null pointer", j);
        VXM_REQUIRE(a.Cw_in > 0 && a.Cw_out > 0 && a.ci_lo >= 0 && a.ci_n > 0 && a.ci_lo + a.ci_n <= a.Cw_in, VXM_ERR_BAD_SHAPE,
                    "vxm_bf16_conv_pack_weights_batch: job %d: channel range [%d, %d) of %d", j, a.ci_lo, a.ci_lo + a.ci_n, a.Cw_in);
        VXM_REQUIRE(bf_al16(a.wpacked), VXM_ERR_BAD_SHAPE, "vxm_bf16_conv_pack_weights_batch: job %d: 16-byte alignment", j);
    }
    for (int j0 = 0; j0 < n_jobs; j0 += BF_PACK_JOBS) {
        BfPackBatch batch;
        batch.n = n_jobs - j0 < BF_PACK_JOBS ? n_jobs - j0 : BF_PACK_JOBS;
        unsigned blocks = 0;
        for (int j = 0; j < batch.n; ++j) {
            const VxmBf16PackJob& a = jobs[j0 + j];
            const int InC = a.transpose_flip ? a.Cw_out : a.ci_n, OutC = a.transpose_flip ? a.ci_n : a.Cw_out;
            const size_t words = vxm_bf16_conv_packed_bytes(InC, OutC) / 16;
            batch.job[j] = {a.w, static_cast<u32x4*>(a.wpacked), a.Cw_in, a.ci_lo, a.transpose_flip ? 1 : 0, InC, OutC, bf_nct(OutC), (InC + 15) / 16,
                            blocks, (unsigned)words};
            blocks += (unsigned)((words + 255) / 256);
        }
        hipLaunchKernelGGL(k_bf16_pack_weights_batch, dim3(blocks), dim3(256), 0, VXM_STREAM(stream), batch);
    }
    return vxm_check_launch("vxm_bf16_conv_pack_weights_batch");
}

int vxm_bf16_conv_fwd(const void* x0, int C0, int x0_up, const void* x1, int C1, const void* wpacked, const float* bias, void* y, int Cout,
                      int out_planar_f32, float leaky_slope, const void* mask, float mask_slope, int B, int D, int H, int W, void* stream) {
    VXM_REQUIRE(x0 && wpacked && y && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_bf16_conv_fwd: null pointer");
    if (int e = bf_check("vxm_bf16_conv_fwd", C0, C1, x0_up, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(bf_al16(x0) && bf_al16(x1) && bf_al16(wpacked) && bf_al16(y) && bf_al16(mask), VXM_ERR_BAD_SHAPE, "vxm_bf16_conv_fwd: 16-byte alignment");
    VXM_REQUIRE((out_planar_f32 & 1) ? (Cout <= 4 && !mask) : (Cout % 16 == 0), VXM_ERR_BAD_SHAPE,
                "vxm_bf16_conv_fwd: %d output channels (blocked outputs: multiples of 16; planar fp32 outputs: at most 4, no mask)", Cout);
    const BfIn in = {x0, x1, C0 / 8, C1 / 8, x0_up ? 1 : 0, (out_planar_f32 & 2) ? 1 : 0};
    out_planar_f32 &= 1;
    hipStream_t s = VXM_STREAM(stream);
    if (out_planar_f32) bf_launch_conv<1, 8, 1>(in, wpacked, bias, y, Cout, leaky_slope, nullptr, 1.0f, B, D, H, W, s);
    else if (bf_nct(Cout) == 1) bf_launch_conv<1, 8, 0>(in, wpacked, bias, y, Cout, leaky_slope, mask, mask_slope, B, D, H, W, s);
    else bf_launch_conv<2, 6, 0>(in, wpacked, bias, y, Cout, leaky_slope, mask, mask_slope, B, D, H, W, s);
    return vxm_check_launch("vxm_bf16_conv_fwd");
}

int vxm_bf16_conv_bwd_data_up(const void* dz, int Cdz, const void* wpacked, void* dx_low, int Cout, const void* mask_low, float mask_slope, int B,
                              int D, int H, int W, void* stream) {
    VXM_REQUIRE(dz && wpacked && dx_low, VXM_ERR_NULL_POINTER, "vxm_bf16_conv_bwd_data_up: null pointer");
    if (int e = bf_check("vxm_bf16_conv_bwd_data_up", Cdz, 0, 1, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(Cout % 16 == 0, VXM_ERR_BAD_SHAPE, "vxm_bf16_conv_bwd_data_up: %d output channels (multiples of 16)", Cout);
    VXM_REQUIRE(bf_al16(dz) && bf_al16(wpacked) && bf_al16(dx_low) && bf_al16(mask_low), VXM_ERR_BAD_SHAPE, "vxm_bf16_conv_bwd_data_up: 16-byte alignment");
    const BfIn in = {dz, nullptr, Cdz / 8, 0, 0, 0};
    hipStream_t s = VXM_STREAM(stream);
    if (bf_nct(Cout) == 1) bf_launch_conv<1, 8, 2>(in, wpacked, nullptr, dx_low, Cout, 1.0f, mask_low, mask_slope, B, D, H, W, s);
    else bf_launch_conv<2, 6, 2>(in, wpacked, nullptr, dx_low, Cout, 1.0f, mask_low, mask_slope, B, D, H, W, s);
    return vxm_check_launch("vxm_bf16_conv_bwd_data_up");
}

size_t vxm_bf16_conv_bwd_weight_workspace_bytes(int Cin, int Cout, int B, int D, int H, int W) {
    if (Cin <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const int Q = (Cin + 15) / 16, NCO = (Cout + 15) / 16;
    int NBLK = 1;
    (void)bwb_tasks(Q, B, D, H, W, NBLK);
    return (size_t)NBLK * Q * BWB_TD * 28 * (16 * NCO) * 16 * sizeof(float);
}

int vxm_bf16_conv_bwd_weight(const void* x0, int C0, int x0_up, const void* x1, int C1, const void* dz, int Cdz, float* gw, int Cin_w, int Cout_w,
                             float* gb, void* work, size_t work_bytes, int B, int D, int H, int W, void* stream) {
    VXM_REQUIRE(x0 && dz && gw && work && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_bf16_conv_bwd_weight: null pointer");
    if (int e = bf_check("vxm_bf16_conv_bwd_weight", C0, C1, x0_up, Cdz, B, D, H, W)) return e;
    VXM_REQUIRE(Cdz % 16 == 0 && Cdz <= 32 && Cout_w > 0 && Cout_w <= Cdz && Cin_w > 0 && Cin_w <= C0 + C1, VXM_ERR_BAD_SHAPE,
                "vxm_bf16_conv_bwd_weight: dz carries %d channels (16 or 32), weight is [%d][%d]", Cdz, Cout_w, Cin_w);
    VXM_REQUIRE(bf_al16(x0) && bf_al16(x1) && bf_al16(dz), VXM_ERR_BAD_SHAPE, "vxm_bf16_conv_bwd_weight: 16-byte alignment");
    const int Q = (C0 + C1) / 16, NCO = Cdz / 16;
    int NBLK = 1;
    const BwbTasks tk = bwb_tasks(Q, B, D, H, W, NBLK);
    VXM_REQUIRE(work_bytes >= (size_t)NBLK * Q * BWB_TD * 28 * (16 * NCO) * 16 * sizeof(float), VXM_ERR_BAD_SHAPE,
                "vxm_bf16_conv_bwd_weight: workspace too small");
    const BfIn in = {x0, x1, C0 / 8, C1 / 8, x0_up ? 1 : 0, 0};
    hipStream_t s = VXM_STREAM(stream);
    float* part = static_cast<float*>(work);
    auto launch = [&](auto kern, int lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(NBLK * Q), dim3(bwb_threads(NCO)), lds, s, in, dz, part, D, H, W, NBLK, tk);
    };
    if (NCO == 1) launch(k_bf16_conv_bwd_weight<1>, bwb_lds_bytes(1));
    else launch(k_bf16_conv_bwd_weight<2>, bwb_lds_bytes(2));
    const int n = 16 * NCO * 16 * Q * 28;
    hipLaunchKernelGGL(k_bf16_reduce_partials, dim3(vxm_blocks(n, 64)), dim3(1024), 0, s, part, gw, gb, Cin_w, Cout_w, Q, NCO, NBLK);
    return vxm_check_launch("vxm_bf16_conv_bwd_weight");
}

int vxm_bf16_maxpool2_fwd(const void* x, void* y, int B, int C, int D, int H, int W, void* stream) {
    VXM_REQUIRE(x && y, VXM_ERR_NULL_POINTER, "vxm_bf16_maxpool2_fwd: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && C % 8 == 0 && D >= 2 && H >= 2 && W >= 2, VXM_ERR_BAD_SHAPE, "vxm_bf16_maxpool2_fwd: bad shape");
    const long long n = (long long)(D / 2) * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(k_bf16_maxpool2_fwd, dim3(vxm_blocks(n, 256), B), dim3(256), 0, VXM_STREAM(stream), static_cast<const u32x4*>(x),
                       static_cast<u32x4*>(y), C / 8, D, H, W);
    return vxm_check_launch("vxm_bf16_maxpool2_fwd");
}

int vxm_bf16_maxpool2_bwd(const void* x, const void* gpool, const void* gskip, void* dz, float slope, int B, int C, int D, int H, int W, void* stream) {
    VXM_REQUIRE(x && gpool && dz, VXM_ERR_NULL_POINTER, "vxm_bf16_maxpool2_bwd: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && C % 8 == 0 && D >= 2 && H >= 2 && W >= 2 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, VXM_ERR_BAD_SHAPE,
                "vxm_bf16_maxpool2_bwd: bad shape");
    const long long n = (long long)(D / 2) * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(k_bf16_maxpool2_bwd, dim3(vxm_blocks(n, 256), B), dim3(256), 0, VXM_STREAM(stream), static_cast<const u32x4*>(x),
                       static_cast<const u32x4*>(gpool), static_cast<const u32x4*>(gskip), static_cast<u32x4*>(dz), slope, C / 8, D, H, W);
    return vxm_check_launch("vxm_bf16_maxpool2_bwd");
}

int vxm_bf16_upsample2_bwd(const void* g, const void* y, void* dz, float slope, int B, int C, int Dl, int Hl, int Wl, void* stream) {
    VXM_REQUIRE(g && dz, VXM_ERR_NULL_POINTER, "vxm_bf16_upsample2_bwd: null pointer");
    VXM_REQUIRE(B > 0 && B <= 65535 && C > 0 && C % 8 == 0 && Dl > 0 && Hl > 0 && Wl > 0, VXM_ERR_BAD_SHAPE, "vxm_bf16_upsample2_bwd: bad shape");
    const long long n = (long long)Dl * Hl * Wl * (C / 8);
    hipLaunchKernelGGL(k_bf16_upsample2_bwd, dim3(vxm_blocks(n, 256), B), dim3(256), 0, VXM_STREAM(stream), static_cast<const u32x4*>(g),
                       static_cast<const u32x4*>(y), static_cast<u32x4*>(dz), slope, C / 8, Dl, Hl, Wl);
    return vxm_check_launch("vxm_bf16_upsample2_bwd");
}

int vxm_bf16_lrelu_bwd(const void* g, const void* y, void* dz, float slope, int64_t n_elems, void* stream) {
    VXM_REQUIRE(g && y && dz, VXM_ERR_NULL_POINTER, "vxm_bf16_lrelu_bwd: null pointer");
    VXM_REQUIRE(n_elems > 0 && n_elems % 8 == 0, VXM_ERR_BAD_SHAPE, "vxm_bf16_lrelu_bwd: element count must be a multiple of 8");
    const long long n = n_elems / 8;
    const unsigned blocks = (unsigned)(n / 256 + 1 > 8192 ? 8192 : n / 256 + 1);
    hipLaunchKernelGGL(k_bf16_lrelu_bwd, dim3(blocks), dim3(256), 0, VXM_STREAM(stream), static_cast<const u32x4*>(g), static_cast<const u32x4*>(y),
                       static_cast<u32x4*>(dz), slope, n);
    return vxm_check_launch("vxm_bf16_lrelu_bwd");
}

}  // extern "C"
