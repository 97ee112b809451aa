// cat([upsample2(x0), x1]) layers of the VxmDense U-Net on the split engine, COLLAPSED: the upsampled segment at low-resolution cost.
//
// Replaces, for the decoder / "remaining" blocks that read cat([Upsample(2, 'nearest')(x0), skip]):
//   voxelmorph/torch/networks.py:133-138  x = upsample(x); x = cat([x, x_history.pop()])
//   voxelmorph/torch/networks.py:299-305  ConvBlock = Conv3d(k3,s1,p1) + LeakyReLU(0.2)   (forward)
// i.e. upsample_nearest3d + cat + convolution + leaky_relu in one launch (round 1-3: k_conv3d_k3_t8u on the fp32 matrix pipe).
//
// Algebra ("upsampled segment at low-resolution cost").  Per axis, a 3-tap convolution over a nearest-x2
// upsampled signal touches only two low-resolution inputs: output o = 2 m + p reads (m-1, m, m) for p = 0 and (m, m, m+1) for p = 1.
// Pre-summing the taps that share an input gives ONE 2 x 2 x 2 kernel per output parity class (pd, ph, pw):
//     y[2 m + p] += sum_{j in {0,1}^3} Wc[p][j] xl[m - 1 + p + j],   Wc[p][j] = sum of the w[k] with k in S(p_a, j_a) per axis a,
//     S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}
// -- 8 instead of 27 multiply-adds per input channel, x0 read at its own resolution, borders unchanged (low-res -1 / M IS the padding).
// The skip segment x1 is a plain 27-tap convolution accumulated into the same outputs.
//
// Mapping.  An MFMA's 16 columns share one weight fragment, so they must share the parity class: WAVE = CLASS.  A block of 8 waves owns
// an 8 x 4 x 32 output tile (low-res 4 x 2 x 16); wave (pd, ph, pw) owns its 8 rows (d = 2 ld + pd, h = 2 lh + ph) of 16 voxels
// w = 2 n + pw.  M = 16 output channels (x NCT tiles), K = 32 = four units of 8 input channels:
//   * upsampled segment: unit = (jh, jw) of one jd, 8 channels -> K-step = jd: two FULL K-steps per 8 channels (K = 64, no padding).
//     The class-specific weight fragments (8 classes x 2 x NP x NCT words per 8 channels: 64 KB per 8 channels for 32 outputs) do not
//     go through LDS: a wave needs only ITS class, and loads it straight from the packed operator (L2-resident) one K-step ahead.
//   * skip segment: the 27 taps of 8 channels as 7 K-steps (27 of 28 unit slots used), weights of the chunk in LDS (shared by all
//     classes).  Its full-resolution haloed tile is stored DE-INTERLEAVED by W parity (word = plane, row, parity, w / 2), so that the
//     stride-2 columns a class reads (w = 2 n + pw + kw - 1) are consecutive 16-byte words: conflict-free ds_read_b128.
// Lane groups that are served together by a ds_read_b128 (kg 0 with 1, kg 2 with 3) read words that are congruent mod 16 (they differ by
// whole rows / planes, whose strides are multiples of 16 words) -- the rule found for conv_s3.hip.
//
// Staging: the upsampled segment is staged in SUPER-CHUNKS of CBU 8-channel blocks (its haloed low-res tile is small: 6 x 4 x 18 voxels
// per block), the skip segment in 8-channel chunks (10 x 6 x 34 voxels): both fill the same 4 x 8 staging registers per thread, so a
// 32 + 16 channel layer is three stages per tile.  Persistent blocks, the next stage in flight in registers under the MFMAs of this one,
// chunk 0 of the block's next tile under the last stage -- as k_s3_conv.
//
// Piece schemes (s3_pieces.h): NP = 2 (fp16 x 2, three products) scales every stage by a power of two from the stage's largest
// magnitude.  Here the accumulators live for the whole tile (8 rows x NCT x 4 registers leave no room for a second set): the scale is a
// RUNNING one -- a stage whose maximum exceeds every earlier one first rescales the accumulators down (exact), a later stage with a
// smaller maximum reuses the running scale (its error floor is then 2^-40 of the tile's running maximum instead of its own).
#include "conv_common.h"
#include "s3_pieces.h"

// developer timing experiments (results wrong): compiled in only with -DVXM_S3_EXP (tools/build_exp.sh), selected per launch by VXM_S3_DBG
#ifdef VXM_S3_EXP
#define SU_DBG(dbg, bit) (((dbg) & (bit)) != 0)
#else
#define SU_DBG(dbg, bit) false
#endif

namespace {

constexpr int SU_THREADS = 512, SU_NI = 4;
constexpr int SU_UP_ROW = 32, SU_UP_PLANE = 4 * SU_UP_ROW, SU_UP_CB = 6 * SU_UP_PLANE;          // low-res haloed tile of one block: [6][4][32 (18 used)] words
constexpr int SU_UP_SLOTS = 6 * 4 * 18;                                                        // 432
constexpr int SU_SK_HALF = 20, SU_SK_ROW = 2 * SU_SK_HALF, SU_SK_PLANE = 6 * SU_SK_ROW;         // full-res haloed tile: [10][6][2 parities][20 (17 used)] words
constexpr int SU_SK_SLOTS = 10 * 6 * 34;                                                       // 2040

// unit (kd, kh, kw) lane group kg multiplies in skip K-step s: pair slot 2 s + (kg >> 1), member kg & 1.  Members of a pair differ by
// whole planes (kd 0 | 1) or two rows (kh 0 | 2): congruent mod 16 words.  The last two pairs are (2,1,0) | (2,1,2) -- one word apart,
// 6 instead of 4 LDS cycles -- and (2,1,1) | zero weights.
struct SuUnit { int kd, kh, kw, valid; };
__host__ __device__ constexpr SuUnit su_skip_unit(int s, int kg) {
    const int ps = 2 * s + (kg >> 1), mem = kg & 1;
    if (ps < 9) return SuUnit{mem, ps / 3, ps % 3, 1};
    if (ps < 12) return SuUnit{2, mem ? 2 : 0, ps - 9, 1};
    if (ps == 12) return SuUnit{2, 1, mem ? 2 : 0, 1};
    return SuUnit{2, 1, 1, mem == 0 ? 1 : 0};
}

template <int NCT, int NP>
struct SuCfg {
    static constexpr int CBU = NP == 2 ? 4 : 2;                                                // 8-channel blocks of an upsampled super-chunk
    static constexpr int XWORDS = CBU * SU_UP_CB > 10 * SU_SK_PLANE ? CBU * SU_UP_CB : 10 * SU_SK_PLANE;
    static constexpr int WSK = 7 * NP * NCT * 64;                                              // skip-chunk weights [s][piece][ct][lane]
    static constexpr int UPW = 8 * 2 * NP * NCT * 64;                                          // upsampled-segment weights of one 8-channel block [class][jd][piece][ct][lane]
    static constexpr int LDS_BYTES = (NP * XWORDS + WSK) * 16 + 64;
    static_assert(CBU * SU_UP_SLOTS <= SU_NI * SU_THREADS && SU_SK_SLOTS <= SU_NI * SU_THREADS, "staging slots");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// wp: [G]{[Q0 blocks][class 8][jd 2][piece NP][NCT][64 lanes], [Q1 chunks][step 7][piece NP][NCT][64 lanes]} 16-byte words, then one trailer
// word {1 / weight scale, weight scale, scratch, -} (k_s3u_pack).  y[b][o][d][h][w], Cout channels.
template <int NCT, int NP>
__global__ void __launch_bounds__(SU_THREADS, 2)
k_s3u_conv(const float* __restrict__ x0, long long bs0, int C0, const float* __restrict__ x1, long long bs1, int C1, const u32x4* __restrict__ wp,
           const float* __restrict__ bias, float* __restrict__ y, long long y_bs, int Cout, float act_slope, int B, int D, int H, int W, int lay, int dbg,
           unsigned char* __restrict__ signs, long long signs_bs) {
    using C = SuCfg<NCT, NP>;
    using P = S3P<NP>;
    VXM_DYN_SMEM(u32x4, smem);
    constexpr int XWORDS = C::XWORDS, WSK = C::WSK, UPW = C::UPW, CBU = C::CBU;
    u32x4* const Xs = smem;                                  // [NP][XWORDS]
    u32x4* const Ws = smem + NP * XWORDS;                    // skip-chunk weights
    float* const Wm = reinterpret_cast<float*>(smem + NP * XWORDS + WSK);
    const int tid = threadIdx.x, tid_ = tid, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, n = lane & 15;
    const int pd = wave >> 2, ph = (wave >> 1) & 1, pw = wave & 1;       // this wave's parity class (wave-uniform)

    const int nw = (W + 31) / 32, nh = (H + 3) / 4, nd = (D + 7) / 8;
    const int ntiles = B * nd * nh * nw;
    int t_lo, t_hi, t_step;
    if (ntiles >= 64) {
        const int x = blockIdx.x & 7;
        t_lo = (int)((long long)ntiles * x / 8) + (int)(blockIdx.x >> 3); t_hi = (int)((long long)ntiles * (x + 1) / 8); t_step = (int)(gridDim.x >> 3);
    } else {
        t_lo = blockIdx.x; t_hi = ntiles; t_step = gridDim.x;
    }
    const int g = blockIdx.y;
    const int V = D * H * W, Dl = D >> 1, Hl = H >> 1, Wl = W >> 1, V0 = Dl * Hl * Wl;
    const int Q0 = C0 >> 3, Q1 = C1 >> 3, NU = (Q0 + CBU - 1) / CBU, NST = NU + Q1;
    const u32x4* const wg = wp + (size_t)g * ((size_t)Q0 * UPW + (size_t)Q1 * WSK);

    // ---- staging roles of this thread (tile-independent): slot i = tid + 512 j.
    //   upsampled super-chunk: i = (cb, hd 6, hh 4, hw 18) of the low-res haloed tile  -> LDS word cb 768 + hd 128 + hh 32 + hw
    //   skip chunk:            i = (hd 10, hh 6, hw 34) of the full-res haloed tile     -> LDS word hd 240 + hh 40 + (hw & 1) 20 + (hw >> 1)
    // kept as (position, LDS word) pairs; -1: no slot
    int up_pos[SU_NI], up_lw[SU_NI], sk_pos[SU_NI], sk_lw[SU_NI];
#pragma unroll
    for (int j = 0; j < SU_NI; ++j) {
        const int i = tid + SU_THREADS * j;
        {
            const int cb = i / SU_UP_SLOTS, rem = i - cb * SU_UP_SLOTS;
            const int hd = rem / 72, r2 = rem - hd * 72, hh = r2 / 18, hw = r2 - hh * 18;
            const bool ok = i < CBU * SU_UP_SLOTS;
            up_pos[j] = ok ? (cb << 16 | hd << 10 | hh << 5 | hw) : -1;
            up_lw[j] = cb * SU_UP_CB + hd * SU_UP_PLANE + hh * SU_UP_ROW + hw;
        }
        {
            const int hd = i / 204, rem = i - hd * 204, hh = rem / 34, hw = rem - hh * 34;
            const bool ok = i < SU_SK_SLOTS;
            sk_pos[j] = ok ? (hd << 10 | hh << 6 | hw) : -1;
            sk_lw[j] = hd * SU_SK_PLANE + hh * SU_SK_ROW + (hw & 1) * SU_SK_HALF + (hw >> 1);
        }
    }
    // ---- per-lane LDS word offsets of the B fragments
    // upsampled: lane group kg = 2 jw + jh reads low-res voxel (ld + pd + jd, lh + ph + jh, n + pw + jw) of the haloed tile
    const int up_base = pd * SU_UP_PLANE + (ph + (kg & 1)) * SU_UP_ROW + pw + (kg >> 1) + n;
    // skip: lane group kg reads full-res haloed voxel (2 ld + pd + kd, 2 lh + ph + kh, 2 n + pw + kw) for the unit (kd, kh, kw) of step s
    int sk_base[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const SuUnit u0 = su_skip_unit(s, 0), u1 = su_skip_unit(s, 1), u2 = su_skip_unit(s, 2), u3 = su_skip_unit(s, 3);
        const SuUnit u = kg == 0 ? u0 : kg == 1 ? u1 : kg == 2 ? u2 : u3;
        const int hw = pw + u.kw;
        sk_base[s] = (pd + u.kd) * SU_SK_PLANE + (ph + u.kh) * SU_SK_ROW + (hw & 1) * SU_SK_HALF + (hw >> 1) + n;
    }

    // ---- the staging tile (runs one stage ahead of the tile being computed, across tiles)
    int d0 = 0, h0 = 0, w0 = 0, bt = 0;
    bool live = false;
    __amdgpu_buffer_rsrc_t r0, r1;
    auto set_tile = [&](int tile) __attribute__((always_inline)) {
        live = tile < t_hi;
        const int tl = live ? tile : t_lo;
        const int tw = tl % nw; int tq = tl / nw;
        const int th = tq % nh; tq /= nh;
        const int td = tq % nd;
        bt = tq / nd;
        d0 = td * 8; h0 = th * 4; w0 = tw * 32;
        r0 = vxm_rsrc(x0 + (size_t)bt * bs0, (unsigned)C0 * (unsigned)V0 * 4u);
        r1 = vxm_rsrc(C1 ? x1 + (size_t)bt * bs1 : x0, (unsigned)C1 * (unsigned)V * 4u);
    };
    float xr[SU_NI][8];
    int voffs[SU_NI];
    auto load_stage = [&](int st) __attribute__((always_inline)) {
        const bool up = st < NU;                                 // wave-uniform
        const __amdgpu_buffer_rsrc_t r = up ? r0 : r1;
        const int Vs = up ? V0 : V;
        const int cbg = up ? st * CBU : st - NU;                 // first 8-channel block of the stage inside its segment
        const int nblk = up ? Q0 : Q1;
#pragma unroll
        for (int j = 0; j < SU_NI; ++j) {
            const int pos = up ? up_pos[j] : sk_pos[j];
            int cb, gd, gh, gw, De, He, We;
            if (up) { cb = pos >> 16; gd = (d0 >> 1) - 1 + ((pos >> 10) & 63); gh = (h0 >> 1) - 1 + ((pos >> 5) & 31); gw = (w0 >> 1) - 1 + (pos & 31); De = Dl; He = Hl; We = Wl; }
            else { cb = 0; gd = d0 - 1 + (pos >> 10); gh = h0 - 1 + ((pos >> 6) & 15); gw = w0 - 1 + (pos & 63); De = D; He = H; We = W; }
            const bool ok = live && pos >= 0 && st < NST && cbg + cb < nblk && (unsigned)gd < (unsigned)De && (unsigned)gh < (unsigned)He &&
                            (unsigned)gw < (unsigned)We;
            voffs[j] = ok ? ((cbg + cb) * 8 * Vs + (gd * He + gh) * We + gw) << 2 : VXM_OOB;
            if (SU_DBG(dbg, 1)) voffs[j] = VXM_OOB;               // timing experiment: no input reads
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[j][e] = vxm_bload(r, voffs[j], (e * Vs) << 2);
        }
    };
    auto keep_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SU_NI; ++j) asm volatile("" ::"v"(voffs[j]));
    };
    auto publish_max = [&]() __attribute__((always_inline)) {
        if constexpr (NP == 2) {
            const float m = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < SU_NI; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) f(xr[j][e]);
            });
            if (lane == 0) Wm[wave] = m;
        }
    };
    // NP = 2: the running scale of the tile.  E_run = exponent field the scale is built from; `ratio` = what the accumulators are
    // multiplied by before the MFMAs of the stage just stored (1 unless that stage raised the running maximum); inv_run = 1 / scale.
    int E_run = 15;
    float ratio = 1.0f, inv_run = 1.0f;
    auto store_stage = [&](int st, bool first) __attribute__((always_inline)) {
        const bool up = st < NU;                                 // wave-uniform
        // the skip chunk's packed weights: requested first, written to LDS after the split arithmetic has covered their latency
        constexpr int WIT = (WSK + SU_THREADS - 1) / SU_THREADS;
        u32x4 wv[WIT];
        const __amdgpu_buffer_rsrc_t rw = vxm_rsrc(reinterpret_cast<const float*>(wg + (size_t)Q0 * UPW + (size_t)(up ? 0 : st - NU) * WSK), up ? 0u : WSK * 16u);
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            wv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tid_ + SU_THREADS * it) * 16, 0, 0));
        float sc = 1.0f;
        if constexpr (NP == 2) {
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(Wm), m1 = *reinterpret_cast<const f32x4*>(Wm + 4);
            const float mx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
            int E = (int)(__float_as_uint(mx) >> 23) & 255;
            E = E < 15 ? 15 : E;
            const int E_new = first ? E : (E > E_run ? E : E_run);
            const int dE = E_new - E_run;                                                          // >= 0 unless `first`
            ratio = (first || dE == 0) ? 1.0f : (dE > 126 ? 0.0f : __uint_as_float((unsigned)(127 - dE) << 23));      // 2^(E_run - E_new) <= 1
            E_run = E_new;
            sc = __uint_as_float((unsigned)(268 - E_run) << 23);
            inv_run = __uint_as_float((unsigned)(E_run - 14) << 23);
        }
#pragma unroll
        for (int j = 0; j < SU_NI; ++j) {
            if (SU_DBG(dbg, 4)) break;                          // timing experiment: no split, no LDS writes of the tile
            unsigned pk[NP][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (NP == 3) s3_split2(xr[j][2 * e], xr[j][2 * e + 1], pk[0][e], pk[1][e], pk[2][e]);
                else s3_split2_f16(xr[j][2 * e], xr[j][2 * e + 1], sc, pk[0][e], pk[1][e]);
            }
            const int pos = up ? up_pos[j] : sk_pos[j];
            const int lw = up ? up_lw[j] : sk_lw[j];
            if (pos >= 0) {                                    // (padding slots are written too: zeros from the out-of-range loads)
#pragma unroll
                for (int p = 0; p < NP; ++p) Xs[p * XWORDS + lw] = (u32x4){pk[p][0], pk[p][1], pk[p][2], pk[p][3]};
            }
        }
        if (!up) {
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int i = tid_ + SU_THREADS * it;
                if (i < WSK) Ws[i] = wv[it];
            }
        }
    };

    set_tile(t_lo);
    load_stage(0);
    if constexpr (NP == 2) { publish_max(); __syncthreads(); }
    store_stage(0, true);
    for (int tile = t_lo; tile < t_hi; tile += t_step) {
    const int cd0 = d0, ch0 = h0, cw0 = w0, cbt = bt;           // the tile being computed
    __syncthreads();                                            // stage 0 of this tile is in LDS
    f32x4 acc[8][NCT];                                          // row r = 2 ld + lh
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float inv_fin = 1.0f;
    for (int st = 0; st < NST; ++st) {
        const bool last = st + 1 == NST;                        // wave-uniform
        if (last) set_tile(tile + t_step);
        load_stage(last ? 0 : st + 1);
        if constexpr (NP == 2) {
            if (ratio != 1.0f) {                                // wave-uniform (every thread derives it from the same maxima)
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[r][ct] *= ratio;
            }
        }
        if (SU_DBG(dbg, 2)) {                                   // timing experiment: no multiply phase
        } else if (st < NU) {
            // ---- upsampled super-chunk: per 8-channel block two K-steps (jd); this class's weight fragments come from global memory,
            // one K-step ahead (register double buffer)
            const int cbg = st * CBU, ncb = min(CBU, Q0 - cbg);
            const u32x4* wa = wg + (size_t)cbg * UPW + (size_t)wave * (2 * NP * NCT * 64) + lane;
            u32x4 a0[NP][NCT], a1[NP][NCT];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) a0[p][ct] = wa[(p * NCT + ct) * 64];
            for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
                for (int jd = 0; jd < 2; ++jd) {
                    u32x4 (&ac)[NP][NCT] = jd ? a1 : a0;
                    u32x4 (&an)[NP][NCT] = jd ? a0 : a1;
                    const bool more = jd == 0 || cb + 1 < ncb;   // wave-uniform
                    const u32x4* wn = jd == 0 ? wa + NP * NCT * 64 : wa + UPW;      // (cb, jd = 1) resp. (cb + 1, jd = 0)
                    if (more && !SU_DBG(dbg, 16)) {              // (16: timing experiment, the class's weight fragments are not fetched)
#pragma unroll
                        for (int p = 0; p < NP; ++p)
#pragma unroll
                            for (int ct = 0; ct < NCT; ++ct) an[p][ct] = wn[(p * NCT + ct) * 64];
                    }
                    const int xo = up_base + cb * SU_UP_CB + jd * SU_UP_PLANE;
                    u32x4 bf[2][NP];
#pragma unroll
                    for (int p = 0; p < NP; ++p) bf[0][p] = Xs[p * XWORDS + xo];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        if (r + 1 < 8) {
#pragma unroll
                            for (int p = 0; p < NP; ++p) bf[(r + 1) & 1][p] = Xs[p * XWORDS + xo + ((r + 1) >> 1) * SU_UP_PLANE + ((r + 1) & 1) * SU_UP_ROW];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                            for (int ct = 0; ct < NCT; ++ct) acc[r][ct] = P::mfma(ac[P::PA[t]][ct], bf[r & 1][P::PB[t]], acc[r][ct]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                wa += UPW;
            }
        } else {
            // ---- skip chunk: 7 K-steps over the 27 taps of 8 channels, weights from LDS
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                u32x4 a[NP][NCT], bf[2][NP];
#pragma unroll
                for (int p = 0; p < NP; ++p) bf[0][p] = Xs[p * XWORDS + sk_base[s]];
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) a[p][ct] = Ws[((s * NP + p) * NCT + ct) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r + 1 < 8) {
#pragma unroll
                        for (int p = 0; p < NP; ++p)
                            bf[(r + 1) & 1][p] = Xs[p * XWORDS + sk_base[s] + ((r + 1) >> 1) * 2 * SU_SK_PLANE + ((r + 1) & 1) * 2 * SU_SK_ROW];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) acc[r][ct] = P::mfma(a[P::PA[t]][ct], bf[r & 1][P::PB[t]], acc[r][ct]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        keep_offsets();
        if (last) inv_fin = inv_run;                // the scale the finished accumulators carry (store_stage below moves on to the next tile)
        publish_max();                              // of the stage in flight
        __syncthreads();                            // every wave is done reading this stage
        if (!last) {
            store_stage(st + 1, false);
            __syncthreads();
        } else if (tile + t_step < t_hi) {
            store_stage(0, true);                   // stage 0 of the next tile; the barrier at the top of the tile loop publishes it
        }
    }

    // ---- epilogue: lane (kg, n) holds, in acc[r][ct][j], channel 16 (g NCT + ct) + 4 kg + j of voxel (cd0 + 2 ld + pd, ch0 + 2 lh + ph,
    // cw0 + 2 n + pw).  bias + LeakyReLU, planar fp32 store through the sample's buffer descriptor (out-of-volume lanes and channels
    // >= Cout get an out-of-range offset).  The two classes that differ in pw write the two halves of every 8-byte pair from the same
    // block at about the same time (merged in L2).
    float unscale = 1.0f;
    if constexpr (NP == 2) {
        const size_t GW = (size_t)Q0 * UPW + (size_t)Q1 * WSK;
        const float inv_w = __uint_as_float(__builtin_amdgcn_readfirstlane((int)wp[(size_t)gridDim.y * GW].x));
        unscale = inv_fin * inv_w;
    }
    const __amdgpu_buffer_rsrc_t ry = vxm_rsrc(y + (size_t)cbt * y_bs, SU_DBG(dbg, 8) ? 0u : (unsigned)Cout * (unsigned)V * 4u);      // (8: timing experiment, nothing stored)
    const int cbase = g * NCT * 16 + kg * 4;
    const int wv_ = cw0 + 2 * n + pw;
    float bz[NCT][4];
    conv_load_bias<NCT>(bz, bias, Cout, g, kg);
    if (lay & VXM_S3_OUT_BLOCKED) {
        // channel-blocked output [Cout / 8][voxel][8] (Cout % 8 == 0): the lane's four channels are 16 contiguous bytes of the voxel's group
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int dd = cd0 + 2 * (r >> 1) + pd, hh = ch0 + 2 * (r & 1) + ph;       // wave-uniform
            if (dd < D && hh < H) {
                const int vox = (dd * H + hh) * W + wv_;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int ch = cbase + 16 * ct;
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acc[r][ct][j];
                        if constexpr (NP == 2) v *= unscale;
                        v += bz[ct][j];
                        o[j] = v > 0.0f ? v : v * act_slope;
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry,
                                                           (wv_ < W && ch < Cout) ? ((((ch >> 3) * V + vox) << 5) + ((kg & 1) << 4)) : VXM_OOB, 0, 0);
                    if (signs) {                                 // the sign tensor of y (include/vxm_hip.h): one byte for this lane's four channels
                        const unsigned nib = (o[0] > 0.0f ? 1u : 0u) | (o[1] > 0.0f ? 2u : 0u) | (o[2] > 0.0f ? 4u : 0u) | (o[3] > 0.0f ? 8u : 0u);
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(signs + (size_t)cbt * signs_bs, 0, (Cout >> 2) * V, 0x00020000);
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)nib, rs, (wv_ < W && ch < Cout) ? (ch >> 2) * V + vox : VXM_OOB, 0, 0);
                    }
                }
            }
        }
    } else {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int dd = cd0 + 2 * (r >> 1) + pd, hh = ch0 + 2 * (r & 1) + ph;       // wave-uniform
        if (dd < D && hh < H) {
            const int vox = (dd * H + hh) * W + wv_;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = cbase + 16 * ct + j;
                    float v = acc[r][ct][j];
                    if constexpr (NP == 2) v *= unscale;
                    v += bz[ct][j];
                    v = v > 0.0f ? v : v * act_slope;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, (wv_ < W && ch < Cout) ? (ch * V + vox) << 2 : VXM_OOB, 0, 0);
                }
        }
    }
    }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_s3u_conv_pc (round 6): the same operator with PRODUCER and CONSUMER waves and a double-buffered staging tile.
// Timing experiments on k_s3u_conv at 160 x 192 x 224 (SU_DBG; rem0, 1.44 ms): with the multiply phase compiled out 0.89 ms, everything but
// the multiply phase compiled out 0.30 ms, the class weights not fetched -0.04 ms -- staging and multiplying ADD: the eight class waves
// issue 32 one-dword loads per thread and stage (they stall at issue when the address unit's queue is full), multiply, meet at a barrier,
// split, meet again.  Here
//   * waves 0 .. 7 (the parity classes) only multiply and write finished tiles; waves 8 .. 11 only fetch, split and write the staged tile;
//   * the staged tile is double-buffered: a stage is 2 upsampled 8-channel blocks (48 KB) or one skip chunk (75 KB), two buffers of 75 KB;
//     the skip chunk's weights come from the packed operator in L2 like the class weights (one K-step ahead, registers), not through LDS;
//   * a staging slot is a PAIR of W neighbours (one 8-byte load per channel instead of two 4-byte ones): 16 / 40 loads per producer thread
//     and stage -- per block and tile 4 x 256 x 112 = 114 k lane-loads of 8 bytes instead of 512 x 96 = 49 k x 2 of 4 bytes, in a third of
//     the wave instructions;
//   * one barrier per stage.  Phase k: the consumers multiply stage k out of buffer k & 1; the producers split stage k + 1 (its maxima were
//     published in phase k - 1) into the other buffer, request stage k + 2, wait for it and publish its maxima.
// The running scale of the tile's accumulators follows the stages as in k_s3u_conv; a producer lane writes {ratio, 1 / scale} of stage k
// into a four-slot table.  fp16 pieces only, W and W / 2 even.
constexpr int SUP_CONS = 8, SUP_PROD = 4, SUP_THREADS = 64 * (SUP_CONS + SUP_PROD), SUP_PT = 64 * SUP_PROD;
constexpr int SUP_CBU = 2;                                                                      // 8-channel blocks of an upsampled stage
// a skip stage is one 8-channel chunk over HALF the tile's depth: output rows 4 hf .. 4 hf + 3, haloed planes 4 hf .. 4 hf + 5 (6 of the tile's 10;
// planes 4, 5 are staged by both halves) -- 648 pair slots = three rounds of 16 raw registers per producer thread instead of five
constexpr int SUP_SK_PLANES = 6;
constexpr int SUP_XWORDS = SUP_CBU * SU_UP_CB > SUP_SK_PLANES * SU_SK_PLANE ? SUP_CBU * SU_UP_CB : SUP_SK_PLANES * SU_SK_PLANE;      // 1536 words per piece and buffer
constexpr int SUP_UP_PAIRS = 10, SUP_UP_SLOTS = SUP_CBU * 6 * 4 * SUP_UP_PAIRS;                 // low-res haloed row w/2 - 2 .. w/2 + 17 as 10 pairs: 480 slots
constexpr int SUP_SK_PAIRS = 18, SUP_SK_SLOTS = SUP_SK_PLANES * 6 * SUP_SK_PAIRS;               // full-res haloed row w0 - 2 .. w0 + 33 as 18 pairs: 648 slots
constexpr int SUP_NRU = (SUP_UP_SLOTS + SUP_PT - 1) / SUP_PT, SUP_NRS = (SUP_SK_SLOTS + SUP_PT - 1) / SUP_PT;      // rounds: 2, 3
constexpr int SUP_SKB_BYTES = SUP_CONS * 7 * 64 * 4;                                             // per class wave: LDS word offsets of its B fragments in the 7 skip K-steps
constexpr int SUP_LDS_BYTES = 2 * 2 * SUP_XWORDS * 16 + 256 + SUP_SKB_BYTES;
static_assert(SUP_LDS_BYTES <= 160 * 1024, "k_s3u_conv_pc: LDS");

template <int NCT>
__global__ void __launch_bounds__(SUP_THREADS)
k_s3u_conv_pc(const float* __restrict__ x0, long long bs0, int C0, const float* __restrict__ x1, long long bs1, int C1, const u32x4* __restrict__ wp,
              const float* __restrict__ bias, float* __restrict__ y, long long y_bs, int Cout, float act_slope, int B, int D, int H, int W, int lay, int dbg,
              unsigned char* __restrict__ signs, long long signs_bs) {
    constexpr int NP = 2;
    using P = S3P<NP>;
    VXM_DYN_SMEM(u32x4, smem);
    constexpr int XWORDS = SUP_XWORDS, WSK = 7 * NP * NCT * 64, UPW = 8 * 2 * NP * NCT * 64, CBU = SUP_CBU;
    // buffer b, piece p: smem + (b * NP + p) * XWORDS.  Table (floats) behind the buffers: [0..7] wave maxima of the stage in flight
    // (slot (stage & 1) * 4 + producer wave), [8 + 2 (stage & 3)] ratio, [9 + 2 (stage & 3)] inverse scale
    float* const Tab = reinterpret_cast<float*>(smem + 2 * NP * XWORDS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = (W + 31) / 32, nh = (H + 3) / 4, nd = (D + 7) / 8;
    const int ntiles = B * nd * nh * nw;
    int t_lo, t_hi, t_step;
    if (ntiles >= 64) {
        const int x = blockIdx.x & 7;
        t_lo = (int)((long long)ntiles * x / 8) + (int)(blockIdx.x >> 3); t_hi = (int)((long long)ntiles * (x + 1) / 8); t_step = (int)(gridDim.x >> 3);
    } else {
        t_lo = blockIdx.x; t_hi = ntiles; t_step = gridDim.x;
    }
    const int g = blockIdx.y;
    const int V = D * H * W, Dl = D >> 1, Hl = H >> 1, Wl = W >> 1, V0 = Dl * Hl * Wl;
    const int Q0 = C0 >> 3, Q1 = C1 >> 3, NU = (Q0 + CBU - 1) / CBU, NST = NU + 2 * Q1;      // skip stage NU + 2 q + hf
    const u32x4* const wg = wp + (size_t)g * ((size_t)Q0 * UPW + (size_t)Q1 * WSK);
    const int my_tiles = t_lo < t_hi ? (t_hi - t_lo + t_step - 1) / t_step : 0;
    const int nstage = my_tiles * NST;                           // stages this block walks; stage k: tile t_lo + (k / NST) t_step, step k % NST
    auto tile_geom = [&](int tile, int& bt, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
        const int tw = tile % nw; int tq = tile / nw;
        const int th = tq % nh; tq /= nh;
        const int td = tq % nd;
        bt = tq / nd; d0 = td * 8; h0 = th * 4; w0 = tw * 32;
    };

    if (wave < SUP_CONS) {
        // ================================================================ consumers: wave = parity class
        // At three waves per SIMD (168 registers) the compiler selects the VGPR form of the MFMAs, whose destination may not overlap its
        // accumulator operand: the 64 accumulators wander through the register file and the multiply loops spill (304 bytes of scratch).
        // An (empty) asm with an accumulation-register operand makes it keep the AGPR form: accumulators in place in a[0:63].
#ifdef SUP_AGPR_FORM
        asm volatile("" ::: "a63");
#endif
        const int kg_ = lane >> 4, n_ = lane & 15;
        const int pd = wave >> 2, ph = (wave >> 1) & 1, pw = wave & 1;
        const int up_base = pd * SU_UP_PLANE + (ph + (kg_ & 1)) * SU_UP_ROW + pw + (kg_ >> 1) + n_;
        // the seven skip-step bases of a lane live in LDS (one ds_read_b32 per K-step): as registers they were spilled to scratch, and the
        // reload in front of every K-step waited for ALL vector memory -- the weight prefetch it had just issued included (5.9 ms)
        int* const skb = reinterpret_cast<int*>(Tab + 64) + wave * (7 * 64) + lane;
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const SuUnit u0 = su_skip_unit(s, 0), u1 = su_skip_unit(s, 1), u2 = su_skip_unit(s, 2), u3 = su_skip_unit(s, 3);
            const SuUnit u = kg_ == 0 ? u0 : kg_ == 1 ? u1 : kg_ == 2 ? u2 : u3;
            const int hw = pw + u.kw;
            skb[s * 64] = (pd + u.kd) * SU_SK_PLANE + (ph + u.kh) * SU_SK_ROW + (hw & 1) * SU_SK_HALF + (hw >> 1) + n_;
        }
        const size_t GW = (size_t)Q0 * UPW + (size_t)Q1 * WSK;
        const __amdgpu_buffer_rsrc_t rwg = vxm_rsrc(reinterpret_cast<const float*>(wg), (unsigned)(GW * 16));
        const int lane16 = lane * 16;
        const float inv_w = __uint_as_float(__builtin_amdgcn_readfirstlane((int)wp[(size_t)gridDim.y * GW].x));
        __syncthreads();                                        // (start-up: the maxima of stage 0 are published)
        __syncthreads();                                        // stage 0 is in buffer 0
        int k = 0;
        for (int tile = t_lo; tile < t_hi; tile += t_step) {
            f32x4 acc[8][NCT];                                  // row r = 2 ld + lh
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float inv_fin = 1.0f;
            for (int st = 0; st < NST; ++st, ++k) {
                const u32x4* const Xs = smem + (k & 1) * NP * XWORDS;
                const float ratio = Tab[8 + 2 * (k & 3)];
                inv_fin = Tab[9 + 2 * (k & 3)];
                {                                               // ratio < 1: this stage raised the tile's running maximum (unconditional: a branch here
                    const float rt = st > 0 ? ratio : 1.0f;     // makes the accumulators loop-carried through a phi and the allocator doubles them)
#pragma unroll
                    for (int r = 0; r < 8; ++r)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) acc[r][ct] *= rt;
                }
                if (SU_DBG(dbg, 2)) {                           // timing experiment: no multiply phase
                } else if (st < NU) {
                    // ---- upsampled stage: per 8-channel block two K-steps (jd); the class's weight fragments from L2, one K-step ahead
                    const int cbg = st * CBU, ncb = min(CBU, Q0 - cbg);
                    // (buffer loads: one descriptor in scalar registers, lane offset lane * 16, the word index in the scalar offset -- per-lane
                    // 64-bit pointers cost the consumers registers they do not have)
                    int wa = (cbg * UPW + wave * (2 * NP * NCT * 64)) * 16;              // byte offset of (cb, class, jd = 0) (wave-uniform)
                    u32x4 a0[NP][NCT], a1[NP][NCT];
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) a0[p][ct] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rwg, lane16, wa + (p * NCT + ct) * 1024, 0));
                    for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
                        for (int jd = 0; jd < 2; ++jd) {
                            u32x4 (&ac)[NP][NCT] = jd ? a1 : a0;
                            u32x4 (&an)[NP][NCT] = jd ? a0 : a1;
                            const bool more = jd == 0 || cb + 1 < ncb;   // wave-uniform
                            const int wn = jd == 0 ? wa + NP * NCT * 1024 : wa + UPW * 16;
                            if (more) {
#pragma unroll
                                for (int p = 0; p < NP; ++p)
#pragma unroll
                                    for (int ct = 0; ct < NCT; ++ct) an[p][ct] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rwg, lane16, wn + (p * NCT + ct) * 1024, 0));
                            }
                            const int xo = up_base + cb * SU_UP_CB + jd * SU_UP_PLANE;
                            u32x4 bf[2][NP];
#pragma unroll
                            for (int p = 0; p < NP; ++p) bf[0][p] = Xs[p * XWORDS + xo];
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                if (r + 1 < 8) {
#pragma unroll
                                    for (int p = 0; p < NP; ++p) bf[(r + 1) & 1][p] = Xs[p * XWORDS + xo + ((r + 1) >> 1) * SU_UP_PLANE + ((r + 1) & 1) * SU_UP_ROW];
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                                    for (int ct = 0; ct < NCT; ++ct) acc[r][ct] = P::mfma(ac[P::PA[t]][ct], bf[r & 1][P::PB[t]], acc[r][ct]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                        wa += UPW * 16;
                    }
                } else {
                    // ---- skip stage (chunk q, depth half hf): 7 K-steps over the 27 taps of 8 channels for output rows 4 hf .. 4 hf + 3; the
                    // weights of a K-step from L2, one K-step ahead
                    const int sq = st - NU, q = sq >> 1;
                    const int wk = (Q0 * UPW + q * WSK) * 16;                            // byte offset of the chunk's step 0 (wave-uniform)
                    auto skip_half = [&](auto hf_) __attribute__((always_inline)) {
                        constexpr int HF = decltype(hf_)::value;
                        u32x4 a0[NP][NCT], a1[NP][NCT];
#pragma unroll
                        for (int p = 0; p < NP; ++p)
#pragma unroll
                            for (int ct = 0; ct < NCT; ++ct) a0[p][ct] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rwg, lane16, wk + (p * NCT + ct) * 1024, 0));
#pragma unroll
                        for (int s = 0; s < 7; ++s) {
                            u32x4 (&ac)[NP][NCT] = (s & 1) ? a1 : a0;
                            u32x4 (&an)[NP][NCT] = (s & 1) ? a0 : a1;
                            __builtin_amdgcn_sched_barrier(0);      // (without it the seven steps' weight loads are all hoisted to the top and spill)
                            if (s + 1 < 7) {
#pragma unroll
                                for (int p = 0; p < NP; ++p)
#pragma unroll
                                    for (int ct = 0; ct < NCT; ++ct)
                                        an[p][ct] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rwg, lane16, wk + (((s + 1) * NP + p) * NCT + ct) * 1024, 0));
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            u32x4 bf[2][NP];
                            const int sb = skb[s * 64];
#pragma unroll
                            for (int p = 0; p < NP; ++p) bf[0][p] = Xs[p * XWORDS + sb];
#pragma unroll
                            for (int rl = 0; rl < 4; ++rl) {                             // local row rl = 2 ld' + lh of the half: accumulator row 4 HF + rl
                                if (rl + 1 < 4) {
#pragma unroll
                                    for (int p = 0; p < NP; ++p)
                                        bf[(rl + 1) & 1][p] = Xs[p * XWORDS + sb + ((rl + 1) >> 1) * 2 * SU_SK_PLANE + ((rl + 1) & 1) * 2 * SU_SK_ROW];
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                                    for (int ct = 0; ct < NCT; ++ct) acc[4 * HF + rl][ct] = P::mfma(ac[P::PA[t]][ct], bf[rl & 1][P::PB[t]], acc[4 * HF + rl][ct]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    };
                    if (sq & 1) skip_half(std::integral_constant<int, 1>{}); else skip_half(std::integral_constant<int, 0>{});
                }
                __syncthreads();                                // stage k is read, stage k + 1 is written
            }
            // ---- epilogue (as k_s3u_conv): bias + LeakyReLU, planar or channel-blocked fp32 store
            int cbt, cd0, ch0, cw0;
            tile_geom(tile, cbt, cd0, ch0, cw0);
            const float unscale = inv_fin * inv_w;
            int kg = kg_, n = n_;                               // opaque per tile: the epilogue's addresses are computed here, not hoisted out of the tile loop
            asm volatile("" : "+v"(kg), "+v"(n));               // (hoisted, they were spilled: 61 dwords of scratch)
            float bz[NCT][4];
            conv_load_bias<NCT>(bz, bias, Cout, g, kg);
            const __amdgpu_buffer_rsrc_t ry = vxm_rsrc(y + (size_t)cbt * y_bs, SU_DBG(dbg, 8) ? 0u : (unsigned)Cout * (unsigned)V * 4u);
            const int cbase = g * NCT * 16 + kg * 4;
            const int wv_ = cw0 + 2 * n + pw;
            if (lay & VXM_S3_OUT_BLOCKED) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int dd = cd0 + 2 * (r >> 1) + pd, hh = ch0 + 2 * (r & 1) + ph;       // wave-uniform
                    if (dd < D && hh < H) {
                        const int vox = (dd * H + hh) * W + wv_;
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) {
                            const int ch = cbase + 16 * ct;
                            f32x4 o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float v = acc[r][ct][j] * unscale + bz[ct][j];
                                o[j] = v > 0.0f ? v : v * act_slope;
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), ry,
                                                                   (wv_ < W && ch < Cout) ? ((((ch >> 3) * V + vox) << 5) + ((kg & 1) << 4)) : VXM_OOB, 0, 0);
                            if (signs) {                         // the sign tensor of y (include/vxm_hip.h): one byte for this lane's four channels
                                const unsigned nib = (o[0] > 0.0f ? 1u : 0u) | (o[1] > 0.0f ? 2u : 0u) | (o[2] > 0.0f ? 4u : 0u) | (o[3] > 0.0f ? 8u : 0u);
                                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(signs + (size_t)cbt * signs_bs, 0, (Cout >> 2) * V, 0x00020000);
                                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)nib, rs, (wv_ < W && ch < Cout) ? (ch >> 2) * V + vox : VXM_OOB, 0, 0);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int dd = cd0 + 2 * (r >> 1) + pd, hh = ch0 + 2 * (r & 1) + ph;       // wave-uniform
                    if (dd < D && hh < H) {
                        const int vox = (dd * H + hh) * W + wv_;
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int ch = cbase + 16 * ct + j;
                                float v = acc[r][ct][j] * unscale + bz[ct][j];
                                v = v > 0.0f ? v : v * act_slope;
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, (wv_ < W && ch < Cout) ? (ch * V + vox) << 2 : VXM_OOB, 0, 0);
                            }
                    }
                }
            }
        }
    } else {
        // ================================================================ producers
        const int pwv = wave - SUP_CONS, ptid = tid - 64 * SUP_CONS;
        __builtin_amdgcn_s_setprio(3);                           // little vector work, and every wave waits for it at the barrier (see k_s3_bww_pc)
        // staging roles (tile-independent): slot i = ptid + 256 j is a PAIR of W neighbours, 8 channels.
        //   upsampled stage: i = (cb, hd 6, hh 4, pp 10): low-res voxels hw = 2 pp - 1, 2 pp of the haloed row (hw = 0 .. 17 kept) -> LDS word cb 768 + hd 128 + hh 32 + hw
        //   skip chunk:      i = (hd 10, hh 6, pp 18):    full-res voxels hw = 2 pp - 1, 2 pp (hw = 0 .. 33 kept)             -> LDS word hd 240 + hh 40 + (hw & 1) 20 + (hw >> 1)
        // pos < 0: no slot.  lw: LDS word of the SECOND voxel of the pair (the first one: up lw - 1; skip: the other parity half)
        int up_pos[SUP_NRU], up_lw[SUP_NRU], sk_pos[SUP_NRS], sk_lw[SUP_NRS];
#pragma unroll
        for (int j = 0; j < SUP_NRU; ++j) {
            const int i = ptid + SUP_PT * j;
            const int cb = i / (6 * 4 * SUP_UP_PAIRS), rem = i - cb * (6 * 4 * SUP_UP_PAIRS);
            const int hd = rem / (4 * SUP_UP_PAIRS), r2 = rem - hd * (4 * SUP_UP_PAIRS), hh = r2 / SUP_UP_PAIRS, pp = r2 - hh * SUP_UP_PAIRS;
            up_pos[j] = i < SUP_UP_SLOTS ? (cb << 16 | hd << 10 | hh << 5 | pp) : -1;
            up_lw[j] = cb * SU_UP_CB + hd * SU_UP_PLANE + hh * SU_UP_ROW + 2 * pp;
        }
#pragma unroll
        for (int j = 0; j < SUP_NRS; ++j) {
            const int i = ptid + SUP_PT * j;
            const int hd = i / (6 * SUP_SK_PAIRS), rem = i - hd * (6 * SUP_SK_PAIRS), hh = rem / SUP_SK_PAIRS, pp = rem - hh * SUP_SK_PAIRS;
            sk_pos[j] = i < SUP_SK_SLOTS ? (hd << 10 | hh << 5 | pp) : -1;
            sk_lw[j] = hd * SU_SK_PLANE + hh * SU_SK_ROW + pp;          // word of hw = 2 pp (parity 0); hw = 2 pp - 1 (parity 1) sits at + SU_SK_HALF - 1
        }
        // TWO raw sets (set = stage & 1): a stage is requested two phases before it is split, so that no phase waits for the loads it issued itself
        float ra[2][SUP_NRS][8], rb[2][SUP_NRS][8];              // first / second voxel of the pair
        int voffs[2][SUP_NRS];
        // the raw loads of stage k (past the block's last stage: nothing fetched)
        auto load_stage = [&](auto set_, int k) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value;
            const bool any = k < nstage;
            const int ti = any ? k / NST : 0, st = k - ti * NST;
            int bt, d0, h0, w0;
            tile_geom(t_lo + ti * t_step, bt, d0, h0, w0);
            const bool up = st < NU;                             // wave-uniform
            const __amdgpu_buffer_rsrc_t r = up ? vxm_rsrc(x0 + (size_t)bt * bs0, (unsigned)C0 * (unsigned)V0 * 4u)
                                                : vxm_rsrc(C1 ? x1 + (size_t)bt * bs1 : x0, (unsigned)C1 * (unsigned)V * 4u);
            const int Vs = up ? V0 : V;
            const int cbg = up ? st * CBU : (st - NU) >> 1, nblk = up ? Q0 : Q1;
            const int dh = up ? 0 : 4 * ((st - NU) & 1);         // first haloed plane of a skip stage's depth half
#pragma unroll
            for (int j = 0; j < SUP_NRS; ++j) {
                if (j >= SUP_NRU && up) { voffs[S][j] = VXM_OOB; continue; }      // (wave-uniform: an upsampled stage has two rounds)
                const int pos = up ? up_pos[j < SUP_NRU ? j : 0] : sk_pos[j];
                int cb, gd, gh, gw, De, He, We;
                if (up) { cb = pos >> 16; gd = (d0 >> 1) - 1 + ((pos >> 10) & 63); gh = (h0 >> 1) - 1 + ((pos >> 5) & 31); gw = (w0 >> 1) - 2 + 2 * (pos & 31); De = Dl; He = Hl; We = Wl; }
                else { cb = 0; gd = d0 - 1 + dh + (pos >> 10); gh = h0 - 1 + ((pos >> 5) & 31); gw = w0 - 2 + 2 * (pos & 31); De = D; He = H; We = W; }
                // a pair starts at an even w (tiles start at multiples of 32 / 16, W and W / 2 are even): inside the volume or outside it as a whole
                const bool ok = any && pos >= 0 && cbg + cb < nblk && (unsigned)gd < (unsigned)De && (unsigned)gh < (unsigned)He && (unsigned)gw < (unsigned)We;
                voffs[S][j] = ok ? ((cbg + cb) * 8 * Vs + (gd * He + gh) * We + gw) << 2 : VXM_OOB;
                if (SU_DBG(dbg, 1)) voffs[S][j] = VXM_OOB;           // timing experiment: no input reads
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const f32x2 t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voffs[S][j], (e * Vs) << 2, 0));
                    ra[S][j][e] = t2.x; rb[S][j][e] = t2.y;
                }
            }
        };
        auto publish_max = [&](auto set_, int k) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value;
            const bool up = (k % NST) < NU;                      // wave-uniform
            const float m = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < SUP_NRS; ++j) {
                    if (j >= SUP_NRU && up) continue;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { f(ra[S][j][e]); f(rb[S][j][e]); }
                }
            });
            if (lane == 0) Tab[(k & 1) * 4 + pwv] = m;
        };
        int E_run = 15;
        // split the raw registers (stage k) into buffer k & 1 with the tile's running scale; table slot k & 3 gets {ratio, 1 / scale}
        auto store_stage = [&](auto set_, int k) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value;
            if (k >= nstage) return;                             // wave-uniform
            const int st = k % NST;
            const bool up = st < NU, first = st == 0;
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(Tab + (k & 1) * 4);
            const float mx = fmaxf(fmaxf(m4.x, m4.y), fmaxf(m4.z, m4.w));
            int E = (int)(__float_as_uint(mx) >> 23) & 255;
            E = E < 15 ? 15 : E;
            const int E_new = first ? E : (E > E_run ? E : E_run);
            const int dE = E_new - E_run;                                                          // >= 0 unless `first`
            const float ratio = (first || dE == 0) ? 1.0f : (dE > 126 ? 0.0f : __uint_as_float((unsigned)(127 - dE) << 23));
            E_run = E_new;
            const float sc = __uint_as_float((unsigned)(268 - E_run) << 23);
            if (ptid == 0) { Tab[8 + 2 * (k & 3)] = ratio; Tab[9 + 2 * (k & 3)] = __uint_as_float((unsigned)(E_run - 14) << 23); }
            if (SU_DBG(dbg, 4)) return;                          // timing experiment: no split, no LDS writes
            u32x4* const Xd = smem + (k & 1) * NP * XWORDS;
#pragma unroll
            for (int j = 0; j < SUP_NRS; ++j) {
                if (j >= SUP_NRU && up) continue;
                unsigned ka[NP][4], kb[NP][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s3_split2_f16(ra[S][j][2 * e], ra[S][j][2 * e + 1], sc, ka[0][e], ka[1][e]);
                    s3_split2_f16(rb[S][j][2 * e], rb[S][j][2 * e + 1], sc, kb[0][e], kb[1][e]);
                }
                const int pos = up ? up_pos[j < SUP_NRU ? j : 0] : sk_pos[j];
                if (pos >= 0) {                                  // (padding voxels are written too: zeros from the out-of-range loads)
                    const int pp = pos & 31;
                    if (up) {
                        const int lw = up_lw[j < SUP_NRU ? j : 0];
                        if (pp > 0) {
#pragma unroll
                            for (int p = 0; p < NP; ++p) Xd[p * XWORDS + lw - 1] = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                        }
                        if (pp < SUP_UP_PAIRS - 1) {
#pragma unroll
                            for (int p = 0; p < NP; ++p) Xd[p * XWORDS + lw] = (u32x4){kb[p][0], kb[p][1], kb[p][2], kb[p][3]};
                        }
                    } else {
                        const int lw = sk_lw[j];
                        if (pp > 0) {
#pragma unroll
                            for (int p = 0; p < NP; ++p) Xd[p * XWORDS + lw + SU_SK_HALF - 1] = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                        }
                        if (pp < SUP_SK_PAIRS - 1) {
#pragma unroll
                            for (int p = 0; p < NP; ++p) Xd[p * XWORDS + lw] = (u32x4){kb[p][0], kb[p][1], kb[p][2], kb[p][3]};
                        }
                    }
                }
            }
        };
        auto keep_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < SUP_NRS; ++j) asm volatile("" ::"v"(voffs[0][j]), "v"(voffs[1][j]));
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        load_stage(S0{}, 0);
        publish_max(S0{}, 0);
        __syncthreads();
        store_stage(S0{}, 0);
        load_stage(S1{}, 1);
        publish_max(S1{}, 1);                                  // (the one place a phase waits for its own loads)
        load_stage(S0{}, 2);
        keep_offsets();
        __syncthreads();
        // phase k: split stage k + 1 (set (k + 1) & 1; its maxima were published in phase k - 1), publish the maxima of stage k + 2 (set k & 1,
        // requested in phase k - 1), request stage k + 3 into the set the split has just freed
        auto phase = [&](auto par_, int k) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_)::value;           // = k & 1
            using SA = std::integral_constant<int, PAR>;
            using SB = std::integral_constant<int, PAR ^ 1>;
            store_stage(SB{}, k + 1);
            publish_max(SA{}, k + 2);
            load_stage(SB{}, k + 3);
            keep_offsets();
            __syncthreads();
        };
        for (int k = 0; k < nstage; k += 2) {
            phase(S0{}, k);
            if (k + 1 < nstage) phase(S1{}, k + 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_s3u_bwd_pc (round 6): BOTH backward-data products of a cat([upsample(x0), x1]) layer from ONE staging of dz.
// k_s3u_dlow (gradient of the low-resolution x0, 0.59 ms at rem0) and the adjoint onto the skip tensor (k_s3p_conv, 0.60 ms) each staged the
// same 0.88 GB dz -- haloed, 1.75 + 1.55 GB through the vector L1, which is what bounds both (section 4.6 of DESIGN.md).  Here a stage is one
// 8-channel chunk of dz over a 4 x 4 x 32 tile (6 x 6 x 34 haloed, the W-parity de-interleaved layout of the forward kernel's skip chunk),
// fetched and split by four producer waves into a double-buffered tile; the eight consumer waves multiply it twice:
//   * wave = (a_h, low-res depth) for the stride-2 4 x 4 x 4 adjoint onto x0 (the arithmetic of k_s3u_dlow: four K-steps a_w, lane group
//     = a_d, two low-res rows per wave; the four a_h partial sums of a row meet in LDS at the end of the tile, fixed order);
//   * wave = parity class for the 27-tap adjoint onto the skip tensor (the arithmetic of the forward kernel's skip segment with the
//     mirrored, transposed operator: SuPack kind 2), four rows per wave.
// 32 accumulator registers in all; the weights of a K-step come from the two packed operators in L2, one K-step ahead.  One barrier per
// stage (protocol of k_s3u_conv_pc) plus one per tile for the a_h exchange.  fp16 pieces, C0 <= 32, C1 <= 32.
constexpr int SBP_PLANES = 6, SBP_XWORDS = SBP_PLANES * SU_SK_PLANE;                              // 1440 words per piece and buffer
constexpr int SBP_VSLOTS = SBP_PLANES * 6 * 34, SBP_PSLOTS = SBP_PLANES * 6 * 18;                 // voxel slots (blocked dz) / W-pair slots (planar dz)
constexpr int SBP_NRV = (SBP_VSLOTS + SUP_PT - 1) / SUP_PT, SBP_NRP = (SBP_PSLOTS + SUP_PT - 1) / SUP_PT;      // rounds: 5 / 3
constexpr int SBP_RS_WORDS = 4 * 4 * 2 * 64;                                                      // a_h exchange: [a_h 4][row 4][ct 2][lane] 16-byte words
constexpr int SBP_LDS_BYTES = 2 * 2 * SBP_XWORDS * 16 + SBP_RS_WORDS * 16 + 256 + SUP_CONS * 7 * 4 * 4;
static_assert(SBP_LDS_BYTES <= 160 * 1024, "k_s3u_bwd_pc: LDS");

template <int NCTL, int NCTS, bool BLK>
__global__ void __launch_bounds__(SUP_THREADS)
k_s3u_bwd_pc(const float* __restrict__ dz, long long dz_bs, int Cout, const u32x4* __restrict__ wlow, float* __restrict__ gxl, long long gxl_bs, int C0,
             const float* __restrict__ mask, long long mask_bs, float mask_slope, const u32x4* __restrict__ wskip, float* __restrict__ gx1,
             long long gx1_bs, int C1, int B, int D, int H, int W, int dbg) {
    constexpr int NP = 2;
    using P = S3P<NP>;
    VXM_DYN_SMEM(u32x4, smem);
    constexpr int XWORDS = SBP_XWORDS;
    constexpr int WCH = 16 * NP * NCTL * 64, WSK = 7 * NP * NCTS * 64;       // words of one dz chunk in the two operators
    f32x4* const Rs = reinterpret_cast<f32x4*>(smem + 2 * NP * XWORDS);
    float* const Tab = reinterpret_cast<float*>(Rs + SBP_RS_WORDS);          // [0..7] wave maxima (two stage slots), [8 + 2 (k & 3)] ratio, [9 + ..] inverse scale
    int* const Skb = reinterpret_cast<int*>(Tab + 64);                        // [wave 8][step 7][kg 4]: B-fragment bases of the skip adjoint
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1, V = D * H * W, Vl = Dl * Hl * Wl;
    const int nw = (W + 31) / 32, nh = (H + 3) / 4, nd = (D + 3) / 4;
    const int ntiles = B * nd * nh * nw;
    int t_lo, t_hi, t_step;
    if (ntiles >= 64) {
        const int x = blockIdx.x & 7;
        t_lo = (int)((long long)ntiles * x / 8) + (int)(blockIdx.x >> 3); t_hi = (int)((long long)ntiles * (x + 1) / 8); t_step = (int)(gridDim.x >> 3);
    } else {
        t_lo = blockIdx.x; t_hi = ntiles; t_step = gridDim.x;
    }
    const int Q = (Cout + 7) >> 3;                               // stages of a tile = 8-channel chunks of dz
    const int my_tiles = t_lo < t_hi ? (t_hi - t_lo + t_step - 1) / t_step : 0;
    const int nstage = my_tiles * Q;
    auto tile_geom = [&](int tile, int& bt, int& d0, int& h0, int& w0) __attribute__((always_inline)) {
        const int tw = tile % nw; int tq = tile / nw;
        const int th = tq % nh; tq /= nh;
        const int td = tq % nd;
        bt = tq / nd; d0 = td * 4; h0 = th * 4; w0 = tw * 32;
    };

    if (wave < SUP_CONS) {
        // ================================================================ consumers
        const int kg_ = lane >> 4, n_ = lane & 15;
        const int kq = wave & 3, rp = wave >> 2;                 // low-resolution adjoint: this wave's a_h index and low-res depth (rows 2 rp, 2 rp + 1)
        const int pd = wave >> 2, ph = (wave >> 1) & 1, pw = wave & 1;       // skip adjoint: this wave's parity class
        // lane group kg = a_d index reads haloed voxel (2 rp + kg, 2 lh + kq, 2 n + a_w) in K-step a_w
        int xb[4];
#pragma unroll
        for (int aw = 0; aw < 4; ++aw) xb[aw] = (2 * rp + kg_) * SU_SK_PLANE + kq * SU_SK_ROW + (aw & 1) * SU_SK_HALF + (aw >> 1) + n_;
        if (lane < 28) {                                         // (s, kg) -> base of the unit lane group kg multiplies in skip K-step s, without n
            const int s = lane >> 2, kg = lane & 3;
            int base = 0;
#pragma unroll
            for (int ss = 0; ss < 7; ++ss)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const SuUnit u = su_skip_unit(ss, kk);
                    const int hw = pw + u.kw;
                    if (ss == s && kk == kg) base = (pd + u.kd) * SU_SK_PLANE + (ph + u.kh) * SU_SK_ROW + (hw & 1) * SU_SK_HALF + (hw >> 1);
                }
            Skb[wave * 28 + lane] = base;
        }
        const int* const skb = Skb + wave * 28 + kg_;
        const __amdgpu_buffer_rsrc_t rwl = vxm_rsrc(reinterpret_cast<const float*>(wlow), (unsigned)((size_t)Q * WCH * 16));
        const __amdgpu_buffer_rsrc_t rws = vxm_rsrc(reinterpret_cast<const float*>(wskip), (unsigned)((size_t)Q * WSK * 16));
        const int lane16 = lane * 16;
        const float inv_wl = __uint_as_float(__builtin_amdgcn_readfirstlane((int)wlow[(size_t)Q * WCH].x));
        const float inv_ws = __uint_as_float(__builtin_amdgcn_readfirstlane((int)wskip[(size_t)Q * WSK].x));
        __syncthreads();                                        // (start-up: the maxima of stage 0 are published)
        __syncthreads();                                        // stage 0 is in buffer 0
        int k = 0;
        constexpr int RS = NCTS == 1 ? 3 : 2;                   // ring depth of the skip adjoint's weight K-steps
        u32x4 al[4][NP][NCTL], as[RS][NP][NCTS];
        {                                                       // chunk 0's four K-steps of the low-resolution adjoint
            const int wo = (kq * 4 * NP * NCTL * 64) * 16;
#pragma unroll
            for (int aw = 0; aw < 4; ++aw)
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int ct = 0; ct < NCTL; ++ct)
                        al[aw][p][ct] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rwl, lane16, wo + ((aw * NP + p) * NCTL + ct) * 1024, 0));
        }
        for (int tile = t_lo; tile < t_hi; tile += t_step) {
            f32x4 accl[2][NCTL], accs[4][NCTS];                 // low-res rows (rp, lh) / skip rows rl = 2 ld + lh of this class
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int ct = 0; ct < NCTL; ++ct) accl[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ct = 0; ct < NCTS; ++ct) accs[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            float inv_fin = 1.0f;
            for (int q = 0; q < Q; ++q, ++k) {
                const u32x4* const Xs = smem + (k & 1) * NP * XWORDS;
                const float rt = q > 0 ? Tab[8 + 2 * (k & 3)] : 1.0f;       // < 1: this stage raised the tile's running maximum
                inv_fin = Tab[9 + 2 * (k & 3)];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int ct = 0; ct < NCTL; ++ct) accl[r][ct] *= rt;
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ct = 0; ct < NCTS; ++ct) accs[r][ct] *= rt;
                // The weights of a K-step come from L2 (~600 cycles) and a K-step is 12 MFMAs of a wave (~200 cycles): one K-step ahead leaves the
                // latency open (first version: 0.91 ms with the dz loads compiled out, for 0.42 ms of MFMAs).  So: the four K-steps of the
                // low-resolution adjoint are requested a whole stage ahead (behind the previous stage's use of the same registers), the skip
                // adjoint's run RS K-steps ahead in a ring, its first RS requested at the top of the stage.
                const int wos = q * WSK * 16;
#pragma unroll
                for (int s = 0; s < RS; ++s)
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int ct = 0; ct < NCTS; ++ct)
                            as[s][p][ct] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, wos + ((s * NP + p) * NCTS + ct) * 1024, 0));
                __builtin_amdgcn_sched_barrier(0);
                // ---- adjoint onto the low-resolution tensor: K-steps a_w = 0 .. 3 of this wave's a_h
                if (!SU_DBG(dbg, 2)) {
#pragma unroll
                    for (int aw = 0; aw < 4; ++aw) {
                        u32x4 bf[2][NP];
#pragma unroll
                        for (int lh = 0; lh < 2; ++lh)
#pragma unroll
                            for (int p = 0; p < NP; ++p) bf[lh][p] = Xs[p * XWORDS + xb[aw] + lh * 2 * SU_SK_ROW];
#pragma unroll
                        for (int lh = 0; lh < 2; ++lh)
#pragma unroll
                            for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                                for (int ct = 0; ct < NCTL; ++ct) accl[lh][ct] = P::mfma(al[aw][P::PA[t]][ct], bf[lh][P::PB[t]], accl[lh][ct]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {                                               // the next stage's (chunk (q + 1) % Q: the next tile starts over) four K-steps
                    const int qn = q + 1 < Q ? q + 1 : 0;
                    const int wo = (qn * WCH + kq * 4 * NP * NCTL * 64) * 16;
                    if (!SU_DBG(dbg, 16)) {
#pragma unroll
                        for (int aw = 0; aw < 4; ++aw)
#pragma unroll
                            for (int p = 0; p < NP; ++p)
#pragma unroll
                                for (int ct = 0; ct < NCTL; ++ct)
                                    al[aw][p][ct] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rwl, lane16, wo + ((aw * NP + p) * NCTL + ct) * 1024, 0));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- adjoint onto the skip tensor: 7 K-steps over the 27 taps
                if (!SU_DBG(dbg, 4)) {
#pragma unroll
                    for (int s = 0; s < 7; ++s) {
                        const int sb = skb[s * 4] + n_;
                        u32x4 bf[2][NP];
#pragma unroll
                        for (int p = 0; p < NP; ++p) bf[0][p] = Xs[p * XWORDS + sb];
#pragma unroll
                        for (int rl = 0; rl < 4; ++rl) {
                            if (rl + 1 < 4) {
#pragma unroll
                                for (int p = 0; p < NP; ++p)
                                    bf[(rl + 1) & 1][p] = Xs[p * XWORDS + sb + ((rl + 1) >> 1) * 2 * SU_SK_PLANE + ((rl + 1) & 1) * 2 * SU_SK_ROW];
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                                for (int ct = 0; ct < NCTS; ++ct) accs[rl][ct] = P::mfma(as[s % RS][P::PA[t]][ct], bf[rl & 1][P::PB[t]], accs[rl][ct]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (s + RS < 7 && !SU_DBG(dbg, 32)) {      // K-step s + RS into the ring slot this one has just left
#pragma unroll
                            for (int p = 0; p < NP; ++p)
#pragma unroll
                                for (int ct = 0; ct < NCTS; ++ct)
                                    as[s % RS][p][ct] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rws, lane16, wos + (((s + RS) * NP + p) * NCTS + ct) * 1024, 0));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();                                // stage k is read, stage k + 1 is written
            }
            // ---- epilogue.  Skip adjoint: lane (kg, n) holds channel 16 ct + 4 kg + j of voxel (d0 + 2 ld + pd, h0 + 2 lh + ph, w0 + 2 n + pw): planar store
            int bt, d0, h0, w0;
            tile_geom(tile, bt, d0, h0, w0);
            int kg = kg_, n = n_;                               // opaque per tile: the epilogue's addresses are computed here, not hoisted and spilled
            asm volatile("" : "+v"(kg), "+v"(n));
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int ct = 0; ct < NCTL; ++ct) Rs[((kq * 4 + 2 * rp + r) * 2 + ct) * 64 + lane] = accl[r][ct];
            {
                const float unscale = inv_fin * inv_ws;
                const __amdgpu_buffer_rsrc_t ry = vxm_rsrc(gx1 + (size_t)bt * gx1_bs, SU_DBG(dbg, 8) ? 0u : (unsigned)C1 * (unsigned)V * 4u);
                const int wv_ = w0 + 2 * n + pw;
#pragma unroll
                for (int rl = 0; rl < 4; ++rl) {
                    const int dd = d0 + 2 * (rl >> 1) + pd, hh = h0 + 2 * (rl & 1) + ph;       // wave-uniform
                    if (dd < D && hh < H) {
                        const int vox = (dd * H + hh) * W + wv_;
#pragma unroll
                        for (int ct = 0; ct < NCTS; ++ct)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int ch = 16 * ct + 4 * kg + j;
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accs[rl][ct][j] * unscale), ry, (wv_ < W && ch < C1) ? (ch * V + vox) << 2 : VXM_OOB, 0, 0);
                            }
                    }
                }
            }
            __syncthreads();                                    // the a_h partial sums of the tile are in LDS
            // low-resolution adjoint: wave w finishes (row = w >> 1 = (ld, lh), ct = w & 1): the four a_h sums in a fixed order, unscale, LeakyReLU'(mask)
            if ((wave & 1) < NCTL) {
                const int row = wave >> 1, ct = wave & 1;
                const f32x4 p0 = Rs[((0 * 4 + row) * 2 + ct) * 64 + lane], p1 = Rs[((1 * 4 + row) * 2 + ct) * 64 + lane];
                const f32x4 p2 = Rs[((2 * 4 + row) * 2 + ct) * 64 + lane], p3 = Rs[((3 * 4 + row) * 2 + ct) * 64 + lane];
                const f32x4 sum = (p0 + p1) + (p2 + p3);
                const float unscale = inv_fin * inv_wl;
                const int dd = (d0 >> 1) + (row >> 1), hh = (h0 >> 1) + (row & 1), ww = (w0 >> 1) + n;
                const __amdgpu_buffer_rsrc_t ry = vxm_rsrc(gxl + (size_t)bt * gxl_bs, (unsigned)C0 * (unsigned)Vl * 4u);
                const __amdgpu_buffer_rsrc_t rm = vxm_rsrc(mask ? mask + (size_t)bt * mask_bs : gxl, (unsigned)C0 * (unsigned)Vl * 4u);
                const bool vok = dd < Dl && hh < Hl && ww < Wl;
                const int vox = (dd * Hl + hh) * Wl + ww;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ch = ct * 16 + kg * 4 + j;
                    const int off = (vok && ch < C0) ? (ch * Vl + vox) << 2 : VXM_OOB;
                    float v = sum[j] * unscale;
                    if (mask) v *= vxm_lrelu_grad(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rm, off, 0, 0)), mask_slope);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, off, 0, 0);
                }
            }
        }
    } else {
        // ================================================================ producers (protocol of k_s3u_conv_pc)
        const int pwv = wave - SUP_CONS, ptid = tid - 64 * SUP_CONS;
        __builtin_amdgcn_s_setprio(3);
        constexpr int NR = BLK ? SBP_NRV : SBP_NRP, RW = BLK ? 8 : 16;          // rounds / raw registers of a round
        // slot i = ptid + 256 j.  Blocked dz: a voxel (hd 6, hh 6, hw 34), 8 channels = two 16-byte loads -> LDS word hd 240 + hh 40 + (hw & 1) 20 + (hw >> 1).
        // Planar dz: a PAIR of W neighbours (hd, hh, pp 18) = voxels hw = 2 pp - 1, 2 pp, one 8-byte load per channel.
        int pos[NR], lw[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int i = ptid + SUP_PT * j;
            if constexpr (BLK) {
                const int hd = i / 204, rem = i - hd * 204, hh = rem / 34, hw = rem - hh * 34;
                pos[j] = i < SBP_VSLOTS ? (hd << 10 | hh << 6 | hw) : -1;
                lw[j] = hd * SU_SK_PLANE + hh * SU_SK_ROW + (hw & 1) * SU_SK_HALF + (hw >> 1);
            } else {
                const int hd = i / 108, rem = i - hd * 108, hh = rem / 18, pp = rem - hh * 18;
                pos[j] = i < SBP_PSLOTS ? (hd << 10 | hh << 6 | pp) : -1;
                lw[j] = hd * SU_SK_PLANE + hh * SU_SK_ROW + pp;       // word of hw = 2 pp (parity 0); hw = 2 pp - 1 sits at + SU_SK_HALF - 1
            }
        }
        // TWO raw sets (set = stage & 1): a stage is requested two phases before it is split, so a phase never waits for its own loads --
        // with one set the producers' chain was split + request + WAIT + maximum in every phase, longer than the consumers' multiply phase
        // (timing experiment: 0.36 ms of the 1.11 with loads, multiplies and stores compiled out)
        float raw[2][NR][RW];
        int voffs[2][NR];
        auto load_stage = [&](auto set_, int k) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value;
            const bool any = k < nstage;
            const int ti = any ? k / Q : 0, q = k - ti * Q;
            int bt, d0, h0, w0;
            tile_geom(t_lo + ti * t_step, bt, d0, h0, w0);
            const __amdgpu_buffer_rsrc_t r = vxm_rsrc(dz + (size_t)bt * dz_bs, (unsigned)Cout * (unsigned)V * 4u);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int gd = d0 - 1 + (pos[j] >> 10), gh = h0 - 1 + ((pos[j] >> 6) & 15);
                if constexpr (BLK) {
                    const int gw = w0 - 1 + (pos[j] & 63);
                    const bool ok = any && pos[j] >= 0 && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
                    voffs[S][j] = (ok && !SU_DBG(dbg, 1)) ? (q * V + (gd * H + gh) * W + gw) << 5 : VXM_OOB;
                    const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffs[S][j], 0, 0));
                    const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voffs[S][j], 16, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { raw[S][j][e] = lo[e]; raw[S][j][4 + e] = hi[e]; }
                } else {
                    const int gw = w0 - 2 + 2 * (pos[j] & 63);
                    const bool ok = any && pos[j] >= 0 && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
                    voffs[S][j] = (ok && !SU_DBG(dbg, 1)) ? (q * 8 * V + (gd * H + gh) * W + gw) << 2 : VXM_OOB;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const f32x2 t2 = (q * 8 + e < Cout) ? __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voffs[S][j], (e * V) << 2, 0)) : (f32x2){0.f, 0.f};
                        raw[S][j][e] = t2.x; raw[S][j][8 + e] = t2.y;
                    }
                }
            }
        };
        auto publish_max = [&](auto set_, int k) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value;
            const float m = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    if constexpr (BLK) {
#pragma unroll
                        for (int e = 0; e < RW; ++e) f(raw[S][j][e]);
                    } else {                                     // the voxels a row's first / last pair drops (hw = -1, 34) stay out of the maximum:
                        const int pp = pos[j] & 63;              // the blocked staging never sees them -- same scale, same bits in both layouts
#pragma unroll
                        for (int e = 0; e < 8; ++e) { f(pp > 0 ? raw[S][j][e] : 0.0f); f(pp < 17 ? raw[S][j][8 + e] : 0.0f); }
                    }
                }
            });
            if (lane == 0) Tab[(k & 1) * 4 + pwv] = m;
        };
        int E_run = 15;
        auto store_stage = [&](auto set_, int k) __attribute__((always_inline)) {
            constexpr int S = decltype(set_)::value;
            if (k >= nstage) return;                             // wave-uniform
            const bool first = (k % Q) == 0;
            const f32x4 m4 = *reinterpret_cast<const f32x4*>(Tab + (k & 1) * 4);
            const float mx = fmaxf(fmaxf(m4.x, m4.y), fmaxf(m4.z, m4.w));
            int E = (int)(__float_as_uint(mx) >> 23) & 255;
            E = E < 15 ? 15 : E;
            const int E_new = first ? E : (E > E_run ? E : E_run);
            const int dE = E_new - E_run;
            const float ratio = (first || dE == 0) ? 1.0f : (dE > 126 ? 0.0f : __uint_as_float((unsigned)(127 - dE) << 23));
            E_run = E_new;
            const float sc = __uint_as_float((unsigned)(268 - E_run) << 23);
            if (ptid == 0) { Tab[8 + 2 * (k & 3)] = ratio; Tab[9 + 2 * (k & 3)] = __uint_as_float((unsigned)(E_run - 14) << 23); }
            u32x4* const Xd = smem + (k & 1) * NP * XWORDS;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                if (pos[j] < 0) continue;
                if constexpr (BLK) {
                    unsigned ka[NP][4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) s3_split2_f16(raw[S][j][2 * e], raw[S][j][2 * e + 1], sc, ka[0][e], ka[1][e]);
#pragma unroll
                    for (int p = 0; p < NP; ++p) Xd[p * XWORDS + lw[j]] = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                } else {
                    unsigned ka[NP][4], kb[NP][4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s3_split2_f16(raw[S][j][2 * e], raw[S][j][2 * e + 1], sc, ka[0][e], ka[1][e]);
                        s3_split2_f16(raw[S][j][8 + 2 * e], raw[S][j][8 + 2 * e + 1], sc, kb[0][e], kb[1][e]);
                    }
                    const int pp = pos[j] & 63;
                    if (pp > 0) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) Xd[p * XWORDS + lw[j] + SU_SK_HALF - 1] = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                    }
                    if (pp < 17) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) Xd[p * XWORDS + lw[j]] = (u32x4){kb[p][0], kb[p][1], kb[p][2], kb[p][3]};
                    }
                }
            }
        };
        auto keep_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NR; ++j) asm volatile("" ::"v"(voffs[0][j]), "v"(voffs[1][j]));
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        load_stage(S0{}, 0);
        publish_max(S0{}, 0);
        __syncthreads();
        store_stage(S0{}, 0);
        load_stage(S1{}, 1);
        publish_max(S1{}, 1);                                  // (the one place a phase waits for its own loads)
        load_stage(S0{}, 2);
        keep_offsets();
        __syncthreads();
        // phase k: split stage k + 1 (set (k + 1) & 1; its maxima were published in phase k - 1), publish the maxima of stage k + 2 (set k & 1,
        // requested in phase k - 1), request stage k + 3 into the set the split has just freed
        auto phase = [&](auto par_, int k) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_)::value;           // = k & 1
            using SA = std::integral_constant<int, PAR>;
            using SB = std::integral_constant<int, PAR ^ 1>;
            store_stage(SB{}, k + 1);
            publish_max(SA{}, k + 2);
            load_stage(SB{}, k + 3);
            keep_offsets();
            __syncthreads();
            if ((k + 1) % Q == 0) __syncthreads();              // the consumers' a_h exchange at the end of a tile
        };
        for (int k = 0; k < nstage; k += 2) {
            phase(S0{}, k);
            if (k + 1 < nstage) phase(S1{}, k + 1);
        }
    }
}

// ---- packed operator.  w: [Cout][C0 + C1][27] fp32 (reference layout).
struct SuPack {
    const float* w; u32x4* wp;
    int C0, C1, Cout, NCT, NP, Q0, Q1, G;
    unsigned words;                          // 16-byte words of the operator proper; the trailer follows
    int kind;                                // 0: forward operator (k_s3u_conv), 1: adjoint onto the low-resolution tensor (k_s3u_dlow),
                                             // 2: adjoint onto the skip tensor in the forward kernel's skip-chunk format (k_s3u_bwd_pc): C0 = 0, C1 = the
                                             //    layer's OUTPUT channels (dz), Cout = its skip channels; adj_lo / adj_cin: first skip channel / row length of w
    int adj_lo, adj_cin;
};
// the 8 values (8 consecutive input channels) of packed word i
__device__ __forceinline__ void su_word_values(const SuPack& jb, size_t i, float (&v)[8], int& piece) {
    const int NP = jb.NP, NCT = jb.NCT;
    const int UPW = 8 * 2 * NP * NCT * 64, WSK = 7 * NP * NCT * 64, Cin = jb.C0 + jb.C1;
    const size_t GW = (size_t)jb.Q0 * UPW + (size_t)jb.Q1 * WSK;
    const int g = (int)(i / GW);
    size_t r = i - (size_t)g * GW;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    if (r < (size_t)jb.Q0 * UPW) {
        const int cb = (int)(r / UPW); int r2 = (int)(r - (size_t)cb * UPW);
        const int cls = r2 / (2 * NP * NCT * 64); r2 -= cls * (2 * NP * NCT * 64);
        const int jd = r2 / (NP * NCT * 64); r2 -= jd * (NP * NCT * 64);
        piece = r2 / (NCT * 64); r2 -= piece * (NCT * 64);
        const int ct = r2 / 64, lane = r2 & 63, kg = lane >> 4, m = lane & 15;
        const int jh = kg & 1, jw = kg >> 1, pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
        const int o = (g * NCT + ct) * 16 + m;
        if (o >= jb.Cout) return;
        // taps of axis parity p that read low-res input j: S(0,0) = {0}, S(0,1) = {1,2}, S(1,0) = {0,1}, S(1,1) = {2}
        const int dlo = pd ? (jd ? 2 : 0) : (jd ? 1 : 0), dhi = pd ? (jd ? 2 : 1) : (jd ? 2 : 0);
        const int hlo = ph ? (jh ? 2 : 0) : (jh ? 1 : 0), hhi = ph ? (jh ? 2 : 1) : (jh ? 2 : 0);
        const int wlo = pw ? (jw ? 2 : 0) : (jw ? 1 : 0), whi = pw ? (jw ? 2 : 1) : (jw ? 2 : 0);
        for (int e = 0; e < 8; ++e) {
            const int ci = cb * 8 + e;
            if (ci >= jb.C0) continue;
            const float* wr = jb.w + ((size_t)o * Cin + ci) * 27;
            float s = 0.0f;
            for (int kd = dlo; kd <= dhi; ++kd)
                for (int kh = hlo; kh <= hhi; ++kh)
                    for (int kw = wlo; kw <= whi; ++kw) s += wr[kd * 9 + kh * 3 + kw];
            v[e] = s;
        }
    } else {
        r -= (size_t)jb.Q0 * UPW;
        const int q = (int)(r / WSK); int r2 = (int)(r - (size_t)q * WSK);
        const int s = r2 / (NP * NCT * 64); r2 -= s * (NP * NCT * 64);
        piece = r2 / (NCT * 64); r2 -= piece * (NCT * 64);
        const int ct = r2 / 64, lane = r2 & 63, kg = lane >> 4, m = lane & 15;
        const SuUnit u = su_skip_unit(s, kg);
        const int o = (g * NCT + ct) * 16 + m;
        if (!u.valid || o >= jb.Cout) return;
        for (int e = 0; e < 8; ++e) {
            const int ci = q * 8 + e;
            if (ci >= jb.C1) continue;
            if (jb.kind == 2) v[e] = jb.w[((size_t)ci * jb.adj_cin + jb.adj_lo + o) * 27 + 26 - (u.kd * 9 + u.kh * 3 + u.kw)];      // W_adj[o][ci][tap] = w[ci][lo + o][mirrored tap]
            else v[e] = jb.w[((size_t)o * Cin + jb.C0 + ci) * 27 + u.kd * 9 + u.kh * 3 + u.kw];
        }
    }
}
// Adjoint of the upsampled segment onto the LOW-resolution tensor (k_s3u_dlow): gxl[ci][m] = sum_o sum_{a in {-1,0,1,2}^3} Wt[a][ci][o] dz[o][2 m + a]
// with, per axis, Wt(-1) = w2, Wt(0) = w1 + w2, Wt(1) = w0 + w1, Wt(2) = w0 (the transpose of the collapsed forward).
// words: [G][Q = Cout / 8 chunks of dz channels][step 16 = (a_h, a_w)][piece][NCT][64 lanes]; lane (kg, m): a_d index kg, input channel
// ci = 16 (g NCT + ct) + m, the 8 values = dz channels o = 8 q + e.
__device__ __forceinline__ void su_dlow_word_values(const SuPack& jb, size_t i, float (&v)[8], int& piece) {
    const int NP = jb.NP, NCT = jb.NCT, Cin = jb.C0 + jb.C1, Q = (jb.Cout + 7) / 8;
    const int per_chunk = 16 * NP * NCT * 64;
    size_t r = i;
    int r2 = (int)(r % per_chunk); r /= per_chunk;
    const int q = (int)(r % Q), g = (int)(r / Q);
    const int step = r2 / (NP * NCT * 64); r2 -= step * (NP * NCT * 64);
    piece = r2 / (NCT * 64); r2 -= piece * (NCT * 64);
    const int ct = r2 / 64, lane = r2 & 63, kg = lane >> 4, m = lane & 15;
    const int ad = kg, ah = step >> 2, aw = step & 3;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    const int ci = (g * NCT + ct) * 16 + m;
    if (ci >= jb.C0) return;
    // taps that reach offset index t (a = t - 1): T(0) = {2}, T(1) = {1,2}, T(2) = {0,1}, T(3) = {0}
    const int dlo = ad == 0 ? 2 : (ad == 1 ? 1 : 0), dhi = ad <= 1 ? 2 : (ad == 2 ? 1 : 0);
    const int hlo = ah == 0 ? 2 : (ah == 1 ? 1 : 0), hhi = ah <= 1 ? 2 : (ah == 2 ? 1 : 0);
    const int wlo = aw == 0 ? 2 : (aw == 1 ? 1 : 0), whi = aw <= 1 ? 2 : (aw == 2 ? 1 : 0);
    for (int e = 0; e < 8; ++e) {
        const int o = q * 8 + e;
        if (o >= jb.Cout) continue;
        const float* wr = jb.w + ((size_t)o * Cin + ci) * 27;
        float s = 0.0f;
        for (int kd = dlo; kd <= dhi; ++kd)
            for (int kh = hlo; kh <= hhi; ++kh)
                for (int kw = wlo; kw <= whi; ++kw) s += wr[kd * 9 + kh * 3 + kw];
        v[e] = s;
    }
}
__device__ __forceinline__ void su_values(const SuPack& jb, size_t i, float (&v)[8], int& piece) {
    if (jb.kind == 1) su_dlow_word_values(jb, i, v, piece); else su_word_values(jb, i, v, piece);
}
// PHASE 0: largest magnitude of the packed values -> trailer.z (float bits, zeroed by the host side first).  PHASE 1: the words.
template <int PHASE>
__global__ void __launch_bounds__(256) k_s3u_pack(const SuPack jb) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float v[8];
    int piece = 0;
    if (PHASE == 0) {
        float m = 0.0f;
        if (i < jb.words) {
            su_values(jb, i, v, piece);
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, s3_finite_mag(v[e]));
        }
        m = s3_wave_max(m);
        if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(&jb.wp[jb.words]) + 2, __float_as_uint(m));
        return;
    }
    if (i >= jb.words) return;
    su_values(jb, i, v, piece);
    unsigned pk[3][4];
    if (jb.NP == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s3_split2(v[2 * e], v[2 * e + 1], pk[0][e], pk[1][e], pk[2][e]);
    } else {
        float sc, inv;
        s3_scale_of(__uint_as_float(jb.wp[jb.words].z), sc, inv);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s3_split2_f16(v[2 * e], v[2 * e + 1], sc, pk[0][e], pk[1][e]); pk[2][e] = 0u; }
        if (i == 0) {                                         // (.z is only read by the other threads)
            unsigned* t = reinterpret_cast<unsigned*>(&jb.wp[jb.words]);
            t[0] = __float_as_uint(inv); t[1] = __float_as_uint(sc);
        }
    }
    jb.wp[i] = (u32x4){pk[piece][0], pk[piece][1], pk[piece][2], pk[piece][3]};
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward-data of the upsampled segment, straight onto the LOW-resolution tensor:
//   convolution_backward (input) + upsample_nearest3d_backward + leaky_relu_backward of the decoder block in one launch
//   (autograd twins of networks.py:133-138, 299-305; rounds 1-3: k_conv3d_k3_dlow on the fp32 matrix pipe).
// gxl[ci][m] = LeakyReLU'(mask[ci][m]) sum_o sum_{a in {-1,0,1,2}^3} Wt[a][ci][o] dz[o][2 m + a]   (su_dlow_word_values): a stride-2, 4x4x4-tap
// convolution of the full-resolution dZ; the full-resolution gradient of those channels is never written.
// Implicit GEMM: M = 16 input channels (x NCT), N = 16 low-res voxels of a W row, K = 32 = four units of 8 dZ channels, unit = a_d
// (lane group), K-step = (a_h, a_w): 16 K-steps per 8 dZ channels.  dZ is staged like the forward kernel's skip segment (haloed
// 10 x 6 x 34 tile, W-parity de-interleaved), output tile = low-res 4 x 2 x 16.  With only 8 output rows per tile a wave cannot own
// rows AND reuse a weight fragment, so the block splits K: wave = (a_h, row half) multiplies its a_h's four K-steps for four output rows
// (a weight fragment serves four rows), and the four a_h partial sums of a row are combined through LDS in a fixed order
// (deterministic) before the epilogue.
template <int NCT, int NP>
struct SdCfg {
    static constexpr int XWORDS = 10 * SU_SK_PLANE;
    static constexpr int WCH = 16 * NP * NCT * 64;                                             // [step 16][piece][ct][lane]
    static constexpr int LDS_BYTES = (NP * XWORDS + WCH) * 16 + 64;
    static_assert(4 * 8 * NCT * 64 <= NP * XWORDS, "the combine buffer reuses the staging tile");
};

// BLK: dz is channel-blocked [Cout / 8][voxel][8] (Cout % 8 == 0): a staging slot is 32 contiguous bytes (two 16-byte loads)
template <int NCT, int NP, bool BLK = false>
__global__ void __launch_bounds__(SU_THREADS, 2)
k_s3u_dlow(const float* __restrict__ dz, long long dz_bs, int Cout, const u32x4* __restrict__ wp, float* __restrict__ gxl, long long gxl_bs, int C0,
           const float* __restrict__ mask, long long mask_bs, float mask_slope, int B, int D, int H, int W) {
    using C = SdCfg<NCT, NP>;
    using P = S3P<NP>;
    VXM_DYN_SMEM(u32x4, smem);
    constexpr int XWORDS = C::XWORDS, WCH = C::WCH;
    u32x4* const Xs = smem;
    u32x4* const Ws = smem + NP * XWORDS;
    float* const Wm = reinterpret_cast<float*>(smem + NP * XWORDS + WCH);
    const int tid = threadIdx.x, tid_ = tid, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 4, n = lane & 15;
    const int kq = wave & 3, rh = wave >> 2;                 // this wave's a_h index and row half (wave-uniform)

    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1, V = D * H * W, Vl = Dl * Hl * Wl;
    const int nw = (Wl + 15) / 16, nh = (Hl + 1) / 2, nd = (Dl + 3) / 4;
    const int ntiles = B * nd * nh * nw;
    int t_lo, t_hi, t_step;
    if (ntiles >= 64) {
        const int x = blockIdx.x & 7;
        t_lo = (int)((long long)ntiles * x / 8) + (int)(blockIdx.x >> 3); t_hi = (int)((long long)ntiles * (x + 1) / 8); t_step = (int)(gridDim.x >> 3);
    } else {
        t_lo = blockIdx.x; t_hi = ntiles; t_step = gridDim.x;
    }
    const int g = blockIdx.y;
    const int Q = (Cout + 7) >> 3;
    const u32x4* const wg = wp + (size_t)g * Q * WCH;

    // staging slots (as the forward kernel's skip chunk): i = tid + 512 j = (hd 10, hh 6, hw 34) -> word hd 240 + hh 40 + (hw & 1) 20 + (hw >> 1)
    int sk_pos[SU_NI], sk_lw[SU_NI];
#pragma unroll
    for (int j = 0; j < SU_NI; ++j) {
        const int i = tid + SU_THREADS * j;
        const int hd = i / 204, rem = i - hd * 204, hh = rem / 34, hw = rem - hh * 34;
        sk_pos[j] = i < SU_SK_SLOTS ? (hd << 10 | hh << 6 | hw) : -1;
        sk_lw[j] = hd * SU_SK_PLANE + hh * SU_SK_ROW + (hw & 1) * SU_SK_HALF + (hw >> 1);
    }
    // B fragments: lane group kg = a_d index reads haloed voxel (2 ld + kg, 2 lh + kq, 2 n + aw) in K-step aw of this wave's a_h
    int xb[4];
#pragma unroll
    for (int aw = 0; aw < 4; ++aw) xb[aw] = kg * SU_SK_PLANE + kq * SU_SK_ROW + (aw & 1) * SU_SK_HALF + (aw >> 1) + n + rh * 4 * SU_SK_PLANE;

    int d0 = 0, h0 = 0, w0 = 0, bt = 0;                       // full-resolution origin of the staging tile
    bool live = false;
    __amdgpu_buffer_rsrc_t rz;
    auto set_tile = [&](int tile) __attribute__((always_inline)) {
        live = tile < t_hi;
        const int tl = live ? tile : t_lo;
        const int tw = tl % nw; int tq = tl / nw;
        const int th = tq % nh; tq /= nh;
        const int td = tq % nd;
        bt = tq / nd;
        d0 = td * 8; h0 = th * 4; w0 = tw * 32;
        rz = vxm_rsrc(dz + (size_t)bt * dz_bs, (unsigned)Cout * (unsigned)V * 4u);
    };
    float xr[SU_NI][8];
    int voffs[SU_NI];
    auto load_stage = [&](int q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SU_NI; ++j) {
            const int pos = sk_pos[j];
            const int gd = d0 - 1 + (pos >> 10), gh = h0 - 1 + ((pos >> 6) & 15), gw = w0 - 1 + (pos & 63);
            const bool ok = live && pos >= 0 && q < Q && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            if constexpr (BLK) {
                voffs[j] = ok ? (q * V + (gd * H + gh) * W + gw) << 5 : VXM_OOB;
                const f32x4 lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, voffs[j], 0, 0));
                const f32x4 hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, voffs[j], 16, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) { xr[j][e] = lo[e]; xr[j][4 + e] = hi[e]; }
                continue;
            }
            voffs[j] = ok ? (q * 8 * V + (gd * H + gh) * W + gw) << 2 : VXM_OOB;
#pragma unroll
            for (int e = 0; e < 8; ++e) xr[j][e] = (q * 8 + e < Cout) ? vxm_bload(rz, voffs[j], (e * V) << 2) : 0.0f;       // (wave-uniform test: Cout % 8 != 0)
        }
    };
    auto keep_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < SU_NI; ++j) asm volatile("" ::"v"(voffs[j]));
    };
    auto publish_max = [&]() __attribute__((always_inline)) {
        if constexpr (NP == 2) {
            const float m = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < SU_NI; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) f(xr[j][e]);
            });
            if (lane == 0) Wm[wave] = m;
        }
    };
    int E_run = 15;
    float ratio = 1.0f, inv_run = 1.0f;
    auto store_stage = [&](int q, bool first) __attribute__((always_inline)) {
        constexpr int WIT = (WCH + SU_THREADS - 1) / SU_THREADS;
        u32x4 wv[WIT];
        const __amdgpu_buffer_rsrc_t rw = vxm_rsrc(reinterpret_cast<const float*>(wg + (size_t)q * WCH), WCH * 16u);
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            wv[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (tid_ + SU_THREADS * it) * 16, 0, 0));
        float sc = 1.0f;
        if constexpr (NP == 2) {
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(Wm), m1 = *reinterpret_cast<const f32x4*>(Wm + 4);
            const float mx = fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w)));
            int E = (int)(__float_as_uint(mx) >> 23) & 255;
            E = E < 15 ? 15 : E;
            const int E_new = first ? E : (E > E_run ? E : E_run);
            const int dE = E_new - E_run;
            ratio = (first || dE == 0) ? 1.0f : (dE > 126 ? 0.0f : __uint_as_float((unsigned)(127 - dE) << 23));
            E_run = E_new;
            sc = __uint_as_float((unsigned)(268 - E_run) << 23);
            inv_run = __uint_as_float((unsigned)(E_run - 14) << 23);
        }
#pragma unroll
        for (int j = 0; j < SU_NI; ++j) {
            unsigned pk[NP][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (NP == 3) s3_split2(xr[j][2 * e], xr[j][2 * e + 1], pk[0][e], pk[1][e], pk[2][e]);
                else s3_split2_f16(xr[j][2 * e], xr[j][2 * e + 1], sc, pk[0][e], pk[1][e]);
            }
            if (sk_pos[j] >= 0) {
#pragma unroll
                for (int p = 0; p < NP; ++p) Xs[p * XWORDS + sk_lw[j]] = (u32x4){pk[p][0], pk[p][1], pk[p][2], pk[p][3]};
            }
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid_ + SU_THREADS * it;
            if (i < WCH) Ws[i] = wv[it];
        }
    };

    set_tile(t_lo);
    load_stage(0);
    if constexpr (NP == 2) { publish_max(); __syncthreads(); }
    store_stage(0, true);
    for (int tile = t_lo; tile < t_hi; tile += t_step) {
    const int cd0 = d0 >> 1, ch0 = h0 >> 1, cw0 = w0 >> 1, cbt = bt;      // low-resolution origin of the tile being computed
    __syncthreads();
    f32x4 acc[4][NCT];                                          // row r = 2 (ld - 2 rh) + lh of this wave's half
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[r][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float inv_fin = 1.0f;
    for (int q = 0; q < Q; ++q) {
        const bool last = q + 1 == Q;
        if (last) set_tile(tile + t_step);
        load_stage(last ? 0 : q + 1);
        if constexpr (NP == 2) {
            if (ratio != 1.0f) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[r][ct] *= ratio;
            }
        }
#pragma unroll
        for (int aw = 0; aw < 4; ++aw) {
            u32x4 a[NP][NCT], bf[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[0][p] = Xs[p * XWORDS + xb[aw]];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) a[p][ct] = Ws[(((kq * 4 + aw) * NP + p) * NCT + ct) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r + 1 < 4) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) bf[(r + 1) & 1][p] = Xs[p * XWORDS + xb[aw] + ((r + 1) >> 1) * 2 * SU_SK_PLANE + ((r + 1) & 1) * 2 * SU_SK_ROW];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < P::NPROD; ++t)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[r][ct] = P::mfma(a[P::PA[t]][ct], bf[r & 1][P::PB[t]], acc[r][ct]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        keep_offsets();
        if (last) inv_fin = inv_run;
        publish_max();
        __syncthreads();                            // every wave is done reading this chunk
        if (!last) {
            store_stage(q + 1, false);
            __syncthreads();
        }
    }
    // ---- combine the four a_h partial sums of every output row through LDS (the staging tile is free), fixed order; then the epilogue:
    // wave w finishes output row w = (ld, lh) = (w >> 1, w & 1): unscale, LeakyReLU'(mask), planar fp32 store at low resolution
    f32x4* const Rs = reinterpret_cast<f32x4*>(smem);            // [kq 4][row 8][ct][lane]
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) Rs[((kq * 8 + rh * 4 + r) * NCT + ct) * 64 + lane] = acc[r][ct];
    __syncthreads();
    {
        float unscale = 1.0f;
        if constexpr (NP == 2) {
            const float inv_w = __uint_as_float(__builtin_amdgcn_readfirstlane((int)wp[(size_t)gridDim.y * Q * WCH].x));
            unscale = inv_fin * inv_w;
        }
        const int ld = wave >> 1, lh = wave & 1;
        const int dd = cd0 + ld, hh = ch0 + lh, ww = cw0 + n;    // low-resolution voxel of this lane
        const __amdgpu_buffer_rsrc_t ry = vxm_rsrc(gxl + (size_t)cbt * gxl_bs, (unsigned)C0 * (unsigned)Vl * 4u);
        const __amdgpu_buffer_rsrc_t rm = vxm_rsrc(mask ? mask + (size_t)cbt * mask_bs : gxl, (unsigned)C0 * (unsigned)Vl * 4u);
        const bool vok = dd < Dl && hh < Hl && ww < Wl;
        const int vox = (dd * Hl + hh) * Wl + ww;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const f32x4 p0 = Rs[((0 * 8 + wave) * NCT + ct) * 64 + lane], p1 = Rs[((1 * 8 + wave) * NCT + ct) * 64 + lane];
            const f32x4 p2 = Rs[((2 * 8 + wave) * NCT + ct) * 64 + lane], p3 = Rs[((3 * 8 + wave) * NCT + ct) * 64 + lane];
            const f32x4 sum = (p0 + p1) + (p2 + p3);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = (g * NCT + ct) * 16 + kg * 4 + j;
                const int off = (vok && ch < C0) ? (ch * Vl + vox) << 2 : VXM_OOB;
                float v = sum[j];
                if constexpr (NP == 2) v *= unscale;
                if (mask) v *= vxm_lrelu_grad(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rm, off, 0, 0)), mask_slope);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, off, 0, 0);
            }
        }
    }
    __syncthreads();                                // the combine buffer is free again
    if (tile + t_step < t_hi) store_stage(0, true); // chunk 0 of the next tile; the barrier at the top of the tile loop publishes it
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// backward-weight of the upsampled segment, collapsed and on the split arithmetic (fp16 scheme):
//   convolution_backward (weight) of cat([upsample(x0), .]) restricted to the x0 channels (autograd twin of networks.py:133-138, 299);
//   rounds 1-3: k_conv3d_k3_bwd_weight_up on the fp32 matrix pipe.
// With G[a][co][ci] = sum over low-res voxels m of dz[co][2 m + a] x0[ci][m] for the 64 offsets a in {-1,0,1,2}^3 (the offsets of
// k_s3u_dlow), the gradient of tap k is the sum of G over a_axis in A(k_axis), A(0) = {1,2}, A(1) = {0,1}, A(2) = {-1,0} per axis
// (k_s3u_bww_reduce): 64 instead of 216 products per (voxel pair of channels), and x0 is read at its own resolution.
// Contraction over VOXELS as k_s3_bwd_weight: M = 16 dz channels, N = 16 x0 channels (x NCI tiles), K = 32 = two low-res rows x 16
// low-res voxels of a W row; both operands live in LDS as [voxel][16 channel] rows and are read with ds_read_b64_tr_b16.  dz is staged
// like k_s3_bwd_weight's X (haloed planes of 6 rows x 34 voxels in a 6-plane depth ring: a tile of one low-res depth needs four
// full-resolution planes and shares two with the next), but stored de-interleaved by W parity, so that the 16 voxels 2 v + a_w an
// offset reads are consecutive rows; x0 has no halo (2 x 16 voxels per tile, double-buffered).  Block = 16 waves = (a_d, a_h); a wave
// keeps its 4 (a_w) x NCI accumulator tiles over all its tiles (per-tile MFMA chains folded into fp32 totals, as k_s3_bwd_weight).
// One block per CU and dz-channel tile walks a range of (column, depth segment) tasks; per-block partial sums of G, reduced and
// mapped onto the 27 taps in a fixed order (deterministic).
// A PHASE covers two low-res depths (two "tiles"): what bounds this kernel is not arithmetic (24 MFMAs per wave and tile) but the latency of
// the loads of the next phase, which has to be hidden behind this one -- so a phase requests as many bytes as the block can hold: four dz
// planes + two x0 tiles on all 1024 threads (one slot each), a 10-plane ring (six in use, four arriving).
constexpr int UW_WAVES = 16, UW_THREADS = 64 * UW_WAVES, UW_RING = 10, UW_TPP = 2;
constexpr int UW_ROWV = 17;                                    // voxels of one parity in a haloed row of 34
constexpr int UW_PLANE = 6 * 2 * UW_ROWV * 32;                 // bytes of one haloed dz plane of one piece: [row 6][parity 2][17][16 co]
constexpr int UW_XPIECE = UW_RING * UW_PLANE;
constexpr int UW_LTILE = 2 * 16 * 32;                          // x0 tile of one (piece, ci tile): [low-res row 2][16 voxels][16 ci]
template <int NCI> struct UwCfg {
    static constexpr int NP = 2;
    static constexpr int LBUF = UW_TPP * NP * NCI * UW_LTILE;  // one x0 buffer: the two tiles of a phase
    static constexpr int TILE_BYTES = NP * UW_XPIECE + 2 * LBUF;
    static constexpr int LDS_BYTES = TILE_BYTES + 256;         // + floats: [0..9] ring planes, [10..11] x0 buffers, [16..] wave maxima (two slots of 16)
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// dz: [B][Cdz][D][H][W], x0: [B][C0][D/2][H/2][W/2]; grid = NBLK x NCOT (16-channel tiles of dz)
template <int NCI>
__global__ void __launch_bounds__(UW_THREADS, 4) k_s3u_bww(const float* __restrict__ x0, long long x0_bs, int C0, const float* __restrict__ dz,
                                                           long long dz_bs, int Cdz, float* __restrict__ part, int D, int H, int W, int NBLK,
                                                           int ncol, int nseg, int seg_len, int nh, int nw, int task_rr, int lay) {
    using CF = UwCfg<NCI>;
    using P = S3P<2>;
    constexpr int NP = 2;
    VXM_DYN_SMEM(char, smem);
    char* const Xs = smem;                                       // [NP][10 ring planes][UW_PLANE]
    char* const Ls = smem + NP * UW_XPIECE;                      // [2 buffers][tile of the phase 2][NP][NCI][UW_LTILE]
    float* const Tab = reinterpret_cast<float*>(smem + CF::TILE_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ad = wave >> 2, ah = wave & 3;                     // this wave's offsets (index = a + 1)
    const int NCOT = gridDim.x / NBLK, xmain = NBLK & ~7;
    int bx, cot;
    if ((int)blockIdx.x < xmain * NCOT) {                        // the tiles of one task range 8 workgroup ids apart: one XCD's L2 serves both
        const int j = blockIdx.x >> 3;
        cot = j % NCOT;
        bx = (j / NCOT) * 8 + (blockIdx.x & 7);
    } else {
        const int r = blockIdx.x - xmain * NCOT;
        bx = xmain + r / NCOT;
        cot = r % NCOT;
    }
    const int Dl = D >> 1, Hl = H >> 1, Wl = W >> 1;
    const int ntask = ncol * nseg;
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
    const int V = D * H * W, HW = H * W, Vl = Dl * Hl * Wl, HWl = Hl * Wl;

    f32x4 tot[4][NCI];
#pragma unroll
    for (int aw = 0; aw < 4; ++aw)
#pragma unroll
        for (int c = 0; c < NCI; ++c) tot[aw][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lp = (4 * (lane >> 4) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;     // lane pattern of the transposing read

    // staging roles (wave-uniform): waves 0 .. 13 stage dz -- slot = (plane of the four 4, row 6, W pair 18, 8-channel half): 864 slots --,
    // waves 14, 15 the two x0 tiles -- slot = (tile 2, low-res row 2, W pair 8, 8-channel block of the NCI x 16 channels)
    constexpr int NXS = 6 * 18 * 2;
    const int s_role = wave <= 13 ? 1 : 2;
    const int s_i = s_role == 1 ? tid : tid - 14 * 64;
    const int x_pl = s_role == 1 ? s_i / NXS : (s_i >> 6), x_r = s_role == 1 ? s_i - x_pl * NXS : (s_i & 63);      // plane of the four / tile of the two
    float ra[8], rb[8];
    unsigned ka[NP][4], kb[NP][4];
    int off0 = VXM_OOB, ldst = 0, vk;

    // (the task loop per staging role and dz layout, compile-time inside it: see k_s3_bwd_weight)
    auto run_tasks = [&](auto xr_, auto bl_) __attribute__((always_inline)) {
        constexpr bool XR = decltype(xr_)::value, BLK = decltype(bl_)::value;
        for (int task = k_lo; task < k_hi; task += k_step) {
            const int seg = task_rr ? task / ncol : task % nseg, col = task_rr ? task - seg * ncol : task / nseg;        // (depth-segment-major when round-robin)
            const int tw = col % nw; int cq = col / nw;
            const int th = cq % nh; const int b = cq / nh;
            const int md0 = seg * seg_len, ntile = min(seg_len, Dl - md0), nphase = (ntile + UW_TPP - 1) / UW_TPP;
            const int mh0 = th * 2, mw0 = tw * 16;
            constexpr bool xrole = XR;                             // (= s_role != 2: the task loop is instantiated per staging role, as k_s3_bwd_weight)
            constexpr bool blk = XR && BLK;                          // dz channel-blocked [Cdz / 8][voxel][8] (lay & VXM_S3_IN1_BLOCKED); x0 stays planar
            constexpr int lsh = blk ? 5 : 2;
            const __amdgpu_buffer_rsrc_t rd = vxm_rsrc(xrole ? dz + (size_t)b * dz_bs : x0 + (size_t)b * x0_bs, (unsigned)(xrole ? Cdz * V : C0 * Vl) * 4u);
            if (s_role == 1) {
                const int cb = x_r & 1, pr = x_r >> 1, hh = pr / 18, pp = pr - hh * 18;
                const int gh = 2 * mh0 - 1 + hh, gw = 2 * mw0 - 2 + 2 * pp;
                const bool live = x_pl < 4 && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W && cot * 16 + cb * 8 < Cdz;
                off0 = live ? (blk ? ((cot * 2 + cb) * V + gh * W + gw) << 5 : ((cot * 16 + cb * 8) * V + gh * W + gw) << 2) : VXM_OOB;
                // first voxel of the pair: haloed column 2 pp - 1 (parity 1, index pp - 1), second: column 2 pp (parity 0, index pp)
                ldst = (((hh * 2 + 1) * UW_ROWV + pp - 1) * 32 + cb * 16) | (pp == 0 ? 1 : 0) | (pp == 17 ? 2 : 0);
            } else {
                const int cbq = x_r & 3, r2 = x_r >> 2, lr = r2 >> 3, pair = r2 & 7;
                const bool live = cbq * 8 < C0 && cbq < 2 * NCI && mh0 + lr < Hl && mw0 + 2 * pair < Wl;
                off0 = live ? (cbq * 8 * Vl + (mh0 + lr) * Wl + mw0 + 2 * pair) << 2 : VXM_OOB;
                ldst = ((cbq >> 1) * UW_LTILE) + (lr * 16 + 2 * pair) * 32 + (cbq & 1) * 16;
            }
            // dz planes p0 .. p0 + np - 1 of this task (plane p = full-resolution depth 2 md0 - 1 + p) and the x0 tiles tl, tl + 1 (tl < 0: none)
            // -> registers; a plane outside the volume / a tile outside the task ORs the out-of-range bit into the lane offsets (branch-free)
            auto load_phase = [&](int p0, int np, int tl) __attribute__((always_inline)) {
                const int gd = 2 * md0 - 1 + p0 + x_pl;                                   // (dz role: per-lane plane)
                const bool pok = x_pl < np && (unsigned)gd < (unsigned)D;
                const int mt = tl + x_pl;                                                 // (x0 role: wave-uniform tile)
                const bool lok = tl >= 0 && mt < ntile;
                vk = xrole ? (pok ? off0 + ((gd * HW) << lsh) : VXM_OOB) : (lok ? off0 : VXM_OOB);
                const int sb = xrole ? 0 : ((md0 + (lok ? mt : 0)) * HWl) << 2;
                const int cs = (xrole ? V : Vl) << 2;
                if constexpr (blk) {
    #pragma unroll
                    for (int k = 0; k < 2; ++k) {                    // first voxel: channels 4 k .. 4 k + 3; second voxel 32 bytes on
                        const f32x4 ta = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vk, 16 * k, 0));
                        const f32x4 tb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vk, 32 + 16 * k, 0));
    #pragma unroll
                        for (int e = 0; e < 4; ++e) { ra[4 * k + e] = ta[e]; rb[4 * k + e] = tb[e]; }
                    }
                } else {
    #pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const f32x2 t2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rd, vk, sb + e * cs, 0));
                        ra[e] = t2.x; rb[e] = t2.y;
                    }
                }
            };
            auto publish_max = [&](int slot) __attribute__((always_inline)) {
                const float m = s3_unit_max([&](auto&& f) __attribute__((always_inline)) {
    #pragma unroll
                    for (int e = 0; e < 8; ++e) { f(ra[e]); f(rb[e]); }
                });
                if (lane == 0) Tab[16 + 16 * slot + wave] = m;
            };
            float sc_role = 1.0f;
            auto take_scales = [&](int slot, int p0, int np, int lbuf) __attribute__((always_inline)) {
                const f32x4* const t4 = reinterpret_cast<const f32x4*>(Tab + 16 + 16 * slot);
                const f32x4 m0 = t4[0], m1 = t4[1], m2 = t4[2], m3 = t4[3];
                const float mxx = fmaxf(fmaxf(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), fmaxf(fmaxf(m1.x, m1.y), fmaxf(m1.z, m1.w))),
                                        fmaxf(fmaxf(fmaxf(m2.x, m2.y), fmaxf(m2.z, m2.w)), fmaxf(m3.x, m3.y)));          // waves 0 .. 13
                const float mxl = fmaxf(m3.z, m3.w);                                                                     // waves 14, 15
                float sx, ix, sl, il;
                s3_scale_of(mxx, sx, ix);
                s3_scale_of(mxl, sl, il);
                sc_role = s_role == 2 ? sl : sx;
                if (tid < np) Tab[(p0 + tid) % UW_RING] = ix;
                if (tid == 64 && lbuf >= 0) Tab[10 + lbuf] = il;
            };
            auto split_tile = [&]() __attribute__((always_inline)) {
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s3_split2_f16(ra[2 * e], ra[2 * e + 1], sc_role, ka[0][e], ka[1][e]);
                    s3_split2_f16(rb[2 * e], rb[2 * e + 1], sc_role, kb[0][e], kb[1][e]);
                }
            };
            auto write_phase = [&](int p0, int np, int lbuf) __attribute__((always_inline)) {
                if (s_role == 1) {
                    if (x_pl < np) {
                        char* const d = Xs + ((p0 + x_pl) % UW_RING) * UW_PLANE + (ldst & ~3);
                        if (!(ldst & 1)) {
    #pragma unroll
                            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(d + p * UW_XPIECE) = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                        }
                        if (!(ldst & 2)) {                           // the second voxel: parity 0, one parity block back and one index on
    #pragma unroll
                            for (int p = 0; p < NP; ++p)
                                *reinterpret_cast<u32x4*>(d + p * UW_XPIECE - UW_ROWV * 32 + 32) = (u32x4){kb[p][0], kb[p][1], kb[p][2], kb[p][3]};
                        }
                    }
                } else if (lbuf >= 0) {
    #pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        char* const d = Ls + lbuf * CF::LBUF + (x_pl * NP + p) * NCI * UW_LTILE + ldst;
                        *reinterpret_cast<u32x4*>(d) = (u32x4){ka[p][0], ka[p][1], ka[p][2], ka[p][3]};
                        *reinterpret_cast<u32x4*>(d + 32) = (u32x4){kb[p][0], kb[p][1], kb[p][2], kb[p][3]};
                    }
                }
            };

            __syncthreads();                                        // every wave is done with the previous task
            load_phase(0, 4, 0);                                    // planes 0 .. 3, tiles 0, 1
            publish_max(0);
            __syncthreads();
            take_scales(0, 0, 4, 0);
            split_tile();
            write_phase(0, 4, 0);
            load_phase(4, 2, -1);                                   // planes 4, 5
            publish_max(1);
            __syncthreads();
            take_scales(1, 4, 2, -1);
            split_tile();
            write_phase(4, 2, -1);
            __syncthreads();
            for (int t = 0; t < nphase; ++t) {
                const bool more = t + 1 < nphase;
                const int lcur = t & 1;
                load_phase(4 * t + 6, 4, more ? UW_TPP * (t + 1) : -1);
                __builtin_amdgcn_sched_barrier(0);
    #pragma nounroll
                for (int u = 0; u < UW_TPP; ++u) {                   // (not unrolled: the two tiles' fragment reads side by side cost 44 spilled registers)
                    const int slot = (4 * t + 2 * u + ad) % UW_RING;
                    const char* const xp = Xs + slot * UW_PLANE;
                    const float unscale = Tab[slot] * Tab[10 + lcur];
                    u32x4 bl[NCI][NP];                               // x0 fragments (B operand): K = (low-res row, 16 voxels)
    #pragma unroll
                    for (int c = 0; c < NCI; ++c)
    #pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const int lb = lcur * CF::LBUF + ((u * NP + p) * NCI + c) * UW_LTILE + lp;
                            const u32x2 lo = s3_tr_read(Ls, lb), hi = s3_tr_read(Ls, lb + 16 * 32);
                            bl[c][p] = (u32x4){lo.x, lo.y, hi.x, hi.y};
                        }
    #pragma unroll
                    for (int aw = 0; aw < 4; ++aw) {
                        u32x4 az[NP];                                // dz fragment (A operand) of offset (ad, ah, aw): rows ah and ah + 2, parity aw & 1, from index aw >> 1
    #pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const int zb = p * UW_XPIECE + ((ah * 2 + (aw & 1)) * UW_ROWV + (aw >> 1)) * 32 + lp;
                            const u32x2 lo = s3_tr_read(xp, zb), hi = s3_tr_read(xp, zb + 2 * 2 * UW_ROWV * 32);
                            az[p] = (u32x4){lo.x, lo.y, hi.x, hi.y};
                        }
    #pragma unroll
                        for (int c = 0; c < NCI; ++c) {
                            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                            for (int tp = 0; tp < P::NPROD; ++tp) acc = P::mfma(az[P::PA[tp]], bl[c][P::PB[tp]], acc);
    #pragma unroll
                            for (int j = 0; j < 4; ++j) tot[aw][c][j] = __builtin_fmaf(acc[j], unscale, tot[aw][c][j]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" ::"v"(vk));
                publish_max(0);
                __syncthreads();                                     // every wave is done reading this phase; the maxima of the next are published
                if (more) {
                    take_scales(0, 4 * t + 6, 4, lcur ^ 1);
                    split_tile();
                    write_phase(4 * t + 6, 4, lcur ^ 1);
                }
                __syncthreads();
            }
        }
    };
    if (s_role != 2) { if (lay & VXM_S3_IN1_BLOCKED) run_tasks(std::true_type{}, std::true_type{}); else run_tasks(std::true_type{}, std::false_type{}); }
    else run_tasks(std::false_type{}, std::false_type{});
    // ---- partials: part[bx][cot][a = (ad, ah, aw)][co 16][ci 16 NCI]; lane (kg, n) holds co = 4 kg + j, ci = 16 c + n
    float* const pp = part + (((size_t)bx * NCOT + cot) * 64 + (ad * 4 + ah) * 4) * (16 * 16 * NCI);
#pragma unroll
    for (int aw = 0; aw < 4; ++aw)
#pragma unroll
        for (int c = 0; c < NCI; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) pp[(size_t)aw * (16 * 16 * NCI) + (4 * (lane >> 4) + j) * (16 * NCI) + c * 16 + (lane & 15)] = tot[aw][c][j];
}

// gw[co][ci][tap] (row length gw_cin, the first C0 input channels) = sum over the blocks' partials and over the offsets of the tap, fixed order.
// Block = 64 consecutive (col, ci-tile, ci) elements of one (dz-channel tile, tap) x 16 slices of the blocks' partials: a thread sums the 8
// offsets of its tap over every 16th block (256-byte coalesced reads), the slices are combined through LDS in a fixed tree (deterministic).
// (The first version -- one thread per output walking all NBLK partials, 108 blocks -- took 123 us for 67 MB of partials.)
__global__ void __launch_bounds__(1024) k_s3u_bww_reduce(const float* __restrict__ part, float* __restrict__ gw, int C0, int Cout, int gw_cin, int NCI,
                                                         int NCOT, int NBLK) {
    __shared__ float sm[16][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const int per_a = 16 * 16 * NCI, chunks = per_a / 64;                 // elements of one (block, dz tile, offset); 64-element chunks of them
    int j = blockIdx.x;
    const int chunk = j % chunks; j /= chunks;
    const int tap = j % 27, cot = j / 27;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    const int el = chunk * 64 + x;                                        // (col 16, c NCI, ci16 16)
    const size_t per_blk = (size_t)NCOT * 64 * per_a;
    const float* const p0 = part + (size_t)cot * 64 * per_a + el;
    // offsets index (a + 1) that feed tap k: k = 0: {2, 3}; k = 1: {1, 2}; k = 2: {0, 1}
    const int d0 = 2 - kd, h0 = 2 - kh, w0 = 2 - kw;
    float s = 0.0f;
    for (int blk = y; blk < NBLK; blk += 16) {
        const float* const pb = p0 + (size_t)blk * per_blk;
        float t = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int a = ((d0 + (i >> 2)) * 4 + (h0 + ((i >> 1) & 1))) * 4 + (w0 + (i & 1));
            t += pb[(size_t)a * per_a];
        }
        s += t;
    }
    sm[y][x] = s;
    __syncthreads();
    if (y != 0) return;
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = sm[u][x];
    const float sum = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
                      (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
    const int col = el / (16 * NCI), c = (el / 16) % NCI, ci = c * 16 + (el & 15), co = cot * 16 + col;
    if (co < Cout && ci < C0) gw[((size_t)co * gw_cin + ci) * 27 + tap] = sum;
}

int su_cus() {
    static const int cus = [] {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }();
    return cus;
}
int su_nct(int Cout) { return Cout > 16 ? 2 : 1; }
size_t su_words(int C0, int C1, int Cout, int NP) {
    const int NCT = su_nct(Cout), G = (Cout + 16 * NCT - 1) / (16 * NCT);
    return (size_t)G * ((size_t)(C0 / 8) * (8 * 2 * NP * NCT * 64) + (size_t)(C1 / 8) * (7 * NP * NCT * 64));
}
long long su_min_tiles() {
    static const long long v = [] { const char* e = getenv("VXM_S3U_MIN_TILES"); return e ? atoll(e) : 512ll; }();
    return v;
}

int su_dbg() {
#ifdef VXM_S3_EXP
    const char* e = getenv("VXM_S3_DBG");
    return e ? atoi(e) : 0;
#else
    return 0;
#endif
}

// does k_s3u_conv_pc take this launch (fp16 pieces)?  Its staging loads W pairs of both resolutions: W and W / 2 even, even strides.
// VXM_S3U_PC=0: k_s3u_conv everywhere (same-box A/B)
bool su_pc_ok(long long bs0, long long bs1, int D, int H, int W) {
    static const bool pc_on = [] { const char* e = getenv("VXM_S3U_PC"); return !(e && e[0] == '0'); }();
    return pc_on && W % 4 == 0 && (bs0 & 1) == 0 && (bs1 & 1) == 0 && ((D / 2) * (H / 2) * (W / 2)) % 2 == 0;
}

template <int NCT, int NP>
void su_launch(const float* x0, long long bs0, int C0, const float* x1, long long bs1, int C1, const void* wp, const float* bias, float* y,
               long long y_bs, int Cout, float slope, int B, int D, int H, int W, hipStream_t s, int lay, unsigned char* signs = nullptr, long long signs_bs = 0) {
    using C = SuCfg<NCT, NP>;
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3u_conv<NCT, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        return true;
    }();
    (void)attr;
    const long long ntiles = (long long)B * ((D + 7) / 8) * ((H + 3) / 4) * ((W + 31) / 32);
    const int G = (Cout + 16 * NCT - 1) / (16 * NCT);
    unsigned gx = ntiles >= 64 ? (unsigned)(8 * ((ntiles + 7) / 8)) : (unsigned)ntiles;
    // one persistent block per CU (the LDS holds one): its blocks walk their XCD's tile range and stage ahead across tiles
    static const int persist = [] { const char* e = getenv("VXM_S3U_PERSIST"); return e ? atoi(e) : 1; }();      // < 0: that many blocks in all (tests)
    if (persist != 0 && ntiles >= 64) {
        const unsigned want = persist > 0 ? (unsigned)(su_cus() * persist / G) : (unsigned)(-persist);
        const unsigned cap = 8 * ((want + 7) / 8);
        if (cap < gx) gx = cap;
    }
    // fp16 pieces, W and W / 2 even (the staging loads W pairs of both resolutions): the producer / consumer kernel.  VXM_S3U_PC=0: k_s3u_conv (A/B)
    if constexpr (NP == 2) {
        if (su_pc_ok(bs0, bs1, D, H, W)) {
            static const bool attr2 = [] {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3u_conv_pc<NCT>), hipFuncAttributeMaxDynamicSharedMemorySize, SUP_LDS_BYTES);
                return true;
            }();
            (void)attr2;
            hipLaunchKernelGGL((k_s3u_conv_pc<NCT>), dim3(gx, G), dim3(SUP_THREADS), SUP_LDS_BYTES, s, x0, bs0, C0, x1, bs1, C1, static_cast<const u32x4*>(wp),
                               bias, y, y_bs, Cout, slope, B, D, H, W, lay, su_dbg(), signs, signs_bs);
            return;
        }
    }
    hipLaunchKernelGGL((k_s3u_conv<NCT, NP>), dim3(gx, G), dim3(SU_THREADS), C::LDS_BYTES, s, x0, bs0, C0, x1, bs1, C1, static_cast<const u32x4*>(wp),
                       bias, y, y_bs, Cout, slope, B, D, H, W, lay, su_dbg(), signs, signs_bs);
}


size_t sd_words(int C0, int Cout, int NP, int NCT) {
    const int G = (C0 + 16 * NCT - 1) / (16 * NCT), Q = (Cout + 7) / 8;
    return (size_t)G * Q * 16 * NP * NCT * 64;
}
template <int NCT, int NP, bool BLK = false>
void sd_launch(const float* dz, long long dz_bs, int Cout, const void* wp, float* gxl, long long gxl_bs, int C0, const float* mask, long long mask_bs,
               float mask_slope, int B, int D, int H, int W, hipStream_t s) {
    using C = SdCfg<NCT, NP>;
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3u_dlow<NCT, NP, BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        return true;
    }();
    (void)attr;
    const long long ntiles = (long long)B * ((D / 2 + 3) / 4) * ((H / 2 + 1) / 2) * ((W / 2 + 15) / 16);
    const int G = (C0 + 16 * NCT - 1) / (16 * NCT);
    unsigned gx = ntiles >= 64 ? (unsigned)(8 * ((ntiles + 7) / 8)) : (unsigned)ntiles;
    static const int persist = [] { const char* e = getenv("VXM_S3U_PERSIST"); return e ? atoi(e) : 1; }();
    if (persist != 0 && ntiles >= 64) {
        const unsigned want = persist > 0 ? (unsigned)(su_cus() * persist / G) : (unsigned)(-persist);
        const unsigned cap = 8 * ((want + 7) / 8);
        if (cap < gx) gx = cap;
    }
    hipLaunchKernelGGL((k_s3u_dlow<NCT, NP, BLK>), dim3(gx, G), dim3(SU_THREADS), C::LDS_BYTES, s, dz, dz_bs, Cout, static_cast<const u32x4*>(wp), gxl, gxl_bs,
                       C0, mask, mask_bs, mask_slope, B, D, H, W);
}


template <int NCTL, int NCTS, bool BLK>
void sbp_launch(const float* dz, long long dz_bs, int Cout, const void* wlow, float* gxl, long long gxl_bs, int C0, const float* mask, long long mask_bs,
                float mask_slope, const void* wskip, float* gx1, long long gx1_bs, int C1, int B, int D, int H, int W, hipStream_t s) {
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3u_bwd_pc<NCTL, NCTS, BLK>), hipFuncAttributeMaxDynamicSharedMemorySize, SBP_LDS_BYTES);
        return true;
    }();
    (void)attr;
    const long long ntiles = (long long)B * ((D + 3) / 4) * ((H + 3) / 4) * ((W + 31) / 32);
    unsigned gx = ntiles >= 64 ? (unsigned)(8 * ((ntiles + 7) / 8)) : (unsigned)ntiles;
    static const int persist = [] { const char* e = getenv("VXM_S3U_PERSIST"); return e ? atoi(e) : 1; }();
    if (persist != 0 && ntiles >= 64) {
        const unsigned want = persist > 0 ? (unsigned)(su_cus() * persist) : (unsigned)(-persist);
        const unsigned cap = 8 * ((want + 7) / 8);
        if (cap < gx) gx = cap;
    }
    hipLaunchKernelGGL((k_s3u_bwd_pc<NCTL, NCTS, BLK>), dim3(gx), dim3(SUP_THREADS), SBP_LDS_BYTES, s, dz, dz_bs, Cout, static_cast<const u32x4*>(wlow), gxl,
                       gxl_bs, C0, mask, mask_bs, mask_slope, static_cast<const u32x4*>(wskip), gx1, gx1_bs, C1, B, D, H, W, su_dbg());
}

struct UwTasks { int ncol, nseg, seg_len, nh, nw, NBLK; };
UwTasks uw_tasks(int ncot, int B, int D, int H, int W) {
    UwTasks tk;
    const int Dl = D / 2, Hl = H / 2, Wl = W / 2;
    tk.nh = (Hl + 1) / 2; tk.nw = (Wl + 15) / 16;
    tk.ncol = B * tk.nh * tk.nw;
    const int nb = su_cus() / ncot > 0 ? su_cus() / ncot : 1;
    int nseg = (4 * nb + tk.ncol - 1) / tk.ncol;
    if (nseg > Dl) nseg = Dl;
    if (nseg < 1) nseg = 1;
    tk.seg_len = (Dl + nseg - 1) / nseg;
    tk.nseg = (Dl + tk.seg_len - 1) / tk.seg_len;
    const long long ntask = (long long)tk.ncol * tk.nseg;
    tk.NBLK = (int)(nb < ntask ? nb : ntask);
    return tk;
}

}  // namespace

extern "C" {

int vxm_conv3d_k3_s3u_ok(int C0, int C1, int Cout, int B, int D, int H, int W, int pieces) {
    if (C0 <= 0 || C1 < 0 || Cout < 8 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || (pieces != 2 && pieces != 3)) return 0;
    if (C0 % 8 || C1 % 8 || (D | H | W) & 1) return 0;
    if ((long long)(C0 + C1 > Cout ? C0 + C1 : Cout) * D * H * W >= (1ll << 29)) return 0;
    const long long ntiles = (long long)B * ((D + 7) / 8) * ((H + 3) / 4) * ((W + 31) / 32);
    return ntiles >= su_min_tiles() ? 1 : 0;
}

size_t vxm_conv3d_k3_s3u_packed_bytes(int C0, int C1, int Cout, int pieces) {
    if (C0 <= 0 || C1 < 0 || Cout <= 0 || C0 % 8 || C1 % 8 || (pieces != 2 && pieces != 3)) return 0;
    return (su_words(C0, C1, Cout, pieces) + 1) * 16;
}

/* 1 when vxm_conv3d_k3_s3u_fwd launches the producer / consumer kernel k_s3u_conv_pc for this call (for profiles and bench regions), else 0 (k_s3u_conv) */
int vxm_conv3d_k3_s3u_fwd_kernel(int64_t x0_bstride, int64_t x1_bstride, int D, int H, int W, int pieces) {
    return ((pieces & 0xff) == 2 && su_pc_ok(x0_bstride, x1_bstride, D, H, W)) ? 1 : 0;
}

int vxm_conv3d_k3_s3u_pack_weights(const float* w, void* wpacked, int C0, int C1, int Cout, int pieces, void* stream) {
    VXM_REQUIRE(w && wpacked, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3u_pack_weights: null pointer");
    VXM_REQUIRE(C0 > 0 && C1 >= 0 && Cout > 0 && C0 % 8 == 0 && C1 % 8 == 0 && (pieces == 2 || pieces == 3) &&
                    (reinterpret_cast<uintptr_t>(wpacked) & 15) == 0, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_pack_weights: %d + %d -> %d channels (segments in multiples of 8), pieces %d, 16-byte aligned destination", C0, C1, Cout, pieces);
    const int NCT = su_nct(Cout);
    SuPack jb = {w, static_cast<u32x4*>(wpacked), C0, C1, Cout, NCT, pieces, C0 / 8, C1 / 8, (Cout + 16 * NCT - 1) / (16 * NCT),
                 (unsigned)su_words(C0, C1, Cout, pieces), 0, 0, 0};
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(jb.wp + jb.words, 0, 16, s);
    const unsigned blocks = (jb.words + 255) / 256;
    if (pieces == 2) hipLaunchKernelGGL(k_s3u_pack<0>, dim3(blocks), dim3(256), 0, s, jb);
    hipLaunchKernelGGL(k_s3u_pack<1>, dim3(blocks), dim3(256), 0, s, jb);
    return vxm_check_launch("vxm_conv3d_k3_s3u_pack_weights");
}

int vxm_conv3d_k3_s3u_fwd(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const void* wpacked,
                          const float* bias, float* y, int64_t y_bstride, int Cout, float leaky_slope, int B, int D, int H, int W, int pieces_and_layout,
                          void* stream) {
    return vxm_conv3d_k3_s3u_fwd_signs(x0, C0, x0_bstride, x1, C1, x1_bstride, wpacked, bias, y, y_bstride, Cout, leaky_slope, B, D, H, W, pieces_and_layout,
                                       nullptr, 0, stream);
}

int vxm_conv3d_k3_s3u_fwd_signs(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const void* wpacked,
                                const float* bias, float* y, int64_t y_bstride, int Cout, float leaky_slope, int B, int D, int H, int W, int pieces_and_layout,
                                unsigned char* signs, int64_t signs_bstride, void* stream) {
    const int pieces = pieces_and_layout & 0xff, lay = pieces_and_layout & ~0xff;
    VXM_REQUIRE(x0 && wpacked && y && (C1 == 0 || x1), VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3u_fwd: null pointer");
    VXM_REQUIRE(signs == nullptr || (lay == VXM_S3_OUT_BLOCKED && Cout % 8 == 0), VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_fwd_signs: the sign tensor goes with a channel-blocked output");
    VXM_REQUIRE(lay == 0 || (lay == VXM_S3_OUT_BLOCKED && Cout % 8 == 0), VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_fwd: layout flags 0x%x (only the output may be channel-blocked; Cout = %d in multiples of 8)", lay, Cout);
    if (int e = check_conv("vxm_conv3d_k3_s3u_fwd", C0, C1, 1, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(C0 % 8 == 0 && C1 % 8 == 0 && (pieces == 2 || pieces == 3), VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_fwd: segments carry multiples of 8 channels (got %d + %d), pieces 2 or 3 (got %d)", C0, C1, pieces);
    VXM_REQUIRE((reinterpret_cast<uintptr_t>(wpacked) & 15) == 0, VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3u_fwd: packed weights must be 16-byte aligned");
    hipStream_t s = VXM_STREAM(stream);
    const int NCT = su_nct(Cout);
#define SU_GO(NCT_)                                                                                                                  \
    do {                                                                                                                             \
        if (pieces == 2) su_launch<NCT_, 2>(x0, x0_bstride, C0, x1, x1_bstride, C1, wpacked, bias, y, y_bstride, Cout, leaky_slope, B, D, H, W, s, lay, signs, signs_bstride); \
        else su_launch<NCT_, 3>(x0, x0_bstride, C0, x1, x1_bstride, C1, wpacked, bias, y, y_bstride, Cout, leaky_slope, B, D, H, W, s, lay, signs, signs_bstride);            \
    } while (0)
    if (NCT == 2) SU_GO(2); else SU_GO(1);
#undef SU_GO
    return vxm_check_launch("vxm_conv3d_k3_s3u_fwd");
}

/* backward-data of the upsampled segment onto the low-resolution tensor (k_s3u_dlow); the fp16 scheme only (the three bf16 pieces of the
 * 16-step weight chunk do not fit the LDS beside the staging tile: pieces = 3 callers keep vxm_conv3d_k3_up_bwd_low) */
int vxm_conv3d_k3_s3u_bwd_low_ok(int C0, int Cout, int B, int D, int H, int W, int pieces) {
    if (C0 <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || pieces != 2) return 0;
    if (C0 % 8 || ((D | H | W) & 1)) return 0;
    if ((long long)(C0 > Cout ? C0 : Cout) * D * H * W >= (1ll << 29)) return 0;
    const long long ntiles = (long long)B * ((D / 2 + 3) / 4) * ((H / 2 + 1) / 2) * ((W / 2 + 15) / 16);
    return ntiles >= su_min_tiles() ? 1 : 0;
}

size_t vxm_conv3d_k3_s3u_bwd_low_packed_bytes(int C0, int Cout, int pieces) {
    if (C0 <= 0 || Cout <= 0 || pieces != 2) return 0;
    return (sd_words(C0, Cout, pieces, su_nct(C0)) + 1) * 16;
}

int vxm_conv3d_k3_s3u_bwd_low_pack_weights(const float* w, void* wpacked, int C0, int Cin, int Cout, int pieces, void* stream) {
    VXM_REQUIRE(w && wpacked, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3u_bwd_low_pack_weights: null pointer");
    VXM_REQUIRE(C0 > 0 && Cin >= C0 && Cout > 0 && pieces == 2 && (reinterpret_cast<uintptr_t>(wpacked) & 15) == 0, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_bwd_low_pack_weights: %d of %d input channels, %d outputs, pieces %d (2), 16-byte aligned destination", C0, Cin, Cout, pieces);
    const int NCT = su_nct(C0);
    SuPack jb = {w, static_cast<u32x4*>(wpacked), C0, Cin - C0, Cout, NCT, pieces, 0, 0, (C0 + 16 * NCT - 1) / (16 * NCT),
                 (unsigned)sd_words(C0, Cout, pieces, NCT), 1, 0, 0};
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(jb.wp + jb.words, 0, 16, s);
    const unsigned blocks = (jb.words + 255) / 256;
    hipLaunchKernelGGL(k_s3u_pack<0>, dim3(blocks), dim3(256), 0, s, jb);
    hipLaunchKernelGGL(k_s3u_pack<1>, dim3(blocks), dim3(256), 0, s, jb);
    return vxm_check_launch("vxm_conv3d_k3_s3u_bwd_low_pack_weights");
}

int vxm_conv3d_k3_s3u_bwd_low(const float* dz, int64_t dz_bstride, int Cout, const void* wpacked, float* gxl, int64_t gxl_bstride, int C0,
                              const float* mask, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W, int pieces_and_layout, void* stream) {
    const int pieces = pieces_and_layout & 0xff, lay = pieces_and_layout & ~0xff;
    VXM_REQUIRE(dz && wpacked && gxl, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3u_bwd_low: null pointer");
    VXM_REQUIRE(lay == 0 || (lay == VXM_S3_IN0_BLOCKED && Cout % 8 == 0), VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_bwd_low: layout flags 0x%x (only dz, the first operand, may be channel-blocked; Cout = %d in multiples of 8)", lay, Cout);
    if (int e = check_conv("vxm_conv3d_k3_s3u_bwd_low", C0, 0, 1, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(pieces == 2 && (reinterpret_cast<uintptr_t>(wpacked) & 15) == 0, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_bwd_low: pieces %d (this kernel runs the fp16 scheme, 2), 16-byte aligned packed weights", pieces);
    hipStream_t s = VXM_STREAM(stream);
    if (su_nct(C0) == 2) {
        if (lay) sd_launch<2, 2, true>(dz, dz_bstride, Cout, wpacked, gxl, gxl_bstride, C0, mask, mask_bstride, mask_slope, B, D, H, W, s);
        else sd_launch<2, 2>(dz, dz_bstride, Cout, wpacked, gxl, gxl_bstride, C0, mask, mask_bstride, mask_slope, B, D, H, W, s);
    } else {
        if (lay) sd_launch<1, 2, true>(dz, dz_bstride, Cout, wpacked, gxl, gxl_bstride, C0, mask, mask_bstride, mask_slope, B, D, H, W, s);
        else sd_launch<1, 2>(dz, dz_bstride, Cout, wpacked, gxl, gxl_bstride, C0, mask, mask_bstride, mask_slope, B, D, H, W, s);
    }
    return vxm_check_launch("vxm_conv3d_k3_s3u_bwd_low");
}

/* both backward-data products of a cat([upsample(x0), x1]) layer from one staging of dz (k_s3u_bwd_pc): the gradient of the low-resolution x0
 * (what _bwd_low computes) and of the skip tensor x1 (what vxm_conv3d_k3_s3_fwd computes with the flipped operator of w[:, C0:]) */
int vxm_conv3d_k3_s3u_bwd_data_ok(int C0, int C1, int Cout, int B, int D, int H, int W, int pieces) {
    // OFF by default: measured at 160 x 192 x 224 (tools/s3_bench.py --only dlow) the one launch takes 1.11 - 1.14 ms against 1.13 - 1.26 ms for
    // k_s3u_dlow + k_s3p_conv, and in the replayed step, beside the weight gradients of the second stream, it is SLOWER (91.7 against 92.5
    // pairs/s, same box).  Why: with 16 skip channels a B fragment read from LDS (1 KB per wave) feeds three MFMAs -- 128 LDS cycles against 96
    // matrix-pipe cycles per K-step row over the CU, so the skip product is bound by LDS reads whatever feeds the tile, and the 4 x 4 x 32 tiles the
    // double buffer allows are re-staged with 2.4 x halo.  VXM_S3U_BWD_PC=1 switches it on (tests call the entry point directly).
    static const bool on = [] { const char* e = getenv("VXM_S3U_BWD_PC"); return e && e[0] == '1'; }();
    if (!on || !vxm_conv3d_k3_s3u_bwd_low_ok(C0, Cout, B, D, H, W, pieces)) return 0;
    if (C0 > 32 || C1 <= 0 || C1 > 32 || C1 % 8 || Cout % 8 || (W & 3)) return 0;
    if ((long long)C1 * D * H * W >= (1ll << 29)) return 0;
    return 1;
}

size_t vxm_conv3d_k3_s3u_bwd_skip_packed_bytes(int C1, int Cout, int pieces) {
    if (C1 <= 0 || Cout <= 0 || Cout % 8 || pieces != 2) return 0;
    const int NCT = su_nct(C1);
    return ((size_t)(Cout / 8) * (7 * 2 * NCT * 64) + 1) * 16;
}

/* w: [Cout][Cin][27] (reference layout); the operator of the skip channels [C0, C0 + C1) transposed and mirrored, in the forward kernel's skip-chunk format */
int vxm_conv3d_k3_s3u_bwd_skip_pack_weights(const float* w, void* wpacked, int C0, int C1, int Cout, int pieces, void* stream) {
    VXM_REQUIRE(w && wpacked, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3u_bwd_skip_pack_weights: null pointer");
    VXM_REQUIRE(C0 >= 0 && C1 > 0 && Cout > 0 && Cout % 8 == 0 && pieces == 2 && (reinterpret_cast<uintptr_t>(wpacked) & 15) == 0, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_bwd_skip_pack_weights: skip channels [%d, %d) of %d outputs (multiple of 8), pieces %d (2), 16-byte aligned destination",
                C0, C0 + C1, Cout, pieces);
    const int NCT = su_nct(C1);
    SuPack jb = {w, static_cast<u32x4*>(wpacked), 0, Cout, C1, NCT, 2, 0, Cout / 8, 1, (unsigned)((size_t)(Cout / 8) * (7 * 2 * NCT * 64)), 2, C0, C0 + C1};
    VXM_REQUIRE(C1 <= 16 * NCT, VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3u_bwd_skip_pack_weights: at most 32 skip channels (got %d)", C1);
    hipStream_t s = VXM_STREAM(stream);
    (void)hipMemsetAsync(jb.wp + jb.words, 0, 16, s);
    const unsigned blocks = (jb.words + 255) / 256;
    hipLaunchKernelGGL(k_s3u_pack<0>, dim3(blocks), dim3(256), 0, s, jb);
    hipLaunchKernelGGL(k_s3u_pack<1>, dim3(blocks), dim3(256), 0, s, jb);
    return vxm_check_launch("vxm_conv3d_k3_s3u_bwd_skip_pack_weights");
}

int vxm_conv3d_k3_s3u_bwd_data(const float* dz, int64_t dz_bstride, int Cout, const void* wlow, float* gxl, int64_t gxl_bstride, int C0, const float* mask,
                               int64_t mask_bstride, float mask_slope, const void* wskip, float* gx1, int64_t gx1_bstride, int C1, int B, int D, int H, int W,
                               int pieces_and_layout, void* stream) {
    const int pieces = pieces_and_layout & 0xff, lay = pieces_and_layout & ~0xff;
    VXM_REQUIRE(dz && wlow && gxl && wskip && gx1, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3u_bwd_data: null pointer");
    VXM_REQUIRE(lay == 0 || lay == VXM_S3_IN0_BLOCKED, VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3u_bwd_data: layout flags 0x%x (only dz, the first operand, may be channel-blocked)", lay);
    if (int e = check_conv("vxm_conv3d_k3_s3u_bwd_data", C0, C1, 1, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(pieces == 2 && C0 % 8 == 0 && C0 <= 32 && C1 % 8 == 0 && C1 <= 32 && Cout % 8 == 0 && W % 4 == 0 && (dz_bstride & 1) == 0 &&
                    ((reinterpret_cast<uintptr_t>(wlow) | reinterpret_cast<uintptr_t>(wskip)) & 15) == 0, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_bwd_data: fp16 pieces, %d + %d -> %d channels (segments <= 32, multiples of 8), W = %d (multiple of 4), 16-byte aligned operators",
                C0, C1, Cout, W);
    hipStream_t s = VXM_STREAM(stream);
    const int nl = su_nct(C0), ns = su_nct(C1);
#define SBP_GO(NL, NS)                                                                                                                                      \
    do {                                                                                                                                                    \
        if (lay) sbp_launch<NL, NS, true>(dz, dz_bstride, Cout, wlow, gxl, gxl_bstride, C0, mask, mask_bstride, mask_slope, wskip, gx1, gx1_bstride, C1, B, D, H, W, s); \
        else sbp_launch<NL, NS, false>(dz, dz_bstride, Cout, wlow, gxl, gxl_bstride, C0, mask, mask_bstride, mask_slope, wskip, gx1, gx1_bstride, C1, B, D, H, W, s);  \
    } while (0)
    if (nl == 2 && ns == 2) SBP_GO(2, 2); else if (nl == 2) SBP_GO(2, 1); else if (ns == 2) SBP_GO(1, 2); else SBP_GO(1, 1);
#undef SBP_GO
    return vxm_check_launch("vxm_conv3d_k3_s3u_bwd_data");
}

/* backward-weight of the upsampled segment, collapsed, on the fp16 scheme (k_s3u_bww): gw[co][0:C0][tap] inside a [Cout][gw_cin][27] array */
int vxm_conv3d_k3_s3u_bwd_weight_ok(int C0, int Cout, int B, int D, int H, int W, int pieces) {
    if (pieces != 2 || (C0 != 16 && C0 != 32) || Cout <= 0 || Cout % 16 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    if (((D | H) & 1) || (W & 3)) return 0;                       // even extents; W / 2 even (the x0 staging loads voxel pairs)
    if ((long long)(C0 > Cout ? C0 : Cout) * D * H * W >= (1ll << 29)) return 0;
    if (Cout / 16 > su_cus()) return 0;
    // measured (tools/s3_bench.py, same box): 0.95 against 1.18 ms at 160x192x224 (6720 tiles of 4 x 2 x 16 low-res voxels), but 0.235 against
    // 0.204 ms at 80x96x112 (840 tiles: a block's task ranges are too short for its two-phase prologue) -- the half-resolution level keeps
    // the fp32-MFMA kernel.  VXM_S3U_BWW_MIN_TILES overrides (tests).
    static const long long min_tiles = [] { const char* e = getenv("VXM_S3U_BWW_MIN_TILES"); return e ? atoll(e) : 4096ll; }();
    const long long ntiles = (long long)B * ((D / 2 + 3) / 4) * ((H / 2 + 1) / 2) * ((W / 2 + 15) / 16);
    return ntiles >= min_tiles ? 1 : 0;
}

size_t vxm_conv3d_k3_s3u_bwd_weight_workspace_bytes(int C0, int Cout, int B, int D, int H, int W) {
    if (C0 <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const int NCOT = (Cout + 15) / 16, NCI = (C0 + 15) / 16;
    const UwTasks tk = uw_tasks(NCOT, B, D, H, W);
    return (size_t)tk.NBLK * NCOT * 64 * 16 * 16 * NCI * sizeof(float);
}

int vxm_conv3d_k3_s3u_bwd_weight(const float* x0, int C0, int64_t x0_bstride, const float* dz, int64_t dz_bstride, int Cout, float* gw, int gw_cin,
                                 void* work, size_t work_bytes, int B, int D, int H, int W, int pieces_and_layout, void* stream) {
    const int pieces = pieces_and_layout & 0xff, phase = pieces_and_layout & (VXM_S3_BW_CONTRACT_ONLY | VXM_S3_BW_REDUCE_ONLY);
    const int lay = pieces_and_layout & ~0xff & ~(VXM_S3_BW_CONTRACT_ONLY | VXM_S3_BW_REDUCE_ONLY);
    VXM_REQUIRE(phase != (VXM_S3_BW_CONTRACT_ONLY | VXM_S3_BW_REDUCE_ONLY), VXM_ERR_BAD_SHAPE, "vxm_conv3d_k3_s3u_bwd_weight: both phase flags set");
    VXM_REQUIRE(x0 && dz && gw && work, VXM_ERR_NULL_POINTER, "vxm_conv3d_k3_s3u_bwd_weight: null pointer");
    VXM_REQUIRE(lay == 0 || lay == VXM_S3_IN1_BLOCKED, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_bwd_weight: layout flags 0x%x (only dz, the second operand, may be channel-blocked)", lay);
    if (int e = check_conv("vxm_conv3d_k3_s3u_bwd_weight", C0, 0, 1, Cout, B, D, H, W)) return e;
    VXM_REQUIRE(pieces == 2 && (C0 == 16 || C0 == 32) && Cout % 16 == 0 && C0 <= gw_cin && W % 4 == 0, VXM_ERR_BAD_SHAPE,
                "vxm_conv3d_k3_s3u_bwd_weight: %d upsampled channels (16 or 32) of %d, %d outputs (multiple of 16), W = %d (multiple of 4), pieces %d (2)",
                C0, gw_cin, Cout, W, pieces);
    const int NCOT = Cout / 16, NCI = C0 / 16;
    const UwTasks tk = uw_tasks(NCOT, B, D, H, W);
    VXM_REQUIRE(work_bytes >= (size_t)tk.NBLK * NCOT * 64 * 16 * 16 * NCI * sizeof(float), VXM_ERR_WORKSPACE, "vxm_conv3d_k3_s3u_bwd_weight: workspace too small");
    static const bool attr = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3u_bww<1>), hipFuncAttributeMaxDynamicSharedMemorySize, UwCfg<1>::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_s3u_bww<2>), hipFuncAttributeMaxDynamicSharedMemorySize, UwCfg<2>::LDS_BYTES);
        return true;
    }();
    (void)attr;
    hipStream_t s = VXM_STREAM(stream);
    float* part = static_cast<float*>(work);
    const char* te = getenv("VXM_S3_BW_TASKS");
    // (round-robin order: -8 .. -16 % at 160x192x224, +3 % at 80x96x112 -- same-box A/B, profiles/r04r_bw_task_order.txt)
    const int task_rr = ((te && te[0] == 'r' && te[1] == 'a') || (long long)D * H * W < (1ll << 21)) ? 0 : 1;
    if (phase == VXM_S3_BW_REDUCE_ONLY) {}
    else if (NCI == 2)
        hipLaunchKernelGGL(k_s3u_bww<2>, dim3(tk.NBLK * NCOT), dim3(UW_THREADS), UwCfg<2>::LDS_BYTES, s, x0, (long long)x0_bstride, C0, dz,
                           (long long)dz_bstride, Cout, part, D, H, W, tk.NBLK, tk.ncol, tk.nseg, tk.seg_len, tk.nh, tk.nw, task_rr, lay);
    else
        hipLaunchKernelGGL(k_s3u_bww<1>, dim3(tk.NBLK * NCOT), dim3(UW_THREADS), UwCfg<1>::LDS_BYTES, s, x0, (long long)x0_bstride, C0, dz,
                           (long long)dz_bstride, Cout, part, D, H, W, tk.NBLK, tk.ncol, tk.nseg, tk.seg_len, tk.nh, tk.nw, task_rr, lay);
    if (phase != VXM_S3_BW_CONTRACT_ONLY)
        hipLaunchKernelGGL(k_s3u_bww_reduce, dim3((unsigned)(NCOT * 27 * (16 * 16 * NCI / 64))), dim3(1024), 0, s, part, gw, C0, Cout, gw_cin, NCI, NCOT, tk.NBLK);
    return vxm_check_launch("vxm_conv3d_k3_s3u_bwd_weight");
}

}  // extern "C"
