// Backward-weight of a full-resolution, non-upsampled 16-channel-multiple segment ("row-sliding" kernel, round 2).
//
// Replaces `convolution_backward` (weight, bias) of ConvBlock / skip segments (voxelmorph/torch/networks.py:290-305 autograd twin)
// where k_conv3d_k3_bwd_weight_vec ran at 58-68 % of the fp32 MFMA peak.  Same product, gW[co][ci][tap] = sum_v dZ[co][v] X[ci][v+tap-1]
// on v_mfma_f32_16x16x4_f32 (M = 16 output channels, N = 16 input channels, K = 4 voxels of a W row; exact fp32), restructured
// after the bf16 kernel of conv_bf16.hip so that the matrix pipe is the only busy unit:
//  * a wave = (depth slice of a 4 x 8 x 16 voxel tile, kd) keeps the 9 (kh, kw) taps x NCO output-channel tiles of its kd in
//    36 NCO accumulator VGPRs over ALL its tiles and slides over the 10 haloed X rows: the 12 B operands of a row (3 kw shifts x
//    4 K-steps) serve kh = 0, 1, 2 with the A operands of output rows r, r-1, r-2 -- 16 + 12 NCO ds_read_b32 and no address
//    arithmetic per 36 NCO MFMAs (every LDS offset is an immediate; channel strides = 2 mod 32 make the reads conflict-free);
//  * 12 waves per block, one block per CU: 3 waves on every SIMD (a 6- or 16-wave block leaves SIMDs unevenly loaded);
//  * the block walks DOWN THE DEPTH of a (b, th, tw) column, the haloed planes live in a 6-slot LDS ring (a tile fetches 4 new
//    planes of its 6), and the loads of tile t+1 are in flight in registers under the 288 NCO MFMAs per wave of tile t;
//    per-lane staging offsets are rebuilt once per column;
//  * the bias gradient is a VALU side-sum of the A operands; partials per (block, depth slice) are summed in a fixed order.
#include "conv_common.h"

namespace {

constexpr int RS_TD = 4, RS_TH = 8, RS_TW = 16, RS_WAVES = RS_TD * 3, RS_THREADS = 64 * RS_WAVES;
constexpr int RS_RING = 6, RS_ROWF = 20;                                  // haloed row: halo at 1, interior at 2..17, halo at 18
constexpr int RS_CSX = RS_RING * (RS_TH + 2) * RS_ROWF + 18;              // 1218 = 2 mod 32: channel stride of the X ring
constexpr int RS_CSZ = RS_TD * RS_TH * RS_TW + 2;                         // 514 = 2 mod 32: channel stride of the dZ tile
constexpr int rs_lds_floats(int nco) { return 16 * RS_CSX + 16 * nco * RS_CSZ; }

struct RsTasks { int ncol, nseg, seg_len, nd, nh, nw; };

template <int NCO>
__global__ void __launch_bounds__(RS_THREADS, 3) k_conv3d_k3_bwd_weight_rs(const float* __restrict__ x, long long x_bs, const float* __restrict__ dz,
                                                                           long long dz_bs, int Cout, float* __restrict__ part, float* __restrict__ bpart,
                                                                           int D, int H, int W, int NBLK, RsTasks tk) {
    VXM_DYN_SMEM(float, smem);
    float* const Xs = smem;                      // [16 ci][6 slots][10 rows][20]
    float* const Zs = smem + 16 * RS_CSX;        // [16 NCO co][4 ds][8 rows][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ds = wave / 3, kd = wave - 3 * ds;
    const int kq = lane >> 4, n = lane & 15;
    const int q = blockIdx.y;                     // 16-channel chunk of the segment
    const int ntask = tk.ncol * tk.nseg;
    const int k_lo = (int)((long long)ntask * blockIdx.x / NBLK), k_hi = (int)((long long)ntask * (blockIdx.x + 1) / NBLK);
    const int HW = H * W, V = D * HW;

    f32x4 acc[3][3][NCO];
    float bsum[NCO];
#pragma unroll
    for (int ct = 0; ct < NCO; ++ct) {
        bsum[ct] = 0.0f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc[kh][kw][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // staging roles, fixed for the kernel: X interior = dwordx4 pieces (ci, plane 0..3, row, quarter), X halo = dwords (ci, plane,
    // row, side), dZ = dwordx4 pieces (co, slice, row, quarter)
    constexpr int NXQ = 16 * 4 * (RS_TH + 2) * 4, NXI = (NXQ + RS_THREADS - 1) / RS_THREADS;          // 2560 -> 4
    constexpr int NXH = 16 * 4 * (RS_TH + 2) * 2, NHI = (NXH + RS_THREADS - 1) / RS_THREADS;          // 1280 -> 2
    constexpr int NZQ = 16 * NCO * RS_TD * RS_TH * 4, NZI = (NZQ + RS_THREADS - 1) / RS_THREADS;      // 2048 NCO -> 3 / 6
    f32x4 xv[NXI], zv[NZI];
    float hv[NHI];
    int xoff[NXI], hoff[NHI], zoff[NZI];          // byte offsets (VXM_OOB: padding); bits 0-1 of xoff / hoff: plane, bits 0-1 of zoff: slice
    int xdst[NXI], hdst[NHI];                     // LDS float offsets without the ring slot
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        const int r = tid + RS_THREADS * j, q4 = r & 3, rowid = r >> 2;
        const int ci = rowid / (4 * (RS_TH + 2)), rem = rowid - ci * 4 * (RS_TH + 2), hr = rem % (RS_TH + 2);
        xdst[j] = ci * RS_CSX + hr * RS_ROWF + 2 + 4 * q4;
    }
#pragma unroll
    for (int j = 0; j < NHI; ++j) {
        const int r = tid + RS_THREADS * j, side = r & 1, rowid = r >> 1;
        const int ci = rowid / (4 * (RS_TH + 2)), rem = rowid - ci * 4 * (RS_TH + 2), hr = rem % (RS_TH + 2);
        hdst[j] = ci * RS_CSX + hr * RS_ROWF + (side ? RS_TW + 2 : 1);
    }

    for (int task = k_lo; task < k_hi; ++task) {
        const int col = task / tk.nseg, seg = task - col * tk.nseg;
        const int tw = col % tk.nw; int cq = col / tk.nw;
        const int th = cq % tk.nh; const int b = cq / tk.nh;
        const int td0 = seg * tk.seg_len, ntile = min(tk.seg_len, tk.nd - td0);
        const int dbase = td0 * RS_TD, h0 = th * RS_TH, w0 = tw * RS_TW;
        const __amdgpu_buffer_rsrc_t rx = vxm_rsrc(x + (size_t)b * x_bs + (size_t)q * 16 * V, 16u * (unsigned)V * 4u);
        const __amdgpu_buffer_rsrc_t rz = vxm_rsrc(dz + (size_t)b * dz_bs, (unsigned)Cout * (unsigned)V * 4u);
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const int r = tid + RS_THREADS * j, q4 = r & 3, rowid = r >> 2;
            const int ci = rowid / (4 * (RS_TH + 2)), rem = rowid - ci * 4 * (RS_TH + 2), pl = rem / (RS_TH + 2), hr = rem - pl * (RS_TH + 2);
            const int gh = h0 - 1 + hr;
            const bool ok = r < NXQ && (unsigned)gh < (unsigned)H;
            xoff[j] = !ok ? VXM_OOB : (((ci * D + pl) * H + gh) * W + w0 + 4 * q4) << 2 | pl;
        }
#pragma unroll
        for (int j = 0; j < NHI; ++j) {
            const int r = tid + RS_THREADS * j, side = r & 1, rowid = r >> 1;
            const int ci = rowid / (4 * (RS_TH + 2)), rem = rowid - ci * 4 * (RS_TH + 2), pl = rem / (RS_TH + 2), hr = rem - pl * (RS_TH + 2);
            const int gh = h0 - 1 + hr, gw = side ? w0 + RS_TW : w0 - 1;
            const bool ok = r < NXH && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
            hoff[j] = !ok ? VXM_OOB : (((ci * D + pl) * H + gh) * W + gw) << 2 | pl;
        }
#pragma unroll
        for (int j = 0; j < NZI; ++j) {
            const int r = tid + RS_THREADS * j, q4 = r & 3, rowid = r >> 2;
            const int co = rowid / (RS_TD * RS_TH), rem = rowid - co * RS_TD * RS_TH, zd = rem / RS_TH, zh = rem - zd * RS_TH;
            const bool ok = r < NZQ && co < Cout && h0 + zh < H;
            zoff[j] = !ok ? VXM_OOB : (((co * D + zd) * H + h0 + zh) * W + w0 + 4 * q4) << 2 | zd;
        }

        // planes p0 .. p0 + 3 of this task (plane p = global depth dbase - 1 + p; only the first `np` of them) -> registers
        auto load_x = [&](int p0, int np) __attribute__((always_inline)) {
            const int gd0 = dbase - 1 + p0;                                // wave-uniform; -1 for the first planes of the volume
            const int g0 = min(max(gd0, 0), D - 1);                        // the scalar offset stays inside the tensor ...
            const int adj = (gd0 - g0) * HW * 4;                           // ... the rest rides in the lane offset of the valid lanes
#pragma unroll
            for (int j = 0; j < NXI; ++j) {
                const int pl = xoff[j] & 3;
                const bool ok = pl < np && (unsigned)(gd0 + pl) < (unsigned)D;
                xv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (ok && xoff[j] >= 0) ? (xoff[j] & ~3) + adj : VXM_OOB, (g0 * HW) << 2, 0));
            }
#pragma unroll
            for (int j = 0; j < NHI; ++j) {
                const int pl = hoff[j] & 3;
                const bool ok = pl < np && (unsigned)(gd0 + pl) < (unsigned)D;
                hv[j] = vxm_bload(rx, (ok && hoff[j] >= 0) ? (hoff[j] & ~3) + adj : VXM_OOB, (g0 * HW) << 2);
            }
        };
        auto store_x = [&](int p0, int np) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NXI; ++j) {
                const int r = tid + RS_THREADS * j, rowid = r >> 2, plr = (rowid % (4 * (RS_TH + 2))) / (RS_TH + 2);
                if (r < NXQ && plr < np) {
                    float* d = Xs + xdst[j] + ((p0 + plr) % RS_RING) * ((RS_TH + 2) * RS_ROWF);
                    *reinterpret_cast<f32x2*>(d) = (f32x2){xv[j].x, xv[j].y};
                    *reinterpret_cast<f32x2*>(d + 2) = (f32x2){xv[j].z, xv[j].w};
                }
            }
#pragma unroll
            for (int j = 0; j < NHI; ++j) {
                const int r = tid + RS_THREADS * j, rowid = r >> 1, plr = (rowid % (4 * (RS_TH + 2))) / (RS_TH + 2);
                if (r < NXH && plr < np) Xs[hdst[j] + ((p0 + plr) % RS_RING) * ((RS_TH + 2) * RS_ROWF)] = hv[j];
            }
        };
        auto load_z = [&](int t) __attribute__((always_inline)) {
            const int gd0 = dbase + t * RS_TD;
#pragma unroll
            for (int j = 0; j < NZI; ++j) {
                const bool ok = gd0 + (zoff[j] & 3) < D;
                zv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, ok ? (zoff[j] & ~3) : VXM_OOB, (gd0 * HW) << 2, 0));
            }
        };
        auto store_z = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NZI; ++j) {
                const int r = tid + RS_THREADS * j, q4 = r & 3, rowid = r >> 2;
                if (r < NZQ) {
                    const int co = rowid / (RS_TD * RS_TH), rem = rowid - co * RS_TD * RS_TH;
                    float* d = Zs + co * RS_CSZ + rem * RS_TW + 4 * q4;
                    *reinterpret_cast<f32x2*>(d) = (f32x2){zv[j].x, zv[j].y};
                    *reinterpret_cast<f32x2*>(d + 2) = (f32x2){zv[j].z, zv[j].w};
                }
            }
        };

        __syncthreads();                                        // every wave is done with the previous task
        load_x(0, 2);
        load_z(0);
        store_x(0, 2);
        load_x(2, 4);
        store_z();
        store_x(2, 4);
        __syncthreads();

        for (int t = 0; t < ntile; ++t) {
            const bool more = t + 1 < ntile;                     // wave-uniform
            if (more) { load_x(4 * t + 6, 4); load_z(t + 1); }   // in flight under the MFMAs below
            __builtin_amdgcn_sched_barrier(0);
            const float* const xp = Xs + ((4 * t + ds + kd) % RS_RING) * ((RS_TH + 2) * RS_ROWF) + n * RS_CSX + kq + 1;   // + hr 20 + 4 s + kw
            const float* const zp = Zs + n * RS_CSZ + ds * RS_TH * RS_TW + kq;                                            // + ct 16 CSZ + row 16 + 4 s
#pragma unroll
            for (int hr = 0; hr < RS_TH + 2; ++hr) {
                float bx[3][4];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int s = 0; s < 4; ++s) bx[kw][s] = xp[hr * RS_ROWF + 4 * s + kw];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int row = hr - kh;
                    if (row >= 0 && row < RS_TH) {
#pragma unroll
                        for (int ct = 0; ct < NCO; ++ct) {
                            float az[4];
#pragma unroll
                            for (int s = 0; s < 4; ++s) az[s] = zp[ct * 16 * RS_CSZ + row * RS_TW + 4 * s];
                            if (kh == 0 && kd == 0) bsum[ct] += (az[0] + az[1]) + (az[2] + az[3]);
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                                for (int s = 0; s < 4; ++s) acc[kh][kw][ct] = vxm_mfma16(az[s], bx[kw][s], acc[kh][kw][ct]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);             // operands of row hr + 1 stay behind this row's MFMAs (register budget)
            }
            __syncthreads();                                    // every wave is done reading tile t: planes 4t..4t+3 and the dZ tile are free
            if (more) { store_x(4 * t + 6, 4); store_z(); }
            __syncthreads();
        }
    }

    // ---- partials: part[blk][q][ds][tap][co 16 NCO][ci 16]; D layout: lane (kq, n) holds co = 4 kq + r, ci = n
    float* const pp = part + ((((size_t)blockIdx.x * gridDim.y + q) * RS_TD + ds) * 27) * (16 * NCO) * 16;
#pragma unroll
    for (int ct = 0; ct < NCO; ++ct) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pp[((size_t)(kd * 9 + kh * 3 + kw) * (16 * NCO) + ct * 16 + 4 * kq + r) * 16 + n] = acc[kh][kw][ct][r];
        // bias partials: lane (kq, n) summed the voxels = kq (mod 4) of output channel 16 ct + n
        if (kd == 0 && q == 0) bpart[(((size_t)blockIdx.x * RS_TD + ds) * (16 * NCO) + ct * 16 + n) * 4 + kq] = bsum[ct];
    }
}

// gw[co][ci_off + ci][tap] = sum over (block, depth slice) partials, gb[co] = sum over (block, slice, k) bias partials; fixed order
__global__ void __launch_bounds__(1024) k_rs_reduce_partials(const float* __restrict__ part, const float* __restrict__ bpart, float* __restrict__ gw,
                                                             float* __restrict__ gb, int gw_cin, int ci_off, int Cseg, int Cout, int Q, int NCO, int NBLK) {
    __shared__ float sm[16][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const int CoP = 16 * NCO, per_q = 27 * CoP * 16, nw = Q * per_q, nb = gb ? CoP : 0;
    const int e = blockIdx.x * 64 + x;
    float tot = 0.0f;
    if (e < nw) {
        const int q = e / per_q, r = e - q * per_q;
        const size_t stride_blk = (size_t)Q * RS_TD * per_q;
        const float* p = part + (size_t)q * RS_TD * per_q + r;
        const int K = RS_TD * NBLK;                              // entry k: block k / 4, slice k % 4
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int k = y;
        for (; k + 16 * 7 < K; k += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k + 16 * u;
                s[u] += p[(size_t)(kk / RS_TD) * stride_blk + (size_t)(kk % RS_TD) * per_q];
            }
        }
        for (; k < K; k += 16) s[0] += p[(size_t)(k / RS_TD) * stride_blk + (size_t)(k % RS_TD) * per_q];
        tot = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    } else if (e < nw + nb) {
        const int co = e - nw;
        const int K = RS_TD * NBLK;
        float s = 0.0f;
        for (int k = y; k < K; k += 16) {
            const float* p = bpart + ((size_t)k * CoP + co) * 4;
            s += (p[0] + p[1]) + (p[2] + p[3]);
        }
        tot = s;
    }
    sm[y][x] = tot;
    __syncthreads();
    if (y == 0 && e < nw + nb) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sm[u][x];
        const float sum = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) +
                          (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        if (e < nw) {
            const int q = e / per_q, r = e - q * per_q;
            const int tap = r / (CoP * 16), co = (r / 16) % CoP, ci = q * 16 + (r & 15);
            if (co < Cout && ci < Cseg) gw[((size_t)co * gw_cin + ci_off + ci) * 27 + tap] = sum;
        } else if (e - nw < Cout) {
            gb[e - nw] = sum;
        }
    }
}

int rs_cus() {
    static const int cus = [] {
        int dev = 0; hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
        return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }();
    return cus;
}
RsTasks rs_tasks(int Q, int B, int D, int H, int W, int& NBLK) {
    RsTasks tk;
    tk.nd = (D + RS_TD - 1) / RS_TD; tk.nh = (H + RS_TH - 1) / RS_TH; tk.nw = (W + RS_TW - 1) / RS_TW;
    tk.ncol = B * tk.nh * tk.nw;
    const int nb = rs_cus() / Q > 0 ? rs_cus() / Q : 1;
    int nseg = (10 * nb + tk.ncol - 1) / tk.ncol;              // ~10 tasks per block: <= 5 % load imbalance
    if (nseg > tk.nd) nseg = tk.nd;
    if (nseg < 1) nseg = 1;
    tk.seg_len = (tk.nd + nseg - 1) / nseg;
    tk.nseg = (tk.nd + tk.seg_len - 1) / tk.seg_len;
    const long long ntask = (long long)tk.ncol * tk.nseg;
    NBLK = (int)(nb < ntask ? nb : ntask);
    return tk;
}

}  // namespace

// VXM_BWDW_RS=0 keeps the 16-wave kernel everywhere (developer A/B switch)
bool vxm_bw_rs_enabled() {
    static const bool on = [] { const char* e = getenv("VXM_BWDW_RS"); return !(e && e[0] == '0'); }();
    return on;
}

// the row-sliding kernel takes a segment when: channels in multiples of 16 on both sides (<= 32 outputs), 16-voxel rows, 16-byte aligned
// rows, and enough tiles for one block per CU to walk several of them
bool vxm_bw_rs_ok(const float* x, int64_t x_bs, int Cseg, const float* dz, int64_t dz_bs, int Cout, int B, int D, int H, int W) {
    if (!vxm_bw_rs_enabled() || bw_force_generic()) return false;
    if (Cseg <= 0 || Cseg % 16 || Cout % 16 || Cout > 32 || (W % RS_TW) || !al16(x) || !al16(dz) || (x_bs & 3) || (dz_bs & 3)) return false;
    if ((((long long)D * H * W) & 3) != 0 || (long long)(Cseg > Cout ? Cseg : Cout) * D * H * W >= (1ll << 29)) return false;
    const long long tiles = (long long)B * ((D + RS_TD - 1) / RS_TD) * ((H + RS_TH - 1) / RS_TH) * (W / RS_TW);
    return tiles >= 4ll * rs_cus() || tiles * (Cseg / 16) >= wide_min_tiles() * 16;      // tests lower wide_min_tiles to force it on small volumes
}

size_t vxm_bw_rs_workspace_floats(int Cseg, int Cout, int B, int D, int H, int W) {
    const int Q = Cseg / 16, NCO = (Cout + 15) / 16;
    if (Q <= 0) return 0;
    int NBLK = 1;
    (void)rs_tasks(Q, B, D, H, W, NBLK);
    return (size_t)NBLK * Q * RS_TD * 27 * (16 * NCO) * 16 + (size_t)NBLK * RS_TD * (16 * NCO) * 4 + 64;
}

// gw[:, ci_off : ci_off + Cseg] (row stride gw_cin) and, if gb, the bias gradient from one segment x [B][Cseg][D][H][W]
void vxm_bw_rs_launch(const float* x, int64_t x_bs, int Cseg, const float* dz, int64_t dz_bs, int Cout, float* gw, int gw_cin, int ci_off, float* gb,
                      float* work, int B, int D, int H, int W, hipStream_t s) {
    const int Q = Cseg / 16, NCO = (Cout + 15) / 16;
    int NBLK = 1;
    const RsTasks tk = rs_tasks(Q, B, D, H, W, NBLK);
    float* part = work;
    float* bpart = work + (size_t)NBLK * Q * RS_TD * 27 * (16 * NCO) * 16;
    auto launch = [&](auto kern, int lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(NBLK, Q), dim3(RS_THREADS), lds, s, x, (long long)x_bs, dz, (long long)dz_bs, Cout, part, bpart, D, H, W, NBLK, tk);
    };
    if (NCO == 1) launch(k_conv3d_k3_bwd_weight_rs<1>, rs_lds_floats(1) * 4);
    else launch(k_conv3d_k3_bwd_weight_rs<2>, rs_lds_floats(2) * 4);
    const int n = Q * 27 * 16 * NCO * 16 + (gb ? 16 * NCO : 0);
    hipLaunchKernelGGL(k_rs_reduce_partials, dim3(vxm_blocks(n, 64)), dim3(1024), 0, s, part, bpart, gw, gb, gw_cin, ci_off, Cseg, Cout, Q, NCO, NBLK);
}
