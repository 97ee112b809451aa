/* libvxm_hip.so — C ABI of the MI355X-native VxmDense hot path (gfx950 only).
 *
 * The reference (voxelmorph/voxelmorph @ 0.2) has no FFI layer: its "operator API" for this
 * path is the Python class surface in voxelmorph/torch/{layers,networks,losses}.py, every
 * method of which expands into PyTorch ATen calls.  Each entry point below replaces one such
 * ATen op chain (reference file:line given per function, paths relative to the reference
 * root).  The Python host mirror (voxelmorph_amd/torch/*) binds these with ctypes; the stub a
 * reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (unless stated) owned by the caller; the library
 *    never allocates, frees or retains device memory.  Workspace sizes come from the
 *    *_workspace_bytes() queries.
 *  - tensors are NCDHW contiguous like the reference's; arguments named
 *    *_bstride are the element stride between batch samples, so that a channel slice of a
 *    larger buffer can be passed without a copy.
 *  - `stream` is a hipStream_t; all calls are asynchronous on it and never synchronise.
 *  - return value: 0 = VXM_OK, otherwise a vxm_status; vxm_last_error_string() describes the
 *    last failure on the calling thread.  Nothing throws or aborts across the ABI.
 */
#ifndef VXM_HIP_H
#define VXM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    VXM_OK = 0,
    VXM_ERR_BAD_SHAPE = 1,     /* mirrors the reference asserts (layers.py:59, networks.py:51,196) */
    VXM_ERR_UNSUPPORTED = 2,
    VXM_ERR_NULL_POINTER = 3,
    VXM_ERR_WORKSPACE = 4,
    VXM_ERR_HIP = 5            /* hipGetLastError() != hipSuccess after a launch */
} vxm_status;

enum { VXM_INTERP_LINEAR = 0, VXM_INTERP_NEAREST = 1 };   /* SpatialTransformer mode, layers.py:11 */
enum { VXM_PENALTY_L1 = 0, VXM_PENALTY_L2 = 1 };          /* Grad penalty, losses.py:98 */

int vxm_version(void);                 /* 10000 major + 100 minor + patch of this ABI: 501 = 0.5.1 (round 6; every 0.4 / 0.5.0 entry point kept) */
const char* vxm_last_error_string(void);

/* ---- the two umbrella names SURVEY.md section 8b lists.  Every op has its own *_workspace_bytes() query next to it; this one dispatches
 * on an op code (0 for an unknown op or a shape the op refuses).  Cin / Cout are those of the FORWARD layer. */
enum { VXM_WS_CONV_BWD_WEIGHT = 1, VXM_WS_S3_BWD_WEIGHT = 2, VXM_WS_S3U_BWD_WEIGHT = 3, VXM_WS_BF16_BWD_WEIGHT = 4, VXM_WS_CONV_BWD_DATA = 5,
       VXM_WS_VECINT_BWD = 6 };
size_t vxm_workspace_bytes(int op, int Cin, int Cout, int B, int D, int H, int W);
/* convolution_backward w.r.t. the input (autograd twin of networks.py:299) for the input channels [ci_lo, ci_lo + ci_n) of w [Cout][Cw_in][27]:
 * gx [B][ci_n][D][H][W] = adjoint conv of dz [B][Cout][D][H][W], times LeakyReLU'(mask) when mask != NULL (the fused leaky_relu_backward of
 * the previous ConvBlock).  It IS vxm_conv3d_k3_fwd on the transposed / flipped operator, which this call packs into wpacked_scratch
 * (vxm_workspace_bytes(VXM_WS_CONV_BWD_DATA, ci_n, Cout, ...) bytes) first; callers that keep the packed operator (the fused U-Net engine,
 * the split kernels' vxm_conv3d_k3_s3_fwd with a transpose_flip pack job) call the forward entry points directly. */
int vxm_conv3d_k3_bwd_data(const float* dz, int Cout, int64_t dz_bstride, const float* w, int Cw_in, int ci_lo, int ci_n, float* wpacked_scratch,
                           float* gx, int64_t gx_bstride, const float* mask, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W,
                           void* stream);

/* ---- SpatialTransformer.forward, layers.py:30-48 (add + normalise + permute + index +
 * grid_sampler_3d, align_corners=True, padding zeros).  out[b,c,p] = sample(src[b,c], p+flow[b,:,p]).
 * mode nearest is bit-exact with the reference (round-half-even after its fp32 normalise /
 * un-normalise round trip). */
int vxm_warp3d_fwd(const float* src, const float* flow, float* out, int B, int C, int D, int H, int W,
                   int mode, void* stream);
/* backward of the above: gflow [B,3,D,H,W] (nullable), gsrc [B,C,D,H,W] (nullable, overwritten). */
int vxm_warp3d_bwd(const float* src, const float* flow, const float* gout, float* gsrc, float* gflow,
                   int B, int C, int D, int H, int W, int mode, void* stream);

/* ---- `fullsize` + the final SpatialTransformer as one kernel, networks.py:275-280 (pos_flow = fullsize(integrate(v)); y = transformer(source,
 * pos_flow)) with layers.py:85-97 and :30-48: out[b,c,p] = sample(src[b,c], p + U(flow_lo)[b,:,p]), U = `factor *` + upsample_trilinear3d
 * (align_corners) of the integrated field flow_lo [B,3,lD,lH,lW].  The full-resolution displacement is evaluated in registers (the resize
 * kernel's expression, then the warp kernel's coordinate arithmetic: results bit-identical to vxm_resize3d_fwd + vxm_warp3d_fwd) and only
 * written when pos_flow != NULL (registration=True returns it).  vxm_warp3d_up_ok: the field is at most ~half as fine as the image per axis.
 * _bwd: gflow_lo = dL/dflow_lo; work: B*3*D*H*W floats (the gradient w.r.t. the full-resolution displacement, consumed by vxm_resize3d_bwd
 * inside the call).  src receives no gradient here (the image inputs of the path need none; callers that want one take the two-kernel path). */
int vxm_warp3d_up_ok(int D, int H, int W, int lD, int lH, int lW);
int vxm_warp3d_up_fwd(const float* src, const float* flow_lo, float* out, float* pos_flow, int B, int C, int D, int H, int W,
                      int lD, int lH, int lW, float factor, int mode, void* stream);
int vxm_warp3d_up_bwd(const float* src, const float* flow_lo, const float* gout, float* gflow_lo, float* work, size_t work_bytes,
                      int B, int C, int D, int H, int W, int lD, int lH, int lW, float factor, int mode, void* stream);

/* ---- VecInt.forward, layers.py:64-68: v0 = vec/2^n; v_{k+1} = v_k + warp(v_k, v_k).
 * steps: [nsteps][B,3,D,H,W]; steps[k] receives v_{k+1}; the result is steps[nsteps-1]. */
int vxm_vecint_fwd(const float* vec, float* steps, int B, int D, int H, int W, int nsteps, void* stream);
/* backward: gout = dL/dv_n; gvec = dL/dvec.  work: 2*B*3*D*H*W + VXM_VECINT_WORK_EXTRA floats of scratch (two gradient buffers and
 * the per-step statistics of the voxels displaced by a voxel or more: they are scattered by a second, deterministic pass -- per output
 * tile into 64-bit fixed-point LDS accumulators -- so the result is bit-reproducible; only steps that displace a voxel by more than 24
 * voxels fall back to global float atomics).  nsteps < 31. */
#define VXM_VECINT_WORK_EXTRA 128
/* work_bytes: size of `work` (VXM_ERR_WORKSPACE when below 2*B*3*D*H*W + VXM_VECINT_WORK_EXTRA floats: the call clears the statistics BEHIND
 * the two gradient buffers, so a caller that sized the scratch by hand used to overrun silently).  With vxm_workspace_bytes(VXM_WS_VECINT_BWD,
 * 0, 0, B, D, H, W) bytes the gather also records each 4 x 8 x 32 tile's largest far displacement and the deterministic pass grows an output
 * tile by what the sender tiles around it need (a handful of outliers no longer make every tile walk the batch maximum); with less it uses
 * the step's global maximum.  Same results either way. */
int vxm_vecint_bwd_ws(const float* vec, const float* steps, const float* gout, float* gvec, float* work, size_t work_bytes,
                      int B, int D, int H, int W, int nsteps, void* stream);
/* ABI 0.4 name, kept: the same call for a caller that vouches for (2*B*3*D*H*W + VXM_VECINT_WORK_EXTRA) floats of scratch. */
int vxm_vecint_bwd(const float* vec, const float* steps, const float* gout, float* gvec, float* work,
                   int B, int D, int H, int W, int nsteps, void* stream);

/* ---- ResizeTransform.forward, layers.py:85-97 (upsample_trilinear3d align_corners=True and the
 * `factor *` rescale, before the resize when factor>1, after when factor<1). */
int vxm_resize3d_fwd(const float* x, float* out, int B, int C, int D, int H, int W, int oD, int oH, int oW,
                     float factor, void* stream);
int vxm_resize3d_bwd(const float* gout, float* gx, int B, int C, int D, int H, int W, int oD, int oH, int oW,
                     float factor, void* stream);

/* ---- ConvBlock / flow conv, networks.py:299-305,211,257: 3x3x3, stride 1, pad 1 conv + bias +
 * LeakyReLU(act_slope) (act_slope = 1 -> no activation).  fp32 MFMA implicit GEMM.
 * The input is a *virtual concat* of up to two channel segments, so that Unet.forward's
 * `cat([upsample(x), skip])` (networks.py:137-138) is never materialised:
 *   segment 0: x0 [B,C0,*] (if x0_up != 0 it is stored at half resolution [D/2,H/2,W/2] and
 *              read through nearest x2 upsampling), segment 1: x1 [B,C1,D,H,W] (nullable, C1=0).
 * wpacked comes from vxm_conv3d_k3_pack_weights.  If mask_src != NULL the result is multiplied
 * by LeakyReLU'(mask_src) (slope mask_slope) — used when this launch computes a backward-data
 * product that feeds the previous ConvBlock (fused leaky_relu_backward). */
size_t vxm_conv3d_k3_packed_elems(int Cin, int Cout);
/* w: [Cout,Cin,3,3,3] (reference layout).  transpose_flip=0: forward operator; 1: the adjoint
 * (backward-data) operator, i.e. w'[ci][co][t] = w[co][ci][26-t], packed as a Cout->Cin conv. */
int vxm_conv3d_k3_pack_weights(const float* w, float* wpacked, int Cin, int Cout, int transpose_flip,
                               void* stream);
/* the same for the input-channel sub-range [ci_lo, ci_lo + ci_n) of w (row stride Cw_in): the operator on those inputs, or with
 * transpose_flip its adjoint onto them (backward-data of one segment of a concatenated input) -- no sliced copy of w needed */
int vxm_conv3d_k3_pack_weights_range(const float* w, float* wpacked, int Cw_in, int Cout, int ci_lo, int ci_n, int transpose_flip,
                                     void* stream);
int vxm_conv3d_k3_fwd(const float* x0, int C0, int64_t x0_bstride, int x0_up,
                      const float* x1, int C1, int64_t x1_bstride,
                      const float* wpacked, const float* bias, float* y, int64_t y_bstride, int Cout,
                      float act_slope, const float* mask_src, int64_t mask_bstride, float mask_slope,
                      int B, int D, int H, int W, void* stream);
/* ConvBlock over cat([upsample2(x0), x1]) (networks.py:137-138,299-305) with the upsampled segment evaluated at LOW
 * resolution: per output parity the 27 taps over the nearest-upsampled x0 collapse onto 2x2x2 low-resolution inputs
 * with pre-summed weights (8 instead of 27 MACs per channel pair and voxel).  x0 [B,C0,D/2,H/2,W/2], x1 [B,C1,D,H,W]
 * (nullable, C1 = 0), y [B,Cout,D,H,W]; weights packed by vxm_conv3d_k3_up_pack_weights from the reference layout
 * w [Cout,C0+C1,3,3,3].  vxm_conv3d_k3_up_ok tells whether the operands qualify (otherwise vxm_conv3d_k3_fwd, x0_up=1). */
int vxm_conv3d_k3_up_ok(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, float* y,
                        int Cout, int B, int D, int H, int W);
size_t vxm_conv3d_k3_up_packed_elems(int C0, int C1, int Cout);
int vxm_conv3d_k3_up_pack_weights(const float* w, float* wpacked, int C0, int C1, int Cout, void* stream);
int vxm_conv3d_k3_up_fwd(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride,
                         const float* wpacked, const float* bias, float* y, int64_t y_bstride, int Cout, float act_slope,
                         int B, int D, int H, int W, void* stream);
/* Backward-data of the upsampled segment of such a ConvBlock, straight to the HALF-resolution gradient:
 * gx_low[b,ci,m] = LeakyReLU'(mask_low[b,ci,m]) * sum_co sum_{delta in {-1,0,1,2}^3} Wt[delta][ci,co] dz[b,co,2m+delta]
 * (a stride-2 4x4x4 conv with per-axis summed kernel taps = conv backward + upsample_nearest3d_backward in one pass).
 * dz [B,Cout,D,H,W]; w [Cout,Cin,3,3,3] (reference layout; the first C0 input channels are the upsampled segment);
 * wpacked: scratch of vxm_conv3d_k3_up_bwd_low_packed_elems(C0,Cout) floats (filled by the call); gx_low, mask_low
 * [B,C0,D/2,H/2,W/2] (mask nullable). */
int vxm_conv3d_k3_up_bwd_low_ok(const float* dz, int64_t dz_bstride, int C0, int Cout, int B, int D, int H, int W);
size_t vxm_conv3d_k3_up_bwd_low_packed_elems(int C0, int Cout);
int vxm_conv3d_k3_up_bwd_low(const float* dz, int64_t dz_bstride, int Cout, const float* w, int C0, int Cin, float* wpacked,
                             float* gx_low, int64_t gx_bstride, const float* mask_low, int64_t mask_bstride, float mask_slope,
                             int B, int D, int H, int W, void* stream);
/* Forward conv with 1..4 output channels (the 16 -> 3 flow conv, networks.py:211,257) on the vector ALUs: x [B,Cin,D,H,W],
 * w [Cout,Cin,3,3,3] in the REFERENCE layout (no packing), y [B,Cout,D,H,W]; act_slope = 1: no activation.
 * vxm_conv3d_k3_fewout_ok tells whether the operands qualify (Cout <= 4, W % 4 == 0, 16-byte aligned). */
int vxm_conv3d_k3_fewout_ok(const float* x, int64_t x_bstride, float* y, int64_t y_bstride, int Cin, int Cout, int W);
int vxm_conv3d_k3_fewout_fwd(const float* x, int Cin, int64_t x_bstride, const float* w, const float* bias, float* y,
                             int64_t y_bstride, int Cout, float act_slope, int B, int D, int H, int W, void* stream);
/* Which kernel a vxm_conv3d_k3_fwd call with these operands dispatches to (for profiling labels):
 * 100 * wide + 10 * CK + NCT, wide = 1: the 8-wave wide-load kernel (k_conv3d_k3_t8<NCT>), 0: k_conv3d_k3<CK,NCT>;
 * 200 + Cin: the few-input-channel kernel (k_conv3d_k3_kpack<Cin>); 300 + NW: the small-volume kernel whose NW waves per block split
 * the input channels (k_conv3d_k3_sm<NW>: grids that would leave more than half of the CUs without a block, i.e. the U-Net levels
 * at 1/8 and 1/16 resolution; VXM_CONV_SMALL_MAX_BLOCKS=0 keeps k_conv3d_k3 there). */
int vxm_conv3d_k3_fwd_variant(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride,
                              const float* wpacked, int Cout, int B, int D, int H, int W);
/* Same for vxm_conv3d_k3_bwd_weight: 10 * kind + NCT, kind = 0: k_conv3d_k3_bwd_weight_dma, 1: ..._vec, 2: collapsed
 * upsampled segment (k_conv3d_k3_bwd_weight_up) + ..._vec on the skip segment, 3: k_fewch_bwd_weight (1-3 channels on one side:
 * first block, flow conv). */
int vxm_conv3d_k3_bwd_weight_variant(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1,
                                     int64_t x1_bstride, const float* dz, int64_t dz_bstride, int Cout, int D, int H, int W);
/* Only the UPSAMPLED segment's share of that gradient, gw[:, 0:C0, :] inside the [Cout][C0+C1][27] array (collapsed low-resolution
 * product), when vxm_conv3d_k3_bwd_weight_variant(...) / 10 == 2; the caller computes the skip segment's share and the bias gradient
 * itself (vxm_conv3d_k3_s3_bwd_weight with ci_off = C0).  x1 is only inspected for alignment. */
int vxm_conv3d_k3_bwd_weight_up_segment(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride,
                                        const float* dz, int64_t dz_bstride, int Cout, float* gw, void* workspace,
                                        size_t workspace_bytes, int B, int D, int H, int W, void* stream);
/* convolution_backward w.r.t. weight and bias: gw [Cout,C0+C1,3,3,3], gb [Cout] (nullable).
 * dz [B,Cout,D,H,W] is the gradient w.r.t. the conv output *before* the activation. */
size_t vxm_conv3d_k3_bwd_weight_workspace_bytes(int Cin, int Cout, int B, int D, int H, int W);
int vxm_conv3d_k3_bwd_weight(const float* x0, int C0, int64_t x0_bstride, int x0_up,
                             const float* x1, int C1, int64_t x1_bstride,
                             const float* dz, int64_t dz_bstride, int Cout, float* gw, float* gb,
                             void* workspace, size_t workspace_bytes, int B, int D, int H, int W, void* stream);

/* dz = g * LeakyReLU'(y)  (leaky_relu_backward; y is the activation OUTPUT, sign(y)=sign(z)). */
int vxm_lrelu_bwd(const float* g, int64_t g_bstride, const float* y, int64_t y_bstride, float* dz,
                  int64_t dz_bstride, float slope, int B, int C, int64_t V, void* stream);

/* ---- general pooling factors: Unet(max_pool=k or a list), networks.py:79-85,130,137-138 (VxmDense itself always builds k = 2, served by
 * the fused entry points above).  Contiguous fp32 tensors, per-axis factors (a 2-D image: D = 1, kd = 1).
 * vxm_maxpool3d_k_fwd: MaxPoolNd(k) -- kernel = stride = k, no padding, floor: x [BC][D][H][W] -> y [BC][D/kd][H/kh][W/kw], ATen's scan
 *   (a later value replaces the maximum when it is greater or NaN).
 * vxm_maxpool3d_k_bwd: gx [BC][D][H][W] = gy routed to the arg-max that scan ends on, zero elsewhere (every element is written).
 * vxm_upsample3d_k_cat: y [B][C0+C1][D][H][W] = cat([Upsample(scale_factor=k, 'nearest')(x0 [B][C0][D/kd][H/kh][W/kw]), x1 [B][C1][D][H][W]]).
 * vxm_upsample3d_k_bwd: gx0 [B][C0][D/kd][H/kh][W/kw] = gy[:, :C0] summed over each voxel's kd x kh x kw children (fixed order). */
int vxm_maxpool3d_k_fwd(const float* x, float* y, int64_t BC, int D, int H, int W, int kd, int kh, int kw, void* stream);
int vxm_maxpool3d_k_bwd(const float* x, const float* gy, float* gx, int64_t BC, int D, int H, int W, int kd, int kh, int kw, void* stream);
int vxm_upsample3d_k_cat(const float* x0, int C0, const float* x1, int C1, float* y, int B, int D, int H, int W, int kd, int kh, int kw,
                         void* stream);
int vxm_upsample3d_k_bwd(const float* gy, int Ctot, int C0, float* gx0, int B, int D, int H, int W, int kd, int kh, int kw, void* stream);

/* ---- MaxPool3d(2), networks.py:83-84,130.  x [B,C,D,H,W] (x_bstride) -> y [B,C,D/2,H/2,W/2]. */
int vxm_maxpool2_fwd(const float* x, int64_t x_bstride, float* y, int B, int C, int D, int H, int W,
                     void* stream);
/* The same, and one 16-bit word per pooled voxel and channel (code [B,C,D/2,H/2,W/2]) with what the backward pass reads of its 2x2x2 block:
 * bit k = 4 dz + 2 dy + dx: x[k] > 0; bits 8..10: the arg-max (first maximum in scan order, a NaN wins).  Even D, H, W. */
int vxm_maxpool2_fwd_code(const float* x, int64_t x_bstride, float* y, uint16_t* code, int B, int C, int D, int H, int W, void* stream);
/* fused backward of {max_pool3d, the skip branch of the concat, leaky_relu}:
 * dz[b,c,p] = (gskip[b,c,p] + (p is the arg-max of its 2x2x2 block ? gpool[b,c,p>>1] : 0)) * LeakyReLU'(x[b,c,p])
 * gskip nullable; slope = 1 gives the plain max-pool backward.  First max in scan order wins ties,
 * like ATen's max_pool3d_with_indices. */
int vxm_maxpool2_bwd(const float* x, int64_t x_bstride, const float* gpool, const float* gskip,
                     int64_t gskip_bstride, float* dz, float slope, int B, int C, int D, int H, int W,
                     void* stream);
/* fused backward of {Upsample(2,'nearest') (networks.py:85,137), leaky_relu}:
 * dz[b,c,q] = (sum over the 2x2x2 children p of q of g[b,c,p]) * LeakyReLU'(y[b,c,q]);  y nullable. */
int vxm_upsample2_bwd(const float* g, int64_t g_bstride, const float* y, float* dz, float slope,
                      int B, int C, int D, int H, int W /* low-res dims */, void* stream);
/* materialise cat([upsample2(x0), x1]) (only needed when a Unet ends on a concat). */
int vxm_upsample2_cat(const float* x0, int C0, const float* x1, int C1, float* out, int B, int D, int H, int W,
                      void* stream);

/* ---- losses (losses.py).  Every *_fwd writes a 0-dim fp32 loss; `acc` is a caller-provided
 * scratch of doubles (zeroed by the call) that the matching *_bwd reads back.  gloss points to
 * the upstream scalar gradient ON DEVICE (no host sync). */
/* NCC.loss, losses.py:15-67 (win^3 zero-padded box sums).  Windows 3..9: one fused kernel marches pixel columns
 * along D (2-D box sums through LDS, a register ring over depth) and keeps the partials (a,b,c) = d cc/d(sum J,
 * sum J^2, sum IJ) for backward in `sums` (3*B*D*H*W floats used, `work` unused).  Larger windows: separable passes;
 * `sums` keeps the five box sums (5*B*D*H*W floats) and `work` is a 5*B*D*H*W scratch. */
/* 1 if (B, win) takes the fused march (sums: 3 planes, work unused), 0 for the separable passes (sums 5, work 5/6 planes):
 * the ONE place that decision is made; callers allocate from it. */
int vxm_ncc_fused(int B, int win);
int vxm_ncc_fwd(const float* I, const float* J, float* loss, float* sums, float* work, double* acc,
                int B, int D, int H, int W, int win, void* stream);
/* gJ = dL/dJ (y_pred) from what vxm_ncc_fwd left in `sums` for the same (I, J, win).  work: unused for windows
 * 3..9, else 6*B*D*H*W floats. */
int vxm_ncc_bwd(const float* I, const float* J, const float* sums, const float* gloss, float* gJ,
                float* work, int B, int D, int H, int W, int win, void* stream);
/* Grad.loss, losses.py:102-135.  mult = loss_mult (1 if None).  acc: 3 * B * VXM_GRAD_SLOTS doubles (per-axis sums, spread over
 * slots so that the blocks' fp64 atomics do not serialise on three addresses). */
#define VXM_GRAD_SLOTS 32
int vxm_gradloss_fwd(const float* y, float* loss, double* acc, int B, int C, int D, int H, int W,
                     int penalty, float mult, void* stream);
int vxm_gradloss_bwd(const float* y, const float* gloss, float* gy, int B, int C, int D, int H, int W,
                     int penalty, float mult, void* stream);
/* MSE.loss, losses.py:75-76.  acc: 1 double. */
int vxm_mse_fwd(const float* a, const float* b, float* loss, double* acc, int64_t n, void* stream);
int vxm_mse_bwd(const float* a, const float* b, const float* gloss, float* ga, float* gb, int64_t n,
                void* stream);
/* Dice.loss, losses.py:84-90.  acc: 2*B*C doubles (top/2, bottom sums). */
int vxm_dice_fwd(const float* yt, const float* yp, float* loss, double* acc, int B, int C, int64_t V,
                 void* stream);
int vxm_dice_bwd(const float* yt, const float* yp, const double* acc, const float* gloss, float* gyt,
                 float* gyp, int B, int C, int64_t V, void* stream);

/* ---- 2-D (planar) variants of the same layers for [B,C,H,W] images with 2-channel flows (channel 0 = row / H
 * displacement, channel 1 = column / W); the reference classes are N-D generic (layers.py:11-97, losses.py:15-135,
 * networks.py:83-85).  The 3x3 convolutions of a 2-D network go through vxm_conv3d_k3_* with D = 1. */
int vxm_warp2d_fwd(const float* src, const float* flow, float* out, int B, int C, int H, int W, int mode, void* stream);
int vxm_warp2d_bwd(const float* src, const float* flow, const float* gout, float* gsrc /* nullable */,
                   float* gflow /* nullable */, int B, int C, int H, int W, int mode, void* stream);
/* steps: [nsteps][B,2,H,W]; work (backward): 2*B*2*H*W floats */
int vxm_vecint2d_fwd(const float* vec, float* steps, int B, int H, int W, int nsteps, void* stream);
int vxm_vecint2d_bwd(const float* vec, const float* steps, const float* gout, float* gvec, float* work,
                     int B, int H, int W, int nsteps, void* stream);
int vxm_resize2d_fwd(const float* x, float* out, int B, int C, int H, int W, int oH, int oW, float factor, void* stream);
int vxm_resize2d_bwd(const float* gout, float* gx, int B, int C, int H, int W, int oH, int oW, float factor, void* stream);
/* MaxPool2d(2) and its gradient (first arg-max of each 2x2 block) */
int vxm_maxpool2d_fwd(const float* x, float* y, int B, int C, int H, int W, void* stream);
int vxm_maxpool2d_bwd(const float* x, const float* gpool, float* gx, int B, int C, int H, int W, void* stream);
/* out = cat([Upsample(2,'nearest')(x0), x1], 1) at H x W; gradient of the upsampled segment from g [B,Ctot,2H,2W] */
int vxm_upsample2d_cat(const float* x0, int C0, const float* x1, int C1, float* out, int B, int H, int W, void* stream);
int vxm_upsample2d_bwd(const float* g, int Ctot, float* gx0, int C0, int B, int H /* low-res */, int W, void* stream);
/* NCC with a win x win window: sums and work are 5 (forward) / 5 and 6 (backward) planes of B*H*W floats */
int vxm_ncc2d_fwd(const float* I, const float* J, float* loss, float* sums, float* work, double* acc,
                  int B, int H, int W, int win, void* stream);
int vxm_ncc2d_bwd(const float* I, const float* J, const float* sums, const float* gloss, float* gJ, float* work,
                  int B, int H, int W, int win, void* stream);
/* NCC on 1-D signals [B,1,L] (voxelmorph/torch/losses.py:15-67 with ndims = 1: conv1d box filter of `win` taps); planes as above */
int vxm_ncc1d_fwd(const float* I, const float* J, float* loss, float* sums, float* work, double* acc, int B, int L, int win, void* stream);
int vxm_ncc1d_bwd(const float* I, const float* J, const float* sums, const float* gloss, float* gJ, float* work, int B, int L, int win,
                  void* stream);
/* NCC.loss with ANY window (voxelmorph/torch/losses.py:26-36,47-67): `w*` taps per axis and `p*` zeros on both sides of it.  The
 * reference pads every axis the tensor has by win[0] // 2 whatever the other window sizes are (:31-36), so non-cubic and even windows
 * change the extent of the box sums: O = S + 2 p - w + 1 per axis; cc (:65) and its mean (:67) are taken on [B, Od, Oh, Ow].  An axis
 * the tensor does not have is passed as extent 1, w 1, p 0.  vxm_ncc_win_elems returns the plane size P the scratch is counted in
 * (0: the shape has no box sums -- the reference's conv raises there too) and *n_out = B*Od*Oh*Ow.
 * sums: 5 * n_out floats (kept for backward); work: 10 * P floats (forward), 6 * P floats (backward); acc: one double. */
int64_t vxm_ncc_win_elems(int B, int D, int H, int W, int wd, int wh, int ww, int pd, int ph, int pw, int64_t* n_out);
int vxm_ncc_win_fwd(const float* I, const float* J, float* loss, float* sums, float* work, double* acc, int B, int D, int H, int W,
                    int wd, int wh, int ww, int pd, int ph, int pw, void* stream);
int vxm_ncc_win_bwd(const float* I, const float* J, const float* sums, const float* gloss, float* gJ, float* work, int B, int D, int H, int W,
                    int wd, int wh, int ww, int pd, int ph, int pw, void* stream);
int vxm_gradloss2d_fwd(const float* y, float* loss, double* acc, int B, int C, int H, int W, int penalty, float mult,
                       void* stream);
int vxm_gradloss2d_bwd(const float* y, const float* gloss, float* gy, int B, int C, int H, int W, int penalty,
                       float mult, void* stream);

/* ---- diagnostics of the fp16-piece engine (csrc/diag.hip; not on the hot path): accumulates into out[0] the sum over the aligned
 * 8-channel x 8 x 8 x 16 voxel tiles of x [B][C][D][H][W] (blocked != 0: channel-blocked [B][C/8][D][H][W][8]) of
 * (non-zero values below 2^-18 of the tile's largest magnitude m) * m^2, into out[1] the sum of x^2, into out[2] the number of such values
 * and into out[3] the number of non-zero values (four doubles, zeroed by the caller): out[2] / out[3] is the share of values that keep an
 * absolute instead of a relative error bound under the per-tile scaling of the two-piece representation (include/vxm_hip.h, pieces = 2). */
int vxm_s3_range_probe(const float* x, int C, int64_t bstride, int blocked, int B, int D, int H, int W, double* out, void* stream);

/* ---- torch.optim.Adam.step (scripts/torch/train.py:161,220) over ONE flat fp32 buffer (which is
 * also the RCCL all-reduce bucket).  g is pre-multiplied by gscale (1/world_size). */
int vxm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float gscale, void* stream);
/* The same update with the step counter in DEVICE memory (torch.optim.Adam(capturable=True)): `state` points at VXM_ADAM_STATE_BYTES
 * zero-initialised, 8-byte aligned bytes {int64 step; float lr / (1 - beta1^step); float sqrt(1 - beta2^step)}; the call advances the
 * counter by one on the stream and applies the update with the new value.  No argument changes from step to step, so the launch pair
 * can be replayed from a hipGraph (voxelmorph_amd/graph.py). */
#define VXM_ADAM_STATE_BYTES 16
int vxm_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                      float beta2, float eps, void* state, float gscale, void* stream);

/* ---- weight / bias gradient of the layers with 1-3 channels on one side (first block 2 -> 16: x0 / x1 the two 1-channel images; flow conv
 * 16 -> 3) on the fp16-piece scheme of the split engine (pieces = 2: two fp16 pieces per fp32 operand, per-tile power-of-two scales, three
 * piece products on v_mfma_f32_16x16x32_f16, fp32 totals; csrc/conv_bwd_weight.hip k_fewch_bwd_weight_h) -- the convolution_backward
 * (weight, bias) of networks.py:299,211 for those two layers.  Operands, results, determinism and workspace
 * (vxm_conv3d_k3_bwd_weight_workspace_bytes) as vxm_conv3d_k3_bwd_weight, which computes the same product on the fp32 matrix pipe. */
int vxm_conv3d_k3_fewch_bwd_weight_ok(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* dz,
                                      int64_t dz_bstride, int Cout, int pieces, int W);
int vxm_conv3d_k3_fewch_bwd_weight(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* dz,
                                   int64_t dz_bstride, int Cout, float* gw, float* gb, void* workspace, size_t workspace_bytes, int B, int D, int H,
                                   int W, int pieces, void* stream);
/* The first ConvBlock's weight / bias gradient straight from the operands of vxm_maxpool2_bwd (networks.py:130,299-305 autograd twins:
 * max_pool3d backward + the skip branch of the concat + leaky_relu_backward + convolution_backward(weight, bias) in one launch): the gradient at the
 * block's pre-activation, dz = (gskip + routed gpool) * LeakyReLU'(y), is formed while its tiles are staged -- from gskip [B,16,D,H,W]
 * (gskip_bstride), gpool [B,16,D/2,H/2,W/2] and the code words of vxm_maxpool2_fwd_code -- instead of being written (0.44 GB at 160x192x224) and
 * read back.  Same arithmetic in the same order as vxm_maxpool2_bwd followed by vxm_conv3d_k3_fewch_bwd_weight: bit-identical gw / gb.
 * Cout = 16, C0 + C1 <= 3, even D, H, W, W % 4 == 0, pieces = 2; workspace as vxm_conv3d_k3_bwd_weight_workspace_bytes(C0 + C1, 16, ...). */
int vxm_conv3d_k3_fewch_bwd_weight_pool(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* gskip,
                                        int64_t gskip_bstride, const float* gpool, const uint16_t* code, float slope, float* gw, float* gb, void* workspace,
                                        size_t workspace_bytes, int B, int D, int H, int W, int pieces, void* stream);

/* ---- the weighted sum of the loss terms, scripts/torch/train.py:205-212 (`loss += loss_function(y_true[n], y_pred[n]) * weights[n]`):
 * total[0] = sum_n terms[n][0] * weights[n], products and running sum in that order in fp32, ONE launch instead of a mul + an add per term.
 * terms: HOST array of n device pointers to the 0-dim loss tensors; weights: HOST array.  running (nullable, device, n + 1 floats):
 * running[n'] += term n' * weight, running[n] += total -- the per-epoch log of train.py:215 without a read-back per step.
 * _bwd: gterms[n'] (device, n floats) = gtotal[0] * weights[n'], the upstream gradient of each term's own backward kernel. */
#define VXM_LOSS_TERMS_MAX 8
int vxm_loss_combine_fwd(const float* const* terms, const float* weights, int n, float* total, float* running, void* stream);
int vxm_loss_combine_bwd(const float* gtotal, const float* weights, int n, float* gterms, void* stream);
/* out[i] = a[i] + b[i]: the gradient of a tensor that two ops consume (preint_flow: Grad and VecInt, networks.py:262-268), so that the
 * captured step carries no ATen accumulation kernel */
int vxm_add2(const float* a, const float* b, float* out, int64_t n, void* stream);
/* zero `bytes` bytes on the stream (a memset node when captured): FlatAdam.zero_grad() without an ATen fill kernel */
int vxm_fill_zero(void* p, size_t bytes, void* stream);

/* ---- bf16 activations / fp32 accumulate path of the U-Net (BASELINE.json configs[1]); csrc/conv_bf16.hip.
 * What torch.autocast(bfloat16) over ConvBlock / flow conv / MaxPool3d / Upsample + cat (networks.py:83-85,130,137-138,211,257,
 * 290-305) would dispatch to MIOpen.  Activations and their gradients are CHANNEL-BLOCKED bf16 tensors
 * [B][C/8][D][H][W][8] (contiguous, C a multiple of 16, 16-byte aligned; passed as void*); weights, biases and parameter
 * gradients stay fp32 in the reference layout [Cout][Cin][3][3][3]. */
/* planar fp32 [B][C0 (+C1)][V] -> blocked bf16 with Cblk >= C0 + C1 channels (the rest zero); and back (first C channels). */
int vxm_bf16_to_blocked(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, void* out,
                        int Cblk, int B, int64_t V, void* stream);
int vxm_bf16_from_blocked(const void* x, int Cblk, float* out, int C, int B, int64_t V, void* stream);
/* weights of an operator with InC inputs / OutC outputs in MFMA fragment order, rounded to bf16.  Forward: inputs = the
 * input-channel range [ci_lo, ci_lo + ci_n) of w, outputs = Cw_out.  transpose_flip: the adjoint onto that range
 * (backward-data): inputs = Cw_out, outputs = ci_n. */
size_t vxm_bf16_conv_packed_bytes(int InC, int OutC);
int vxm_bf16_conv_pack_weights(const float* w, int Cw_in, int Cw_out, int ci_lo, int ci_n, int transpose_flip, void* wpacked,
                               void* stream);
/* the same for n_jobs operators in one launch (the job table is host memory, read before the call returns): what a training step
 * uses after every optimiser step, when all the packed copies of a network are stale at once */
typedef struct VxmBf16PackJob {
    const float* w;
    void* wpacked;
    int Cw_in, Cw_out, ci_lo, ci_n, transpose_flip;
} VxmBf16PackJob;
int vxm_bf16_conv_pack_weights_batch(const VxmBf16PackJob* jobs, int n_jobs, void* stream);
/* conv3d(k3,p1) over the virtual concat [x0 (optionally nearest-x2 upsampled) | x1] + bias + LeakyReLU(leaky_slope).
 * out_planar_f32 = 0: y blocked bf16 with Cout (multiple of 16) channels, optionally multiplied by LeakyReLU'(mask)
 * (mask: blocked, Cout channels; the fused leaky_relu_backward of backward-data).  out_planar_f32 = 1: y fp32
 * [B][Cout <= 4][D][H][W] (the flow head).  Backward-data = this entry point with transpose_flip-packed weights.
 * out_planar_f32 | 2: scheduling hint, walk the output tiles from the end of the tensor (as VXM_S3_REVERSE_TILES; same results). */
int vxm_bf16_conv_fwd(const void* x0, int C0, int x0_up, const void* x1, int C1, const void* wpacked, const float* bias, void* y,
                      int Cout, int out_planar_f32, float leaky_slope, const void* mask, float mask_slope,
                      int B, int D, int H, int W, void* stream);
/* backward-data onto a nearest-x2 UPSAMPLED segment in one kernel (replaces vxm_bf16_conv_fwd at full resolution followed by
 * vxm_bf16_upsample2_bwd; same rounding points): dz blocked [B][Cdz/8][D][H][W][8], wpacked = the transpose_flip pack of the segment's
 * channel range, dx_low blocked [B][Cout/8][D/2][H/2][W/2][8] = (sum over the 2x2x2 children of the adjoint conv) * LeakyReLU'(mask_low)
 * (mask_low: blocked, low resolution, or NULL).  D, H, W even. */
int vxm_bf16_conv_bwd_data_up(const void* dz, int Cdz, const void* wpacked, void* dx_low, int Cout, const void* mask_low, float mask_slope,
                              int B, int D, int H, int W, void* stream);
/* gw[Cout_w][Cin_w][27], gb[Cout_w] (nullable) fp32 from the blocked input (virtual concat, C0 + C1 >= Cin_w channels) and the
 * blocked gradient dz (Cdz = 16 or 32 >= Cout_w channels).  Deterministic (fixed-order partial sums in `work`). */
size_t vxm_bf16_conv_bwd_weight_workspace_bytes(int Cin, int Cout, int B, int D, int H, int W);
int vxm_bf16_conv_bwd_weight(const void* x0, int C0, int x0_up, const void* x1, int C1, const void* dz, int Cdz, float* gw,
                             int Cin_w, int Cout_w, float* gb, void* work, size_t work_bytes, int B, int D, int H, int W,
                             void* stream);
/* MaxPool3d(2) forward; its backward fused with the skip-branch gradient (nullable) and LeakyReLU'(x); Upsample(2) backward
 * fused with LeakyReLU'(y) (y nullable); plain leaky_relu_backward.  All on blocked tensors; D, H, W = input (full) extents,
 * Dl.. = low-resolution extents. */
int vxm_bf16_maxpool2_fwd(const void* x, void* y, int B, int C, int D, int H, int W, void* stream);
int vxm_bf16_maxpool2_bwd(const void* x, const void* gpool, const void* gskip, void* dz, float slope, int B, int C,
                          int D, int H, int W, void* stream);
int vxm_bf16_upsample2_bwd(const void* g, const void* y, void* dz, float slope, int B, int C, int Dl, int Hl, int Wl,
                           void* stream);
int vxm_bf16_lrelu_bwd(const void* g, const void* y, void* dz, float slope, int64_t n_elems, void* stream);

/* ---- fp32 convolutions on the bf16 matrix pipe ("bf16x3" split, csrc/conv_s3.hip).
 * Same operator, operands and layouts as vxm_conv3d_k3_fwd (ConvBlock networks.py:299-305 forward, and convolution_backward
 * w.r.t. the input with transpose_flip-packed weights): planar fp32 NCDHW in and out, virtual concat [x0 (optionally nearest-x2
 * upsampled) | x1], bias + LeakyReLU(act_slope) (+ LeakyReLU'(mask_src) = fused leaky_relu_backward).  Every fp32 operand is
 * split into three bf16 pieces (exactly: x = h + m + l up to 2^-24 |x|) while it is staged, and a product is accumulated in fp32
 * from its six leading piece products on v_mfma_f32_16x16x32_bf16: fp32-level accuracy (same rel-L2 <= 1e-5 gate against an fp64
 * evaluation as the fp32-MFMA kernels), 2.67x the matrix-pipe rate.  C0, C1 multiples of 8.
 * `pieces` selects the split: 3 = the three bf16 pieces above; 2 = TWO fp16 pieces (x s = h + l up to 2^-22 |x s|, three products
 * hh + hl + lh on v_mfma_f32_16x16x32_f16: half the matrix instructions).  fp16 has a 5-bit exponent, so every staged tile is scaled by
 * the power of two s that brings its largest magnitude into [2^14, 2^15) and an MFMA chain lives for one staged chunk before the vector
 * ALU folds it, unscaled, into the fp32 totals; weights carry one scale per packed operator.  Same accuracy gates as pieces = 3; the
 * difference is range: inside ONE staged tile, values below 2^-18 of the tile's largest magnitude lose relative precision (absolute
 * error 2^-40 of that magnitude), which bf16's 8-bit exponent does not.  The packed weights of the two schemes differ (pass the same
 * `pieces` to packed_bytes / the pack job / the launch).
 * Non-finite values (round 6): a staged unit's scale is taken over its FINITE values, so an Inf / NaN among the activations or gradients stays
 * an Inf / NaN piece and poisons exactly the outputs whose 3 x 3 x 3 window contains it (for a weight gradient: its channel's slice), as ATen's
 * convolution does; the other outputs of the unit keep their fp32-level values.  (Which outputs a non-finite value next to the volume border
 * reaches through the zero padding -- 0 x Inf = NaN -- depends on which operand a kernel pads; ATen's vol2col has the same freedom.)
 * vxm_conv3d_k3_s3_ok: 1 when the split kernel takes a launch of this shape (otherwise use vxm_conv3d_k3_fwd). */
int vxm_conv3d_k3_s3_ok(int C0, int C1, int Cout, int B, int D, int H, int W);
/* Layout flags, OR-ed into the `pieces` argument of the launch entry points of the split engine (fp16 scheme only).  A flagged tensor is
 * CHANNEL-BLOCKED: [B][C/8][D][H][W][8] fp32 (element (c, v) of a sample at ((c / 8) V + v) 8 + c % 8; C % 8 == 0; same batch strides as
 * NCDHW).  Used BETWEEN the kernels of the fused U-Net for tensors only split kernels touch: a haloed row of a staged tile is then one
 * contiguous run instead of eight 72-byte pieces.  Values and results are those of the planar launch, bit for bit.
 * Which operand a flag names is stated at each entry point; vxm_conv3d_k3_s3_layout_ok tells whether vxm_conv3d_k3_s3_fwd takes them. */
#define VXM_S3_IN0_BLOCKED 0x100   /* first tensor operand  */
#define VXM_S3_IN1_BLOCKED 0x200   /* second tensor operand */
#define VXM_S3_OUT_BLOCKED 0x400   /* output tensor and, where there is one, the mask tensor of the fused leaky_relu_backward */
/* Phase flags of the two backward-weight entry points of the split engine (vxm_conv3d_k3_s3_bwd_weight, vxm_conv3d_k3_s3u_bwd_weight), OR-ed
 * into `pieces` like the layout flags.  A weight gradient is a contraction kernel (one partial sum per block into `work`) followed by a small
 * reduction kernel (work -> gw, gb).  Run back to back on one stream, the few blocks of the reduction wait behind the persistent blocks of
 * whatever shares the chip, and the NEXT contraction of that stream waits with them (measured: 450 - 585 us stalls, three per step).  With
 * CONTRACT_ONLY the call stops after the contraction; a second call with REDUCE_ONLY (same arguments, same `work`, any stream that waits
 * for the first) finishes it -- the fused U-Net backward sends the reductions to a third stream.  Neither flag: both, as before. */
/* Scheduling hint of vxm_conv3d_k3_s3_fwd, OR-ed into `pieces` like the flags above: walk the output tiles from the END of the tensor.  Same
 * results bit for bit; a launch that reads what the previous launch has just written starts where its producer stopped and finds that
 * part of the tensor in the memory-side cache (256 MB against 440 - 880 MB tensors). */
#define VXM_S3_REVERSE_TILES 0x4000
/* SIGN tensors (round 6), OR-ed into `pieces` of vxm_conv3d_k3_s3_fwd (and written by vxm_conv3d_k3_s3u_fwd_signs): leaky_relu_backward needs
 * one bit of the activation it is taken at.  A sign tensor is [B][C/4][D][H][W] BYTES, bit j of byte (q, v) = (y[4 q + j][v] > 0); batch stride in bytes.
 * VXM_S3_OUT_SIGNS: a forward launch with a channel-blocked output also writes the sign tensor of that output, passed in the `mask_src` /
 * `mask_bstride` arguments (which a forward launch does not otherwise use).  VXM_S3_MASK_SIGNS: `mask_src` of a backward-data launch with a
 * channel-blocked output IS such a sign tensor (C = Cout of the launch) instead of the fp32 activation: same products, 1/32 of the mask bytes. */
#define VXM_S3_MASK_SIGNS 0x8000
#define VXM_S3_OUT_SIGNS 0x10000
#define VXM_S3_BW_CONTRACT_ONLY 0x1000
#define VXM_S3_BW_REDUCE_ONLY 0x2000
/* Round 6, late: the layout flags at the few-channel kernels either side of the LAST ConvBlock of the fused U-Net (networks.py:211,257: the
 * 16 -> 3 flow conv reads that block's output; its backward-data writes the block's gradient; its weight gradient reads the output again), so
 * that this activation, its gradient and its LeakyReLU mask can take the channel-blocked / sign-tensor form too.  Same values, same arithmetic
 * in the same order as the planar launches: bit-identical results.
 *   vxm_conv3d_k3_fewout_fwd_layout: vxm_conv3d_k3_fewout_fwd with `layout` = VXM_S3_IN0_BLOCKED (x channel-blocked; Cin % 8 == 0) or 0.
 *   vxm_conv3d_k3_fwd_layout: vxm_conv3d_k3_fwd with `layout` = VXM_S3_OUT_BLOCKED (y channel-blocked, and mask_src in the same layout) optionally
 *     | VXM_S3_MASK_SIGNS (mask_src is the sign tensor of the activation, mask_bstride in bytes), or 0.  Flags are accepted where
 *     vxm_conv3d_k3_fwd_layout_ok returns 1: launches the few-input-channel kernel takes (C0 + C1 <= 4, W % 4 == 0, aligned) with Cout % 8 == 0.
 *   vxm_conv3d_k3_fewch_bwd_weight: `pieces` may carry VXM_S3_IN0_BLOCKED when x0 is the 16-channel operand (flow conv: C0 = 16, C1 = 0, Cout <= 3). */
int vxm_conv3d_k3_fewout_fwd_layout(const float* x, int Cin, int64_t x_bstride, const float* w, const float* bias, float* y,
                                    int64_t y_bstride, int Cout, float act_slope, int B, int D, int H, int W, int layout, void* stream);
int vxm_conv3d_k3_fwd_layout_ok(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const float* wpacked,
                                int Cout, int B, int D, int H, int W);
int vxm_conv3d_k3_fwd_layout(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                             const float* wpacked, const float* bias, float* y, int64_t y_bstride, int Cout, float act_slope,
                             const float* mask_src, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W, int layout, void* stream);
int vxm_conv3d_k3_s3_layout_ok(int C0, int C1, int x0_up, int Cout, int H, int pieces);
int vxm_conv3d_k3_s3_variant(int Cout);                     /* 10 * NCT + CB of the kernel instance (profiling labels) */
int vxm_conv3d_k3_s3_tile_rows(int Cout, int pieces, int H); /* rows of its output tile: 8 x 8 x 16 on the fp16 scheme, else 8 x 4 x 16 */
int vxm_conv3d_k3_s3_tile_rows_at(int Cout, int pieces, int B, int D, int H, int W); /* ... of a launch of this shape (a 32-channel operator takes
                                                                                       * 4 rows where 8 would leave CUs without a block) */
/* 1 when the launch runs k_s3p_conv (the same tile, operator pack and results with producer and consumer waves: 16-output-channel
 * forward launches of the fp16 scheme on large volumes; profiling labels) */
int vxm_conv3d_k3_s3_producer_consumer(int Cout, int pieces, int has_mask, int B, int D, int H, int W);
/* packed, pre-split weights of one operator: seg0 / seg1 = input channels of the two segments of the virtual concat it reads */
size_t vxm_conv3d_k3_s3_packed_bytes(int seg0, int seg1, int OutC, int pieces);
typedef struct VxmS3PackJob {
    const float* w;                                         /* [Cw_out][Cw_in][3][3][3], reference layout */
    void* wpacked;                                          /* vxm_conv3d_k3_s3_packed_bytes(seg0, InC - seg0, OutC) bytes, 16-byte aligned */
    int Cw_in, Cw_out, ci_lo, ci_n, transpose_flip;         /* as vxm_bf16_conv_pack_weights: forward operator on input channels
                                                             * [ci_lo, ci_lo + ci_n) (InC = ci_n, OutC = Cw_out), or its adjoint onto them
                                                             * (transpose_flip: InC = Cw_out, OutC = ci_n) */
    int seg0;                                               /* operator input channels that belong to segment 0 (InC for one tensor) */
    int pieces;                                             /* 3: bf16 (h, m, l); 2: fp16 (h, l) with the operator's power-of-two scale */
} VxmS3PackJob;
int vxm_conv3d_k3_s3_pack_weights_batch(const VxmS3PackJob* jobs, int n_jobs, void* stream);
/* `pieces` may carry VXM_S3_IN0_BLOCKED (x0) and / or VXM_S3_OUT_BLOCKED (y and mask_src) where vxm_conv3d_k3_s3_layout_ok says so. */
int vxm_conv3d_k3_s3_fwd(const float* x0, int C0, int64_t x0_bstride, int x0_up, const float* x1, int C1, int64_t x1_bstride,
                         const void* wpacked, const float* bias, float* y, int64_t y_bstride, int Cout, float act_slope,
                         const float* mask_src, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W, int pieces,
                         void* stream);
/* convolution_backward w.r.t. weight and bias (autograd twin of networks.py:299) of ONE full-resolution tensor x [B,C,D,H,W] on the
 * same split arithmetic (contraction over voxels, K = 32 voxels of a W row per MFMA): gw[co][ci_off + ci][tap] for ci < C inside a
 * [Cout][gw_cin][3][3][3] array (the channel sub-range a segment of a virtual concat owns), gb[Cout] (nullable).  C, Cout multiples
 * of 16, W even (the staging loads W-neighbouring voxel pairs; an odd W is VXM_ERR_BAD_SHAPE and _ok returns 0).  Deterministic
 * (fixed-order partial sums in `work`).  _ok: 1 when the split kernel takes a launch of this shape.  On the fp16 scheme `pieces` may carry
 * VXM_S3_IN0_BLOCKED (x) and / or VXM_S3_IN1_BLOCKED (dz). */
int vxm_conv3d_k3_s3_bwd_weight_ok(int C, int Cout, int B, int D, int H, int W);
size_t vxm_conv3d_k3_s3_bwd_weight_workspace_bytes(int C, int Cout, int B, int D, int H, int W);
/* which kernel a launch of this shape runs (profiles, bench regions): 0 = k_s3_bwd_weight<pieces>; round 6, fp16 pieces: 1 = k_s3_bww_pc<false, 2>
 * (haloed x, two dz tiles per block), 2 = k_s3_bww_pc<true, 2> (haloed dz, two x chunks), 3 = k_s3_bww_pc<false, 1> */
int vxm_conv3d_k3_s3_bwd_weight_kernel(int C, int Cout, int pieces);
int vxm_conv3d_k3_s3_bwd_weight(const float* x, int C, int64_t x_bstride, const float* dz, int64_t dz_bstride, int Cout, float* gw,
                                int gw_cin, int ci_off, float* gb, void* work, size_t work_bytes, int B, int D, int H, int W,
                                int pieces, void* stream);

/* ---- cat([Upsample(2,'nearest')(x0), x1]) -> ConvBlock on the split engine, COLLAPSED (csrc/conv_s3u.hip): upsample_nearest3d + cat +
 * convolution + leaky_relu of voxelmorph/torch/networks.py:133-138,299-305 in one launch.  The upsampled segment is evaluated at
 * low-resolution cost -- per output parity class a 2x2x2 kernel of pre-summed taps on the low-resolution grid (8 instead of 27
 * multiply-adds per input channel) --, the skip segment x1 as the plain 27-tap convolution, both with the split arithmetic selected by
 * `pieces` (see vxm_conv3d_k3_s3_fwd).  x0: [B,C0,D/2,H/2,W/2], x1: [B,C1,D,H,W] (C1 may be 0), y: [B,Cout,D,H,W]; C0, C1 multiples of
 * 8, D, H, W even.  The operator is packed from the reference-layout weights [Cout][C0+C1][3][3][3] by _pack_weights (per `pieces`).
 * _ok: 1 when this kernel takes a launch of this shape.  `pieces` of _fwd may carry VXM_S3_OUT_BLOCKED (y; Cout a multiple of 8). */
int vxm_conv3d_k3_s3u_ok(int C0, int C1, int Cout, int B, int D, int H, int W, int pieces);
size_t vxm_conv3d_k3_s3u_packed_bytes(int C0, int C1, int Cout, int pieces);
int vxm_conv3d_k3_s3u_pack_weights(const float* w, void* wpacked, int C0, int C1, int Cout, int pieces, void* stream);
int vxm_conv3d_k3_s3u_fwd(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const void* wpacked,
                          const float* bias, float* y, int64_t y_bstride, int Cout, float act_slope, int B, int D, int H, int W, int pieces,
                          void* stream);
/* _fwd, and with signs != NULL (channel-blocked y only) the sign tensor of y beside it: [B][Cout/4][D][H][W] bytes, batch stride signs_bstride bytes */
int vxm_conv3d_k3_s3u_fwd_signs(const float* x0, int C0, int64_t x0_bstride, const float* x1, int C1, int64_t x1_bstride, const void* wpacked,
                                const float* bias, float* y, int64_t y_bstride, int Cout, float act_slope, int B, int D, int H, int W, int pieces,
                                unsigned char* signs, int64_t signs_bstride, void* stream);
/* 1 when _fwd runs this call on k_s3u_conv_pc (round 6: producer / consumer waves, double-buffered staging tile; fp16 pieces, W % 4 == 0,
 * even batch strides), 0: k_s3u_conv -- for profiles and bench regions */
int vxm_conv3d_k3_s3u_fwd_kernel(int64_t x0_bstride, int64_t x1_bstride, int D, int H, int W, int pieces);

/* BOTH backward-data products of such a layer from one staging of dz (round 6, csrc/conv_s3u.hip k_s3u_bwd_pc; fp16 pieces): gxl as
 * vxm_conv3d_k3_s3u_bwd_low below (operator `wlow` from vxm_conv3d_k3_s3u_bwd_low_pack_weights), and gx1 [B,C1,D,H,W] = convolution_backward
 * (input) restricted to the skip channels [C0, C0 + C1) of the weights (operator `wskip` from _bwd_skip_pack_weights; no mask: the skip tensor's
 * LeakyReLU' is applied where its gradients are summed).  C0, C1 <= 32, multiples of 8; W a multiple of 4; `pieces` may carry VXM_S3_IN0_BLOCKED (dz).
 * _ok returns 0 unless VXM_S3U_BWD_PC=1: measured, the one launch is not faster than the two it replaces (callers keep those; csrc/conv_s3u.hip says why). */
int vxm_conv3d_k3_s3u_bwd_data_ok(int C0, int C1, int Cout, int B, int D, int H, int W, int pieces);
size_t vxm_conv3d_k3_s3u_bwd_skip_packed_bytes(int C1, int Cout, int pieces);
int vxm_conv3d_k3_s3u_bwd_skip_pack_weights(const float* w, void* wpacked, int C0, int C1, int Cout, int pieces, void* stream);
int vxm_conv3d_k3_s3u_bwd_data(const float* dz, int64_t dz_bstride, int Cout, const void* wlow, float* gxl, int64_t gxl_bstride, int C0, const float* mask,
                               int64_t mask_bstride, float mask_slope, const void* wskip, float* gx1, int64_t gx1_bstride, int C1, int B, int D, int H, int W,
                               int pieces, void* stream);

/* convolution_backward (input) of the UPSAMPLED segment of such a layer, straight onto the low-resolution tensor it was upsampled from:
 * conv backward + upsample_nearest3d_backward + leaky_relu_backward(mask_src) in one launch on the split arithmetic (csrc/conv_s3u.hip:
 * a stride-2, 4x4x4-tap convolution of the full-resolution dz [B,Cout,D,H,W] with the transposed collapsed weights; the full-resolution
 * gradient of those channels is never written).  gxl: [B,C0,D/2,H/2,W/2]; w: [Cout][Cin][27] whose first C0 input channels are the
 * upsampled segment.  pieces = 2 only (_ok returns 0 otherwise: callers keep vxm_conv3d_k3_up_bwd_low); it may carry VXM_S3_IN0_BLOCKED (dz;
 * Cout a multiple of 8). */
int vxm_conv3d_k3_s3u_bwd_low_ok(int C0, int Cout, int B, int D, int H, int W, int pieces);
size_t vxm_conv3d_k3_s3u_bwd_low_packed_bytes(int C0, int Cout, int pieces);
int vxm_conv3d_k3_s3u_bwd_low_pack_weights(const float* w, void* wpacked, int C0, int Cin, int Cout, int pieces, void* stream);
int vxm_conv3d_k3_s3u_bwd_low(const float* dz, int64_t dz_bstride, int Cout, const void* wpacked, float* gxl, int64_t gxl_bstride, int C0,
                              const float* mask_src, int64_t mask_bstride, float mask_slope, int B, int D, int H, int W, int pieces,
                              void* stream);

/* convolution_backward (weight) of the UPSAMPLED segment of such a layer, collapsed and on the split arithmetic (csrc/conv_s3u.hip:
 * k_s3u_bww): gw[co][0:C0][tap] inside a [Cout][gw_cin][3][3][3] array from x0 [B,C0,D/2,H/2,W/2] and dz [B,Cout,D,H,W] -- per offset
 * a in {-1,0,1,2}^3 the contraction sum_m dz[2 m + a] x0[m] over the low-resolution voxels, then the taps as sums of offsets.  The
 * caller computes the skip segment's share and the bias gradient (vxm_conv3d_k3_s3_bwd_weight with ci_off = C0).  C0 = 16 or 32, Cout a
 * multiple of 16, D, H even, W a multiple of 4, pieces = 2 (it may carry VXM_S3_IN1_BLOCKED: dz); deterministic (fixed-order partial sums
 * in `work`). */
int vxm_conv3d_k3_s3u_bwd_weight_ok(int C0, int Cout, int B, int D, int H, int W, int pieces);
size_t vxm_conv3d_k3_s3u_bwd_weight_workspace_bytes(int C0, int Cout, int B, int D, int H, int W);
int vxm_conv3d_k3_s3u_bwd_weight(const float* x0, int C0, int64_t x0_bstride, const float* dz, int64_t dz_bstride, int Cout, float* gw,
                                 int gw_cin, void* work, size_t work_bytes, int B, int D, int H, int W, int pieces, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VXM_HIP_H */
