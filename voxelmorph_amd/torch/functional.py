"""torch.autograd.Function wrappers over the C ABI (include/vxm_hip.h).

Each Function replaces one ATen op chain of the reference (file:line in the docstrings) with
calls into libvxm_hip.so on the current HIP stream.  PyTorch is used for device memory
(torch.empty through the caching allocator), streams and autograd bookkeeping only.
"""
import ctypes
import math
import os

import torch

from .. import _lib
from .. import profiler as _prof
from .._lib import call, ptr, require_device, stream

SPLIT_48 = True     # conv_bwd_data: produce 48-channel results as 32 + 16 (module-level switch for A/B timing)
OVERLAP_SMALL_LEVELS = os.environ.get("VXM_NO_OVERLAP", "") != "1"     # UnetFn.backward: weight gradients of the coarse levels on a second HIP stream
# first U-Net level whose weight gradients go to the second stream.  0 (default since round 3): all of them -- the tails of the big
# full-resolution launches fill with each other's blocks (-1.6 % per step, measured); 1 keeps the full-resolution launches serialised, which
# is what bench.py's per-kernel pass sets so that a launch's duration is its own.
OVERLAP_MIN_LEVEL = int(os.environ.get("VXM_OVERLAP_MIN_LEVEL", "0"))
# fp32 engine of the 3-D convolutions.  The split engines run the plain full-resolution layers on the 16-bit matrix pipe with every fp32
# operand split into pieces while it is staged (csrc/conv_s3.hip; fp32-level accuracy, gated against fp64 in tests/test_gpu_s3.py):
#   "f16x2"  two fp16 pieces (11 + 11 significand bits), three piece products, per-tile power-of-two scaling (round 4, default)
#   "split"  three bf16 pieces (8 + 8 + 8 bits), six piece products (round 3; alias "bf16x3")
#   "native" v_mfma_f32_16x16x4_f32 everywhere (csrc/conv_fwd.hip, rounds 1-2)
# VXM_S3_UP=1 also sends cat([upsample(x0), x1]) layers through the split kernel (nominal FLOPs, gather through the upsampling)
# instead of the collapsed-weight fp32 kernel.
FP32_ENGINE = os.environ.get("VXM_FP32_ENGINE", "f16x2")
if FP32_ENGINE == "bf16x3":
    FP32_ENGINE = "split"
if FP32_ENGINE not in ("f16x2", "split", "native"):
    raise ValueError("VXM_FP32_ENGINE must be 'f16x2', 'split' (= 'bf16x3') or 'native', got %r" % FP32_ENGINE)


def s3_pieces():
    """piece scheme of the split kernels for the engine selected NOW (bench.py and the tests switch FP32_ENGINE inside a process)"""
    return 2 if FP32_ENGINE == "f16x2" else 3

S3_UP = os.environ.get("VXM_S3_UP", "") == "1"
# cat([upsample(x0), x1]) forwards on the split + collapsed kernel (csrc/conv_s3u.hip: the upsampled segment at low-resolution cost AND on
# the 16-bit matrix pipe); VXM_S3U=0 keeps them on the round-3 kernels (collapsed fp32-MFMA kernel / split kernel through the gather)
S3U = os.environ.get("VXM_S3U", "1") != "0"
# channel-blocked interior tensors of the fused U-Net (tensors that only split kernels write and read: _blocked_tensors); VXM_BLOCKED=0: all planar
BLOCKED = os.environ.get("VXM_BLOCKED", "1") != "0"
# ... and the LAST activation of the fused U-Net (read by the 16 -> 3 flow conv), its gradient and its sign tensor, through the layout flags of the
# few-channel kernels (round 6, late); VXM_BLOCKED_LAST=0: that tensor stays planar
BLOCKED_LAST = os.environ.get("VXM_BLOCKED_LAST", "1") != "0"
# UnetFn.backward: enqueue a layer's weight gradient (second stream) AFTER its backward-data launches (main stream) instead of before them.
# Same dependencies; in a captured graph the order of node creation decides which child of the dz producer stays on its queue.
# Default: "after" while a hipGraph is being captured (same-box A/B of the replayed step: 80.4 / 81.3 -> 83.4 / 82.4 pairs/s; the replay then keeps
# the main chain on one queue and every weight gradient on a second one, as launch-by-launch submission does: 7.0 ms of overlap and 0.05 ms
# idle per step instead of 3.7 and 0.14), "before" otherwise (launch by launch: 83.4 / 81.6 against 82.0 / 81.2).  VXM_DW_ORDER=before|after forces one.
DW_ORDER = os.environ.get("VXM_DW_ORDER", "")
BW_REDUCE_STREAM = os.environ.get("VXM_BW_REDUCE_STREAM", "") == "1"     # UnetFn.backward: weight-gradient reductions on a third stream (A/B: slower)
# few-channel weight gradients (first block, flow conv) on the fp16 pieces (csrc/conv_bwd_weight.hip k_fewch_bwd_weight_h); VXM_FEWCH_H=0: fp32 MFMA
FEWCH_H = os.environ.get("VXM_FEWCH_H", "1") != "0"
# the first block's weight gradient forms the gradient at its pre-activation on the fly from the operands of the pooling backward (16-bit codes from
# the forward pooling, vxm_conv3d_k3_fewch_bwd_weight_pool): no vxm_maxpool2_bwd launch, no 0.44 GB tensor written and read back; VXM_POOL_FUSE=0: off
POOL_FUSE = os.environ.get("VXM_POOL_FUSE", "1") != "0"
_SIDE_STREAMS = {}


def split_engine():
    return FP32_ENGINE != "native"


def fp32_engine_note():
    if FP32_ENGINE == "native":
        return "native: v_mfma_f32_16x16x4_f32 for every conv product"
    if FP32_ENGINE == "f16x2":
        return ("f16x2: the three conv products of the plain full-resolution layers as 2 x fp16 pieces per fp32 operand (per-tile power-of-two "
                "scale), 3 piece products on v_mfma_f32_16x16x32_f16, chains folded into fp32 totals per staged chunk (fp32-level accuracy, "
                "parity-gated against fp64); every other conv product on v_mfma_f32_16x16x4_f32")
    return ("split: the three conv products of the plain full-resolution layers as 3 x bf16 pieces per fp32 operand, 6 piece products on "
            "v_mfma_f32_16x16x32_bf16 with fp32 accumulation (fp32-level accuracy, parity-gated); every other conv product on "
            "v_mfma_f32_16x16x4_f32")


def _side_stream(dev, which=0):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), which)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[key]
INTERP = {"bilinear": 0, "nearest": 1}
PENALTY = {"l1": 0, "l2": 1}


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _vol3(t, what):
    if t.dim() != 5:
        raise NotImplementedError("%s: the MI355X path implements 3-D volumes [B,C,D,H,W]; got %d-D "
                                  "(2-D support is listed as 'next' in DESIGN.md)" % (what, t.dim() - 2))


def vecint_work_elems(shape):
    """floats of scratch vxm_vecint_bwd_ws wants for a [B,3,D,H,W] field: two gradient buffers, the per-step statistics and the per-tile
    displacement records (vxm_workspace_bytes(VXM_WS_VECINT_BWD, ...))"""
    B, _, D, H, W = (int(v) for v in shape)
    return int(_lib.lib().vxm_workspace_bytes(6, 0, 0, B, D, H, W)) // 4


# --------------------------------------------------------------------------- layers
class WarpFn(torch.autograd.Function):
    """SpatialTransformer.forward (voxelmorph/torch/layers.py:30-48)."""

    @staticmethod
    def forward(ctx, src, flow, mode):
        _vol3(src, "SpatialTransformer")
        require_device(src, flow)
        src, flow = _c(src), _c(flow)
        B, C, D, H, W = src.shape
        if tuple(flow.shape) != (B, 3, D, H, W):
            raise ValueError("flow shape %s does not match src %s" % (tuple(flow.shape), tuple(src.shape)))
        out = torch.empty_like(src)
        with _prof.region("warp3d_fwd", nbytes=4.0 * B * D * H * W * (2 * C + 3)):
            call("vxm_warp3d_fwd", ptr(src), ptr(flow), ptr(out), B, C, D, H, W, INTERP[mode], stream())
        ctx.save_for_backward(src, flow)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, gout):
        src, flow = ctx.saved_tensors
        B, C, D, H, W = src.shape
        gout = _c(gout)
        gsrc = torch.empty_like(src) if ctx.needs_input_grad[0] else None
        gflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        with _prof.region("warp3d_bwd", nbytes=4.0 * B * D * H * W * (2 * C + 6 + (C if gsrc is not None else 0))):
            call("vxm_warp3d_bwd", ptr(src), ptr(flow), ptr(gout), ptr(gsrc), ptr(gflow), B, C, D, H, W,
                 INTERP[ctx.mode], stream())
        return gsrc, gflow, None


# VXM_FUSE_UPWARP=0: `fullsize` and the final SpatialTransformer of VxmDense as two kernels with the full-resolution pos_flow between them (rounds 1-5)
FUSE_UPWARP = os.environ.get("VXM_FUSE_UPWARP", "1") != "0"


def warp_up_ok(src, flow_lo):
    """can `transformer(src, fullsize(flow_lo))` run as the fused kernel (vxm_warp3d_up_fwd)?  3-D, a field at most about half as fine as the
    image, and no gradient wanted for src (the fused backward produces none: the images of the path need none)"""
    return (FUSE_UPWARP and src.dim() == 5 and flow_lo.dim() == 5 and not (src.requires_grad and torch.is_grad_enabled())
            and bool(_lib.lib().vxm_warp3d_up_ok(*[int(v) for v in src.shape[2:]], *[int(v) for v in flow_lo.shape[2:]])))


class WarpUpFn(torch.autograd.Function):
    """`transformer(source, fullsize(flow))` (voxelmorph/torch/networks.py:275-280 with layers.py:85-97 and :30-48) as ONE kernel: the
    full-resolution displacement is evaluated in registers from the integrated half-resolution field and never stored.  Same values as
    ResizeFn + WarpFn bit for bit.  Returns the moved image; `want_pos` also returns the displacement (registration=True)."""

    @staticmethod
    def forward(ctx, src, flow_lo, factor, mode, want_pos):
        _vol3(src, "SpatialTransformer")
        require_device(src, flow_lo)
        src, flow_lo = _c(src), _c(flow_lo)
        B, C, D, H, W = src.shape
        lD, lH, lW = (int(v) for v in flow_lo.shape[2:])
        if flow_lo.shape[0] != B or flow_lo.shape[1] != 3 or (D, H, W) != tuple(int(math.floor(s * factor)) for s in (lD, lH, lW)):
            raise ValueError("flow %s resized by %g does not match src %s" % (tuple(flow_lo.shape), factor, tuple(src.shape)))
        out = torch.empty_like(src)
        pos = torch.empty((B, 3, D, H, W), dtype=src.dtype, device=src.device) if want_pos else None
        V, lV = D * H * W, lD * lH * lW
        with _prof.region("warp3d_up_fwd", nbytes=4.0 * B * (V * (2 * C + (3 if want_pos else 0)) + 3 * lV)):
            call("vxm_warp3d_up_fwd", ptr(src), ptr(flow_lo), ptr(out), ptr(pos), B, C, D, H, W, lD, lH, lW, float(factor), INTERP[mode], stream())
        ctx.save_for_backward(src, flow_lo)
        ctx.args = (float(factor), mode)
        if want_pos:
            ctx.mark_non_differentiable(pos)
            return out, pos
        return out

    @staticmethod
    def backward(ctx, gout, *_):
        src, flow_lo = ctx.saved_tensors
        factor, mode = ctx.args
        B, C, D, H, W = src.shape
        lD, lH, lW = (int(v) for v in flow_lo.shape[2:])
        gout = _c(gout)
        gflow = torch.empty_like(flow_lo)
        work = torch.empty((B, 3, D, H, W), dtype=src.dtype, device=src.device)
        V, lV = D * H * W, lD * lH * lW
        # algorithmic bytes of the fused op: src, gout, the low-resolution field and its gradient (the full-resolution gradient the call keeps in
        # `work` between its two kernels is an implementation round trip, not counted)
        with _prof.region("warp3d_up_bwd", nbytes=4.0 * B * (V * 2 * C + 6 * lV)):
            call("vxm_warp3d_up_bwd", ptr(src), ptr(flow_lo), ptr(gout), ptr(gflow), ptr(work), work.numel() * 4, B, C, D, H, W, lD, lH, lW,
                 factor, INTERP[mode], stream())
        return None, gflow, None, None, None


class VecIntFn(torch.autograd.Function):
    """VecInt.forward (voxelmorph/torch/layers.py:64-68), nsteps >= 1."""

    @staticmethod
    def forward(ctx, vec, nsteps):
        _vol3(vec, "VecInt")
        require_device(vec)
        vec = _c(vec)
        B, C, D, H, W = vec.shape
        if C != 3:
            raise ValueError("VecInt expects a 3-channel field, got %d" % C)
        steps = torch.empty((nsteps,) + tuple(vec.shape), dtype=vec.dtype, device=vec.device)
        with _prof.region("vecint_fwd", nbytes=24.0 * B * D * H * W * nsteps):
            call("vxm_vecint_fwd", ptr(vec), ptr(steps), B, D, H, W, nsteps, stream())
        ctx.save_for_backward(vec, steps)
        ctx.nsteps = nsteps
        return steps[nsteps - 1]

    @staticmethod
    def backward(ctx, gout):
        vec, steps = ctx.saved_tensors
        B, _, D, H, W = vec.shape
        gout = _c(gout)
        gvec = torch.empty_like(vec)
        work = torch.empty(vecint_work_elems(vec.shape), dtype=vec.dtype, device=vec.device)
        with _prof.region("vecint_bwd", nbytes=36.0 * B * D * H * W * ctx.nsteps):
            call("vxm_vecint_bwd_ws", ptr(vec), ptr(steps), ptr(gout), ptr(gvec), ptr(work), work.numel() * 4, B, D, H, W, ctx.nsteps, stream())
        return gvec, None


class ResizeFn(torch.autograd.Function):
    """ResizeTransform.forward (voxelmorph/torch/layers.py:85-97), factor != 1."""

    @staticmethod
    def forward(ctx, x, factor):
        _vol3(x, "ResizeTransform")
        require_device(x)
        x = _c(x)
        B, C, D, H, W = x.shape
        oD, oH, oW = (int(math.floor(s * factor)) for s in (D, H, W))
        out = torch.empty((B, C, oD, oH, oW), dtype=x.dtype, device=x.device)
        with _prof.region("resize3d_fwd", nbytes=4.0 * B * C * (D * H * W + oD * oH * oW)):
            call("vxm_resize3d_fwd", ptr(x), ptr(out), B, C, D, H, W, oD, oH, oW, float(factor), stream())
        ctx.shape = (B, C, D, H, W, oD, oH, oW)
        ctx.factor = float(factor)
        return out

    @staticmethod
    def backward(ctx, gout):
        B, C, D, H, W, oD, oH, oW = ctx.shape
        gout = _c(gout)
        gx = torch.empty((B, C, D, H, W), dtype=gout.dtype, device=gout.device)
        with _prof.region("resize3d_bwd", nbytes=4.0 * B * C * (D * H * W + oD * oH * oW)):
            call("vxm_resize3d_bwd", ptr(gout), ptr(gx), B, C, D, H, W, oD, oH, oW, ctx.factor, stream())
        return gx, None


# --------------------------------------------------------------------------- losses
class NCCFn(torch.autograd.Function):
    """NCC.loss (voxelmorph/torch/losses.py:15-67)."""

    @staticmethod
    def forward(ctx, y_true, y_pred, win):
        _vol3(y_true, "NCC")
        require_device(y_true, y_pred)
        I, J = _c(y_true), _c(y_pred)
        B, C, D, H, W = I.shape
        if C != 1 or I.shape != J.shape:
            raise ValueError("NCC: expected two [B,1,D,H,W] tensors (the reference's box filter has one "
                             "input channel, losses.py:29), got %s / %s" % (tuple(I.shape), tuple(J.shape)))
        loss = torch.empty((), dtype=I.dtype, device=I.device)
        fused = bool(_lib.lib().vxm_ncc_fused(B, win))    # the C side decides (window AND batch): fused march keeps (a, b, c), 3 planes
        sums = torch.empty((3 if fused else 5, B, D, H, W), dtype=I.dtype, device=I.device)
        work = torch.empty((1,) if fused else (5, B, D, H, W), dtype=I.dtype, device=I.device)
        acc = torch.empty(1, dtype=torch.float64, device=I.device)
        with _prof.region("ncc_fwd", nbytes=8.0 * B * D * H * W):
            call("vxm_ncc_fwd", ptr(I), ptr(J), ptr(loss), ptr(sums), ptr(work), ptr(acc), B, D, H, W, win, stream())
        ctx.save_for_backward(I, J, sums)
        ctx.win = win
        ctx.fused = fused
        return loss

    @staticmethod
    def backward(ctx, gloss):
        I, J, sums = ctx.saved_tensors
        B, _, D, H, W = I.shape
        gloss = _c(gloss)
        gI = gJ = None
        work = torch.empty((1,) if ctx.fused else (6, B, D, H, W), dtype=I.dtype, device=I.device)
        if ctx.needs_input_grad[1]:
            gJ = torch.empty_like(J)
            with _prof.region("ncc_bwd", nbytes=12.0 * B * D * H * W):
                call("vxm_ncc_bwd", ptr(I), ptr(J), ptr(sums), ptr(gloss), ptr(gJ), ptr(work), B, D, H, W, ctx.win, stream())
        if ctx.needs_input_grad[0]:
            # cc is symmetric in (I, J): swap the roles
            if ctx.fused:              # (a, b, c) were kept for J only: one more forward march with the roles swapped
                swapped = torch.empty_like(sums)
                tmp_loss = torch.empty((), dtype=I.dtype, device=I.device)
                acc = torch.empty(1, dtype=torch.float64, device=I.device)
                call("vxm_ncc_fwd", ptr(J), ptr(I), ptr(tmp_loss), ptr(swapped), ptr(work), ptr(acc), B, D, H, W, ctx.win, stream())
            else:                      # box-sum planes 0<->1 and 2<->3
                swapped = torch.stack([sums[1], sums[0], sums[3], sums[2], sums[4]])
            gI = torch.empty_like(I)
            call("vxm_ncc_bwd", ptr(J), ptr(I), ptr(swapped), ptr(gloss), ptr(gI), ptr(work), B, D, H, W, ctx.win, stream())
        return gI, gJ, None


class NCCWinFn(torch.autograd.Function):
    """NCC.loss (voxelmorph/torch/losses.py:15-67) with ANY window on 1-, 2- or 3-D volumes: the reference pads every axis by
    win[0] // 2 (:31-36), so windows that are not odd and of one size change the extent of the box sums, and with it the
    shape cc is averaged over (:65-67).  `win` is the per-axis list, len(win) == number of spatial axes."""

    @staticmethod
    def forward(ctx, y_true, y_pred, win):
        require_device(y_true, y_pred)
        I, J = _c(y_true), _c(y_pred)
        nd = I.dim() - 2
        if nd not in (1, 2, 3) or I.shape[1] != 1 or I.shape != J.shape or len(win) != nd:
            raise ValueError("NCC: expected two [B,1,*vol] tensors (the reference's box filter has one input channel, losses.py:29) "
                             "and one window size per axis, got %s / %s, win=%s" % (tuple(I.shape), tuple(J.shape), list(win)))
        B = I.shape[0]
        size = [1] * (3 - nd) + [int(v) for v in I.shape[2:]]
        w3 = [1] * (3 - nd) + [int(v) for v in win]
        pad = int(win[0]) // 2                                            # losses.py:31
        p3 = [0] * (3 - nd) + [pad] * nd
        geo = (B, *size, *w3, *p3)
        n_out = ctypes.c_int64(0)
        plane = int(_lib.lib().vxm_ncc_win_elems(*geo, ctypes.byref(n_out)))
        if plane <= 0:
            raise ValueError("NCC: window %s with padding %d leaves no box sums on a volume of shape %s (the reference's conv raises "
                             "for a kernel larger than the padded input too)" % (list(win), pad, tuple(I.shape[2:])))
        loss = torch.empty((), dtype=I.dtype, device=I.device)
        sums = torch.empty((5, n_out.value), dtype=I.dtype, device=I.device)
        work = torch.empty(10 * plane, dtype=I.dtype, device=I.device)
        acc = torch.empty(1, dtype=torch.float64, device=I.device)
        call("vxm_ncc_win_fwd", ptr(I), ptr(J), ptr(loss), ptr(sums), ptr(work), ptr(acc), *geo, stream())
        ctx.save_for_backward(I, J, sums)
        ctx.geo, ctx.plane = geo, plane
        return loss

    @staticmethod
    def backward(ctx, gloss):
        I, J, sums = ctx.saved_tensors
        gloss = _c(gloss)
        work = torch.empty(6 * ctx.plane, dtype=I.dtype, device=I.device)
        gI = gJ = None
        if ctx.needs_input_grad[1]:
            gJ = torch.empty_like(J)
            call("vxm_ncc_win_bwd", ptr(I), ptr(J), ptr(sums), ptr(gloss), ptr(gJ), ptr(work), *ctx.geo, stream())
        if ctx.needs_input_grad[0]:      # cc is symmetric in (I, J): box-sum planes 0<->1 and 2<->3
            swapped = torch.stack([sums[1], sums[0], sums[3], sums[2], sums[4]])
            gI = torch.empty_like(I)
            call("vxm_ncc_win_bwd", ptr(J), ptr(I), ptr(swapped), ptr(gloss), ptr(gI), ptr(work), *ctx.geo, stream())
        return gI, gJ, None


class GradLossFn(torch.autograd.Function):
    """Grad.loss (voxelmorph/torch/losses.py:102-135)."""

    @staticmethod
    def forward(ctx, y, penalty, mult):
        _vol3(y, "Grad")
        require_device(y)
        y = _c(y)
        B, C, D, H, W = y.shape
        loss = torch.empty((), dtype=y.dtype, device=y.device)
        acc = torch.empty(3 * B * 32, dtype=torch.float64, device=y.device)      # 3 * B * VXM_GRAD_SLOTS (include/vxm_hip.h)
        call("vxm_gradloss_fwd", ptr(y), ptr(loss), ptr(acc), B, C, D, H, W, PENALTY[penalty], float(mult), stream())
        ctx.save_for_backward(y)
        ctx.args = (penalty, float(mult))
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (y,) = ctx.saved_tensors
        B, C, D, H, W = y.shape
        gy = torch.empty_like(y)
        call("vxm_gradloss_bwd", ptr(y), ptr(_c(gloss)), ptr(gy), B, C, D, H, W, PENALTY[ctx.args[0]], ctx.args[1], stream())
        return gy, None, None


class MSEFn(torch.autograd.Function):
    """MSE.loss (voxelmorph/torch/losses.py:75-76)."""

    @staticmethod
    def forward(ctx, a, b):
        require_device(a, b)
        a, b = _c(a), _c(b)
        if a.shape != b.shape:
            raise ValueError("MSE: shapes differ %s / %s" % (tuple(a.shape), tuple(b.shape)))
        loss = torch.empty((), dtype=a.dtype, device=a.device)
        acc = torch.empty(1, dtype=torch.float64, device=a.device)
        call("vxm_mse_fwd", ptr(a), ptr(b), ptr(loss), ptr(acc), a.numel(), stream())
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        call("vxm_mse_bwd", ptr(a), ptr(b), ptr(_c(gloss)), ptr(ga), ptr(gb), a.numel(), stream())
        return ga, gb


class DiceFn(torch.autograd.Function):
    """Dice.loss (voxelmorph/torch/losses.py:84-90)."""

    @staticmethod
    def forward(ctx, y_true, y_pred):
        require_device(y_true, y_pred)
        yt, yp = _c(y_true), _c(y_pred)
        if yt.shape != yp.shape or yt.dim() < 3:
            raise ValueError("Dice: expected two [B,C,*vol] tensors of one shape")
        B, C = yt.shape[:2]
        V = yt[0, 0].numel()
        loss = torch.empty((), dtype=yt.dtype, device=yt.device)
        acc = torch.empty(2 * B * C, dtype=torch.float64, device=yt.device)
        call("vxm_dice_fwd", ptr(yt), ptr(yp), ptr(loss), ptr(acc), B, C, V, stream())
        ctx.save_for_backward(yt, yp, acc)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        yt, yp, acc = ctx.saved_tensors
        B, C = yt.shape[:2]
        gyt = torch.empty_like(yt) if ctx.needs_input_grad[0] else None
        gyp = torch.empty_like(yp) if ctx.needs_input_grad[1] else None
        call("vxm_dice_bwd", ptr(yt), ptr(yp), ptr(acc), ptr(_c(gloss)), ptr(gyt), ptr(gyp), B, C, yt[0, 0].numel(), stream())
        return gyt, gyp


class ForkFn(torch.autograd.Function):
    """x -> (x, x) for a tensor that two ops consume: autograd would add the two incoming gradients with an ATen kernel; here that sum is
    vxm_add2.  `preint_flow` (networks.py:262-268) is regularised by Grad AND integrated by VecInt."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        require_device(ga, gb)
        ga, gb = _c(ga), _c(gb)
        out = torch.empty_like(ga)
        call("vxm_add2", ptr(ga), ptr(gb), ptr(out), ga.numel(), stream())
        return out


class WeightedSumFn(torch.autograd.Function):
    """`loss = 0; loss += loss_function(y_true[n], y_pred[n]) * weights[n]` (scripts/torch/train.py:205-212) as one launch forward and one
    backward (vxm_loss_combine_fwd / _bwd) instead of a mul + an add of ATen per term.  `running` (optional, n + 1 floats on the device)
    accumulates the weighted terms and the total for a per-epoch log."""

    @staticmethod
    def forward(ctx, weights, running, *terms):
        require_device(*terms)
        n = len(terms)
        if n < 1 or n > 8 or len(weights) != n or any(t.numel() != 1 for t in terms):
            raise ValueError("weighted_sum: 1 .. 8 scalar loss terms with one weight each, got %d terms / %d weights" % (n, len(weights)))
        if running is not None:
            require_device(running)
            if running.numel() < n + 1 or not running.is_contiguous():
                raise ValueError("weighted_sum: running needs %d contiguous floats" % (n + 1))
        terms = [_c(t) for t in terms]
        tp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
        wv = (ctypes.c_float * n)(*[float(w) for w in weights])
        total = torch.empty((), dtype=terms[0].dtype, device=terms[0].device)
        call("vxm_loss_combine_fwd", ctypes.cast(tp, ctypes.c_void_p), ctypes.cast(wv, ctypes.c_void_p), n, ptr(total), ptr(running), stream())
        ctx.weights = [float(w) for w in weights]
        ctx.shapes = [t.shape for t in terms]
        return total

    @staticmethod
    def backward(ctx, gtotal):
        n = len(ctx.weights)
        gtotal = _c(gtotal)
        wv = (ctypes.c_float * n)(*ctx.weights)
        gterms = torch.empty(n, dtype=gtotal.dtype, device=gtotal.device)
        call("vxm_loss_combine_bwd", ptr(gtotal), ctypes.cast(wv, ctypes.c_void_p), n, ptr(gterms), stream())
        return (None, None) + tuple(gterms[i].view(ctx.shapes[i]) if ctx.needs_input_grad[2 + i] else None for i in range(n))


# --------------------------------------------------------------------------- conv helpers
def pack_weights(w, flip, lo=0, hi=None):
    """[Cout,Cin,3,3,3] -> MFMA A-fragment order of the operator on the input channels [lo, hi) (forward), or of its adjoint
    onto them (flip: backward-data of that segment); the channel range is read in place, no sliced copy of w."""
    cout, cin = w.shape[:2]
    hi = cin if hi is None else hi
    n = _lib.lib().vxm_conv3d_k3_packed_elems(cout if flip else hi - lo, hi - lo if flip else cout)
    wp = torch.empty(n, dtype=w.dtype, device=w.device)
    call("vxm_conv3d_k3_pack_weights_range", ptr(_c(w)), ptr(wp), cin, cout, lo, hi - lo, 1 if flip else 0, stream())
    return wp


def pack_weights_cached(w, flip, lo=0, hi=None):
    """pack_weights with the packed operator cached on the weight tensor until its version moves (as the split operators): the fused U-Net
    builds the ones it will ask for on the second stream at the start of a step (_prepack_plan) instead of one small launch in front of
    every fp32-MFMA conv of the main chain"""
    cin = w.shape[1]
    hi = cin if hi is None else hi
    cache = w.__dict__.setdefault("_vxm_s3_packs", {})
    key = ("fp32", bool(flip), lo, hi)
    hit = cache.get(key)
    if hit is not None and hit[0] == _pack_ver(w) and hit[1].device == w.device:
        return hit[1]
    cout = w.shape[0]
    n = _lib.lib().vxm_conv3d_k3_packed_elems(cout if flip else hi - lo, hi - lo if flip else cout)
    wp = hit[1] if hit is not None and hit[1].device == w.device and hit[1].numel() == n else None
    if wp is None:
        wp = torch.empty(n, dtype=w.dtype, device=w.device)
        _FRESH_PACKS.append(wp)
    call("vxm_conv3d_k3_pack_weights_range", ptr(_c(w)), ptr(wp), cin, cout, lo, hi - lo, 1 if flip else 0, stream())
    cache[key] = (_pack_ver(w), wp)
    return wp


def _pack_ver(t):
    """cache key of a packed copy of weight tensor t: its version counter + the generation bumped by writers that bypass it"""
    from .functional_bf16 import _ver
    return _ver(t)


def s3_route(c0, up0, c1, cout, B, D, H, W):
    """does this conv launch (forward operator over the virtual concat, or an adjoint as a forward) go to the split kernel?"""
    if not split_engine():
        return False
    # cat([upsample(x0), x1]): through the upsampling gather the split kernel executes the nominal FLOPs, the fp32 kernel 8 / 27 of the
    # upsampled segment's -- measured, the split kernel wins when the skip segment is at least as wide as the upsampled one (dec3 at L1:
    # 0.42 vs 0.53 ms) and ties when it is half as wide (rem0: 2.60 vs 2.56 ms: stays on the collapsed kernel); VXM_S3_UP=1 forces it
    if up0 and not (S3_UP or c1 >= c0):
        return False
    return bool(_lib.lib().vxm_conv3d_k3_s3_ok(c0, c1, cout, B, D, H, W))


_RANGE_PROBE = None    # voxelmorph_amd.diagnostics.range_report(): list of (tag, four doubles on the device) while a report is being taken


def _probe(tag, t, C, blocked):
    """dynamic range of one tensor the split kernels read, as the fp16-piece scheme sees it (csrc/diag.hip); only under range_report()"""
    if _RANGE_PROBE is None or C < 8 or t.dim() != 5:
        return
    out = torch.zeros(4, dtype=torch.float64, device=t.device)
    B, _, D, H, W = t.shape
    call("vxm_s3_range_probe", ptr(t), C, t[0].numel(), 1 if blocked else 0, B, D, H, W, ptr(out), stream())
    _RANGE_PROBE.append((tag, out))


_FRESH_PACKS = []      # pack buffers allocated since the last _adopt_fresh_packs(): allocated under the second stream, read on the main one


def _new_pack(nbytes, device):
    t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    _FRESH_PACKS.append(t)
    return t


def _adopt_fresh_packs(stream_):
    """tell the caching allocator that `stream_` reads the pack buffers allocated (under another stream) since the last call"""
    while _FRESH_PACKS:
        t = _FRESH_PACKS.pop()
        if stream_ is not None and t.is_cuda:
            t.record_stream(stream_)


def s3_prepack(jobs):
    """Pack every stale split operator of `jobs` = [(w, lo, hi, flip, seg0), ...] in ONE launch (vxm_conv3d_k3_s3_pack_weights_batch).
    The packed, pre-split copies are cached on the weight tensor until its version moves (once per optimiser step)."""
    import ctypes
    stale = []
    for w, lo, hi, flip, seg0 in jobs:
        cache = w.__dict__.setdefault("_vxm_s3_packs", {})
        key = (lo, hi, bool(flip), seg0, s3_pieces())
        hit = cache.get(key)
        if hit is not None and hit[0] == _pack_ver(w) and hit[1].device == w.device:
            continue
        if any(k is cache and kk == key for (_, _, _, _, _, _, _, _, k, kk, _) in stale):
            continue
        cout, cin = w.shape[:2]
        inc, outc = (cout, hi - lo) if flip else (hi - lo, cout)
        nbytes = _lib.lib().vxm_conv3d_k3_s3_packed_bytes(seg0, inc - seg0, outc, s3_pieces())
        wp = hit[1] if hit is not None and hit[1].device == w.device and hit[1].numel() == nbytes else _new_pack(nbytes, w.device)
        stale.append((_c(w), cin, cout, lo, hi - lo, bool(flip), seg0, wp, cache, key, _pack_ver(w)))
    if not stale:
        return
    table = (_lib.S3PackJob * len(stale))()
    for j, (w, cin, cout, lo, n, flip, seg0, wp, _, _, _) in enumerate(stale):
        table[j] = _lib.S3PackJob(w.data_ptr(), wp.data_ptr(), cin, cout, lo, n, 1 if flip else 0, seg0, s3_pieces())
    call("vxm_conv3d_k3_s3_pack_weights_batch", ctypes.cast(table, ctypes.c_void_p), len(stale), stream())
    for _, _, _, _, _, _, _, wp, cache, key, ver in stale:
        cache[key] = (ver, wp)


def s3_pack(w, flip, lo, hi, seg0):
    s3_prepack([(w, lo, hi, flip, seg0)])
    return w.__dict__["_vxm_s3_packs"][(lo, hi, bool(flip), seg0, s3_pieces())][1]


def s3u_route(c0, c1, cout, B, D, H, W):
    """does a cat([upsample(x0), x1]) forward go to the split + collapsed kernel?"""
    return S3U and split_engine() and bool(_lib.lib().vxm_conv3d_k3_s3u_ok(c0, c1, cout, B, D, H, W, s3_pieces()))


def s3u_pack(w, c0, c1):
    """packed operator of the split + collapsed forward kernel (collapsed 2x2x2 weights per output parity class for the upsampled
    segment, the 27 taps for the skip segment, pre-split), cached on the weight tensor until its version moves"""
    cache = w.__dict__.setdefault("_vxm_s3_packs", {})
    key = ("s3u", c0, c1, s3_pieces())
    hit = cache.get(key)
    if hit is not None and hit[0] == _pack_ver(w) and hit[1].device == w.device:
        return hit[1]
    cout = w.shape[0]
    nbytes = _lib.lib().vxm_conv3d_k3_s3u_packed_bytes(c0, c1, cout, s3_pieces())
    wp = hit[1] if hit is not None and hit[1].device == w.device and hit[1].numel() == nbytes else _new_pack(nbytes, w.device)
    call("vxm_conv3d_k3_s3u_pack_weights", ptr(_c(w)), ptr(wp), c0, c1, cout, s3_pieces(), stream())
    cache[key] = (_pack_ver(w), wp)
    return wp


def s3u_launch(x0, c0, bs0, x1, c1, bs1, wp, bias, y, ybs, cout, slope, B, D, H, W, lay=0, signs=None):
    name = None
    if _prof.ACTIVE is not None:
        pc = _lib.lib().vxm_conv3d_k3_s3u_fwd_kernel(bs0, bs1, D, H, W, s3_pieces())
        name = ("k_s3u_conv_pc<%d>" % (1 if cout <= 16 else 2)) if pc else "k_s3u_conv<%d,%d>" % (1 if cout <= 16 else 2, s3_pieces())
    with _prof.region(name, flops=2.0 * (8 * c0 + 27 * c1) * cout * B * D * H * W, nominal=2.0 * 27 * (c0 + c1) * cout * B * D * H * W):
        if signs is not None:
            call("vxm_conv3d_k3_s3u_fwd_signs", ptr(x0), c0, bs0, ptr(x1), c1, bs1, ptr(wp), ptr(bias), ptr(y), ybs, cout, float(slope), B, D, H, W,
                 s3_pieces() | lay, ptr(signs), (cout // 4) * D * H * W, stream())
        else:
            call("vxm_conv3d_k3_s3u_fwd", ptr(x0), c0, bs0, ptr(x1), c1, bs1, ptr(wp), ptr(bias), ptr(y), ybs, cout, float(slope), B, D, H, W,
                 s3_pieces() | lay, stream())


def s3u_bwd_low_route(c0, cout, B, D, H, W):
    """does the backward-data of an upsampled segment go to the split kernel (csrc/conv_s3u.hip: k_s3u_dlow)?"""
    return S3U and split_engine() and bool(_lib.lib().vxm_conv3d_k3_s3u_bwd_low_ok(c0, cout, B, D, H, W, s3_pieces()))


def s3u_bwd_low_pack(w, c0, cin):
    """packed transposed-collapsed operator of k_s3u_dlow for the first c0 input channels of w [cout][cin][27], cached on the weight tensor until
    its version moves"""
    cout = w.shape[0]
    cache = w.__dict__.setdefault("_vxm_s3_packs", {})
    key = ("s3u_low", c0, cin, s3_pieces())
    hit = cache.get(key)
    if hit is None or hit[0] != _pack_ver(w) or hit[1].device != w.device:
        nbytes = _lib.lib().vxm_conv3d_k3_s3u_bwd_low_packed_bytes(c0, cout, s3_pieces())
        wp = hit[1] if hit is not None and hit[1].device == w.device and hit[1].numel() == nbytes else _new_pack(nbytes, w.device)
        call("vxm_conv3d_k3_s3u_bwd_low_pack_weights", ptr(_c(w)), ptr(wp), c0, cin, cout, s3_pieces(), stream())
        cache[key] = hit = (_pack_ver(w), wp)
    return hit[1]


def s3u_bwd_low(dz, cout, w, c0, cin, gxl, mask, mask_slope, B, D, H, W, lay=0):
    """gxl [B,c0,D/2,H/2,W/2] = LeakyReLU'(mask) * (conv backward + upsample backward of dz [B,cout,D,H,W]) for the first c0 input channels
    of w [cout][cin][27]"""
    wp = s3u_bwd_low_pack(w, c0, cin)
    V = D * H * W
    with _prof.region("k_s3u_dlow<%d,%d>" % (1 if c0 <= 16 else 2, s3_pieces()), flops=2.0 * 8 * c0 * cout * B * V, nominal=2.0 * 27 * c0 * cout * B * V):
        call("vxm_conv3d_k3_s3u_bwd_low", ptr(dz), cout * V, cout, ptr(wp), ptr(gxl), c0 * (V // 8), c0, ptr(mask), c0 * (V // 8), float(mask_slope),
             B, D, H, W, s3_pieces() | lay, stream())


def s3u_bwd_data_route(c0, c1, cout, B, D, H, W):
    """do BOTH backward-data products of a cat([upsample(x0), x1]) layer come from one staging of dz (csrc/conv_s3u.hip: k_s3u_bwd_pc)?"""
    return S3U and split_engine() and bool(_lib.lib().vxm_conv3d_k3_s3u_bwd_data_ok(c0, c1, cout, B, D, H, W, s3_pieces()))


def s3u_bwd_skip_pack(w, c0, c1):
    """packed adjoint operator of the skip channels [c0, c0 + c1) of w [cout][c0 + c1][27] for k_s3u_bwd_pc, cached on the weight tensor until its
    version moves"""
    cout = w.shape[0]
    cache = w.__dict__.setdefault("_vxm_s3_packs", {})
    key = ("s3u_skip_adj", c0, c1, s3_pieces())
    hit = cache.get(key)
    if hit is None or hit[0] != _pack_ver(w) or hit[1].device != w.device:
        nbytes = _lib.lib().vxm_conv3d_k3_s3u_bwd_skip_packed_bytes(c1, cout, s3_pieces())
        wp = hit[1] if hit is not None and hit[1].device == w.device and hit[1].numel() == nbytes else _new_pack(nbytes, w.device)
        call("vxm_conv3d_k3_s3u_bwd_skip_pack_weights", ptr(_c(w)), ptr(wp), c0, c1, cout, s3_pieces(), stream())
        cache[key] = hit = (_pack_ver(w), wp)
    return hit[1]


def s3u_bwd_data(dz, cout, w, c0, c1, gxl, mask, mask_slope, gx1, B, D, H, W, lay=0):
    """gxl [B,c0,D/2,H/2,W/2] as s3u_bwd_low, and gx1 [B,c1,D,H,W] = conv backward of dz onto the skip channels of w [cout][c0 + c1][27], from one
    staging of dz [B,cout,D,H,W]"""
    wl, wk = s3u_bwd_low_pack(w, c0, c0 + c1), s3u_bwd_skip_pack(w, c0, c1)
    V = D * H * W
    with _prof.region("k_s3u_bwd_pc<%d,%d>" % (1 if c0 <= 16 else 2, 1 if c1 <= 16 else 2), flops=2.0 * (8 * c0 + 27 * c1) * cout * B * V,
                      nominal=2.0 * 27 * (c0 + c1) * cout * B * V):
        call("vxm_conv3d_k3_s3u_bwd_data", ptr(dz), cout * V, cout, ptr(wl), ptr(gxl), c0 * (V // 8), c0, ptr(mask), c0 * (V // 8), float(mask_slope),
             ptr(wk), ptr(gx1), c1 * V, c1, B, D, H, W, s3_pieces() | lay, stream())


# layout flags of include/vxm_hip.h (OR-ed into `pieces`): a flagged tensor is channel-blocked [B][C/8][D][H][W][8]
S3_IN0_BLOCKED, S3_IN1_BLOCKED, S3_OUT_BLOCKED = 0x100, 0x200, 0x400
S3_REVERSE_TILES = 0x4000                                     # scheduling hint of vxm_conv3d_k3_s3_fwd (include/vxm_hip.h)
S3_MASK_SIGNS, S3_OUT_SIGNS = 0x8000, 0x10000                 # sign tensors (include/vxm_hip.h): [B][C/4][D][H][W] bytes, bit j = (y[4 q + j] > 0)
# channel-blocked activations of the fused U-Net get a sign tensor from their forward epilogue; the backward-data epilogue that applies their
# LeakyReLU' reads it instead of the fp32 activation (round 6: that mask was 0.12 of rem1's 0.68 ms and 0.045 of rem2's 0.375).  VXM_S3_SIGNS=0: off
SIGNS = os.environ.get("VXM_S3_SIGNS", "1") != "0"
# consecutive full-resolution launches of the fused U-Net walk their tensors in alternating directions (a consumer starts where its producer
# stopped: the last-written part of a 440 - 880 MB tensor is what the 256 MB memory-side cache still holds); VXM_S3_SNAKE=0: all forward
SNAKE = os.environ.get("VXM_S3_SNAKE", "1") != "0"
S3_BW_CONTRACT_ONLY, S3_BW_REDUCE_ONLY = 0x1000, 0x2000       # phase flags of the split backward-weight entry points (include/vxm_hip.h)


def to_blocked(x):
    """NCDHW -> channel-blocked [B][C/8][D][H][W][8] (same shape attribute, different element order); tests and tools"""
    B, C = x.shape[:2]
    return x.reshape(B, C // 8, 8, -1).permute(0, 1, 3, 2).contiguous().view(x.shape)


def from_blocked(x):
    B, C = x.shape[:2]
    return x.reshape(B, C // 8, -1, 8).permute(0, 1, 3, 2).contiguous().view(x.shape)


def s3_launch(x0, c0, bs0, up0, x1, c1, bs1, wp, bias, y, ybs, cout, slope, mask, mask_bs, mask_slope, B, D, H, W, lay=0):
    v = _lib.lib().vxm_conv3d_k3_s3_variant(cout)
    name = None
    if _prof.ACTIVE is not None:          # label the region with the kernel the C ABI will dispatch to
        rows = _lib.lib().vxm_conv3d_k3_s3_tile_rows_at(cout, s3_pieces(), B, D, H, W)
        pc = _lib.lib().vxm_conv3d_k3_s3_producer_consumer(cout, s3_pieces(), 0 if (mask is None or lay & (S3_MASK_SIGNS | S3_OUT_SIGNS)) else 1, B, D, H, W)
        name = "k_s3p_conv<%d,%d>" % (v // 10, s3_pieces()) if pc else "k_s3_conv<%d,%d,%d,%d>" % (v // 10, rows, v % 10, s3_pieces())
    with _prof.region(name, flops=2.0 * 27 * (c0 + c1) * cout * B * D * H * W):
        call("vxm_conv3d_k3_s3_fwd", ptr(x0), c0, bs0, 1 if up0 else 0, ptr(x1), c1, bs1, ptr(wp), ptr(bias), ptr(y), ybs,
             cout, float(slope), ptr(mask), mask_bs, float(mask_slope), B, D, H, W, s3_pieces() | lay, stream())


def conv_launch(x0, c0, bs0, up0, x1, c1, bs1, wp, bias, y, ybs, cout, slope, mask, mask_bs, mask_slope, B, D, H, W, lay=0):
    """lay: S3_OUT_BLOCKED (| S3_MASK_SIGNS) on launches the few-input-channel kernel takes (vxm_conv3d_k3_fwd_layout_ok), else 0"""
    name = None
    if _prof.ACTIVE is not None:          # label the region with the kernel the C ABI will dispatch to
        v = _lib.lib().vxm_conv3d_k3_fwd_variant(ptr(x0), c0, bs0, ptr(x1), c1, bs1, ptr(wp), cout, B, D, H, W)
        name = "k_conv3d_k3_kpack<%d>" % (v - 200) if 200 <= v < 300 else (
            "k_conv3d_k3_sm<%d>" % (v - 300) if v >= 300 else (
            "k_conv3d_k3_t8<%d>" % (v % 10) if v >= 100 else "k_conv3d_k3<%d,%d>" % (v // 10, v % 10)))
    with _prof.region(name, flops=2.0 * 27 * (c0 + c1) * cout * B * D * H * W):
        if lay:
            call("vxm_conv3d_k3_fwd_layout", ptr(x0), c0, bs0, 1 if up0 else 0, ptr(x1), c1, bs1, ptr(wp), ptr(bias), ptr(y), ybs,
                 cout, float(slope), ptr(mask), mask_bs, float(mask_slope), B, D, H, W, lay, stream())
        else:
            call("vxm_conv3d_k3_fwd", ptr(x0), c0, bs0, 1 if up0 else 0, ptr(x1), c1, bs1, ptr(wp), ptr(bias), ptr(y), ybs,
                 cout, float(slope), ptr(mask), mask_bs, float(mask_slope), B, D, H, W, stream())


def _need_split_kernel(lay, what):
    if lay:
        raise RuntimeError("%s: a channel-blocked operand (layout flags 0x%x) needs the split kernels; the plan of the fused U-Net only marks "
                           "tensors whose producer and consumers run on them" % (what, lay))


def conv_forward(x0, c0, bs0, up0, x1, c1, bs1, w, bias, y, ybs, cout, slope, B, D, H, W, lay=0, signs=None):
    """ConvBlock / flow conv forward (networks.py:299-305, 211,257) from the reference-layout weights: the MFMA implicit
    GEMM (weights packed per call), or the vector-ALU kernel when there are at most 4 output channels (flow conv).
    lay: S3_IN0_BLOCKED (x0) / S3_OUT_BLOCKED (y) for channel-blocked tensors between split kernels (fused U-Net only)."""
    if (lay & ~S3_IN0_BLOCKED) == 0 and cout <= 4 and x1 is None and not up0 and _lib.lib().vxm_conv3d_k3_fewout_ok(ptr(x0), bs0, ptr(y), ybs, c0, cout, W):
        with _prof.region("k_conv3d_k3_fewout<%d>" % cout, flops=2.0 * 27 * c0 * cout * B * D * H * W):
            if lay:      # x0 channel-blocked: the last activation of the fused U-Net (_blocked_tensors)
                call("vxm_conv3d_k3_fewout_fwd_layout", ptr(x0), c0, bs0, ptr(_c(w)), ptr(bias), ptr(y), ybs, cout, float(slope), B, D, H, W, lay, stream())
            else:
                call("vxm_conv3d_k3_fewout_fwd", ptr(x0), c0, bs0, ptr(_c(w)), ptr(bias), ptr(y), ybs, cout, float(slope), B, D, H, W, stream())
        return
    if signs is not None and not (lay & S3_OUT_BLOCKED):
        raise RuntimeError("conv_forward: a sign tensor goes with a channel-blocked output")
    if up0 and s3u_route(c0, c1, cout, B, D, H, W):
        s3u_launch(x0, c0, bs0, x1, c1, bs1, s3u_pack(w, c0, c1), bias, y, ybs, cout, slope, B, D, H, W, lay=lay, signs=signs)
        return
    if s3_route(c0, up0, c1, cout, B, D, H, W):
        if signs is not None:        # (a forward launch has no mask: the sign tensor of its output travels in that argument, flag OUT_SIGNS)
            s3_launch(x0, c0, bs0, up0, x1, c1, bs1, s3_pack(w, False, 0, c0 + c1, c0), bias, y, ybs, cout, slope, signs, (cout // 4) * D * H * W, 1.0,
                      B, D, H, W, lay=lay | S3_OUT_SIGNS)
        else:
            s3_launch(x0, c0, bs0, up0, x1, c1, bs1, s3_pack(w, False, 0, c0 + c1, c0), bias, y, ybs, cout, slope, None, 0, 1.0, B, D, H, W, lay=lay)
        return
    _need_split_kernel(lay, "conv_forward")
    if up0 and _lib.lib().vxm_conv3d_k3_up_ok(ptr(x0), c0, bs0, ptr(x1), c1, bs1, ptr(y), cout, B, D, H, W):
        # upsampled segment at low-resolution cost: collapsed 2x2x2 weights per output parity (conv_fwd.hip: k_conv3d_k3_t8u)
        wp = torch.empty(_lib.lib().vxm_conv3d_k3_up_packed_elems(c0, c1, cout), dtype=w.dtype, device=w.device)
        call("vxm_conv3d_k3_up_pack_weights", ptr(_c(w)), ptr(wp), c0, c1, cout, stream())
        with _prof.region("k_conv3d_k3_t8u<1>", flops=2.0 * (8 * c0 + 27 * c1) * cout * B * D * H * W,
                          nominal=2.0 * 27 * (c0 + c1) * cout * B * D * H * W):
            call("vxm_conv3d_k3_up_fwd", ptr(x0), c0, bs0, ptr(x1), c1, bs1, ptr(wp), ptr(bias), ptr(y), ybs, cout, float(slope),
                 B, D, H, W, stream())
        return
    conv_launch(x0, c0, bs0, up0, x1, c1, bs1, pack_weights_cached(w, False), bias, y, ybs, cout, slope, None, 0, 1.0, B, D, H, W)


def conv_bwd_data(dz, cout, w, gx, cin, mask, mask_slope, B, D, H, W, w_lo=0, lay=0):
    """convolution_backward w.r.t. the input = the forward kernel with the flipped / transposed weights
    (networks.py:299 autograd twin), optionally multiplied by LeakyReLU'(mask) of the previous ConvBlock.
    A 48-channel result (16 mod 32) is produced as 32 + 16 channels: two launches of the 8-wave kernel's 2- and
    1-tile instances beat one launch of the 3-tile instance (which only exists in the 4-wave kernel)."""
    V = D * H * W
    bounds = _bwd_bounds(cin)
    for lo, hi in zip(bounds[:-1], bounds[1:]):         # w_lo: gx covers the input channels [w_lo, w_lo + cin) of w
        if s3_route(cout, False, 0, hi - lo, B, D, H, W):
            if lay & S3_OUT_BLOCKED and (lo, hi) != (0, cin):   # (S3_REVERSE_TILES needs no such care: every launch of the pair walks backwards)
                raise RuntimeError("conv_bwd_data: a channel-blocked gradient is written by one launch")
            if lay & S3_MASK_SIGNS:      # `mask` is the sign tensor [B][cin / 4][D][H][W] (bytes) of the activation; one launch (blocked output)
                s3_launch(dz, cout, cout * V, False, None, 0, 0, s3_pack(w, True, w_lo + lo, w_lo + hi, cout), None, gx, cin * V, cin,
                          1.0, mask, (cin // 4) * V, mask_slope, B, D, H, W, lay=lay)
                continue
            s3_launch(dz, cout, cout * V, False, None, 0, 0, s3_pack(w, True, w_lo + lo, w_lo + hi, cout), None, gx[:, lo:hi], cin * V, hi - lo,
                      1.0, mask[:, lo:hi] if mask is not None else None, cin * V, mask_slope, B, D, H, W, lay=lay)
            continue
        lay &= ~S3_REVERSE_TILES                         # a scheduling hint of the split kernels only
        if lay & S3_OUT_BLOCKED and not (lay & S3_IN0_BLOCKED) and (lo, hi) == (0, cin):
            # the flow conv's adjoint (3 -> 16, few-input-channel kernel) onto a channel-blocked gradient, LeakyReLU' from the activation in the
            # same layout or from its sign tensor
            wpk = pack_weights_cached(w, True, w_lo, w_lo + cin)
            if _lib.lib().vxm_conv3d_k3_fwd_layout_ok(ptr(dz), cout, cout * V, None, 0, 0, ptr(wpk), cin, B, D, H, W):
                conv_launch(dz, cout, cout * V, False, None, 0, 0, wpk, None, gx, cin * V, cin, 1.0, mask,
                            ((cin // 4) * V if lay & S3_MASK_SIGNS else cin * V) if mask is not None else 0, mask_slope, B, D, H, W, lay=lay)
                continue
        _need_split_kernel(lay, "conv_bwd_data")
        conv_launch(dz, cout, cout * V, False, None, 0, 0, pack_weights_cached(w, True, w_lo + lo, w_lo + hi), None, gx[:, lo:hi], cin * V, hi - lo, 1.0,
                    mask[:, lo:hi] if mask is not None else None, cin * V, mask_slope, B, D, H, W)


def _versions(tensors):
    """Version counters for the hand-made in-place-modification check of the fused engines (None for inference tensors,
    which have none: reading `._version` of a tensor made under torch.inference_mode() raises)."""
    return [None if t.is_inference() else t._version for t in tensors]


def _claim_sink(p):
    """The optimiser's flat-bucket view of parameter p if it has not been written since the last zero_grad()."""
    sink = getattr(p, "_vxm_grad_sink", None)
    if sink is None or getattr(p, "_vxm_sink_written", False):
        return None
    p._vxm_sink_written = True
    return sink


class _Workspace:
    """Grow-only scratch shared by the bwd-weight launches of one backward pass.  `deferred`: a list that collects the reductions of the
    split weight-gradient kernels instead of running them behind their contraction (each then keeps a buffer of its own until it has run:
    UnetFn.backward sends them to a third stream, see VXM_S3_BW_CONTRACT_ONLY in include/vxm_hip.h)."""

    def __init__(self, device, deferred=None):
        self.buf = None
        self.device = device
        self.deferred = deferred

    def get(self, nbytes):
        if self.deferred is not None:
            return torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self.buf


def s3_bwd_weight(ws, x, c, bs, dz, cout, gw, gw_cin, ci_off, gb, B, D, H, W, lay=0):
    """weight / bias gradient of one full-resolution tensor on the split kernel (vxm_conv3d_k3_s3_bwd_weight)"""
    need = _lib.lib().vxm_conv3d_k3_s3_bwd_weight_workspace_bytes(c, cout, B, D, H, W)
    buf = ws.get(need)
    args = (ptr(x), c, bs, ptr(dz), cout * D * H * W, cout, ptr(gw), gw_cin, ci_off, ptr(gb), ptr(buf), buf.numel(), B, D, H, W)
    flags = s3_pieces() | lay
    name = None
    if _prof.ACTIVE is not None:               # the region carries the name of the kernel the library launches for this shape (profiles/*_counters.json are keyed by it)
        kern = _lib.lib().vxm_conv3d_k3_s3_bwd_weight_kernel(c, cout, s3_pieces())
        name = {0: "k_s3_bwd_weight<%d>" % s3_pieces(), 1: "k_s3_bww_pc<false,2>", 2: "k_s3_bww_pc<true,2>", 3: "k_s3_bww_pc<false,1>"}[kern]
    with _prof.region(name, flops=2.0 * 27 * c * cout * B * D * H * W):
        call("vxm_conv3d_k3_s3_bwd_weight", *args, flags | (S3_BW_CONTRACT_ONLY if ws.deferred is not None else 0), stream())
    if ws.deferred is not None:
        ws.deferred.append(("vxm_conv3d_k3_s3_bwd_weight", args, flags | S3_BW_REDUCE_ONLY, (x, dz, gw, gb, buf)))


def s3u_bwd_weight_route(c0, cout, B, D, H, W):
    """does the weight gradient of an upsampled segment go to the split + collapsed kernel (csrc/conv_s3u.hip: k_s3u_bww)?"""
    return S3U and split_engine() and bool(_lib.lib().vxm_conv3d_k3_s3u_bwd_weight_ok(c0, cout, B, D, H, W, s3_pieces()))


def s3u_bwd_weight(ws, x0, c0, bs0, dz, cout, gw, gw_cin, B, D, H, W, lay=0):
    """gw[:, 0:c0] of a [cout][gw_cin][27] array: weight gradient of the x2-upsampled segment x0 [B,c0,D/2,H/2,W/2] against dz [B,cout,D,H,W]"""
    need = _lib.lib().vxm_conv3d_k3_s3u_bwd_weight_workspace_bytes(c0, cout, B, D, H, W)
    buf = ws.get(need)
    args = (ptr(x0), c0, bs0, ptr(dz), cout * D * H * W, cout, ptr(gw), gw_cin, ptr(buf), buf.numel(), B, D, H, W)
    flags = s3_pieces() | lay
    with _prof.region("k_s3u_bww<%d>" % (c0 // 16), flops=2.0 * 8 * c0 * cout * B * D * H * W, nominal=2.0 * 27 * c0 * cout * B * D * H * W):
        call("vxm_conv3d_k3_s3u_bwd_weight", *args, flags | (S3_BW_CONTRACT_ONLY if ws.deferred is not None else 0), stream())
    if ws.deferred is not None:
        ws.deferred.append(("vxm_conv3d_k3_s3u_bwd_weight", args, flags | S3_BW_REDUCE_ONLY, (x0, dz, gw, buf)))


def conv_bwd_weight(ws, x0, c0, bs0, up0, x1, c1, bs1, dz, cout, gw, gb, B, D, H, W, lay=0):
    """lay: S3_IN0_BLOCKED (x0 of a plain conv) / S3_IN1_BLOCKED (dz) for channel-blocked tensors between split kernels (fused U-Net only)"""
    if split_engine() and not up0 and x1 is None and _lib.lib().vxm_conv3d_k3_s3_bwd_weight_ok(c0, cout, B, D, H, W):
        s3_bwd_weight(ws, x0, c0, bs0, dz, cout, gw, c0, 0, gb, B, D, H, W, lay=lay)
        return
    if up0 and x1 is not None and s3u_bwd_weight_route(c0, cout, B, D, H, W) and _lib.lib().vxm_conv3d_k3_s3_bwd_weight_ok(c1, cout, B, D, H, W):
        # cat([upsample(x0), x1]) on the split engine: the upsampled segment through the collapsed split kernel (conv_s3u.hip: k_s3u_bww), the
        # full-resolution skip segment (and the bias gradient) through k_s3_bwd_weight, each into its channel range of gw
        s3u_bwd_weight(ws, x0, c0, bs0, dz, cout, gw, c0 + c1, B, D, H, W, lay=lay & S3_IN1_BLOCKED)
        s3_bwd_weight(ws, x1, c1, bs1, dz, cout, gw, c0 + c1, c0, gb, B, D, H, W, lay=lay & S3_IN1_BLOCKED)
        return
    if FEWCH_H and split_engine() and s3_pieces() == 2 and not up0 and _lib.lib().vxm_conv3d_k3_fewch_bwd_weight_ok(
            ptr(x0), c0, bs0, ptr(x1), c1, bs1, ptr(dz), cout * D * H * W, cout, 2, W):
        if lay & ~S3_IN0_BLOCKED or (lay and not (c0 == 16 and c1 == 0 and cout <= 3)):
            _need_split_kernel(lay, "conv_bwd_weight")
        # first block (2 -> 16) / flow conv (16 -> 3): the few-channel kernel on the fp16 pieces (round 6; the fp32-MFMA form spent 62 % of its
        # time with the matrix pipe busy, and the first block's launch is the last thing a step waits for)
        need = _lib.lib().vxm_conv3d_k3_bwd_weight_workspace_bytes(c0 + c1, cout, B, D, H, W)
        buf = ws.get(need)
        with _prof.region("k_fewch_bwd_weight_h", flops=2.0 * 27 * (c0 + c1) * cout * B * D * H * W):
            call("vxm_conv3d_k3_fewch_bwd_weight", ptr(x0), c0, bs0, ptr(x1), c1, bs1, ptr(dz), cout * D * H * W, cout, ptr(gw), ptr(gb), ptr(buf),
                 buf.numel(), B, D, H, W, 2 | lay, stream())
        return
    _need_split_kernel(lay, "conv_bwd_weight")
    if split_engine() and up0 and x1 is not None and _lib.lib().vxm_conv3d_k3_s3_bwd_weight_ok(c1, cout, B, D, H, W) and \
            _lib.lib().vxm_conv3d_k3_bwd_weight_variant(ptr(x0), c0, bs0, 1, ptr(x1), c1, bs1, ptr(dz), cout * D * H * W, cout, D, H, W) // 10 == 2:
        # cat([upsample(x0), x1]): the upsampled segment through the collapsed fp32-MFMA kernel, the full-resolution skip segment (and the
        # bias gradient) on the split kernel, each into its channel range of gw
        need = _lib.lib().vxm_conv3d_k3_bwd_weight_workspace_bytes(c0 + c1, cout, B, D, H, W)
        buf = ws.get(need)
        with _prof.region("k_conv3d_k3_bwd_weight_up<%d>" % (1 if cout <= 16 else 2), flops=2.0 * 8 * c0 * cout * B * D * H * W,
                          nominal=2.0 * 27 * c0 * cout * B * D * H * W):
            call("vxm_conv3d_k3_bwd_weight_up_segment", ptr(x0), c0, bs0, ptr(x1), c1, bs1, ptr(dz), cout * D * H * W, cout, ptr(gw), ptr(buf),
                 buf.numel(), B, D, H, W, stream())
        s3_bwd_weight(ws, x1, c1, bs1, dz, cout, gw, c0 + c1, c0, gb, B, D, H, W)
        return
    need = _lib.lib().vxm_conv3d_k3_bwd_weight_workspace_bytes(c0 + c1, cout, B, D, H, W)
    buf = ws.get(need)
    name, nominal = None, 2.0 * 27 * (c0 + c1) * cout * B * D * H * W
    flops = nominal
    if _prof.ACTIVE is not None:
        v = _lib.lib().vxm_conv3d_k3_bwd_weight_variant(ptr(x0), c0, bs0, 1 if up0 else 0, ptr(x1), c1, bs1, ptr(dz), cout * D * H * W, cout, D, H, W)
        kind = {0: "dma", 1: "vec", 2: "up+vec", 3: "fewch"}[v // 10]
        name = "k_fewch_bwd_weight" if v // 10 == 3 else "k_conv3d_k3_bwd_weight_%s<%d>" % (kind, v % 10)
        if v // 10 == 2:
            flops = 2.0 * (8 * c0 + 27 * c1) * cout * B * D * H * W
    with _prof.region(name, flops=flops, nominal=nominal):
        call("vxm_conv3d_k3_bwd_weight", ptr(x0), c0, bs0, 1 if up0 else 0, ptr(x1), c1, bs1, ptr(dz), cout * D * H * W,
             cout, ptr(gw), ptr(gb), ptr(buf), buf.numel(), B, D, H, W, stream())


def _factors3(k, nd):
    """per-axis factors (kd, kh, kw) of an int or per-axis pooling factor for a 2-D image (depth 1) or a 3-D volume"""
    ks = [int(v) for v in k] if isinstance(k, (tuple, list)) else [int(k)] * nd
    if len(ks) != nd or any(v < 1 for v in ks):
        raise ValueError("pooling factor %r does not fit %d spatial axes" % (k, nd))
    return [1] * (3 - nd) + ks


class MaxPoolKFn(torch.autograd.Function):
    """MaxPoolNd(k): kernel = stride = k, no padding, floor (voxelmorph/torch/networks.py:83-84,130 with max_pool != 2; images and volumes)."""

    @staticmethod
    def forward(ctx, x, k):
        require_device(x)
        x = _c(x)
        nd = x.dim() - 2
        kd, kh, kw = _factors3(k, nd)
        D, H, W = [1] * (3 - nd) + list(x.shape[2:])
        if D < kd or H < kh or W < kw:
            raise RuntimeError("max_pool: window %s does not fit input %s" % ((kd, kh, kw)[3 - nd:], tuple(x.shape[2:])))
        y = torch.empty(tuple(x.shape[:2]) + tuple(n // f for n, f in zip((D, H, W), (kd, kh, kw)))[3 - nd:], dtype=x.dtype, device=x.device)
        call("vxm_maxpool3d_k_fwd", ptr(x), ptr(y), x.shape[0] * x.shape[1], D, H, W, kd, kh, kw, stream())
        ctx.save_for_backward(x)
        ctx.geom = (D, H, W, kd, kh, kw)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        D, H, W, kd, kh, kw = ctx.geom
        gx = torch.empty_like(x)
        call("vxm_maxpool3d_k_bwd", ptr(x), ptr(_c(gy)), ptr(gx), x.shape[0] * x.shape[1], D, H, W, kd, kh, kw, stream())
        return gx, None


class UpsampleCatKFn(torch.autograd.Function):
    """cat([Upsample(scale_factor=k, 'nearest')(x), skip], 1) (voxelmorph/torch/networks.py:85,137-138 with max_pool != 2); a skip whose
    extents differ from the upsampled ones is refused as torch.cat refuses it in the reference."""

    @staticmethod
    def forward(ctx, x, skip, k):
        require_device(x, skip)
        x, skip = _c(x), _c(skip)
        nd = x.dim() - 2
        kd, kh, kw = _factors3(k, nd)
        up = tuple(n * f for n, f in zip(x.shape[2:], (kd, kh, kw)[3 - nd:]))
        if tuple(skip.shape[2:]) != up or skip.shape[0] != x.shape[0]:
            raise RuntimeError("Sizes of tensors must match except in dimension 1: upsampled %s, skip connection %s (Unet max_pool)"
                               % ((x.shape[0],) + up, (skip.shape[0],) + tuple(skip.shape[2:])))
        B, C0, C1 = x.shape[0], x.shape[1], skip.shape[1]
        D, H, W = [1] * (3 - nd) + list(up)
        out = torch.empty((B, C0 + C1) + up, dtype=x.dtype, device=x.device)
        call("vxm_upsample3d_k_cat", ptr(x), C0, ptr(skip), C1, ptr(out), B, D, H, W, kd, kh, kw, stream())
        ctx.geom = (B, C0, C1, D, H, W, kd, kh, kw, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        B, C0, C1, D, H, W, kd, kh, kw, xshape = ctx.geom
        g = _c(g)
        gx = torch.empty(xshape, dtype=g.dtype, device=g.device)
        call("vxm_upsample3d_k_bwd", ptr(g), C0 + C1, C0, ptr(gx), B, D, H, W, kd, kh, kw, stream())
        gskip = g[:, C0:].contiguous() if ctx.needs_input_grad[1] else None
        return gx, gskip, None


class ConvFn(torch.autograd.Function):
    """One ConvBlock / bare conv (voxelmorph/torch/networks.py:299-305, 211): conv3d(k3,p1) + bias +
    LeakyReLU(slope) (slope=1: no activation).  Generic single-tensor form used by the standalone
    modules; the U-Net uses the fused engine below."""

    @staticmethod
    def forward(ctx, x, w, b, slope):
        _vol3(x, "ConvBlock")
        require_device(x, w, b)
        x, w = _c(x), _c(w)
        B, cin, D, H, W = x.shape
        cout = w.shape[0]
        if w.shape[1] != cin or tuple(w.shape[2:]) != (3, 3, 3):
            raise ValueError("conv weight %s does not fit input with %d channels (3x3x3 kernels only)" % (tuple(w.shape), cin))
        V = D * H * W
        y = torch.empty((B, cout, D, H, W), dtype=x.dtype, device=x.device)
        conv_forward(x, cin, cin * V, False, None, 0, 0, w, b, y, cout * V, cout, slope, B, D, H, W)
        ctx.save_for_backward(x, w, y)
        ctx.slope = slope
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        B, cin, D, H, W = x.shape
        cout = w.shape[0]
        V = D * H * W
        gy = _c(gy)
        if ctx.slope != 1.0:
            dz = torch.empty_like(gy)
            call("vxm_lrelu_bwd", ptr(gy), cout * V, ptr(y), cout * V, ptr(dz), cout * V, float(ctx.slope), B, cout, V, stream())
        else:
            dz = gy
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            conv_bwd_data(dz, cout, w, gx, cin, None, 1.0, B, D, H, W)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw = torch.empty_like(w)
            gb = torch.empty(cout, dtype=w.dtype, device=w.device) if ctx.has_bias else None
            conv_bwd_weight(_Workspace(x.device), x, cin, cin * V, False, None, 0, 0, dz, cout, gw, gb, B, D, H, W)
        return gx, gw, gb, None


# --------------------------------------------------------------------------- fused U-Net engine
class UnetPlan:
    """Static op list of `Unet.forward` (voxelmorph/torch/networks.py:122-144) plus optional trailing
    bare convs (the VxmDense flow conv, networks.py:211,257).

    Tensors are numbered; ids 0/1 are the network inputs (virtual concat of source and target,
    networks.py:253).  A conv reads a *virtual tensor* (seg0 id, seg0 upsampled x2?, seg1 id or None),
    which is how `cat([upsample(x), skip])` (networks.py:137-138) is consumed without being built.
    """

    def __init__(self, in_channels, enc_nf, dec_nf, final_nf, nb_levels, nb_conv_per_level=1, half_res=False,
                 extra=()):
        self.ops = []
        self.ch = {}
        self.lvl = {}
        self.convs = []          # (cin, cout, slope) in parameter order
        nid = 0
        for c in in_channels:
            self.ch[nid] = c
            self.lvl[nid] = 0
            nid += 1
        self.n_inputs = nid
        cur = (0, False, 1 if nid == 2 else None)
        hist = []

        def vt_ch(vt):
            return self.ch[vt[0]] + (self.ch[vt[2]] if vt[2] is not None else 0)

        def vt_lvl(vt):
            return self.lvl[vt[0]] - (1 if vt[1] else 0)

        def add_conv(vt, cout, slope):
            nonlocal nid
            dst = nid
            nid += 1
            self.ch[dst] = cout
            self.lvl[dst] = vt_lvl(vt)
            self.ops.append(dict(kind="conv", k=len(self.convs), src=vt, dst=dst, slope=slope))
            self.convs.append((vt_ch(vt), cout, slope))
            return (dst, False, None)

        for level in range(nb_levels - 1):                      # encoder, networks.py:125-130
            for c in range(nb_conv_per_level):
                cur = add_conv(cur, enc_nf[level * nb_conv_per_level + c], 0.2)
            hist.append(cur[0])
            dst = nid
            nid += 1
            self.ch[dst] = self.ch[cur[0]]
            self.lvl[dst] = self.lvl[cur[0]] + 1
            self.ops.append(dict(kind="pool", src=cur[0], dst=dst))
            cur = (dst, False, None)
        for level in range(nb_levels - 1):                      # decoder, networks.py:133-138
            for c in range(nb_conv_per_level):
                cur = add_conv(cur, dec_nf[level * nb_conv_per_level + c], 0.2)
            if not half_res or level < (nb_levels - 2):
                cur = (cur[0], True, hist.pop())
        for nf in final_nf:                                     # remaining, networks.py:141-142
            cur = add_conv(cur, nf, 0.2)
        self.unet_out_channels = vt_ch(cur)
        for cout, slope in extra:
            cur = add_conv(cur, cout, slope)
        if cur[1] or cur[2] is not None:                        # network ends on a concat: materialise it
            dst = nid
            nid += 1
            self.ch[dst] = vt_ch(cur)
            self.lvl[dst] = vt_lvl(cur)
            self.ops.append(dict(kind="cat", src=cur, dst=dst))
            cur = (dst, False, None)
        self.out = cur[0]
        self.n_tensors = nid
        # consumers of every tensor (to decide where gradients come from in backward)
        self.consumers = {i: [] for i in range(nid)}
        for n, op in enumerate(self.ops):
            if op["kind"] == "pool":
                self.consumers[op["src"]].append(n)
            else:
                for sid in (op["src"][0], op["src"][2]):
                    if sid is not None:
                        self.consumers[sid].append(n)
        self.producer = {op["dst"]: n for n, op in enumerate(self.ops)}


def _dims(shape3, lvl):
    return tuple(s >> lvl for s in shape3)


def _bwd_bounds(cin):
    if cin > 32 and cin % 32 == 16 and SPLIT_48:
        return list(range(0, cin - 16, 32)) + [cin - 16, cin]
    return [0, cin]


def _s3_jobs(plan, params, B, shape3, with_backward, input_grads):
    """The split-kernel operators one pass over `plan` will ask for (forward operator of every eligible conv; for a training step
    the adjoints `conv_bwd_data` launches), so that they are packed in ONE launch.  A job listed here and not used costs a few
    microseconds; one not listed is packed on demand."""
    jobs = []
    for op in plan.ops:
        if op["kind"] != "conv":
            continue
        s0, up0, s1 = op["src"]
        w = params[2 * op["k"]]
        cout, c0 = plan.ch[op["dst"]], plan.ch[s0]
        c1 = plan.ch[s1] if s1 is not None else 0
        if s0 < plan.n_inputs:
            c0, c1 = sum(plan.ch[i] for i in range(plan.n_inputs)), 0 if plan.n_inputs == 1 else plan.ch[1]
            c0 -= c1
        D, H, W = _dims(shape3, plan.lvl[op["dst"]])
        if cout > 4 and not (up0 and s3u_route(c0, c1, cout, B, D, H, W)) and s3_route(c0, up0, c1, cout, B, D, H, W):
            jobs.append((w, 0, c0 + c1, False, c0))
        if not with_backward or (s0 < plan.n_inputs and not input_grads):
            continue
        if up0:
            # UnetFn.backward: the upsampled segment goes straight onto the low-resolution tensor when one of the two fused kernels takes the
            # shape (then only the skip segment needs an adjoint here); otherwise the adjoint of the WHOLE virtual concat runs at this level
            # (round 5: that pack was missing from the list, and was built on demand in the middle of the backward-data chain -- 0.23 ms behind
            # the persistent blocks of the weight gradients, every step, at the 40x48x56 level of the default U-Net)
            V = D * H * W
            up_fused = plan.ops[plan.producer[s0]]["kind"] == "conv" and len(plan.consumers[s0]) == 1 if s0 in plan.producer else False
            low = up_fused and (s3u_bwd_low_route(c0, cout, B, D, H, W) or
                                bool(_lib.lib().vxm_conv3d_k3_up_bwd_low_ok(256, cout * V, c0, cout, B, D, H, W)))     # (256: any 16-byte aligned address)
            ranges = ([(c0, c1)] if s1 is not None else []) if low else [(0, c0 + c1)]
            if low and s1 is not None and up_fused and s3u_bwd_data_route(c0, c1, cout, B, D, H, W):
                ranges = []                                         # k_s3u_bwd_pc computes the skip segment's gradient too (its operator: _prepack_plan, group B2)
        else:
            ranges = [(0, c0 + c1)]                                 # (first channel, count) handed to conv_bwd_data
        for w_lo, cin in ranges:
            bounds = _bwd_bounds(cin)
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                if s3_route(cout, False, 0, hi - lo, B, D, H, W):
                    jobs.append((w, w_lo + lo, w_lo + hi, True, cout))
    return jobs


def _prepack_plan(plan, params, B, shape3, with_backward, input_grads, dry=False, group=None):
    """Every packed operator one pass over `plan` will ask for, built now, in three groups: "A" -- the split operators, in one batched launch
    (_s3_jobs: weight scales + pre-split pieces, ~40 us); "B1" -- the fp32-MFMA operators of the forward convs the split engine does not take
    (coarse levels: ~6 launches of 5 us); "B2" -- the collapsed operators of the cat([upsample, skip]) layers (forward: k_s3u_conv; backward-data
    onto the low-resolution tensor: k_s3u_dlow) and the fp32-MFMA adjoints of the backward (~20 launches, ~150 us in a row).  group=None builds
    all.  Returns the index of the first op that reads an operator of each group (A, B1, B2) -- len(ops) when none does; dry: only the indices.
    Round 6: the groups have their own events.  In the default U-Net the first reader of A is the second layer (op 2), of B1 the first
    coarse-level conv (op 6), of B2 the first cat([upsample, skip]) layer (op 11): with ONE event the main chain sat idle for 0.23 ms per
    replayed step behind the first layer's pooling, waiting for all 27 launches (rocprofv3 dispatch list, profiles/r06q_dispatch_graph.txt)."""
    do_a, do_b1, do_b = (not dry) and group in (None, "A"), (not dry) and group in (None, "B1"), (not dry) and group in (None, "B2")
    if do_a:
        s3_prepack(_s3_jobs(plan, params, B, shape3, with_backward, input_grads))
    first_a = first_b1 = first_b = len(plan.ops)
    for n, op in enumerate(plan.ops):
        if op["kind"] != "conv":
            continue
        s0, up0, s1 = op["src"]
        w = params[2 * op["k"]]
        cout, c0 = plan.ch[op["dst"]], plan.ch[s0]
        c1 = plan.ch[s1] if s1 is not None else 0
        if s0 < plan.n_inputs:
            c0, c1 = sum(plan.ch[i] for i in range(plan.n_inputs)), 0 if plan.n_inputs == 1 else plan.ch[1]
            c0 -= c1
        D, H, W = _dims(shape3, plan.lvl[op["dst"]])
        uses = cout > 4 and s3_route(c0, up0, c1, cout, B, D, H, W)
        if up0 and cout > 4 and s3u_route(c0, c1, cout, B, D, H, W):
            if do_b:
                s3u_pack(w, c0, c1)
            first_b = min(first_b, n)
        elif uses:
            first_a = min(first_a, n)
        if do_b and up0 and with_backward and s3u_bwd_low_route(c0, cout, B, D, H, W) and plan.ops[plan.producer[s0]]["kind"] == "conv" \
                and len(plan.consumers[s0]) == 1:
            s3u_bwd_low_pack(w, c0, c0 + c1)
            if s1 is not None and s3u_bwd_data_route(c0, c1, cout, B, D, H, W):
                s3u_bwd_skip_pack(w, c0, c1)
        # the fp32-MFMA operators of the convs the split engine does not take (coarse levels, first layer, flow head): packed here too, so
        # that the main chain does not carry a 5 us pack launch in front of each of them.  A guess that turns out unused costs that launch on
        # the second stream; one that is missing is packed on demand, as before.
        if n > 0 and cout > 4 and not uses and not up0:      # (op 0 needs its operator at once: packed on demand on the main stream)
            if do_b1:
                pack_weights_cached(w, False)
            first_b1 = min(first_b1, n)                       # the main stream must have joined the second one before this launch
        if do_b and with_backward and s0 >= plan.n_inputs:
            whole = not up0
            for w_lo, cin_r in ([(0, c0 + c1)] if whole else ([(c0, c1)] if s1 is not None else [])):
                bounds = _bwd_bounds(cin_r)
                for lo, hi in zip(bounds[:-1], bounds[1:]):
                    if not s3_route(cout, False, 0, hi - lo, B, D, H, W):
                        pack_weights_cached(w, True, w_lo + lo, w_lo + hi)
    return first_a, first_b1, first_b


def _blocked_tensors(plan, B, shape3):
    """Which activations of the fused U-Net are kept CHANNEL-BLOCKED ([B][C/8][D][H][W][8], include/vxm_hip.h VXM_S3_*_BLOCKED) instead of
    NCDHW.  The tensors never leave the engine, so their layout is its own business; a haloed row of a blocked tensor is one contiguous run
    and the split kernels, which are bound by the vector L1's sector requests, stage it 15 - 20 % faster.  A tensor t
    qualifies when every kernel that writes or reads t, or the gradient DZ[t] (which takes the same layout), is a split kernel with a
    blocked variant at this shape: t is the output of a ConvBlock, read by exactly one plain ConvBlock (forward, weight gradient, and the
    fused backward-data that produces DZ[t] with t as its mask), and its producer's backward (weight gradient and backward-data, which read
    DZ[t]) runs on the split kernels too.  In the default VxmDense U-Net: the outputs of remaining[0] and remaining[1]."""
    if not (BLOCKED and split_engine() and s3_pieces() == 2):
        return frozenset()
    key = (B, tuple(shape3), FP32_ENGINE, S3U, S3_UP, SPLIT_48, BLOCKED_LAST, FEWCH_H)          # (the route predicates below depend on these switches)
    cache = plan.__dict__.setdefault("_blocked_cache", {})
    if key in cache:
        return cache[key]
    L = _lib.lib()
    out = set()

    def plain_ok(cin, cout, D, H, W):
        """forward-type launch cin -> cout of a plain single-segment tensor on an 8-row split instance"""
        return cout > 4 and s3_route(cin, False, 0, cout, B, D, H, W) and bool(L.vxm_conv3d_k3_s3_layout_ok(cin, 0, 0, cout, H, 2))

    for t, pn in plan.producer.items():
        prod = plan.ops[pn]
        cons = plan.consumers[t]
        if prod["kind"] != "conv" or t == plan.out or len(cons) != 1:
            continue
        cop = plan.ops[cons[0]]
        if cop["kind"] != "conv" or tuple(cop["src"]) != (t, False, None):
            continue
        C, cc = plan.ch[t], plan.ch[cop["dst"]]
        D, H, W = _dims(shape3, plan.lvl[t])
        if C % 16 or _bwd_bounds(C) != [0, C]:
            continue
        # the consumer: forward (reads t), weight gradient (x = t), fused backward-data (mask t, writes DZ[t])
        if cc <= 4:
            # a few-output-channel conv (the flow conv, networks.py:211,257): k_conv3d_k3_fewout / k_fewch_bwd_weight_h / k_conv3d_k3_kpack take the
            # layout flags (round 6, late); the *_ok predicates only look at alignment, which every torch allocation has (a stand-in address)
            V = D * H * W
            if not (BLOCKED_LAST and C == 16 and FEWCH_H and L.vxm_conv3d_k3_fewout_ok(256, C * V, 256, cc * V, C, cc, W)
                    and L.vxm_conv3d_k3_fewch_bwd_weight_ok(256, C, C * V, None, 0, 0, 256, cc * V, cc, 2, W)
                    and L.vxm_conv3d_k3_fwd_layout_ok(256, cc, cc * V, None, 0, 0, 256, C, B, D, H, W)):
                continue
        elif not (plain_ok(C, cc, D, H, W) and L.vxm_conv3d_k3_s3_bwd_weight_ok(C, cc, B, D, H, W) and plain_ok(cc, C, D, H, W)):
            continue
        # the producer: forward (writes t), weight gradient (dz = DZ[t]), backward-data (reads DZ[t])
        s0, up0, s1 = prod["src"]
        if s0 < plan.n_inputs:
            continue
        c0 = plan.ch[s0]
        c1 = plan.ch[s1] if s1 is not None else 0
        if up0:
            up_fused = plan.ops[plan.producer[s0]]["kind"] == "conv" and len(plan.consumers[s0]) == 1
            if not (s1 is not None and up_fused and s3u_route(c0, c1, C, B, D, H, W) and s3u_bwd_low_route(c0, C, B, D, H, W)
                    and s3u_bwd_weight_route(c0, C, B, D, H, W) and L.vxm_conv3d_k3_s3_bwd_weight_ok(c1, C, B, D, H, W)
                    and _bwd_bounds(c1) == [0, c1] and plain_ok(C, c1, D, H, W)):
                continue
        else:
            if s1 is not None or _bwd_bounds(c0) != [0, c0]:
                continue
            if not (plain_ok(c0, C, D, H, W) and L.vxm_conv3d_k3_s3_bwd_weight_ok(c0, C, B, D, H, W) and plain_ok(C, c0, D, H, W)):
                continue
        out.add(t)
    cache[key] = frozenset(out)
    return cache[key]


class UnetFn(torch.autograd.Function):
    """Whole U-Net (+ trailing convs) forward/backward on the HIP kernels: 12 MFMA conv launches,
    4 pool launches forward; backward = per conv one bwd-data launch (the forward kernel with the
    adjoint weights and the previous block's LeakyReLU' fused in the epilogue) + one bwd-weight
    launch, plus the fused pool / upsample gradient kernels.  No ATen op is issued."""

    @staticmethod
    def forward(ctx, plan, *tensors):
        inputs = [_c(t) for t in tensors[:plan.n_inputs]]
        params = tensors[plan.n_inputs:]
        require_device(*inputs)
        require_device(*params)
        for t in inputs:
            _vol3(t, "Unet")
        B = inputs[0].shape[0]
        shape3 = tuple(inputs[0].shape[2:])
        nlev = max(plan.lvl.values())
        if any(s % (1 << nlev) for s in shape3):
            raise ValueError("Unet: volume %s must be divisible by %d (MaxPool floors and the skip concat "
                             "would not line up, networks.py:130,138)" % (shape3, 1 << nlev))
        dev, dt = inputs[0].device, inputs[0].dtype
        T = {i: t for i, t in enumerate(inputs)}
        for i, t in T.items():
            if t.shape[1] != plan.ch[i] or t.shape[0] != B or tuple(t.shape[2:]) != shape3:
                raise ValueError("Unet: input %d has shape %s, expected [%d,%d,%s]" % (i, tuple(t.shape), B, plan.ch[i], shape3))
        ready = [None, None, None]               # events: the operators of group A / B1 / B2 are built (second stream, in that order)
        first = [len(plan.ops)] * 3
        packs_late = False
        want_bwd = any(ctx.needs_input_grad[1:])
        if split_engine():
            # The packed operators (weight scales, pre-split pieces, collapsed upsample operators: ~27 small launches, 0.2 ms in a row) are
            # rebuilt once per optimiser step.  They go to the second stream and run beside the layers that do not read them; the main stream
            # waits for each GROUP (_prepack_plan) in front of the first launch that reads it.
            want_bwd, want_in = any(ctx.needs_input_grad[1:]), any(ctx.needs_input_grad[1:1 + plan.n_inputs])
            if OVERLAP_SMALL_LEVELS:
                main, side = torch.cuda.current_stream(dev), _side_stream(dev)
                fork = torch.cuda.Event()
                fork.record(main)

                def launch_packs(groups=("A", "B1", "B2")):
                    side.wait_event(fork)
                    evs = [None] * (3 - len(groups))
                    with torch.cuda.stream(side):
                        for grp in groups:
                            _prepack_plan(plan, params, B, shape3, want_bwd, want_in, group=grp)
                            evs.append(torch.cuda.Event())
                            evs[-1].record(side)
                    _adopt_fresh_packs(main)
                    return evs
                first = list(_prepack_plan(plan, params, B, shape3, want_bwd, want_in, dry=True))
                # While a hipGraph is being captured the packs are enqueued BEHIND the layers that do not read them (a graph runs by its edges,
                # not by enqueue order, and the first child created of a node keeps its queue: the main chain should, see DW_ORDER); launch by
                # launch they must be enqueued first to run beside those layers.
                packs_late = torch.cuda.is_current_stream_capturing() and min(first) > 0
                if not packs_late:
                    ready = launch_packs()
                else:
                    # Group A (two launches, ~35 us) on the MAIN stream in front of the first layer: in a replayed graph the second branch does
                    # not start before the first layer's pooling has run, whatever the creation order of its nodes (round 6, rocprofv3 dispatch
                    # lists r06r / r06s), and the second layer then waited 82 us for these two launches
                    _prepack_plan(plan, params, B, shape3, want_bwd, want_in, group="A")
            else:
                packs_late = False
                _prepack_plan(plan, params, B, shape3, want_bwd, want_in)
                _adopt_fresh_packs(None)
        blocked = _blocked_tensors(plan, B, shape3)
        SG = {}                      # tensor id -> sign tensor of a channel-blocked activation (S3_OUT_SIGNS / S3_MASK_SIGNS)
        PC = {}                      # tensor id -> 16-bit codes of its pooling (vxm_maxpool2_fwd_code), for the fused pooling backward + weight gradient
        walked_back = set()          # tensors whose producer walked its tiles from the end (S3_REVERSE_TILES)
        for n_op, op in enumerate(plan.ops):
            if packs_late and n_op >= 1:
                # (captured right behind the FIRST op: on this runtime a branch of a replayed graph starts once everything the main chain had
                # enqueued before the branch's nodes were created has run -- created behind op 0 and its pooling, the packs began when the
                # pooling ended and the second layer waited 47 us for group A; created between the two they run beside the pooling)
                ready = launch_packs(("B1", "B2"))
                packs_late = False
            for g in (2, 1, 0):                      # a later group's event is recorded behind the earlier ones on the same stream
                if ready[g] is not None and n_op >= first[g]:
                    torch.cuda.current_stream(dev).wait_event(ready[g])
                    for e in range(g + 1):
                        ready[e] = None
            dst = op["dst"]
            D, H, W = _dims(shape3, plan.lvl[dst])
            V = D * H * W
            out = torch.empty((B, plan.ch[dst], D, H, W), dtype=dt, device=dev)
            if op["kind"] == "conv":
                s0, up0, s1 = op["src"]
                w, b = params[2 * op["k"]], params[2 * op["k"] + 1]
                x0, x1 = T[s0], (T[s1] if s1 is not None else None)
                lay = (S3_IN0_BLOCKED if s0 in blocked else 0) | (S3_OUT_BLOCKED if dst in blocked else 0)
                if SNAKE and not up0 and s1 is None and s0 >= plan.n_inputs and plan.lvl[dst] == 0 and plan.ch[dst] > 4 and s0 not in walked_back \
                        and s3_route(plan.ch[s0], False, 0, plan.ch[dst], B, D, H, W):
                    lay |= S3_REVERSE_TILES
                    walked_back.add(dst)
                sg = None
                if SIGNS and want_bwd and dst in blocked and op["slope"] != 1.0 and plan.ch[dst] % 8 == 0 and s3_pieces() == 2:
                    # the one reader of this activation's sign is the backward-data epilogue of its consumer (a blocked split launch, or it
                    # would not be in `blocked`): one byte per four channels and voxel instead of the fp32 tensor there
                    sg = SG[dst] = torch.empty((B, plan.ch[dst] // 4, D, H, W), dtype=torch.uint8, device=dev)
                conv_forward(x0, plan.ch[s0], x0[0].numel(), up0, x1, plan.ch[s1] if s1 is not None else 0,
                             x1[0].numel() if x1 is not None else 0, w, b, out, plan.ch[dst] * V, plan.ch[dst], op["slope"], B, D, H, W, lay=lay, signs=sg)
            elif op["kind"] == "pool":
                src = T[op["src"]]
                sD, sH, sW = src.shape[2:]
                if want_bwd and _pool_fusable(plan, op["src"], shape3, any(ctx.needs_input_grad[1:1 + plan.n_inputs]), blocked):
                    PC[op["src"]] = torch.empty((B, plan.ch[dst], sD // 2, sH // 2, sW // 2), dtype=torch.int16, device=dev)
                    call("vxm_maxpool2_fwd_code", ptr(src), src[0].numel(), ptr(out), ptr(PC[op["src"]]), B, plan.ch[dst], sD, sH, sW, stream())
                else:
                    call("vxm_maxpool2_fwd", ptr(src), src[0].numel(), ptr(out), B, plan.ch[dst], sD, sH, sW, stream())
            else:
                s0, up0, s1 = op["src"]
                call("vxm_upsample2_cat", ptr(T[s0]), plan.ch[s0], ptr(T[s1]), plan.ch[s1], ptr(out), B, D, H, W, stream())
            T[dst] = out
            if _RANGE_PROBE is not None and op["kind"] == "conv":
                _probe("activation of conv %d (%d channels, level %d)" % (op["k"], plan.ch[dst], plan.lvl[dst]), out, plan.ch[dst], dst in blocked)
        # The returned tensor must not be reachable from ctx except through save_for_backward: out.grad_fn is this
        # node, so `ctx.T[plan.out] = out` would be a reference cycle that keeps EVERY activation of the step alive
        # until Python's cyclic GC runs (tens of GB per step at 160x192x224).
        if packs_late:
            ready = launch_packs(("B1", "B2"))
        if ready[2] is not None:                # no op of this plan read a group-B2 operator (the backward's are in it): still join the second stream
            torch.cuda.current_stream(dev).wait_event(ready[2])
        out = T.pop(plan.out)
        ctx.save_for_backward(out)
        ctx.plan, ctx.T, ctx.params, ctx.shape3, ctx.B, ctx.blocked, ctx.SG, ctx.PC = plan, T, params, shape3, B, blocked, SG, PC
        # activations and parameters are held as plain attributes (the returned tensor alone goes through
        # save_for_backward, see above), so autograd's version-counter check is done by hand in backward
        # (inference tensors -- torch.inference_mode() -- carry no version counter and can never reach backward)
        ctx.versions = _versions(list(params) + list(T.values()))
        return out

    @staticmethod
    def backward(ctx, gout):
        plan, params, shape3, B, blocked, SG, PC = ctx.plan, ctx.params, ctx.shape3, ctx.B, ctx.blocked, ctx.SG, ctx.PC
        if ctx.T is None:
            raise RuntimeError("UnetFn: backward a second time: the saved activations were released by the first pass "
                               "(a retained graph is not supported by the fused engine)")
        T = dict(ctx.T)
        if _versions(list(params) + list(T.values())) != ctx.versions:
            raise RuntimeError("UnetFn: a parameter or a saved activation was modified in place between forward and "
                               "backward (e.g. an optimizer step before loss.backward()); gradients would be wrong")
        ctx.T = None                # released with this pass, not when the graph node dies
        T[plan.out] = ctx.saved_tensors[0]
        dev, dt = gout.device, gout.dtype
        gout = _c(gout)
        ws = _Workspace(dev)
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if OVERLAP_SMALL_LEVELS else None
        # The weight gradients run on a second stream beside the backward-data chain.  Their small reduction kernels (partials -> gw, gb) go to
        # a THIRD stream: behind a contraction on the second stream, the few blocks of a reduction waited for the persistent blocks of the main
        # stream's kernel to leave the chip -- and the next contraction waited with them (rocprofv3 trace, round 5: 450 - 585 us, three times per
        # step; the second stream ended the step 0.45 ms after the main one).
        # (same-box A/B after the change: 84.6 -> 83.2 pairs/s eager, 82.5 -> 81.3 replayed.  The contractions then share the chip with the main
        # chain ALL the time, both slow down, and the main chain ends 0.38 ms later: the chip is saturated either way, the stalls were not idle
        # capacity.  Kept behind VXM_BW_REDUCE_STREAM=1 with its measurement; default: reductions behind their contraction, as before.)
        red = _side_stream(dev, 1) if (side is not None and BW_REDUCE_STREAM) else None
        pending = []
        ws_side = _Workspace(dev, deferred=pending if red is not None else None)
        n_in = plan.n_inputs
        grads = [None] * (n_in + len(params))
        dz_back = set()   # tensor ids whose gradient DZ[.] was written from the end (S3_REVERSE_TILES)
        DZ = {}      # tensor id -> gradient w.r.t. the pre-activation of its producing conv
        GP = {}      # pool-output id -> gradient w.r.t. the pooled tensor
        GS = {}      # tensor id -> (buffer, element offset, batch stride): skip-branch gradient view
        GC = {}      # decoder tensor id -> (buffer, batch stride): gradient of its upsampled copy

        def finish_conv_output(tid, graw, graw_bs):
            """graw = dL/d(activation output) of tensor tid -> DZ[tid] (leaky_relu_backward)."""
            op = plan.ops[plan.producer[tid]]
            D, H, W = _dims(shape3, plan.lvl[tid])
            V, C = D * H * W, plan.ch[tid]
            if op["slope"] == 1.0 and graw_bs == C * V:
                DZ[tid] = graw
                return
            dz = torch.empty((B, C, D, H, W), dtype=dt, device=dev)
            call("vxm_lrelu_bwd", ptr(graw), graw_bs, ptr(T[tid]), C * V, ptr(dz), C * V, float(op["slope"]), B, C, V, stream())
            DZ[tid] = dz

        # gradient of the network output
        out_op = plan.ops[plan.producer[plan.out]]
        if out_op["kind"] == "conv":
            finish_conv_output(plan.out, gout, gout[0].numel())
        else:
            GCAT = gout

        try:
            deferred_dw = None
            for n in range(len(plan.ops) - 1, -1, -1):
                if deferred_dw is not None:
                    deferred_dw()
                    deferred_dw = None
                op = plan.ops[n]
                dst = op["dst"]
                D, H, W = _dims(shape3, plan.lvl[dst])
                V = D * H * W
                if op["kind"] == "cat":
                    s0, _, s1 = op["src"]
                    c0 = plan.ch[s0]
                    GC[s0] = (GCAT, GCAT[0].numel())
                    GS[s1] = (GCAT, c0 * V, GCAT[0].numel())
                    _resolve_decoder(plan, T, DZ, GC, s0, B, shape3, dt, dev)
                    continue
                if op["kind"] == "pool":
                    src = op["src"]
                    sD, sH, sW = _dims(shape3, plan.lvl[src])
                    C = plan.ch[src]
                    gs = GS.get(src)
                    prod = plan.ops[plan.producer[src]] if src in plan.producer else None
                    slope = prod["slope"] if prod is not None and prod["kind"] == "conv" else 1.0
                    gskip = None
                    gs_bs = 0
                    if gs is not None:
                        gskip = gs[0].view(-1)[gs[1]:]
                        gs_bs = gs[2]
                    if src in PC and gskip is not None:
                        # the one reader of this gradient is the first block's weight gradient, which forms it on the fly (conv op below)
                        DZ[src] = _PoolGrad(gskip, gs_bs, GP[dst], PC[src], float(slope), gs[0])
                        continue
                    dz = torch.empty((B, C, sD, sH, sW), dtype=dt, device=dev)
                    call("vxm_maxpool2_bwd", ptr(T[src]), T[src][0].numel(), ptr(GP[dst]), ptr(gskip), gs_bs, ptr(dz),
                         float(slope), B, C, sD, sH, sW, stream())
                    DZ[src] = dz
                    continue
                # ---- conv
                s0, up0, s1 = op["src"]
                w, b = params[2 * op["k"]], params[2 * op["k"] + 1]
                cout = plan.ch[dst]
                c0 = plan.ch[s0]
                c1 = plan.ch[s1] if s1 is not None else 0
                cin = c0 + c1
                dz = DZ.pop(dst)
                x0, x1 = T[s0], (T[s1] if s1 is not None else None)
                if isinstance(dz, _PoolGrad):
                    # first block, nothing behind it: weight / bias gradient from the pooling backward's operands in one launch (main stream: the
                    # last launch of the pass, see last_on_main below)
                    gw_sink, gb_sink = _claim_sink(w), _claim_sink(b)
                    gw = gw_sink if gw_sink is not None else torch.empty_like(w)
                    gb = gb_sink if gb_sink is not None else torch.empty_like(b)
                    need = _lib.lib().vxm_conv3d_k3_bwd_weight_workspace_bytes(cin, cout, B, D, H, W)
                    buf = ws.get(need)
                    with _prof.region("k_fewch_bwd_weight_h+pool", flops=2.0 * 27 * cin * cout * B * V):
                        call("vxm_conv3d_k3_fewch_bwd_weight_pool", ptr(x0), c0, x0[0].numel(), ptr(x1), c1, x1[0].numel() if x1 is not None else 0,
                             ptr(dz.gskip), dz.gs_bs, ptr(dz.gpool), ptr(dz.code), dz.slope, ptr(gw), ptr(gb), ptr(buf), buf.numel(), B, D, H, W, 2, stream())
                    grads[n_in + 2 * op["k"]] = None if gw_sink is not None else gw
                    grads[n_in + 2 * op["k"] + 1] = None if gb_sink is not None else gb
                    continue
                if _RANGE_PROBE is not None:
                    _probe("gradient at conv %d (%d channels, level %d)" % (op["k"], cout, plan.lvl[dst]), dz, cout, dst in blocked)
                # channel-blocked tensors (_blocked_tensors): the activation s0 of a plain conv and / or this conv's own output, whose DZ shares its layout
                dz_blk, x_blk = dst in blocked, s0 in blocked
                lay_w = (S3_IN0_BLOCKED if x_blk else 0) | (S3_IN1_BLOCKED if dz_blk else 0)
                lay_d = S3_IN0_BLOCKED if dz_blk else 0
                # parameter gradients go straight into the optimiser's flat bucket when one is attached
                # (voxelmorph_amd.optim.FlatAdam): no per-tensor accumulate / copy launches
                # The kernel OVERWRITES its destination: only the first gradient of a parameter since zero_grad() may land in
                # the bucket; later ones (a second backward / a model called twice before one backward) go through autograd's
                # accumulation into p.grad, which FlatAdam.step() folds into the bucket.
                gw_sink, gb_sink = _claim_sink(w), _claim_sink(b)
                gw = gw_sink if gw_sink is not None else torch.empty_like(w)
                gb = gb_sink if gb_sink is not None else torch.empty_like(b)
                # the FIRST layer's weight gradient is the last launch of the pass (its dz is the last thing the backward-data chain produces, and
                # when the inputs need no gradient nothing follows it): on the main stream it starts the moment that dz exists, beside whatever
                # the second stream still has queued, instead of behind it (rocprofv3 trace, round 5: the step ended 0.38 ms after the main chain)
                last_on_main = s0 < n_in and not any(ctx.needs_input_grad[1 + i] for i in range(n_in))
                if side is not None and plan.lvl[dst] >= OVERLAP_MIN_LEVEL and not last_on_main:
                    # Below full resolution neither product fills the chip (a few hundred tiles on 256 CUs): the weight gradient
                    # of this block runs on a second stream beside the backward-data chain it does not feed.
                    ev = torch.cuda.Event()
                    ev.record(main)                 # dz is final here: the weight gradient may start, whenever it is enqueued

                    def launch_dw(ev=ev, x0=x0, c0=c0, up0=up0, x1=x1, c1=c1, dz=dz, cout=cout, gw=gw, gb=gb, gw_sink=gw_sink, gb_sink=gb_sink,
                                  D=D, H=H, W=W, lay_w=lay_w):
                        side.wait_event(ev)
                        with torch.cuda.stream(side):
                            conv_bwd_weight(ws_side, x0, c0, x0[0].numel(), up0, x1, c1, x1[0].numel() if x1 is not None else 0, dz, cout,
                                            gw, gb, B, D, H, W, lay=lay_w)
                            if pending:
                                ev2 = torch.cuda.Event()
                                ev2.record(side)
                        dz.record_stream(side)          # dz is released by the main-stream chain before the side stream may be done
                        for g, sink in ((gw, gw_sink), (gb, gb_sink)):
                            if sink is None:            # allocated on the main stream, written on the side stream
                                g.record_stream(side)
                        if pending:                     # the reductions of the contractions just launched: third stream, behind them
                            red.wait_event(ev2)
                            with torch.cuda.stream(red):
                                for name, args, flags, keep in pending:
                                    call(name, *args, flags, stream())
                            for _, _, _, keep in pending:
                                for t in keep:
                                    if t is not None:
                                        t.record_stream(red)
                            pending.clear()
                    if DW_ORDER == "after" or (DW_ORDER != "before" and torch.cuda.is_current_stream_capturing()):
                        deferred_dw = launch_dw     # enqueued behind this op's backward-data launches (top of the next iteration)
                    else:
                        launch_dw()
                else:
                    conv_bwd_weight(ws, x0, c0, x0[0].numel(), up0, x1, c1, x1[0].numel() if x1 is not None else 0, dz, cout,
                                    gw, gb, B, D, H, W, lay=lay_w)
                grads[n_in + 2 * op["k"]] = None if gw_sink is not None else gw
                grads[n_in + 2 * op["k"] + 1] = None if gb_sink is not None else gb
                feeds_inputs = s0 < n_in
                if feeds_inputs and not any(ctx.needs_input_grad[1 + i] for i in range(n_in)):
                    continue
                fuse = (not up0) and s1 is None and (not feeds_inputs) and len(plan.consumers[s0]) == 1 \
                    and plan.ops[plan.producer[s0]]["kind"] == "conv"
                up_fused = up0 and plan.ops[plan.producer[s0]]["kind"] == "conv" and len(plan.consumers[s0]) == 1
                if up_fused and (s3u_bwd_low_route(c0, cout, B, D, H, W) or
                                 _lib.lib().vxm_conv3d_k3_up_bwd_low_ok(ptr(dz), cout * V, c0, cout, B, D, H, W)):
                    # upsampled segment: straight to the half-resolution gradient of the decoder block (stride-2 4x4x4 conv =
                    # conv backward + upsample_nearest3d_backward + leaky_relu_backward in one kernel: conv_s3u.hip k_s3u_dlow on the split
                    # engine, conv_fwd.hip k_conv3d_k3_dlow otherwise)
                    pslope = plan.ops[plan.producer[s0]]["slope"]
                    lD, lH, lW = D // 2, H // 2, W // 2
                    dzl = torch.empty((B, c0, lD, lH, lW), dtype=dt, device=dev)
                    gxs = None
                    if s1 is not None and s3u_bwd_data_route(c0, c1, cout, B, D, H, W):
                        # both products from one staging of dz (k_s3u_bwd_pc): the skip segment's gradient comes with it
                        gxs = torch.empty((B, c1, D, H, W), dtype=dt, device=dev)
                        s3u_bwd_data(dz, cout, w, c0, c1, dzl, T[s0] if pslope != 1.0 else None, pslope, gxs, B, D, H, W, lay=lay_d)
                    elif s3u_bwd_low_route(c0, cout, B, D, H, W):
                        s3u_bwd_low(dz, cout, w, c0, cin, dzl, T[s0] if pslope != 1.0 else None, pslope, B, D, H, W, lay=lay_d)
                    else:
                        _need_split_kernel(lay_d, "backward-data onto the low-resolution tensor")
                        wpk = torch.empty(_lib.lib().vxm_conv3d_k3_up_bwd_low_packed_elems(c0, cout), dtype=dt, device=dev)
                        with _prof.region("k_conv3d_k3_dlow<%d>" % (1 if c0 <= 16 else 2), flops=2.0 * 8 * c0 * cout * B * V,
                                          nominal=2.0 * 27 * c0 * cout * B * V):
                            call("vxm_conv3d_k3_up_bwd_low", ptr(dz), cout * V, cout, ptr(_c(w)), c0, cin, ptr(wpk), ptr(dzl), c0 * lD * lH * lW,
                                 ptr(T[s0]) if pslope != 1.0 else None, c0 * lD * lH * lW, float(pslope), B, D, H, W, stream())
                    DZ[s0] = dzl
                    if s1 is not None:                       # skip segment: regular backward-data of its channels only
                        if gxs is None:
                            gxs = torch.empty((B, c1, D, H, W), dtype=dt, device=dev)
                            conv_bwd_data(dz, cout, w, gxs, c1, None, 1.0, B, D, H, W, w_lo=c0, lay=lay_d)
                        GS[s1] = (gxs, 0, c1 * V)
                        if not any(plan.ops[m]["kind"] == "pool" for m in plan.consumers[s1]):
                            g = GS.pop(s1)
                            finish_conv_output(s1, g[0].view(-1)[g[1]:], g[2])
                    continue
                gx = torch.empty((B, cin, D, H, W), dtype=dt, device=dev)
                if fuse:   # dX * LeakyReLU'(y_prev) in the epilogue == DZ of the previous ConvBlock
                    pslope = plan.ops[plan.producer[s0]]["slope"]
                    rev = 0
                    if SNAKE and plan.lvl[dst] == 0 and dst not in dz_back and _bwd_bounds(cin) == [0, cin] and s3_route(cout, False, 0, cin, B, D, H, W):
                        rev = S3_REVERSE_TILES          # (dropped by conv_bwd_data when the launch is not a split kernel's)
                        dz_back.add(s0)
                    if x_blk and pslope != 1.0 and s0 in SG and _bwd_bounds(cin) == [0, cin] and (cout <= 4 or s3_route(cout, False, 0, cin, B, D, H, W)):
                        conv_bwd_data(dz, cout, w, gx, cin, SG[s0], pslope, B, D, H, W, lay=lay_d | S3_OUT_BLOCKED | S3_MASK_SIGNS | rev)
                    else:
                        conv_bwd_data(dz, cout, w, gx, cin, T[s0] if pslope != 1.0 else None, pslope, B, D, H, W,
                                      lay=lay_d | (S3_OUT_BLOCKED if x_blk else 0) | rev)
                    DZ[s0] = gx
                    continue
                if x_blk:
                    raise RuntimeError("UnetFn: a channel-blocked activation reached the unfused backward-data path")
                conv_bwd_data(dz, cout, w, gx, cin, None, 1.0, B, D, H, W, lay=lay_d)
                if feeds_inputs:
                    for i, sid in enumerate([s0] + ([s1] if s1 is not None else [])):
                        lo = 0 if i == 0 else c0
                        grads[sid] = gx[:, lo:lo + plan.ch[sid]]
                    continue
                if up0:
                    GC[s0] = (gx, cin * V)
                    _resolve_decoder(plan, T, DZ, GC, s0, B, shape3, dt, dev)
                else:
                    prod = plan.ops[plan.producer[s0]]
                    if prod["kind"] == "pool":
                        GP[s0] = gx if s1 is None else gx[:, :c0].contiguous()
                    else:      # conv output with several consumers (not produced by the reference topologies)
                        finish_conv_output(s0, gx, cin * V)
                if s1 is not None:
                    GS[s1] = (gx, c0 * V, cin * V)
                    if not any(plan.ops[m]["kind"] == "pool" for m in plan.consumers[s1]):
                        g = GS.pop(s1)
                        finish_conv_output(s1, g[0].view(-1)[g[1]:], g[2])
            if deferred_dw is not None:
                deferred_dw()
        finally:
            # also on an exception half-way: later main-stream work (zero_grad, Adam on the bucket) must not race with
            # side-stream launches that are still writing parameter gradients
            if side is not None:
                main.wait_stream(side)          # gradients (and the activations the side stream read) are final past this point
                if red is not None:
                    main.wait_stream(red)
        return (None,) + tuple(grads)


class _PoolGrad:
    """The gradient at the pre-activation of a pooled ConvBlock, not materialised: the operands of vxm_maxpool2_bwd, handed to the one kernel
    that reads it (vxm_conv3d_k3_fewch_bwd_weight_pool)."""
    __slots__ = ("gskip", "gs_bs", "gpool", "code", "slope", "keep")

    def __init__(self, gskip, gs_bs, gpool, code, slope, keep):
        self.gskip, self.gs_bs, self.gpool, self.code, self.slope, self.keep = gskip, gs_bs, gpool, code, slope, keep


def _pool_fusable(plan, src, shape3, inputs_need_grad, blocked):
    """True when the gradient of tensor `src` (pooled AND a skip connection) has exactly one reader that can form it on the fly: the weight
    gradient of the FIRST ConvBlock (its inputs are network inputs that need no gradient) on the few-channel fp16-piece kernel."""
    if not (POOL_FUSE and FEWCH_H and split_engine() and s3_pieces() == 2 and _RANGE_PROBE is None) or inputs_need_grad or src in blocked:
        return False
    if src not in plan.producer:
        return False
    prod = plan.ops[plan.producer[src]]
    if prod["kind"] != "conv":
        return False
    s0, up0, s1 = prod["src"]
    if up0 or s0 >= plan.n_inputs or (s1 is not None and s1 >= plan.n_inputs):
        return False
    cin = plan.ch[s0] + (plan.ch[s1] if s1 is not None else 0)
    D, H, W = _dims(shape3, plan.lvl[src])
    cons = [plan.ops[m] for m in plan.consumers[src]]        # its pooling and the conv that reads it as the skip segment of a virtual concat
    if sorted(o["kind"] for o in cons) != ["conv", "pool"] or [o for o in cons if o["kind"] == "conv"][0]["src"][2] != src:
        return False
    return plan.ch[src] == 16 and cin <= 3 and not ((D | H | W) & 1) and W % 4 == 0


def _resolve_decoder(plan, T, DZ, GC, tid, B, shape3, dt, dev):
    """Upsample(2,'nearest') backward (sum over the 2x2x2 children) fused with LeakyReLU' of the
    decoder ConvBlock that produced tensor `tid`."""
    buf, bs = GC.pop(tid)
    D, H, W = _dims(shape3, plan.lvl[tid])
    C = plan.ch[tid]
    prod = plan.ops[plan.producer[tid]]
    dz = torch.empty((B, C, D, H, W), dtype=dt, device=dev)
    if prod["kind"] == "conv":
        call("vxm_upsample2_bwd", ptr(buf), bs, ptr(T[tid]) if prod["slope"] != 1.0 else None, ptr(dz),
             float(prod["slope"]), B, C, D, H, W, stream())
        DZ[tid] = dz
    else:
        raise NotImplementedError("upsampling a non-conv tensor is not produced by Unet topologies")
