"""2-D (planar) autograd wrappers over the C ABI: the same reference classes (`voxelmorph/torch/layers.py`,
`losses.py`, `networks.py` are N-D generic) on `[B,C,H,W]` images with 2-channel flows.

Layers and losses call the `vxm_*2d_*` kernels (csrc/planar.hip, losses.hip).  The 3x3 convolutions of a 2-D network run on
the 3-D MFMA kernels with a depth of one: a `[Cout,Cin,3,3]` weight is the middle depth slice of a `[Cout,Cin,3,3,3]`
kernel whose other two slices only ever meet zero padding.  2-D slices are tiny next to the 160x192x224 volumes of the
benchmark path, so the 2-D U-Net is composed from per-op Functions instead of the fused 3-D engine.
"""
import math

import torch
import torch.nn.functional as F

from .. import profiler as _prof
from .._lib import call, ptr, require_device, stream
from .functional import INTERP, PENALTY, ConvFn, _c


def _img(t, what):
    if t.dim() != 4:
        raise ValueError("%s: expected a [B,C,H,W] tensor, got %d dimensions" % (what, t.dim()))


class Warp2dFn(torch.autograd.Function):
    """SpatialTransformer.forward (voxelmorph/torch/layers.py:30-48), 2-D."""

    @staticmethod
    def forward(ctx, src, flow, mode):
        _img(src, "SpatialTransformer")
        require_device(src, flow)
        src, flow = _c(src), _c(flow)
        B, C, H, W = src.shape
        if tuple(flow.shape) != (B, 2, H, W):
            raise ValueError("flow shape %s does not match src %s" % (tuple(flow.shape), tuple(src.shape)))
        out = torch.empty_like(src)
        with _prof.region("warp2d_fwd", nbytes=4.0 * B * H * W * (2 * C + 2)):
            call("vxm_warp2d_fwd", ptr(src), ptr(flow), ptr(out), B, C, H, W, INTERP[mode], stream())
        ctx.save_for_backward(src, flow)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, gout):
        src, flow = ctx.saved_tensors
        B, C, H, W = src.shape
        gsrc = torch.empty_like(src) if ctx.needs_input_grad[0] else None
        gflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        call("vxm_warp2d_bwd", ptr(src), ptr(flow), ptr(_c(gout)), ptr(gsrc), ptr(gflow), B, C, H, W, INTERP[ctx.mode], stream())
        return gsrc, gflow, None


class VecInt2dFn(torch.autograd.Function):
    """VecInt.forward (voxelmorph/torch/layers.py:64-68), 2-D, nsteps >= 1."""

    @staticmethod
    def forward(ctx, vec, nsteps):
        _img(vec, "VecInt")
        require_device(vec)
        vec = _c(vec)
        B, C, H, W = vec.shape
        if C != 2:
            raise ValueError("VecInt expects a 2-channel field for 2-D images, got %d" % C)
        steps = torch.empty((nsteps,) + tuple(vec.shape), dtype=vec.dtype, device=vec.device)
        call("vxm_vecint2d_fwd", ptr(vec), ptr(steps), B, H, W, nsteps, stream())
        ctx.save_for_backward(vec, steps)
        ctx.nsteps = nsteps
        return steps[nsteps - 1]

    @staticmethod
    def backward(ctx, gout):
        vec, steps = ctx.saved_tensors
        B, _, H, W = vec.shape
        gvec = torch.empty_like(vec)
        work = torch.empty(2 * vec.numel(), dtype=vec.dtype, device=vec.device)
        call("vxm_vecint2d_bwd", ptr(vec), ptr(steps), ptr(_c(gout)), ptr(gvec), ptr(work), B, H, W, ctx.nsteps, stream())
        return gvec, None


class Resize2dFn(torch.autograd.Function):
    """ResizeTransform.forward (voxelmorph/torch/layers.py:85-97), 'bilinear', factor != 1."""

    @staticmethod
    def forward(ctx, x, factor):
        _img(x, "ResizeTransform")
        require_device(x)
        x = _c(x)
        B, C, H, W = x.shape
        oH, oW = (int(math.floor(s * factor)) for s in (H, W))
        out = torch.empty((B, C, oH, oW), dtype=x.dtype, device=x.device)
        call("vxm_resize2d_fwd", ptr(x), ptr(out), B, C, H, W, oH, oW, float(factor), stream())
        ctx.shape = (B, C, H, W, oH, oW)
        ctx.factor = float(factor)
        return out

    @staticmethod
    def backward(ctx, gout):
        B, C, H, W, oH, oW = ctx.shape
        gx = torch.empty((B, C, H, W), dtype=gout.dtype, device=gout.device)
        call("vxm_resize2d_bwd", ptr(_c(gout)), ptr(gx), B, C, H, W, oH, oW, ctx.factor, stream())
        return gx, None


class NCC2dFn(torch.autograd.Function):
    """NCC.loss (voxelmorph/torch/losses.py:15-67) with a win x win window."""

    @staticmethod
    def forward(ctx, y_true, y_pred, win):
        _img(y_true, "NCC")
        require_device(y_true, y_pred)
        I, J = _c(y_true), _c(y_pred)
        B, C, H, W = I.shape
        if C != 1 or I.shape != J.shape:
            raise ValueError("NCC: expected two [B,1,H,W] tensors (the reference's box filter has one input channel, "
                             "losses.py:29), got %s / %s" % (tuple(I.shape), tuple(J.shape)))
        loss = torch.empty((), dtype=I.dtype, device=I.device)
        sums = torch.empty((5, B, H, W), dtype=I.dtype, device=I.device)
        work = torch.empty((5, B, H, W), dtype=I.dtype, device=I.device)
        acc = torch.empty(1, dtype=torch.float64, device=I.device)
        call("vxm_ncc2d_fwd", ptr(I), ptr(J), ptr(loss), ptr(sums), ptr(work), ptr(acc), B, H, W, win, stream())
        ctx.save_for_backward(I, J, sums)
        ctx.win = win
        return loss

    @staticmethod
    def backward(ctx, gloss):
        I, J, sums = ctx.saved_tensors
        B, _, H, W = I.shape
        gloss = _c(gloss)
        work = torch.empty((6, B, H, W), dtype=I.dtype, device=I.device)
        gI = gJ = None
        if ctx.needs_input_grad[1]:
            gJ = torch.empty_like(J)
            call("vxm_ncc2d_bwd", ptr(I), ptr(J), ptr(sums), ptr(gloss), ptr(gJ), ptr(work), B, H, W, ctx.win, stream())
        if ctx.needs_input_grad[0]:      # cc is symmetric in (I, J): box-sum planes 0<->1 and 2<->3
            swapped = torch.stack([sums[1], sums[0], sums[3], sums[2], sums[4]])
            gI = torch.empty_like(I)
            call("vxm_ncc2d_bwd", ptr(J), ptr(I), ptr(swapped), ptr(gloss), ptr(gI), ptr(work), B, H, W, ctx.win, stream())
        return gI, gJ, None


class NCC1dFn(torch.autograd.Function):
    """NCC.loss (voxelmorph/torch/losses.py:15-67) on 1-D signals [B,1,L] with a window of `win` taps."""

    @staticmethod
    def forward(ctx, y_true, y_pred, win):
        require_device(y_true, y_pred)
        I, J = _c(y_true), _c(y_pred)
        if I.dim() != 3 or I.shape[1] != 1 or I.shape != J.shape or I.dtype != torch.float32:
            raise ValueError("NCC: expected two fp32 [B,1,L] tensors (the reference's box filter has one input channel, "
                             "losses.py:29), got %s / %s" % (tuple(I.shape), tuple(J.shape)))
        B, _, L = I.shape
        loss = torch.empty((), dtype=I.dtype, device=I.device)
        sums = torch.empty((5, B, L), dtype=I.dtype, device=I.device)
        work = torch.empty((5, B, L), dtype=I.dtype, device=I.device)
        acc = torch.empty(1, dtype=torch.float64, device=I.device)
        call("vxm_ncc1d_fwd", ptr(I), ptr(J), ptr(loss), ptr(sums), ptr(work), ptr(acc), B, L, win, stream())
        ctx.save_for_backward(I, J, sums)
        ctx.win = win
        return loss

    @staticmethod
    def backward(ctx, gloss):
        I, J, sums = ctx.saved_tensors
        B, _, L = I.shape
        gloss = _c(gloss)
        work = torch.empty((6, B, L), dtype=I.dtype, device=I.device)
        gI = gJ = None
        if ctx.needs_input_grad[1]:
            gJ = torch.empty_like(J)
            call("vxm_ncc1d_bwd", ptr(I), ptr(J), ptr(sums), ptr(gloss), ptr(gJ), ptr(work), B, L, ctx.win, stream())
        if ctx.needs_input_grad[0]:      # cc is symmetric in (I, J): box-sum planes 0<->1 and 2<->3
            swapped = torch.stack([sums[1], sums[0], sums[3], sums[2], sums[4]])
            gI = torch.empty_like(I)
            call("vxm_ncc1d_bwd", ptr(J), ptr(I), ptr(swapped), ptr(gloss), ptr(gI), ptr(work), B, L, ctx.win, stream())
        return gI, gJ, None


class GradLoss2dFn(torch.autograd.Function):
    """Grad.loss (voxelmorph/torch/losses.py:102-135), 2-D."""

    @staticmethod
    def forward(ctx, y, penalty, mult):
        _img(y, "Grad")
        require_device(y)
        y = _c(y)
        B, C, H, W = y.shape
        loss = torch.empty((), dtype=y.dtype, device=y.device)
        acc = torch.empty(3 * B * 32, dtype=torch.float64, device=y.device)      # 3 * B * VXM_GRAD_SLOTS (include/vxm_hip.h)
        call("vxm_gradloss2d_fwd", ptr(y), ptr(loss), ptr(acc), B, C, H, W, PENALTY[penalty], float(mult), stream())
        ctx.save_for_backward(y)
        ctx.args = (penalty, float(mult))
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (y,) = ctx.saved_tensors
        B, C, H, W = y.shape
        gy = torch.empty_like(y)
        call("vxm_gradloss2d_bwd", ptr(y), ptr(_c(gloss)), ptr(gy), B, C, H, W, PENALTY[ctx.args[0]], ctx.args[1], stream())
        return gy, None, None


class MaxPool2dFn(torch.autograd.Function):
    """MaxPool2d(2) (voxelmorph/torch/networks.py:83-84,130)."""

    @staticmethod
    def forward(ctx, x):
        _img(x, "MaxPool")
        require_device(x)
        x = _c(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
        call("vxm_maxpool2d_fwd", ptr(x), ptr(y), B, C, H, W, stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        call("vxm_maxpool2d_bwd", ptr(x), ptr(_c(gy)), ptr(gx), B, C, H, W, stream())
        return gx


class UpsampleCat2dFn(torch.autograd.Function):
    """cat([Upsample(2,'nearest')(x), skip], 1) (voxelmorph/torch/networks.py:85,137-138); skip may be None."""

    @staticmethod
    def forward(ctx, x, skip):
        _img(x, "Upsample")
        require_device(x, skip)
        x = _c(x)
        skip = _c(skip) if skip is not None else None
        B, C0, h, w = x.shape
        C1 = skip.shape[1] if skip is not None else 0
        if skip is not None and tuple(skip.shape[2:]) != (2 * h, 2 * w):
            raise ValueError("Unet: skip %s does not match the upsampled %s" % (tuple(skip.shape), (2 * h, 2 * w)))
        out = torch.empty((B, C0 + C1, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
        call("vxm_upsample2d_cat", ptr(x), C0, ptr(skip), C1, ptr(out), B, 2 * h, 2 * w, stream())
        ctx.dims = (B, C0, C1, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C0, C1, h, w = ctx.dims
        g = _c(g)
        gx = torch.empty((B, C0, h, w), dtype=g.dtype, device=g.device)
        call("vxm_upsample2d_bwd", ptr(g), C0 + C1, ptr(gx), C0, B, h, w, stream())
        gskip = g[:, C0:].contiguous() if C1 and ctx.needs_input_grad[1] else None
        return gx, gskip


def conv2d(x, w, b, slope):
    """conv2d(k3, p1) + bias + LeakyReLU(slope) through the 3-D MFMA kernels at depth one."""
    _img(x, "ConvBlock")
    if tuple(w.shape[2:]) != (3, 3):
        raise ValueError("conv weight %s: 3x3 kernels only" % (tuple(w.shape),))
    w3 = F.pad(w.unsqueeze(2), (0, 0, 0, 0, 1, 1))          # [Cout,Cin,3,3] -> middle slice of [Cout,Cin,3,3,3]
    return ConvFn.apply(x.unsqueeze(2), w3, b, slope).squeeze(2)
