"""Drop-in mirror of `voxelmorph/torch/layers.py` (reference) on the MI355X HIP kernels.

Same class names, constructor signatures, attributes and buffers as the reference; `forward` dispatches to libvxm_hip.so
through `functional.py` (3-D volumes) or `planar.py` (2-D images).  There is no CPU / ATen path behind these modules.
"""
import torch
from torch import nn

from . import functional as VF
from . import planar as VP

_LINEAR_MODE = {1: 'linear', 2: 'bilinear', 3: 'trilinear'}


def _identity_grid(size):
    """[1, N, *size] fp32 voxel-index grid in 'ij' order (what the reference registers as `grid`, layers.py:17-28)."""
    axes = [torch.arange(int(s), dtype=torch.float32) for s in size]
    return torch.stack(torch.meshgrid(*axes, indexing='ij')).unsqueeze(0)


class SpatialTransformer(nn.Module):
    """N-D spatial transformer (reference: layers.py:6-48): samples `src` at `identity + flow`, 'bilinear' or 'nearest'.

    `grid` exists only for state-dict parity with the reference (it is in `state_dict()`, stripped by `LoadableModel.save`):
    the kernels compute voxel indices in registers, reproduce the reference's normalise / un-normalise round trip in fp32
    (`vxm_src_coord`) and never read the buffer.
    """

    def __init__(self, size, mode='bilinear'):
        super().__init__()
        if mode not in VF.INTERP:
            raise ValueError("mode should be 'bilinear' or 'nearest', got %r" % (mode,))
        self.mode = mode
        self.register_buffer('grid', _identity_grid(size))

    def forward(self, src, flow):
        expected = tuple(self.grid.shape[2:])
        if tuple(flow.shape[2:]) != expected:
            raise RuntimeError("flow spatial shape %s does not match the transformer size %s" % (tuple(flow.shape[2:]), expected))
        warp = VP.Warp2dFn if src.dim() == 4 else VF.WarpFn
        return warp.apply(src, flow, self.mode)


class VecInt(nn.Module):
    """Integrates a stationary velocity field by scaling and squaring (reference: layers.py:51-68):
    `v <- v / 2^nsteps`, then `nsteps` times `v <- v + v o (id + v)`."""

    def __init__(self, inshape, nsteps):
        super().__init__()
        assert nsteps >= 0, 'nsteps should be >= 0, found: %d' % nsteps
        self.nsteps = nsteps
        self.scale = 1.0 / (2 ** nsteps)
        self.transformer = SpatialTransformer(inshape)      # kept as a sub-module: its grid is part of the reference state dict

    def forward(self, vec):
        if self.nsteps == 0:
            return vec * self.scale                          # scale == 1: nothing to integrate
        integrate = VP.VecInt2dFn if vec.dim() == 4 else VF.VecIntFn
        return integrate.apply(vec, self.nsteps)


class ResizeTransform(nn.Module):
    """Resize a displacement field: resample by `1 / vel_resize` (linear, align_corners) AND rescale the vectors by the same
    factor (reference: layers.py:71-97; the rescale comes after the resampling when shrinking, before it when enlarging)."""

    def __init__(self, vel_resize, ndims):
        super().__init__()
        self.factor = 1.0 / vel_resize
        self.mode = _LINEAR_MODE.get(ndims, 'linear')

    def forward(self, x):
        if self.factor == 1:
            return x
        resize = VP.Resize2dFn if x.dim() == 4 else VF.ResizeFn
        return resize.apply(x, self.factor)
