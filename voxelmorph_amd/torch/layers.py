"""Drop-in mirror of `voxelmorph/torch/layers.py` (reference) on the MI355X HIP kernels.

Same class names, constructor signatures, attributes and buffers; `forward` dispatches to
libvxm_hip.so through `functional.py` (3-D volumes) or `planar.py` (2-D images).
"""
import torch
import torch.nn as nn

from . import functional as VF
from . import planar as VP


class SpatialTransformer(nn.Module):
    """N-D Spatial Transformer (reference: layers.py:6-48).

    `grid` is kept as a registered fp32 buffer for state-dict parity (layers.py:17-28; stripped
    by `LoadableModel.save`), but the kernel computes voxel indices in registers and never
    reads it.
    """

    def __init__(self, size, mode='bilinear'):
        super().__init__()
        if mode not in VF.INTERP:
            raise ValueError("mode should be 'bilinear' or 'nearest', got %r" % (mode,))
        self.mode = mode
        vectors = [torch.arange(0, s) for s in size]
        grid = torch.stack(torch.meshgrid(*vectors, indexing='ij')).unsqueeze(0).type(torch.FloatTensor)
        self.register_buffer('grid', grid)

    def forward(self, src, flow):
        if tuple(flow.shape[2:]) != tuple(self.grid.shape[2:]):
            raise RuntimeError("flow spatial shape %s does not match the transformer size %s"
                               % (tuple(flow.shape[2:]), tuple(self.grid.shape[2:])))
        if src.dim() == 4:
            return VP.Warp2dFn.apply(src, flow, self.mode)
        return VF.WarpFn.apply(src, flow, self.mode)


class VecInt(nn.Module):
    """Integrates a vector field via scaling and squaring (reference: layers.py:51-68)."""

    def __init__(self, inshape, nsteps):
        super().__init__()
        assert nsteps >= 0, 'nsteps should be >= 0, found: %d' % nsteps
        self.nsteps = nsteps
        self.scale = 1.0 / (2 ** self.nsteps)
        self.transformer = SpatialTransformer(inshape)

    def forward(self, vec):
        if self.nsteps == 0:
            return vec * self.scale          # scale == 1
        if vec.dim() == 4:
            return VP.VecInt2dFn.apply(vec, self.nsteps)
        return VF.VecIntFn.apply(vec, self.nsteps)


class ResizeTransform(nn.Module):
    """Resize a transform: resample the field *and* rescale it (reference: layers.py:71-97)."""

    def __init__(self, vel_resize, ndims):
        super().__init__()
        self.factor = 1.0 / vel_resize
        self.mode = 'linear'
        if ndims == 2:
            self.mode = 'bi' + self.mode
        elif ndims == 3:
            self.mode = 'tri' + self.mode

    def forward(self, x):
        if self.factor == 1:
            return x
        if x.dim() == 4:
            return VP.Resize2dFn.apply(x, self.factor)
        return VF.ResizeFn.apply(x, self.factor)
