"""Drop-in mirror of `voxelmorph/torch/networks.py` (reference): Unet, ConvBlock, VxmDense.

Parameter names, shapes and initialisation follow the reference so that `state_dict()` keys match
(`unet_model.encoder.{l}.{c}.main.weight`, ..., `flow.weight`); the forward/backward run on the
fused HIP engine of `functional.py` instead of ~680 ATen launches per step.
"""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.distributions.normal import Normal

from . import functional as VF
from . import functional_bf16 as VB
from . import layers
from . import planar as VP
from .modelio import LoadableModel, store_config_args


def default_unet_features():
    """reference: voxelmorph/py/utils.py:16-21"""
    return [[16, 32, 32, 32], [32, 32, 32, 32, 32, 16, 16]]


class _Conv3dParams(nn.Module):
    """Weight/bias holder initialised like torch.nn.Conv3d / Conv2d (kaiming_uniform(a=sqrt(5)) + fan-in bias)."""

    def __init__(self, in_channels, out_channels, ndims=3):
        super().__init__()
        self.in_channels, self.out_channels, self.ndims = in_channels, out_channels, ndims
        self.weight = nn.Parameter(torch.empty((out_channels, in_channels) + (3,) * ndims))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_channels * 3 ** ndims)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x, slope=1.0):
        if self.ndims == 2:
            return VP.conv2d(x, self.weight, self.bias, slope)
        return VF.ConvFn.apply(x, self.weight, self.bias, slope)


class ConvBlock(nn.Module):
    """ConvNd(3, stride, pad 1) + LeakyReLU(0.2) (reference: networks.py:290-305).  The U-Net only ever builds stride 1; a strided block
    (accepted by the reference's constructor) is the stride-1 block's output sampled at every stride-th voxel -- with kernel 3 and padding
    1 output i of the strided convolution IS output stride * i of the stride-1 one, and LeakyReLU commutes with the sampling -- so it runs
    the same HIP kernels and takes a strided view (autograd zero-fills the skipped voxels on the way back)."""

    def __init__(self, ndims, in_channels, out_channels, stride=1):
        super().__init__()
        if ndims not in (2, 3):
            raise NotImplementedError("the MI355X ConvBlock implements 2-D / 3-D kernel-3 convolutions")
        if int(stride) != stride or stride < 1:
            raise ValueError("ConvBlock: stride must be a positive integer, got %r" % (stride,))
        self.stride = int(stride)
        self.main = _Conv3dParams(in_channels, out_channels, ndims)
        self.activation = nn.LeakyReLU(0.2)          # attribute kept for parity; fused into the conv epilogue

    def forward(self, x):
        y = self.main(x, slope=0.2)
        if self.stride != 1:
            y = y[(slice(None), slice(None)) + (slice(None, None, self.stride),) * (y.dim() - 2)]
        return y


def _unet_feature_plan(nb_features, nb_levels, feat_mult, nb_conv_per_level):
    """Feature bookkeeping of the reference U-Net (networks.py:57-77) as a pure function.

    `nb_features` is either `[encoder features, decoder features]` (decoder entries beyond the encoder's length are the extra
    full-resolution convolutions) or an int, in which case level l has `round(nb_features * feat_mult**l)` features, every
    level holds `nb_conv_per_level` convolutions and `nb_levels` is required.  Returns (enc, dec, final, levels)."""
    if nb_features is None:
        nb_features = default_unet_features()
    if isinstance(nb_features, int):
        if nb_levels is None:
            raise ValueError('must provide unet nb_levels if nb_features is an integer')
        per_level = [int(f) for f in np.round(nb_features * feat_mult ** np.arange(nb_levels))]
        enc = [f for f in per_level[:-1] for _ in range(nb_conv_per_level)]
        dec = [f for f in reversed(per_level) for _ in range(nb_conv_per_level)]
    else:
        if nb_levels is not None:
            raise ValueError('cannot use nb_levels if nb_features is not an integer')
        enc, dec = ([int(f) for f in group] for group in nb_features)
    n = len(enc)
    return enc, dec[:n], dec[n:], n // nb_conv_per_level + 1


class Unet(nn.Module):
    """U-Net of the reference (networks.py:12-144): per level `nb_conv_per_level` ConvBlocks, MaxPool(2) down, nearest x2 up
    with a skip concat, then the remaining full-resolution ConvBlocks.  Default features: encoder [16, 32, 32, 32], decoder
    [32, 32, 32, 32, 32, 16, 16].  Module tree and parameter names are the reference's (`encoder.{level}.{conv}.main.weight`, ...).
    """

    def __init__(self, inshape=None, infeats=None, nb_features=None, nb_levels=None, max_pool=2, feat_mult=1,
                 nb_conv_per_level=1, half_res=False):
        super().__init__()
        ndims = len(inshape)
        assert ndims in [1, 2, 3], 'ndims should be one of 1, 2, or 3. found: %d' % ndims
        if ndims == 1:
            raise NotImplementedError("the MI355X Unet implements 2-D images and 3-D volumes")
        enc_nf, dec_nf, final_nf, self.nb_levels = _unet_feature_plan(nb_features, nb_levels, feat_mult, nb_conv_per_level)
        # networks.py:79-85: one factor per level, an int is repeated.  Level l pools by max_pool[l] on the way down and upsamples by
        # max_pool[l] on the way up (the SAME index, not the mirrored one -- as the reference); factors other than 2 take the op-by-op path
        self.max_pool = [max_pool] * self.nb_levels if isinstance(max_pool, int) else list(max_pool)
        if len(self.max_pool) < self.nb_levels - 1:
            raise IndexError("Unet: max_pool lists %d factors for %d pooling levels" % (len(self.max_pool), self.nb_levels - 1))
        self._pool2 = all(isinstance(p, int) and p == 2 for p in self.max_pool[:self.nb_levels - 1])
        self.ndims, self.half_res, self.nb_conv_per_level = ndims, half_res, nb_conv_per_level
        self._enc_nf, self._dec_nf, self._final_nf, self._infeats = enc_nf, dec_nf, final_nf, infeats

        def level_blocks(features, level, width):
            """ModuleList of the ConvBlocks of one level; returns it with the channel count it ends on."""
            blocks = nn.ModuleList()
            for nf in features[level * nb_conv_per_level:(level + 1) * nb_conv_per_level]:
                blocks.append(ConvBlock(ndims, width, nf))
                width = nf
            return blocks, width

        width = infeats
        skips = [infeats]                            # channels available for the skip concat at every resolution, coarsest last
        self.encoder = nn.ModuleList()
        for level in range(self.nb_levels - 1):
            blocks, width = level_blocks(enc_nf, level, width)
            self.encoder.append(blocks)
            skips.append(width)
        self.decoder = nn.ModuleList()
        for level in range(self.nb_levels - 1):
            blocks, width = level_blocks(dec_nf, level, width)
            self.decoder.append(blocks)
            if not half_res or level < self.nb_levels - 2:      # upsample + concat with the encoder output of that resolution
                width += skips[-1 - level]
        self.remaining = nn.ModuleList()
        for nf in final_nf:
            self.remaining.append(ConvBlock(ndims, width, nf))
            width = nf
        self.final_nf = width
        self._plans = {}

    def conv_params(self):
        """[w0, b0, w1, b1, ...] in execution order (encoder, decoder, remaining)."""
        out = []
        for group in list(self.encoder) + list(self.decoder):
            for blk in group:
                out += [blk.main.weight, blk.main.bias]
        for blk in self.remaining:
            out += [blk.main.weight, blk.main.bias]
        return out

    def plan(self, in_channels, extra=()):
        key = (tuple(in_channels), tuple(extra))
        if key not in self._plans:
            if sum(in_channels) != self._infeats:
                raise ValueError("Unet built for %d input features, got %s" % (self._infeats, list(in_channels)))
            self._plans[key] = VF.UnetPlan(list(in_channels), self._enc_nf, self._dec_nf, self._final_nf, self.nb_levels,
                                           self.nb_conv_per_level, self.half_res, extra=extra)
        return self._plans[key]

    def forward(self, x):
        if not self._pool2:
            return self._forward_general(x)
        if self.ndims == 2:
            return self._forward_planar(x)
        engine = VB.UnetBf16Fn if VB.enabled() else VF.UnetFn       # torch.autocast(bfloat16): blocked-bf16 activations
        return engine.apply(self.plan([x.shape[1]]), x, *self.conv_params())

    def _forward_general(self, x):
        """networks.py:122-144 op by op for pooling factors other than 2 (images and volumes): ConvBlocks on the HIP conv kernels,
        MaxPoolNd(k) and Upsample(k, 'nearest') + cat on the general-factor kernels (csrc/pool.hip).  Not reachable through VxmDense."""
        x_history = [x]
        for level, convs in enumerate(self.encoder):
            for conv in convs:
                x = conv(x)
            x_history.append(x)
            x = VF.MaxPoolKFn.apply(x, self.max_pool[level])
        for level, convs in enumerate(self.decoder):
            for conv in convs:
                x = conv(x)
            if not self.half_res or level < (self.nb_levels - 2):
                x = VF.UpsampleCatKFn.apply(x, x_history.pop(), self.max_pool[level])
        for conv in self.remaining:
            x = conv(x)
        return x

    def _forward_planar(self, x):
        """networks.py:122-144 op by op (2-D slices are small: per-op autograd nodes instead of the fused 3-D engine)."""
        nlev = self.nb_levels - 1
        if any(int(d) % (1 << nlev) for d in x.shape[2:]):
            raise ValueError("Unet: image %s must be divisible by %d (MaxPool floors and the skip concat would not "
                             "line up, networks.py:130,138)" % (tuple(x.shape[2:]), 1 << nlev))
        x_history = [x]
        for convs in self.encoder:
            for conv in convs:
                x = conv(x)
            x_history.append(x)
            x = VP.MaxPool2dFn.apply(x)
        for level, convs in enumerate(self.decoder):
            for conv in convs:
                x = conv(x)
            if not self.half_res or level < (self.nb_levels - 2):
                x = VP.UpsampleCat2dFn.apply(x, x_history.pop())
        for conv in self.remaining:
            x = conv(x)
        return x


class VxmDense(LoadableModel):
    """VoxelMorph network for (unsupervised) nonlinear registration between two images
    (reference: networks.py:147-287; same constructor arguments, attributes and return values)."""

    @store_config_args
    def __init__(self, inshape, nb_unet_features=None, nb_unet_levels=None, unet_feat_mult=1, nb_unet_conv_per_level=1,
                 int_steps=7, int_downsize=2, bidir=False, use_probs=False, src_feats=1, trg_feats=1, unet_half_res=False):
        super().__init__()
        if use_probs:                                # as the reference: the probabilistic variant exists only in its TF backend
            raise NotImplementedError('Flow variance has not been implemented in pytorch - set use_probs to False')
        ndims = len(inshape)
        assert ndims in [1, 2, 3], 'ndims should be one of 1, 2, or 3. found: %d' % ndims
        self.training = True                         # the reference sets the flag explicitly (networks.py:192)
        self.ndims, self.bidir = ndims, bidir
        self._feats = (src_feats, trg_feats)

        self.unet_model = Unet(inshape, infeats=src_feats + trg_feats, nb_features=nb_unet_features, nb_levels=nb_unet_levels,
                               feat_mult=unet_feat_mult, nb_conv_per_level=nb_unet_conv_per_level, half_res=unet_half_res)
        # flow head: ndims output channels, weights ~ N(0, 1e-5), zero bias (networks.py:210-215): training starts at the identity
        self.flow = _Conv3dParams(self.unet_model.final_nf, ndims, ndims)
        with torch.no_grad():
            self.flow.weight.copy_(Normal(0, 1e-5).sample(self.flow.weight.shape))
            self.flow.bias.zero_()

        # The velocity field is integrated at 1 / int_downsize of the image resolution: shrink it first (unless the U-Net
        # already stops at half resolution), integrate, bring the result back to full size (networks.py:223-242).
        integrating, reduced = int_steps > 0, int_downsize > 1
        self.resize = layers.ResizeTransform(int_downsize, ndims) if (integrating and reduced and not unet_half_res) else None
        self.fullsize = layers.ResizeTransform(1 / int_downsize, ndims) if (integrating and reduced) else None
        self.integrate = layers.VecInt([int(extent / int_downsize) for extent in inshape], int_steps) if integrating else None
        self.transformer = layers.SpatialTransformer(inshape)

    def _forward_all(self, source, target, need_disp=True):
        """Everything `forward` can return (reference: networks.py:244-287): (moved source, moved target or None, the field
        the smoothness loss sees, positive displacement, negative displacement or None).  need_disp=False (the training outputs): the
        full-resolution displacements are not returned, and when `fullsize` upsamples an integrated field they are never materialised --
        `fullsize` + `transformer` run as one kernel (functional.WarpUpFn) on the half-resolution field."""
        if self.ndims == 2:
            field = self.flow(self.unet_model(torch.cat([source, target], dim=1)))
        else:
            # U-Net + flow head as ONE fused autograd node; source and target enter as a virtual concat
            plan = self.unet_model.plan(self._feats, extra=((self.flow.out_channels, 1.0),))
            # under torch.autocast(bfloat16) the activations between the convolutions are blocked bf16 (functional_bf16.py);
            # the field that leaves the flow head, and everything that consumes it, stays fp32
            engine = VB.UnetBf16Fn if VB.enabled() else VF.UnetFn
            field = engine.apply(plan, source, target, *self.unet_model.conv_params(), self.flow.weight, self.flow.bias)
        velocity = self.resize(field) if self.resize is not None else field       # what Grad regularises ("preint_flow")
        v_int = velocity
        if self.ndims == 3 and velocity.requires_grad and velocity.is_cuda and velocity.dtype == torch.float32:
            # the field has two consumers (the caller's Grad loss; the integration or, with int_steps = 0, the warp itself): their gradients are
            # summed by this library's kernel instead of autograd's ATen add
            velocity, v_int = VF.ForkFn.apply(velocity)

        def displacement(v):
            """velocity -> full-resolution displacement: scaling and squaring, then back to the image grid"""
            if self.integrate is not None:
                v = self.integrate(v)
                if self.fullsize is not None:
                    v = self.fullsize(v)
            return v

        def moved(image, v):
            """image warped by the displacement of velocity v -> (moved image, displacement or None)"""
            if self.ndims == 3 and self.integrate is not None and self.fullsize is not None and self.fullsize.factor != 1:
                low = self.integrate(v)
                # (a displacement that is RETURNED while gradients are recorded must stay differentiable -- the semi-supervised head warps
                # labels with it: that case keeps the two-kernel path, whose pos_flow is an autograd output)
                if VF.warp_up_ok(image, low) and not (need_disp and torch.is_grad_enabled() and low.requires_grad):
                    if need_disp:
                        return VF.WarpUpFn.apply(image, low, self.fullsize.factor, self.transformer.mode, True)
                    return VF.WarpUpFn.apply(image, low, self.fullsize.factor, self.transformer.mode, False), None
                disp = self.fullsize(low)
            else:
                disp = displacement(v)
            return self.transformer(image, disp), disp

        moved_source, forward_disp = moved(source, v_int)
        moved_target, backward_disp = moved(target, -v_int) if self.bidir else (None, None)
        return moved_source, moved_target, velocity, forward_disp, backward_disp

    def forward(self, source, target, registration=False):
        moved_source, moved_target, velocity, forward_disp, _ = self._forward_all(source, target, need_disp=registration)
        if registration:
            return moved_source, forward_disp
        return (moved_source, moved_target, velocity) if self.bidir else (moved_source, velocity)


class VxmDenseSemiSupervisedSeg(LoadableModel):
    """VoxelMorph network for (semi-supervised) nonlinear registration between two images: VxmDense plus a warped
    down-sampled source segmentation for an auxiliary Dice loss.

    The reference implements this model only for its TensorFlow backend (voxelmorph/tf/networks.py:287-388, trained by
    scripts/tf/train_semisupervised_seg.py:117-140); this is the same wiring on the torch-side API of this package:
    `seg_flow = RescaleTransform(1/seg_resolution)(pos_flow)` (tf/networks.py:336-337) is `ResizeTransform(seg_resolution)`
    here (torch/layers.py:76-97: resample by 1/seg_resolution and rescale), and the one-hot segmentation
    [B, nb_labels, *inshape/seg_resolution] is warped with a linear SpatialTransformer (tf/networks.py:338-339).

    forward(source, target, seg_src[, seg_trg]) returns VxmDense's training outputs followed by the warped
    segmentation(s): (y_source, [y_target,] preint_flow, y_seg_src[, y_seg_trg]).
    """

    @store_config_args
    def __init__(self, inshape, nb_labels, nb_unet_features=None, seg_resolution=2, bidir=False, bidir_labels=False, **kwargs):
        super().__init__()
        if bidir_labels:                       # tf/networks.py:319-321
            bidir = True
        self.vxm_model = VxmDense(inshape, nb_unet_features=nb_unet_features, bidir=bidir, **kwargs)
        ndims = len(inshape)
        self.nb_labels = nb_labels
        self.bidir, self.bidir_labels = bidir, bidir_labels
        inshape_ds = [int(s) for s in (np.array(inshape) / seg_resolution).astype(int)]      # tf/networks.py:332
        self.seg_resize = layers.ResizeTransform(seg_resolution, ndims)
        self.seg_transformer = layers.SpatialTransformer(inshape_ds)

    def forward(self, source, target, seg_src, seg_trg=None):
        y_source, y_target, preint_flow, pos_flow, neg_flow = self.vxm_model._forward_all(source, target)
        outs = [y_source] + ([y_target] if self.bidir else []) + [preint_flow]
        outs.append(self.seg_transformer(seg_src, self.seg_resize(pos_flow)))
        if self.bidir_labels:
            if seg_trg is None:
                raise ValueError("bidir_labels=True needs the target segmentation as fourth input")
            outs.append(self.seg_transformer(seg_trg, self.seg_resize(neg_flow)))
        return tuple(outs)

    def register(self, src, trg):
        """Predicts the transform from src to trg tensors (tf/networks.py:370-374)."""
        return self.vxm_model(src, trg, registration=True)[1]

    def apply_transform(self, src, trg, img, interp_method='linear'):
        """Predicts the transform from src to trg and applies it to img (tf/networks.py:376-388);
        interp_method 'nearest' gives the bit-exact label warp used for Dice evaluation."""
        flow = self.register(src, trg)
        mode = 'bilinear' if interp_method == 'linear' else interp_method
        return layers.SpatialTransformer(flow.shape[2:], mode=mode).to(flow.device)(img, flow)
