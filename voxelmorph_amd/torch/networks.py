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
    """ConvNd(3, stride 1, pad 1) + LeakyReLU(0.2) (reference: networks.py:290-305)."""

    def __init__(self, ndims, in_channels, out_channels, stride=1):
        super().__init__()
        if ndims not in (2, 3) or stride != 1:
            raise NotImplementedError("the MI355X ConvBlock implements 2-D / 3-D, stride-1, kernel-3 convolutions")
        self.main = _Conv3dParams(in_channels, out_channels, ndims)
        self.activation = nn.LeakyReLU(0.2)          # attribute kept for parity; fused into the conv epilogue

    def forward(self, x):
        return self.main(x, slope=0.2)


class Unet(nn.Module):
    """U-Net with the reference's feature bookkeeping (networks.py:12-144).

    Default features: encoder [16, 32, 32, 32], decoder [32, 32, 32, 32, 32, 16, 16].
    """

    def __init__(self, inshape=None, infeats=None, nb_features=None, nb_levels=None, max_pool=2, feat_mult=1,
                 nb_conv_per_level=1, half_res=False):
        super().__init__()
        ndims = len(inshape)
        assert ndims in [1, 2, 3], 'ndims should be one of 1, 2, or 3. found: %d' % ndims
        if ndims == 1:
            raise NotImplementedError("the MI355X Unet implements 2-D images and 3-D volumes")
        self.ndims = ndims
        self.half_res = half_res
        if nb_features is None:
            nb_features = default_unet_features()
        if isinstance(nb_features, int):
            if nb_levels is None:
                raise ValueError('must provide unet nb_levels if nb_features is an integer')
            feats = np.round(nb_features * feat_mult ** np.arange(nb_levels)).astype(int)
            nb_features = [np.repeat(feats[:-1], nb_conv_per_level), np.repeat(np.flip(feats), nb_conv_per_level)]
        elif nb_levels is not None:
            raise ValueError('cannot use nb_levels if nb_features is not an integer')
        enc_nf, dec_nf = [[int(f) for f in fs] for fs in nb_features]
        nb_dec_convs = len(enc_nf)
        final_convs = dec_nf[nb_dec_convs:]
        dec_nf = dec_nf[:nb_dec_convs]
        self.nb_levels = int(nb_dec_convs / nb_conv_per_level) + 1
        if isinstance(max_pool, int):
            max_pool = [max_pool] * self.nb_levels
        if any(p != 2 for p in max_pool):
            raise NotImplementedError("the MI355X Unet implements max_pool=2")
        self.nb_conv_per_level = nb_conv_per_level
        self._enc_nf, self._dec_nf, self._final_nf = enc_nf, dec_nf, list(final_convs)
        self._infeats = infeats

        prev_nf = infeats
        encoder_nfs = [prev_nf]
        self.encoder = nn.ModuleList()
        for level in range(self.nb_levels - 1):
            convs = nn.ModuleList()
            for conv in range(nb_conv_per_level):
                nf = enc_nf[level * nb_conv_per_level + conv]
                convs.append(ConvBlock(ndims, prev_nf, nf))
                prev_nf = nf
            self.encoder.append(convs)
            encoder_nfs.append(prev_nf)
        encoder_nfs = encoder_nfs[::-1]
        self.decoder = nn.ModuleList()
        for level in range(self.nb_levels - 1):
            convs = nn.ModuleList()
            for conv in range(nb_conv_per_level):
                nf = dec_nf[level * nb_conv_per_level + conv]
                convs.append(ConvBlock(ndims, prev_nf, nf))
                prev_nf = nf
            self.decoder.append(convs)
            if not half_res or level < (self.nb_levels - 2):
                prev_nf += encoder_nfs[level]
        self.remaining = nn.ModuleList()
        for nf in final_convs:
            self.remaining.append(ConvBlock(ndims, prev_nf, nf))
            prev_nf = nf
        self.final_nf = prev_nf
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
        if self.ndims == 2:
            return self._forward_planar(x)
        return VF.UnetFn.apply(self.plan([x.shape[1]]), x, *self.conv_params())

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
        self.training = True
        ndims = len(inshape)
        assert ndims in [1, 2, 3], 'ndims should be one of 1, 2, or 3. found: %d' % ndims
        self.unet_model = Unet(inshape, infeats=(src_feats + trg_feats), nb_features=nb_unet_features,
                               nb_levels=nb_unet_levels, feat_mult=unet_feat_mult,
                               nb_conv_per_level=nb_unet_conv_per_level, half_res=unet_half_res)
        self.flow = _Conv3dParams(self.unet_model.final_nf, ndims, ndims)
        self.ndims = ndims
        self.flow.weight = nn.Parameter(Normal(0, 1e-5).sample(self.flow.weight.shape))
        self.flow.bias = nn.Parameter(torch.zeros(self.flow.bias.shape))
        if use_probs:
            raise NotImplementedError('Flow variance has not been implemented in pytorch - set use_probs to False')
        if not unet_half_res and int_steps > 0 and int_downsize > 1:
            self.resize = layers.ResizeTransform(int_downsize, ndims)
        else:
            self.resize = None
        if int_steps > 0 and int_downsize > 1:
            self.fullsize = layers.ResizeTransform(1 / int_downsize, ndims)
        else:
            self.fullsize = None
        self.bidir = bidir
        down_shape = [int(dim / int_downsize) for dim in inshape]
        self.integrate = layers.VecInt(down_shape, int_steps) if int_steps > 0 else None
        self.transformer = layers.SpatialTransformer(inshape)
        self._feats = (src_feats, trg_feats)

    def _forward_all(self, source, target):
        """(y_source, y_target, preint_flow, pos_flow, neg_flow) of networks.py:244-287 (None where not bidir)."""
        if self.ndims == 2:
            flow_field = self.flow(self.unet_model(torch.cat([source, target], dim=1)))
        else:
            # U-Net + flow conv as ONE fused autograd node; source/target enter as a virtual concat
            plan = self.unet_model.plan(self._feats, extra=((self.flow.out_channels, 1.0),))
            flow_field = VF.UnetFn.apply(plan, source, target, *self.unet_model.conv_params(), self.flow.weight, self.flow.bias)
        pos_flow = flow_field
        if self.resize:
            pos_flow = self.resize(pos_flow)
        preint_flow = pos_flow
        neg_flow = -pos_flow if self.bidir else None
        if self.integrate:
            pos_flow = self.integrate(pos_flow)
            neg_flow = self.integrate(neg_flow) if self.bidir else None
            if self.fullsize:
                pos_flow = self.fullsize(pos_flow)
                neg_flow = self.fullsize(neg_flow) if self.bidir else None
        y_source = self.transformer(source, pos_flow)
        y_target = self.transformer(target, neg_flow) if self.bidir else None
        return y_source, y_target, preint_flow, pos_flow, neg_flow

    def forward(self, source, target, registration=False):
        y_source, y_target, preint_flow, pos_flow, _ = self._forward_all(source, target)
        if not registration:
            return (y_source, y_target, preint_flow) if self.bidir else (y_source, preint_flow)
        return y_source, pos_flow


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
