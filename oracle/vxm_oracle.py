"""TEST INFRASTRUCTURE ONLY — CPU oracle for the VxmDense hot path.

A functional restatement, on CPU tensors, of the reference torch backend
(`/root/reference/voxelmorph/torch/{layers,networks,losses}.py`).  It is the
*checker* for the HIP path: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  The product package
(`voxelmorph_amd`) never does.

Pinning: `tests/test_oracle_golden.py` checks every function here against
fixtures under `tests/golden/` that were produced by the *unmodified*
reference (torch 2.10.0 CPU) via `tests/golden/make_golden.py`, and — when
`/root/reference` is present — against the live reference.  The reference
itself ships no tests or golden vectors.

Where the arithmetic really lives is PyTorch ATen (`grid_sampler_3d`,
`upsample_trilinear3d`, `convolution`, `max_pool3d`, `upsample_nearest3d`);
the `*_explicit` functions below restate that arithmetic in numpy, operation
by operation, so the HIP kernels can be checked against something that is not
ATen (and bit-exactly for the nearest-neighbour warp).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# layers.py
# ----------------------------------------------------------------------------


def identity_grid(size):
    """`SpatialTransformer.__init__` layers.py:17-21: ij-meshgrid of arange, fp32, [1,N,*size]."""
    axes = [torch.arange(0, s) for s in size]
    mesh = torch.meshgrid(*axes, indexing="ij")
    return torch.stack(mesh).unsqueeze(0).to(torch.float32)


def spatial_transformer(src, flow, mode="bilinear"):
    """`SpatialTransformer.forward` layers.py:30-48 (N-D, N in {2,3})."""
    shape = flow.shape[2:]
    locs = identity_grid(shape).to(flow) + flow                      # :32
    chans = []
    for i, s in enumerate(shape):                                      # :36-37
        chans.append(2 * (locs[:, i] / (s - 1) - 0.5))
    locs = torch.stack(chans, dim=-1)                                  # channels last  :45
    locs = locs.flip(-1)                                               # (x,y,z) order   :46
    return F.grid_sample(src, locs, align_corners=True, mode=mode)    # :48


def vecint(vec, nsteps):
    """`VecInt.forward` layers.py:64-68 (scaling and squaring)."""
    assert nsteps >= 0                                                 # :59
    vec = vec * (1.0 / (2 ** nsteps))                                  # :61,:65
    for _ in range(nsteps):
        vec = vec + spatial_transformer(vec, vec)                      # :66-67
    return vec


def resize_transform(x, vel_resize):
    """`ResizeTransform.forward` layers.py:85-97."""
    ndims = x.dim() - 2
    factor = 1.0 / vel_resize                                          # :78
    mode = {1: "linear", 2: "bilinear", 3: "trilinear"}[ndims]        # :79-83
    if factor < 1:                                                     # :86-89
        x = F.interpolate(x, align_corners=True, scale_factor=factor, mode=mode)
        x = factor * x
    elif factor > 1:                                                   # :91-94
        x = factor * x
        x = F.interpolate(x, align_corners=True, scale_factor=factor, mode=mode)
    return x


# ----------------------------------------------------------------------------
# losses.py
# ----------------------------------------------------------------------------


def ncc_loss(y_true, y_pred, win=None, dtype=None):
    """`NCC.loss` losses.py:15-67 with the filter created on the input's device
    (the reference hard-codes "cuda" at :29).  `dtype=torch.float64` gives the
    arbiter the fp32 formula is judged against (SURVEY.md §7 "NCC conditioning")."""
    I, J = y_true, y_pred
    if dtype is not None:
        I, J = I.to(dtype), J.to(dtype)
    ndims = I.dim() - 2
    assert ndims in (1, 2, 3)
    win = [9] * ndims if win is None else list(win)
    filt = torch.ones([1, 1, *win], dtype=I.dtype, device=I.device)
    pad = math.floor(win[0] / 2)
    conv = getattr(F, "conv%dd" % ndims)

    def box(t):
        return conv(t, filt, stride=1, padding=pad)

    I_sum, J_sum = box(I), box(J)
    I2_sum, J2_sum, IJ_sum = box(I * I), box(J * J), box(I * J)
    n = float(np.prod(win))
    u_I, u_J = I_sum / n, J_sum / n
    cross = IJ_sum - u_J * I_sum - u_I * J_sum + u_I * u_J * n        # :61
    I_var = I2_sum - 2 * u_I * I_sum + u_I * u_I * n                   # :62
    J_var = J2_sum - 2 * u_J * J_sum + u_J * u_J * n                   # :63
    cc = cross * cross / (I_var * J_var + 1e-5)                        # :65
    return -torch.mean(cc)                                             # :67


def mse_loss(y_true, y_pred):
    """`MSE.loss` losses.py:75-76."""
    return torch.mean((y_true - y_pred) ** 2)


def dice_loss(y_true, y_pred):
    """`Dice.loss` losses.py:84-90."""
    axes = list(range(2, y_pred.dim()))
    top = 2 * (y_true * y_pred).sum(dim=axes)
    bottom = torch.clamp((y_true + y_pred).sum(dim=axes), min=1e-5)
    return -torch.mean(top / bottom)


def grad_loss(y_pred, penalty="l1", loss_mult=None):
    """`Grad.loss` losses.py:102-135 (first argument of the reference is ignored)."""
    assert penalty in ("l1", "l2")
    ndims = y_pred.dim() - 2
    per_axis = []
    for i in range(ndims):
        ax = i + 2
        n = y_pred.shape[ax]
        d = y_pred.narrow(ax, 1, n - 1) - y_pred.narrow(ax, 0, n - 1)  # :111
        d = d.abs() if penalty == "l1" else d * d                     # :123-127
        per_axis.append(d.flatten(1).mean(dim=-1))                    # :129
    g = sum(per_axis) / len(per_axis)                                  # :130
    if loss_mult is not None:
        g = g * loss_mult                                              # :132-133
    return g.mean()                                                    # :135


def dice_metric(a, b, labels=None, include_zero=False):
    """`voxelmorph/py/utils.py:265-287` hard Dice per label (numpy)."""
    a, b = np.asarray(a), np.asarray(b)
    if labels is None:
        labels = np.sort(np.unique(np.concatenate([np.unique(a), np.unique(b)])))
    labels = np.asarray(labels)
    if not include_zero:
        labels = labels[labels != 0]
    out = np.zeros(len(labels))
    for k, lab in enumerate(labels):
        top = 2 * np.sum(np.logical_and(a == lab, b == lab))
        bottom = np.sum(a == lab) + np.sum(b == lab)
        out[k] = top / np.maximum(bottom, np.finfo(float).eps)
    return out


# ----------------------------------------------------------------------------
# networks.py (functional, parameters passed as a reference-keyed state dict)
# ----------------------------------------------------------------------------


def default_unet_features():
    """`voxelmorph/py/utils.py:16-21`."""
    return [[16, 32, 32, 32], [32, 32, 32, 32, 32, 16, 16]]


def unet_plan(nb_features=None, nb_levels=None, feat_mult=1, nb_conv_per_level=1):
    """Feature bookkeeping of `Unet.__init__` networks.py:57-77."""
    if nb_features is None:
        nb_features = default_unet_features()
    if isinstance(nb_features, int):
        if nb_levels is None:
            raise ValueError("must provide unet nb_levels if nb_features is an integer")
        feats = np.round(nb_features * feat_mult ** np.arange(nb_levels)).astype(int)
        nb_features = [np.repeat(feats[:-1], nb_conv_per_level),
                       np.repeat(np.flip(feats), nb_conv_per_level)]
    elif nb_levels is not None:
        raise ValueError("cannot use nb_levels if nb_features is not an integer")
    enc_nf, dec_nf = [list(map(int, f)) for f in nb_features]
    n_dec = len(enc_nf)
    return enc_nf, dec_nf[:n_dec], dec_nf[n_dec:], int(n_dec / nb_conv_per_level) + 1


# ---- emulation of bf16 activations / fp32 accumulate (BASELINE.json configs[1]).  Not reference code: the reference has no
# reduced-precision path; this restates what `torch.autocast(bfloat16)` over networks.py:290-305 means arithmetically
# (operands rounded to bf16, products accumulated in fp32 or better, results rounded once) so that the HIP bf16 engine
# can be checked at its own rounding points instead of only against the fp32 result.
_BF16 = [False]


class bf16_activations:
    """Context: conv_block / the flow conv round their input, weights and output activation to bf16 (forward) and the
    gradients w.r.t. their pre-activation and their input to bf16 (backward); biases, accumulation and parameter gradients
    keep the dtype of the surrounding computation."""

    def __enter__(self):
        self.prev = _BF16[0]
        _BF16[0] = True

    def __exit__(self, *exc):
        _BF16[0] = self.prev


def _rbf(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _RoundFwd(torch.autograd.Function):            # round forward, straight-through backward
    @staticmethod
    def forward(ctx, x):
        return _rbf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):            # identity forward, the gradient is rounded
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rbf(g)


def conv_bf16(x, w, b, slope=0.2, round_out=True):
    conv = getattr(F, "conv%dd" % (x.dim() - 2))
    z = conv(_RoundFwd.apply(_RoundBwd.apply(x)), _RoundFwd.apply(w), b, stride=1, padding=1)
    a = F.leaky_relu(_RoundBwd.apply(z), slope) if slope != 1.0 else _RoundBwd.apply(z)
    return _RoundFwd.apply(a) if round_out else a


# Conditioning probe (tests only): the network is piecewise linear; a pre-activation within rounding of zero, or two entries of a
# pooling window within rounding of each other, makes LeakyReLU' / the arg-max -- and with them every upstream gradient -- depend on
# the LAST BITS of the forward pass.  No two fp32 evaluations (different summation orders) agree on such a sample, so a parity test
# has to know when it is looking at one.  `with conditioning() as m:` records, over every ConvBlock / pooling of a forward pass,
# the smallest |pre-activation| and the smallest top-2 gap of a pooling window, each relative to the rms of its tensor.
_MARGIN = [None]


class conditioning:
    def __enter__(self):
        self.margins = {"lrelu": float("inf"), "pool": float("inf")}
        _MARGIN[0] = self.margins
        return self.margins

    def __exit__(self, *exc):
        _MARGIN[0] = None
        return False


def _note_margin(kind, values, ref):
    if _MARGIN[0] is not None:
        rms = float(ref.detach().double().pow(2).mean().sqrt())
        _MARGIN[0][kind] = min(_MARGIN[0][kind], float(values.detach().double().abs().min()) / max(rms, 1e-30))


def conv_block(x, w, b, slope=0.2):
    """`ConvBlock.forward` networks.py:302-305 (k3, stride 1, pad 1, LeakyReLU 0.2)."""
    if _BF16[0]:
        return conv_bf16(x, w, b, slope)
    conv = getattr(F, "conv%dd" % (x.dim() - 2))
    z = conv(x, w, b, stride=1, padding=1)
    _note_margin("lrelu", z, z)
    return F.leaky_relu(z, slope)


def _pool_margin(x, nd):
    if _MARGIN[0] is None:
        return
    xd = x.detach()
    for a in range(nd):                      # windows as a trailing axis of 2^nd entries
        xd = xd.unflatten(2 + a + a, (xd.shape[2 + a + a] // 2, 2))
    win = xd.permute([0, 1] + [2 + 2 * a for a in range(nd)] + [3 + 2 * a for a in range(nd)]).flatten(2 + nd)
    top = win.topk(2, dim=-1).values
    _note_margin("pool", top[..., 0] - top[..., 1], x)


def unet_forward(x, sd, prefix="unet_model.", nb_features=None, nb_levels=None, feat_mult=1,
                 nb_conv_per_level=1, half_res=False, max_pool=2):
    """`Unet.forward` networks.py:122-144 over reference state-dict keys.  `max_pool` (networks.py:79-85): one factor per level (an int is
    repeated); level l pools by max_pool[l] going down and upsamples by max_pool[l] -- the same index -- going up."""
    enc_nf, dec_nf, final_nf, levels = unet_plan(nb_features, nb_levels, feat_mult, nb_conv_per_level)
    nd = x.dim() - 2
    pool = getattr(F, "max_pool%dd" % nd)
    pools = [max_pool] * levels if isinstance(max_pool, int) else list(max_pool)
    hist = [x]
    for lvl in range(levels - 1):                                      # :125-130
        for c in range(nb_conv_per_level):
            k = "%sencoder.%d.%d.main." % (prefix, lvl, c)
            x = conv_block(x, sd[k + "weight"], sd[k + "bias"])
        hist.append(x)
        if pools[lvl] == 2:
            _pool_margin(x, nd)
        x = pool(x, pools[lvl])
    for lvl in range(levels - 1):                                      # :133-138
        for c in range(nb_conv_per_level):
            k = "%sdecoder.%d.%d.main." % (prefix, lvl, c)
            x = conv_block(x, sd[k + "weight"], sd[k + "bias"])
        if not half_res or lvl < (levels - 2):
            k = pools[lvl]
            x = F.interpolate(x, scale_factor=[float(v) for v in k] if isinstance(k, (tuple, list)) else float(k), mode="nearest")
            x = torch.cat([x, hist.pop()], dim=1)
    for n in range(len(final_nf)):                                     # :141-142
        k = "%sremaining.%d.main." % (prefix, n)
        x = conv_block(x, sd[k + "weight"], sd[k + "bias"])
    return x


def vxm_dense_forward(source, target, sd, int_steps=7, int_downsize=2, bidir=False,
                      registration=False, unet_half_res=False, return_all=False, **unet_kwargs):
    """`VxmDense.forward` networks.py:244-287.  return_all: (y_source, y_target, preint_flow, pos_flow) of ONE evaluation (what
    the semi-supervised composition needs: the training outputs AND the positive flow)."""
    nd = source.dim() - 2
    x = unet_forward(torch.cat([source, target], dim=1), sd, half_res=unet_half_res, **unet_kwargs)
    conv = getattr(F, "conv%dd" % nd)
    if _BF16[0]:
        pos = conv_bf16(x, sd["flow.weight"], sd["flow.bias"], slope=1.0, round_out=False)    # the field itself stays fp32
    else:
        pos = conv(x, sd["flow.weight"], sd["flow.bias"], padding=1)   # :257
    if (not unet_half_res) and int_steps > 0 and int_downsize > 1:    # :223,:261
        pos = resize_transform(pos, int_downsize)
    preint = pos                                                       # :264
    neg = -pos if bidir else None
    if int_steps > 0:                                                  # :270-277
        pos = vecint(pos, int_steps)
        neg = vecint(neg, int_steps) if bidir else None
        if int_downsize > 1:
            pos = resize_transform(pos, 1 / int_downsize)
            neg = resize_transform(neg, 1 / int_downsize) if bidir else None
    y_source = spatial_transformer(source, pos)                        # :280
    y_target = spatial_transformer(target, neg) if bidir else None
    if return_all:
        return y_source, y_target, preint, pos
    if not registration:
        return (y_source, y_target, preint) if bidir else (y_source, preint)
    return y_source, pos


def vxm_semisupervised_forward(source, target, seg_src, sd, seg_resolution=2, int_steps=7, int_downsize=2, **unet_kwargs):
    """`VxmDenseSemiSupervisedSeg` (TF backend only in the reference: tf/networks.py:287-388) restated with the torch
    layers of the path: seg_flow = RescaleTransform(1/seg_resolution)(pos_flow) (tf/networks.py:336-337) is
    `ResizeTransform(seg_resolution)` (torch/layers.py:76-97); the down-sampled one-hot source segmentation is warped
    with a linear SpatialTransformer (tf/networks.py:338-339).  Returns (y_source, preint_flow, y_seg_src, pos_flow)."""
    y_source, _, preint, pos = vxm_dense_forward(source, target, sd, int_steps=int_steps, int_downsize=int_downsize, return_all=True,
                                                 **unet_kwargs)
    seg_flow = resize_transform(pos, seg_resolution)
    return y_source, preint, spatial_transformer(seg_src, seg_flow), pos


def state_dict_shapes(inshape, src_feats=1, trg_feats=1, **unet_kwargs):
    """Parameter names/shapes in the reference's state-dict order."""
    nd = len(inshape)
    enc_nf, dec_nf, final_nf, levels = unet_plan(**{k: v for k, v in unet_kwargs.items()
                                                     if k in ("nb_features", "nb_levels", "feat_mult",
                                                              "nb_conv_per_level")})
    per = unet_kwargs.get("nb_conv_per_level", 1)
    half_res = unet_kwargs.get("half_res", False)
    shapes = []
    prev = src_feats + trg_feats
    enc_hist = [prev]
    for lvl in range(levels - 1):
        for c in range(per):
            nf = enc_nf[lvl * per + c]
            shapes.append(("unet_model.encoder.%d.%d.main" % (lvl, c), prev, nf))
            prev = nf
        enc_hist.append(prev)
    enc_hist = enc_hist[::-1]
    for lvl in range(levels - 1):
        for c in range(per):
            nf = dec_nf[lvl * per + c]
            shapes.append(("unet_model.decoder.%d.%d.main" % (lvl, c), prev, nf))
            prev = nf
        if not half_res or lvl < (levels - 2):
            prev += enc_hist[lvl]
    for n, nf in enumerate(final_nf):
        shapes.append(("unet_model.remaining.%d.main" % n, prev, nf))
        prev = nf
    shapes.append(("flow", prev, nd))
    out = []
    for name, cin, cout in shapes:
        out.append((name + ".weight", (cout, cin) + (3,) * nd))
        out.append((name + ".bias", (cout,)))
    return out


def seeded_state_dict(inshape, seed=0, flow_std=1e-5, **kw):
    """Deterministic numpy-seeded parameters (portable across hosts, unlike torch's
    init RNG): conv weights U(-b, b) with b = 1/sqrt(fan_in) like kaiming-uniform(a=sqrt 5)
    (networks.py:299 default init), flow weights N(0, flow_std), flow bias 0 (:214-215)."""
    rng = np.random.default_rng(seed)
    sd = {}
    fan_in = 1
    for name, shape in state_dict_shapes(inshape, **kw):
        if name.endswith(".weight"):
            fan_in = int(np.prod(shape[1:]))          # the bias that follows shares it
        if name.startswith("flow."):
            arr = rng.standard_normal(shape) * flow_std if name.endswith("weight") else np.zeros(shape)
        else:
            bound = 1.0 / math.sqrt(fan_in)
            arr = rng.uniform(-bound, bound, size=shape)
        sd[name] = torch.from_numpy(arr.astype(np.float32))
    return sd


# ----------------------------------------------------------------------------
# Explicit (non-ATen) restatements — SURVEY.md Appendix B
# ----------------------------------------------------------------------------


def _src_coords_explicit(flow):
    """Per-axis source coordinate with the reference+ATen op order, all fp32:
    loc = i + f; c = 2*(loc/(S-1) - 0.5); x = ((c+1)/2)*(S-1)."""
    f32 = np.float32
    flow = np.asarray(flow, dtype=f32)
    B, nd = flow.shape[:2]
    size = flow.shape[2:]
    xs = []
    for a, S in enumerate(size):
        shp = [1] * nd
        shp[a] = S
        idx = np.arange(S, dtype=f32).reshape(shp)
        loc = (idx + flow[:, a]).astype(f32)
        c = (f32(2) * ((loc / f32(S - 1)).astype(f32) - f32(0.5)).astype(f32)).astype(f32)
        x = ((((c + f32(1)).astype(f32)) / f32(2)).astype(f32) * f32(S - 1)).astype(f32)
        xs.append(x)
    return xs


def warp_explicit(src, flow, mode="bilinear"):
    """numpy fp32 restatement of layers.py:30-48 + ATen grid_sampler_3d
    (align_corners=True, padding zeros).  Nearest = round-half-even."""
    f32 = np.float32
    src = np.asarray(src, dtype=f32)
    B, C = src.shape[:2]
    size = src.shape[2:]
    nd = len(size)
    xs = _src_coords_explicit(flow)
    out = np.zeros((B, C) + tuple(flow.shape[2:]), dtype=f32)
    bidx = np.arange(B).reshape((B,) + (1,) * nd)
    if mode == "nearest":
        js = [np.rint(x).astype(np.int64) for x in xs]
        ok = np.ones(js[0].shape, dtype=bool)
        for j, S in zip(js, size):
            ok &= (j >= 0) & (j < S)
        jc = [np.clip(j, 0, S - 1) for j, S in zip(js, size)]
        for c in range(C):
            g = src[:, c][(np.broadcast_to(bidx, jc[0].shape),) + tuple(jc)]
            out[:, c] = np.where(ok, g, f32(0))
        return out
    x0 = [np.floor(x) for x in xs]
    for corner in range(2 ** nd):
        w = np.ones(xs[0].shape, dtype=f32)
        ok = np.ones(xs[0].shape, dtype=bool)
        jc = []
        for a in range(nd):
            bit = (corner >> (nd - 1 - a)) & 1
            ca = x0[a] + f32(bit)
            w = (w * (f32(1) - np.abs(xs[a] - ca).astype(f32))).astype(f32)
            j = ca.astype(np.int64)
            ok &= (j >= 0) & (j < size[a])
            jc.append(np.clip(j, 0, size[a] - 1))
        for c in range(C):
            g = src[:, c][(np.broadcast_to(bidx, jc[0].shape),) + tuple(jc)]
            out[:, c] += np.where(ok, g * w, f32(0)).astype(f32)
    return out


def resize_explicit(x, vel_resize):
    """Separable-lerp restatement of layers.py:85-97 / ATen upsample_trilinear3d
    (align_corners=True): out = floor(in*factor); src = dst*(in-1)/(out-1)."""
    f32 = np.float32
    x = np.asarray(x, dtype=f32)
    factor = 1.0 / vel_resize
    if factor == 1:
        return x
    if factor > 1:
        x = (f32(factor) * x).astype(f32)
    nd = x.ndim - 2
    for a in range(nd):
        ax = a + 2
        n_in = x.shape[ax]
        n_out = int(math.floor(n_in * factor))
        scale = f32(n_in - 1) / f32(n_out - 1) if n_out > 1 else f32(0)
        pos = (scale * np.arange(n_out, dtype=f32)).astype(f32)
        i0 = np.minimum(np.floor(pos).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (pos - i0.astype(f32)).astype(f32)
        l0 = (f32(1) - l1).astype(f32)
        shp = [1] * x.ndim
        shp[ax] = n_out
        x = (np.take(x, i0, axis=ax) * l0.reshape(shp) + np.take(x, i1, axis=ax) * l1.reshape(shp)).astype(f32)
    if factor < 1:
        x = (f32(factor) * x).astype(f32)
    return x


def boxsum_explicit(x, win=9):
    """Zero-padded box sum over the trailing 3 axes in fp64 (NCC arbiter helper)."""
    x = np.asarray(x, dtype=np.float64)
    r = win // 2
    for ax in (-3, -2, -1):
        n = x.shape[ax]
        pad = [(0, 0)] * x.ndim
        pad[ax] = (r + 1, r)
        cs = np.cumsum(np.pad(x, pad), axis=ax)
        hi = np.take(cs, np.arange(win, win + n), axis=ax)
        lo = np.take(cs, np.arange(0, n), axis=ax)
        x = hi - lo
    return x


def ncc_explicit(I, J, win=9):
    """fp64 separable-sum restatement of losses.py:47-67 (arbiter)."""
    I = np.asarray(I, dtype=np.float64)
    J = np.asarray(J, dtype=np.float64)
    n = float(win ** 3)
    Is, Js = boxsum_explicit(I, win), boxsum_explicit(J, win)
    I2, J2, IJ = boxsum_explicit(I * I, win), boxsum_explicit(J * J, win), boxsum_explicit(I * J, win)
    uI, uJ = Is / n, Js / n
    cross = IJ - uJ * Is - uI * Js + uI * uJ * n
    Iv = I2 - 2 * uI * Is + uI * uI * n
    Jv = J2 - 2 * uJ * Js + uJ * uJ * n
    return -np.mean(cross * cross / (Iv * Jv + 1e-5))


def _box_axis_explicit(x, ax, w, p, adjoint=False):
    """One axis of the reference's box filter in fp64: `conv(ones[w], stride 1, padding p)` (losses.py:47-55) takes an axis of S entries
    to O = S + 2p - w + 1 sums, out[o] = sum_k in[o - p + k]; `adjoint=True` applies its transpose (O entries back to S)."""
    S = x.shape[ax]
    if adjoint:                        # x has O entries; S_in of the forward filter is O - 2p + w - 1
        O, S = S, S - 2 * p + w - 1
        out_shape = list(x.shape)
        out_shape[ax] = S
        out = np.zeros(out_shape, dtype=np.float64)
        for o in range(O):
            lo, hi = max(0, o - p), min(S, o - p + w)
            if hi > lo:
                sl = [slice(None)] * x.ndim
                sl[ax] = slice(lo, hi)
                out[tuple(sl)] += np.take(x, [o], axis=ax)
        return out
    O = S + 2 * p - w + 1
    if O <= 0:
        raise ValueError("window %d larger than the padded axis (%d + 2*%d)" % (w, S, p))
    pad = [(0, 0)] * x.ndim
    pad[ax] = (p + 1, p)
    cs = np.cumsum(np.pad(x, pad), axis=ax)
    return np.take(cs, np.arange(w, w + O), axis=ax) - np.take(cs, np.arange(0, O), axis=ax)


def ncc_explicit_win(I, J, win, grad=False):
    """fp64 separable restatement of losses.py:26-67 for ANY window: every axis padded by win[0] // 2 (:31-36), box sums of extent
    S + 2 pad - win + 1 per axis, cc and its mean on that shape.  With `grad=True` also returns dL/dJ and dL/dI through the
    transposed box filters (the arbiter of the HIP kernels' separable passes)."""
    I = np.asarray(I, dtype=np.float64)
    J = np.asarray(J, dtype=np.float64)
    nd = I.ndim - 2
    win = [int(w) for w in win]
    assert len(win) == nd
    pad = win[0] // 2
    axes = list(range(2, 2 + nd))

    def box(x):
        for ax, w in zip(axes, win):
            x = _box_axis_explicit(x, ax, w, pad)
        return x

    def box_t(x):
        for ax, w in zip(axes, win):
            x = _box_axis_explicit(x, ax, w, pad, adjoint=True)
        return x

    n = float(np.prod(win))
    Is, Js, I2, J2, IJ = box(I), box(J), box(I * I), box(J * J), box(I * J)
    uI, uJ = Is / n, Js / n
    cross = IJ - uJ * Is - uI * Js + uI * uJ * n
    Iv = I2 - 2 * uI * Is + uI * uI * n
    Jv = J2 - 2 * uJ * Js + uJ * uJ * n
    den = Iv * Jv + 1e-5
    loss = -np.mean(cross * cross / den)
    if not grad:
        return loss
    N = cross.size
    t = cross / den

    def side(Xs, Ys, Xv, X, Y):          # d(-mean cc)/dY with cross = IJ - Xs Ys / n, Yv = Y2 - Ys^2 / n
        a = 2 * t * (-Xs / n) + t * t * Xv * (2 * Ys / n)
        b = -(t * t) * Xv
        c = 2 * t
        return -(box_t(a) + 2 * Y * box_t(b) + X * box_t(c)) / N

    return loss, side(Is, Js, Iv, I, J), side(Js, Is, Jv, J, I)


# ----------------------------------------------------------------------------
# Training step (the north-star path; scripts/torch/train.py:194-223)
# ----------------------------------------------------------------------------


def train_step_loss(source, target, sd, image_loss="ncc", lam=1.0, int_steps=7, int_downsize=2,
                    **kw):
    """Loss of one train.py step: image loss on (target, y_source) + lam * Grad('l2',
    loss_mult=int_downsize) on preint_flow (train.py:164-181,207-212)."""
    y_source, preint = vxm_dense_forward(source, target, sd, int_steps=int_steps,
                                         int_downsize=int_downsize, **kw)
    img = ncc_loss(target, y_source) if image_loss == "ncc" else mse_loss(target, y_source)
    reg = grad_loss(preint, "l2", loss_mult=int_downsize)
    return img + lam * reg, (img, reg, y_source, preint)


def adam_step_explicit(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (train.py:161, defaults) single-tensor update, fp32."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v
