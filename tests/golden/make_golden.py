#!/usr/bin/env python
"""Generate tests/golden/*.npz from the UNMODIFIED reference torch backend.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Every array stored is either a seeded input or an output/gradient computed by the
reference's own modules (`voxelmorph.torch.layers/losses/networks`) on CPU,
torch version recorded in `meta`.  Parameters of the network cases come from
`oracle.vxm_oracle.seeded_state_dict` (numpy-seeded, so they can be rebuilt
anywhere without storing 1.3 MB of weights).
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_loader            # noqa: E402
from oracle import vxm_oracle as orc     # noqa: E402

warnings.filterwarnings("ignore")
vxm = ref_loader.load_reference()
L, N, LS = vxm.layers, vxm.networks, vxm.losses


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def smooth_flow(rng, shape, amp, nd=3):
    """Low-frequency + small noise displacement field [B,nd,*shape], |v| up to ~amp."""
    B = shape[0]
    vol = shape[1:]
    axes = np.meshgrid(*[np.linspace(0, 1, s) for s in vol], indexing="ij")
    out = np.zeros((B, nd) + tuple(vol), dtype=np.float32)
    for b in range(B):
        for c in range(nd):
            ph = rng.uniform(0, 2 * np.pi, size=nd)
            fr = rng.uniform(0.5, 2.0, size=nd)
            f = sum(np.sin(2 * np.pi * fr[a] * axes[a] + ph[a]) for a in range(nd)) / nd
            out[b, c] = amp * f + 0.05 * amp * rng.standard_normal(vol)
    return out


def save(name, **arrs):
    meta = "torch=%s numpy=%s reference=voxelmorph@0.2" % (torch.__version__, np.__version__)
    np.savez_compressed(os.path.join(HERE, name), meta=np.array(meta), **arrs)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrs.items()})


def gold_layers():
    rng = np.random.default_rng(11)
    vol = (10, 12, 14)
    # --- trilinear warp, C=2, flow large enough to leave the volume at the borders
    src = rng.random((2, 2) + vol).astype(np.float32)
    flow = smooth_flow(rng, (2,) + vol, amp=3.0)
    st = L.SpatialTransformer(vol)
    s, f = t(src).requires_grad_(), t(flow).requires_grad_()
    out = st(s, f)
    gout = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(t(gout))
    # --- nearest warp on tie (half-integer) flows: the bit-exact case
    seg = rng.integers(0, 30, size=(1, 1) + vol).astype(np.float32)
    tie = (rng.integers(-4, 5, size=(1, 3) + vol) * 0.5).astype(np.float32)
    stn = L.SpatialTransformer(vol, mode="nearest")
    near = stn(t(seg), t(tie))
    near_id = stn(t(seg), torch.zeros(1, 3, *vol))
    # --- 2-D bilinear
    vol2 = (12, 10)
    src2 = rng.random((1, 2) + vol2).astype(np.float32)
    flow2 = smooth_flow(rng, (1,) + vol2, amp=2.0, nd=2)
    out2 = L.SpatialTransformer(vol2)(t(src2), t(flow2))
    # --- VecInt 7 steps
    vec = smooth_flow(rng, (1,) + vol, amp=2.5)
    vi = L.VecInt(vol, 7)
    v = t(vec).requires_grad_()
    iv = vi(v)
    giv = rng.standard_normal(iv.shape).astype(np.float32)
    iv.backward(t(giv))
    # --- ResizeTransform down (vel_resize=2) and up (vel_resize=1/2)
    volr = (8, 12, 10)
    xr = rng.standard_normal((1, 3) + volr).astype(np.float32)
    xd = t(xr).requires_grad_()
    down = L.ResizeTransform(2, 3)(xd)
    gdown = rng.standard_normal(down.shape).astype(np.float32)
    down.backward(t(gdown))
    xu = t(xr).requires_grad_()
    up = L.ResizeTransform(0.5, 3)(xu)
    gup = rng.standard_normal(up.shape).astype(np.float32)
    up.backward(t(gup))
    save("layers.npz",
         warp_src=src, warp_flow=flow, warp_out=out.detach().numpy(), warp_gout=gout,
         warp_gsrc=s.grad.numpy(), warp_gflow=f.grad.numpy(),
         near_seg=seg, near_flow=tie, near_out=near.numpy(), near_identity=near_id.numpy(),
         warp2_src=src2, warp2_flow=flow2, warp2_out=out2.numpy(),
         vecint_in=vec, vecint_out=iv.detach().numpy(), vecint_gout=giv, vecint_gin=v.grad.numpy(),
         resize_in=xr, resize_down=down.detach().numpy(), resize_gdown=gdown,
         resize_down_gin=xd.grad.numpy(), resize_up=up.detach().numpy(), resize_gup=gup,
         resize_up_gin=xu.grad.numpy())


def gold_losses():
    rng = np.random.default_rng(12)
    vol = (12, 14, 16)
    I = rng.random((2, 1) + vol).astype(np.float32)
    J = (0.6 * I + 0.4 * rng.random((2, 1) + vol)).astype(np.float32)
    with ref_loader.cuda_alias_to_cpu():
        Jt = t(J).requires_grad_()
        ncc = LS.NCC().loss(t(I), Jt)
        ncc.backward()
        Jt5 = t(J).requires_grad_()
        ncc5 = LS.NCC(win=[5, 5, 5]).loss(t(I), Jt5)
        ncc5.backward()
    Jm = t(J).requires_grad_()
    mse = LS.MSE().loss(t(I), Jm)
    mse.backward()
    flow = smooth_flow(rng, (2,) + vol, amp=1.5)
    res = {}
    for pen, mult in (("l1", None), ("l2", 2)):
        fl = t(flow).requires_grad_()
        g = LS.Grad(pen, loss_mult=mult).loss(None, fl)
        g.backward()
        res["grad_%s" % pen] = g.detach().numpy()
        res["grad_%s_g" % pen] = fl.grad.numpy()
    lab = rng.integers(0, 5, size=(2,) + vol)
    yt = np.stack([(lab == k) for k in range(5)], axis=1).astype(np.float32)
    yp = rng.random((2, 5) + vol).astype(np.float32)
    yp[:, 4] = 0
    yt[:, 4] = 0                       # an empty channel exercises the clamp(min=1e-5)
    ypt = t(yp).requires_grad_()
    dice = LS.Dice().loss(t(yt), ypt)
    dice.backward()
    save("losses.npz", I=I, J=J, ncc=ncc.detach().numpy(), ncc_gJ=Jt.grad.numpy(),
         ncc5=ncc5.detach().numpy(), ncc5_gJ=Jt5.grad.numpy(),
         mse=mse.detach().numpy(), mse_gJ=Jm.grad.numpy(), flow=flow,
         dice_true=yt, dice_pred=yp, dice=dice.detach().numpy(), dice_g=ypt.grad.numpy(), **res)


KEEP_FULL = ["flow.weight", "flow.bias", "unet_model.encoder.0.0.main.weight",
             "unet_model.decoder.3.0.main.bias", "unet_model.remaining.2.main.weight",
             "unet_model.encoder.2.0.main.bias"]


def gold_network():
    inshape = (16, 32, 16)
    rng = np.random.default_rng(13)
    src = rng.random((2, 1) + inshape).astype(np.float32)
    trg = rng.random((2, 1) + inshape).astype(np.float32)
    out = dict(source=src, target=trg, inshape=np.array(inshape))
    cases = {
        "diffeo": dict(int_steps=7, int_downsize=2, bidir=False, loss="ncc", lam=1.0),
        "dense": dict(int_steps=0, int_downsize=2, bidir=False, loss="mse", lam=0.01),
        "bidir": dict(int_steps=3, int_downsize=2, bidir=True, loss="mse", lam=0.01),
    }
    for tag, cfg in cases.items():
        sd = orc.seeded_state_dict(inshape, seed=5, flow_std=0.2)
        model = N.VxmDense(inshape, int_steps=cfg["int_steps"], int_downsize=cfg["int_downsize"],
                           bidir=cfg["bidir"])
        missing = model.load_state_dict(sd, strict=False)
        assert all(k.endswith(".grid") for k in missing.missing_keys), missing
        assert not missing.unexpected_keys
        model.train()
        pred = model(t(src), t(trg))
        with ref_loader.cuda_alias_to_cpu():
            img_fn = LS.NCC().loss if cfg["loss"] == "ncc" else LS.MSE().loss
            if cfg["bidir"]:
                img = 0.5 * img_fn(t(trg), pred[0]) + 0.5 * img_fn(t(src), pred[1])
            else:
                img = img_fn(t(trg), pred[0])
            reg = LS.Grad("l2", loss_mult=cfg["int_downsize"]).loss(None, pred[-1])
            loss = img + cfg["lam"] * reg
            loss.backward()
        out[tag + "_y_source"] = pred[0].detach().numpy()
        if cfg["bidir"]:
            out[tag + "_y_target"] = pred[1].detach().numpy()
        out[tag + "_preint"] = pred[-1].detach().numpy()
        out[tag + "_loss"] = np.array([loss.item(), img.item(), reg.item()])
        names, norms = [], []
        for k, p in model.named_parameters():
            names.append(k)
            norms.append(float(p.grad.double().norm()))
            if k in KEEP_FULL:
                out[tag + "_grad_" + k] = p.grad.numpy().copy()
        out[tag + "_grad_names"] = np.array(names)
        out[tag + "_grad_norms"] = np.array(norms)
        with torch.no_grad():
            ys, pos = model(t(src), t(trg), registration=True)
        out[tag + "_pos_flow"] = pos.numpy()
        # one Adam step (train.py:161,218-220) -> post-step flow weights
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        opt.step()
        out[tag + "_flow_weight_after_adam"] = model.flow.weight.detach().numpy().copy()
        out[tag + "_enc0_weight_after_adam"] = model.unet_model.encoder[0][0].main.weight.detach().numpy().copy()
    # state-dict key order / shapes of the default model (checkpoint interop, modelio.py:58-77)
    m = N.VxmDense((16, 16, 16))
    keys = [k for k in m.state_dict().keys()]
    out["state_keys"] = np.array(keys)
    out["state_shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
    out["n_params"] = np.array(sum(p.numel() for p in m.parameters()))
    save("network.npz", **out)


def gold_dice_metric():
    rng = np.random.default_rng(14)
    a = rng.integers(0, 6, size=(8, 9, 10))
    b = np.where(rng.random(a.shape) < 0.7, a, rng.integers(0, 6, size=a.shape))
    d = vxm.py.utils.dice(a, b, labels=[1, 2, 3, 5])
    save("dice_metric.npz", a=a, b=b, dice=d)


def gold_planar():
    """2-D cases of the same classes (the reference is N-D generic): layers, losses and a small VxmDense."""
    rng = np.random.default_rng(21)
    img = (24, 20)
    out = {}
    # --- bilinear warp, C=2, flow leaving the image at the borders
    src = rng.random((2, 2) + img).astype(np.float32)
    flow = smooth_flow(rng, (2,) + img, amp=3.0, nd=2)
    s, f = t(src).requires_grad_(), t(flow).requires_grad_()
    w = L.SpatialTransformer(img)(s, f)
    gout = rng.standard_normal(w.shape).astype(np.float32)
    w.backward(t(gout))
    out.update(warp_src=src, warp_flow=flow, warp_out=w.detach().numpy(), warp_gout=gout, warp_gsrc=s.grad.numpy(),
               warp_gflow=f.grad.numpy())
    # --- nearest warp on tie flows (bit-exact case)
    seg = rng.integers(0, 30, size=(1, 1) + img).astype(np.float32)
    tie = (rng.integers(-4, 5, size=(1, 2) + img) * 0.5).astype(np.float32)
    out.update(near_seg=seg, near_flow=tie, near_out=L.SpatialTransformer(img, mode="nearest")(t(seg), t(tie)).numpy())
    # --- VecInt 5 steps
    vec = smooth_flow(rng, (2,) + img, amp=2.5, nd=2)
    v = t(vec).requires_grad_()
    iv = L.VecInt(img, 5)(v)
    giv = rng.standard_normal(iv.shape).astype(np.float32)
    iv.backward(t(giv))
    out.update(vecint_in=vec, vecint_out=iv.detach().numpy(), vecint_gout=giv, vecint_gin=v.grad.numpy())
    # --- ResizeTransform down / up
    xr = rng.standard_normal((2, 2) + img).astype(np.float32)
    xd = t(xr).requires_grad_()
    down = L.ResizeTransform(2, 2)(xd)
    gdown = rng.standard_normal(down.shape).astype(np.float32)
    down.backward(t(gdown))
    xu = t(xr).requires_grad_()
    up = L.ResizeTransform(0.5, 2)(xu)
    gup = rng.standard_normal(up.shape).astype(np.float32)
    up.backward(t(gup))
    out.update(resize_in=xr, resize_down=down.detach().numpy(), resize_gdown=gdown, resize_down_gin=xd.grad.numpy(),
               resize_up=up.detach().numpy(), resize_gup=gup, resize_up_gin=xu.grad.numpy())
    # --- losses
    I = rng.random((2, 1) + img).astype(np.float32)
    J = (0.6 * I + 0.4 * rng.random((2, 1) + img)).astype(np.float32)
    with ref_loader.cuda_alias_to_cpu():
        for tag, win in (("ncc", None), ("ncc5", [5, 5])):
            Jt = t(J).requires_grad_()
            ncc = LS.NCC(win=win).loss(t(I), Jt)
            ncc.backward()
            out[tag] = ncc.detach().numpy()
            out[tag + "_gJ"] = Jt.grad.numpy()
    for pen, mult in (("l1", None), ("l2", 2)):
        fl = t(flow).requires_grad_()
        g = LS.Grad(pen, loss_mult=mult).loss(None, fl)
        g.backward()
        out["grad_%s" % pen] = g.detach().numpy()
        out["grad_%s_g" % pen] = fl.grad.numpy()
    out.update(I=I, J=J)
    # --- VxmDense on 32 x 48 images
    inshape = (32, 48)
    srcn = rng.random((2, 1) + inshape).astype(np.float32)
    trgn = rng.random((2, 1) + inshape).astype(np.float32)
    out.update(source=srcn, target=trgn, inshape=np.array(inshape))
    cases = {
        "diffeo": dict(int_steps=5, int_downsize=2, bidir=False, loss="ncc", lam=1.0),
        "dense": dict(int_steps=0, int_downsize=2, bidir=False, loss="mse", lam=0.01),
        "bidir": dict(int_steps=3, int_downsize=2, bidir=True, loss="mse", lam=0.01),
    }
    for tag, cfg in cases.items():
        sd = orc.seeded_state_dict(inshape, seed=7, flow_std=0.2)
        model = N.VxmDense(inshape, int_steps=cfg["int_steps"], int_downsize=cfg["int_downsize"], bidir=cfg["bidir"])
        missing = model.load_state_dict(sd, strict=False)
        assert all(k.endswith(".grid") for k in missing.missing_keys) and not missing.unexpected_keys, missing
        model.train()
        pred = model(t(srcn), t(trgn))
        with ref_loader.cuda_alias_to_cpu():
            img_fn = LS.NCC().loss if cfg["loss"] == "ncc" else LS.MSE().loss
            if cfg["bidir"]:
                imgl = 0.5 * img_fn(t(trgn), pred[0]) + 0.5 * img_fn(t(srcn), pred[1])
            else:
                imgl = img_fn(t(trgn), pred[0])
            reg = LS.Grad("l2", loss_mult=cfg["int_downsize"]).loss(None, pred[-1])
            loss = imgl + cfg["lam"] * reg
            loss.backward()
        out[tag + "_y_source"] = pred[0].detach().numpy()
        if cfg["bidir"]:
            out[tag + "_y_target"] = pred[1].detach().numpy()
        out[tag + "_preint"] = pred[-1].detach().numpy()
        out[tag + "_loss"] = np.array([loss.item(), imgl.item(), reg.item()])
        names, norms = [], []
        for k, p in model.named_parameters():
            names.append(k)
            norms.append(float(p.grad.double().norm()))
            if k in KEEP_FULL:
                out[tag + "_grad_" + k] = p.grad.numpy().copy()
        out[tag + "_grad_names"] = np.array(names)
        out[tag + "_grad_norms"] = np.array(norms)
        with torch.no_grad():
            _, pos = model(t(srcn), t(trgn), registration=True)
        out[tag + "_pos_flow"] = pos.numpy()
    m = N.VxmDense((16, 16))
    out["state_keys"] = np.array(list(m.state_dict().keys()))
    out["state_shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
    save("planar.npz", **out)


def gold_ncc_windows():
    """NCC with windows that are not odd and of one size per axis (losses.py:26-36: every axis is padded by win[0] // 2, so the box sums
    change shape), from the unmodified reference: value, gradient onto y_pred, and for one case onto y_true."""
    rng = np.random.default_rng(21)
    out = {}
    vol = (10, 13, 15)
    I3 = rng.random((2, 1) + vol).astype(np.float32)
    J3 = (0.6 * I3 + 0.4 * rng.random((2, 1) + vol)).astype(np.float32)
    img = (14, 19)
    I2 = rng.random((2, 1) + img).astype(np.float32)
    J2 = (0.6 * I2 + 0.4 * rng.random((2, 1) + img)).astype(np.float32)
    I1 = rng.random((3, 1, 37)).astype(np.float32)
    J1 = (0.6 * I1 + 0.4 * rng.random((3, 1, 37))).astype(np.float32)
    out.update(I3=I3, J3=J3, I2=I2, J2=J2, I1=I1, J1=J1)
    cases = [("w995", I3, J3, [9, 9, 5]), ("w579", I3, J3, [5, 7, 9]), ("w444", I3, J3, [4, 4, 4]), ("w357", I3, J3, [3, 5, 7]),
             ("w73", I2, J2, [7, 3]), ("w66", I2, J2, [6, 6]), ("w4", I1, J1, [4])]
    with ref_loader.cuda_alias_to_cpu():
        for tag, I, J, win in cases:
            It, Jt = t(I).requires_grad_(), t(J).requires_grad_()
            ncc = LS.NCC(win=win).loss(It, Jt)
            ncc.backward()
            out[tag] = ncc.detach().numpy()
            out[tag + "_gJ"] = Jt.grad.numpy()
            out[tag + "_gI"] = It.grad.numpy()
            out[tag + "_win"] = np.array(win)
    out["cases"] = np.array([c[0] for c in cases])
    save("ncc_windows.npz", **out)


UNET_POOL_CASES = {
    # tag: (inshape, infeats, nb_features, max_pool)
    "p3_vol": ((18, 9, 18), 2, [[8, 8], [8, 8, 8]], 3),
    "aniso_vol": ((4, 16, 12), 2, [[8, 8], [8, 8, 8]], [(1, 2, 2), (1, 2, 2), (1, 2, 2)]),
    "p3_img": ((27, 18), 1, [[8, 8], [8, 8, 8]], 3),
    "p42_img": ((16, 24), 2, [[8], [8, 8]], [4, 2]),
}


def gold_unet_pools():
    """Unet(max_pool = k != 2, per-level lists, per-axis tuples) of the unmodified reference (networks.py:79-85,122-144): output and the
    gradients of sum(y * r) onto the input and every parameter (norms; the first encoder weight in full)."""
    out = {}
    for tag, (inshape, infeats, feats, pool) in UNET_POOL_CASES.items():
        rng = np.random.default_rng({"p3_vol": 31, "aniso_vol": 32, "p3_img": 33, "p42_img": 34}[tag])
        torch.manual_seed(40 + len(tag))
        model = N.Unet(inshape=inshape, infeats=infeats, nb_features=feats, max_pool=pool)
        x = rng.standard_normal((2, infeats) + inshape).astype(np.float32)
        xt = t(x).requires_grad_()
        y = model(xt)
        r = rng.standard_normal(tuple(y.shape)).astype(np.float32)
        (y * t(r)).sum().backward()
        out[tag + "_x"] = x
        out[tag + "_r"] = r
        out[tag + "_y"] = y.detach().numpy()
        out[tag + "_gx"] = xt.grad.numpy()
        names, norms = [], []
        for k, p in model.named_parameters():
            names.append(k)
            norms.append(float(p.grad.double().norm()))
            out[tag + "_param_" + k] = p.detach().numpy().copy()
        out[tag + "_grad_names"] = np.array(names)
        out[tag + "_grad_norms"] = np.array(norms)
        out[tag + "_gw_enc0"] = model.encoder[0][0].main.weight.grad.numpy().copy()
    save("unet_pools.npz", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = sys.argv[1:]          # e.g. `make_golden.py planar` regenerates one file and leaves the others untouched
    for name, fn in (("layers", gold_layers), ("losses", gold_losses), ("network", gold_network),
                     ("dice_metric", gold_dice_metric), ("planar", gold_planar), ("ncc_windows", gold_ncc_windows),
                     ("unet_pools", gold_unet_pools)):
        if not only or name in only:
            fn()
