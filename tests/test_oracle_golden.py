"""Pin the oracle (oracle/vxm_oracle.py + oracle/vxm_oracle.c) to fixtures produced by the
unmodified reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import c_oracle, ref_loader
from oracle import vxm_oracle as orc


def T(a):
    return torch.from_numpy(np.asarray(a))


# ------------------------------------------------------------------ layers
def test_warp_trilinear_matches_reference(g_layers):
    s, f = T(g_layers["warp_src"]).requires_grad_(), T(g_layers["warp_flow"]).requires_grad_()
    out = orc.spatial_transformer(s, f)
    assert torch.equal(out.detach(), T(g_layers["warp_out"]))          # same ATen calls: bit-exact
    out.backward(T(g_layers["warp_gout"]))
    np.testing.assert_allclose(s.grad.numpy(), g_layers["warp_gsrc"], atol=1e-6)
    np.testing.assert_allclose(f.grad.numpy(), g_layers["warp_gflow"], atol=1e-5)


def test_warp_explicit_and_c(g_layers):
    for fn in (orc.warp_explicit, c_oracle.warp3d):
        out = fn(g_layers["warp_src"], g_layers["warp_flow"])
        np.testing.assert_allclose(out, g_layers["warp_out"], atol=2e-6, rtol=0)


def test_warp_nearest_bit_exact(g_layers):
    for fn in (orc.warp_explicit, c_oracle.warp3d):
        out = fn(g_layers["near_seg"], g_layers["near_flow"], mode="nearest")
        assert np.array_equal(out, g_layers["near_out"]), fn
        ident = fn(g_layers["near_seg"], np.zeros_like(g_layers["near_flow"]), mode="nearest")
        assert np.array_equal(ident, g_layers["near_identity"])
        assert np.array_equal(ident, g_layers["near_seg"])             # index grid is exact
    out = orc.spatial_transformer(T(g_layers["near_seg"]), T(g_layers["near_flow"]), mode="nearest")
    assert np.array_equal(out.numpy(), g_layers["near_out"])


def test_warp_2d(g_layers):
    out = orc.spatial_transformer(T(g_layers["warp2_src"]), T(g_layers["warp2_flow"]))
    assert torch.equal(out, T(g_layers["warp2_out"]))
    np.testing.assert_allclose(orc.warp_explicit(g_layers["warp2_src"], g_layers["warp2_flow"]),
                               g_layers["warp2_out"], atol=2e-6)


def test_identity_grid_is_exact_integers():
    g = orc.identity_grid((5, 6, 7))
    assert g.dtype == torch.float32 and g.shape == (1, 3, 5, 6, 7)
    assert g[0, 0, 3, 1, 2] == 3 and g[0, 1, 3, 1, 2] == 1 and g[0, 2, 3, 1, 2] == 2


def test_vecint(g_layers):
    v = T(g_layers["vecint_in"]).requires_grad_()
    out = orc.vecint(v, 7)
    assert torch.equal(out.detach(), T(g_layers["vecint_out"]))
    out.backward(T(g_layers["vecint_gout"]))
    np.testing.assert_allclose(v.grad.numpy(), g_layers["vecint_gin"], atol=1e-5)
    np.testing.assert_allclose(c_oracle.vecint3d(g_layers["vecint_in"], 7), g_layers["vecint_out"], atol=1e-4)
    with pytest.raises(AssertionError):
        orc.vecint(v, -1)


def test_resize(g_layers):
    x = T(g_layers["resize_in"]).requires_grad_()
    down = orc.resize_transform(x, 2)
    assert torch.equal(down.detach(), T(g_layers["resize_down"]))
    down.backward(T(g_layers["resize_gdown"]))
    np.testing.assert_allclose(x.grad.numpy(), g_layers["resize_down_gin"], atol=1e-6)
    x2 = T(g_layers["resize_in"]).requires_grad_()
    up = orc.resize_transform(x2, 0.5)
    assert torch.equal(up.detach(), T(g_layers["resize_up"]))
    up.backward(T(g_layers["resize_gup"]))
    np.testing.assert_allclose(x2.grad.numpy(), g_layers["resize_up_gin"], atol=1e-5)
    for fn in (orc.resize_explicit, c_oracle.resize3d):
        np.testing.assert_allclose(fn(g_layers["resize_in"], 2), g_layers["resize_down"], atol=2e-6)
        np.testing.assert_allclose(fn(g_layers["resize_in"], 0.5), g_layers["resize_up"], atol=2e-6)
    assert orc.resize_transform(x, 1) is x


# ------------------------------------------------------------------ losses
def test_ncc(g_losses):
    I, J = T(g_losses["I"]), T(g_losses["J"]).requires_grad_()
    l = orc.ncc_loss(I, J)
    np.testing.assert_allclose(l.item(), g_losses["ncc"], rtol=1e-6)
    l.backward()
    np.testing.assert_allclose(J.grad.numpy(), g_losses["ncc_gJ"], atol=1e-7)
    l5 = orc.ncc_loss(I, T(g_losses["J"]), win=[5, 5, 5])
    np.testing.assert_allclose(l5.item(), g_losses["ncc5"], rtol=1e-6)
    # fp64 arbiters (python and C) agree with each other tightly, and with the fp32 reference loosely
    a = orc.ncc_explicit(g_losses["I"], g_losses["J"])
    c = c_oracle.ncc_loss(g_losses["I"], g_losses["J"])
    d = orc.ncc_loss(I, T(g_losses["J"]), dtype=torch.float64).item()
    assert abs(a - c) < 1e-12 and abs(a - d) < 1e-12
    assert abs(a - float(g_losses["ncc"])) < 1e-3


def test_ncc_any_window_matches_reference(g_nccwin):
    """Windows that are not odd and of one size (losses.py:26-36 pads every axis by win[0] // 2): the torch oracle is the reference's
    call sequence (same ATen calls: tight), the explicit fp64 restatement carries the shape rule and the transposed filters the HIP
    kernels implement."""
    for tag in g_nccwin["cases"]:
        tag = str(tag)
        nd = len(g_nccwin[tag + "_win"])
        win = [int(w) for w in g_nccwin[tag + "_win"]]
        Ia, Ja = g_nccwin["I%d" % nd], g_nccwin["J%d" % nd]
        I, J = T(Ia).requires_grad_(), T(Ja).requires_grad_()
        l = orc.ncc_loss(I, J, win=win)
        np.testing.assert_allclose(l.item(), g_nccwin[tag], rtol=1e-6, err_msg=tag)
        l.backward()
        np.testing.assert_allclose(J.grad.numpy(), g_nccwin[tag + "_gJ"], atol=5e-7, rtol=1e-5, err_msg=tag)   # (thread-count dependent sums)
        np.testing.assert_allclose(I.grad.numpy(), g_nccwin[tag + "_gI"], atol=5e-7, rtol=1e-5, err_msg=tag)
        le, gJ, gI = orc.ncc_explicit_win(Ia, Ja, win, grad=True)
        Id, Jd = T(Ia).double().requires_grad_(), T(Ja).double().requires_grad_()
        ld = orc.ncc_loss(Id, Jd, win=win)
        ld.backward()
        assert abs(le - ld.item()) < 1e-12, tag
        assert np.abs(gJ - Jd.grad.numpy()).max() < 1e-12 and np.abs(gI - Id.grad.numpy()).max() < 1e-12, tag
        assert abs(le - float(g_nccwin[tag])) < 1e-5, tag                 # fp64 arbiter vs the reference's fp32 evaluation
    # odd windows of one size: the general restatement is the cubic one
    assert abs(orc.ncc_explicit_win(g_nccwin["I3"], g_nccwin["J3"], [5, 5, 5]) - orc.ncc_explicit(g_nccwin["I3"], g_nccwin["J3"], 5)) < 1e-14
    with pytest.raises(ValueError):
        orc.ncc_explicit_win(g_nccwin["I2"][..., :4], g_nccwin["J2"][..., :4], [3, 9])      # 4 + 2 * (3 // 2) < 9: no box sums along W
    with pytest.raises(RuntimeError):
        orc.ncc_loss(T(g_nccwin["I2"][..., :4]), T(g_nccwin["J2"][..., :4]), win=[3, 9])    # ... and the reference's conv raises


@pytest.mark.parametrize("tag", ["p3_vol", "aniso_vol", "p3_img", "p42_img"])
def test_unet_with_other_pooling_factors_matches_reference(g_unetpools, tag):
    """Unet(max_pool = 3, per-axis tuples, per-level lists; networks.py:79-85,122-144): the restatement against the output and the
    gradients of the unmodified reference"""
    from conftest import UNET_POOL_CASES
    g = g_unetpools
    inshape, infeats, feats, pool = UNET_POOL_CASES[tag]
    names = [str(n) for n in g[tag + "_grad_names"]]
    sd = {"unet_model." + n: T(g[tag + "_param_" + n]).requires_grad_() for n in names}
    x = T(g[tag + "_x"]).requires_grad_()
    y = orc.unet_forward(x, sd, nb_features=feats, max_pool=pool)
    np.testing.assert_allclose(y.detach().numpy(), g[tag + "_y"], atol=2e-6, rtol=1e-5)
    (y * T(g[tag + "_r"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g[tag + "_gx"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(sd["unet_model.encoder.0.0.main.weight"].grad.numpy(), g[tag + "_gw_enc0"], atol=2e-4, rtol=1e-4)
    for n, ref in zip(names, g[tag + "_grad_norms"]):
        assert abs(float(sd["unet_model." + n].grad.double().norm()) - ref) <= 1e-4 * ref + 1e-6, n


def test_mse_grad_dice(g_losses):
    I, J = T(g_losses["I"]), T(g_losses["J"])
    np.testing.assert_allclose(orc.mse_loss(I, J).item(), g_losses["mse"], rtol=1e-6)
    fl = T(g_losses["flow"])
    np.testing.assert_allclose(orc.grad_loss(fl, "l1").item(), g_losses["grad_l1"], rtol=1e-6)
    np.testing.assert_allclose(orc.grad_loss(fl, "l2", 2).item(), g_losses["grad_l2"], rtol=1e-6)
    np.testing.assert_allclose(c_oracle.grad_loss(g_losses["flow"], "l1"), g_losses["grad_l1"], rtol=1e-5)
    np.testing.assert_allclose(c_oracle.grad_loss(g_losses["flow"], "l2", 2), g_losses["grad_l2"], rtol=1e-5)
    f2 = T(g_losses["flow"]).requires_grad_()
    orc.grad_loss(f2, "l2", 2).backward()
    np.testing.assert_allclose(f2.grad.numpy(), g_losses["grad_l2_g"], atol=1e-8)
    yp = T(g_losses["dice_pred"]).requires_grad_()
    d = orc.dice_loss(T(g_losses["dice_true"]), yp)
    np.testing.assert_allclose(d.item(), g_losses["dice"], rtol=1e-6)
    d.backward()
    np.testing.assert_allclose(yp.grad.numpy(), g_losses["dice_g"], atol=1e-9)
    with pytest.raises(AssertionError):
        orc.grad_loss(fl, "l3")


def test_dice_metric(g_dice):
    np.testing.assert_array_equal(orc.dice_metric(g_dice["a"], g_dice["b"], labels=[1, 2, 3, 5]), g_dice["dice"])


# ------------------------------------------------------------------ network
@pytest.mark.parametrize("tag,kw,loss,lam", [
    ("diffeo", dict(int_steps=7, int_downsize=2), "ncc", 1.0),
    ("dense", dict(int_steps=0, int_downsize=2), "mse", 0.01),
])
def test_vxm_dense(g_network, tag, kw, loss, lam):
    inshape = tuple(int(v) for v in g_network["inshape"])
    sd = orc.seeded_state_dict(inshape, seed=5, flow_std=0.2)
    for v in sd.values():
        v.requires_grad_()
    src, trg = T(g_network["source"]), T(g_network["target"])
    total, (img, reg, ys, pre) = orc.train_step_loss(src, trg, sd, image_loss=loss, lam=lam, **kw)
    np.testing.assert_allclose(ys.detach().numpy(), g_network[tag + "_y_source"], atol=1e-6)
    np.testing.assert_allclose(pre.detach().numpy(), g_network[tag + "_preint"], atol=1e-6)
    np.testing.assert_allclose([total.item(), img.item(), reg.item()], g_network[tag + "_loss"], rtol=1e-5)
    total.backward()
    names = [str(n) for n in g_network[tag + "_grad_names"]]
    norms = g_network[tag + "_grad_norms"]
    for n, ref in zip(names, norms):
        got = float(sd[n].grad.double().norm())
        assert abs(got - ref) <= 1e-4 * max(ref, 1e-12), (n, got, ref)
    with torch.no_grad():
        _, pos = orc.vxm_dense_forward(src, trg, sd, registration=True, **kw)
    np.testing.assert_allclose(pos.numpy(), g_network[tag + "_pos_flow"], atol=1e-5)


def test_vxm_dense_bidir(g_network):
    inshape = tuple(int(v) for v in g_network["inshape"])
    sd = orc.seeded_state_dict(inshape, seed=5, flow_std=0.2)
    ys, yt, pre = orc.vxm_dense_forward(T(g_network["source"]), T(g_network["target"]), sd,
                                        int_steps=3, int_downsize=2, bidir=True)
    np.testing.assert_allclose(ys.numpy(), g_network["bidir_y_source"], atol=1e-6)
    np.testing.assert_allclose(yt.numpy(), g_network["bidir_y_target"], atol=1e-6)
    np.testing.assert_allclose(pre.numpy(), g_network["bidir_preint"], atol=1e-6)


def test_state_dict_layout(g_network):
    keys = [str(k) for k in g_network["state_keys"] if not str(k).endswith(".grid")]
    shapes = [str(s) for k, s in zip(g_network["state_keys"], g_network["state_shapes"])
              if not str(k).endswith(".grid")]
    mine = orc.state_dict_shapes((16, 16, 16))
    assert [k for k, _ in mine] == keys
    assert [str(tuple(s)) for _, s in mine] == shapes
    assert sum(int(np.prod(s)) for _, s in mine) == int(g_network["n_params"]) == 327331


def test_conv_c_arbiter():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 3, 5, 6, 7)).astype(np.float32)
    w = rng.standard_normal((4, 3, 3, 3, 3)).astype(np.float32)
    b = rng.standard_normal(4).astype(np.float32)
    ref = orc.conv_block(T(x), T(w), T(b)).numpy()
    np.testing.assert_allclose(c_oracle.conv3d_k3(x, w, b), ref, atol=2e-5)


# ------------------------------------------------------------------ 2-D (planar) cases of the same classes
def test_planar_layers(g_planar):
    g = g_planar
    s, f = T(g["warp_src"]).requires_grad_(), T(g["warp_flow"]).requires_grad_()
    out = orc.spatial_transformer(s, f)
    assert torch.equal(out.detach(), T(g["warp_out"]))
    out.backward(T(g["warp_gout"]))
    np.testing.assert_allclose(s.grad.numpy(), g["warp_gsrc"], atol=1e-6)
    np.testing.assert_allclose(f.grad.numpy(), g["warp_gflow"], atol=1e-5)
    np.testing.assert_allclose(orc.warp_explicit(g["warp_src"], g["warp_flow"]), g["warp_out"], atol=2e-6)
    assert np.array_equal(orc.warp_explicit(g["near_seg"], g["near_flow"], mode="nearest"), g["near_out"])
    v = T(g["vecint_in"]).requires_grad_()
    iv = orc.vecint(v, 5)
    assert torch.equal(iv.detach(), T(g["vecint_out"]))
    iv.backward(T(g["vecint_gout"]))
    np.testing.assert_allclose(v.grad.numpy(), g["vecint_gin"], atol=1e-5)
    assert torch.equal(orc.resize_transform(T(g["resize_in"]), 2), T(g["resize_down"]))
    assert torch.equal(orc.resize_transform(T(g["resize_in"]), 0.5), T(g["resize_up"]))
    np.testing.assert_allclose(orc.resize_explicit(g["resize_in"], 2), g["resize_down"], atol=2e-6)
    np.testing.assert_allclose(orc.resize_explicit(g["resize_in"], 0.5), g["resize_up"], atol=2e-6)


def test_planar_losses(g_planar):
    g = g_planar
    I, J = T(g["I"]), T(g["J"]).requires_grad_()
    l = orc.ncc_loss(I, J)
    np.testing.assert_allclose(l.item(), g["ncc"], rtol=1e-6)
    l.backward()
    np.testing.assert_allclose(J.grad.numpy(), g["ncc_gJ"], atol=1e-7)
    np.testing.assert_allclose(orc.ncc_loss(I, T(g["J"]), win=[5, 5]).item(), g["ncc5"], rtol=1e-6)
    fl = T(g["warp_flow"]).requires_grad_()
    np.testing.assert_allclose(orc.grad_loss(fl, "l1").item(), g["grad_l1"], rtol=1e-6)
    l2 = orc.grad_loss(fl, "l2", 2)
    np.testing.assert_allclose(l2.item(), g["grad_l2"], rtol=1e-6)
    l2.backward()
    np.testing.assert_allclose(fl.grad.numpy(), g["grad_l2_g"], atol=1e-8)


@pytest.mark.parametrize("tag,kw,loss,lam", [
    ("diffeo", dict(int_steps=5, int_downsize=2), "ncc", 1.0),
    ("dense", dict(int_steps=0, int_downsize=2), "mse", 0.01),
])
def test_planar_vxm_dense(g_planar, tag, kw, loss, lam):
    g = g_planar
    inshape = tuple(int(v) for v in g["inshape"])
    sd = orc.seeded_state_dict(inshape, seed=7, flow_std=0.2)
    for v in sd.values():
        v.requires_grad_()
    src, trg = T(g["source"]), T(g["target"])
    total, (img, reg, ys, pre) = orc.train_step_loss(src, trg, sd, image_loss=loss, lam=lam, **kw)
    np.testing.assert_allclose(ys.detach().numpy(), g[tag + "_y_source"], atol=1e-6)
    np.testing.assert_allclose(pre.detach().numpy(), g[tag + "_preint"], atol=1e-6)
    np.testing.assert_allclose([total.item(), img.item(), reg.item()], g[tag + "_loss"], rtol=1e-5)
    total.backward()
    for n, ref in zip([str(n) for n in g[tag + "_grad_names"]], g[tag + "_grad_norms"]):
        got = float(sd[n].grad.double().norm())
        assert abs(got - ref) <= 1e-4 * max(ref, 1e-12), (n, got, ref)
    keys = [str(k) for k in g["state_keys"] if not str(k).endswith(".grid")]
    assert [k for k, _ in orc.state_dict_shapes((16, 16))] == keys


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference tree not present")
def test_live_reference_agrees():
    vxm = ref_loader.load_reference()
    rng = np.random.default_rng(21)
    vol = (8, 9, 10)
    src = T(rng.random((1, 1) + vol).astype(np.float32))
    flow = T((rng.standard_normal((1, 3) + vol) * 2).astype(np.float32))
    assert torch.equal(vxm.layers.SpatialTransformer(vol)(src, flow), orc.spatial_transformer(src, flow))
    assert torch.equal(vxm.layers.VecInt(vol, 5)(flow), orc.vecint(flow, 5))
    with ref_loader.cuda_alias_to_cpu():
        a = vxm.losses.NCC().loss(src, src * 0.5 + 0.1)
    assert torch.equal(a, orc.ncc_loss(src, src * 0.5 + 0.1))


@pytest.mark.skipif(not ref_loader.reference_available(), reason="reference tree not present")
def test_reference_one_dimensional_path_is_not_runnable():
    """Why 1-D is not a row of this build: the reference's torch backend accepts ndims = 1 in its asserts (networks.py:51,196,
    layers.py:79-83) but cannot execute it -- SpatialTransformer passes a 3-D tensor to grid_sample (layers.py:41-48 permutes
    only 2-D / 3-D grids), which ATen rejects."""
    vxm = ref_loader.load_reference()
    x = torch.rand(1, 1, 64)
    with pytest.raises(RuntimeError, match="grid_sampler"):
        vxm.layers.SpatialTransformer((64,))(x, torch.zeros(1, 1, 64))
    with pytest.raises(RuntimeError, match="grid_sampler"):
        vxm.networks.VxmDense((64,), int_steps=0)(x, x)
