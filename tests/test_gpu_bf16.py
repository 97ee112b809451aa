"""bf16-activation path (BASELINE.json configs[1]) through the C ABI against the oracle.  Run on the MI355X box: `pytest -m gpu`.

Two references per case:
  * the oracle with the SAME rounding points (`oracle.vxm_oracle.bf16_activations`: operands rounded to bf16, accumulation in
    fp64, one rounding of the result) — the kernels may differ from it only by fp32-vs-fp64 accumulation, i.e. by an
    occasional flip of the final bf16 rounding (1 bf16 ulp = 2^-8 relative on single elements, ~1e-3 rel-L2);
  * the plain fp32 oracle — the stated tolerance of SURVEY.md §8c for bf16 mode: conv activations rel-L2 <= 1e-2.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vxm_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vxm():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    import voxelmorph_amd
    from voxelmorph_amd import _lib
    _lib.lib()
    return voxelmorph_amd


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def N(t):
    return t.detach().float().cpu().numpy()


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def rbf(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float64)


def to_blocked(x, cblk=None):
    """planar fp32 cuda [B,C,D,H,W] -> blocked bf16 [B,cblk/8,D,H,W,8] through the C ABI."""
    from voxelmorph_amd._lib import call, ptr, stream
    B, C = x.shape[:2]
    dims = tuple(x.shape[2:])
    cblk = cblk or (C + 15) // 16 * 16
    out = torch.empty((B, cblk // 8) + dims + (8,), dtype=torch.bfloat16, device=x.device)
    call("vxm_bf16_to_blocked", ptr(x), C, x[0].numel(), None, 0, 0, ptr(out), cblk, B, int(np.prod(dims)), stream())
    return out


def from_blocked(xb, C):
    from voxelmorph_amd._lib import call, ptr, stream
    B = xb.shape[0]
    dims = tuple(xb.shape[2:5])
    out = torch.empty((B, C) + dims, dtype=torch.float32, device=xb.device)
    call("vxm_bf16_from_blocked", ptr(xb), xb.shape[1] * 8, ptr(out), C, B, int(np.prod(dims)), stream())
    return out


def test_blocked_layout_roundtrip(vxm):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 19, 4, 6, 10)).astype(np.float32)
    xb = to_blocked(G(x), 32)
    assert xb.shape == (2, 4, 4, 6, 10, 8) and xb.dtype == torch.bfloat16
    ref = torch.from_numpy(x).to(torch.bfloat16)                                       # torch's RNE conversion
    got = xb.cpu().permute(0, 1, 5, 2, 3, 4).reshape(2, 32, 4, 6, 10)
    assert torch.equal(got[:, :19], ref) and float(got[:, 19:].abs().max()) == 0.0
    assert torch.equal(from_blocked(xb, 19).cpu(), ref.float())


CONV_CASES = [
    # c0, up0, c1, cout, vol, slope
    (16, False, 0, 16, (8, 8, 16), 0.2),
    (16, False, 0, 32, (10, 6, 20), 0.2),          # ragged tiles: D, H, W not multiples of the 8 x 6 x 16 (or 8 x 8 x 16) tile
    (32, False, 0, 16, (8, 12, 16), 0.2),
    (32, True, 16, 32, (16, 8, 32), 0.2),           # cat([upsample(x0), x1])
    (32, True, 32, 32, (8, 8, 16), 0.2),
    (32, False, 0, 32, (2, 4, 6), 1.0),             # coarsest level, no activation
    (16, False, 0, 3, (9, 7, 18), 1.0),             # flow head: planar fp32 output
]


def _conv_ref(x0, up0, x1, w, b, slope):
    x = torch.from_numpy(x0).double()
    x = rbf(x0)
    if up0:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    if x1 is not None:
        x = torch.cat([x, rbf(x1)], 1)
    z = F.conv3d(x, rbf(w), torch.from_numpy(b).double(), padding=1)
    return F.leaky_relu(z, slope) if slope != 1.0 else z


@pytest.mark.parametrize("c0,up0,c1,cout,vol,slope", CONV_CASES)
def test_bf16_conv_forward_and_mask_vs_oracle(vxm, c0, up0, c1, cout, vol, slope):
    from voxelmorph_amd.torch import functional_bf16 as VB
    rng = np.random.default_rng(c0 + 3 * c1 + cout)
    B = 2
    D, H, W = vol
    lo = tuple(s // 2 for s in vol)
    x0 = rng.standard_normal((B, c0) + (lo if up0 else vol)).astype(np.float32)
    x1 = rng.standard_normal((B, c1) + vol).astype(np.float32) if c1 else None
    w = (rng.standard_normal((cout, c0 + c1, 3, 3, 3)) / np.sqrt(27 * (c0 + c1))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    ref = _conv_ref(x0, up0, x1, w, b, slope)
    planar = cout <= 4
    wg = G(w)
    wp = VB.pack_weights(wg, 0, c0 + c1, False)
    x0b, x1b = to_blocked(G(x0)), (to_blocked(G(x1)) if c1 else None)
    if planar:
        y = torch.full((B, cout, D, H, W), float("nan"), device="cuda")
        VB.conv(x0b, c0, up0, x1b, c1, wp, G(b), y, cout, True, slope, None, 1.0, B, D, H, W)
        assert rel_l2(N(y), ref.numpy()) < 1e-5           # fp32 accumulation of exact bf16 products, no output rounding
        return
    yb = VB._blocked(B, cout, vol, "cuda")
    VB.conv(x0b, c0, up0, x1b, c1, wp, G(b), yb, cout, False, slope, None, 1.0, B, D, H, W)
    got = N(from_blocked(yb, cout))
    want = ref.to(torch.bfloat16).double().numpy()
    assert rel_l2(got, want) < 2e-3
    assert np.abs(got - want).max() <= 2.0 ** -7 * np.abs(want).max()                 # single elements: at most a rounding flip
    assert rel_l2(got, ref.numpy()) < 4e-3                                            # vs the unrounded result: bf16 resolution
    # the fused leaky_relu_backward mask of backward-data: y * LeakyReLU'(mask)
    m = rng.standard_normal((B, cout) + vol).astype(np.float32)
    VB.conv(x0b, c0, up0, x1b, c1, wp, G(b), yb, cout, False, slope, to_blocked(G(m)), 0.2, B, D, H, W)
    mk = np.where(rbf(m).numpy() > 0, 1.0, 0.2)
    assert rel_l2(N(from_blocked(yb, cout)), (ref.numpy() * mk)) < 4e-3


@pytest.mark.parametrize("c0,up0,c1,cout,vol", [(16, False, 0, 16, (16, 24, 48)), (32, True, 16, 32, (12, 10, 36)), (16, False, 0, 3, (9, 16, 40))])
def test_bf16_conv_reversed_tile_order_bit_exact(vxm, c0, up0, c1, cout, vol):
    """vxm_bf16_conv_fwd with `out_planar_f32 | 2` walks its output tiles from the end of the tensor (the snake order of consecutive
    full-resolution launches): scheduling, not arithmetic -- the same bits, blocked and planar outputs, with and without the fused mask"""
    from voxelmorph_amd import _lib
    from voxelmorph_amd.torch import functional_bf16 as VB
    rng = np.random.default_rng(7)
    B = 2
    D, H, W = vol
    lo = tuple(s // 2 for s in vol)
    x0b = to_blocked(G(rng.standard_normal((B, c0) + (lo if up0 else vol)).astype(np.float32)))
    x1b = to_blocked(G(rng.standard_normal((B, c1) + vol).astype(np.float32))) if c1 else None
    w = G((rng.standard_normal((cout, c0 + c1, 3, 3, 3)) / np.sqrt(27 * (c0 + c1))).astype(np.float32))
    b = G(rng.standard_normal(cout).astype(np.float32) * 0.1)
    wp = VB.pack_weights(w, 0, c0 + c1, False)
    planar = cout <= 4
    masks = [None] if planar else [None, to_blocked(G(rng.standard_normal((B, cout) + vol).astype(np.float32)))]
    for mask in masks:
        outs = []
        for rev in (0, 2):
            y = torch.full((B, cout, D, H, W), float("nan"), device="cuda") if planar else VB._blocked(B, cout, vol, "cuda")
            _lib.call("vxm_bf16_conv_fwd", _lib.ptr(x0b), c0, 1 if up0 else 0, _lib.ptr(x1b), c1, _lib.ptr(wp), _lib.ptr(b), _lib.ptr(y), cout,
                      (1 if planar else 0) | rev, 0.2, _lib.ptr(mask), 0.2, B, D, H, W, _lib.stream())
            outs.append(y)
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int16) if not planar else outs[0], outs[1].view(torch.int16) if not planar else outs[1])


@pytest.mark.parametrize("c0,up0,c1,cout,vol", [(16, False, 0, 16, (8, 8, 32)), (32, True, 16, 32, (12, 10, 36)), (16, False, 0, 3, (6, 16, 40)),
                                                (32, False, 0, 32, (2, 2, 2)), (2, False, 0, 16, (8, 8, 16))])
def test_bf16_conv_backward_data_and_weight_vs_oracle(vxm, c0, up0, c1, cout, vol):
    """gW, gb from blocked x / dz, and the adjoint (backward-data) per input segment, against autograd of the fp64 conv on the
    bf16-rounded operands.  (2 -> 16: the first layer, whose blocked input carries 14 zero channels; 16 -> 3: the flow head,
    whose blocked gradient carries 13.)"""
    from voxelmorph_amd.torch import functional_bf16 as VB
    from voxelmorph_amd.torch.functional import _Workspace
    rng = np.random.default_rng(7 * c0 + cout)
    B = 2
    D, H, W = vol
    lo = tuple(s // 2 for s in vol)
    x0 = rng.standard_normal((B, c0) + (lo if up0 else vol)).astype(np.float32)
    x1 = rng.standard_normal((B, c1) + vol).astype(np.float32) if c1 else None
    w = (rng.standard_normal((cout, c0 + c1, 3, 3, 3)) / np.sqrt(27 * (c0 + c1))).astype(np.float32)
    dz = rng.standard_normal((B, cout) + vol).astype(np.float32)
    X0 = rbf(x0).requires_grad_()
    X1 = rbf(x1).requires_grad_() if c1 else None
    Wt = rbf(w).requires_grad_()
    x = F.interpolate(X0, scale_factor=2, mode="nearest") if up0 else X0
    if c1:
        x = torch.cat([x, X1], 1)
    bias = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    F.conv3d(x, Wt, bias, padding=1).backward(rbf(dz))
    c0p, cdz = (c0 + 15) // 16 * 16, (cout + 15) // 16 * 16
    x0b, x1b, dzb = to_blocked(G(x0)), (to_blocked(G(x1)) if c1 else None), to_blocked(G(dz))
    gw = torch.full((cout, c0 + c1, 3, 3, 3), float("nan"), device="cuda")
    gb = torch.full((cout,), float("nan"), device="cuda")
    VB.conv_bwd_weight(_Workspace(torch.device("cuda")), x0b, c0p, up0, x1b, c1, dzb, cdz, gw, gb, B, D, H, W)
    assert rel_l2(N(gw), Wt.grad.numpy()) < 1e-5
    assert rel_l2(N(gb), bias.grad.numpy()) < 1e-5
    gw2 = torch.empty_like(gw)
    VB.conv_bwd_weight(_Workspace(torch.device("cuda")), x0b, c0p, up0, x1b, c1, dzb, cdz, gw2, gb, B, D, H, W)
    assert torch.equal(gw, gw2)                                                        # fixed-order partial sums
    wg = G(w)
    for lo_c, n_c, ref in ((0, c0, X0.grad), (c0, c1, X1.grad if c1 else None)):
        if not n_c:
            continue
        npad = (n_c + 15) // 16 * 16
        gx = VB._blocked(B, npad, vol, "cuda")
        VB.conv(dzb, cdz, False, None, 0, VB.pack_weights(wg, lo_c, n_c, True), None, gx, npad, False, 1.0, None, 1.0, B, D, H, W)
        got = from_blocked(gx, n_c).double().cpu()
        if up0 and lo_c == 0:               # gradient of the upsampled segment: sum over the 2x2x2 children (fp64 here; the
            got = got.reshape(B, n_c, lo[0], 2, lo[1], 2, lo[2], 2).sum(dim=(3, 5, 7))          # kernel under test comes next)
            assert rel_l2(got.numpy(), ref.numpy()) < 4e-3
            dzl = VB._blocked(B, n_c, lo, "cuda")
            from voxelmorph_amd._lib import call, ptr, stream
            call("vxm_bf16_upsample2_bwd", ptr(gx), None, ptr(dzl), 1.0, B, n_c, lo[0], lo[1], lo[2], stream())
            assert rel_l2(N(from_blocked(dzl, n_c)), got.numpy()) < 4e-3
        else:
            assert rel_l2(got.numpy(), ref.numpy()) < 4e-3


def test_bf16_pool_kernels_vs_oracle(vxm):
    from voxelmorph_amd._lib import call, ptr, stream
    from voxelmorph_amd.torch import functional_bf16 as VB
    rng = np.random.default_rng(3)
    B, C, vol = 2, 16, (4, 6, 8)
    lo = tuple(s // 2 for s in vol)
    x = rng.integers(-3, 4, size=(B, C) + vol).astype(np.float32)       # small integers: many ties inside the 2x2x2 blocks
    gp = rng.standard_normal((B, C) + lo).astype(np.float32)
    gs = rng.standard_normal((B, C) + vol).astype(np.float32)
    xb = to_blocked(G(x))
    yb = VB._blocked(B, C, lo, "cuda")
    call("vxm_bf16_maxpool2_fwd", ptr(xb), ptr(yb), B, C, *vol, stream())
    X = torch.from_numpy(x).double().requires_grad_()
    P = F.max_pool3d(X, 2)
    assert np.array_equal(N(from_blocked(yb, C)), P.detach().numpy())
    P.backward(rbf(gp))
    want = (X.grad + rbf(gs)) * torch.where(X.detach() > 0, 1.0, 0.2)       # first arg-max routing (ATen) + skip add + LeakyReLU'
    dz = VB._blocked(B, C, vol, "cuda")
    call("vxm_bf16_maxpool2_bwd", ptr(xb), ptr(to_blocked(G(gp))), ptr(to_blocked(G(gs))), ptr(dz), 0.2, B, C, *vol, stream())
    assert rel_l2(N(from_blocked(dz, C)), want.numpy()) < 3e-3
    call("vxm_bf16_maxpool2_bwd", ptr(xb), ptr(to_blocked(G(gp))), None, ptr(dz), 1.0, B, C, *vol, stream())
    assert np.array_equal(N(from_blocked(dz, C)), X.grad.to(torch.bfloat16).double().numpy())


def _dense_step(vxm, model, src, trg, lam=0.01):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, flow = model(src, trg)
        loss = vxm.losses.MSE().loss(trg, y) + lam * vxm.losses.Grad("l2", loss_mult=2).loss(None, flow)
    loss.backward()
    return y, flow, loss


@pytest.mark.parametrize("inshape,kw", [((32, 48, 32), {}), ((32, 32, 64), dict(nb_unet_features=[[16, 32], [32, 32, 16]]))])
def test_vxm_dense_bf16_vs_emulated_and_fp32_oracle(vxm, inshape, kw):
    """BASELINE.json configs[1] wiring at a small size: VxmDense(int_steps=0) under torch.autocast(bfloat16), MSE + 0.01 Grad."""
    rng = np.random.default_rng(11)
    src = rng.random((2, 1) + inshape).astype(np.float32)
    trg = rng.random((2, 1) + inshape).astype(np.float32)
    okw = dict(nb_features=kw["nb_unet_features"]) if kw else {}
    sd = orc.seeded_state_dict(inshape, seed=4, flow_std=0.05, **okw)
    model = vxm.networks.VxmDense(inshape, int_steps=0, **kw)
    model.load_state_dict(sd, strict=False)
    model = model.cuda()
    y, flow, loss = _dense_step(vxm, model, G(src), G(trg))
    assert flow.dtype == torch.float32 and flow.shape == (2, 3) + inshape and y.dtype == torch.float32
    res = {}
    for tag, emulate in (("emulated", True), ("fp32", False)):
        sdo = {k: v.clone().double().requires_grad_() for k, v in sd.items()}
        S, T = torch.from_numpy(src).double(), torch.from_numpy(trg).double()
        if emulate:
            with orc.bf16_activations():
                ref, (_, _, ys, fl) = orc.train_step_loss(S, T, sdo, "mse", 0.01, int_steps=0, **okw)
        else:
            ref, (_, _, ys, fl) = orc.train_step_loss(S, T, sdo, "mse", 0.01, int_steps=0, **okw)
        ref.backward()
        gerr = {n: rel_l2(N(p.grad), sdo[n].grad.numpy()) for n, p in model.named_parameters()}
        res[tag] = (rel_l2(N(flow), fl.detach().numpy()), abs(float(loss) - float(ref)), max(gerr.values()), gerr)
        print("%s: flow rel-L2 %.2e, |dloss| %.2e, worst grad rel-L2 %.2e (%s)" % (tag, res[tag][0], res[tag][1], res[tag][2],
                                                                                 max(gerr, key=gerr.get)))
    assert res["emulated"][0] < 5e-3 and res["emulated"][1] < 1e-5 and res["emulated"][2] < 3e-2
    assert res["fp32"][0] < 1e-2 and res["fp32"][1] < 1e-4                           # bf16 conv activations rel-L2 <= 1e-2
    for n, e in res["fp32"][3].items():
        assert e < 0.15, (n, e)                                                       # bf16 gradients against fp32 ones (coarse levels: few voxels)


@pytest.mark.parametrize("cdz,cup,vol,masked", [(32, 32, (12, 6, 20), True), (32, 16, (8, 8, 16), True), (16, 32, (18, 10, 36), False),
                                                 (32, 32, (2, 2, 2), True)])
def test_fused_upsampled_segment_backward_data_equals_the_two_kernel_form(vxm, cdz, cup, vol, masked):
    """`vxm_bf16_conv_bwd_data_up` (adjoint conv + adjoint of nearest-x2 upsampling + leaky_relu_backward, one kernel, nothing
    written at full resolution) against `vxm_bf16_conv_fwd` followed by `vxm_bf16_upsample2_bwd`: same rounding points (every
    full-resolution value rounded to bf16 before the 2x2x2 sum), so the results may differ only by the order of an fp32 sum of
    eight bf16 numbers."""
    from voxelmorph_amd._lib import call, ptr, stream
    from voxelmorph_amd.torch import functional_bf16 as VB
    rng = np.random.default_rng(cdz + cup + vol[0])
    B = 2
    D, H, W = vol
    low = (D // 2, H // 2, W // 2)
    w = G((rng.standard_normal((cdz, cup + 16, 3, 3, 3)) / np.sqrt(27 * cdz)).astype(np.float32))      # Conv3d(cup + 16 -> cdz): segment 0 = upsampled
    dz = to_blocked(G(rng.standard_normal((B, cdz) + vol).astype(np.float32)))
    ylow = to_blocked(G(rng.standard_normal((B, cup) + low).astype(np.float32)))
    wp = VB.pack_weights(w, 0, cup, True)
    gfull = VB._blocked(B, cup, vol, "cuda")
    VB.conv(dz, cdz, False, None, 0, wp, None, gfull, cup, False, 1.0, None, 1.0, B, D, H, W)
    want = VB._blocked(B, cup, low, "cuda")
    call("vxm_bf16_upsample2_bwd", ptr(gfull), ptr(ylow) if masked else None, ptr(want), 0.2, B, cup, *low, stream())
    got = torch.full_like(want, float("nan"))
    call("vxm_bf16_conv_bwd_data_up", ptr(dz), cdz, ptr(wp), ptr(got), cup, ptr(ylow) if masked else None, 0.2, B, D, H, W, stream())
    a, b = got.float().cpu().numpy(), want.float().cpu().numpy()
    assert np.isfinite(a).all()
    assert (a == b).mean() > 0.995 and np.abs(a - b).max() <= 2.0 ** -7 * np.abs(b).max()
    with pytest.raises(Exception, match="even extents"):
        call("vxm_bf16_conv_bwd_data_up", ptr(dz), cdz, ptr(wp), ptr(got), cup, None, 0.2, B, D, H, W - 1, stream())


def test_batched_weight_packing_equals_per_operator_packing(vxm):
    """`vxm_bf16_conv_pack_weights_batch` (one launch for a whole network; > 48 jobs: more than one kernel-argument table) writes
    the same bytes as the per-operator entry point, and `prepack` fills the cache `pack_weights` reads."""
    import ctypes
    from voxelmorph_amd import _lib
    from voxelmorph_amd._lib import call, stream
    from voxelmorph_amd.torch import functional_bf16 as VB
    rng = np.random.default_rng(3)
    ws = [G(rng.standard_normal(sh).astype(np.float32)) for sh in [(16, 2, 3, 3, 3), (32, 16, 3, 3, 3), (32, 48, 3, 3, 3), (3, 16, 3, 3, 3)]]
    specs = []
    for w in ws:
        cin = w.shape[1]
        specs += [(w, 0, cin, False), (w, 0, cin, True)]
        if cin == 48:
            specs += [(w, 0, 32, True), (w, 32, 16, True)]
    specs = specs * 6                                                     # 60 jobs
    singles = []
    for w, lo, n, flip in specs:
        cout, cin = w.shape[:2]
        inc, outc = (cout, n) if flip else (n, cout)
        wp = torch.zeros(_lib.lib().vxm_bf16_conv_packed_bytes(inc, outc), dtype=torch.uint8, device="cuda")
        call("vxm_bf16_conv_pack_weights", w.data_ptr(), cin, cout, lo, n, 1 if flip else 0, wp.data_ptr(), stream())
        singles.append(wp)
    outs = [torch.full_like(x, 0xAB) for x in singles]
    table = (_lib.Bf16PackJob * len(specs))()
    for j, ((w, lo, n, flip), o) in enumerate(zip(specs, outs)):
        table[j] = _lib.Bf16PackJob(w.data_ptr(), o.data_ptr(), w.shape[1], w.shape[0], lo, n, 1 if flip else 0)
    call("vxm_bf16_conv_pack_weights_batch", ctypes.cast(table, ctypes.c_void_p), len(specs), stream())
    for a, b in zip(singles, outs):
        assert torch.equal(a, b)
    table[3].ci_n = 99
    with pytest.raises(Exception, match="channel range"):
        call("vxm_bf16_conv_pack_weights_batch", ctypes.cast(table, ctypes.c_void_p), len(specs), stream())
    VB.prepack(specs[:10])
    for (w, lo, n, flip), ref in zip(specs[:10], singles):
        ver, wp = w._vxm_bf16_packs[(lo, n, flip)]
        assert ver == VB._ver(w) and ver[0] == w._version and torch.equal(wp, ref) and VB.pack_weights(w, lo, n, flip) is wp


def test_bare_unet_bf16_vs_emulated_oracle(vxm):
    """`Unet.forward` alone under autocast (one input tensor, the last activation handed back as fp32 NCDHW), forward and
    gradients of the parameters and of the input, against the oracle with the same rounding points."""
    inshape = (32, 32, 32)
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 2) + inshape).astype(np.float32)
    sd = {k: v for k, v in orc.seeded_state_dict(inshape, seed=9).items() if k.startswith("unet_model.")}
    net = vxm.networks.Unet(inshape, infeats=2)
    net.load_state_dict({k[len("unet_model."):]: v for k, v in sd.items()})
    net = net.cuda()
    xg = G(x).requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(xg)
    assert y.dtype == torch.float32 and y.shape == (2, 16) + inshape
    # a cotangent with a non-zero mean: with a white-noise one every parameter gradient is a cancelling sum over the voxels
    # (condition ~ sqrt(#voxels)) and the two bf16 evaluations differ by 2-3 %, each 7-14 % from the fp64 one — measured, DESIGN.md §2
    gout = (rng.standard_normal(y.shape) + 1.0).astype(np.float32)
    y.backward(G(gout))
    sdo = {k: v.clone().double().requires_grad_() for k, v in sd.items()}
    xo = torch.from_numpy(x).double().requires_grad_()
    with orc.bf16_activations():
        yo = orc.unet_forward(xo, sdo)
    yo.backward(rbf(gout))
    assert rel_l2(N(y), yo.detach().numpy()) < 5e-3
    # d/dx: the first layer's dz is stored once more in bf16 after the pooled and the skip contribution are added (the oracle adds
    # the two rounded contributions in fp64); dx is a 432-term signed sum of it, so that 2^-9 shows up ~20x larger
    assert rel_l2(N(xg.grad), xo.grad.numpy()) < 3e-2
    gerr = {n: rel_l2(N(p.grad), sdo["unet_model." + n].grad.numpy()) for n, p in net.named_parameters()}
    print("bare Unet bf16: worst grad rel-L2 %.2e (%s)" % (max(gerr.values()), max(gerr, key=gerr.get)), sorted(gerr.items(), key=lambda kv: -kv[1])[:4])
    assert max(gerr.values()) < 6e-3, gerr


def test_vxm_dense_bf16_engine_is_selected_by_autocast_only(vxm):
    from voxelmorph_amd.torch import functional_bf16 as VB
    assert not VB.enabled()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert VB.enabled()
    with torch.autocast("cuda", dtype=torch.float16):
        assert not VB.enabled()
    VB.set_activation_dtype("bf16")
    try:
        assert VB.enabled()
    finally:
        VB.set_activation_dtype(None)
    model = vxm.networks.VxmDense((16, 16, 16), int_steps=0, nb_unet_features=[[4, 8], [8, 8, 4]]).cuda()
    x = torch.rand(1, 1, 16, 16, 16, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16), pytest.raises(Exception, match="multiples of 16"):
        model(x, x)


def test_full_size_dense_bf16_step_vs_oracle(vxm):
    """BASELINE.json configs[1] at the size it is quoted on: VxmDense 160x192x224, int_steps=0, MSE + 0.01 Grad('l2', x2), bf16
    activations; one training step against the oracle with the same rounding points (host cores, fp32 accumulation there)."""
    from voxelmorph_amd.optim import FlatAdam
    FULL = (160, 192, 224)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    rng = np.random.default_rng(99)
    src = rng.random((1, 1) + FULL).astype(np.float32)
    trg = rng.random((1, 1) + FULL).astype(np.float32)
    sd = orc.seeded_state_dict(FULL, seed=13, flow_std=0.02)
    model = vxm.networks.VxmDense(FULL, int_steps=0)
    model.load_state_dict(sd, strict=False)
    model = model.cuda()
    opt = FlatAdam(model, lr=1e-4)
    opt.zero_grad()
    y, flow, loss = _dense_step(vxm, model, G(src), G(trg))
    grads = {n: g.clone() for (n, _), g in zip(model.named_parameters(), opt._views)}
    opt.step()
    torch.cuda.synchronize()
    sdo = {k: v.clone().requires_grad_() for k, v in sd.items()}
    with orc.bf16_activations():
        ref, (_, _, ys, fl) = orc.train_step_loss(torch.from_numpy(src), torch.from_numpy(trg), sdo, "mse", 0.01, int_steps=0)
    ref.backward()
    e_flow = rel_l2(N(flow), fl.detach().numpy())
    gerr = {n: rel_l2(N(g), sdo[n].grad.numpy()) for n, g in grads.items()}
    worst = max(gerr, key=gerr.get)
    print("full-size bf16 dense step: loss hip=%.7f oracle=%.7f | flow rel-L2 %.2e | max|dy| %.2e | worst grad rel-L2 %.2e (%s)"
          % (float(loss), float(ref), e_flow, float((y.detach().cpu() - ys.detach()).abs().max()), gerr[worst], worst))
    assert e_flow < 1e-2 and abs(float(loss) - float(ref)) < 1e-4
    for n, e in gerr.items():
        assert e < 3e-2, (n, e)
