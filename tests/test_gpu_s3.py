"""Parity of the split-fp32 convolution kernels (csrc/conv_s3.hip: every fp32 operand as two fp16 pieces / three products with a
per-tile power-of-two scale -- engine "f16x2", the default -- or as three bf16 pieces / six products -- engine "split" --, fp32
accumulation) -- the forward / backward-data / backward-weight products of ConvBlock (voxelmorph/torch/networks.py:299-305)
-- against fp64 evaluations of the same operator, through the C ABI.  Every direct test runs on BOTH piece schemes.  `pytest -m gpu`.

Gate: the conv tolerance of the fp32-MFMA kernels (rel-L2 <= 1e-5 against fp64) AND "fp32-level": the error
must stay within a small factor of what the exact-fp32 kernel leaves on the same operands (measured ~1e-7 for both).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=["f16x2", "split"])
def VF(request):
    """the functional module with its split engine switched to one piece scheme for the duration of the module's tests"""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    from voxelmorph_amd import _lib
    _lib.lib()
    from voxelmorph_amd.torch import functional
    keep = functional.FP32_ENGINE
    functional.FP32_ENGINE = request.param
    yield functional
    functional.FP32_ENGINE = keep


@pytest.fixture()
def VF16():
    """the functional module on the fp16 piece scheme (the channel-blocked operands exist there only)"""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    from voxelmorph_amd import _lib as lib_
    lib_.lib()
    from voxelmorph_amd.torch import functional
    keep = functional.FP32_ENGINE
    functional.FP32_ENGINE = "f16x2"
    yield functional
    functional.FP32_ENGINE = keep


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _ref_conv(x0, up0, x1, w, bias, slope):
    """fp64 ConvBlock over cat([upsample2(x0) if up0 else x0, x1]) on the host"""
    xin = x0.double()
    if up0:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2, mode="nearest")
    if x1 is not None:
        xin = torch.cat([xin, x1.double()], 1)
    y = torch.nn.functional.conv3d(xin, w.double(), None if bias is None else bias.double(), padding=1)
    return torch.nn.functional.leaky_relu(y, slope) if slope != 1.0 else y


def _s3_forward(VF, x0, up0, x1, w, bias, slope, cout, vol, B):
    """vxm_conv3d_k3_s3_fwd directly (whatever the shape: the size threshold of the dispatcher does not apply)"""
    c0, c1 = x0.shape[1], (x1.shape[1] if x1 is not None else 0)
    D, H, W = vol
    V = D * H * W
    y = torch.empty((B, cout) + tuple(vol), device="cuda")
    VF.s3_launch(x0, c0, x0[0].numel(), up0, x1, c1, c1 * V, VF.s3_pack(w, False, 0, c0 + c1, c0), bias, y, cout * V, cout, slope,
                 None, 0, 1.0, B, D, H, W)
    return y


CASES = [
    # c0, up0, c1, cout, vol, slope
    (16, False, 0, 16, (8, 8, 16), 0.2),
    (32, False, 0, 16, (9, 6, 20), 0.2),          # partial tiles in every direction
    (16, False, 0, 32, (8, 12, 16), 0.2),
    (32, False, 0, 32, (5, 7, 33), 1.0),
    (48, False, 0, 32, (8, 4, 32), 0.2),
    (8, False, 8, 16, (8, 8, 16), 0.2),           # two full-resolution segments
    (32, True, 16, 32, (16, 8, 32), 0.2),         # cat([upsample2(x0), x1]) through the upsampling gather
    (16, True, 0, 16, (8, 8, 16), 0.2),
    (24, False, 0, 8, (6, 9, 18), 0.2),           # channel counts that do not fill the chunks / the 16-channel tile
    (16, False, 40, 24, (4, 8, 16), 1.0),
]


@pytest.mark.parametrize("c0,up0,c1,cout,vol,slope", CASES)
def test_s3_forward_vs_fp64(VF, c0, up0, c1, cout, vol, slope):
    from voxelmorph_amd import _lib
    if _lib.lib().vxm_conv3d_k3_s3_variant(cout) % 10 == 2 and c1 and c0 % 16:
        pytest.skip("16-channel chunks need C0 % 16 == 0 beside a second segment (the dispatcher falls back to the fp32-MFMA kernel)")
    B = 2
    torch.manual_seed(1000 + c0 + 7 * cout)
    lo = tuple(s // 2 for s in vol)
    x0 = torch.randn(B, c0, *(lo if up0 else vol), device="cuda")
    x1 = torch.randn(B, c1, *vol, device="cuda") if c1 else None
    w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
    bias = torch.randn(cout, device="cuda")
    y = _s3_forward(VF, x0, up0, x1, w, bias, slope, cout, vol, B)
    ref = _ref_conv(x0.cpu(), up0, x1.cpu() if c1 else None, w.cpu(), bias.cpu(), slope)
    e = rel_l2(y.cpu().numpy(), ref.numpy())
    # the exact-fp32 evaluation of the same operator on the same device (torch's conv3d, fp32) as the yardstick for "fp32-level"
    y32 = _ref_conv(x0.cpu(), up0, x1.cpu() if c1 else None, w.cpu(), bias.cpu(), slope).float()
    e32 = rel_l2(y32.numpy(), ref.numpy())
    print("s3 forward [%s] (%d%s+%d -> %d, %s): rel-L2 vs fp64 %.2e (fp32 rounding of the exact result alone: %.2e)"
          % (VF.FP32_ENGINE, c0, "^" if up0 else "", c1, cout, "x".join(map(str, vol)), e, e32))
    assert e <= 1e-5, e
    assert e <= 1e-6, e                      # measured ~1e-7: a dropped piece product (2^-16 relative) would be ~1e-5


UP_CASES = [
    # c0 (upsampled), c1 (skip), cout, full-resolution volume, batch
    (32, 16, 32, (16, 8, 64), 2),          # rem0's channel split
    (32, 32, 32, (8, 12, 32), 1),          # dec3's
    (16, 8, 16, (10, 6, 36), 2),           # partial tiles in every direction, one 16-channel output tile
    (8, 0, 16, (8, 4, 32), 1),             # no skip segment
    (40, 24, 24, (4, 8, 34), 1),           # channel counts that do not fill the super-chunk / the output tiles
    (16, 16, 48, (8, 4, 32), 1),           # two output-channel groups (blockIdx.y)
]


def _s3u_forward(VF, x0, x1, w, bias, slope, cout, vol, B):
    c0, c1 = x0.shape[1], (x1.shape[1] if x1 is not None else 0)
    D, H, W = vol
    V = D * H * W
    y = torch.full((B + 1, cout + 3) + tuple(vol), 7.25, device="cuda")          # guard channels and a guard sample around the destination
    VF.s3u_launch(x0, c0, x0[0].numel(), x1, c1, c1 * V, VF.s3u_pack(w, c0, c1), bias, y, (cout + 3) * V, cout, slope, B, D, H, W)
    assert bool((y[:B, cout:] == 7.25).all()) and bool((y[B] == 7.25).all())
    return y[:B, :cout]


@pytest.mark.parametrize("c0,c1,cout,vol,B", UP_CASES)
def test_s3u_collapsed_forward_vs_fp64(VF, c0, c1, cout, vol, B):
    """cat([upsample2(x0), x1]) -> ConvBlock on the split + collapsed kernel (csrc/conv_s3u.hip) against the fp64 evaluation of the
    reference's op sequence (networks.py:133-138, 299-305), both piece schemes, output guards."""
    torch.manual_seed(2000 + c0 + 3 * c1 + 7 * cout)
    lo = tuple(s // 2 for s in vol)
    x0 = torch.randn(B, c0, *lo, device="cuda")
    x1 = torch.randn(B, c1, *vol, device="cuda") if c1 else None
    w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
    bias = torch.randn(cout, device="cuda")
    y = _s3u_forward(VF, x0, x1, w, bias, 0.2, cout, vol, B)
    ref = _ref_conv(x0.cpu(), True, x1.cpu() if c1 else None, w.cpu(), bias.cpu(), 0.2)
    e = rel_l2(y.cpu().numpy(), ref.numpy())
    print("s3u forward [%s] (%d^+%d -> %d, %s, B=%d): rel-L2 vs fp64 %.2e" % (VF.FP32_ENGINE, c0, c1, cout, "x".join(map(str, vol)), B, e))
    assert e <= 1e-6, e


def test_s3u_collapsed_forward_many_tiles_and_scales(VF):
    """several tiles per persistent block (VXM_S3U_PERSIST=-16 in the subprocess test), channel blocks of very different magnitude (the
    running scale of the fp16 scheme moves between stages) and exact scale invariance under powers of two"""
    torch.manual_seed(88)
    vol, B, c0, c1, cout = (24, 20, 72), 2, 32, 16, 32
    lo = tuple(s // 2 for s in vol)
    x0 = torch.randn(B, c0, *lo, device="cuda")
    x1 = torch.randn(B, c1, *vol, device="cuda")
    x0[:, 8:16] *= 2.0 ** 12
    x1[:, 8:] *= 2.0 ** -9
    w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
    bias = torch.randn(cout, device="cuda")
    y = _s3u_forward(VF, x0, x1, w, bias, 0.2, cout, vol, B).clone()
    e = rel_l2(y.cpu().numpy(), _ref_conv(x0.cpu(), True, x1.cpu(), w.cpu(), bias.cpu(), 0.2).numpy())
    assert e <= 1e-6, e
    ys = _s3u_forward(VF, x0 * 2.0 ** -30, x1 * 2.0 ** -30, w, None, 1.0, cout, vol, B).clone()
    y1 = _s3u_forward(VF, x0, x1, w, None, 1.0, cout, vol, B)
    assert torch.equal(ys, y1 * 2.0 ** -30)


@pytest.mark.parametrize("c0,c1,cout,vol,B", [(32, 16, 32, (16, 8, 64), 2), (16, 8, 24, (10, 6, 36), 2), (8, 0, 16, (8, 4, 32), 1), (40, 24, 32, (12, 8, 34), 1)])
def test_s3u_backward_data_onto_the_low_resolution_tensor_vs_fp64(VF, c0, c1, cout, vol, B):
    """k_s3u_dlow: conv backward + upsample_nearest3d_backward + leaky_relu_backward of the upsampled segment in one launch against fp64
    autograd of the reference's op sequence (networks.py:133-138, 299-305); destination guards.  fp16 scheme only (the bf16 scheme keeps
    the fp32-MFMA kernel: _ok says so)."""
    from voxelmorph_amd import _lib
    D, H, W = vol
    if VF.FP32_ENGINE != "f16x2":
        assert _lib.lib().vxm_conv3d_k3_s3u_bwd_low_ok(c0, cout, B, 64, 64, 64, 3) == 0
        pytest.skip("k_s3u_dlow runs the fp16 scheme")
    torch.manual_seed(4000 + c0 + cout)
    lo = tuple(s // 2 for s in vol)
    w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
    dz = torch.randn(B, cout, *vol, device="cuda")
    act = torch.randn(B, c0, *lo, device="cuda")                      # the decoder block's activation (mask source)
    big = torch.full((B + 1, c0 + 3) + lo, 7.25, device="cuda")
    gxl = big[:B, :c0]
    # (s3u_bwd_low takes contiguous per-sample strides of c0 * V / 8: give it its own tensor and copy into the guarded one)
    out = torch.empty(B, c0, *lo, device="cuda")
    VF.s3u_bwd_low(dz, cout, w, c0, c0 + c1, out, act, 0.2, B, D, H, W)
    xl = torch.zeros(B, c0, *lo, dtype=torch.float64, requires_grad=True)
    xin = torch.nn.functional.interpolate(xl, scale_factor=2, mode="nearest")
    if c1:
        xin = torch.cat([xin, torch.zeros(B, c1, *vol, dtype=torch.float64)], 1)
    torch.nn.functional.conv3d(xin, w.cpu().double(), None, padding=1).backward(dz.cpu().double())
    want = xl.grad * torch.where(act.cpu().double() > 0, 1.0, 0.2)
    e = rel_l2(out.cpu().numpy(), want.numpy())
    print("s3u backward-data-low [%s] (%d^(+%d) <- %d, %s, B=%d): rel-L2 vs fp64 %.2e" % (VF.FP32_ENGINE, c0, c1, cout, "x".join(map(str, vol)), B, e))
    assert e <= 1e-6, e
    del gxl, big


@pytest.mark.parametrize("c0,c1,cout,vol,B", [(32, 16, 32, (16, 8, 64), 2), (32, 32, 32, (8, 12, 36), 1), (16, 8, 24, (10, 6, 40), 2), (8, 16, 16, (4, 4, 32), 1),
                                              (32, 16, 32, (24, 20, 72), 1)])
def test_s3u_both_backward_data_products_from_one_staging_vs_fp64(VF, c0, c1, cout, vol, B):
    """k_s3u_bwd_pc (round 6): the gradient of the low-resolution x0 (with LeakyReLU' of its activation) AND of the skip tensor x1 of
    cat([upsample(x0), x1]) -> Conv3d from one staging of dz, against fp64 autograd of the reference's op sequence (networks.py:133-138, 299-305);
    planar and channel-blocked dz give the same bits; partial tiles in every direction; several tiles per block in the subprocess re-run
    (VXM_S3U_PERSIST=-16).  fp16 scheme only.  The fused step does not route to this kernel by default (measured: no faster than the two launches
    it replaces, csrc/conv_s3u.hip vxm_conv3d_k3_s3u_bwd_data_ok); the entry point is called directly here."""
    from voxelmorph_amd import _lib
    D, H, W = vol
    V = D * H * W
    assert _lib.lib().vxm_conv3d_k3_s3u_bwd_data_ok(c0, c1, cout, B, 64, 64, 64, 3) == 0
    if VF.FP32_ENGINE != "f16x2":
        pytest.skip("k_s3u_bwd_pc runs the fp16 scheme")
    torch.manual_seed(4100 + c0 + c1 + cout)
    lo = tuple(s // 2 for s in vol)
    w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
    dz = torch.randn(B, cout, *vol, device="cuda")
    dz[:, : max(8, cout // 2)] *= 2.0 ** 9                             # chunks of very different magnitude: the running scale moves between stages
    act = torch.randn(B, c0, *lo, device="cuda")
    gxl, gx1 = torch.full((B, c0) + lo, 7.25, device="cuda"), torch.full((B, c1) + tuple(vol), 7.25, device="cuda")
    VF.s3u_bwd_data(dz, cout, w, c0, c1, gxl, act, 0.2, gx1, B, D, H, W)
    xl = torch.zeros(B, c0, *lo, dtype=torch.float64, requires_grad=True)
    xs = torch.zeros(B, c1, *vol, dtype=torch.float64, requires_grad=True)
    xin = torch.cat([torch.nn.functional.interpolate(xl, scale_factor=2, mode="nearest"), xs], 1)
    torch.nn.functional.conv3d(xin, w.cpu().double(), None, padding=1).backward(dz.cpu().double())
    want_l = xl.grad * torch.where(act.cpu().double() > 0, 1.0, 0.2)
    e_l, e_s = rel_l2(gxl.cpu().numpy(), want_l.numpy()), rel_l2(gx1.cpu().numpy(), xs.grad.numpy())
    print("s3u backward-data, both products [%s] (%d^+%d <- %d, %s, B=%d): rel-L2 vs fp64 low %.2e skip %.2e"
          % (VF.FP32_ENGINE, c0, c1, cout, "x".join(map(str, vol)), B, e_l, e_s))
    assert e_l <= 1e-6 and e_s <= 1e-6, (e_l, e_s)
    for lay, dzz in ((0, dz), (VF.S3_IN0_BLOCKED, VF.to_blocked(dz))):
        g2, s2 = torch.full_like(gxl, float("nan")), torch.full_like(gx1, float("nan"))
        VF.s3u_bwd_data(dzz, cout, w, c0, c1, g2, act, 0.2, s2, B, D, H, W, lay=lay)
        assert torch.equal(g2, gxl) and torch.equal(s2, gx1), hex(lay)
    g3, s3 = torch.empty_like(gxl), torch.empty_like(gx1)              # no mask: LeakyReLU slope 1 of the producer
    VF.s3u_bwd_data(dz, cout, w, c0, c1, g3, None, 1.0, s3, B, D, H, W)
    assert rel_l2(g3.cpu().numpy(), xl.grad.numpy()) <= 1e-6 and torch.equal(s3, gx1)


@pytest.mark.parametrize("c0,cout,vol,B", [(32, 32, (8, 8, 64), 2), (16, 16, (6, 12, 36), 1), (32, 16, (10, 4, 32), 2), (16, 48, (4, 6, 68), 1)])
def test_s3u_backward_weight_of_the_upsampled_segment_vs_fp64(VF, c0, cout, vol, B):
    """k_s3u_bww: weight gradient of the x2-upsampled segment (64 offset contractions on the low-resolution grid, mapped onto the 27 taps)
    against fp64 autograd of upsample + conv (networks.py:133-138, 299); only its channel range of a wider gradient array is written;
    bit-wise run-to-run determinism.  fp16 scheme only."""
    from voxelmorph_amd import _lib
    D, H, W = vol
    if VF.FP32_ENGINE != "f16x2":
        assert _lib.lib().vxm_conv3d_k3_s3u_bwd_weight_ok(c0, cout, B, 64, 64, 64, 3) == 0
        pytest.skip("k_s3u_bww runs the fp16 scheme")
    torch.manual_seed(5000 + c0 + cout)
    lo = tuple(s // 2 for s in vol)
    x0 = torch.randn(B, c0, *lo, device="cuda")
    dz = torch.randn(B, cout, *vol, device="cuda")
    pad = 16
    gw = torch.full((cout, c0 + pad, 3, 3, 3), 7.25, device="cuda")
    ws = VF._Workspace(x0.device)
    VF.s3u_bwd_weight(ws, x0, c0, x0[0].numel(), dz, cout, gw, c0 + pad, B, D, H, W)
    assert bool((gw[:, c0:] == 7.25).all())
    wr = torch.zeros(cout, c0, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    xin = torch.nn.functional.interpolate(x0.cpu().double(), scale_factor=2, mode="nearest")
    torch.nn.functional.conv3d(xin, wr, None, padding=1).backward(dz.cpu().double())
    e = rel_l2(gw[:, :c0].cpu().numpy(), wr.grad.numpy())
    print("s3u backward-weight [%s] (%d^ x %d, %s, B=%d): rel-L2 vs fp64 %.2e" % (VF.FP32_ENGINE, c0, cout, "x".join(map(str, vol)), B, e))
    assert e <= 3e-6, e
    gw2 = torch.full_like(gw, 7.25)
    VF.s3u_bwd_weight(ws, x0, c0, x0[0].numel(), dz, cout, gw2, c0 + pad, B, D, H, W)
    assert torch.equal(gw, gw2)


def test_s3_backward_data_with_fused_mask_and_output_guard(VF):
    """The adjoint operator (transpose_flip pack) with the previous block's LeakyReLU' fused in the epilogue, written into a
    channel slice of a larger buffer: nothing outside the slice, the next sample or the tail may be touched."""
    B, cin, cout, vol = 2, 24, 32, (8, 9, 20)
    D, H, W = vol
    V = D * H * W
    torch.manual_seed(5)
    w = torch.randn(cout, cin + 8, 3, 3, 3, device="cuda") / (27 * cin) ** 0.5       # gradient onto input channels [8, 8 + cin) of a wider layer
    dz = torch.randn(B, cout, D, H, W, device="cuda")
    mask = torch.randn(B, cin, D, H, W, device="cuda")
    pad = 3
    big = torch.full((B + 1, cin + pad, D, H, W), 7.25, device="cuda")
    gx = big[:B, :cin]
    VF.s3_launch(dz, cout, cout * V, False, None, 0, 0, VF.s3_pack(w, True, 8, 8 + cin, cout), None, gx, (cin + pad) * V, cin, 1.0,
                 mask, cin * V, 0.2, B, D, H, W)
    assert bool((big[:B, cin:] == 7.25).all()) and bool((big[B] == 7.25).all())
    xr = torch.zeros(B, cin + 8, D, H, W, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(xr, w.cpu().double(), None, padding=1).backward(dz.cpu().double())
    want = xr.grad[:, 8:] * torch.where(mask.cpu().double() > 0, 1.0, 0.2)
    e = rel_l2(gx.cpu().numpy(), want.numpy())
    assert e <= 1e-6, e


def test_s3_scale_invariance_and_small_magnitudes(VF):
    """The split is exact under scaling by powers of two (no piece under- or overflows at the magnitudes gradients have: 1e-12 of
    an activation), so conv(2^k x) == 2^k conv(x) bit for bit; and operands of mixed magnitude keep the fp32-level error."""
    B, c, cout, vol = 1, 16, 16, (8, 8, 16)
    torch.manual_seed(9)
    x = torch.randn(B, c, *vol, device="cuda")
    w = torch.randn(cout, c, 3, 3, 3, device="cuda") / (27 * c) ** 0.5
    y = _s3_forward(VF, x, False, None, w, None, 1.0, cout, vol, B)
    for k in (-40, 17):
        ys = _s3_forward(VF, x * 2.0 ** k, False, None, w, None, 1.0, cout, vol, B)
        assert torch.equal(ys, y * 2.0 ** k), k
    xm = x * torch.exp(8.0 * torch.randn_like(x))                # magnitudes spread over ~ 20 orders
    ym = _s3_forward(VF, xm, False, None, w, None, 1.0, cout, vol, B)
    e = rel_l2(ym.cpu().numpy(), _ref_conv(xm.cpu(), False, None, w.cpu(), None, 1.0).numpy())
    assert e <= 1e-6, e
    z = _s3_forward(VF, torch.zeros_like(x), False, None, w, None, 1.0, cout, vol, B)
    assert bool((z == 0).all())
    # fp32's own range: magnitudes near the ends of it (the per-tile scale of f16x2 / the pieces of bf16x3 must not over- or underflow)
    for k in (-100, 100):
        yk = _s3_forward(VF, x * 2.0 ** k, False, None, w, None, 1.0, cout, vol, B)
        assert torch.equal(yk, y * 2.0 ** k), k


def test_s3_dynamic_range_inside_one_tile(VF):
    """What the two piece schemes do with an outlier INSIDE a staged tile (documented in include/vxm_hip.h): one activation 2^24 times the
    rest.  Outputs that see the outlier are dominated by it (fp32-level relative error in both schemes).  Outputs of the same tile that do
    NOT see it: bf16x3 keeps fp32-level relative error (8-bit exponents per piece); f16x2 keeps an ABSOLUTE error of 2^-40 of the tile's
    largest magnitude per term -- asserted here as the bound -- which is a relative error of ~1e-5 on those outputs for this 2^24 outlier.
    The whole tensor stays inside the 1e-6 rel-L2 gate either way."""
    B, c, cout, vol = 1, 16, 16, (8, 8, 16)
    torch.manual_seed(21)
    x = torch.randn(B, c, *vol, device="cuda")
    x[0, 3, 4, 2, 5] = 2.0 ** 24
    w = torch.randn(cout, c, 3, 3, 3, device="cuda") / (27 * c) ** 0.5
    y = _s3_forward(VF, x, False, None, w, None, 1.0, cout, vol, B).cpu().double()
    ref = _ref_conv(x.cpu(), False, None, w.cpu(), None, 1.0)
    assert rel_l2(y.numpy(), ref.numpy()) <= 1e-6
    far = torch.ones(vol, dtype=torch.bool)
    far[3:6, 1:4, 4:7] = False                                   # voxels whose 3 x 3 x 3 window does not contain the outlier
    err = (y - ref)[0][:, far].abs().max().item()
    scale = ref[0][:, far].abs().mean().item()
    l1 = w.abs().sum(dim=(1, 2, 3, 4)).max().item()
    print("dynamic range [%s]: max abs error away from a 2^24 outlier %.3e (typical output %.3e)" % (VF.FP32_ENGINE, err, scale))
    if VF.FP32_ENGINE == "split":
        assert err <= 1e-5 * scale
    else:
        assert err <= 2.0 ** -38 * 2.0 ** 24 * l1              # 2^-40 of the tile maximum per term (x2 for the two roundings), summed against |w|


def _same_footprint_and_finite_values(y, ref, what, gate=1e-6):
    """non-finite exactly where the fp64 evaluation of the reference operator is non-finite; everything else at the fp32-level gate"""
    y, ref = y.detach().cpu().double(), ref.detach().cpu().double()
    fy, fr = torch.isfinite(y), torch.isfinite(ref)
    assert torch.equal(fy, fr), "%s: %d non-finite outputs, the reference has %d (%d differ)" % (what, int((~fy).sum()), int((~fr).sum()), int((fy != fr).sum()))
    assert int((~fr).sum()) > 0, what
    e = rel_l2(y[fr].numpy(), ref[fr].numpy())
    assert e <= gate, (what, e)
    return int((~fr).sum())


def test_s3_non_finite_values_poison_what_the_reference_poisons(VF):
    """Round-5 verdict item 6: an Inf (or NaN) among the activations used to set the power-of-two scale of its whole staged unit on the fp16
    scheme (exponent field 255 -> 2^-114: every finite value of the unit flushed), where ATen's convolution confines it to the outputs whose
    3 x 3 x 3 window contains it.  The unit's scale is now taken over its FINITE values (s3_pieces.h s3_unit_max): forward (plain, two segments,
    upsample + cat), backward-data onto the low-resolution tensor and both weight gradients produce non-finite results exactly where the fp64
    reference does, and the fp32-level result everywhere else -- on both piece schemes."""
    torch.manual_seed(5)
    B, vol = 1, (16, 16, 32)
    D, H, W = vol
    V = D * H * W

    def poison(t, corner=True):
        t = t.clone()
        t[0, 3, t.shape[2] // 2, 2, 5] = float("inf")
        t[0, 9 % t.shape[1], 1, t.shape[3] - 3, t.shape[4] - 7] = float("nan")
        if corner:
            t[0, 5, 0, 0, 0] = float("-inf")
        return t
    # plain forward (k_s3_conv; k_s3p_conv in the VXM_S3_PC=1 re-run), 16 and 32 output channels
    for c0, cout in ((16, 16), (32, 16), (16, 32)):
        x = poison(torch.randn(B, c0, *vol, device="cuda"))
        w = torch.randn(cout, c0, 3, 3, 3, device="cuda") / (27 * c0) ** 0.5
        bias = torch.randn(cout, device="cuda")
        n = _same_footprint_and_finite_values(_s3_forward(VF, x, False, None, w, bias, 0.2, cout, vol, B), _ref_conv(x.cpu(), False, None, w.cpu(), bias.cpu(), 0.2),
                                              "forward %d -> %d" % (c0, cout))
        assert n <= 3 * 27 * cout
    # cat([upsample(x0), x1]) forward on the collapsed kernel: a poisoned low-resolution voxel reaches the windows of its eight children
    c0, c1, cout = 32, 16, 32
    x0 = poison(torch.randn(B, c0, D // 2, H // 2, W // 2, device="cuda"))
    x1 = poison(torch.randn(B, c1, *vol, device="cuda"))
    w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
    bias = torch.randn(cout, device="cuda")
    _same_footprint_and_finite_values(_s3u_forward(VF, x0, x1, w, bias, 0.2, cout, vol, B), _ref_conv(x0.cpu(), True, x1.cpu(), w.cpu(), bias.cpu(), 0.2),
                                      "upsample + cat forward")
    # weight gradients: a poisoned x poisons its input channel's slice of gw, a poisoned dz its output channel's slice (and gb entry).  (Poison
    # away from the volume border here: a product of a non-finite value with the ZERO PADDING of the other operand is NaN, and which operand a
    # kernel -- or ATen's vol2col -- pads is an implementation choice; k_s3_bww_pc<true> pads dz, the reference formula pads x.)
    for c, co in ((16, 16), (32, 16), (16, 32)):
        x, dz = poison(torch.randn(B, c, *vol, device="cuda"), corner=False), torch.randn(B, co, *vol, device="cuda")
        dz[0, 2, 3, 4, 5] = float("inf")
        ws = VF._Workspace(x.device)
        gw, gb = torch.empty(co, c, 3, 3, 3, device="cuda"), torch.empty(co, device="cuda")
        VF.s3_bwd_weight(ws, x, c, c * V, dz, co, gw, c, 0, gb, B, D, H, W)
        ref = torch.nn.grad.conv3d_weight(x.double().cpu(), (co, c, 3, 3, 3), dz.double().cpu(), padding=1)
        _same_footprint_and_finite_values(gw, ref, "weight gradient %d x %d" % (c, co), gate=3e-6)
        _same_footprint_and_finite_values(gb, dz.double().cpu().sum(dim=(0, 2, 3, 4)), "bias gradient %d" % co, gate=3e-6)
    if VF.s3u_bwd_weight_route(32, 32, B, D, H, W):
        x0, dz = poison(torch.randn(B, 32, D // 2, H // 2, W // 2, device="cuda"), corner=False), poison(torch.randn(B, 32, *vol, device="cuda"), corner=False)
        ws = VF._Workspace(x0.device)
        gw = torch.zeros(32, 48, 3, 3, 3, device="cuda")
        VF.s3u_bwd_weight(ws, x0, 32, x0[0].numel(), dz, 32, gw, 48, B, D, H, W)
        up = torch.nn.functional.interpolate(x0.double().cpu(), scale_factor=2, mode="nearest")
        ref = torch.nn.grad.conv3d_weight(up, (32, 32, 3, 3, 3), dz.double().cpu(), padding=1)
        _same_footprint_and_finite_values(gw[:, :32], ref, "weight gradient of the upsampled segment", gate=3e-6)
    if VF.s3u_bwd_low_route(32, 32, B, D, H, W):
        dz = poison(torch.randn(B, 32, *vol, device="cuda"))
        w = torch.randn(32, 48, 3, 3, 3, device="cuda") / (27 * 48) ** 0.5
        gxl = torch.empty(B, 32, D // 2, H // 2, W // 2, device="cuda")
        VF.s3u_bwd_low(dz, 32, w, 32, 48, gxl, None, 1.0, B, D, H, W)
        gup = torch.nn.grad.conv3d_input((B, 48) + vol, w.double().cpu(), dz.double().cpu(), padding=1)[:, :32]
        ref = gup.reshape(B, 32, D // 2, 2, H // 2, 2, W // 2, 2).sum(dim=(3, 5, 7))
        _same_footprint_and_finite_values(gxl, ref, "backward-data onto the low-resolution tensor", gate=3e-6)


def test_s3_forward_many_tiles(VF):
    """a volume of 2 x 3 x 9 x 5 = 270 tiles (partial in every direction) against the fp64 reference: with VXM_S3_PERSIST=-16 (subprocess
    test below) each of the 16 blocks walks ~17 tiles of its XCD's range"""
    torch.manual_seed(77)
    vol, B, cin, cout = (20, 35, 72), 2, 16, 32
    x = torch.randn(B, cin, *vol, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") / (27 * cin) ** 0.5
    bias = torch.randn(cout, device="cuda")
    y = _s3_forward(VF, x, False, None, w, bias, 0.2, cout, vol, B)
    e = rel_l2(y.cpu().numpy(), _ref_conv(x.cpu(), False, None, w.cpu(), bias.cpu(), 0.2).numpy())
    assert e <= 1e-6, e


@pytest.mark.parametrize("c,cout,vol,B", [(16, 16, (4, 8, 32), 2), (32, 16, (5, 7, 40), 2), (16, 32, (6, 4, 64), 1), (48, 32, (3, 9, 34), 2)])
def test_s3_backward_weight_vs_fp64(VF, c, cout, vol, B):
    """vxm_conv3d_k3_s3_bwd_weight directly: weight and bias gradient against fp64 autograd, partial tiles in every direction, a
    destination that is a channel sub-range of a wider weight array (nothing else written), and bit-wise run-to-run determinism."""
    D, H, W = vol
    V = D * H * W
    torch.manual_seed(300 + c + cout)
    x = torch.randn(B, c, D, H, W, device="cuda")
    dz = torch.randn(B, cout, D, H, W, device="cuda")
    pad = 16
    gw = torch.full((cout, c + pad, 3, 3, 3), 7.25, device="cuda")
    gb = torch.empty(cout, device="cuda")
    ws = VF._Workspace(x.device)
    VF.s3_bwd_weight(ws, x, c, c * V, dz, cout, gw, c + pad, pad, gb, B, D, H, W)
    assert bool((gw[:, :pad] == 7.25).all())
    wr = torch.zeros(cout, c, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    br = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(x.cpu().double(), wr, br, padding=1).backward(dz.cpu().double())
    e_w, e_b = rel_l2(gw[:, pad:].cpu().numpy(), wr.grad.numpy()), rel_l2(gb.cpu().numpy(), br.grad.numpy())
    print("s3 backward-weight [%s] (%d -> %d, %s, B=%d): rel-L2 vs fp64 gw %.2e gb %.2e" % (VF.FP32_ENGINE, c, cout, "x".join(map(str, vol)), B, e_w, e_b))
    assert e_w <= 3e-6 and e_b <= 3e-6, (e_w, e_b)          # fp32 accumulation over B V voxels (measured ~3e-7)
    gw2 = torch.full_like(gw, 7.25)
    gb2 = torch.empty_like(gb)
    VF.s3_bwd_weight(ws, x, c, c * V, dz, cout, gw2, c + pad, pad, gb2, B, D, H, W)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)


def test_s3_backward_weight_refuses_odd_width(VF):
    """the staging of k_s3_bwd_weight loads W-neighbouring voxel pairs: an odd W is not eligible (the dispatcher keeps the fp32-MFMA kernel)
    and a direct launch is a shape error, not a wrong gradient"""
    from voxelmorph_amd import _lib
    assert _lib.lib().vxm_conv3d_k3_s3_bwd_weight_ok(16, 16, 4, 64, 64, 66) == 1
    assert _lib.lib().vxm_conv3d_k3_s3_bwd_weight_ok(16, 16, 4, 64, 64, 65) == 0
    x = torch.randn(1, 16, 4, 8, 33, device="cuda")
    gw, gb = torch.empty(16, 16, 3, 3, 3, device="cuda"), torch.empty(16, device="cuda")
    with pytest.raises(Exception, match="even"):
        VF.s3_bwd_weight(VF._Workspace(x.device), x, 16, x[0].numel(), x, 16, gw, 16, 0, gb, 1, 4, 8, 33)


# ------------------------------------------------------------------ channel-blocked operands (VXM_S3_*_BLOCKED)
def _eq(a, b):
    return torch.equal(a, b)


@pytest.mark.parametrize("vol", [(8, 16, 32), (9, 21, 37), (16, 24, 48)])
@pytest.mark.parametrize("c0,cout", [(16, 16), (32, 16), (16, 32), (8, 24)])
def test_s3_conv_channel_blocked_operands_bit_exact(VF16, c0, cout, vol):
    """include/vxm_hip.h VXM_S3_IN0_BLOCKED / VXM_S3_OUT_BLOCKED on the 8-row instances of k_s3_conv (and k_s3p_conv: re-run with
    VXM_S3_PC=1 in the subprocess test): the same values in the channel-blocked layout give the same bits -- forward operators (bias,
    LeakyReLU) and adjoints with a fused mask, input / output / both, one and two samples, a 24-channel output (partial second tile)."""
    VF = VF16
    D, H, W = vol
    V = D * H * W
    torch.manual_seed(5)
    for flip in (False, True):
        for B in (1, 2):
            x = torch.randn(B, c0, D, H, W, device="cuda")
            w = torch.randn(*((c0, cout) if flip else (cout, c0)), 3, 3, 3, device="cuda") / (27 * c0) ** 0.5
            bias = None if flip else torch.randn(cout, device="cuda")
            mask = torch.randn(B, cout, D, H, W, device="cuda") if flip else None
            wp = VF.s3_pack(w, flip, 0, cout if flip else c0, c0)
            sl = 1.0 if flip else 0.2
            y0 = torch.empty(B, cout, D, H, W, device="cuda")
            VF.s3_launch(x, c0, c0 * V, False, None, 0, 0, wp, bias, y0, cout * V, cout, sl, mask, cout * V, 0.2, B, D, H, W)
            for lay in (VF.S3_IN0_BLOCKED, VF.S3_OUT_BLOCKED, VF.S3_IN0_BLOCKED | VF.S3_OUT_BLOCKED):
                xin = VF.to_blocked(x) if lay & VF.S3_IN0_BLOCKED else x
                mk = (VF.to_blocked(mask) if lay & VF.S3_OUT_BLOCKED else mask) if mask is not None else None
                y1 = torch.full_like(y0, float("nan"))
                VF.s3_launch(xin, c0, c0 * V, False, None, 0, 0, wp, bias, y1, cout * V, cout, sl, mk, cout * V, 0.2, B, D, H, W, lay=lay)
                assert _eq(y0, VF.from_blocked(y1) if lay & VF.S3_OUT_BLOCKED else y1), (flip, B, hex(lay))


@pytest.mark.parametrize("c0,cout", [(16, 32), (16, 16), (32, 16)])
def test_s3_sign_tensors_written_by_forward_and_read_by_backward_data_epilogues(VF16, c0, cout):
    """include/vxm_hip.h VXM_S3_OUT_SIGNS / VXM_S3_MASK_SIGNS (round 6): a forward launch with a channel-blocked output writes, beside it, one
    byte per four channels and voxel with the signs of what it stored -- the output itself unchanged; a backward-data launch that reads those
    bytes instead of the fp32 activation produces the same bits (LeakyReLU' takes nothing else from the activation).  k_s3_conv here, k_s3p_conv in
    the subprocess re-run with VXM_S3_PC=1, and the collapsed forward (k_s3u_conv_pc / k_s3u_conv); one and two samples, partial tiles."""
    VF = VF16
    D, H, W = 12, 20, 40
    V = D * H * W
    torch.manual_seed(11 + c0 + cout)

    def signs_of(y):
        return sum(((y[:, j::4] > 0).to(torch.uint8) << j) for j in range(4)).contiguous()
    for B in (1, 2):
        x = torch.randn(B, c0, D, H, W, device="cuda")
        w = torch.randn(cout, c0, 3, 3, 3, device="cuda") / (27 * c0) ** 0.5
        bias = torch.randn(cout, device="cuda")
        y0, y1 = torch.empty(B, cout, D, H, W, device="cuda"), torch.empty(B, cout, D, H, W, device="cuda")
        sg = torch.full((B, cout // 4, D, H, W), 255, dtype=torch.uint8, device="cuda")
        wp = VF.s3_pack(w, False, 0, c0, c0)
        VF.s3_launch(x, c0, c0 * V, False, None, 0, 0, wp, bias, y0, cout * V, cout, 0.2, None, 0, 1.0, B, D, H, W, lay=VF.S3_OUT_BLOCKED)
        VF.s3_launch(x, c0, c0 * V, False, None, 0, 0, wp, bias, y1, cout * V, cout, 0.2, sg, (cout // 4) * V, 1.0, B, D, H, W,
                     lay=VF.S3_OUT_BLOCKED | VF.S3_OUT_SIGNS)
        assert torch.equal(y0, y1) and torch.equal(sg, signs_of(VF.from_blocked(y0))), (B, "forward")
        dz = VF.to_blocked(torch.randn(B, cout, D, H, W, device="cuda"))
        wa = torch.randn(cout, c0, 3, 3, 3, device="cuda") / (27 * cout) ** 0.5
        act = torch.randn(B, c0, D, H, W, device="cuda")
        act[:, :, ::3] = 0.0                                           # exact zeros take the slope, as y > 0 ? 1 : slope does
        g0, g1 = torch.empty(B, c0, D, H, W, device="cuda"), torch.full((B, c0, D, H, W), float("nan"), device="cuda")
        wpa = VF.s3_pack(wa, True, 0, c0, cout)
        lay = VF.S3_IN0_BLOCKED | VF.S3_OUT_BLOCKED
        VF.s3_launch(dz, cout, cout * V, False, None, 0, 0, wpa, None, g0, c0 * V, c0, 1.0, VF.to_blocked(act), c0 * V, 0.2, B, D, H, W, lay=lay)
        VF.s3_launch(dz, cout, cout * V, False, None, 0, 0, wpa, None, g1, c0 * V, c0, 1.0, signs_of(act), (c0 // 4) * V, 0.2, B, D, H, W,
                     lay=lay | VF.S3_MASK_SIGNS)
        assert torch.equal(g0, g1), (B, "backward-data")
    if cout == 32:
        c1, B = 16, 2
        x0, x1 = torch.randn(B, c0, D // 2, H // 2, W // 2, device="cuda"), torch.randn(B, c1, D, H, W, device="cuda")
        w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
        bias = torch.randn(cout, device="cuda")
        y0, y1 = torch.empty(B, cout, D, H, W, device="cuda"), torch.empty(B, cout, D, H, W, device="cuda")
        sg = torch.full((B, cout // 4, D, H, W), 255, dtype=torch.uint8, device="cuda")
        wp = VF.s3u_pack(w, c0, c1)
        VF.s3u_launch(x0, c0, x0[0].numel(), x1, c1, c1 * V, wp, bias, y0, cout * V, cout, 0.2, B, D, H, W, lay=VF.S3_OUT_BLOCKED)
        VF.s3u_launch(x0, c0, x0[0].numel(), x1, c1, c1 * V, wp, bias, y1, cout * V, cout, 0.2, B, D, H, W, lay=VF.S3_OUT_BLOCKED, signs=sg)
        assert torch.equal(y0, y1) and torch.equal(sg, signs_of(VF.from_blocked(y0))), "collapsed forward"


def test_s3_layout_flags_are_refused_where_no_blocked_variant_exists(VF16):
    VF = VF16
    x = torch.randn(1, 16, 8, 4, 32, device="cuda")               # H = 4: the 4-row instance
    w = torch.randn(16, 16, 3, 3, 3, device="cuda")
    y = torch.empty_like(x)
    with pytest.raises(Exception, match="layout flags"):
        VF.s3_launch(x, 16, x[0].numel(), False, None, 0, 0, VF.s3_pack(w, False, 0, 16, 16), None, y, y[0].numel(), 16, 0.2, None, 0, 1.0,
                     1, 8, 4, 32, lay=VF.S3_IN0_BLOCKED)
    from voxelmorph_amd import _lib
    assert not _lib.lib().vxm_conv3d_k3_s3_layout_ok(16, 16, 0, 16, 16, 2)       # two segments
    assert not _lib.lib().vxm_conv3d_k3_s3_layout_ok(16, 0, 0, 16, 16, 3)        # bf16 pieces
    assert _lib.lib().vxm_conv3d_k3_s3_layout_ok(32, 0, 0, 16, 16, 2)


@pytest.mark.parametrize("vol", [(8, 16, 32), (10, 20, 38), (16, 24, 64)])
@pytest.mark.parametrize("c,cout", [(16, 16), (32, 16), (16, 32)])
def test_s3_backward_weight_channel_blocked_operands_bit_exact(VF16, c, cout, vol):
    """k_s3_bwd_weight with x and / or dz channel-blocked (the task loop is instantiated per staging role and layout): same bits"""
    VF = VF16
    D, H, W = vol
    V = D * H * W
    torch.manual_seed(6)
    for B in (1, 2):
        x, dz = torch.randn(B, c, D, H, W, device="cuda"), torch.randn(B, cout, D, H, W, device="cuda")
        ws = VF._Workspace(x.device)
        g0, b0 = torch.empty(cout, c, 3, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
        VF.s3_bwd_weight(ws, x, c, c * V, dz, cout, g0, c, 0, b0, B, D, H, W)
        for lay in (VF.S3_IN0_BLOCKED, VF.S3_IN1_BLOCKED, VF.S3_IN0_BLOCKED | VF.S3_IN1_BLOCKED):
            g1, b1 = torch.full_like(g0, float("nan")), torch.full_like(b0, float("nan"))
            VF.s3_bwd_weight(ws, VF.to_blocked(x) if lay & VF.S3_IN0_BLOCKED else x, c, c * V, VF.to_blocked(dz) if lay & VF.S3_IN1_BLOCKED else dz,
                             cout, g1, c, 0, b1, B, D, H, W, lay=lay)
            assert _eq(g0, g1) and _eq(b0, b1), (B, hex(lay))


@pytest.mark.parametrize("vol,B", [((16, 32, 64), 2), ((24, 36, 72), 1), ((9, 50, 96), 1)])
def test_few_channel_kernels_channel_blocked_operands_bit_exact(VF16, vol, B):
    """Round 6, late: the kernels either side of the LAST ConvBlock of the fused U-Net take the layout flags, so that its activation, gradient and
    LeakyReLU mask can be channel-blocked / a sign tensor too (include/vxm_hip.h: vxm_conv3d_k3_fewout_fwd_layout, vxm_conv3d_k3_fwd_layout,
    VXM_S3_IN0_BLOCKED of vxm_conv3d_k3_fewch_bwd_weight).  The 16 -> 1..3 forward (networks.py:211,257) from a blocked x, its adjoint 3 -> 16 onto
    a blocked gradient with the mask blocked or as signs (8 x 4 x 16 and 8 x 2 x 32 tiles, partial tiles), and its weight gradient from a blocked
    x: every result bit-identical to the planar launch."""
    VF = VF16
    from voxelmorph_amd import _lib
    D, H, W = vol
    V = D * H * W
    torch.manual_seed(D + H + W)
    x = torch.randn(B, 16, D, H, W, device="cuda")
    xb = VF.to_blocked(x)
    for cout in (3, 1, 2):
        w = torch.randn(cout, 16, 3, 3, 3, device="cuda") / (27 * 16) ** 0.5
        bias = torch.randn(cout, device="cuda")
        y0, y1 = torch.empty(B, cout, D, H, W, device="cuda"), torch.full((B, cout, D, H, W), float("nan"), device="cuda")
        VF.conv_forward(x, 16, 16 * V, False, None, 0, 0, w, bias, y0, cout * V, cout, 1.0, B, D, H, W)
        VF.conv_forward(xb, 16, 16 * V, False, None, 0, 0, w, bias, y1, cout * V, cout, 1.0, B, D, H, W, lay=VF.S3_IN0_BLOCKED)
        assert torch.equal(y0, y1), ("forward", cout)
        # the adjoint onto the 16 channels, LeakyReLU'(act) in the epilogue
        dz = torch.randn(B, cout, D, H, W, device="cuda")
        act = torch.randn(B, 16, D, H, W, device="cuda")
        act[:, :, ::3] = 0.0
        signs = sum(((act[:, j::4] > 0).to(torch.uint8) << j) for j in range(4)).contiguous()
        wpk = VF.pack_weights_cached(w, True, 0, 16)
        assert _lib.lib().vxm_conv3d_k3_fwd_variant(VF.ptr(dz), cout, cout * V, None, 0, 0, VF.ptr(wpk), 16, B, D, H, W) >= 200      # the few-input-channel kernel
        g0 = torch.empty(B, 16, D, H, W, device="cuda")
        VF.conv_bwd_data(dz, cout, w, g0, 16, act, 0.2, B, D, H, W)
        for lay, mk in ((VF.S3_OUT_BLOCKED, VF.to_blocked(act)), (VF.S3_OUT_BLOCKED | VF.S3_MASK_SIGNS, signs), (VF.S3_OUT_BLOCKED, None)):
            g1 = torch.full((B, 16, D, H, W), float("nan"), device="cuda")
            if mk is None:
                gref = torch.empty(B, 16, D, H, W, device="cuda")
                VF.conv_bwd_data(dz, cout, w, gref, 16, None, 1.0, B, D, H, W)
            else:
                gref = g0
            VF.conv_bwd_data(dz, cout, w, g1, 16, mk, 0.2 if mk is not None else 1.0, B, D, H, W, lay=lay)
            assert torch.equal(gref, VF.from_blocked(g1)), ("backward-data", cout, hex(lay), mk is None)
        # weight / bias gradient
        ws = VF._Workspace(x.device)
        gw0, gb0 = torch.empty_like(w), torch.empty_like(bias)
        gw1, gb1 = torch.full_like(w, float("nan")), torch.full_like(bias, float("nan"))
        VF.conv_bwd_weight(ws, x, 16, 16 * V, False, None, 0, 0, dz, cout, gw0, gb0, B, D, H, W)
        VF.conv_bwd_weight(ws, xb, 16, 16 * V, False, None, 0, 0, dz, cout, gw1, gb1, B, D, H, W, lay=VF.S3_IN0_BLOCKED)
        assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1), ("backward-weight", cout)
        assert rel_l2(gw0.cpu(), torch.nn.grad.conv3d_weight(x.double().cpu(), w.shape, dz.double().cpu(), padding=1)) < 1e-5


def test_few_channel_layout_flags_are_refused_where_they_do_not_apply(VF16):
    VF = VF16
    D, H, W = 8, 8, 32
    V = D * H * W
    x = torch.randn(1, 12, D, H, W, device="cuda")
    w = torch.randn(3, 12, 3, 3, 3, device="cuda")
    y = torch.empty(1, 3, D, H, W, device="cuda")
    with pytest.raises(Exception, match="layout flags"):
        VF.call("vxm_conv3d_k3_fewout_fwd_layout", VF.ptr(x), 12, 12 * V, VF.ptr(w), None, VF.ptr(y), 3 * V, 3, 1.0, 1, D, H, W, VF.S3_IN0_BLOCKED, VF.stream())
    x = torch.randn(1, 16, D, H, W, device="cuda")                 # 16 input channels: not the few-input-channel kernel
    w = torch.randn(16, 16, 3, 3, 3, device="cuda")
    y = torch.empty(1, 16, D, H, W, device="cuda")
    with pytest.raises(Exception, match="layout flags"):
        VF.conv_launch(x, 16, 16 * V, False, None, 0, 0, VF.pack_weights(w, False), None, y, 16 * V, 16, 0.2, None, 0, 1.0, 1, D, H, W, lay=VF.S3_OUT_BLOCKED)
    x2, dz = torch.randn(1, 1, D, H, W, device="cuda"), torch.randn(1, 16, D, H, W, device="cuda")
    ws = VF._Workspace(x.device)
    with pytest.raises(Exception, match="layout flags|channel-blocked"):   # the first block (2 -> 16): its 16-channel operand is dz, not x0
        VF.conv_bwd_weight(ws, x2, 1, V, False, x2, 1, V, dz, 16, torch.empty(16, 2, 3, 3, 3, device="cuda"), torch.empty(16, device="cuda"), 1, D, H, W,
                           lay=VF.S3_IN0_BLOCKED)


@pytest.mark.parametrize("vol", [(8, 8, 32), (10, 12, 36), (16, 24, 64)])
@pytest.mark.parametrize("c0,c1,cout", [(32, 16, 32), (16, 16, 16), (32, 32, 32), (16, 8, 24)])
def test_s3u_kernels_channel_blocked_operands_bit_exact(VF16, c0, c1, cout, vol):
    """the cat([upsample, skip]) kernels: channel-blocked OUTPUT of the forward (k_s3u_conv), channel-blocked dz of the backward-data onto
    the low-resolution tensor (k_s3u_dlow) and of the weight gradient of the upsampled segment (k_s3u_bww): same bits as the planar launch"""
    VF = VF16
    D, H, W = vol
    V = D * H * W
    torch.manual_seed(7)
    for B in (1, 2):
        x0, x1 = torch.randn(B, c0, D // 2, H // 2, W // 2, device="cuda"), torch.randn(B, c1, D, H, W, device="cuda")
        w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
        bias = torch.randn(cout, device="cuda")
        wp = VF.s3u_pack(w, c0, c1)
        y0 = torch.empty(B, cout, D, H, W, device="cuda")
        y1 = torch.full_like(y0, float("nan"))
        VF.s3u_launch(x0, c0, x0[0].numel(), x1, c1, c1 * V, wp, bias, y0, cout * V, cout, 0.2, B, D, H, W)
        VF.s3u_launch(x0, c0, x0[0].numel(), x1, c1, c1 * V, wp, bias, y1, cout * V, cout, 0.2, B, D, H, W, lay=VF.S3_OUT_BLOCKED)
        assert _eq(y0, VF.from_blocked(y1)), B
        dz = torch.randn(B, cout, D, H, W, device="cuda")
        dzb = VF.to_blocked(dz)
        act = torch.randn(B, c0, D // 2, H // 2, W // 2, device="cuda")
        g0, g1 = torch.empty_like(act), torch.full_like(act, float("nan"))
        VF.s3u_bwd_low(dz, cout, w, c0, c0 + c1, g0, act, 0.2, B, D, H, W)
        VF.s3u_bwd_low(dzb, cout, w, c0, c0 + c1, g1, act, 0.2, B, D, H, W, lay=VF.S3_IN0_BLOCKED)
        assert _eq(g0, g1), B
        if c0 in (16, 32) and cout % 16 == 0 and W % 4 == 0 and D % 2 == 0 and H % 2 == 0:
            ws = VF._Workspace(dz.device)
            gw0, gw1 = torch.zeros(cout, c0 + c1, 3, 3, 3, device="cuda"), torch.zeros(cout, c0 + c1, 3, 3, 3, device="cuda")
            VF.s3u_bwd_weight(ws, x0, c0, x0[0].numel(), dz, cout, gw0, c0 + c1, B, D, H, W)
            VF.s3u_bwd_weight(ws, x0, c0, x0[0].numel(), dzb, cout, gw1, c0 + c1, B, D, H, W, lay=VF.S3_IN1_BLOCKED)
            assert _eq(gw0, gw1), B


def _rerun(env_extra, select, files=("tests/test_gpu_s3.py",), timeout=900):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu"] + [os.path.join(ROOT, f) for f in files] + ["-k", select],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("vol,B", [((8, 16, 32), 1), ((9, 21, 37), 2), ((24, 24, 80), 1), ((40, 48, 64), 1)])
@pytest.mark.parametrize("c0,cout", [(16, 16), (32, 16), (16, 32)])
def test_s3_conv_reversed_tile_order_bit_exact(VF16, c0, cout, vol, B):
    """include/vxm_hip.h VXM_S3_REVERSE_TILES: walking the output tiles from the end of the tensor is scheduling, not arithmetic -- the same
    bits, forward operators and adjoints with a fused mask, alone and together with the channel-blocked layouts, small grids (one tile per
    block) and persistent ones (the last shape: 1800 tiles of the 16-channel kernel, i.e. the producer / consumer kernel's range too)."""
    VF = VF16
    D, H, W = vol
    V = D * H * W
    torch.manual_seed(8)
    for flip in (False, True):
        x = torch.randn(B, c0, D, H, W, device="cuda")
        w = torch.randn(*((c0, cout) if flip else (cout, c0)), 3, 3, 3, device="cuda") / (27 * c0) ** 0.5
        bias = None if flip else torch.randn(cout, device="cuda")
        mask = torch.randn(B, cout, D, H, W, device="cuda") if flip else None
        wp = VF.s3_pack(w, flip, 0, cout if flip else c0, c0)
        sl = 1.0 if flip else 0.2
        y0 = torch.empty(B, cout, D, H, W, device="cuda")
        VF.s3_launch(x, c0, c0 * V, False, None, 0, 0, wp, bias, y0, cout * V, cout, sl, mask, cout * V, 0.2, B, D, H, W)
        for lay in (0, VF.S3_IN0_BLOCKED | VF.S3_OUT_BLOCKED):
            xin = VF.to_blocked(x) if lay else x
            mk = (VF.to_blocked(mask) if lay else mask) if mask is not None else None
            y1 = torch.full_like(y0, float("nan"))
            VF.s3_launch(xin, c0, c0 * V, False, None, 0, 0, wp, bias, y1, cout * V, cout, sl, mk, cout * V, 0.2, B, D, H, W, lay=lay | VF.S3_REVERSE_TILES)
            assert _eq(y0, VF.from_blocked(y1) if lay else y1), (flip, hex(lay))


@pytest.mark.parametrize("c,cout,vol,B", [(16, 16, (8, 16, 32), 2), (32, 16, (10, 20, 38), 1), (16, 32, (16, 24, 64), 1)])
def test_s3_backward_weight_in_two_calls_bit_exact(VF16, c, cout, vol, B):
    """include/vxm_hip.h VXM_S3_BW_CONTRACT_ONLY / VXM_S3_BW_REDUCE_ONLY: a weight gradient as two calls (contraction into the workspace,
    then the reduction -- possibly on another stream) is the one call, bit for bit; same for the collapsed kernel of an upsampled segment."""
    VF = VF16
    D, H, W = vol
    V = D * H * W
    torch.manual_seed(9)
    x, dz = torch.randn(B, c, D, H, W, device="cuda"), torch.randn(B, cout, D, H, W, device="cuda")
    g0, b0 = torch.empty(cout, c, 3, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
    VF.s3_bwd_weight(VF._Workspace(x.device), x, c, c * V, dz, cout, g0, c, 0, b0, B, D, H, W)
    pending = []
    g1, b1 = torch.full_like(g0, float("nan")), torch.full_like(b0, float("nan"))
    VF.s3_bwd_weight(VF._Workspace(x.device, deferred=pending), x, c, c * V, dz, cout, g1, c, 0, b1, B, D, H, W)
    assert len(pending) == 1 and bool(torch.isnan(g1).all())           # nothing reduced yet
    name, args, flags, keep = pending[0]
    VF.call(name, *args, flags, VF.stream())
    assert _eq(g0, g1) and _eq(b0, b1)
    if VF.s3u_bwd_weight_route(c, cout, B, 2 * D, 2 * H, 2 * W):
        dz2 = torch.randn(B, cout, 2 * D, 2 * H, 2 * W, device="cuda")
        h0, h1 = torch.empty(cout, c, 3, 3, 3, device="cuda"), torch.full((cout, c, 3, 3, 3), float("nan"), device="cuda")
        VF.s3u_bwd_weight(VF._Workspace(x.device), x, c, c * V, dz2, cout, h0, c, B, 2 * D, 2 * H, 2 * W)
        pending = []
        VF.s3u_bwd_weight(VF._Workspace(x.device, deferred=pending), x, c, c * V, dz2, cout, h1, c, B, 2 * D, 2 * H, 2 * W)
        name, args, flags, keep = pending[0]
        VF.call(name, *args, flags, VF.stream())
        assert _eq(h0, h1)


def test_s3_other_kernel_instances_in_subprocess():
    """The packed layout depends on the kernel instance, which is chosen once per process: 16-channel chunks (VXM_S3_CB=2) and
    32-channel operators as two 16-channel groups (VXM_S3_NCT=1) re-run the direct tests above."""
    if any(os.environ.get(k) for k in ("VXM_S3_CB", "VXM_S3_NCT", "VXM_S3_PERSIST", "VXM_S3U_PERSIST", "VXM_S3_PC", "VXM_S3_BW_PC", "VXM_S3U_PC")):
        pytest.skip("already inside a variant run")
    _rerun({"VXM_S3_CB": "2"}, "forward_vs_fp64 or fused_mask or scale_invariance")
    _rerun({"VXM_S3_NCT": "1"}, "forward_vs_fp64 or fused_mask")
    # 16 blocks in all: every block walks several tiles of its XCD's range (the default grid only does so on volumes with > 2048 tiles);
    # the same for the collapsed kernels (k_s3u_conv_pc / k_s3u_dlow)
    _rerun({"VXM_S3_PERSIST": "-16", "VXM_S3U_PERSIST": "-16"}, "forward_vs_fp64 or fused_mask or many_tiles or s3u_collapsed or both_backward_data")
    _rerun({"VXM_S3_PERSIST": "0", "VXM_S3U_PERSIST": "0"}, "many_tiles")
    # the producer / consumer kernel (k_s3p_conv: by default from 2048 tiles of 8 x 8 x 16 up) on every eligible launch, with one block per
    # tile and with 16 blocks in all (every block streams several tiles through its two LDS buffers); and the alternating kernel everywhere
    _rerun({"VXM_S3_PC": "1"}, "forward_vs_fp64 or fused_mask or scale_invariance or dynamic_range or non_finite or many_tiles or s3_conv_channel_blocked or sign_tensors")
    _rerun({"VXM_S3_PC": "1", "VXM_S3P_BLOCKS": "16"}, "forward_vs_fp64 or fused_mask or dynamic_range or many_tiles")
    # the kernels the producer / consumer ones of round 6 replaced by default stay reachable (other piece scheme, odd extents, A/B):
    # k_s3_bwd_weight<2> for k_s3_bww_pc, k_s3u_conv<., 2> for k_s3u_conv_pc, k_s3_conv for k_s3p_conv
    _rerun({"VXM_S3_PC": "0", "VXM_S3_BW_PC": "0", "VXM_S3U_PC": "0"}, "many_tiles or backward_weight or s3u_collapsed or s3u_kernels or sign_tensors or non_finite")


def test_s3_through_the_dispatcher_on_small_volumes_in_subprocess():
    """VXM_S3_MIN_TILES=1 sends every eligible forward / backward-data launch of the ordinary parity tests through the split kernel
    (normally reserved for the large layers): ConvBlock vs the fp64 reference incl. gradients, output guards, the five U-Net
    topologies, and the VxmDense goldens generated by the unmodified reference."""
    if os.environ.get("VXM_S3_MIN_TILES"):
        pytest.skip("already inside the forced run")
    for engine in ("f16x2", "split"):
        _rerun({"VXM_S3_MIN_TILES": "1", "VXM_S3U_MIN_TILES": "1", "VXM_S3U_BWW_MIN_TILES": "1", "VXM_FP32_ENGINE": engine},
               "conv_block_vs_oracle or conv_block_output_guard or unet_vs_oracle or vxm_dense_golden or collapsed_weights", files=("tests/test_gpu_parity.py",))
        _rerun({"VXM_S3_MIN_TILES": "1", "VXM_S3U": "0", "VXM_FP32_ENGINE": engine, "VXM_S3_UP": "1"},
               "unet_vs_oracle or collapsed_weights", files=("tests/test_gpu_parity.py",))
    _rerun({"VXM_S3_MIN_TILES": "1", "VXM_S3U_MIN_TILES": "1", "VXM_S3_PC": "1", "VXM_S3P_BLOCKS": "16"},
           "conv_block_vs_oracle or conv_block_output_guard or unet_vs_oracle or vxm_dense_golden", files=("tests/test_gpu_parity.py",))


def test_other_fp32_engines_at_full_size_in_subprocess():
    """The engines that are not the default keep their full-size coverage -- the adjoint identity of every full-resolution conv product
    and the whole headline step against the reference restatement on the host: VXM_FP32_ENGINE=native (the exact-fp32 MFMA kernels of
    conv_fwd.hip) and VXM_FP32_ENGINE=split (three bf16 pieces, the default of round 3)."""
    if os.environ.get("VXM_FP32_ENGINE") in ("native", "split", "bf16x3"):
        pytest.skip("already inside a non-default engine run")
    for engine in ("native", "split"):
        _rerun({"VXM_FP32_ENGINE": engine}, "(full_size_conv_adjoint_identity and not four_pairs) or full_size_train_step_vs_oracle_noise_pair",
               files=("tests/test_gpu_parity.py",), timeout=1500)


def test_packed_operator_cache_follows_reseated_and_invalidated_weights(VF):
    """the split engine caches its packed operators on the weight tensor.  `w.data = other` (no version bump) must be seen
    through the storage address in the key; a write into the SAME storage through `.data` is seen after `voxelmorph_amd.invalidate_packs`."""
    import voxelmorph_amd
    torch.manual_seed(3)
    vol, B = (8, 8, 16), 1
    x = torch.randn((B, 16) + vol, device="cuda")
    w = torch.nn.Parameter(torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.1)
    y0 = _s3_forward(VF, x, False, None, w, None, 1.0, 16, vol, B).clone()
    w.data = (w.data * 2.0).clone()                                   # re-seated storage, same version counter
    y1 = _s3_forward(VF, x, False, None, w, None, 1.0, 16, vol, B).clone()
    assert rel_l2(y1.cpu().numpy(), 2.0 * y0.cpu().numpy()) < 1e-6
    w.data.mul_(0.5)                                                  # same storage, same counter: the cache cannot know ...
    voxelmorph_amd.invalidate_packs([w])                              # ... until it is told
    y2 = _s3_forward(VF, x, False, None, w, None, 1.0, 16, vol, B)
    assert rel_l2(y2.cpu().numpy(), y0.cpu().numpy()) < 1e-6
