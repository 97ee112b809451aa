"""Developer check + timing of the channel-blocked operand variants of the split kernels against their planar launches (bit-exact);
BLK_TIME=1 adds timings at 160x192x224.  The same comparisons run as tests in tests/test_gpu_s3.py."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from voxelmorph_amd.torch import functional as VF
torch.manual_seed(0)
def run(c0, cout, shape, flip, lay, B=1):
    D, H, W = shape; V = D * H * W
    x = torch.randn(B, c0, D, H, W, device="cuda")
    w = torch.randn(*((c0, cout) if flip else (cout, c0)), 3, 3, 3, device="cuda") / (27 * c0) ** 0.5
    bias = None if flip else torch.randn(cout, device="cuda")
    mask = torch.randn(B, cout, D, H, W, device="cuda") if flip else None
    wp = VF.s3_pack(w, flip, 0, cout if flip else c0, c0)
    y0 = torch.empty(B, cout, D, H, W, device="cuda")
    VF.s3_launch(x, c0, c0 * V, False, None, 0, 0, wp, bias, y0, cout * V, cout, 1.0 if flip else 0.2, mask, cout * V, 0.2, B, D, H, W)
    xin = VF.to_blocked(x) if lay & VF.S3_IN0_BLOCKED else x
    mk = (VF.to_blocked(mask) if lay & VF.S3_OUT_BLOCKED else mask) if mask is not None else None
    y1 = torch.full((B, cout, D, H, W), float("nan"), device="cuda")
    VF.s3_launch(xin, c0, c0 * V, False, None, 0, 0, wp, bias, y1, cout * V, cout, 1.0 if flip else 0.2, mk, cout * V, 0.2, B, D, H, W, lay=lay)
    if lay & VF.S3_OUT_BLOCKED: y1 = VF.from_blocked(y1)
    return torch.equal(y0, y1), float((y0 - y1).abs().max())
for shape in ((8, 16, 32), (9, 21, 37), (16, 24, 48)):
    for c0, cout in ((16, 16), (32, 16), (16, 32), (8, 24)):
        for flip in (False, True):
            for lay in (VF.S3_IN0_BLOCKED, VF.S3_OUT_BLOCKED, VF.S3_IN0_BLOCKED | VF.S3_OUT_BLOCKED):
                for B in (1, 2):
                    ok, d = run(c0, cout, shape, flip, lay, B)
                    if not ok: print("MISMATCH", shape, c0, cout, flip, hex(lay), B, d)
print("done", os.environ.get("VXM_S3_PC"))

# ---- backward-weight: blocked x / dz operands against the planar launch (bit-exact), then timing at full size
def bww(c, cout, shape, lay, B=1, time_it=False):
    D, H, W = shape; V = D * H * W
    x = torch.randn(B, c, D, H, W, device="cuda"); dz = torch.randn(B, cout, D, H, W, device="cuda")
    ws = VF._Workspace(x.device)
    g0, b0 = torch.empty(cout, c, 3, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
    g1, b1 = torch.full_like(g0, float("nan")), torch.full_like(b0, float("nan"))
    VF.s3_bwd_weight(ws, x, c, c * V, dz, cout, g0, c, 0, b0, B, D, H, W)
    xb = VF.to_blocked(x) if lay & VF.S3_IN0_BLOCKED else x
    zb = VF.to_blocked(dz) if lay & VF.S3_IN1_BLOCKED else dz
    VF.s3_bwd_weight(ws, xb, c, c * V, zb, cout, g1, c, 0, b1, B, D, H, W, lay=lay)
    ok = torch.equal(g0, g1) and torch.equal(b0, b1)
    t = None
    if time_it:
        def timed(fn, n=10):
            fn(); fn(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(n): fn()
            e.record(); torch.cuda.synchronize()
            return s.elapsed_time(e) / n
        t = (timed(lambda: VF.s3_bwd_weight(ws, x, c, c * V, dz, cout, g0, c, 0, b0, B, D, H, W)),
             timed(lambda: VF.s3_bwd_weight(ws, xb, c, c * V, zb, cout, g1, c, 0, b1, B, D, H, W, lay=lay)))
    return ok, float((g0 - g1).abs().max()), t
for shape in ((8, 16, 32), (10, 20, 38), (16, 24, 64)):
    for c, cout in ((16, 16), (32, 16), (16, 32)):
        for lay in (VF.S3_IN0_BLOCKED, VF.S3_IN1_BLOCKED, VF.S3_IN0_BLOCKED | VF.S3_IN1_BLOCKED):
            for B in (1, 2):
                ok, d, _ = bww(c, cout, shape, lay, B)
                if not ok: print("BWW MISMATCH", shape, c, cout, hex(lay), B, d)
if os.environ.get("BLK_TIME"):
    for c, cout in ((32, 16), (16, 16)):
        for lay in (VF.S3_IN0_BLOCKED, VF.S3_IN1_BLOCKED, VF.S3_IN0_BLOCKED | VF.S3_IN1_BLOCKED):
            ok, d, t = bww(c, cout, (160, 192, 224), lay, 1, True)
            print("bww %d->%d lay %s ok %s planar %.3f ms blocked %.3f ms" % (c, cout, hex(lay), ok, t[0], t[1]))
print("bww done")

# ---- the cat([upsample, skip]) kernels: blocked output of the forward, blocked dz of the backward-data onto the low-resolution tensor and of
# the weight gradient of the upsampled segment
def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def s3u_all(c0, c1, cout, shape, B=1, time_it=False):
    D, H, W = shape; V = D * H * W
    x0 = torch.randn(B, c0, D // 2, H // 2, W // 2, device="cuda"); x1 = torch.randn(B, c1, D, H, W, device="cuda")
    w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
    bias = torch.randn(cout, device="cuda")
    wp = VF.s3u_pack(w, c0, c1)
    y0 = torch.empty(B, cout, D, H, W, device="cuda"); y1 = torch.full_like(y0, float("nan"))
    f0 = lambda: VF.s3u_launch(x0, c0, x0[0].numel(), x1, c1, c1 * V, wp, bias, y0, cout * V, cout, 0.2, B, D, H, W)
    f1 = lambda: VF.s3u_launch(x0, c0, x0[0].numel(), x1, c1, c1 * V, wp, bias, y1, cout * V, cout, 0.2, B, D, H, W, lay=VF.S3_OUT_BLOCKED)
    f0(); f1()
    res = {"fwd": torch.equal(y0, VF.from_blocked(y1))}
    dz = torch.randn(B, cout, D, H, W, device="cuda"); dzb = VF.to_blocked(dz)
    act = torch.randn(B, c0, D // 2, H // 2, W // 2, device="cuda")
    g0 = torch.empty_like(act); g1 = torch.full_like(act, float("nan"))
    d0 = lambda: VF.s3u_bwd_low(dz, cout, w, c0, c0 + c1, g0, act, 0.2, B, D, H, W)
    d1 = lambda: VF.s3u_bwd_low(dzb, cout, w, c0, c0 + c1, g1, act, 0.2, B, D, H, W, lay=VF.S3_IN0_BLOCKED)
    d0(); d1()
    res["dlow"] = torch.equal(g0, g1)
    ws = VF._Workspace(dz.device)
    gw0 = torch.zeros(cout, c0 + c1, 3, 3, 3, device="cuda"); gw1 = torch.zeros_like(gw0)
    b0 = lambda: VF.s3u_bwd_weight(ws, x0, c0, x0[0].numel(), dz, cout, gw0, c0 + c1, B, D, H, W)
    b1 = lambda: VF.s3u_bwd_weight(ws, x0, c0, x0[0].numel(), dzb, cout, gw1, c0 + c1, B, D, H, W, lay=VF.S3_IN1_BLOCKED)
    if c0 in (16, 32) and cout % 16 == 0 and W % 4 == 0 and D % 2 == 0 and H % 2 == 0:
        b0(); b1()
        res["bww"] = torch.equal(gw0, gw1)
    if time_it:
        res["t"] = {k: (round(timed(a), 3), round(timed(b), 3)) for k, (a, b) in dict(fwd=(f0, f1), dlow=(d0, d1), bww=(b0, b1)).items()}
    return res
for shape in ((8, 8, 32), (10, 12, 36), (16, 24, 64)):
    for c0, c1, cout in ((32, 16, 32), (16, 16, 16), (32, 32, 32), (16, 8, 24)):
        for B in (1, 2):
            r = s3u_all(c0, c1, cout, shape, B)
            if not all(v for v in r.values()): print("S3U MISMATCH", shape, c0, c1, cout, B, r)
print("s3u done")
if os.environ.get("BLK_TIME"):
    print("s3u rem0 planar / blocked ms:", s3u_all(32, 16, 32, (160, 192, 224), 1, True))
    def conv_t(c0, cout, flip, lay):
        D, H, W = 160, 192, 224; V = D * H * W; B = 1
        x = torch.randn(B, c0, D, H, W, device="cuda")
        w = torch.randn(*((c0, cout) if flip else (cout, c0)), 3, 3, 3, device="cuda") / (27 * c0) ** 0.5
        bias = None if flip else torch.randn(cout, device="cuda")
        mask = torch.randn(B, cout, D, H, W, device="cuda") if flip else None
        wp = VF.s3_pack(w, flip, 0, cout if flip else c0, c0)
        y = torch.empty(B, cout, D, H, W, device="cuda")
        sl = 1.0 if flip else 0.2
        return (timed(lambda: VF.s3_launch(x, c0, c0 * V, False, None, 0, 0, wp, bias, y, cout * V, cout, sl, mask, cout * V, 0.2, B, D, H, W)),
                timed(lambda: VF.s3_launch(x, c0, c0 * V, False, None, 0, 0, wp, bias, y, cout * V, cout, sl, mask, cout * V, 0.2, B, D, H, W, lay=lay)))
    for name, c0, cout, flip, lay in (("rem1 fwd 32->16 in+out", 32, 16, False, 0x500), ("rem2 fwd 16->16 in", 16, 16, False, 0x100),
                                      ("rem2 bwd-data 16->16 out+mask", 16, 16, True, 0x400), ("rem1 bwd-data 16->32 in+out+mask", 16, 32, True, 0x500),
                                      ("rem0 bwd-skip 32->16 in (no mask in the engine)", 32, 16, False, 0x100)):
        t = conv_t(c0, cout, flip, lay)
        print("%-50s planar %.3f ms blocked %.3f ms" % (name, t[0], t[1]))
