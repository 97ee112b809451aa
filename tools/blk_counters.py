#!/usr/bin/env python
"""Developer tool: run ONE full-resolution launch shape of a split kernel a few times, planar or channel-blocked, so that rocprofv3 --pmc
can attribute memory-system counters to it.

    rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum ... -- python tools/blk_counters.py {fwd|bww} {planar|blocked}
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from voxelmorph_amd import _lib  # noqa: E402
if os.environ.get("BLK_LIB"):                       # a developer build (tools/build_exp.sh) instead of libvxm_hip.so
    _lib.LIB_PATH = os.path.abspath(os.environ["BLK_LIB"])
from voxelmorph_amd.torch import functional as VF  # noqa: E402

op, layout = sys.argv[1], sys.argv[2]
blk = layout == "blocked"
D, H, W = 160, 192, 224
V, B = D * H * W, 1
torch.manual_seed(0)
if op == "fwd":                      # rem1 forward 32 -> 16 (k_s3p_conv)
    c0, cout = 32, 16
    x = torch.randn(B, c0, D, H, W, device="cuda")
    w = torch.randn(cout, c0, 3, 3, 3, device="cuda") / (27 * c0) ** 0.5
    bias = torch.randn(cout, device="cuda")
    wp = VF.s3_pack(w, False, 0, c0, c0)
    y = torch.empty(B, cout, D, H, W, device="cuda")
    xin = VF.to_blocked(x) if blk else x
    lay = (VF.S3_IN0_BLOCKED | VF.S3_OUT_BLOCKED) if blk else 0
    fn = lambda: VF.s3_launch(xin, c0, c0 * V, False, None, 0, 0, wp, bias, y, cout * V, cout, 0.2, None, 0, 1.0, B, D, H, W, lay=lay)
else:                                # rem1 backward-weight 32 -> 16 (k_s3_bwd_weight)
    c, cout = 32, 16
    x, dz = torch.randn(B, c, D, H, W, device="cuda"), torch.randn(B, cout, D, H, W, device="cuda")
    ws = VF._Workspace(x.device)
    gw, gb = torch.empty(cout, c, 3, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
    xin, zin = (VF.to_blocked(x), VF.to_blocked(dz)) if blk else (x, dz)
    lay = (VF.S3_IN0_BLOCKED | VF.S3_IN1_BLOCKED) if blk else 0
    fn = lambda: VF.s3_bwd_weight(ws, xin, c, c * V, zin, cout, gw, c, 0, gb, B, D, H, W, lay=lay)
for _ in range(5):
    fn()
torch.cuda.synchronize()
if os.environ.get("BLK_TIME"):
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(20):
        fn()
    e_.record()
    torch.cuda.synchronize()
    print("%s %s dbg=%s: %.4f ms" % (op, layout, os.environ.get("VXM_S3_DBG", "0"), s_.elapsed_time(e_) / 20))
