#!/usr/bin/env python
"""Developer timing (not a test): what does the fp32 LeakyReLU mask cost the backward-data launches of the full-resolution layers?  Same launch
with and without the mask tensor (channel-blocked operands as in the fused step), HIP events."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voxelmorph_amd.torch import functional as VF

def timed(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

D, H, W = 160, 192, 224
V = D * H * W
for name, c0, cout in (("rem1 bwd-data dz16 -> dx32", 16, 32), ("rem2 bwd-data dz16 -> dx16", 16, 16)):
    torch.manual_seed(0)
    x = VF.to_blocked(torch.randn(1, c0, D, H, W, device="cuda"))
    w = torch.randn(c0, cout, 3, 3, 3, device="cuda") / (27 * c0) ** 0.5
    mask = VF.to_blocked(torch.randn(1, cout, D, H, W, device="cuda"))
    wp = VF.s3_pack(w, True, 0, cout, c0)
    y = torch.empty(1, cout, D, H, W, device="cuda")
    lay = VF.S3_IN0_BLOCKED | VF.S3_OUT_BLOCKED
    res = []
    for _ in range(2):
        for mk in (mask, None):
            res.append(timed(lambda: VF.s3_launch(x, c0, c0 * V, False, None, 0, 0, wp, None, y, cout * V, cout, 1.0, mk, cout * V, 0.2, 1, D, H, W, lay=lay)))
    print("%-30s with mask %.3f / %.3f ms | without %.3f / %.3f ms" % (name, res[0], res[2], res[1], res[3]), flush=True)
    del x, mask, y
