#!/usr/bin/env python
"""Developer check (not a test, not the product): the conv launches of the U-Net's coarse levels (<= 20x24x28) through the C ABI against an
fp64 evaluation on the host, with HIP-event timings.  VXM_CONV_SMALL_MAX_BLOCKS=0 gives the kernel k_conv3d_k3_sm replaces.

    python tools/small_conv_check.py [--iters 50]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# name, (c0, up0, c1), cout, (D, H, W), slope, with mask
CASES = [
    ("enc3", (32, False, 0), 32, (20, 24, 28), 0.2, False),
    ("dec0", (32, False, 0), 32, (10, 12, 14), 0.2, False),
    ("dec1", (32, True, 32), 32, (20, 24, 28), 0.2, False),
    ("wide", (32, False, 0), 64, (20, 24, 28), 0.2, True),
    ("odd", (40, False, 0), 24, (9, 11, 13), 0.2, True),
    ("odd2", (24, True, 16), 40, (6, 10, 18), 1.0, False),
]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    from voxelmorph_amd.torch import functional as VF
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    B = args.batch
    worst = 0.0
    for name, (c0, up0, c1), cout, (D, H, W), slope, with_mask in CASES:
        V = D * H * W
        cin = c0 + c1
        x0 = torch.randn(B, c0, *((D // 2, H // 2, W // 2) if up0 else (D, H, W)), device=dev)
        x1 = torch.randn(B, c1, D, H, W, device=dev) if c1 else None
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * (1.0 / (27 * cin) ** 0.5)
        b = torch.randn(cout, device=dev)
        y = torch.empty(B, cout, D, H, W, device=dev)
        dz = torch.randn(B, cout, D, H, W, device=dev)
        gx = torch.empty(B, cin, D, H, W, device=dev)
        mask = torch.randn(B, cin, D, H, W, device=dev) if with_mask else None

        def fwd():
            VF.conv_forward(x0, c0, x0[0].numel(), up0, x1, c1, x1[0].numel() if x1 is not None else 0, w, b, y, cout * V, cout, slope, B, D, H, W)

        def bwd():
            VF.conv_bwd_data(dz, cout, w, gx, cin, mask, 0.2, B, D, H, W)

        fwd(); bwd()
        torch.cuda.synchronize()
        xin = x0.cpu().double()
        if up0:
            xin = F.interpolate(xin, scale_factor=2, mode="nearest")
        if x1 is not None:
            xin = torch.cat([xin, x1.cpu().double()], 1)
        pre = F.conv3d(xin, w.cpu().double(), b.cpu().double(), padding=1)
        ref = F.leaky_relu(pre, slope) if slope != 1.0 else pre
        gref = F.conv_transpose3d(dz.cpu().double(), w.cpu().double(), padding=1)
        if mask is not None:
            gref = gref * torch.where(mask.cpu() > 0, 1.0, 0.2).double()

        def rel(a, r):
            return float((a.cpu().double() - r).norm() / r.norm())
        ey, eg = rel(y, ref), rel(gx, gref)
        worst = max(worst, ey, eg)
        print("%-5s %2d%s+%2d -> %2d  %2dx%2dx%2d  fwd %6.1f us  err %.1e | bwd-data %6.1f us  err %.1e"
              % (name, c0, "^" if up0 else " ", c1, cout, D, H, W, timed(fwd, args.iters), ey, timed(bwd, args.iters), eg), flush=True)
    print("worst rel-L2 %.2e" % worst)
    return 0 if worst < 1e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
