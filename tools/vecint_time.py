#!/usr/bin/env python
"""Developer microbenchmark: VecInt (7 scaling-and-squaring steps) forward / backward on the half-resolution field of the headline
config (80x96x112), HIP-event timed; a near-zero field (random-init network) and a smooth field with displacements of several voxels
(trained network: the last steps have senders that are not "near").

    python tools/vecint_time.py [--iters 50] [--shape 80,96,112] [--batch 1] [--nsteps 7]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def timed(fn, iters):
    for _ in range(50):
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
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--shape", type=str, default="80,96,112")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--nsteps", type=int, default=7)
    ap.add_argument("--case", type=str, default="both", choices=("zero", "smooth", "both"))
    args = ap.parse_args()
    from voxelmorph_amd.torch import functional as VF
    call, ptr, stream = VF.call, VF.ptr, VF.stream
    shape = tuple(int(s) for s in args.shape.split(","))
    B, n = args.batch, args.nsteps
    D, H, W = shape
    V = D * H * W
    torch.manual_seed(0)
    low = torch.randn(B, 3, *(max(2, s // 16) for s in shape), device="cuda")
    smooth = torch.nn.functional.interpolate(low, size=shape, mode="trilinear", align_corners=True)
    cases = []
    if args.case in ("zero", "both"):
        cases.append(("near-zero field (std 1e-3)", 1e-3 * torch.randn(B, 3, *shape, device="cuda")))
    if args.case in ("smooth", "both"):
        for amp in (1.6, 3.0):
            cases.append(("smooth field, max |v| %.1f voxels" % float(amp * smooth.abs().max()), (amp * smooth).contiguous()))
    if args.case in ("smooth", "both"):
        # the first smooth field + two localised ~20-voxel bumps (verdict round 5, item 5).  Before round 6 every tile of the far pass walked the
        # batch maximum (~200 x its volume for a 20-voxel maximum); now a tile's radius is what the sender tiles around it need.  The second
        # line of the case passes the ABI-0.4 scratch size, which selects the global radius: the old behaviour, for comparison.
        zz, yy, xx = torch.meshgrid(*[torch.arange(s, dtype=torch.float32, device="cuda") for s in shape], indexing="ij")
        bumpy = (1.6 * smooth).contiguous().clone()
        for (fz, fy, fx, c, a) in ((0.2, 0.3, 0.25, 2, 20.0), (0.7, 0.6, 0.8, 0, -20.0)):
            bumpy[:, c] += a * torch.exp(-((zz - fz * D) ** 2 + (yy - fy * H) ** 2 + (xx - fx * W) ** 2) / (2 * 5.0 ** 2))
        cases.append(("the 5.2-voxel field + two 20-voxel bumps", bumpy))
        cases.append(("  the same, global radius (ABI 0.4 scratch)", bumpy))
    for name, vec in cases:
        gout = torch.randn_like(vec)
        steps = torch.empty((n,) + tuple(vec.shape), device="cuda")
        gvec = torch.empty_like(vec)
        work = torch.zeros(2 * vec.numel() + 128 if "global radius" in name else VF.vecint_work_elems(vec.shape), device="cuda")

        def fwd():
            call("vxm_vecint_fwd", ptr(vec), ptr(steps), B, D, H, W, n, stream())

        def bwd():
            call("vxm_vecint_bwd_ws", ptr(vec), ptr(steps), ptr(gout), ptr(gvec), ptr(work), work.numel() * 4, B, D, H, W, n, stream())
        fwd()
        t_f = timed(fwd, args.iters)
        t_b = timed(bwd, args.iters)
        print("%-40s fwd %7.1f us (%5.2f TB/s of 24 B/voxel/step) | bwd %7.1f us (%5.2f TB/s of 36 B/voxel/step) | max |out| %.2f"
              % (name, t_f, 24.0 * B * V * n / t_f / 1e6, t_b, 36.0 * B * V * n / t_b / 1e6, float(steps[n - 1].abs().max())), flush=True)


if __name__ == "__main__":
    main()
