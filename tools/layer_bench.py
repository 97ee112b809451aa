#!/usr/bin/env python
"""Standalone timing of the HBM-bound layers of the path at the benchmark shape: SpatialTransformer,
VecInt, ResizeTransform, NCC, Grad — forward and backward through the C ABI, HIP events on the launch stream, achieved
GB/s on the ALGORITHMIC bytes of SURVEY §8d against the 8 TB/s HBM peak.  Flows: all-zero (best-case gather locality),
a smooth field with max |v| = 5 voxels (Gaussian-filtered noise, sigma = 8 voxels, seed 0: the representative case) and
white noise of the same amplitude (worst case).

    python tools/layer_bench.py [--iters 20] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from voxelmorph_amd.torch import functional as VF  # noqa: E402

HBM_PEAK = 8000.0
FULL = (160, 192, 224)
HALF = (80, 96, 112)


def smooth_field(shape, amp, sigma, seed):
    """Gaussian-filtered white noise scaled to max |v| = amp (separable 1-D convolutions on the device: set-up only)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    v = torch.randn((1, 3) + tuple(shape), device="cuda", generator=g)
    r = int(3 * sigma)
    k = torch.exp(-0.5 * (torch.arange(-r, r + 1, device="cuda", dtype=torch.float32) / sigma) ** 2)
    k = (k / k.sum())
    for ax in range(3):
        kshape = [1, 1, 1, 1, 1]
        kshape[2 + ax] = 2 * r + 1
        pad = [0, 0, 0, 0, 0, 0]
        pad[2 * (2 - ax)] = pad[2 * (2 - ax) + 1] = r
        v = F.conv3d(F.pad(v.transpose(0, 1), pad, mode="replicate"), k.view(kshape)).transpose(0, 1)
    return (v * (amp / v.norm(dim=1).max())).contiguous()


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", type=str, default="")
    args = ap.parse_args()
    import voxelmorph_amd as vxm
    from voxelmorph_amd._lib import call, ptr, stream
    V, Vh = FULL[0] * FULL[1] * FULL[2], HALF[0] * HALF[1] * HALF[2]
    rows = []

    def report(name, flow_kind, ms, nbytes):
        gbs = nbytes / ms / 1e6
        rows.append(dict(layer=name, flow=flow_kind, ms=ms, gbs=gbs, frac_of_hbm_peak=gbs / HBM_PEAK))
        print("%-28s %-10s %8.3f ms  %7.0f GB/s  %5.1f %% of 8 TB/s" % (name, flow_kind, ms, gbs, 100 * gbs / HBM_PEAK), flush=True)

    torch.manual_seed(1234)
    img = torch.rand((1, 1) + FULL, device="cuda")
    flows = {"zero": torch.zeros((1, 3) + FULL, device="cuda"), "smooth5": smooth_field(FULL, 5.0, 8.0, 0)}
    flows["noise5"] = (torch.randn((1, 3) + FULL, device="cuda") * (5.0 / 3 ** 0.5 / 3)).clamp_(-5, 5)
    out, gflow = torch.empty_like(img), torch.empty((1, 3) + FULL, device="cuda")
    gout = torch.randn_like(img)
    for kind, fl in flows.items():
        ms = timed(lambda: call("vxm_warp3d_fwd", ptr(img), ptr(fl), ptr(out), 1, 1, *FULL, 0, stream()), args.iters)
        report("SpatialTransformer fwd", kind, ms, 4.0 * V * (2 + 3))
        ms = timed(lambda: call("vxm_warp3d_bwd", ptr(img), ptr(fl), ptr(gout), None, ptr(gflow), 1, 1, *FULL, 0, stream()), args.iters)
        report("SpatialTransformer bwd", kind, ms, 4.0 * V * (2 + 6))
    # VecInt on half-resolution velocity fields (what the network integrates: the resized flow, |v| up to 2.5 voxels)
    n = 3 * Vh
    for kind, amp in (("zero", 0.0), ("smooth2.5", 2.5)):
        vel = smooth_field(HALF, amp, 4.0, 1) if amp > 0 else torch.zeros((1, 3) + HALF, device="cuda")
        steps = torch.empty((7, 1, 3) + HALF, device="cuda")
        gv, work, g7 = torch.empty_like(vel), torch.empty(VF.vecint_work_elems(vel.shape), device="cuda"), torch.randn_like(vel)
        ms = timed(lambda: call("vxm_vecint_fwd", ptr(vel), ptr(steps), 1, *HALF, 7, stream()), args.iters)
        report("VecInt(7) fwd", kind, ms, 7 * 24.0 * Vh)
        ms = timed(lambda: call("vxm_vecint_bwd_ws", ptr(vel), ptr(steps), ptr(g7), ptr(gv), ptr(work), work.numel() * 4, 1, *HALF, 7, stream()), args.iters)
        report("VecInt(7) bwd", kind, ms, 7 * 36.0 * Vh)
    # ResizeTransform down (vel_resize 2) and up (1/2)
    xf, xh = torch.randn((1, 3) + FULL, device="cuda"), torch.randn((1, 3) + HALF, device="cuda")
    of, oh = torch.empty_like(xf), torch.empty_like(xh)
    nb = 12.0 * (V + Vh)
    report("ResizeTransform down fwd", "-", timed(lambda: call("vxm_resize3d_fwd", ptr(xf), ptr(oh), 1, 3, *FULL, *HALF, 0.5, stream()), args.iters), nb)
    report("ResizeTransform down bwd", "-", timed(lambda: call("vxm_resize3d_bwd", ptr(xh), ptr(of), 1, 3, *FULL, *HALF, 0.5, stream()), args.iters), nb)
    report("ResizeTransform up fwd", "-", timed(lambda: call("vxm_resize3d_fwd", ptr(xh), ptr(of), 1, 3, *HALF, *FULL, 2.0, stream()), args.iters), nb)
    report("ResizeTransform up bwd", "-", timed(lambda: call("vxm_resize3d_bwd", ptr(xf), ptr(oh), 1, 3, *HALF, *FULL, 2.0, stream()), args.iters), nb)
    # losses
    J = torch.rand_like(img).requires_grad_()
    ncc, grad = vxm.losses.NCC().loss, vxm.losses.Grad("l2", loss_mult=2).loss
    report("NCC fwd+bwd (9^3)", "-", timed(lambda: ncc(img, J).backward(), args.iters), 20.0 * V)
    pf = torch.randn((1, 3) + HALF, device="cuda", requires_grad=True)
    report("Grad fwd+bwd", "-", timed(lambda: grad(None, pf).backward(), args.iters), 24.0 * Vh)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
