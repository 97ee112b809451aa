#!/usr/bin/env python
"""Developer check (not a test, not the product): accuracy of the backward-weight kernels AT FULL SIZE, where an accumulator sums
millions of products -- split kernel (bf16 pipe, fp32 MFMA accumulation) and fp32-MFMA kernel against an fp64 evaluation on the host
(27 shifted fp64 GEMMs).  Random and structured (smooth, non-zero-mean: cancellation-free) operands.

    python tools/bw_accuracy.py [--shape 160,192,224] [--c 32] [--cout 16]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def ref_fp64(x, dz):
    """gw[co, ci, kd, kh, kw] = sum_v dz[co, v] x[ci, v + tap - 1] in fp64 on the host"""
    C, D, H, W = x.shape
    xp = torch.zeros(C, D + 2, H + 2, W + 2, dtype=torch.float64)
    xp[:, 1:-1, 1:-1, 1:-1] = x.double()
    z = dz.double().reshape(dz.shape[0], -1)
    out = torch.empty(dz.shape[0], C, 3, 3, 3, dtype=torch.float64)
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                out[:, :, kd, kh, kw] = z @ xp[:, kd:kd + D, kh:kh + H, kw:kw + W].reshape(C, -1).T
    return out, z.sum(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="160,192,224")
    ap.add_argument("--c", type=int, default=32)
    ap.add_argument("--cout", type=int, default=16)
    args = ap.parse_args()
    from voxelmorph_amd.torch import functional as VF
    D, H, W = (int(s) for s in args.shape.split(","))
    V = D * H * W
    c, cout = args.c, args.cout
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    for kind in ("random N(0,1)", "non-zero mean (1 + 0.1 N): every product positive, no cancellation"):
        torch.manual_seed(0)
        x = torch.randn(1, c, D, H, W, device="cuda")
        dz = torch.randn(1, cout, D, H, W, device="cuda")
        if kind.startswith("non"):
            x, dz = 1.0 + 0.1 * x, 1.0 + 0.1 * dz
        ws = VF._Workspace(x.device)
        res = {}
        for eng in ("split", "native"):
            gw, gb = torch.empty(cout, c, 3, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
            keep = VF.FP32_ENGINE
            VF.FP32_ENGINE = eng
            VF.conv_bwd_weight(ws, x, c, c * V, False, None, 0, 0, dz, cout, gw, gb, 1, D, H, W)
            VF.FP32_ENGINE = keep
            res[eng] = (gw.cpu().double(), gb.cpu().double())
        ref, refb = ref_fp64(x[0].cpu(), dz[0].cpu())
        for eng, (gw, gb) in res.items():
            e = float((gw - ref).norm() / ref.norm())
            eb = float((gb - refb).norm() / refb.norm())
            bias = float(((gw - ref) / ref.abs().clamp_min(1e-30)).mean())
            print("%-70s %-7s gw rel-L2 %.2e (mean signed rel. error %+.2e)  gb rel-L2 %.2e" % (kind, eng, e, bias, eb), flush=True)


if __name__ == "__main__":
    main()
