#!/usr/bin/env python
"""Developer microbenchmark (not a test, not the product): the weight-gradient launches of the plain tensors of the headline step, with the
layouts the fused step uses (csrc/conv_s3.hip: k_s3_bww_pc, or k_s3_bwd_weight with VXM_S3_BW_PC=0), timed with HIP events.

    [VXM_S3_BW_PC=0] python tools/bw_pc_bench.py [--iters 5] [--shape 160,192,224] [--only rem0]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def timed(fn, iters):
    fn()
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
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--shape", type=str, default="160,192,224")
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--lib", type=str, default="", help="developer build (tools/build_exp.sh) to load instead of libvxm_hip.so")
    ap.add_argument("--dbg", type=str, default="", help="comma list of VXM_S3_DBG words (1: no loads, 2: no multiply phase, 4: no split / LDS writes; --lib build)")
    args = ap.parse_args()
    from voxelmorph_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from voxelmorph_amd.torch import functional as VF
    shape = tuple(int(s) for s in args.shape.split(","))
    IN0, IN1 = VF.S3_IN0_BLOCKED, VF.S3_IN1_BLOCKED
    cases = (("rem0-skip 16 x 32 (x planar, dz blocked)", 16, 32, 0, IN1), ("rem1 32 x 16 (both blocked)", 32, 16, 0, IN0 | IN1),
             ("rem2 16 x 16 (x blocked)", 16, 16, 0, IN0), ("enc1 16 x 32 (L1, planar)", 16, 32, 1, 0), ("dec3-skip 32 x 32 (L1, planar)", 32, 32, 1, 0),
             ("enc2 32 x 32 (L2)", 32, 32, 2, 0))
    print("VXM_S3_BW_PC=%s" % os.environ.get("VXM_S3_BW_PC", "(default: on)"))
    for name, c, cout, lvl, lay in cases:
        if args.only and args.only not in name:
            continue
        D, H, W = (s >> lvl for s in shape)
        V = D * H * W
        torch.manual_seed(1)
        x = torch.randn(1, c, D, H, W, device="cuda")
        dz = torch.randn(1, cout, D, H, W, device="cuda")
        xb = VF.to_blocked(x) if lay & IN0 else x
        zb = VF.to_blocked(dz) if lay & IN1 else dz
        gw, gb = torch.empty(cout, c, 3, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
        gw_n, gb_n = torch.empty_like(gw), torch.empty_like(gb)
        ws = VF._Workspace(x.device)
        t = timed(lambda: VF.s3_bwd_weight(ws, xb, c, c * V, zb, cout, gw, c, 0, gb, 1, D, H, W, lay=lay), args.iters)
        if args.dbg:
            res = {}
            for v in [0] + [int(w) for w in args.dbg.split(",")]:
                os.environ["VXM_S3_DBG"] = str(v)
                res[v] = timed(lambda: VF.s3_bwd_weight(ws, xb, c, c * V, zb, cout, gw, c, 0, gb, 1, D, H, W, lay=lay), args.iters)
            os.environ["VXM_S3_DBG"] = "0"
            print("    dbg: " + " | ".join("%d: %.3f" % kv for kv in res.items()), flush=True)
            VF.s3_bwd_weight(ws, xb, c, c * V, zb, cout, gw, c, 0, gb, 1, D, H, W, lay=lay)
        keep = VF.FP32_ENGINE
        VF.FP32_ENGINE = "native"
        VF.conv_bwd_weight(ws, x, c, c * V, False, None, 0, 0, dz, cout, gw_n, gb_n, 1, D, H, W)
        VF.FP32_ENGINE = keep
        ew = float((gw.double() - gw_n.double()).norm() / gw_n.double().norm())
        eb = float((gb.double() - gb_n.double()).norm() / gb_n.double().norm())
        gf = 2.0 * 27 * c * cout * V / 1e9
        print("%-44s %7.1f GFLOP  %.3f ms = %6.1f TF-eq | rel-L2 vs fp32-MFMA: gw %.2e gb %.2e" % (name, gf, t, gf / t, ew, eb), flush=True)
        del x, dz, xb, zb


if __name__ == "__main__":
    main()
