#!/usr/bin/env python
"""Developer microbenchmark (not a test, not the product): the forward / backward-data conv products of the full-resolution layers
of the default VxmDense U-Net at 160x192x224 on the split-fp32 kernel (csrc/conv_s3.hip) and on the exact-fp32 MFMA kernels
(csrc/conv_fwd.hip), timed with HIP events; fp32-equivalent TFLOP/s, and the difference between the two results.

    [VXM_S3_CB=2] [VXM_S3_NCT=1] python tools/s3_bench.py [--iters 5] [--shape 160,192,224] [--batch 1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# name, c0, up0, c1, cout, level, flip (operator = adjoint of a layer with `c0` outputs... see below), mask
# forward operators: x [c0 (+c1)] -> cout.  backward-data operators are forward launches with flipped packs: dz [c0] -> cout.
OPS = [
    ("rem1 fwd      32->16", 32, False, 0, 16, 0, False),
    ("rem2 fwd      16->16", 16, False, 0, 16, 0, False),
    ("rem2 bwd-data 16->16", 16, False, 0, 16, 0, True),
    ("rem1 bwd-data 16->32", 16, False, 0, 32, 0, True),
    ("rem0 bwd-skip 32->16", 32, False, 0, 16, 0, True),
    ("enc1 fwd      16->32 (L1)", 16, False, 0, 32, 1, False),
    ("enc1 bwd-data 32->16 (L1)", 32, False, 0, 16, 1, True),
    ("rem0 fwd 32^+16->32", 32, True, 16, 32, 0, False),
    ("dec3 fwd 32^+32->32 (L1)", 32, True, 32, 32, 1, False),
]


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
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--json", type=str, default="")
    ap.add_argument("--lib", type=str, default="", help="developer build to load instead of libvxm_hip.so (tools/build_exp.sh)")
    ap.add_argument("--dbg", type=str, default="", help="comma list of VXM_S3_DBG words to time each forward / backward-data row with (--lib build)")
    args = ap.parse_args()
    from voxelmorph_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from voxelmorph_amd.torch import functional as VF
    shape = tuple(int(s) for s in args.shape.split(","))
    B = args.batch
    torch.manual_seed(0)
    rows = []
    print("instance for 16 / 32 output channels: %d / %d (10 NCT + CB)" % (_lib.lib().vxm_conv3d_k3_s3_variant(16), _lib.lib().vxm_conv3d_k3_s3_variant(32)))
    for name, c0, up0, c1, cout, lvl, flip in OPS:
        if args.only and args.only not in name:
            continue
        D, H, W = (s >> lvl for s in shape)
        V = D * H * W
        x0 = torch.randn(B, c0, *((D // 2, H // 2, W // 2) if up0 else (D, H, W)), device="cuda")
        x1 = torch.randn(B, c1, D, H, W, device="cuda") if c1 else None
        cin = c0 + c1
        # a weight tensor whose operator on (cin -> cout) is what we time: forward reads w[cout][cin], the adjoint reads w[cin][cout]
        w = torch.randn(*((cin, cout) if flip else (cout, cin)), 3, 3, 3, device="cuda") / (27 * cin) ** 0.5
        bias = None if flip else torch.randn(cout, device="cuda")
        mask = torch.randn(B, cout, D, H, W, device="cuda") if flip else None
        y_s3 = torch.empty(B, cout, D, H, W, device="cuda")
        y_nat = torch.empty_like(y_s3)
        slope = 1.0 if flip else 0.2
        wp_s3 = VF.s3_pack(w, flip, 0, cout if flip else cin, cin if flip else c0)

        keep = VF.FP32_ENGINE

        def run_s3():
            VF.FP32_ENGINE = keep                  # the piece scheme follows the engine at call time: the one wp_s3 was packed for
            VF.s3_launch(x0, c0, x0[0].numel(), up0, x1, c1, c1 * V, wp_s3, bias, y_s3, cout * V, cout, slope, mask, cout * V, 0.2, B, D, H, W)

        if up0 and not flip:
            def run_nat():
                VF.FP32_ENGINE = "native"
                VF.conv_forward(x0, c0, x0[0].numel(), True, x1, c1, c1 * V, w, bias, y_nat, cout * V, cout, slope, B, D, H, W)
                VF.FP32_ENGINE = keep
        else:
            wp_nat = VF.pack_weights(w, flip, 0, cout if flip else cin)

            def run_nat():
                VF.conv_launch(x0, c0, x0[0].numel(), up0, x1, c1, c1 * V, wp_nat, bias, y_nat, cout * V, cout, slope, mask, cout * V, 0.2, B, D, H, W)
        ok = bool(_lib.lib().vxm_conv3d_k3_s3_ok(c0, c1, cout, B, D, H, W))
        t_s3 = timed(run_s3, args.iters)
        t_nat = timed(run_nat, args.iters)
        VF.FP32_ENGINE = keep
        if args.dbg:                                       # timing experiments (results wrong): alternate with the plain kernel, two rounds
            words = [int(v) for v in args.dbg.split(",")]
            res = {v: [] for v in [0] + words}
            for _ in range(2):
                for v in [0] + words:
                    os.environ["VXM_S3_DBG"] = str(v)
                    res[v].append(timed(run_s3, args.iters))
            os.environ["VXM_S3_DBG"] = "0"
            print("    dbg: " + " | ".join("%d: %s" % (v, "/".join("%.3f" % t for t in ts)) for v, ts in res.items()), flush=True)
            run_s3()
        gf = 2.0 * 27 * cin * cout * B * V / 1e9
        diff = float((y_s3.double() - y_nat.double()).norm() / y_nat.double().norm())
        row = dict(op=name, gflop=gf, s3_ms=t_s3, s3_tflops=gf / t_s3, native_ms=t_nat, native_tflops=gf / t_nat, rel_l2_s3_vs_native=diff, s3_eligible=ok)
        rows.append(row)
        print("%-28s %7.1f GFLOP | split %.3f ms = %6.1f TF-eq | fp32-MFMA %.3f ms = %6.1f TF | x%.2f | rel-L2(split, fp32) %.2e"
              % (name, gf, t_s3, gf / t_s3, t_nat, gf / t_nat, t_nat / t_s3, diff), flush=True)
        del x0, x1, y_s3, y_nat, mask
    # cat([upsample(x0), x1]) forwards: the split + collapsed kernel (conv_s3u.hip) against the collapsed fp32-MFMA kernel (t8u)
    for name, c0, c1, cout, lvl in (("rem0 fwd 32^+16->32 s3u", 32, 16, 32, 0), ("dec3 fwd 32^+32->32 s3u (L1)", 32, 32, 32, 1),
                                    ("dec2 fwd 32^+32->32 s3u (L2)", 32, 32, 32, 2)):
        if args.only and args.only not in name:
            continue
        D, H, W = (s >> lvl for s in shape)
        V = D * H * W
        x0 = torch.randn(B, c0, D // 2, H // 2, W // 2, device="cuda")
        x1 = torch.randn(B, c1, D, H, W, device="cuda")
        w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
        bias = torch.randn(cout, device="cuda")
        y_s3, y_nat = torch.empty(B, cout, D, H, W, device="cuda"), torch.empty(B, cout, D, H, W, device="cuda")
        wp = VF.s3u_pack(w, c0, c1)
        keep = VF.FP32_ENGINE

        def run_s3():
            VF.s3u_launch(x0, c0, x0[0].numel(), x1, c1, c1 * V, wp, bias, y_s3, cout * V, cout, 0.2, B, D, H, W)

        def run_nat():
            VF.FP32_ENGINE = "native"
            VF.conv_forward(x0, c0, x0[0].numel(), True, x1, c1, c1 * V, w, bias, y_nat, cout * V, cout, 0.2, B, D, H, W)
            VF.FP32_ENGINE = keep
        t_s3, t_nat = timed(run_s3, args.iters), timed(run_nat, args.iters)
        if args.dbg:                                       # timing experiments of k_s3u_conv (SU_DBG in csrc/conv_s3u.hip; --lib build), blocked output as in the step
            y_blk = torch.empty_like(y_s3)
            res = {}
            for v in [0] + [int(v) for v in args.dbg.split(",")]:
                os.environ["VXM_S3_DBG"] = str(v)
                res[v] = timed(lambda: VF.s3u_launch(x0, c0, x0[0].numel(), x1, c1, c1 * V, wp, bias, y_blk, cout * V, cout, 0.2, B, D, H, W, lay=VF.S3_OUT_BLOCKED), args.iters)
            os.environ["VXM_S3_DBG"] = "0"
            print("    dbg (blocked output): " + " | ".join("%d: %.3f" % kv for kv in res.items()), flush=True)
            del y_blk
        gf = 2.0 * (8 * c0 + 27 * c1) * cout * B * V / 1e9
        diff = float((y_s3.double() - y_nat.double()).norm() / y_nat.double().norm())
        rows.append(dict(op=name, gflop_executed=gf, s3_ms=t_s3, s3_tflops=gf / t_s3, native_ms=t_nat, native_tflops=gf / t_nat, rel_l2_s3_vs_native=diff))
        print("%-34s %7.1f GFLOP executed | split+collapsed %.3f ms = %6.1f TF-eq | fp32-MFMA collapsed %.3f ms = %6.1f TF | x%.2f | rel-L2 %.2e"
              % (name, gf, t_s3, gf / t_s3, t_nat, gf / t_nat, t_nat / t_s3, diff), flush=True)
        del x0, x1, y_s3, y_nat
    # backward-data of the upsampled segment onto the low-resolution tensor: k_s3u_dlow against k_conv3d_k3_dlow
    for name, c0, c1, cout, lvl in (("rem0 dlow 32^ <- 32 s3u", 32, 16, 32, 0), ("dec3 dlow 32^ <- 32 s3u (L1)", 32, 32, 32, 1)):
        if args.only and args.only not in name:
            continue
        D, H, W = (s >> lvl for s in shape)
        V = D * H * W
        if not VF.s3u_bwd_low_route(c0, cout, B, D, H, W):
            continue
        dz = torch.randn(B, cout, D, H, W, device="cuda")
        act = torch.randn(B, c0, D // 2, H // 2, W // 2, device="cuda")
        w = torch.randn(cout, c0 + c1, 3, 3, 3, device="cuda") / (27 * (c0 + c1)) ** 0.5
        g_s3, g_nat = torch.empty_like(act), torch.empty_like(act)
        wpk = torch.empty(_lib.lib().vxm_conv3d_k3_up_bwd_low_packed_elems(c0, cout), device="cuda")

        def run_s3():
            VF.s3u_bwd_low(dz, cout, w, c0, c0 + c1, g_s3, act, 0.2, B, D, H, W)

        def run_nat():
            VF.call("vxm_conv3d_k3_up_bwd_low", VF.ptr(dz), cout * V, cout, VF.ptr(w), c0, c0 + c1, VF.ptr(wpk), VF.ptr(g_nat), c0 * (V // 8),
                    VF.ptr(act), c0 * (V // 8), 0.2, B, D, H, W, VF.stream())
        t_s3, t_nat = timed(run_s3, args.iters), timed(run_nat, args.iters)
        gf = 2.0 * 8 * c0 * cout * B * V / 1e9
        diff = float((g_s3.double() - g_nat.double()).norm() / g_nat.double().norm())
        rows.append(dict(op=name, gflop_executed=gf, s3_ms=t_s3, s3_tflops=gf / t_s3, native_ms=t_nat, native_tflops=gf / t_nat, rel_l2_s3_vs_native=diff))
        print("%-34s %7.1f GFLOP executed | split %.3f ms = %6.1f TF-eq | fp32-MFMA %.3f ms = %6.1f TF | x%.2f | rel-L2 %.2e"
              % (name, gf, t_s3, gf / t_s3, t_nat, gf / t_nat, t_nat / t_s3, diff), flush=True)
        # round 6: both backward-data products from one staging (k_s3u_bwd_pc) against the two separate launches, dz channel-blocked as in the step
        if os.environ.get("VXM_S3U_BWD_PC") == "1" and VF.s3u_bwd_data_route(c0, c1, cout, B, D, H, W):
            dzb = VF.to_blocked(dz)
            gxs, gxs2, g2 = torch.empty(B, c1, D, H, W, device="cuda"), torch.empty(B, c1, D, H, W, device="cuda"), torch.empty_like(act)

            def run_two():
                VF.s3u_bwd_low(dzb, cout, w, c0, c0 + c1, g_s3, act, 0.2, B, D, H, W, lay=VF.S3_IN0_BLOCKED)
                VF.conv_bwd_data(dzb, cout, w, gxs, c1, None, 1.0, B, D, H, W, w_lo=c0, lay=VF.S3_IN0_BLOCKED)

            def run_one():
                VF.s3u_bwd_data(dzb, cout, w, c0, c1, g2, act, 0.2, gxs2, B, D, H, W, lay=VF.S3_IN0_BLOCKED)
            t2, t1 = timed(run_two, args.iters), timed(run_one, args.iters)
            t2b, t1b = timed(run_two, args.iters), timed(run_one, args.iters)
            if args.dbg:
                res = {}
                for v in [0] + [int(v) for v in args.dbg.split(",")]:
                    os.environ["VXM_S3_DBG"] = str(v)
                    res[v] = timed(run_one, args.iters)
                os.environ["VXM_S3_DBG"] = "0"
                print("    dbg k_s3u_bwd_pc: " + " | ".join("%d: %.3f" % kv for kv in res.items()), flush=True)
            print("    %-30s low + skip in two launches %.3f / %.3f ms | k_s3u_bwd_pc %.3f / %.3f ms | rel-L2 low %.2e skip %.2e"
                  % (name.split(" dlow")[0] + " bwd-data", t2, t2b, t1, t1b, float((g2.double() - g_s3.double()).norm() / g_s3.double().norm()),
                     float((gxs2.double() - gxs.double()).norm() / gxs.double().norm())), flush=True)
            del dzb, gxs, gxs2, g2
        del dz, act, g_s3, g_nat
    # weight gradient of the upsampled segment: k_s3u_bww against k_conv3d_k3_bwd_weight_up
    for name, c0, c1, cout, lvl in (("rem0 bww-up 32^ x 32 s3u", 32, 16, 32, 0), ("dec3 bww-up 32^ x 32 s3u (L1)", 32, 32, 32, 1)):
        if args.only and args.only not in name:
            continue
        D, H, W = (s >> lvl for s in shape)
        V = D * H * W
        if not (VF.s3u_bwd_weight_route(c0, cout, B, D, H, W) or os.environ.get("VXM_S3U_BWW_MIN_TILES")):
            continue
        x0 = torch.randn(B, c0, D // 2, H // 2, W // 2, device="cuda")
        x1 = torch.randn(B, c1, D, H, W, device="cuda")
        dz = torch.randn(B, cout, D, H, W, device="cuda")
        gw_s, gw_n = torch.zeros(cout, c0 + c1, 3, 3, 3, device="cuda"), torch.zeros(cout, c0 + c1, 3, 3, 3, device="cuda")
        ws = VF._Workspace(dz.device)
        need = _lib.lib().vxm_conv3d_k3_bwd_weight_workspace_bytes(c0 + c1, cout, B, D, H, W)

        def run_s3():
            VF.s3u_bwd_weight(ws, x0, c0, x0[0].numel(), dz, cout, gw_s, c0 + c1, B, D, H, W)

        def run_nat():
            buf = ws.get(need)
            VF.call("vxm_conv3d_k3_bwd_weight_up_segment", VF.ptr(x0), c0, x0[0].numel(), VF.ptr(x1), c1, c1 * V, VF.ptr(dz), cout * V, cout,
                    VF.ptr(gw_n), VF.ptr(buf), buf.numel(), B, D, H, W, VF.stream())
        t_s3, t_nat = timed(run_s3, args.iters), timed(run_nat, args.iters)
        gf = 2.0 * 8 * c0 * cout * B * V / 1e9
        diff = float((gw_s[:, :c0].double() - gw_n[:, :c0].double()).norm() / gw_n[:, :c0].double().norm())
        rows.append(dict(op=name, gflop_executed=gf, s3_ms=t_s3, s3_tflops=gf / t_s3, native_ms=t_nat, native_tflops=gf / t_nat, rel_l2_s3_vs_native=diff))
        print("%-34s %7.1f GFLOP executed | split %.3f ms = %6.1f TF-eq | fp32-MFMA %.3f ms = %6.1f TF | x%.2f | rel-L2 %.2e"
              % (name, gf, t_s3, gf / t_s3, t_nat, gf / t_nat, t_nat / t_s3, diff), flush=True)
        del x0, x1, dz
    # backward-weight of the plain full-resolution tensors: split kernel vs the fp32-MFMA kernels
    for name, c, cout, lvl in (("rem1 bwd-weight 32->16", 32, 16, 0), ("rem2 bwd-weight 16->16", 16, 16, 0), ("rem0-skip bwd-weight 16->32", 16, 32, 0),
                               ("enc1 bwd-weight 16->32 (L1)", 16, 32, 1), ("dec3-skip bwd-weight 32->32 (L1)", 32, 32, 1)):
        if args.only and args.only not in name:
            continue
        D, H, W = (s >> lvl for s in shape)
        V = D * H * W
        x = torch.randn(B, c, D, H, W, device="cuda")
        dz = torch.randn(B, cout, D, H, W, device="cuda")
        gw_s, gb_s = torch.empty(cout, c, 3, 3, 3, device="cuda"), torch.empty(cout, device="cuda")
        gw_n, gb_n = torch.empty_like(gw_s), torch.empty_like(gb_s)
        ws = VF._Workspace(x.device)
        keep = VF.FP32_ENGINE

        def run_s3():
            VF.s3_bwd_weight(ws, x, c, c * V, dz, cout, gw_s, c, 0, gb_s, B, D, H, W)

        def run_nat():
            VF.FP32_ENGINE = "native"
            VF.conv_bwd_weight(ws, x, c, c * V, False, None, 0, 0, dz, cout, gw_n, gb_n, B, D, H, W)
            VF.FP32_ENGINE = keep
        t_s3, t_nat = timed(run_s3, args.iters), timed(run_nat, args.iters)
        if args.dbg:                                       # timing experiments (results wrong), alternating with the plain kernel
            words = [int(v) for v in args.dbg.split(",")]
            res = {v: [] for v in [0] + words}
            for _ in range(2):
                for v in [0] + words:
                    os.environ["VXM_S3_DBG"] = str(v)
                    res[v].append(timed(run_s3, args.iters))
            os.environ["VXM_S3_DBG"] = "0"
            print("    dbg: " + " | ".join("%d: %s" % (v, "/".join("%.3f" % t for t in ts)) for v, ts in res.items()), flush=True)
            run_s3()
        if os.environ.get("VXM_S3_BW_AB"):                 # same-box A/B of the backward-weight pipelines (alternating)
            ab = {"0": [], "1": []}
            for _ in range(3):
                for mode in ("0", "1"):
                    os.environ["VXM_S3_BW_PIPE"] = mode
                    ab[mode].append(timed(run_s3, args.iters))
            os.environ.pop("VXM_S3_BW_PIPE")
            print("    A/B one-barrier %s ms | two-barrier %s ms" % (" ".join("%.3f" % v for v in ab["0"]), " ".join("%.3f" % v for v in ab["1"])))
        gf = 2.0 * 27 * c * cout * B * V / 1e9
        diff = float((gw_s.double() - gw_n.double()).norm() / gw_n.double().norm())
        rows.append(dict(op=name, gflop=gf, s3_ms=t_s3, s3_tflops=gf / t_s3, native_ms=t_nat, native_tflops=gf / t_nat, rel_l2_s3_vs_native=diff))
        print("%-34s %7.1f GFLOP | split %.3f ms = %6.1f TF-eq | fp32-MFMA %.3f ms = %6.1f TF | x%.2f | rel-L2(split, fp32) %.2e"
              % (name, gf, t_s3, gf / t_s3, t_nat, gf / t_nat, t_nat / t_s3, diff), flush=True)
        del x, dz
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
