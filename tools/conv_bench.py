#!/usr/bin/env python
"""Developer microbenchmark (not a test, not the product): every conv layer of the default VxmDense U-Net
at 160x192x224 through the C ABI — forward, backward-data, backward-weight — timed with HIP events on the
launch stream, TFLOP/s against the fp32 MFMA peak, and (with --check) compared with torch's conv3d on the
same device (MIOpen; a second opinion besides the oracle-based parity tests in tests/).

    python tools/conv_bench.py [--check] [--iters 5] [--only rem0,rem1] [--shape 160,192,224]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PEAK = 157.3

# name, (c0, up0, c1), cout, level, slope
LAYERS = [
    ("enc0", (1, False, 1), 16, 0, 0.2),
    ("enc1", (16, False, 0), 32, 1, 0.2),
    ("enc2", (32, False, 0), 32, 2, 0.2),
    ("enc3", (32, False, 0), 32, 3, 0.2),
    ("dec0", (32, False, 0), 32, 4, 0.2),
    ("dec1", (32, True, 32), 32, 3, 0.2),
    ("dec2", (32, True, 32), 32, 2, 0.2),
    ("dec3", (32, True, 32), 32, 1, 0.2),
    ("rem0", (32, True, 16), 32, 0, 0.2),
    ("rem1", (32, False, 0), 16, 0, 0.2),
    ("rem2", (16, False, 0), 16, 0, 0.2),
    ("flow", (16, False, 0), 3, 0, 1.0),
]
# synthetic single-segment layers for kernel studies (only with --only)
EXTRA = [
    ("up32", (32, True, 0), 32, 0, 0.2),       # pure x2-upsampled input: the collapsed-weight path alone
    ("sk16", (16, False, 0), 32, 0, 0.2),      # the skip segment of rem0 alone
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
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--shape", type=str, default="160,192,224")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--json", type=str, default="")
    args = ap.parse_args()
    from voxelmorph_amd import _lib
    if os.environ.get("VXM_LIB"):                 # developer experiments: alternative build of the library
        _lib.LIB_PATH = os.path.abspath(os.environ["VXM_LIB"])
    from voxelmorph_amd.torch import functional as VF
    shape = tuple(int(s) for s in args.shape.split(","))
    only = set(args.only.split(",")) if args.only else None
    B = args.batch
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    rows = []
    tot = {"fwd": 0.0, "bwd_data": 0.0, "bwd_weight": 0.0}
    totf = {"fwd": 0.0, "bwd_data": 0.0, "bwd_weight": 0.0}
    for name, (c0, up0, c1), cout, lvl, slope in LAYERS + EXTRA:
        if (only and name not in only) or (not only and name in [e[0] for e in EXTRA]):
            continue
        D, H, W = (s >> lvl for s in shape)
        V = D * H * W
        cin = c0 + c1
        if up0:
            x0 = torch.randn(B, c0, D // 2, H // 2, W // 2, device=dev)
        else:
            x0 = torch.randn(B, c0, D, H, W, device=dev)
        x1 = torch.randn(B, c1, D, H, W, device=dev) if c1 else None
        w = torch.randn(cout, cin, 3, 3, 3, device=dev) * (1.0 / (27 * cin) ** 0.5)
        b = torch.randn(cout, device=dev)
        y = torch.empty(B, cout, D, H, W, device=dev)
        dz = torch.randn(B, cout, D, H, W, device=dev)
        gx = torch.empty(B, cin, D, H, W, device=dev)
        gw = torch.empty_like(w)
        gb = torch.empty_like(b)
        ws = VF._Workspace(dev)
        flops = 2.0 * 27 * cin * cout * B * V

        def fwd():
            VF.conv_forward(x0, c0, x0[0].numel(), up0, x1, c1, x1[0].numel() if x1 is not None else 0, w, b, y, cout * V,
                            cout, slope, B, D, H, W)

        def bwd_data():
            VF.conv_bwd_data(dz, cout, w, gx, cin, None, 1.0, B, D, H, W)

        def bwd_weight():
            VF.conv_bwd_weight(ws, x0, c0, x0[0].numel(), up0, x1, c1, x1[0].numel() if x1 is not None else 0, dz, cout,
                               gw, gb, B, D, H, W)

        row = {"layer": name, "cin": cin, "cout": cout, "vox": V, "gflop": flops / 1e9}
        for k, fn in (("fwd", fwd), ("bwd_data", bwd_data), ("bwd_weight", bwd_weight)):
            ms = timed(fn, args.iters)
            row[k + "_ms"] = ms
            row[k + "_tf"] = flops / (ms * 1e-3) / 1e12
            tot[k] += ms
            totf[k] += flops
        if args.check:
            xin = x0
            if up0:
                xin = F.interpolate(x0, scale_factor=2, mode="nearest")
            if x1 is not None:
                xin = torch.cat([xin, x1], 1)
            xin = xin.detach().requires_grad_()
            wr = w.detach().requires_grad_()
            br = b.detach().requires_grad_()
            pre = F.conv3d(xin, wr, br, padding=1)
            ref = F.leaky_relu(pre, slope) if slope != 1.0 else pre
            pre.backward(dz)

            def rel(a, r):
                return float((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-30))
            fwd(); bwd_data(); bwd_weight()
            torch.cuda.synchronize()
            row["err_y"] = rel(y, ref)
            row["err_gx"] = rel(gx, xin.grad)
            row["err_gw"] = rel(gw, wr.grad)
            row["err_gb"] = rel(gb, br.grad)
            del xin, pre, ref
        rows.append(row)
        print("%-5s cin=%2d cout=%2d V=%8d  fwd %7.3f ms %6.1f TF | bwd_data %7.3f ms %6.1f TF | bwd_weight %7.3f ms %6.1f TF"
              % (name, cin, cout, V, row["fwd_ms"], row["fwd_tf"], row["bwd_data_ms"], row["bwd_data_tf"],
                 row["bwd_weight_ms"], row["bwd_weight_tf"])
              + ("  err y %.1e gx %.1e gw %.1e gb %.1e" % (row["err_y"], row["err_gx"], row["err_gw"], row["err_gb"])
                 if args.check else ""), flush=True)
        del x0, x1, y, dz, gx
        torch.cuda.empty_cache()
    for k in tot:
        if tot[k] > 0:
            print("total %-10s %8.3f ms  %6.1f TF  (%.1f%% of %.1f)" % (k, tot[k], totf[k] / tot[k] / 1e9, 100 * totf[k] / tot[k] / 1e9 / PEAK, PEAK))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
