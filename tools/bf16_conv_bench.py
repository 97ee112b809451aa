#!/usr/bin/env python
"""Per-layer timing of the bf16 conv kernels on the default VxmDense U-Net at 160x192x224 (B = 1): forward, backward-data per
input segment, backward-weight.  Prints ms, TFLOP/s on the reference formulation's FLOPs and GB/s on the algorithmic bytes
(every operand read once, every result written once, bf16 activations).  Usage: python tools/bf16_conv_bench.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from voxelmorph_amd.torch import functional_bf16 as VB  # noqa: E402
from voxelmorph_amd.torch.functional import _Workspace  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
FULL = (160, 192, 224)
LAYERS = [  # name, c0 (blocked), real c0, up0, c1, cout, level
    ("enc0", 16, 2, False, 0, 16, 0), ("enc1", 16, 16, False, 0, 32, 1), ("enc2", 32, 32, False, 0, 32, 2), ("enc3", 32, 32, False, 0, 32, 3),
    ("dec0", 32, 32, False, 0, 32, 4), ("dec1", 32, 32, True, 32, 32, 3), ("dec2", 32, 32, True, 32, 32, 2), ("dec3", 32, 32, True, 32, 32, 1),
    ("rem0", 32, 32, True, 16, 32, 0), ("rem1", 32, 32, False, 0, 16, 0), ("rem2", 16, 16, False, 0, 16, 0), ("flow", 16, 16, False, 0, 3, 0),
]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(REPS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / REPS


def main():
    dev = torch.device("cuda")
    ws = _Workspace(dev)
    tot = dict(fwd=0.0, bwd_data=0.0, bwd_weight=0.0)
    print("%-5s %-9s %9s %9s %9s" % ("layer", "pass", "ms", "TFLOP/s", "GB/s"))
    for name, c0, c0r, up0, c1, cout, lvl in LAYERS:
        D, H, W = (s >> lvl for s in FULL)
        V = D * H * W
        lo = (D // 2, H // 2, W // 2)
        x0 = (torch.randn((1, c0 // 8) + (lo if up0 else (D, H, W)) + (8,), device=dev)).to(torch.bfloat16)
        x1 = torch.randn((1, c1 // 8, D, H, W, 8), device=dev).to(torch.bfloat16) if c1 else None
        w = torch.randn(cout, c0r + c1, 3, 3, 3, device=dev) / (27 * (c0r + c1)) ** 0.5
        b = torch.zeros(cout, device=dev)
        planar = cout <= 4
        cdz = (cout + 15) // 16 * 16
        y = torch.empty((1, cout, D, H, W), device=dev) if planar else VB._blocked(1, cout, (D, H, W), dev)
        dz = torch.randn((1, cdz // 8, D, H, W, 8), device=dev).to(torch.bfloat16)
        flops = 2.0 * 27 * (c0r + c1) * cout * V
        v0 = V // 8 if up0 else V
        wp = VB.pack_weights(w, 0, c0r + c1, False)
        t = timed(lambda: VB.conv(x0, c0, up0, x1, c1, wp, b, y, cout, planar, 1.0 if planar else 0.2, None, 1.0, 1, D, H, W))
        byts = 2.0 * (c0 * v0 + c1 * V) + (4.0 if planar else 2.0) * cout * V
        print("%-5s %-9s %9.3f %9.1f %9.0f" % (name, "fwd", t, flops / t / 1e9, byts / t / 1e6))
        tot["fwd"] += t
        if name != "enc0":
            for seg, (lo_c, n_c) in enumerate(((0, c0r), (c0r, c1))):
                if not n_c:
                    continue
                gx = VB._blocked(1, n_c, (D, H, W), dev)
                wpf = VB.pack_weights(w, lo_c, n_c, True)
                t = timed(lambda: VB.conv(dz, cdz, False, None, 0, wpf, None, gx, n_c, False, 1.0, None, 1.0, 1, D, H, W))
                print("%-5s %-9s %9.3f %9.1f %9.0f" % (name, "bwd_data%d" % seg, t, 2.0 * 27 * n_c * cout * V / t / 1e9, 2.0 * (cdz + n_c) * V / t / 1e6))
                tot["bwd_data"] += t
        gw, gb = torch.empty_like(w), torch.empty(cout, device=dev)
        t = timed(lambda: VB.conv_bwd_weight(ws, x0, c0, up0, x1, c1, dz, cdz, gw, gb, 1, D, H, W))
        print("%-5s %-9s %9.3f %9.1f %9.0f" % (name, "bwd_wgt", t, flops / t / 1e9, 2.0 * (c0 * v0 + c1 * V + cdz * V) / t / 1e6))
        tot["bwd_weight"] += t
    print("totals (ms): " + ", ".join("%s %.3f" % kv for kv in tot.items()) + ", sum %.3f" % sum(tot.values()))


if __name__ == "__main__":
    main()
