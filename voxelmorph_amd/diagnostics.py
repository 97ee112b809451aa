"""How does the fp16-piece conv engine see this model / batch?  

The default engine (`VXM_FP32_ENGINE=f16x2`) writes every fp32 operand of the big convolutions as two fp16 pieces with one power-of-two scale
per staged tile.  Inside one tile (8 channels x 8 x 8 x 16 voxels) a value below 2^-18 of the tile's largest magnitude keeps an ABSOLUTE
error (2^-40 of that magnitude) instead of a relative one.  The tensor's norm never notices (the large values carry it: `tensor_rel_l2_bound`
below is ~1e-11 even for pathological inputs); a consumer that normalises locally, as the windowed NCC does, could.

`range_report(fn)` runs `fn` (a forward + backward of the model) with a probe (csrc/diag.hip) on every activation and gradient tensor of the
fused U-Net and returns, per tensor, the share of its non-zero values in that regime.  Measured (round 5): <= 4.5e-4 on noise pairs (the
bench workload; the worst tensor is a coarse-level gradient), 0.48 on the real scan with 0.1 % of its voxels multiplied by 2^12 .. 2^20.
On that heavy-tailed step the three engines were compared against the fp64-NCC arbiter at full size
(tests/test_gpu_parity.py::test_full_size_step_with_heavy_tailed_activations_on_all_three_engines): worst parameter gradient 8.9e-2 on the
fp16 pieces, 1.3e-2 on three bf16 pieces, 9.7e-3 on the exact fp32 MFMA (round 6; a reference-order fp32 evaluation on the host: 8.1e-3) -- on
such a batch the fp16 pieces cost about one digit.  The guard -- `guard_engine`, which `GraphedStep` runs on its first eager step (default ON
since round 6; `VXM_RANGE_GUARD=0` / `range_guard=False` switch it off) -- moves the process to the three-piece engine (fp32's exponent
range, 25 % slower) when more than `share_limit` of a tensor is in the absolute-error regime: a one-off cost of one probe launch per
tensor (~3 ms at 160x192x224), nothing in the captured step.
"""
import torch

from .torch import functional as VF

SHARE_LIMIT = 1e-3


def range_report(fn, share_limit=SHARE_LIMIT):
    if VF._RANGE_PROBE is not None:
        raise RuntimeError("range_report: already probing")
    VF._RANGE_PROBE = []
    try:
        fn()
        torch.cuda.synchronize()
        rows = []
        for tag, out in VF._RANGE_PROBE:
            e_below, energy, n_below, n_nonzero = (float(v) for v in out.cpu())
            rows.append({"tensor": tag, "share_below_2^-18_of_tile_max": n_below / n_nonzero if n_nonzero else 0.0,
                         "tensor_rel_l2_bound": (e_below * 2.0 ** -80 / energy) ** 0.5 if energy else 0.0})
    finally:
        VF._RANGE_PROBE = None
    worst = max(rows, key=lambda r: r["share_below_2^-18_of_tile_max"]) if rows else None
    return {"tensors": rows, "worst": worst, "share_limit": share_limit,
            "recommended_engine": "split" if worst is not None and worst["share_below_2^-18_of_tile_max"] > share_limit else "f16x2"}


def guard_engine(fn, share_limit=SHARE_LIMIT):
    """Run `fn` under the probe when the fp16-piece engine is selected; switch this process to the three-piece engine (and say so) when the
    report asks for it.  Returns the report (None when another engine is selected already)."""
    import warnings
    if VF.FP32_ENGINE != "f16x2":
        return None
    rep = range_report(fn, share_limit)
    if rep["recommended_engine"] == "split":
        w = rep["worst"]
        warnings.warn("voxelmorph_amd: %.2g of the non-zero values of %s lie below 2^-18 of their tile's largest magnitude: the fp16-piece conv "
                      "engine would keep only an absolute error bound for them; switching to the three-piece bf16 engine (VXM_FP32_ENGINE=split)"
                      % (w["share_below_2^-18_of_tile_max"], w["tensor"]))
        VF.FP32_ENGINE = "split"
    return rep
