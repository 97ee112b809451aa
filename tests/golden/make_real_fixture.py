#!/usr/bin/env python
"""Makes tests/golden/real_scan_u8.npz from the reference's only real-image fixture at the benchmark shape,
`/root/reference/data/test_scan.npz` (`vol`, `seg`: 160x192x224 float64) and `data/labels.npz` (the 30 evaluated labels).

The volume is k/255 to within one float64 ulp and the label map holds integers < 256, so both are stored as uint8
(3 MB instead of 2 x 55 MB); `(vol_u8 / 255.0).astype(float32)` is BIT-IDENTICAL to `vol.astype(float32)`, which is what the
path sees (`scripts/torch/train.py:200`: `.float()`).  The GPU box has no /root/reference: this derived fixture is what travels.
Run in the build container: `python tests/golden/make_real_fixture.py`."""
import os

import numpy as np

REF = "/root/reference/data"
d = np.load(os.path.join(REF, "test_scan.npz"))
vol, seg = d["vol"], d["seg"]
labels = np.load(os.path.join(REF, "labels.npz"))["labels"]
vol_u8 = np.round(vol * 255.0).astype(np.uint8)
seg_u8 = seg.astype(np.uint8)
assert np.array_equal((vol_u8.astype(np.float64) / 255.0).astype(np.float32), vol.astype(np.float32)), "volume is not k/255 in fp32"
assert np.array_equal(seg_u8.astype(np.float64), seg), "label map does not fit uint8"
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "real_scan_u8.npz")
np.savez_compressed(out, vol_u8=vol_u8, seg_u8=seg_u8, labels=labels.astype(np.int32))
print("wrote", out, os.path.getsize(out), "bytes; shape", vol.shape, "labels", len(labels))
