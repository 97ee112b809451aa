#!/usr/bin/env python
"""TEST INFRASTRUCTURE: outputs of the UNMODIFIED reference generators (voxelmorph/generators.py:71-194: scan_to_scan,
scan_to_atlas, semisupervised) on tiny synthetic volumes, committed as tests/golden/generators.npz.  The device loaders of
`voxelmorph_amd.data` are compared with them on the GPU box (tests/test_gpu_parity.py), where /root/reference does not exist;
tests/test_cpu_contracts.py re-runs this script against the live reference and checks the committed file is what it produces.

Only deterministic configurations are recorded (one-volume file lists, fixed atlas): the reference draws indices from numpy's
global generator (generators.py:53), the device loaders from a per-rank generator, so random picks are not comparable.

    python tests/golden/make_generators_golden.py [out.npz]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SHAPE = (8, 10, 12)
LABELS = [2, 3, 41]


def inputs():
    rng = np.random.default_rng(2024)
    vol = rng.random(SHAPE)                                   # float64 in [0, 1), as the reference's data (py/utils.py:69-129 passes arrays through)
    seg = rng.choice([0, 2, 3, 41, 7], size=SHAPE).astype(np.float64)
    atlas_vol = rng.random(SHAPE)
    atlas_seg = rng.choice([0, 2, 3, 41], size=SHAPE).astype(np.float64)
    return vol, seg, atlas_vol, atlas_seg


def generate(tmpdir):
    from oracle import ref_loader
    ref = ref_loader.load_reference()
    gen = ref.generators
    vol, seg, atlas_vol, atlas_seg = inputs()
    out = {"vol": vol, "seg": seg, "atlas_vol": atlas_vol, "atlas_seg": atlas_seg, "labels": np.asarray(LABELS)}

    def put(tag, pair):
        invols, outvols = pair
        for i, a in enumerate(invols):
            out["%s_in%d" % (tag, i)] = np.asarray(a)
        for i, a in enumerate(outvols):
            out["%s_out%d" % (tag, i)] = np.asarray(a)
        out[tag + "_n"] = np.asarray([len(invols), len(outvols)])

    put("s2s", next(gen.scan_to_scan([vol], batch_size=2)))
    put("s2s_bidir", next(gen.scan_to_scan([vol], bidir=True, batch_size=1)))
    put("s2s_nowarp", next(gen.scan_to_scan([vol], no_warp=True, batch_size=1)))
    atlas = atlas_vol[np.newaxis, ..., np.newaxis]            # scripts/torch/train.py:105-106: add_batch_axis + add_feat_axis
    put("s2a", next(gen.scan_to_atlas([vol], atlas, batch_size=2)))
    put("s2a_bidir", next(gen.scan_to_atlas([vol], atlas, bidir=True, batch_size=1)))
    put("s2a_segs", next(gen.scan_to_atlas([vol], atlas, batch_size=1, segs=[seg])))
    put("semi", next(gen.semisupervised([vol], [seg], LABELS)))
    apath = os.path.join(tmpdir, "atlas.npz")
    np.savez(apath, vol=atlas_vol, seg=atlas_seg)
    put("semi_atlas", next(gen.semisupervised([vol], [seg], LABELS, atlas_file=apath)))
    return out


if __name__ == "__main__":
    import tempfile
    target = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "generators.npz")
    with tempfile.TemporaryDirectory() as td:
        np.savez_compressed(target, **generate(td))
    print("wrote", target)
