"""Host-side contracts pinned to the LIVE reference (CPU; skipped where /root/reference is absent -- the GPU box):
the generator fixtures the device loaders are compared with, and the Jacobian-determinant check of scripts/register.py."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ref_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_ref = pytest.mark.skipif(not ref_loader.reference_available(), reason="reference tree not present")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@needs_ref
def test_generator_fixture_is_what_the_live_reference_yields(tmp_path):
    """tests/golden/generators.npz == a fresh run of voxelmorph/generators.py:71-194 (scan_to_scan, scan_to_atlas, semisupervised)
    through the committed script: the fixture the device loaders are checked against on the GPU box is pinned to the reference."""
    mk = _load(os.path.join(ROOT, "tests", "golden", "make_generators_golden.py"), "make_generators_golden")
    fresh = mk.generate(str(tmp_path))
    g = np.load(os.path.join(ROOT, "tests", "golden", "generators.npz"))
    assert sorted(g.files) == sorted(fresh)
    for k in g.files:
        assert g[k].dtype == np.asarray(fresh[k]).dtype and np.array_equal(g[k], fresh[k]), k
    # the structure the loaders' docstrings promise, read off the reference's own output
    assert list(g["s2s_n"]) == [2, 2] and list(g["s2s_bidir_n"]) == [2, 3] and list(g["s2s_nowarp_n"]) == [2, 1]
    assert list(g["semi_n"]) == [3, 3] and g["semi_in2"].shape == (1, 4, 5, 6, 3)
    assert not g["s2s_out1"].any() and g["s2s_out1"].shape == (2, 8, 10, 12, 3)              # the zero flow target, generators.py:98-101


def _smooth_disp(shape, seed=3, amp=2.0):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    f = np.stack([gaussian_filter(rng.standard_normal(shape), 2.5, mode="nearest") for _ in range(len(shape))], -1)
    return (amp * f / np.abs(f).max()).astype(np.float32)


def test_register_jacobian_matches_numpy_restatement():
    """scripts/register.py --jacobian: det(I + grad disp) with central differences == the numpy formula of py/utils.py:473-516
    (np.gradient of disp + identity grid, cofactor expansion), on a smooth field with folds (amplitude 2 voxels -> some det <= 0)."""
    reg = _load(os.path.join(ROOT, "scripts", "register.py"), "vxm_register_script")
    shape = (12, 14, 16)
    disp = _smooth_disp(shape, amp=6.0)                                  # [*vol, 3]
    grid = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1)      # pystrum.pynd.ndutils.volsize2ndgrid
    J = np.gradient(disp.astype(np.float64) + grid)
    dx, dy, dz = J[0], J[1], J[2]
    det = (dx[..., 0] * (dy[..., 1] * dz[..., 2] - dy[..., 2] * dz[..., 1]) - dx[..., 1] * (dy[..., 0] * dz[..., 2] - dy[..., 2] * dz[..., 0])
           + dx[..., 2] * (dy[..., 0] * dz[..., 1] - dy[..., 1] * dz[..., 0]))
    got = reg.jacobian_determinant(torch.from_numpy(np.moveaxis(disp, -1, 0).copy()))
    np.testing.assert_allclose(got.numpy(), det, rtol=1e-4, atol=1e-4)
    frac = reg.nonpositive_jacobian_fraction(torch.from_numpy(np.moveaxis(disp, -1, 0).copy()))
    assert abs(frac - float((det <= 0).mean())) <= 1.0 / det.size + 1e-9 and 0.0 < frac < 0.5


@needs_ref
def test_register_jacobian_matches_live_reference():
    """The same against the reference function itself (py/utils.py:473-516).  Its one third-party call, pystrum's
    `volsize2ndgrid`, is absent from this image and stubbed with its documented behaviour (an 'ij' meshgrid of aranges)."""
    ref = ref_loader.load_reference()
    nd = sys.modules["pystrum.pynd.ndutils"]
    if not hasattr(nd, "volsize2ndgrid"):
        nd.volsize2ndgrid = lambda volsize: np.meshgrid(*[np.arange(e) for e in volsize], indexing="ij")
    if not hasattr(ref.py.utils.nd, "volsize2ndgrid"):
        ref.py.utils.nd.volsize2ndgrid = nd.volsize2ndgrid
    reg = _load(os.path.join(ROOT, "scripts", "register.py"), "vxm_register_script")
    disp = _smooth_disp((10, 12, 14), seed=5, amp=5.0)
    want = ref.py.utils.jacobian_determinant(disp.astype(np.float64))
    got = reg.jacobian_determinant(torch.from_numpy(np.moveaxis(disp, -1, 0).copy()))
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-4, atol=1e-4)
