"""TEST INFRASTRUCTURE ONLY — loader for the *real* reference package.

Imports the unmodified upstream torch backend from /root/reference with the
import stubs SURVEY.md Appendix A verified (neurite / pystrum / skimage are
import-time gates only, `voxelmorph/__init__.py:13-19`, `py/utils.py:10,13`).
/root/reference exists only in the build container: this module is used by
`tests/golden/make_golden.py` (fixture generation) and by CPU tests that are
skipped when the reference is absent.  Nothing in the product imports it.
"""
import contextlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VXM_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "voxelmorph", "torch"))


def load_reference():
    """Return the upstream `voxelmorph` module (pytorch backend)."""
    if "voxelmorph" in sys.modules and getattr(sys.modules["voxelmorph"], "_vxm_is_reference", False):
        return sys.modules["voxelmorph"]
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    os.environ["VXM_BACKEND"] = "pytorch"            # voxelmorph/py/utils.py:24-29
    ne = types.ModuleType("neurite")
    ne.__version__ = "0.2"                           # voxelmorph/__init__.py:13-19
    sys.modules.setdefault("neurite", ne)
    for name in ["pystrum", "pystrum.pynd", "pystrum.pynd.ndutils", "skimage", "skimage.measure"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["pystrum"].pynd = sys.modules["pystrum.pynd"]
    sys.modules["pystrum.pynd"].ndutils = sys.modules["pystrum.pynd.ndutils"]
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import voxelmorph as vxm
    finally:
        sys.path.remove(REFERENCE_ROOT)
    vxm._vxm_is_reference = True
    return vxm


@contextlib.contextmanager
def cuda_alias_to_cpu():
    """Let the reference's `NCC.loss` run on a GPU-less host.

    `voxelmorph/torch/losses.py:29` hard-codes `torch.ones(...).to("cuda")`.
    Inside this context `.to("cuda")` is a no-op so the *unmodified* reference
    code executes on CPU tensors.
    """
    import torch
    orig = torch.Tensor.to

    def patched(self, *args, **kwargs):
        if args and isinstance(args[0], str) and args[0].startswith("cuda"):
            args = args[1:]
            if not args and not kwargs:
                return self
        return orig(self, *args, **kwargs)

    torch.Tensor.to = patched
    try:
        yield
    finally:
        torch.Tensor.to = orig
