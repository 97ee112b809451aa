"""Identity of the kernel sources a measurement belongs to (bench.py `roofline.traffic`, tools/rocprof_summary.py `_meta`)."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def csrc_sha():
    """sha1 (16 hex digits) over voxelmorph_amd/csrc/*.{hip,h,cpp,sh}, file names included, sorted by name."""
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(_CSRC, "*.*"))):
        if f.rsplit(".", 1)[-1] in ("hip", "h", "cpp", "sh"):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]
