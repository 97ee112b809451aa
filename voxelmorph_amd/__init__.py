"""voxelmorph_amd — MI355X-native VxmDense training/inference path.

Mirrors the reference's pytorch-backend surface (`voxelmorph/__init__.py:32-45`):
`voxelmorph_amd.layers`, `.networks`, `.losses` expose the same classes as
`voxelmorph.layers/networks/losses`, backed by hand-written gfx950 HIP kernels behind the C ABI
of `include/vxm_hip.h` (libvxm_hip.so).  No CPU or ATen fallback exists for the hot path.
"""
from .torch import layers, losses, networks  # noqa: F401
from .torch.networks import default_unet_features  # noqa: F401
from . import torch  # noqa: F401
from .torch.functional_bf16 import invalidate_packs  # noqa: F401

__version__ = "0.1.0"
