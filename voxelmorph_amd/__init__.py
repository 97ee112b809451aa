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
from .graph import GraphedStep  # noqa: F401              (a training step as one hipGraph launch)
from .diagnostics import range_report  # noqa: F401      (how the fp16-piece conv engine sees a model / batch)

__version__ = "0.5.0"
