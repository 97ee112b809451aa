"""Checkpoint / config plumbing that interoperates with the reference's files both ways.

On-disk format (reference `voxelmorph/torch/modelio.py:58-77`): `torch.save({'config': <constructor arguments>,
'model_state': <state_dict without the SpatialTransformer identity grids>})`; loading rebuilds the model from `config` and
tolerates the missing grid buffers.  The behaviour is the reference's; the implementation here binds the constructor call
through `inspect.signature` instead of walking the argspec by hand.
"""
import functools
import inspect

import torch
from torch import nn

_GRID_SUFFIX = '.grid'          # buffers of SpatialTransformer: rebuilt by the constructor, never stored


def store_config_args(init):
    """Decorator for `__init__`: keeps the effective constructor arguments (defaults, then positionals, then keywords) in
    `self.config`, which is all `LoadableModel.load` needs to rebuild the network (reference: modelio.py:7-35)."""
    signature = inspect.signature(init)
    named = [p for p in list(signature.parameters.values())[1:]
             if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]

    @functools.wraps(init)
    def init_and_record(self, *args, **kwargs):
        config = {p.name: p.default for p in named if p.default is not p.empty}
        config.update(zip((p.name for p in named), args))
        config.update(kwargs)                       # explicit keywords, including those swallowed by a **kwargs parameter
        self.config = config
        return init(self, *args, **kwargs)

    return init_and_record


class LoadableModel(nn.Module):
    """`nn.Module` that can be saved and re-created from a single file (reference: modelio.py:38-77).  Subclasses decorate
    their constructor with `@store_config_args` (or set `self.config` themselves) before calling `super().__init__()`."""

    def __init__(self, *args, **kwargs):
        if getattr(self, 'config', None) is None:
            raise RuntimeError('models that inherit from LoadableModel must decorate the '
                               'constructor with @store_config_args')
        super().__init__(*args, **kwargs)

    def save(self, path):
        """Write `{'config', 'model_state'}`; identity-grid buffers are left out (they depend only on the image shape)."""
        weights = {name: tensor for name, tensor in self.state_dict().items() if not name.endswith(_GRID_SUFFIX)}
        torch.save({'config': self.config, 'model_state': weights}, path)

    @classmethod
    def load(cls, path, device):
        """Rebuild the model from a checkpoint written by `save` (or by the reference) and map it to `device`."""
        blob = torch.load(path, map_location=torch.device(device))
        model = cls(**blob['config'])
        model.load_state_dict(blob['model_state'], strict=False)      # the grids are missing by design
        return model
