"""Checkpoint / config plumbing with the reference's on-disk format
(`voxelmorph/torch/modelio.py`): `torch.save({'config': ..., 'model_state': ...})`, identity-grid
buffers stripped on save, `strict=False` on load — so files interoperate both ways."""
import functools
import inspect

import torch
import torch.nn as nn


def store_config_args(func):
    """Record every constructor argument in `self.config` (reference: modelio.py:7-35)."""
    spec = inspect.getfullargspec(func)

    @functools.wraps(func)
    def wrapper(self, *args, **kwargs):
        self.config = {}
        if spec.defaults:
            for name, val in zip(reversed(spec.args), reversed(spec.defaults)):
                self.config[name] = val
        for name, val in zip(spec.args[1:], args):
            self.config[name] = val
        for name, val in (kwargs or {}).items():
            self.config[name] = val
        return func(self, *args, **kwargs)
    return wrapper


class LoadableModel(nn.Module):
    """Base class whose subclasses can be rebuilt from a checkpoint alone (reference: modelio.py:38-77)."""

    def __init__(self, *args, **kwargs):
        if not hasattr(self, 'config'):
            raise RuntimeError('models that inherit from LoadableModel must decorate the '
                               'constructor with @store_config_args')
        super().__init__(*args, **kwargs)

    def save(self, path):
        sd = self.state_dict().copy()
        for key in [k for k in sd.keys() if k.endswith('.grid')]:
            sd.pop(key)
        torch.save({'config': self.config, 'model_state': sd}, path)

    @classmethod
    def load(cls, path, device):
        checkpoint = torch.load(path, map_location=torch.device(device))
        model = cls(**checkpoint['config'])
        model.load_state_dict(checkpoint['model_state'], strict=False)
        return model
