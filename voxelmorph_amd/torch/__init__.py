from . import layers, losses, modelio, networks  # noqa: F401
