import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def g_layers():
    return load_golden("layers.npz")


@pytest.fixture(scope="session")
def g_losses():
    return load_golden("losses.npz")


@pytest.fixture(scope="session")
def g_network():
    return load_golden("network.npz")


@pytest.fixture(scope="session")
def g_dice():
    return load_golden("dice_metric.npz")


@pytest.fixture(scope="session")
def g_planar():
    return load_golden("planar.npz")


@pytest.fixture(scope="session")
def g_nccwin():
    return load_golden("ncc_windows.npz")
