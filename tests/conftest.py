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


@pytest.fixture(scope="session")
def g_unetpools():
    return load_golden("unet_pools.npz")


# tag -> (inshape, infeats, nb_features, max_pool) of tests/golden/make_golden.py:gold_unet_pools
UNET_POOL_CASES = {
    "p3_vol": ((18, 9, 18), 2, [[8, 8], [8, 8, 8]], 3),
    "aniso_vol": ((4, 16, 12), 2, [[8, 8], [8, 8, 8]], [(1, 2, 2), (1, 2, 2), (1, 2, 2)]),
    "p3_img": ((27, 18), 1, [[8, 8], [8, 8, 8]], 3),
    "p42_img": ((16, 24), 2, [[8], [8, 8]], [4, 2]),
}
