import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")
    config.addinivalue_line("markers", "slow: long CPU test")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def planner_options():
    """set(key, value): change a planner option of the library (gan_set_option, include/gan_amd.h) for this test only."""
    from gan_amd import _lib as L
    saved = []

    def set_(key, value):
        saved.append((key, L.set_option(key, value)))
    yield set_
    for key, value in reversed(saved):
        L.set_option(key, value)
