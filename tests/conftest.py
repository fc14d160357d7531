import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    # the CPU oracle (LSTM-bound torch graphs) is 4-5x SLOWER with one thread per hardware thread on
    # the GPU box's 256 logical CPUs than with 16 (bench.py's cpu_baseline sweep): cap it
    import torch
    if (os.cpu_count() or 1) > 32:
        torch.set_num_threads(16)


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_lib():
    """Make sure libedgedict_hip.so exists (builds it if hipcc is around) and load it."""
    from edgedict_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_all(verbose=False)
    return _lib.load()
