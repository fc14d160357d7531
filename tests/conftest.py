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


@pytest.fixture(autouse=True)
def _gpu_test_leaves_nothing_running(request):
    """After EVERY GPU test: (1) wait for the stream the test itself used - what its last ``.cpu()`` orders -, (2) ask
    the library which of its INTERNAL streams still have work (edgedict_streams_busy): every entry point joins the
    streams it used before it returns and every borrower of the auxiliary stream joins it too, so a busy stream here is
    work that nothing was ordered behind - its buffers may already belong to the next test (the mechanism VERDICT r4
    suspected behind the one unreproduced failure of the suite) - and fails THIS test; (3) synchronise the device, so
    that whatever happened the next test starts on an idle one; (4) no bounded in-kernel wait may have given up."""
    yield
    if "gpu" not in request.keywords:
        return
    import ctypes
    import torch
    if not torch.cuda.is_available():
        return
    from edgedict_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        return
    lib = _lib.load()
    torch.cuda.current_stream().synchronize()
    mask = ctypes.c_uint(0)
    rc = lib.edgedict_streams_busy(ctypes.byref(mask))
    torch.cuda.synchronize()
    code = lib.edgedict_stack_wsr_error()
    assert rc == 0, "edgedict_streams_busy failed: %s" % lib.edgedict_last_error()
    assert mask.value == 0, ("the test returned with work still running on the library's internal streams (mask 0x%x: "
                             "bit 0 recurrence, 1 chunk GEMMs, 2 auxiliary) - nothing is ordered behind it" % mask.value)
    assert code == 0, "a bounded in-kernel wait gave up during this test (code %d)" % code
