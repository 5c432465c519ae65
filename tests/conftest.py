import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: timing expectations (tests/test_zz_gpu_perf.py): run after the parity files, "
                                       "reported instead of asserted unless AGPU_PERF_STRICT=1")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ag():
    """The product namespace on a GPU box (fails loudly — no CPU fallback — if the HIP library or device is missing)."""
    import arrow_gpu_amd

    arrow_gpu_amd.GPU_DEVICE()
    return arrow_gpu_amd
