import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: the gpu-marked tests are skipped instead of failing their fixtures."""
    try:
        import swarmmap_amd
        have_gpu = swarmmap_amd.device_count() > 0
    except Exception:  # library not built: the CPU tests that need it say so themselves
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
