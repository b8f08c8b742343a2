import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Without a GPU (this container, any CPU machine) a plain `pytest` run skips the gpu-marked tests instead of failing
    them; on the GPU box nothing is skipped -- the HIP path has no fallback to hide behind."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    try:
        from bloomfiltertrie_amd import _lib
        have = _lib.load().bft_gpu_device_count() > 0
    except Exception:
        return  # a library that does not load is a failure to show, not a reason to skip
    if not have:
        skip = pytest.mark.skip(reason="no HIP device: gpu-marked tests need a real MI355X")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as O
    O.build()
    return O
