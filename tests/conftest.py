import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests are only meaningful on a GPU box; elsewhere they are skipped (the driver selects with -m).
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def dev_lib(monkeypatch):
    """Route this test's engine calls to librappas_place_dev.so (built with -DRK_DEV_KNOBS): the product library reads no
    environment variable, so tests that steer the engine with a developer knob (RK_WG_PASSES, RK_WINDOW_ALWAYS, the shard-failure
    injector ...) must say so by asking for this fixture.  Handles created before the test keep the library they were made with."""
    from rappas_amd import _lib
    _lib.load()
    monkeypatch.setattr(_lib, "_LIB", _lib.load_dev())
    yield
