import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# a flag wait of the peer-mapped exchange polls for 600 s by default before the call falls back to the collective; in the tests
# (several ranks time-sliced on ONE device) a stall should cost a minute, not ten (inherited by the spawned ranks)
os.environ.setdefault("CLID_P2P_TIMEOUT_S", "60")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a real MI355X and the built HIP library: skip them (instead of failing on the first
    one) when a plain `pytest` is run on a CPU-only box."""
    try:
        import torch

        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    lib = os.path.join(ROOT, "clid-slam_amd", "lib", "libclid_native.so")
    if have_gpu and os.path.exists(lib):
        return
    why = "no GPU visible" if not have_gpu else f"{lib} has not been built"
    skip = pytest.mark.skip(reason=f"gpu test: {why}")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
