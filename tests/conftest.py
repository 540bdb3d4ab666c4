import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ebfi-be_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The suite runs in the PRODUCT configuration -- the one bench.py and the entry points run in: the library and the package
# honour their development switches (EBFI_WGRAD_TR, EBFI_CONV_*, EBFI_NO_*: kernel selection for A/B runs) only in a process
# started with EBFI_DEV=1, and the tests do not set it (round-3 verdict: the tested and the benched process must be configured
# alike).  Kernel variants are covered through shapes that select them by the product rules, not through switches.
os.environ.pop("EBFI_DEV", None)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible and -m gpu was not requested."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
