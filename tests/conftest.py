import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _ensure_library_built():
    """The in-tree libcolvo.so travels with the tree, but a fresh checkout has none: build it (hipcc cross-compiles
    without a GPU, seconds when objects are cached)."""
    from coivo_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
