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
    """The in-tree libcolvo.so travels with the tree, but a fresh checkout has none and an edited kernel must not run
    against a stale binary: (re)build unless the library's recorded source hash matches the tree (hipcc cross-compiles
    without a GPU)."""
    from coivo_amd import build
    build.ensure()


@pytest.fixture(autouse=True)
def _restore_aux_stream_limit(request):
    """data.PairLoader and ddp.GradBuckets switch the library's auxiliary side stream off for the process
    (colvo_set_aux_side_streams(0): they bring a third hardware queue of their own).  Tests share one process: put the default
    back after each GPU test so that the ones that follow still exercise the two-side-stream schedule."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from coivo_amd import _lib
        if _lib._lib is not None:
            _lib._lib.colvo_set_aux_side_streams(3)
