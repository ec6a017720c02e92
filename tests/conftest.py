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
def _restore_stream_policy(request):
    """data.PairLoader and ddp.GradBuckets claim a hardware queue from the stream policy (coivo_amd/streams.py), which switches
    the library's auxiliary side stream off while a claim is held.  Tests share one process: drop whatever a test left claimed
    so that the ones that follow still exercise the two-side-stream schedule."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from coivo_amd import _lib, streams
        if _lib._lib is not None:
            streams.reset()
