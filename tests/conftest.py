import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="module")
def gpu():
    """The HIP device of the -m gpu tests; builds the in-tree library if it is stale."""
    import torch
    assert torch.cuda.is_available(), "these tests need a HIP device"
    import __graft_entry__
    __graft_entry__.build()                  # no-op when the in-tree .so matches its sources
    from audiodec_amd import native
    native.lib()
    return "cuda:0"


@pytest.fixture(scope="module")
def ckpt_root(tmp_path_factory):
    return str(tmp_path_factory.mktemp("audiodec_ckpt"))
