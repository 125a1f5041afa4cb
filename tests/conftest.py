import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libmrt_ref.so (build container only)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    from tests import orc
    skip_ref = pytest.mark.skip(reason="oracle/_ref/libmrt_ref.so not present (reference sources do not travel)")
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    has_gpu = None
    for item in items:
        if "ref" in item.keywords and not orc.have_ref():
            item.add_marker(skip_ref)
        if "gpu" in item.keywords:
            if has_gpu is None:
                has_gpu = _has_gpu()
            if not has_gpu:
                item.add_marker(skip_gpu)


@pytest.fixture(scope="session")
def built():
    """The native libraries must exist (built in-tree by __graft_entry__.build())."""
    from raytracer_amd import abi
    abi.hip()
    abi.host()
    from tests import orc
    orc.oracle()
    return True
