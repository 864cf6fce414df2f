import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive variants of a case the default -m gpu run already covers; they run "
                            "with SASSD_FULL_TESTS=1 (tools/gpu_full_tests.sh) -- the default run stays under ten minutes")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("SASSD_FULL_TESTS", "") not in ("", "0"):
        return
    skip = pytest.mark.skip(reason="exhaustive variant: set SASSD_FULL_TESTS=1 (tools/gpu_full_tests.sh)")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
