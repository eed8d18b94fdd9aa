import os
import sys

import pytest

try:      # before anything loads libgpslam_hip.so: torch ships its own HIP runtime, and whichever loads first must be it
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_tests.json")) as f:
        return json.load(f)
