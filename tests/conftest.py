import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:   # tests/parity.py
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")
    # torch's own DataLoader pin-memory thread trips a torch deprecation once per tensor
    config.addinivalue_line("filterwarnings", "ignore:The argument 'device' of Tensor:DeprecationWarning")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
