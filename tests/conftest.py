import os
import sys

import pytest

# plans in the tests use the library tile heuristic (deterministic, fast); the tuner has its own test
os.environ.setdefault("FACENET_AUTOTUNE", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def lib():
    from facenet_amd import _lib
    return _lib.load()
