import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TESTDATA = os.path.join(GOLDEN, "testdata")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load(name):
    with open(os.path.join(TESTDATA, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    return pyoracle
