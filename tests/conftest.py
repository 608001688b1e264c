import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


# The engine leaves small problems (fewer than 1e7 (spike, inducing point) pairs) to its single pair-by-pair launch; the
# parity tests are small problems, and they are there to check the class path of the spike pass too: force it for the
# session (tests/test_gpu_parity.py::test_small_problems_take_the_pair_pass checks the default policy).
os.environ.setdefault("SVGPFA_SERIES_MIN_PAIRS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
