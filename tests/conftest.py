import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("GOBBLET_HIP_LIB"):  # scripts/experiments/ab_sweep.sh: the suite against an experiment's own build of the library
        import gobblet_rl_amd as G
        G._native.use_library(os.environ["GOBBLET_HIP_LIB"])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
