import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    """The product package (raytracer-in-cpp_amd), loaded under an importable name."""
    import rtpkg
    return rtpkg.load()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/librt_oracle.so).  Built here if missing; used ONLY as the checker."""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def scenes():
    return SCENES
