import os
import sys

import pytest

# PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so.7).  Whichever copy a process loads first is
# the one everything else binds to, and torch cannot find the GPU if libecrad_hip.so has already brought in
# /opt/rocm's: import torch before the library is loaded (bench.py does the same).
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA_DIR = os.path.join(ROOT, "data")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    pyoracle.build(ref=True)
    return pyoracle
