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


# Parity before plumbing.  With `-x` a failing infrastructure test (the pool, the launchers, the Fortran builds) must never again leave the
# oracle / golden comparisons behind it unreached (round 5: 169 parity tests not run).  Files not named keep their alphabetical place between
# the two groups.
_FIRST = ["test_hip_parity", "test_hip_rrtmg", "test_hip_spartacus", "test_synthetic_workload", "test_reference_goldens", "test_oracle_golden",
          "test_rrtmg_golden", "test_hip_tiling", "test_mixed_gas", "test_hip_f32_boundary", "test_ifs_scheme", "test_reference_suites",
          "test_reference_targets"]
_LAST = ["test_fortran_conformance", "test_fortran_host", "test_fortran_netcdf", "test_fortran_dropin", "test_driver_outputs", "test_hdf5_output",
         "test_parallel_gloo", "test_parallel_rccl", "test_bench_line", "test_bench_launcher", "test_hip_pool", "test_host_memory"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if name in _FIRST:
            return (0, _FIRST.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=rank)      # (stable: the order inside a file is kept)
