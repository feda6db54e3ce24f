"""Every golden output file the reference's own tests hold for this path (test/ifs/*_out_REFERENCE.nc, byte-identical
copies under tests/golden/), reproduced end to end
  * by the oracle on the CPU (pins the restatement: McICA, Tripleclouds and Cloudless solvers, RRTMG and ecCKD gas
    optics, SOCRATES/Fu and general cloud optics, aerosols, Exp-Ran and Exp-Exp overlap, spectral flux profiles), and
  * by the HIP path through the C-ABI (`-m gpu`).
The files are float32, so agreement is limited to float32 rounding (2^-24 = 6e-8 relative); the tolerance is 2e-7
relative to max(|ref|, 1e-3 max|ref|), i.e. ~1000x tighter than the reference's ctest thresholds
(test/ifs/CMakeLists.txt:14-20) and 5x tighter than the 1e-6 parity bar.  EVERY variable of every file must be
produced with the file's shape and compared: a variable that is missing or differently shaped fails the test."""
import os

import numpy as np
import pytest

from ecrad_amd.driver import flux_to_output_dict
from ecrad_amd.ncfile import NcFile
from helpers import GOLDEN_CASES, GOLDEN_DIR, make_golden_config, rel_err, run_case

FLOAT32_TOL = 2.0e-7
N_VARIABLES = {"ecckd_mcica": 21, "default": 21, "noaer": 21, "expexp": 21, "tripleclouds": 27, "cloudless": 25}


def golden_path(name):
    return os.path.join(GOLDEN_DIR, f"ecrad_meridian_{name}_out_REFERENCE.nc")


def check_against_golden(name, out):
    worst = {}
    with NcFile(golden_path(name)) as g:
        names = list(g._f.variables)
        assert len(names) == N_VARIABLES[name]
        for v in names:
            ref = g.get(v)
            assert v in out, f"{name}: golden variable {v} is not produced"
            got = np.asarray(out[v])
            assert got.shape == ref.shape, f"{name}: {v} has shape {got.shape}, the golden file {ref.shape}"
            worst[v] = rel_err(got, ref)
    bad = {k: e for k, e in worst.items() if not e < FLOAT32_TOL}
    assert not bad, f"{name}: beyond float32 rounding: {bad}"
    return worst


def test_golden_table_is_complete():
    files = sorted(f for f in os.listdir(GOLDEN_DIR) if f.endswith("_out_REFERENCE.nc"))
    assert files == sorted(os.path.basename(golden_path(n)) for n in GOLDEN_CASES)


@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_oracle_reproduces_reference_golden(name, oracle_lib):
    config = make_golden_config(name)
    if GOLDEN_CASES[name][0] == "rrtmg":
        if not oracle_lib.have_ref_rrtm():
            pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
        backend = oracle_lib.make_rrtmg_backend(config)
    else:
        backend = oracle_lib.backend
    flux, th, _ = run_case(config, backend)
    worst = check_against_golden(name, flux_to_output_dict(config, th, flux))
    print(name, "oracle vs golden: max", max(worst.values()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_hip_reproduces_reference_golden(name):
    config = make_golden_config(name)
    flux, th, rad = run_case(config, "hip")
    worst = check_against_golden(name, flux_to_output_dict(config, th, flux))
    rad.close()
    print(name, "HIP vs golden: max", max(worst.values()))
