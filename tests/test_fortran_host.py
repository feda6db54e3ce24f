"""The Fortran host layer (ecrad_amd/fortran, built with amdflang): ISO_C_BINDING types must match the
C-ABI byte for byte (CPU), and the Fortran driver calling radiation_hip() block by block must
reproduce the oracle (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import compare_flux, load_meridian, make_config, rel_err, run_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "ecrad_amd", "fortran")
HAVE_FLANG = os.path.exists("/opt/rocm/bin/amdflang")


def _build():
    if not (os.path.exists(os.path.join(FDIR, "abi_check")) and os.path.exists(os.path.join(FDIR, "ecrad_hip_driver"))):
        if not HAVE_FLANG:
            pytest.skip("amdflang not available and Fortran host not prebuilt")
        import __graft_entry__ as g
        g.build()


def test_fortran_interoperable_types_match_c_abi():
    _build()
    p = subprocess.run([os.path.join(FDIR, "abi_check")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ABI OK" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("solver,nblocksize", [("Tripleclouds", 8), ("McICA", 32), ("Homogeneous", 5)])
def test_fortran_driver_matches_oracle(tmp_path, oracle_lib, solver, nblocksize):
    """driver/ecrad_driver.F90-style loop: radiation_hip(ncol,nlev,istartcol,iendcol,...) over blocks."""
    _build()
    from ecrad_amd.casefile import read_records, write_case
    from ecrad_amd.interface import setup_radiation
    config = make_config(solver)
    setup_radiation(config)
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    gas.set_units(1)
    th.calc_saturation_wrt_liquid()
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    write_case(case, config, ncol, nlev, sl, th, gas, cloud, aer)
    p = subprocess.run([os.path.join(FDIR, "ecrad_hip_driver"), case, out, str(nblocksize)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Time elapsed in radiative transfer" in p.stdout
    got = read_records(out)
    f_ora, _, _ = run_case(make_config(solver), oracle_lib.backend)
    for name, a in f_ora.arrays.items():
        if name in got:
            assert rel_err(got[name], a) < 1e-8, name
    assert {"lw_up", "sw_dn", "cloud_cover_sw"} <= set(got)
