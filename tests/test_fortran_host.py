"""The Fortran host layer (ecrad_amd/fortran, built with amdflang): ISO_C_BINDING types must match the
C-ABI byte for byte (CPU), and the Fortran driver calling radiation_hip() block by block must
reproduce the oracle and the reference's golden output (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN_DIR, compare_flux, load_meridian, make_config, make_golden_config, rel_err, run_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FDIR = os.path.join(ROOT, "ecrad_amd", "fortran")
RRTMG_DRIVER = os.path.join(ROOT, "tests", "_build", "ecrad_hip_driver_rrtmg")
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


def _run_driver(tmp_path, config, nblocksize, exe=None, extra=()):
    from ecrad_amd.casefile import read_records, write_case
    from ecrad_amd.interface import setup_radiation
    setup_radiation(config)
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    gas.set_units(0 if getattr(config, "rrtmg", None) is not None else 1)
    th.calc_saturation_wrt_liquid()
    case, out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    write_case(case, config, ncol, nlev, sl, th, gas, cloud, aer)
    p = subprocess.run([exe or os.path.join(FDIR, "ecrad_hip_driver"), case, out, str(nblocksize), "1", *extra],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "Time elapsed in radiative transfer" in p.stdout
    return read_records(out)


def _compare(got, f_ora, tol=1e-8):
    checked = 0
    for name, a in f_ora.arrays.items():
        assert name in got, f"{name} not written by the Fortran driver"
        assert rel_err(got[name], a) < tol, name
        checked += 1
    assert checked >= 20


@pytest.mark.gpu
@pytest.mark.parametrize("solver,nblocksize", [("Tripleclouds", 8), ("McICA", 32), ("Homogeneous", 5)])
def test_fortran_driver_matches_oracle(tmp_path, oracle_lib, solver, nblocksize):
    """driver/ecrad_driver.F90-style loop: radiation_hip(ncol,nlev,istartcol,iendcol,...) over blocks."""
    _build()
    got = _run_driver(tmp_path, make_config(solver), nblocksize)
    f_ora, _, _ = run_case(make_config(solver), oracle_lib.backend)
    _compare(got, f_ora)
    assert {"lw_up", "sw_dn", "cloud_cover_sw"} <= set(got)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(sw_solver="Tripleclouds", do_toa_spectral_flux=True, do_nearest_spectral_sw_albedo=True, do_nearest_spectral_lw_emiss=True),
    dict(sw_solver="SPARTACUS", max_cloud_od=12.0, do_lw_derivatives=True, do_3d_effects=True),
    dict(sw_solver="McICA", do_lw_aerosol_scattering=True, use_beta_overlap=True),
], ids=["toa_spectral_nearest", "spartacus_max_cloud_od", "mcica_lw_aerosol_scattering"])
def test_fortran_driver_forwards_the_whole_configuration(tmp_path, oracle_lib, kw):
    """Options the first version of the wrapper dropped or hard-coded: TOA spectral fluxes and their output arrays, the
    nearest-interval index tables, max_cloud_od / min_gas_od, the SPARTACUS configuration and the cloud effective sizes."""
    _build()
    kw = dict(kw)
    sw = kw.pop("sw_solver")
    got = _run_driver(tmp_path, make_config(sw, **kw), 16)
    f_ora, _, _ = run_case(make_config(sw, **kw), oracle_lib.backend)
    _compare(got, f_ora)
    if kw.get("do_toa_spectral_flux"):
        assert {"lw_up_toa_band", "sw_dn_toa_band", "sw_up_toa_band", "sw_up_toa_clear_band"} <= set(got)


@pytest.mark.gpu
def test_fortran_driver_rrtmg_reproduces_the_reference_golden(tmp_path):
    """BASELINE configs[2] from Fortran: the driver linked with the reference's own ifsrrtm library runs RRTM_INIT_140GP /
    SRTM_INIT, hands the module tables over (radiation_hip_rrtmg::fill_rrtmg_hip) and must reproduce the reference's
    golden output of its default configuration (test/ifs/ecrad_meridian_default_out_REFERENCE.nc, float32)."""
    if not os.path.exists(RRTMG_DRIVER):
        pytest.skip("tests/_build/ecrad_hip_driver_rrtmg not prebuilt (needs the reference's ifsrrtm library: oracle/build_ref_rrtm.sh)")
    from ecrad_amd.driver import flux_to_output_dict
    from ecrad_amd.types import Flux
    config = make_golden_config("default")
    got = _run_driver(tmp_path, config, 8, exe=RRTMG_DRIVER, extra=(os.path.join(ROOT, "data"),))
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    flux = Flux.allocate(config, ncol, nlev)
    for name, a in flux.arrays.items():
        assert name in got, name
        a[...] = got[name]
    from test_reference_goldens import check_against_golden
    worst = check_against_golden("default", flux_to_output_dict(config, th, flux))
    print("Fortran RRTMG driver vs the reference golden: max", max(worst.values()))
