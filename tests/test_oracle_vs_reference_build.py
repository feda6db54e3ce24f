"""Cross-check of the oracle against the reference's OWN code, end to end, for the configurations that have no golden file --
above all the SPARTACUS solvers, whose restatement (oracle/oracle_spartacus.c) is otherwise pinned only
piecewise.

tests/_build/reference/ecrad_ref (tools/build_dropin.py --reference, built where /root/reference exists) is ecmwf-ifs/ecrad
1.7.1 compiled UNMODIFIED with amdflang: driver, namelist reader, setup_radiation, every solver.  The one thing under it that
is not the reference's is the netCDF library: this image has no libnetcdff, and utilities/easy_netcdf.F90 is linked against this
repo's netcdf module (ecrad_amd/fortran/netcdf.F90 + nc_classic.c, tests/test_fortran_netcdf.py), which only moves files.
By the letter of the task that makes this a CROSS-CHECK, NOT A PIN (the reference build rests on a library of the repo's);
it is recorded as what it is: every variable of the reference's double-precision output file of each target of
test/ifs/Makefile against the oracle through the Python host."""
import os
import subprocess

import numpy as np
import pytest

from ecrad_amd.driver import flux_to_output_dict
from ecrad_amd.ncfile import NcFile
from helpers import make_config, make_config_rrtmg, rel_err, run_case
from test_fortran_dropin import MERIDIAN, OTHER_TARGETS, RRTMG, TARGETS, write_namelist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "_build", "reference", "ecrad_ref")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="tests/_build/reference/ecrad_ref has not been built (tools/build_dropin.py --reference)")

# what the two may differ by: rounding (flang -O3 vs gcc, contraction) amplified by the solver's own conditioning -- the
# 9x9 / 6x6 matrix exponentials and unpivoted solves of SPARTACUS (the longwave 3-D terms most: radiation_spartacus_lw.F90:700-740)
TOL = {"test_spartacus": 1.0e-8, "test_spartacus_maxentr": 1.0e-8, "test_ecckd_spartacus": 1.0e-7}


def run_reference(tmp_path, name, family, edits):
    nam, out = str(tmp_path / f"{name}.nam"), str(tmp_path / f"{name}_ref.nc")
    write_namelist(nam, family, edits)
    text = open(nam).read()
    assert text.count("do_write_double_precision = false") == 1
    open(nam, "w").write(text.replace("do_write_double_precision = false", "do_write_double_precision = true"))
    env = dict(os.environ, OMP_NUM_THREADS="4", OMP_STACKSIZE="1G")
    p = subprocess.run(f"ulimit -s unlimited; exec {REF} {nam} {MERIDIAN} {out}", shell=True, capture_output=True, text=True,
                       cwd=str(tmp_path), env=env, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    return out


@pytest.mark.parametrize("target", sorted(OTHER_TARGETS))
def test_oracle_against_the_unmodified_reference_executable(tmp_path, target, oracle_lib):
    family, edits, (fam, solver, kw) = OTHER_TARGETS[target]
    if fam == "rrtmg" and not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    out = run_reference(tmp_path, target, family, edits)
    kw = dict(do_save_spectral_flux=True, do_lw_aerosol_scattering=False, **kw)
    cfg = make_config_rrtmg(solver, **kw) if fam == "rrtmg" else make_config(solver, **kw)
    flux, th, _ = run_case(cfg, oracle_lib.make_rrtmg_backend(cfg) if fam == "rrtmg" else oracle_lib.backend)
    want = flux_to_output_dict(cfg, th, flux)
    worst = {}
    with NcFile(out) as o:
        names = list(o._f.variables)
        assert len(names) >= 20
        for v in names:
            assert v in want, f"{target}: the oracle's host does not produce {v}"
            got, ref = o.get(v), np.asarray(want[v])
            assert got.shape == ref.shape, (v, got.shape, ref.shape)
            worst[v] = rel_err(ref, got)
    tol = TOL.get(target, 1.0e-10)
    bad = {k: e for k, e in worst.items() if not e < tol}
    assert not bad, f"{target}: oracle vs the reference executable: {bad}"
    print(target, "oracle vs reference executable: max", max(worst.values()))


def test_the_reference_executable_reproduces_its_own_golden_file(tmp_path):
    """(that the build is the reference: test_tripleclouds through it equals test/ifs/ecrad_meridian_tripleclouds_out_REFERENCE.nc)"""
    family, edits = TARGETS["tripleclouds"]
    nam, out = str(tmp_path / "tc.nam"), str(tmp_path / "tc_out.nc")
    write_namelist(nam, family, edits)
    env = dict(os.environ, OMP_NUM_THREADS="4", OMP_STACKSIZE="1G")
    p = subprocess.run(f"ulimit -s unlimited; exec {REF} {nam} {MERIDIAN} {out}", shell=True, capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    golden = os.path.join(ROOT, "tests", "golden", "ecrad_meridian_tripleclouds_out_REFERENCE.nc")
    with NcFile(golden) as g, NcFile(out) as o:
        assert sorted(g._f.variables) == sorted(o._f.variables)
        for v in g._f.variables:
            assert rel_err(o.get(v), g.get(v)) < 2.0e-7, v


def test_the_reference_driver_writes_netcdf4_through_the_repos_netcdf_module(tmp_path):
    """SURVEY section 8 row f4 in the FORTRAN host: the reference's unmodified driver with `do_write_hdf5 = true` (driver/
    ecrad_driver.F90:400 -> save_fluxes(..., is_hdf5_file) -> utilities/easy_netcdf.F90:212-245 -> nf90_create(NF90_HDF5)) on top of
    this repo's netcdf module: the output is a netCDF-4 / HDF5 file (nc_classic.c: ecnc_h5_enddef) that the HDF5 library of this
    image reads, holding the same variables, dimensions and NUMBERS as the classic file of the same run."""
    from test_hdf5_output import H5
    h5 = H5()
    family, edits = TARGETS["tripleclouds"]
    outs = {}
    for tag, hdf5 in (("classic", "false"), ("nc4", "true")):
        nam, out = str(tmp_path / f"{tag}.nam"), str(tmp_path / f"{tag}_out.nc")
        write_namelist(nam, family, edits)
        text = open(nam).read()
        assert text.count("&radiation_driver\n") == 1
        open(nam, "w").write(text.replace("&radiation_driver\n", f"&radiation_driver\ndo_write_hdf5 = {hdf5},\n"))
        env = dict(os.environ, OMP_NUM_THREADS="4", OMP_STACKSIZE="1G")
        p = subprocess.run(f"ulimit -s unlimited; exec {REF} {nam} {MERIDIAN} {out}", shell=True, capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=900)
        assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
        outs[tag] = out
    assert open(outs["classic"], "rb").read(3) == b"CDF" and open(outs["nc4"], "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    f = h5.open(outs["nc4"])
    with NcFile(outs["classic"]) as c:
        names, dims = set(c._f.variables), dict(c.dims())
        assert len(names) >= 20 and set(h5.names(f)) == names | set(dims)
        for n in names:
            _, a, size = h5.read(f, n)
            ref = c.get(n)
            assert size == c._f.variables[n].data.dtype.itemsize and np.array_equal(a, np.asarray(ref, dtype=np.float64)), n
            for att in ("units", "long_name"):
                if att in c._f.variables[n]._attributes:
                    assert h5.string_attr(f, n, att) == c._f.variables[n]._attributes[att].decode(), (n, att)
        for dn, n in dims.items():
            d, a, _ = h5.read(f, dn, keep=True)
            assert a.shape == (n,) and h5.hl.H5DSis_scale(d) > 0, dn
            h5.h5.H5Dclose(d)
    h5.h5.H5Fclose(f)


# ---- SPARTACUS with two regions (config%nregions = 2) -------------------------------------------------------------------------
_SP2 = {"sw_solver_name": '"SPARTACUS"', "lw_solver_name": '"SPARTACUS"', "n_regions": "2"}
TWO_REGION_CASES = {
    # name: (family, namelist edits, Python-host keywords, spectra compared, tolerance)
    "ecckd_1d": ({}, dict(_SP2, do_3d_effects="false"), ("ecckd", dict(do_3d_effects=False)), ("sw", "lw"), 1.0e-9),
    "ecckd_1d_beta_overlap": ({}, dict(_SP2, do_3d_effects="false", use_beta_overlap="true"),
                              ("ecckd", dict(do_3d_effects=False, use_beta_overlap=True)), ("sw", "lw"), 1.0e-9),
    "rrtmg_1d": (RRTMG, dict(_SP2, do_3d_effects="false", do_sw_delta_scaling_with_gases="false"),
                 ("rrtmg", dict(do_3d_effects=False, do_sw_delta_scaling_with_gases=False)), ("sw", "lw"), 1.0e-9),
    "ecckd_3d_explicit_entrapment": ({}, dict(_SP2, do_3d_effects="true"), ("ecckd", dict(do_3d_effects=True)), ("sw",), 1.0e-7),
    "rrtmg_3d_maximum_entrapment": (RRTMG, dict(_SP2, do_3d_effects="true", sw_entrapment_name='"Maximum"', do_sw_delta_scaling_with_gases="false"),
                                    ("rrtmg", dict(do_3d_effects=True, i_3d_sw_entrapment=4, do_sw_delta_scaling_with_gases=False)), ("sw",), 1.0e-8),
}


@pytest.mark.parametrize("case", sorted(TWO_REGION_CASES))
def test_oracle_with_two_spartacus_regions_against_the_unmodified_reference_executable(tmp_path, case, oracle_lib):
    """The oracle runs config%nregions = 2 as three regions of which the third is EMPTY (oracle/oracle_cloud.c:
    oracle_calc_region_properties_2) -- the construction the HIP kernels use.  That it is the reference's two-region scheme is
    checked here against the reference's own code run with n_regions = 2: every variable of its double-precision output file.
    With 3-D effects only the shortwave: the reference's longwave is unusable at two regions (radiation_spartacus_lw.F90:509-519
    reads the unassigned edge_length(3,jlev) and stores into transfer_rate(1,3) of a 2 x 2 array; it returns 1e27 W m-2 here)."""
    family, edits, (fam, kw), spectra, tol = TWO_REGION_CASES[case]
    if fam == "rrtmg" and not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    out = run_reference(tmp_path, case, family, edits)
    kw = dict(do_save_spectral_flux=True, do_lw_aerosol_scattering=False, nregions=2, **kw)
    cfg = make_config_rrtmg("SPARTACUS", **kw) if fam == "rrtmg" else make_config("SPARTACUS", **kw)
    assert cfg.nregions == 2
    flux, th, _ = run_case(cfg, oracle_lib.make_rrtmg_backend(cfg) if fam == "rrtmg" else oracle_lib.backend)
    want = flux_to_output_dict(cfg, th, flux)
    worst = {}
    with NcFile(out) as o:
        names = list(o._f.variables)
        assert len(names) >= 20
        for v in names:
            if "lw" in v and "lw" not in spectra:
                continue
            assert v in want, f"{case}: the oracle's host does not produce {v}"
            got, ref = o.get(v), np.asarray(want[v])
            assert got.shape == ref.shape, (v, got.shape, ref.shape)
            worst[v] = rel_err(ref, got)
    bad = {k: e for k, e in worst.items() if not e < tol}
    assert not bad, f"{case}: oracle vs the reference executable: {bad}"
    print(case, "two regions, oracle vs reference executable: max", max(worst.values()))
