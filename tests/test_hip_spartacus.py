"""SPARTACUS on the GPU (BASELINE configs[4]): the HIP solver kernels
(ecrad_amd/csrc/kernel_spartacus.hip), called through the C-ABI, against the oracle's restatement
(oracle/oracle_spartacus.c) on the reference's 32-column meridian slice, whose input file carries the cloud
effective sizes that drive the 3-D effects.

Parity status (see the oracle's header and DESIGN.md): the reference holds no golden output of a SPARTACUS run, so the
oracle's solver body is pinned piecewise (its matrix algebra against the reference's own radiation_matrix.F90, its
one-dimensional limit against the golden-pinned Tripleclouds); these tests pin the HIP path to that oracle.

Tolerances.  Double precision: 1e-8, as for every other solver.  Single precision (i_precision = single, the reference's
PARKIND1_SINGLE build): the solver carries float rounding (6e-8) through a 9x9 matrix exponential and LU solves without
pivoting -- the reference itself warns that SPARTACUS may be unstable in single precision (radiation_config.F90:1144) --
so the bar is set by what the single-precision ORACLE itself does when only its last bits change (its build with
floating-point contraction against the plain one), with a factor 4 margin, and the HIP path must also stay within
2e-3 of the double-precision answer."""
import os

import numpy as np
import pytest

from ecrad_amd.config import (IEntrapmentEdgeOnly, IEntrapmentExplicit, IEntrapmentExplicitNonFractal,
                              IEntrapmentMaximum, IEntrapmentZero, IPrecisionSingle)
from ecrad_amd.types import Flux
from helpers import compare_flux, make_config, rel_err, run_case

pytestmark = pytest.mark.gpu
TOL = 1.0e-8

CASES = {
    "explicit": dict(i_3d_sw_entrapment=IEntrapmentExplicit),
    "explicit_non_fractal": dict(i_3d_sw_entrapment=IEntrapmentExplicitNonFractal),
    "edge_only": dict(i_3d_sw_entrapment=IEntrapmentEdgeOnly),
    "zero": dict(i_3d_sw_entrapment=IEntrapmentZero),
    "maximum": dict(i_3d_sw_entrapment=IEntrapmentMaximum),
    "no_3d": dict(do_3d_effects=False),
    "no_3d_zero_uncapped": dict(do_3d_effects=False, i_3d_sw_entrapment=IEntrapmentZero, max_cloud_od=1.0e30),
    "expm_everywhere": dict(use_expm_everywhere=True),
    "lw_multilayer": dict(do_3d_lw_multilayer_effects=True),
    "no_3d_lw_multilayer": dict(do_3d_effects=False, do_3d_lw_multilayer_effects=True),
    "clear_to_thick": dict(clear_to_thick_fraction=0.3, overhang_factor=1.0, overhead_sun_factor=0.06),
    "no_side_emissivity": dict(do_lw_side_emissivity=False),
    "no_lw_scattering": dict(do_lw_cloud_scattering=False, do_lw_aerosol_scattering=False),
    "lw_aerosol_scattering": dict(do_lw_aerosol_scattering=True),
    "noaer": dict(use_aerosols=False),
    "tight_caps": dict(max_3d_transfer_rate=1.0, max_gas_od_3d=0.5),
    "noclear": dict(do_clear=False, do_sw_direct=False),
    "lognormal_beta": dict(i_cloud_pdf_shape=0, use_beta_overlap=True),
    "sw64": dict(gas_optics_sw_override_file_name="ecckd-1.2_sw_climate_window-64b_ckd-definition.nc"),
    # 96 shortwave g-points: three launches of 32 lanes per column (see also the RRTMG cases of test_hip_rrtmg.py)
    "sw96": dict(gas_optics_sw_override_file_name="ecckd-1.4_sw_climate_vfine-96b_ckd-definition.nc"),
    # spectral flux profiles (do_save_spectral_flux is on in both of the reference's test namelists, hence in its
    # test_spartacus / test_ecckd_spartacus runs): one interval per g-point, and per band (per-g temporaries summed afterwards)
    "spectral": dict(do_save_spectral_flux=True),
    "spectral_bands": dict(do_save_spectral_flux=True, do_cloud_aerosol_per_sw_g_point=False, do_cloud_aerosol_per_lw_g_point=False),
    "spectral_noclear_no3d": dict(do_save_spectral_flux=True, do_clear=False, do_3d_effects=False),
    "sw96_tight_caps": dict(gas_optics_sw_override_file_name="ecckd-1.4_sw_climate_vfine-96b_ckd-definition.nc", max_gas_od_3d=0.5),
    # config%nregions = 2 (one homogeneous cloudy region; the third region of the kernels' arrays stays empty).  The oracle's
    # two-region path is checked against the reference's own executable in tests/test_oracle_vs_reference_build.py, the HIP path
    # against that executable in tests/test_fortran_dropin.py; here the longwave with 3-D effects is covered as well (the
    # reference's own is unusable at two regions)
    "two_regions": dict(nregions=2),
    "two_regions_no_3d": dict(nregions=2, do_3d_effects=False),
    "two_regions_maximum": dict(nregions=2, i_3d_sw_entrapment=IEntrapmentMaximum),
    "two_regions_explicit_non_fractal_clear_to_thick": dict(nregions=2, i_3d_sw_entrapment=IEntrapmentExplicitNonFractal, clear_to_thick_fraction=0.3),
    "two_regions_lw_multilayer_beta": dict(nregions=2, do_3d_lw_multilayer_effects=True, use_beta_overlap=True),
    "two_regions_spectral": dict(nregions=2, do_save_spectral_flux=True),
}


def _config(**kw):
    # (the reference's test namelist switches the 3-D effects off: every case here asks for them unless it says otherwise)
    return make_config("SPARTACUS", **{**dict(do_lw_derivatives=True, do_3d_effects=True), **kw})


@pytest.mark.parametrize("name", sorted(CASES))
def test_spartacus_matches_oracle(name, oracle_lib):
    f_ora, _, _ = run_case(_config(**CASES[name]), oracle_lib.backend)
    f_hip, _, _ = run_case(_config(**CASES[name]), "hip")
    for k, a in f_hip.arrays.items():
        assert np.all(np.isfinite(a)), k
    compare_flux(f_hip, f_ora, TOL)


def test_the_3d_effects_are_on_in_these_cases(oracle_lib):
    """Guard: the cases must exercise the matrix exponentials (a configuration read from the test namelist has them off)."""
    assert _config().do_3d_effects and not _config(do_3d_effects=False).do_3d_effects
    a, _, _ = run_case(_config(), oracle_lib.backend)
    b, _, _ = run_case(_config(do_3d_effects=False), oracle_lib.backend)
    day = a.arrays["sw_dn"][0] > 0
    assert np.abs(a.arrays["sw_up"][0][day] - b.arrays["sw_up"][0][day]).max() > 0.5       # W m-2
    assert np.abs(a.arrays["lw_up"][0] - b.arrays["lw_up"][0]).max() > 0.05


def test_spartacus_shortwave_with_another_longwave_solver(oracle_lib):
    """The two spectra choose their solvers independently (radiation_interface.F90:422-499)."""
    for sw, lw in (("SPARTACUS", "Tripleclouds"), ("Tripleclouds", "SPARTACUS"), ("SPARTACUS", "McICA")):
        f_ora, _, _ = run_case(make_config(sw, lw, do_3d_effects=True), oracle_lib.backend)
        f_hip, _, _ = run_case(make_config(sw, lw, do_3d_effects=True), "hip")
        compare_flux(f_hip, f_ora, TOL)


def test_spartacus_column_subrange_and_surface_first_levels(oracle_lib):
    f_all, _, _ = run_case(_config(), oracle_lib.backend)
    f_ora, _, _ = run_case(_config(), oracle_lib.backend, columns=(9, 23))
    f_sub, _, _ = run_case(_config(), "hip", columns=(9, 23))
    compare_flux(f_sub, f_ora, TOL)
    # radiation_reverse: the caller's arrays from the surface upwards
    from helpers import load_meridian
    cfg = _config()
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(cfg)
    for obj, names in ((th, ("pressure_hl", "temperature_hl")), (gas, ("mixing_ratio",)),
                       (cloud, ("mixing_ratio", "effective_radius", "fraction", "fractional_std", "overlap_param",
                                "inv_cloud_effective_size", "inv_inhom_effective_size")), (aer, ("mixing_ratio",))):
        for n in names:
            a = getattr(obj, n, None)
            if a is not None:
                setattr(obj, n, np.ascontiguousarray(a[..., ::-1, :]))     # the level axis is the second to last
    f_rev, _, _ = run_case(cfg, "hip", inputs=(ncol, nlev, sl, th, gas, cloud, aer))
    for name in ("lw_up", "lw_dn", "sw_up", "sw_dn", "sw_dn_direct", "lw_up_clear", "sw_up_clear", "lw_derivatives"):
        assert rel_err(f_rev.arrays[name][::-1], f_all.arrays[name]) < TOL, name
    for name in ("sw_dn_diffuse_surf_g", "lw_dn_surf_g", "sw_up_toa_g", "cloud_cover_sw"):
        assert rel_err(f_rev.arrays[name], f_all.arrays[name]) < TOL, name


STABLE_IN_SINGLE = ("sw_", "lw_up_clear", "lw_dn_clear", "lw_dn_surf_clear_g", "lw_up_toa_clear_g", "cloud_cover")


@pytest.mark.parametrize("name", ["explicit", "maximum", "no_3d", "lw_multilayer"])
def test_spartacus_single_precision(name, oracle_lib):
    """i_precision = single is the reference's PARKIND1_SINGLE build of the solver, instabilities included: with 3-D effects
    the LONGWAVE solver solves 6x6 systems with the (nearly singular, for optically thin g-points) exponent matrix by
    unpivoted LU and subtracts the huge particular solutions again (radiation_spartacus_lw.F90:700-740); in float that
    is chaotic -- the reference only warns (radiation_config.F90:1144).  The oracle's single-precision build shows it: a few
    per cent of the all-sky longwave values are off by more than 1e-3 and change completely when only floating-point
    contraction changes.  So: everything is checked against the single-precision oracle within that oracle's own
    last-bit sensitivity, element by element; the shortwave and the clear-sky longwave (stable) must also be as close to
    the double-precision answer (2e-3: float rounding through the 9x9 exponential and the unpivoted solves); and where the 3-D effects are off, so
    must the all-sky longwave."""
    kw = dict(CASES[name], i_precision=IPrecisionSingle)
    f_dp, _, _ = run_case(_config(**CASES[name]), oracle_lib.backend)
    f_sp, _, _ = run_case(_config(**kw), oracle_lib.make_variant_backend("sp"))
    f_sp_fma, _, _ = run_case(_config(**kw), oracle_lib.make_variant_backend("sp_fma"))
    f_hip, _, _ = run_case(_config(**kw), "hip")
    chaotic = _config(**kw).do_3d_effects
    for k, a in f_hip.arrays.items():
        ref, alt, dp = f_sp.arrays[k], f_sp_fma.arrays[k], f_dp.arrays[k]
        scale = np.maximum(np.abs(dp), 1.0e-3 * np.abs(dp).max() + 1.0e-300)
        sens = np.abs(alt - ref) / scale                      # what the oracle itself does when its last bits change
        got = np.abs(a - ref) / scale
        stable = k.startswith(STABLE_IN_SINGLE) or not chaotic
        if stable:
            assert np.all(np.isfinite(a)), k
            assert got.max() <= max(4.0 * sens.max(), 5.0e-6), (k, got.max(), sens.max())
            assert (np.abs(a - dp) / scale).max() < 2.0e-3, k      # float rounding through the 9x9 exponential and the unpivoted solves
        else:
            # element by element: within the oracle's own sensitivity there, and tight wherever the oracle is stable
            calm = sens < 1.0e-5
            assert calm.mean() > 0.8, (k, calm.mean())
            ok = np.isfinite(a[calm])
            assert ok.all(), k
            assert np.median(got[calm]) < 2.0e-6, (k, np.median(got[calm]))
            assert (got[calm] <= 1.0e-3).mean() > 0.999, (k, (got[calm] > 1.0e-3).mean())
    off = np.abs(f_sp.arrays["lw_up"] - f_dp.arrays["lw_up"]) / f_dp.arrays["lw_up"] > 1.0e-3
    print(f"{name}: single-precision ORACLE, all-sky lw_up off by more than 1e-3 from double in {100.0 * off.mean():.2f} % of the values")


def test_single_precision_division_by_reciprocal_changes_nothing_that_matters(oracle_lib):
    """The single-precision kernels divide by the hardware reciprocal (2.5 units of the last place) off the albedo / flux recurrences
    and by reciprocal + one residual correction (one unit) on them (spartacus_device.h: pdiv; Makefile: SP_FAST_DIV).  The variant
    tests/_build/variants/nopack is built without that flag -- correctly rounded float division everywhere.  On the meridian slice the two
    builds are compared with the double-precision oracle, field by field for everything that is stable in single precision: the
    shipped build's largest difference from double within 1.5 x the correctly rounded build's (+ 1e-5; and below the 2e-3 of
    test_spartacus_single_precision), its median difference within 1.25 x (+ 1e-7), and the two must differ somewhere (the variant
    IS another build)."""
    from ecrad_amd.interface import Radiation
    from helpers import load_meridian
    nopack = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_build", "variants", "nopack", "libecrad_hip.so")
    assert os.path.exists(nopack), "tests/_build/variants/nopack/libecrad_hip.so is missing: run __graft_entry__.build()"
    kw = dict(CASES["explicit"], i_precision=IPrecisionSingle)
    f_dp, _, _ = run_case(_config(**CASES["explicit"]), oracle_lib.backend)
    out = []
    for path in (None, nopack):
        config = _config(**kw)
        rad = Radiation(config, backend="hip", lib_path=path)
        ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
        rad.set_gas_units(gas)
        th.calc_saturation_wrt_liquid()
        flux = Flux.allocate(config, ncol, nlev)
        rad.radiation(ncol, nlev, 1, ncol, sl, th, gas, cloud, aer, flux)
        rad.close()
        out.append(flux)
    differ = 0.0
    for k, dp in f_dp.arrays.items():
        if not k.startswith(STABLE_IN_SINGLE):
            continue
        scale = np.maximum(np.abs(dp), 1.0e-3 * np.abs(dp).max() + 1.0e-300)
        d_fast, d_ieee = np.abs(out[0].arrays[k] - dp) / scale, np.abs(out[1].arrays[k] - dp) / scale
        differ = max(differ, float(np.abs(out[0].arrays[k] - out[1].arrays[k]).max()))
        print(f"{k}: largest / median difference from double: shipped {d_fast.max():.2e} / {np.median(d_fast):.2e}, "
              f"correctly rounded division {d_ieee.max():.2e} / {np.median(d_ieee):.2e}")
        # (the largest element is one number; the median says what the division did to the field)
        assert d_fast.max() < 2.0e-3 and d_fast.max() <= 1.5 * d_ieee.max() + 1.0e-5, (k, d_fast.max(), d_ieee.max())
        assert np.median(d_fast) <= 1.25 * np.median(d_ieee) + 1.0e-7, (k, np.median(d_fast), np.median(d_ieee))
    assert differ > 0.0, "the two builds are identical: the variant is not a build with the compiler's own float division"


@pytest.mark.parametrize("kw", [dict(), dict(max_gas_od_3d=0.5), dict(do_lw_aerosol_scattering=True)], ids=["default", "tight_caps", "aerosol_scat"])
def test_spartacus_wide_longwave_spectrum(kw, tmp_path, oracle_lib):
    """96 longwave g-points (the 32-term model with every g-point three times: the reference ships no wider ecCKD
    longwave model): three launches per kernel, broadband profiles from per-chunk partial sums, derivatives normalised
    by the surface flux of the whole spectrum afterwards, and the g-point that switches the 3-D treatment off searched
    over the whole spectrum (radiation_spartacus_lw.F90, section 3.2)."""
    from test_hip_parity import _replicated_lw_model
    kw = dict(kw, gas_optics_lw_override_file_name=_replicated_lw_model(tmp_path))
    f_ora, _, _ = run_case(_config(**kw), oracle_lib.backend)
    f_hip, _, rad = run_case(_config(**kw), "hip")
    rad.close()
    compare_flux(f_hip, f_ora, TOL)


REF_DP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_build", "reference", "ecrad_ref")
REF_SP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "_build", "reference_sp", "ecrad_ref")


@pytest.mark.skipif(not (os.path.exists(REF_DP) and os.path.exists(REF_SP)),
                    reason="tests/_build/reference{,_sp}/ecrad_ref have not been built (tools/build_dropin.py --reference [--single])")
@pytest.mark.parametrize("do_3d", [True, False], ids=["3d", "1d"])
def test_single_precision_against_the_reference_single_precision_executable(tmp_path, do_3d):
    """The yardstick of single precision that is not the repo's own: ecRad 1.7.1 compiled UNMODIFIED with -DPARKIND1_SINGLE
    (tests/_build/reference_sp/ecrad_ref, CPU) and in double (tests/_build/reference/ecrad_ref), both on the reference's
    test_ecckd_spartacus configuration, against the HIP path with i_precision = single on the same namelist.  What the
    reference's formulation allows in float shows in its own float build; the GPU must be as close to the reference's DOUBLE
    answer as the reference's own float executable is: per variable, the largest difference from double within 3x the
    executable's (+ 2e-5), for everything that is stable in single precision; the all-sky longwave with 3-D effects -- chaotic
    in float in the reference itself -- by its median.  (A cross-check in the sense of tests/test_oracle_vs_reference_build.py:
    the executables sit on the repo's netCDF module.)"""
    import subprocess
    from ecrad_amd.driver import flux_to_output_dict
    from ecrad_amd.ncfile import NcFile
    from test_fortran_dropin import MERIDIAN, write_namelist
    edits = {"sw_solver_name": '"SPARTACUS"', "lw_solver_name": '"SPARTACUS"', "do_3d_effects": "true" if do_3d else "false"}
    outs = {}
    for tag, exe in (("dp", REF_DP), ("sp", REF_SP)):
        nam, out = str(tmp_path / f"{tag}.nam"), str(tmp_path / f"{tag}.nc")
        write_namelist(nam, {}, edits)
        text = open(nam).read()
        assert text.count("do_write_double_precision = false") == 1
        open(nam, "w").write(text.replace("do_write_double_precision = false", "do_write_double_precision = true"))
        p = subprocess.run(f"ulimit -s unlimited; exec {exe} {nam} {MERIDIAN} {out}", shell=True, capture_output=True, text=True,
                           cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="4", OMP_STACKSIZE="1G"), timeout=900)
        assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
        outs[tag] = out
    cfg = _config(do_3d_effects=do_3d, i_precision=IPrecisionSingle, do_save_spectral_flux=True)
    flux, th, rad = run_case(cfg, "hip")
    rad.close()
    hip = flux_to_output_dict(cfg, th, flux)
    report = []
    with NcFile(outs["dp"]) as d, NcFile(outs["sp"]) as s:
        names = [v for v in d._f.variables if v in hip and v not in ("pressure_hl",)]
        assert len(names) >= 15, names
        for v in names:
            dp, sp, got = d.get(v).astype(np.float64), s.get(v).astype(np.float64), np.asarray(hip[v], dtype=np.float64)
            assert got.shape == dp.shape, (v, got.shape, dp.shape)
            scale = np.maximum(np.abs(dp), 1.0e-3 * np.abs(dp).max() + 1.0e-300)
            e_ref, e_hip = np.abs(sp - dp) / scale, np.abs(got - dp) / scale
            chaotic = do_3d and "lw" in v and "clear" not in v
            # (profiles per spectral interval, (column, half level, interval): a weak interval's last bits are a per cent of the 1e-3
            #  floor of `scale`; they are judged by their 99.9th percentile, the broadband variables by their maximum)
            spectral = dp.ndim == 3
            top = (lambda e: float(np.quantile(e, 0.999))) if spectral else (lambda e: float(e.max()))
            ok = bool(np.all(np.isfinite(got)))
            if chaotic:
                ok = ok and np.median(e_hip) <= 3.0 * np.median(e_ref) + 2.0e-6
            else:
                ok = ok and top(e_hip) <= 3.0 * top(e_ref) + 2.0e-5
            report.append((v, top(e_hip), top(e_ref), float(np.median(e_hip)), float(np.median(e_ref)), ok))
    for v, eh, er, mh, mr, ok in sorted(report, key=lambda r: -r[1])[:6]:
        print(f"single precision, do_3d_effects={do_3d}: {v}: |x - double| (max, or 99.9 % of a spectral profile) HIP {eh:.2e}, reference float "
              f"executable {er:.2e}; medians {mh:.2e} / {mr:.2e}{'' if ok else '  <-- FAILS'}")
    assert all(r[5] for r in report), [r for r in report if not r[5]]
