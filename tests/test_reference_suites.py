"""The reference's two other test directories, next to test/ifs:

* test/ckdmip -- 50 clear-sky CKDMIP profiles, with LINE-BY-LINE fluxes of the same profiles (longwave, and shortwave at five
  solar zenith angles) in two more files.  The reference compares by eye (Matlab scripts); here the comparison is a
  known-answer test of the whole path (driver -> gas optics -> cloudless solvers) against an INDEPENDENT calculation: the
  thresholds below are what the ecCKD / RRTMG gas models deliver against line-by-line (a few tenths of a W m-2 at the
  boundaries) with a factor of two of margin -- a hundred times tighter than what a wrong unit, a missing gas or a mis-indexed
  table would do.  Input variable names (`vmr_suffix_str`), a black surface and the solar zenith
  angle come from the driver namelist, as in the reference's run.
* test/i3rc -- the I3RC cumulus profile (164 levels, cloud effective sizes, gases as scalars) over 46 solar zenith angles,
  which the reference runs through SPARTACUS (3-D and 1-D, every entrapment option), Tripleclouds and McICA on RRTMG's
  spectra: every target of that Makefile, HIP against the oracle.  (The file `i3rc_mls_cumulus_ECRAD_ICA_OUT.nc` next to it
  is the independent-column result over the full cloud-resolving-model scene, made in 2015 with spartacus-0.9.22: it cannot
  be made from the one-dimensional profile and is not a golden vector of this path.)

The four data files are byte-identical copies of the reference's (tests/golden/{ckdmip,i3rc}/, data/README.md); the two
namelists are written for these tests."""
import os

import numpy as np
import pytest
from scipy.io import netcdf_file

from ecrad_amd.config import (Config, IEntrapmentEdgeOnly, IEntrapmentExplicit, IEntrapmentExplicitNonFractal, IGasModelECCKD,
                              IGasModelIFSRRTMG, ISolverMcICA, ISolverTripleclouds)
from ecrad_amd.driver import DriverConfig, out_of_physical_bounds, read_input
from helpers import DATA_DIR, GOLDEN_DIR, compare_flux

CKDMIP = os.path.join(GOLDEN_DIR, "ckdmip")
I3RC = os.path.join(GOLDEN_DIR, "i3rc")
MU0 = (0.1, 0.3, 0.5, 0.7, 0.9)
# W m-2 over the 50 profiles: largest |mean difference| and rms difference at the top of the atmosphere / the surface, and the
# largest difference of the net longwave flux anywhere in a profile.  Measured with the oracle: ecCKD 0.03 / 0.42 (LW),
# 0.51 / 0.53 (SW, worst at mu0 = 0.1), 6.1; RRTMG 0.40 / 0.76, 1.3 / 1.7, 9.2.
ECCKD_VS_LBL = dict(lw_mean=0.15, lw_rms=0.8, sw_mean=0.8, sw_rms=0.8, net_max=10.0)
RRTMG_VS_LBL = dict(lw_mean=0.8, lw_rms=1.5, sw_mean=2.5, sw_rms=3.0, net_max=15.0)


# ---- CKDMIP ------------------------------------------------------------------------------------------------------------
def _ckdmip_run(backend_of, gas_model, mu0, do_sw=True, do_lw=True):
    nam = os.path.join(CKDMIP, "ckdmip.nam")
    config, dc = Config.read(nam), DriverConfig.read(nam)
    config.directory_name = DATA_DIR
    config.i_gas_model_sw = config.i_gas_model_lw = gas_model
    config.do_sw, config.do_lw = do_sw, do_lw                  # (the reference's make targets: do_sw=false / do_lw=false)
    dc.cos_sza_override = mu0
    def inputs():
        got = read_input(os.path.join(CKDMIP, "ckdmip_evaluation1_concentrations_present_reduced.nc"), config, dc)
        assert got[0] == 50 and got[1] == 54
        return got
    return _run(config, backend_of(config), inputs)


def _run(config, backend, make_inputs):
    """As the driver does: set-up first (the input reader wants the consolidated configuration: do_clouds, ...), then read."""
    from ecrad_amd.interface import Radiation
    from ecrad_amd.types import Flux
    rad = Radiation(config, backend=backend)
    ncol, nlev, sl, th, gas, cloud, aer = make_inputs()
    rad.set_gas_units(gas)
    th.calc_saturation_wrt_liquid()
    flux = Flux.allocate(config, ncol, nlev)
    rad.radiation(ncol, nlev, 1, ncol, sl, th, gas, cloud, aer, flux)
    if hasattr(rad, "close"):
        rad.close()
    return flux


def _lbl(name, var, imu=None):
    f = netcdf_file(os.path.join(CKDMIP, f"ckdmip_evaluation1_{name}_fluxes_present_reduced.nc"), "r", mmap=False)
    a = np.asarray(f.variables[var].data, dtype=np.float64)
    return (a if imu is None else a[:, imu, :]).T               # (half_level, column)


def _check_against_line_by_line(flux, mu0s, lw_mean, lw_rms, sw_mean, sw_rms, net_max):
    """Bias and rms difference (W m-2) over the 50 profiles at the top of the atmosphere and at the surface."""
    out = {}
    if flux["lw"] is not None:
        f = flux["lw"]
        for name, mine, lev in (("lw_up_toa", f.lw_up, 0), ("lw_dn_surf", f.lw_dn, -1)):
            d = mine[lev] - _lbl("lw", "flux_" + name[3:5] + "_lw")[lev]
            out[name] = (float(d.mean()), float(np.sqrt((d * d).mean())))
            assert abs(d.mean()) < lw_mean and np.sqrt((d * d).mean()) < lw_rms, (name, out[name])
        hr_err = np.abs((f.lw_dn - f.lw_up) - (_lbl("lw", "flux_dn_lw") - _lbl("lw", "flux_up_lw")))
        out["net_lw_profile_max"] = float(hr_err.max())         # (the mesosphere is where CKD models are worst)
        assert hr_err.max() < net_max, hr_err.max()         # net flux profile everywhere (the mesosphere is where CKD models are worst)
    for mu0 in mus(mu0s, flux):
        f, imu = flux[mu0], MU0.index(mu0)
        assert np.abs(f.sw_dn[0] - _lbl("sw", "flux_dn_sw", imu)[0]).max() < 1e-3          # same sun at the top
        for name, mine, var, lev in (("sw_up_toa", f.sw_up, "flux_up_sw", 0), ("sw_dn_surf", f.sw_dn, "flux_dn_sw", -1),
                                     ("sw_dn_direct_surf", f.sw_dn_direct, "flux_dn_direct_sw", -1)):
            d = mine[lev] - _lbl("sw", var, imu)[lev]
            out[f"{name}@{mu0}"] = (float(d.mean()), float(np.sqrt((d * d).mean())))
            assert abs(d.mean()) < sw_mean and np.sqrt((d * d).mean()) < sw_rms, (name, mu0, out[f"{name}@{mu0}"])
    return out


def mus(mu0s, flux):
    return [m for m in mu0s if m in flux]


def test_ckdmip_ecckd_oracle_against_line_by_line(oracle_lib):
    flux = {"lw": _ckdmip_run(lambda c: oracle_lib.backend, IGasModelECCKD, 0.5, do_sw=False)}
    for mu0 in MU0:
        flux[mu0] = _ckdmip_run(lambda c: oracle_lib.backend, IGasModelECCKD, mu0, do_lw=False)
    out = _check_against_line_by_line(flux, MU0, **ECCKD_VS_LBL)
    print(out)


def test_ckdmip_rrtmg_oracle_against_line_by_line(oracle_lib):
    if not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    flux = {"lw": _ckdmip_run(oracle_lib.make_rrtmg_backend, IGasModelIFSRRTMG, 0.5, do_sw=False),
            0.5: _ckdmip_run(oracle_lib.make_rrtmg_backend, IGasModelIFSRRTMG, 0.5, do_lw=False)}
    out = _check_against_line_by_line(flux, MU0, **RRTMG_VS_LBL)       # (RRTMG is the older, coarser model)
    print(out)


@pytest.mark.gpu
@pytest.mark.parametrize("gas_model", ["ecckd", "rrtmg"])
def test_ckdmip_hip_against_line_by_line_and_oracle(gas_model, oracle_lib):
    rrtmg = gas_model == "rrtmg"
    if rrtmg and not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    model = IGasModelIFSRRTMG if rrtmg else IGasModelECCKD
    ora = oracle_lib.make_rrtmg_backend if rrtmg else (lambda c: oracle_lib.backend)
    flux = {"lw": _ckdmip_run(lambda c: "hip", model, 0.5, do_sw=False)}
    for mu0 in (0.1, 0.5, 0.9):
        flux[mu0] = _ckdmip_run(lambda c: "hip", model, mu0, do_lw=False)
    if rrtmg:
        _check_against_line_by_line(flux, MU0, **RRTMG_VS_LBL)
    else:
        _check_against_line_by_line(flux, MU0, **ECCKD_VS_LBL)
    both = _ckdmip_run(lambda c: "hip", model, 0.3)                       # both spectra in one call, against the oracle
    worst = compare_flux(both, _ckdmip_run(ora, model, 0.3), 1.0)
    # (1e-8 on the broadband profiles, the bar itself on the per-g-point values: see tests/test_reference_targets.py)
    bad = {k: v for k, v in worst.items() if v > (1.0e-6 if k.endswith(("_g", "_band", "_canopy")) else 1.0e-8)}
    assert not bad, bad


# ---- I3RC ---------------------------------------------------------------------------------------------------------------
def _cos_sza_46():
    """Solar zenith angles 0, 2, ..., 88 degrees and cos = 0.01 (the reference's duplicate_profiles.sh, to its six digits)."""
    return np.array([float(f"{np.cos(np.radians(a)):.6g}") for a in range(0, 90, 2)] + [0.01])


# Makefile target -> (edits of the radiation namelist, edits of the driver namelist)
I3RC_TARGETS = {
    "3reg_3d": (dict(do_3d_effects=True, do_3d_lw_multilayer_effects=True), {}),
    "3reg_1d": (dict(do_3d_effects=False, do_3d_lw_multilayer_effects=False), {}),
    "3reg_3d_clustering": (dict(do_3d_effects=True), dict(effective_size_scaling=1.449)),
    "3reg_3d_explicit": (dict(i_3d_sw_entrapment=IEntrapmentExplicit), {}),
    "3reg_3d_nonfractal": (dict(i_3d_sw_entrapment=IEntrapmentExplicitNonFractal), {}),
    "3reg_3d_edgeonly": (dict(i_3d_sw_entrapment=IEntrapmentEdgeOnly), {}),
    "3reg_1d_explicit": (dict(do_3d_effects=False, i_3d_sw_entrapment=IEntrapmentExplicit), {}),
    "3reg_1d_edgeonly": (dict(do_3d_effects=False, i_3d_sw_entrapment=IEntrapmentEdgeOnly), {}),
    "3reg_3d_explicit_ohf1": (dict(i_3d_sw_entrapment=IEntrapmentExplicit, overhang_factor=1.0), {}),
    "tc": (dict(do_3d_effects=False, i_solver_sw=ISolverTripleclouds, i_solver_lw=ISolverTripleclouds), {}),
    "mcica": (dict(i_solver_sw=ISolverMcICA, i_solver_lw=ISolverMcICA), {}),
}


def _i3rc_run(backend_of, target):
    nam = os.path.join(I3RC, "i3rc.nam")
    config, dc = Config.read(nam), DriverConfig.read(nam)
    config.directory_name = DATA_DIR
    edits, driver_edits = I3RC_TARGETS[target]
    for k, v in edits.items():
        assert hasattr(config, k), k
        setattr(config, k, v)
    for k, v in driver_edits.items():
        setattr(dc, k, v)
    mu0 = _cos_sza_46()

    def inputs():
        from test_hip_parity import _replicate
        one = read_input(os.path.join(I3RC, "i3rc_mls_cumulus.nc"), config, dc)
        assert one[:2] == (1, 164)
        got = _replicate(one, mu0.size)
        got[2].cos_sza = mu0.copy()
        got[3].calc_saturation_wrt_liquid()
        assert not out_of_physical_bounds(1, mu0.size, False, *got[2:], out=lambda m: None)
        return got
    return _run(config, backend_of(config), inputs), mu0


def test_i3rc_oracle_three_dimensional_effects_have_the_known_signs(oracle_lib):
    """Hogan et al. (2016), the figures this directory of the reference reproduces: 3-D effects (cloud-side illumination)
    raise the reflected sunlight of a cumulus field at low sun and lower it (entrapment / side escape) at overhead sun, and
    raise the longwave cloud radiative effect at the surface."""
    if not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    f3, mu0 = _i3rc_run(oracle_lib.make_rrtmg_backend, "3reg_3d")
    f1, _ = _i3rc_run(oracle_lib.make_rrtmg_backend, "3reg_1d")
    cre3 = (f3.sw_up[0] - f3.sw_up_clear[0]) / (1366.0 * mu0)           # TOA shortwave cloud radiative effect, per unit incoming
    cre1 = (f1.sw_up[0] - f1.sw_up_clear[0]) / (1366.0 * mu0)
    assert np.all(cre1 > 0.0) and np.all(cre3 > 0.0)
    low, high = mu0 < 0.35, mu0 > 0.95
    assert np.all(cre3[low & (mu0 > 0.02)] > cre1[low & (mu0 > 0.02)]) and np.all(cre3[high] < cre1[high])
    assert np.abs(f3.sw_up_clear - f1.sw_up_clear).max() < 1e-9             # (the clear-sky calculation does not know about sides)
    lw3, lw1 = f3.lw_dn[-1] - f3.lw_dn_clear[-1], f1.lw_dn[-1] - f1.lw_dn_clear[-1]
    assert np.all(lw3 > lw1) and np.all(lw1 > 0.0)
    assert np.abs(np.diff(f3.lw_up[0])).max() < 1e-9                        # the longwave does not depend on the sun
    assert 0.2 < f3.arrays["cloud_cover_sw"][0] < 0.3                       # (the scene's cloud cover is 0.23)


@pytest.mark.gpu
@pytest.mark.parametrize("target", sorted(I3RC_TARGETS))
def test_i3rc_target_hip_matches_oracle(target, oracle_lib):
    if not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    f_hip, _ = _i3rc_run(lambda c: "hip", target)
    f_ora, _ = _i3rc_run(oracle_lib.make_rrtmg_backend, target)
    worst = compare_flux(f_hip, f_ora, 1.0)
    bad = {k: v for k, v in worst.items() if v > (1.0e-6 if k.endswith(("_g", "_band", "_canopy")) else 1.0e-8)}
    assert not bad, bad


# ---- the driver's namelist names and its check of the inputs -------------------------------------------------------------
def test_driver_namelist_uses_the_references_names(tmp_path):
    nam = tmp_path / "d.nam"
    nam.write_text("&radiation_driver\nfractional_std = 0.75,\noverlap_decorr_length = 1500.0,\ninv_effective_size = 0.002,\n"
                   "low_inv_effective_size = 0.004,\nsw_albedo = 0.2,\nlw_emissivity = 0.9,\nq_liquid_scaling = 0.5,\n"
                   "skin_temperature = 290.0,\ncos_solar_zenith_angle = 0.25,\ndo_correct_unphysical_inputs = true,\n"
                   "co2_scaling = 2.0,\n/\n")
    dc = DriverConfig.read(str(nam))
    assert (dc.fractional_std_override, dc.overlap_decorr_length_override, dc.sw_albedo_override, dc.lw_emissivity_override,
            dc.q_liq_scaling, dc.skin_temperature_override, dc.cos_sza_override) == (0.75, 1500.0, 0.2, 0.9, 0.5, 290.0, 0.25)
    assert (dc.high_inv_effective_size_override, dc.middle_inv_effective_size_override, dc.low_inv_effective_size_override) \
        == (0.002, 0.002, 0.004)
    assert dc.do_correct_unphysical_inputs and dc.gas_scaling == {"co2": 2.0}
    nam.write_text("&radiation_driver\nhigh_inv_effective_size = 0.001,\n/\n")
    with pytest.raises(ValueError):
        DriverConfig.read(str(nam))


def test_out_of_physical_bounds_reports_and_corrects(oracle_lib):
    from helpers import load_meridian, make_config
    config = make_config("Tripleclouds")
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    th.calc_saturation_wrt_liquid()
    msgs = []
    assert not out_of_physical_bounds(1, ncol, False, sl, th, gas, cloud, aer, out=msgs.append) and not msgs
    cloud.fraction[5, 3], th.temperature_hl[7, 2], th.pressure_hl[-1, 4] = 1.5, 50.0, 120000.0
    sl.cos_sza[9] = 1.25
    assert not out_of_physical_bounds(11, ncol, True, sl, th, gas, cloud, aer, out=msgs.append) and not msgs   # other columns
    assert out_of_physical_bounds(1, ncol, False, sl, th, gas, cloud, aer, out=msgs.append)
    assert len(msgs) == 4 and cloud.fraction[5, 3] == 1.5 and not any("corrected" in m for m in msgs)
    assert any(m.startswith("*** Warning: cloud%fraction range") and "is out of physical range" in m for m in msgs)
    assert out_of_physical_bounds(1, ncol, True, sl, th, gas, cloud, aer, out=msgs.append)
    assert cloud.fraction[5, 3] == 1.0 and th.temperature_hl[7, 2] == 100.0 and sl.cos_sza[9] == 1.0
    assert th.pressure_hl[-1, 4] == 120000.0                   # pressure is reported, never clipped


def test_rrtmg_gpoint_reordering_tables_and_where_they_apply():
    """radiation_ifs_rrtm.F90:51-72, :122-130, :167-174: permutations of the g-points, used for a spectrum that SPARTACUS solves and
    for nothing else; the band table the rest of the path works with is the band of the REORDERED g-point."""
    from ecrad_amd.interface import setup_radiation
    from ecrad_amd.rrtmg import GPOINT_REORDERING_LW, GPOINT_REORDERING_SW
    from helpers import make_config_rrtmg
    assert sorted(GPOINT_REORDERING_LW) == list(range(1, 141)) and sorted(GPOINT_REORDERING_SW) == list(range(1, 113))
    c = make_config_rrtmg("SPARTACUS")
    setup_radiation(c)
    r = c.rrtmg
    assert np.array_equal(r.i_g_from_reordered_g_sw, GPOINT_REORDERING_SW) and np.array_equal(r.i_g_from_reordered_g_lw, GPOINT_REORDERING_LW)
    assert np.array_equal(c.i_band_from_reordered_g_sw, r.i_band_from_g_sw[GPOINT_REORDERING_SW - 1])
    assert np.array_equal(c.i_band_from_reordered_g_lw, r.i_band_from_g_lw[GPOINT_REORDERING_LW - 1])
    assert len(set(c.i_band_from_reordered_g_sw[:14])) > 5          # the first positions: the most transparent g-point of many bands
    c2 = make_config_rrtmg("Tripleclouds", "SPARTACUS")              # per spectrum
    setup_radiation(c2)
    assert c2.rrtmg.i_g_from_reordered_g_sw is None and c2.rrtmg.i_g_from_reordered_g_lw is not None
    assert np.array_equal(c2.i_band_from_reordered_g_sw, c2.rrtmg.i_band_from_g_sw)


def _gpoint_profiles_case(solver, make_backend):
    from helpers import make_config_rrtmg, run_case
    cfg = make_config_rrtmg(solver, do_save_spectral_flux=True, do_save_gpoint_flux=True, do_3d_effects=False,
                            i_3d_sw_entrapment=0, max_cloud_od=1.0e30)
    flux, _, rad = run_case(cfg, make_backend(cfg))
    if hasattr(rad, "close"):
        rad.close()
    return cfg, flux


def _check_gpoint_profiles_native_order(make_backend):
    """SPARTACUS on RRTMG takes the g-points reordered, but stores per-g-point spectral flux PROFILES by the native RRTMG
    g-point (radiation_ifs_rrtm.F90:139-141: i_spec_from_reordered_g => i_g_from_reordered_g).  Without 3-D effects the
    shortwave solver is Tripleclouds (tests/test_oracle_spartacus.py), which runs un-reordered: profile g of the one must be
    profile g of the other -- a permuted store would be off by whole g-points."""
    from helpers import rel_err
    csp, sp = _gpoint_profiles_case("SPARTACUS", make_backend)
    ctc, tc = _gpoint_profiles_case("Tripleclouds", make_backend)
    assert not np.array_equal(csp.i_spec_from_reordered_g_sw, np.arange(1, 113)) and sorted(csp.i_spec_from_reordered_g_sw) == list(range(1, 113))
    assert np.array_equal(csp.i_spec_from_reordered_g_sw, csp.rrtmg.i_g_from_reordered_g_sw)
    assert np.array_equal(csp.i_spec_from_reordered_g_lw, csp.rrtmg.i_g_from_reordered_g_lw)
    assert np.array_equal(ctc.i_spec_from_reordered_g_sw, np.arange(1, 113))
    perm = csp.rrtmg.i_g_from_reordered_g_sw - 1
    for name in ("sw_up_band", "sw_dn_band", "sw_dn_direct_band", "sw_up_clear_band", "sw_dn_clear_band"):
        a, b = sp.arrays[name], tc.arrays[name]
        assert a.shape == b.shape and a.shape[-1] == 112       # numpy order of (nspec, ncol, nlev+1)
        assert rel_err(a, b) < 1.0e-10, name
        assert rel_err(a[..., perm], b) > 1.0e-2, name      # ... and it IS a test of the order: permuted, the same data are far off
    # longwave: the clear-sky profiles of the two solvers agree closely (different clear-layer routines), a permutation would not
    for name in ("lw_up_clear_band", "lw_dn_clear_band"):
        assert rel_err(sp.arrays[name], tc.arrays[name]) < 1.0e-3, name


def test_gpoint_flux_profiles_of_reordered_spartacus_are_in_native_order(oracle_lib):
    if not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    _check_gpoint_profiles_native_order(lambda cfg: oracle_lib.make_rrtmg_backend(cfg))


@pytest.mark.gpu
def test_gpoint_flux_profiles_of_reordered_spartacus_are_in_native_order_hip():
    _check_gpoint_profiles_native_order(lambda cfg: "hip")
