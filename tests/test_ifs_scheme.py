"""The IFS-style caller: ecrad_amd/ifs.py against

* the reference's OWN parametrisations compiled from /root/reference into oracle/_ref/libecrad_refifs.so
  (ifs/liquid_effective_radius.F90, ifs/ice_effective_radius.F90, ifs/cloud_overlap_decorr_len.F90) -- 1e-12;
* the ordinary driver: with the reference's BITIDENTITY_TESTING arguments (effective radii, overlap parameter and seeds
  passed through) `radiation_scheme` must give the net fluxes of `radiation()` on the same file, which is the
  comparison the reference's `make test_ifsdriver` / `test_ifsdriver_blocked` targets print (test/ifs/Makefile:40-51);
* itself: NPROMA-blocked == unblocked, one batch == block by block, column sub-ranges leave other columns alone.
CPU cases run the oracle as the backend (host logic); the `gpu` cases call the HIP library through the same code."""
import os

import numpy as np
import pytest

from ecrad_amd import ifs
from ecrad_amd.cases import DATA_DIR, MERIDIAN, NAMELIST, load_meridian, make_config, run_case
from ecrad_amd.ncfile import NcFile
from helpers import rel_err

NET = ("sw_up", "lw_up", "sw_up_clear", "lw_up_clear")        # where the drivers park the net fluxes


def _columns(seed=7, klon=48, klev=60):
    rng = np.random.default_rng(seed)
    p_half = np.linspace(0.0, 101325.0, klev + 1)[:, None] * (1.0 + 0.02 * rng.standard_normal(klon))[None, :]
    p = 0.5 * (p_half[:-1] + p_half[1:])
    t = 200.0 + 90.0 * (p / 101325.0) ** 0.3 + rng.standard_normal((klev, klon))
    frac = np.clip(rng.random((klev, klon)) * 1.4 - 0.5, 0.0, 1.0)
    frac[rng.random((klev, klon)) < 0.1] = 5.0e-4          # below both routines' thresholds
    ql = np.where(rng.random((klev, klon)) < 0.7, 10.0 ** rng.uniform(-8, -3.5, (klev, klon)), 0.0)
    qr = np.where(rng.random((klev, klon)) < 0.3, 10.0 ** rng.uniform(-9, -4, (klev, klon)), 0.0)
    qi = np.where(rng.random((klev, klon)) < 0.7, 10.0 ** rng.uniform(-9, -4, (klev, klon)), 0.0)
    qs = np.where(rng.random((klev, klon)) < 0.3, 10.0 ** rng.uniform(-9, -4, (klev, klon)), 0.0)
    land = (rng.random(klon) < 0.5).astype(np.float64)
    ccn_land = rng.uniform(100.0, 1500.0, klon)
    ccn_sea = rng.uniform(20.0, 300.0, klon)
    gemu = rng.uniform(-1.0, 1.0, klon)
    return p, t, frac, ql, qr, qi, qs, land, ccn_land, ccn_sea, gemu


@pytest.mark.parametrize("nradlp,lccn", [(0, True), (1, True), (2, True), (2, False)])
def test_liquid_effective_radius_is_the_references(oracle_lib, nradlp, lccn):
    if not oracle_lib.have_ref_ifs():
        pytest.skip("oracle/_ref/libecrad_refifs.so not built (needs /root/reference)")
    p, t, frac, ql, qr, qi, qs, land, ccn_land, ccn_sea, gemu = _columns()
    y = ifs.TERAD(NRADLP=nradlp, LCCNL=lccn, LCCNO=lccn)
    mine = ifs.liquid_effective_radius(y, p, t, frac, ql, qr, land, ccn_land, ccn_sea)
    ref = oracle_lib.ref_liquid_effective_radius(y, p, t, frac, ql, qr, land, ccn_land, ccn_sea)
    assert mine.shape == ref.shape and np.isfinite(ref).all()
    assert np.abs(mine / ref - 1.0).max() < 1.0e-12
    if nradlp == 2:
        assert ref.min() >= 4.0 and ref.max() <= 30.0 and np.unique(np.round(ref, 6)).size > 50


@pytest.mark.parametrize("nradip,nminice", [(0, 1), (1, 1), (2, 1), (3, 1), (3, 0)])
def test_ice_effective_radius_is_the_references(oracle_lib, nradip, nminice):
    if not oracle_lib.have_ref_ifs():
        pytest.skip("oracle/_ref/libecrad_refifs.so not built (needs /root/reference)")
    p, t, frac, ql, qr, qi, qs, land, ccn_land, ccn_sea, gemu = _columns(seed=11)
    y = ifs.TERAD(NRADIP=nradip, NMINICE=nminice)
    mine = ifs.ice_effective_radius(y, p, t, frac, qi, qs, gemu)
    ref = oracle_lib.ref_ice_effective_radius(y, p, t, frac, qi, qs, gemu)
    assert np.abs(mine / ref - 1.0).max() < 1.0e-12
    if nradip == 3:
        assert np.unique(np.round(ref, 6)).size > 50


@pytest.mark.parametrize("kdecolat", [0, 1, 2])
def test_overlap_decorrelation_length_is_the_references(oracle_lib, kdecolat):
    if not oracle_lib.have_ref_ifs():
        pytest.skip("oracle/_ref/libecrad_refifs.so not built (needs /root/reference)")
    gemu = np.linspace(-1.0, 1.0, 41)
    mine, ratio = ifs.cloud_overlap_decorr_len(gemu, kdecolat)
    ref, ref_ratio = oracle_lib.ref_cloud_overlap_decorr_len(gemu, kdecolat)
    assert np.abs(mine / ref - 1.0).max() < 1.0e-12 and ratio == ref_ratio


def test_overlap_param_from_a_decorrelation_length_both_level_orders():
    """set_overlap_param (radiation_cloud.F90:195-385): the file's own overlap_param was made this way from a 2 km
    decorrelation length by the reference's tools, to float32; surface-first input gives the mirrored array."""
    config = make_config("Tripleclouds")
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    op = ifs.set_overlap_param(th, 2000.0)
    assert op.shape == cloud.overlap_param.shape
    from ecrad_amd.types import Thermodynamics
    rev = Thermodynamics(np.ascontiguousarray(th.pressure_hl[::-1]), np.ascontiguousarray(th.temperature_hl[::-1]))
    op_rev = ifs.set_overlap_param(rev, np.full(ncol, 2000.0))
    assert np.abs(op_rev[::-1] - op).max() < 1.0e-14
    assert np.all((op > 0.0) & (op < 1.0))
    # the reference's own input file was made with decorrelation lengths of the same order: same shape of profile
    k = 100
    implied = -1.0 / np.log(cloud.overlap_param[k]) * (-np.log(op[k]) * 2000.0)       # decorrelation length of the file (m)
    assert np.all((implied > 500.0) & (implied < 5000.0))


def test_setup_radiation_scheme_translates_the_host_switches(oracle_lib):
    y = ifs.TERAD(NSWSOLVER=0, NLWSOLVER=0, NLWSCATTERING=2, NCLOUDOVERLAP=2, LAPPROXSWUPDATE=True)
    yr = ifs.TRADIATION(yrerad=y)
    from ecrad_amd.config import (IGasModelECCKD, IOverlapExponential, ISolverMcICA)
    yr.rad_config.i_gas_model_sw = yr.rad_config.i_gas_model_lw = IGasModelECCKD
    yr.rad_config.cloud_type_name = ["mie_droplet", "baum-general-habit-mixture_ice"]
    ifs.setup_radiation_scheme(yr, directory_name=DATA_DIR, backend=oracle_lib.backend)
    c = yr.rad_config
    assert c.i_solver_sw == ISolverMcICA and c.i_solver_lw == ISolverMcICA and c.i_overlap_scheme == IOverlapExponential
    assert c.do_lw_cloud_scattering and c.do_lw_aerosol_scattering and c.do_lw_derivatives and c.do_canopy_fluxes_sw
    assert c.n_aerosol_types == 12 and c.i_aerosol_type_map[:4] == [-1, -2, -3, 7] and c.use_aerosols
    assert c.n_canopy_bands_sw == 6 and c.do_nearest_spectral_lw_emiss and not c.do_nearest_spectral_sw_albedo
    assert c.sw_albedo_weights.shape == (c.n_bands_sw, 6)
    # UV and PAR weights: fractions of each g-point in the range, PAR inside 0.4-0.7 um carries ~40 % of the sun
    assert yr.nweight_uv > 0 and yr.nweight_par > 0
    assert np.all((yr.weight_par > 0) & (yr.weight_par <= 1.0 + 1e-12))
    sol = np.asarray(c.gas_optics_sw.spectral_def.solar_irradiance)
    par_fraction = (sol[yr.iband_par - 1] * yr.weight_par).sum() / sol.sum()
    assert 0.35 < par_fraction < 0.45
    with pytest.raises(ifs.ConfigError):
        ifs.setup_radiation_scheme(ifs.TRADIATION(yrerad=ifs.TERAD(NSWSOLVER=2, NLWSOLVER=1)), directory_name=DATA_DIR,
                                   backend=oracle_lib.backend)


def _net_of_ordinary_driver(backend, **kw):
    cfg = make_config("Tripleclouds", do_lw_derivatives=True, **kw)
    flux, th, _ = run_case(cfg, backend)
    return flux


def _check_against_ordinary(flux_ifs, flux, tol):
    assert rel_err(flux_ifs.sw_dn - flux_ifs.sw_up, flux.sw_dn - flux.sw_up) < tol
    assert rel_err(flux_ifs.lw_dn - flux_ifs.lw_up, flux.lw_dn - flux.lw_up) < tol
    assert rel_err(flux_ifs.sw_dn_clear - flux_ifs.sw_up_clear, flux.sw_dn_clear - flux.sw_up_clear) < tol
    assert rel_err(flux_ifs.lw_dn_clear - flux_ifs.lw_up_clear, flux.lw_dn_clear - flux.lw_up_clear) < tol
    assert rel_err(flux_ifs.lw_derivatives, flux.lw_derivatives) < tol
    for n in ("sw_dn", "lw_dn", "sw_dn_clear", "lw_dn_clear", "sw_dn_direct", "sw_dn_direct_clear"):
        assert rel_err(flux_ifs.arrays[n][-1], flux.arrays[n][-1]) < tol, n
    assert rel_err(flux_ifs.sw_dn[0], flux.sw_dn[0]) < tol


def test_ifs_driver_reproduces_the_ordinary_driver_with_bitidentity_arguments(oracle_lib, tmp_path):
    out = str(tmp_path / "ifs_out.nc")
    c, th, flux_ifs, diag = ifs.run_ifs_driver(NAMELIST, MERIDIAN, out, bitidentity=True, backend=oracle_lib.backend,
                                               directory_name=DATA_DIR)
    flux = _net_of_ordinary_driver(oracle_lib.backend)
    _check_against_ordinary(flux_ifs, flux, 1.0e-12)
    # the single-level diagnostics of RADIATION_SCHEME
    day = flux.sw_dn[0] > 0
    assert np.allclose(diag["flux_sw_direct_normal"][day] * th.pressure_hl[0, day] * 0 + diag["flux_sw_direct_normal"][day],
                       flux.sw_dn_direct[-1][day] / load_meridian(c)[2].cos_sza[day], rtol=1e-12)
    assert np.all(diag["flux_sw_direct_normal"][~day] == 0.0)
    assert np.all((diag["emissivity_out"] >= 0.8) & (diag["emissivity_out"] <= 0.99))
    assert np.all(diag["flux_par"][day] < flux.sw_dn[-1][day]) and np.all(diag["flux_par"][day] > 0.3 * flux.sw_dn[-1][day])
    assert np.all(diag["flux_uv"][day] < diag["flux_par"][day])
    assert np.all(diag["flux_par_clear"] >= diag["flux_par"] - 1e-9)
    # the file is the net-flux file of the reference's drivers (save_net_fluxes with the spectral/canopy switches off)
    with NcFile(out) as f:
        names = set(f._f.variables)
        assert {"pressure_hl", "flux_net_lw", "flux_net_sw", "flux_net_lw_clear", "flux_net_sw_clear", "lw_derivative",
                "flux_dn_sw_surf", "flux_dn_lw_surf", "flux_dn_sw_toa", "flux_dn_direct_sw_surf"} <= names
        assert not any(n.startswith("canopy") or n.startswith("spectral") for n in names)
        assert rel_err(f.get("flux_net_sw").T, flux.sw_dn - flux.sw_up) < 1e-6     # float32 file


@pytest.mark.parametrize("per_block", [False, True])
def test_blocked_driver_equals_the_unblocked_one(oracle_lib, per_block):
    kw = dict(bitidentity=True, backend=oracle_lib.backend, directory_name=DATA_DIR)
    _, _, flux_a, diag_a = ifs.run_ifs_driver(NAMELIST, MERIDIAN, None, **kw)
    _, _, flux_b, diag_b = ifs.run_ifs_driver(NAMELIST, MERIDIAN, None, blocked=True, per_block=per_block, **kw)
    for n in NET + ("lw_derivatives",):
        assert np.array_equal(flux_a.arrays[n], flux_b.arrays[n]), n
    for n in ("sw_dn", "lw_dn", "sw_dn_clear", "lw_dn_clear", "sw_dn_direct", "sw_dn_direct_clear"):
        assert np.array_equal(flux_a.arrays[n][-1], flux_b.arrays[n][-1]), n
    for n in diag_a:
        assert np.array_equal(diag_a[n], diag_b[n]), n


def test_blocked_array_layout_follows_the_reference():
    """ifs_setup_indices (driver/ifs_blocking.F90:55-282): inputs first, then outputs, then the diagnostic-only fields;
    NPROMA that does not divide the column count leaves a padded last block."""
    yr = ifs.TRADIATION()
    yr.rad_config.n_aerosol_types = 12
    ic = ifs.ifs_setup_indices(yr, 137)
    assert ic.igi == -1 and ic.imu0 == 0 and ic.iamu0 == 1 and ic.iemiss == 2 and ic.its == 4
    assert ic.iald == 14 and ic.ialp == 20 and ic.iti == 26 and ic.ipr == 26 + 137
    assert ic.ifrsod < ic.ifrso < ic.iaero < ic.iaer < ic.ioz < ic.icl4
    assert ic.ifldstot == ic.icl4 + 137
    icb = ifs.ifs_setup_indices(yr, 137, bitidentity=True)
    assert icb.ifldstot == ic.ifldstot + 137 + 137 + 136


def test_ifs_parametrisations_change_the_answer_plausibly(oracle_lib):
    """Without the bit-identity arguments the scheme derives effective radii and overlap itself (NRADLP=2, NRADIP=3,
    NDECOLAT=2): fluxes move by W m-2, not by tens of per cent, and clear-sky fluxes do not move at all."""
    kw = dict(backend=oracle_lib.backend, directory_name=DATA_DIR)
    _, _, flux_a, _ = ifs.run_ifs_driver(NAMELIST, MERIDIAN, None, bitidentity=True, **kw)
    _, _, flux_b, _ = ifs.run_ifs_driver(NAMELIST, MERIDIAN, None, **kw)
    assert np.abs(flux_a.sw_up_clear - flux_b.sw_up_clear).max() < 1e-9
    assert np.abs(flux_a.lw_up_clear - flux_b.lw_up_clear).max() < 1e-9
    d = np.abs(flux_a.sw_up - flux_b.sw_up).max()
    assert 0.1 < d < 80.0


def test_radiation_scheme_leaves_other_columns_alone(oracle_lib):
    yr = ifs.TRADIATION()
    yr.rad_config.read_into(NAMELIST)
    ifs.setup_radiation_scheme(yr, file_name=NAMELIST, directory_name=DATA_DIR, backend=oracle_lib.backend)
    c = yr.rad_config
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(c)
    from ecrad_amd.types import IMassMixingRatio
    gas.set_units(IMassMixingRatio)
    pf = 0.5 * (th.pressure_hl[:-1] + th.pressure_hl[1:])
    tf = 0.5 * (th.temperature_hl[:-1] + th.temperature_hl[1:])
    z = np.zeros((nlev, ncol))
    g = lambda i: gas.mixing_ratio[i - 1]
    from ecrad_amd.tables import IH2O, ICO2, ICH4, IN2O, INO2, ICFC11, ICFC12, IHCFC22, ICCl4, IO3
    args = dict(PSOLAR_IRRADIANCE=sl.solar_irradiance, PMU0=sl.cos_sza, PTEMPERATURE_SKIN=sl.skin_temperature,
                PALBEDO_DIF=sl.sw_albedo, PALBEDO_DIR=sl.sw_albedo_direct, PSPECTRALEMISS=sl.lw_emissivity,
                PCCN_LAND=np.full(ncol, 900.0), PCCN_SEA=np.full(ncol, 50.0), PGELAM=np.zeros(ncol), PGEMU=np.zeros(ncol),
                PLAND_SEA_MASK=np.zeros(ncol), PPRESSURE=pf, PTEMPERATURE=tf, PPRESSURE_H=th.pressure_hl,
                PTEMPERATURE_H=th.temperature_hl, PQ=g(IH2O), PCO2=g(ICO2), PCH4=g(ICH4), PN2O=g(IN2O), PNO2=g(INO2),
                PCFC11=g(ICFC11), PCFC12=g(ICFC12), PHCFC22=g(IHCFC22), PCCL4=g(ICCl4), PO3=g(IO3),
                PCLOUD_FRAC=cloud.fraction, PQ_LIQUID=cloud.mixing_ratio[0], PQ_ICE=cloud.mixing_ratio[1], PQ_RAIN=z, PQ_SNOW=z,
                PAEROSOL_OLD=np.zeros((nlev, 6, ncol)), PAEROSOL=aer.mixing_ratio)
    full = ifs.radiation_scheme(yr, 1, ncol, ncol, nlev, 12, **args)
    out = ifs.allocate_ifs_outputs(yr, ncol, nlev)
    for a in out.values():
        a[...] = -777.0
    ifs.radiation_scheme(yr, 9, 20, ncol, nlev, 12, out=out, **args)
    for name in ifs.IFS_OUTPUTS_PROFILE + ifs.IFS_OUTPUTS_SURFACE:
        a, b = out[name], full[name]
        if name == "PLWDERIVATIVE" and not yr.yrerad.LAPPROXLWUPDATE:
            continue
        assert np.array_equal(a[..., 8:20], b[..., 8:20]), name
        assert np.all(a[..., :8] == -777.0) and np.all(a[..., 20:] == -777.0), name


@pytest.mark.gpu
@pytest.mark.parametrize("blocked", [False, True])
def test_ifs_driver_on_the_gpu(oracle_lib, blocked):
    """The same drivers with the HIP library as the operator: against the oracle run through the same host code."""
    kw = dict(bitidentity=True, directory_name=DATA_DIR, blocked=blocked)
    _, _, flux_hip, diag_hip = ifs.run_ifs_driver(NAMELIST, MERIDIAN, None, backend="hip", **kw)
    _, _, flux_ora, diag_ora = ifs.run_ifs_driver(NAMELIST, MERIDIAN, None, backend=oracle_lib.backend, **kw)
    for n in NET + ("lw_derivatives",):
        assert rel_err(flux_hip.arrays[n], flux_ora.arrays[n]) < 1.0e-8, n
    for n in diag_ora:
        if np.abs(diag_ora[n]).max() > 0:
            assert rel_err(diag_hip[n], diag_ora[n]) < 1.0e-8, n
    flux = _net_of_ordinary_driver("hip")
    _check_against_ordinary(flux_hip, flux, 1.0e-10)


@pytest.mark.gpu
def test_ifs_parametrised_clouds_on_the_gpu(oracle_lib):
    """Effective radii and overlap from the IFS parametrisations (no bit-identity arguments), McICA + RRTMG-free ecCKD."""
    kw = dict(directory_name=DATA_DIR)
    _, _, flux_hip, _ = ifs.run_ifs_driver(NAMELIST, MERIDIAN, None, backend="hip", blocked=True, **kw)
    _, _, flux_ora, _ = ifs.run_ifs_driver(NAMELIST, MERIDIAN, None, backend=oracle_lib.backend, **kw)
    for n in NET:
        assert rel_err(flux_hip.arrays[n], flux_ora.arrays[n]) < 1.0e-8, n
