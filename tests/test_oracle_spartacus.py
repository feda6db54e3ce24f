"""The SPARTACUS restatement of the oracle (oracle/oracle_spartacus.c).

The reference holds no golden output of a SPARTACUS run, and its solver modules cannot be compiled here (config_type
needs netCDF), so the solver BODY is pinned through what can be pinned:
  * its matrix algebra against the reference's own radiation_matrix.F90 (tests/test_oracle_matrix.py, 1e-12);
  * the identity the reference documents (radiation_config.F90:73 "IEntrapmentZero: No entrapment, as Tripleclouds"):
    with 3-D effects off the shortwave solver IS Tripleclouds -- the same regions, overlap matrices, Meador-Weaver
    layers and matrix adding method -- and Tripleclouds is pinned end to end by the reference's golden file.  Agreement:
    1e-13 with max_cloud_od lifted, i.e. sections 3.2b, 3.3b, 4.1 (matrix adding), 4.2 (edge-only / zero entrapment
    branches) and 5 of the restatement are exercised against golden-pinned code to rounding;
  * the longwave solver without 3-D effects against Tripleclouds to 1e-4 (the two reference solvers differ there by
    construction: calc_reflectance_transmittance_lw vs calc_no_scattering_transmittance_lw in clear layers);
  * conservation and sign properties of the 3-D terms, every entrapment option, single precision vs double."""
import numpy as np
import pytest

from ecrad_amd.config import (IEntrapmentEdgeOnly, IEntrapmentExplicit, IEntrapmentExplicitNonFractal,
                              IEntrapmentMaximum, IEntrapmentZero)
from helpers import make_config, rel_err, run_case


@pytest.fixture(scope="module")
def tripleclouds(oracle_lib):
    f, _, _ = run_case(make_config("Tripleclouds", do_lw_derivatives=True), oracle_lib.backend)
    return f


def spartacus(oracle_lib, **kw):
    f, _, _ = run_case(make_config("SPARTACUS", do_lw_derivatives=True, **kw), oracle_lib.backend)
    for name, a in f.arrays.items():
        assert np.all(np.isfinite(a)), name
    return f


@pytest.mark.parametrize("entrapment", [IEntrapmentZero, IEntrapmentEdgeOnly, IEntrapmentExplicit, IEntrapmentExplicitNonFractal])
def test_shortwave_without_3d_effects_is_tripleclouds(oracle_lib, tripleclouds, entrapment):
    sp = spartacus(oracle_lib, do_3d_effects=False, i_3d_sw_entrapment=entrapment, max_cloud_od=1.0e30)
    for name in ("sw_up", "sw_dn", "sw_dn_direct", "sw_up_clear", "sw_dn_clear", "sw_dn_direct_clear", "cloud_cover_sw",
                 "sw_up_toa_g", "sw_up_toa_clear_g", "sw_dn_diffuse_surf_g", "sw_dn_direct_surf_g",
                 "sw_dn_diffuse_surf_clear_g", "sw_dn_direct_surf_clear_g"):
        assert rel_err(sp.arrays[name], tripleclouds.arrays[name]) < 1.0e-13, name
    # with the reference's default cap of the in-region optical depth (max_cloud_od = 16) a few thick layers differ
    sp = spartacus(oracle_lib, do_3d_effects=False, i_3d_sw_entrapment=entrapment)
    assert rel_err(sp.arrays["sw_up"], tripleclouds.arrays["sw_up"]) < 1.0e-10


def test_longwave_without_3d_effects_is_close_to_tripleclouds(oracle_lib, tripleclouds):
    sp = spartacus(oracle_lib, do_3d_effects=False, max_cloud_od=1.0e30)
    for name in ("lw_up", "lw_dn", "lw_up_clear", "lw_dn_clear", "lw_dn_surf_g", "lw_up_toa_g"):
        assert rel_err(sp.arrays[name], tripleclouds.arrays[name]) < 1.0e-4, name
    assert np.array_equal(sp.arrays["cloud_cover_lw"], tripleclouds.arrays["cloud_cover_lw"])
    cloudy = sp.arrays["cloud_cover_lw"] > 0.0       # (the reference's Tripleclouds treats cloud-free columns differently)
    assert rel_err(sp.arrays["lw_derivatives"][:, cloudy], tripleclouds.arrays["lw_derivatives"][:, cloudy]) < 1.0e-3


def test_longwave_derivative_in_a_cloud_free_column_is_the_weighted_transmittance(oracle_lib):
    """calc_lw_derivatives_matrix (radiation_lw_derivatives.F90:138-193), known answer: without clouds the derivative of
    the upwelling flux at a half level with respect to the surface value is sum_g w_g prod_l T_gl."""
    import ctypes as C
    from ecrad_amd.interface import Radiation, build_inputs_struct
    from helpers import load_meridian
    config = make_config("SPARTACUS", do_3d_effects=False, do_lw_derivatives=True)
    f, th, rad = run_case(config, oracle_lib.backend)
    inp = load_meridian(config)
    rad.set_gas_units(inp[4]); inp[3].calc_saturation_wrt_liquid()
    cin, keep = build_inputs_struct(config, *inp)
    o = oracle_lib.optics(config, rad.cconfig, 32, 137, 1, 32, cin)
    clear_cols = np.nonzero(f.arrays["cloud_cover_lw"] == 0.0)[0]
    assert len(clear_cols) >= 2
    for col in clear_cols:
        T = np.exp(-1.66 * o["od_lw"][col])
        fus = o["lw_emission"][col] + o["lw_albedo"][col] * f.arrays["lw_dn_surf_g"][col]
        w = fus / fus.sum()
        want = np.array([(w * T[l:].prod(axis=0)).sum() for l in range(138)])
        assert np.abs(f.arrays["lw_derivatives"][:, col] - want).max() < 2.0e-4


@pytest.mark.parametrize("entrapment", [IEntrapmentZero, IEntrapmentEdgeOnly, IEntrapmentExplicit,
                                        IEntrapmentExplicitNonFractal, IEntrapmentMaximum])
def test_3d_effects_conserve_energy_and_keep_clear_sky_untouched(oracle_lib, tripleclouds, entrapment):
    sp = spartacus(oracle_lib, do_3d_effects=True, i_3d_sw_entrapment=entrapment)
    a = sp.arrays
    # clear-sky profiles do not see the clouds at all
    for name in ("sw_up_clear", "sw_dn_clear", "sw_dn_direct_clear"):
        assert rel_err(a[name], tripleclouds.arrays[name]) < 1.0e-13, name
    day = a["sw_dn"][0] > 0.0
    assert day.sum() >= 20
    # fluxes are non-negative, the direct beam only weakens downwards, nothing is created: absorbed = in - out >= 0
    assert a["sw_up"].min() >= 0.0 and a["sw_dn"].min() >= 0.0
    assert np.all(np.diff(a["sw_dn_direct"][:, day], axis=0) <= 1e-9)
    absorbed = (a["sw_dn"][0] - a["sw_up"][0]) - (a["sw_dn"][-1] - a["sw_up"][-1])
    assert np.all(absorbed[day] > 0.0)
    assert np.all(a["sw_up"][0] <= a["sw_dn"][0] + 1e-9)
    # the per-g-point surface and TOA values add up to the broadband ones
    assert rel_err(a["sw_up_toa_g"].sum(axis=1), a["sw_up"][0]) < 1e-12
    assert rel_err(a["sw_dn_diffuse_surf_g"].sum(axis=1) + a["sw_dn_direct_surf_g"].sum(axis=1), a["sw_dn"][-1]) < 1e-12
    assert rel_err(a["lw_dn_surf_g"].sum(axis=1), a["lw_dn"][-1]) < 1e-12
    # 3-D effects change cloudy columns by W m-2, not by orders of magnitude, and leave cloud-free columns alone
    cloud_free = a["cloud_cover_sw"] == 0.0
    assert rel_err(a["sw_up"][:, cloud_free & day], tripleclouds.arrays["sw_up"][:, cloud_free & day]) < 1e-10
    d = np.abs(a["sw_up"][0] - tripleclouds.arrays["sw_up"][0])
    assert 0.01 < d[day & ~cloud_free].max() < 80.0


def test_entrapment_orders_the_reflected_flux(oracle_lib):
    """Hogan et al. (2019): more entrapment = less reflection to space.  Zero >= Edge-only >= Explicit >= Maximum in the
    mean over the sunlit columns of the slice."""
    toa = {}
    for e in (IEntrapmentZero, IEntrapmentEdgeOnly, IEntrapmentExplicit, IEntrapmentMaximum):
        a = spartacus(oracle_lib, do_3d_effects=True, i_3d_sw_entrapment=e).arrays
        toa[e] = a["sw_up"][0][a["sw_dn"][0] > 0].mean()
    assert toa[IEntrapmentZero] > toa[IEntrapmentEdgeOnly] > toa[IEntrapmentExplicit] > toa[IEntrapmentMaximum]


@pytest.mark.parametrize("kw", [dict(use_expm_everywhere=True), dict(do_3d_lw_multilayer_effects=True),
                                dict(clear_to_thick_fraction=0.3, overhang_factor=1.0, overhead_sun_factor=0.06),
                                dict(do_lw_side_emissivity=False), dict(do_lw_cloud_scattering=False, do_lw_aerosol_scattering=False),
                                dict(use_aerosols=False), dict(max_3d_transfer_rate=1.0, max_gas_od_3d=0.5)])
def test_other_options_run_and_stay_physical(oracle_lib, kw):
    a = spartacus(oracle_lib, do_3d_effects=True, **kw).arrays
    assert a["sw_up"].min() >= 0.0 and a["lw_up"].min() > 0.0 and a["lw_dn"].min() >= 0.0
    assert np.all(a["lw_up"][-1] > a["lw_up"][0] * 0.3)


def test_expm_everywhere_agrees_with_meador_weaver_in_clear_layers(oracle_lib):
    """use_expm_everywhere sends the clear layers through the matrix exponential too: the same two-stream equations
    solved by Pade-7 scaling and squaring ("accurate only to single precision", radiation_matrix.F90:800-801)."""
    a = spartacus(oracle_lib, do_3d_effects=True).arrays
    b = spartacus(oracle_lib, do_3d_effects=True, use_expm_everywhere=True).arrays
    for name in ("sw_up", "sw_dn", "lw_up", "lw_dn"):
        assert rel_err(b[name], a[name]) < 5.0e-5, name


@pytest.mark.parametrize("per_band", [False, True], ids=["gpoints", "bands"])
def test_spectral_flux_profiles(oracle_lib, per_band):
    """do_save_spectral_flux (radiation_spartacus_sw.F90:1403-1424, :1472-1493, :1557-1572; _lw.F90:956-967, :1033-1048; on in
    both of the reference's test namelists, hence in its test_spartacus / test_ecckd_spartacus runs): the shortwave profiles
    of the 1-D limit are those of the (golden-pinned) Tripleclouds restatement, and the intervals of every profile add up to
    the broadband one."""
    kw = dict(do_save_spectral_flux=True)
    if per_band:
        kw.update(do_cloud_aerosol_per_sw_g_point=False, do_cloud_aerosol_per_lw_g_point=False)
    tc, _, _ = run_case(make_config("Tripleclouds", **kw), oracle_lib.backend)
    sp1d, _, _ = run_case(make_config("SPARTACUS", do_3d_effects=False, i_3d_sw_entrapment=IEntrapmentZero, max_cloud_od=1.0e30, **kw),
                          oracle_lib.backend)
    for name in ("sw_up_band", "sw_dn_band", "sw_dn_direct_band", "sw_up_clear_band", "sw_dn_clear_band", "sw_dn_direct_clear_band"):
        assert sp1d.arrays[name].shape == tc.arrays[name].shape
        assert rel_err(sp1d.arrays[name], tc.arrays[name]) < 1.0e-12, name
    sp3d, _, _ = run_case(make_config("SPARTACUS", do_3d_effects=True, **kw), oracle_lib.backend)
    a = sp3d.arrays
    for spec, broad in (("sw_up_band", "sw_up"), ("sw_dn_band", "sw_dn"), ("sw_dn_direct_band", "sw_dn_direct"),
                        ("sw_up_clear_band", "sw_up_clear"), ("sw_dn_clear_band", "sw_dn_clear"), ("sw_dn_direct_clear_band", "sw_dn_direct_clear"),
                        ("lw_up_band", "lw_up"), ("lw_dn_band", "lw_dn"), ("lw_up_clear_band", "lw_up_clear"), ("lw_dn_clear_band", "lw_dn_clear")):
        assert rel_err(a[spec].sum(axis=-1), a[broad]) < 1.0e-12, spec      # numpy (half_level, column, interval)
