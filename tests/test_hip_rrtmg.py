"""GPU parity for gas_model_name = "RRTMG-IFS" (BASELINE configs[2]): the HIP path
through the C-ABI against
  * the reference's OWN RRTMG routines (oracle/_ref/libecrad_refrrtm.so, compiled from /root/reference unmodified)
    for the gas-optics stage arrays, and the committed golden vectors made with them;
  * the oracle (those routines + the C restatement of everything downstream) for the fluxes of every solver;
  * (the reference's golden output files of its RRTMG configurations: tests/test_reference_goldens.py).
Tolerance as in test_hip_parity.py: 1e-8 relative on fluxes (bar: 1e-6)."""
import ctypes as C
import os

import numpy as np
import pytest

from ecrad_amd import abi
from ecrad_amd.driver import flux_to_output_dict
from ecrad_amd.interface import Radiation, build_inputs_struct
from ecrad_amd.ncfile import NcFile
from helpers import GOLDEN_DIR, compare_flux, load_meridian, make_config_rrtmg, rel_err, run_case

pytestmark = pytest.mark.gpu
TOL = 1.0e-8
TOL_SPECTRAL = 1.0e-6

CASES = {
    "mcica_default": dict(sw_solver="McICA", do_lw_aerosol_scattering=False),          # test/ifs/configCY49R1.nam
    "mcica_noaer": dict(sw_solver="McICA", use_aerosols=False, do_lw_aerosol_scattering=False),
    "mcica_lw_aerosol_scat": dict(sw_solver="McICA"),
    "mcica_general_cloud_optics": dict(sw_solver="McICA", use_general_cloud_optics=True, do_lw_aerosol_scattering=False),
    "mcica_expexp": dict(sw_solver="McICA", i_overlap_scheme=2, do_lw_aerosol_scattering=False),
    "tripleclouds": dict(sw_solver="Tripleclouds", do_lw_aerosol_scattering=False),
    "tripleclouds_lw_aerosol_scat": dict(sw_solver="Tripleclouds"),
    "tripleclouds_spectral": dict(sw_solver="Tripleclouds", do_save_spectral_flux=True, do_lw_aerosol_scattering=False),
    "homogeneous": dict(sw_solver="Homogeneous", do_lw_aerosol_scattering=False),
    "cloudless": dict(sw_solver="Cloudless", do_lw_aerosol_scattering=False),
    "cloudless_noaer": dict(sw_solver="Cloudless", use_aerosols=False, do_lw_aerosol_scattering=False),
    "cloudless_lw_aerosol_scat": dict(sw_solver="Cloudless"),
    "no_lw_cloud_scattering": dict(sw_solver="McICA", do_lw_cloud_scattering=False, do_lw_aerosol_scattering=False),
    "delta_scaling_with_gases": dict(sw_solver="Tripleclouds", do_sw_delta_scaling_with_gases=True, do_lw_aerosol_scattering=False),
    "mcica_delta_scaling_with_gases": dict(sw_solver="McICA", do_sw_delta_scaling_with_gases=True, do_lw_aerosol_scattering=False),
    "homogeneous_lw_aerosol_scat": dict(sw_solver="Homogeneous"),
    # the other band cloud-optics schemes of radiation_cloud_optics.F90; model codes of radiation_config.F90:109-133
    "slingo_liquid": dict(sw_solver="McICA", i_liq_model=2, do_lw_aerosol_scattering=False),
    "baran_ice": dict(sw_solver="McICA", i_ice_model=2, do_lw_aerosol_scattering=False),
    "baran2016_ice": dict(sw_solver="Tripleclouds", i_ice_model=3, do_lw_aerosol_scattering=False),
    "baran2017_ice": dict(sw_solver="McICA", i_ice_model=4, do_lw_aerosol_scattering=False),
    "yi_ice": dict(sw_solver="Tripleclouds", i_ice_model=5, do_lw_aerosol_scattering=False),
    "slingo_yi_no_lw_scattering": dict(sw_solver="McICA", i_liq_model=2, i_ice_model=5, do_lw_cloud_scattering=False, do_lw_aerosol_scattering=False),
    # SPARTACUS on the 140/112-point spectra (five / two launches of the layer and sweep kernels per spectrum): the
    # reference's own test_spartacus and test_spartacus_maxentr targets (test/ifs/Makefile:76-90) -- the 3-D effects on,
    # do_sw_delta_scaling_with_gases off as in configCY49R1.nam -- and variants.  entrapment codes: radiation_config.F90:98-104
    "spartacus": dict(sw_solver="SPARTACUS", do_3d_effects=True, do_lw_derivatives=True, do_lw_aerosol_scattering=False),
    "spartacus_maxentr": dict(sw_solver="SPARTACUS", do_3d_effects=True, i_3d_sw_entrapment=4, do_lw_derivatives=True, do_lw_aerosol_scattering=False),
    "spartacus_lw_aerosol_scat": dict(sw_solver="SPARTACUS", do_3d_effects=True, do_lw_derivatives=True),
    # (a low threshold: the g-point from which the 3-D treatment is switched off lies in an EARLIER launch for most chunks)
    "spartacus_tight_caps": dict(sw_solver="SPARTACUS", do_3d_effects=True, max_gas_od_3d=0.05, max_3d_transfer_rate=1.0, do_lw_aerosol_scattering=False),
    "spartacus_expm_everywhere_noaer": dict(sw_solver="SPARTACUS", do_3d_effects=True, use_expm_everywhere=True, use_aerosols=False, do_lw_aerosol_scattering=False),
    "spartacus_spectral": dict(sw_solver="SPARTACUS", do_3d_effects=True, do_save_spectral_flux=True, do_lw_aerosol_scattering=False),   # test_spartacus as its namelist has it
    "spartacus_no_3d": dict(sw_solver="SPARTACUS", do_3d_effects=False, do_lw_aerosol_scattering=False),
}


@pytest.fixture(scope="module")
def ref(oracle_lib):
    if not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    return oracle_lib


@pytest.mark.parametrize("case", sorted(CASES))
def test_hip_matches_oracle_with_rrtmg(case, ref):
    kw = dict(CASES[case])
    sw = kw.pop("sw_solver")
    c1, c2 = make_config_rrtmg(sw, **kw), make_config_rrtmg(sw, **kw)
    f_hip, _, rad = run_case(c1, "hip")
    f_ora, _, _ = run_case(c2, ref.make_rrtmg_backend(c2))
    worst = compare_flux(f_hip, f_ora, 1.0)
    rad.close()
    # Broadband profiles: 1e-8.  Per-g-point / per-band surface and TOA values: the bar itself (1e-6).  Several RRTMG
    # shortwave g-points are almost purely Rayleigh (ssa = 1 - O(1e-9) without aerosols), where the two-stream
    # coefficients depend on (1 - ssa) and amplify the last-bit differences of the optical depths to ~1e-7.
    bad = {k: v for k, v in worst.items()
           if v > (TOL_SPECTRAL if k.endswith(("_g", "_band", "_canopy")) else TOL)}
    assert not bad, bad
    print(case, "max rel diff", max(worst.values()))


def _hip_optics(config):
    rad = Radiation(config, backend="hip")
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    rad.set_gas_units(gas)
    th.calc_saturation_wrt_liquid()
    cin, keep = build_inputs_struct(config, ncol, nlev, sl, th, gas, cloud, aer)
    # (cloud.fraction is aliased, not copied, by build_inputs_struct: the inputs must outlive the calls)
    return rad, ncol, nlev, cin, (keep, sl, th, gas, cloud, aer)


def test_gas_optics_stage_matches_the_reference_routines(ref):
    """od_lw, planck_hl, lw_emission, od_sw, ssa_sw, incoming_sw without clouds or aerosols = what gas_optics of
    radiation_ifs_rrtm.F90 returns; compared with the reference's routines on all 32 columns."""
    config = make_config_rrtmg("Cloudless", use_aerosols=False, do_lw_aerosol_scattering=False)
    rad, ncol, nlev, cin, keep = _hip_optics(config)
    want = ref.rrtmg_gas_stage(config, ncol, nlev, cin)
    out = abi.Optics()
    got = {k: np.zeros(v) for k, v in ref.optics_shapes(config, nlev, ncol).items()}
    for k, a in got.items():
        setattr(out, k, abi.dptr(a))
    st = rad.lib.ecrad_hip_optics(rad.handle, ncol, nlev, 1, ncol, C.byref(cin), C.byref(out))
    assert st == 0, rad.lib.ecrad_hip_last_error(rad.handle)
    day = np.ctypeslib.as_array(cin.cos_sza, shape=(ncol,)) > 0
    for k in ("od_lw", "planck_hl"):
        assert rel_err(got[k], want[k], floor_frac=1e-12) < 1e-9, k
    assert rel_err(got["lw_emission"], want["lw_emission"] * (1.0 - got["lw_albedo"]), floor_frac=1e-12) < 1e-9
    for k in ("od_sw", "ssa_sw", "incoming_sw"):
        assert rel_err(got[k][day], want[k][day], floor_frac=1e-12) < 1e-9, k
    assert np.all(got["incoming_sw"][~day] == 0.0)
    # and the committed golden vectors (8 of the columns, raw outputs of the reference routines)
    g = np.load(os.path.join(GOLDEN_DIR, "rrtmg_gas_optics.npz"))
    cols = g["columns"]
    assert rel_err(got["od_lw"][cols], np.maximum(g["od_lw"][:, ::-1, :], 1e-15), floor_frac=1e-12) < 1e-9
    gday = g["cos_sza"] > 0
    assert rel_err(got["od_sw"][cols][gday], np.transpose(g["od_sw"], (2, 1, 0))[:, ::-1, :][gday], floor_frac=1e-12) < 1e-9
    rad.close()


def test_band_cloud_optics_and_aerosols_match_oracle(ref):
    """SOCRATES/Fu band cloud optics (row a9) and the per-band aerosol merge, stage by stage."""
    config = make_config_rrtmg("McICA", do_lw_aerosol_scattering=False)
    rad, ncol, nlev, cin, keep = _hip_optics(config)
    out = abi.Optics()
    got = {k: np.zeros(v) for k, v in ref.optics_shapes(config, nlev, ncol).items()}
    for k, a in got.items():
        setattr(out, k, abi.dptr(a))
    st = rad.lib.ecrad_hip_optics(rad.handle, ncol, nlev, 1, ncol, C.byref(cin), C.byref(out))
    assert st == 0, rad.lib.ecrad_hip_last_error(rad.handle)
    stage = ref.rrtmg_gas_stage(config, ncol, nlev, cin)
    st_struct = abi.Optics()
    for k, a in stage.items():
        setattr(st_struct, k, abi.dptr(a))
    ref.lib().ecrad_oracle_set_gas_stage(C.byref(st_struct))
    try:
        want = ref.optics(config, rad.cconfig, ncol, nlev, 1, ncol, cin)
    finally:
        ref.lib().ecrad_oracle_set_gas_stage(None)
    day = np.ctypeslib.as_array(cin.cos_sza, shape=(ncol,)) > 0
    for k in got:
        if k in ("ssa_lw", "g_lw"):
            continue
        a, b = (got[k][day], want[k][day]) if k in ("od_sw", "ssa_sw", "g_sw", "incoming_sw") else (got[k], want[k])
        assert rel_err(a, b, floor_frac=1e-9) < 1e-9, k
    rad.close()


# ---- the edge cases of test_hip_parity.py, with the RRTMG gas optics in front of the solvers ---------------------
def test_rrtmg_column_subrange(ref):
    """istartcol..iendcol inside the arrays: columns outside keep their values (the gas-optics pass, its work arrays
    and the stage arrays are indexed by the local column)."""
    from test_hip_parity import TOL as TOL_E  # noqa: F401  (same tolerance)
    c1, c2 = make_config_rrtmg("McICA", do_lw_aerosol_scattering=False), make_config_rrtmg("McICA", do_lw_aerosol_scattering=False)
    f_hip, _, rad = run_case(c1, "hip", columns=(6, 21))
    for name, a in f_hip.arrays.items():
        if name.startswith("cloud_cover"):
            assert np.all(a[:5] == -1.0) and np.all(a[21:] == -1.0)
        elif name in abi.FLUX_PROFILE_FIELDS:
            assert np.all(a[:, :5] == 0.0) and np.all(a[:, 21:] == 0.0), name
        else:
            assert np.all(a[:5] == 0.0) and np.all(a[21:] == 0.0), name
    f_ora, _, _ = run_case(c2, ref.make_rrtmg_backend(c2), columns=(6, 21))
    worst = compare_flux(f_hip, f_ora, 1.0, cols=(6, 21))
    bad = {k: v for k, v in worst.items() if v > (TOL_SPECTRAL if k.endswith(("_g", "_band", "_canopy")) else TOL)}
    assert not bad, bad
    rad.close()


@pytest.mark.parametrize("solver", ["McICA", "Tripleclouds"])
def test_rrtmg_surface_first_level_order(solver, ref):
    """radiation_reverse with RRTMG: the setcoef pass walks the levels from the surface whatever the caller's order."""
    from test_hip_parity import _reverse_levels
    c1, c2 = make_config_rrtmg(solver, do_lw_aerosol_scattering=False), make_config_rrtmg(solver, do_lw_aerosol_scattering=False)
    f_rev, _, rad = run_case(c1, "hip", inputs=_reverse_levels(load_meridian(c1)))
    rad.close()
    f_ora, _, _ = run_case(c2, ref.make_rrtmg_backend(c2))
    for name, a in f_rev.arrays.items():
        b = f_ora.arrays[name]
        if name in abi.FLUX_PROFILE_FIELDS:
            a = a[::-1, :]
        assert rel_err(a, b) <= (TOL_SPECTRAL if name.endswith(("_g", "_band", "_canopy")) else TOL), name


@pytest.mark.parametrize("solver", ["McICA", "Tripleclouds"])
def test_rrtmg_many_columns_bitwise(solver):
    """Size-independent property: 300 copies of the 32 meridian columns (9 600 columns: 150 column tiles per level in
    the gas-optics pass, several column groups per persistent solver block, 9/7 chunk launches) give 300
    bit-identical copies of the 32-column result."""
    from test_hip_parity import _replicate
    config = make_config_rrtmg(solver, do_lw_aerosol_scattering=False)
    f32, _, rad = run_case(config, "hip")
    rad.close()
    times = 300
    config2 = make_config_rrtmg(solver, do_lw_aerosol_scattering=False)
    f_big, _, rad2 = run_case(config2, "hip", inputs=_replicate(load_meridian(config2), times))
    rad2.close()
    for name, a in f_big.arrays.items():
        b = f32.arrays[name]
        if a.ndim == 1:
            want = np.concatenate([b] * times)
        elif a.shape[-1] == 32 * times:
            want = np.concatenate([b] * times, axis=-1)
        else:
            want = np.concatenate([b] * times, axis=0)
        assert np.array_equal(a, want), name


def test_rrtmg_device_memory_mode_matches_host_memory_mode():
    """ECRAD_MEM_DEVICE (what bench.py times) gives the same bits as ECRAD_MEM_HOST, incl. a column sub-range."""
    import torch
    from ecrad_amd.device import DeviceCase
    from ecrad_amd.types import Flux
    config = make_config_rrtmg("McICA", do_lw_aerosol_scattering=False)
    f_host, _, rad = run_case(config, "hip", columns=(3, 30))
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    rad.set_gas_units(gas)
    th.calc_saturation_wrt_liquid()
    flux = Flux.allocate(config, ncol, nlev)
    case = DeviceCase(config, ncol, nlev, sl, th, gas, cloud, aer, flux)
    st = rad.lib.ecrad_hip_radiation(rad.handle, ncol, nlev, 3, 30, C.byref(case.inputs), C.byref(case.flux))
    assert st == 0, rad.lib.ecrad_hip_last_error(rad.handle).decode()
    rad.lib.ecrad_hip_synchronize(rad.handle)
    torch.cuda.synchronize()
    case.flux_to_host(flux)
    for name, a in flux.arrays.items():
        assert np.array_equal(a, f_host.arrays[name]), name
    rad.close()


@pytest.mark.parametrize("nkeep", [120, 100, 90])
def test_rrtmg_other_level_counts(nkeep, ref):
    """Level counts other than 137 (the lowest nkeep levels).  The top must stay above 95.6 hPa: in a column that lies
    wholly in RRTMG's "lower atmosphere" the reference never assigns the solar source term of the bands that look
    for it in the upper atmosphere (srtm_taumol16.F90 etc.; ZSFLXZEN of srtm_gas_optical_depth.F90:109 is an
    uninitialised local), so there is nothing defined to compare with; the HIP path returns zero there
    (test_rrtmg_column_below_the_tropopause_is_defined)."""
    from test_hip_parity import _bottom_levels
    c1, c2 = make_config_rrtmg("McICA", do_lw_aerosol_scattering=False), make_config_rrtmg("McICA", do_lw_aerosol_scattering=False)
    f_hip, _, rad = run_case(c1, "hip", inputs=_bottom_levels(load_meridian(c1), nkeep))
    rad.close()
    f_ora, _, _ = run_case(c2, ref.make_rrtmg_backend(c2), inputs=_bottom_levels(load_meridian(c2), nkeep))
    worst = compare_flux(f_hip, f_ora, 1.0)
    bad = {k: v for k, v in worst.items() if v > (TOL_SPECTRAL if k.endswith(("_g", "_band", "_canopy")) else TOL)}
    assert not bad, bad


def test_rrtmg_column_below_the_tropopause_is_defined():
    """33 lowest levels only: every layer is "lower atmosphere".  The reference's result is undefined there (see
    above); the HIP path gives finite fluxes, with no incoming flux in the bands whose source term is never assigned."""
    from test_hip_parity import _bottom_levels
    c1 = make_config_rrtmg("McICA", do_lw_aerosol_scattering=False)
    f_hip, _, rad = run_case(c1, "hip", inputs=_bottom_levels(load_meridian(c1), 33))
    rad.close()
    for name, a in f_hip.arrays.items():
        assert np.all(np.isfinite(a)), name
    day = f_hip.arrays["sw_dn"][0] > 0
    assert day.any()
    band = f_hip.arrays["sw_dn_surf_band"][day]
    assert np.all(band[:, [0, 1, 11, 12, 13]] == 0.0) and np.all(band[:, 2:11] > 0.0)
