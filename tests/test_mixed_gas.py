"""Different gas-optics models in the two spectra (sw_gas_model_name / lw_gas_model_name; the reference's test_mixed_gas
target, test/ifs/Makefile:112-120 with configCY49R1_mixed.nam: general cloud and aerosol optics in both spectra, per band
where RRTMG is, per g-point where ecCKD is).

What makes the case more than a combination of two tested halves is the units of gas%mixing_ratio: with RRTMG in either
spectrum set_gas_units makes them MASS mixing ratios (radiation_interface.F90:177-181) and the ecCKD model gets them with
the concentration_scaling of gas%get_scaling (radiation_ecckd_interface.F90:249-255, radiation_gas.F90:471-486,
radiation_ecckd.F90:518-625).

Pin: the reference holds no golden output of a mixed run, but each spectrum of a mixed configuration must reproduce the
same spectrum of the unmixed configuration with the same options -- whose outputs ARE pinned by the reference's goldens
(tests/test_reference_goldens.py): bitwise for the RRTMG spectrum, to the rounding of the unit conversion for ecCKD."""
import numpy as np
import pytest

from ecrad_amd.config import IGasModelECCKD
from helpers import compare_flux, make_config, make_config_rrtmg, rel_err, run_case

BASE = dict(use_general_cloud_optics=True, do_lw_aerosol_scattering=False, do_lw_derivatives=True)
# the shortwave / longwave options configCY49R1.nam has and configCY49R1_ecckd.nam has not, so that the unmixed ecCKD
# configuration below has the same options as the ecCKD spectrum of the mixed one
ECCKD_LIKE = dict(do_nearest_spectral_lw_emiss=True, do_surface_sw_spectral_flux=True, do_weighted_surface_mapping=False,
                  do_lw_aerosol_scattering=False, do_lw_derivatives=True)
SW_PROFILES = ("sw_up", "sw_dn", "sw_dn_direct", "sw_up_clear", "sw_dn_clear", "sw_dn_direct_clear")
LW_PROFILES = ("lw_up", "lw_dn", "lw_up_clear", "lw_dn_clear", "lw_derivatives")


def mixed_config(solver, ecckd_spectrum, **kw):
    """configCY49R1.nam with general cloud optics and the ecCKD model in one spectrum"""
    if ecckd_spectrum == "sw":
        return make_config_rrtmg(solver, i_gas_model_sw=IGasModelECCKD, do_cloud_aerosol_per_sw_g_point=True, **{**BASE, **kw})
    return make_config_rrtmg(solver, i_gas_model_lw=IGasModelECCKD, do_cloud_aerosol_per_lw_g_point=True, **{**BASE, **kw})


@pytest.mark.parametrize("solver", ["Tripleclouds", "McICA", "Homogeneous"])
@pytest.mark.parametrize("ecckd_spectrum", ["sw", "lw"])
def test_oracle_mixed_spectra_reproduce_the_unmixed_ones(solver, ecckd_spectrum, oracle_lib):
    if not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    cm = mixed_config(solver, ecckd_spectrum)
    f_mix, _, _ = run_case(cm, oracle_lib.make_rrtmg_backend(cm))
    cr = make_config_rrtmg(solver, **BASE)
    f_rrtmg, _, _ = run_case(cr, oracle_lib.make_rrtmg_backend(cr))
    f_ecckd, _, _ = run_case(make_config(solver, **ECCKD_LIKE), oracle_lib.backend)
    ck_names, rr_names = (SW_PROFILES, LW_PROFILES) if ecckd_spectrum == "sw" else (LW_PROFILES, SW_PROFILES)
    for k in rr_names:      # the RRTMG spectrum does not know about the other one
        assert np.array_equal(f_mix.arrays[k], f_rrtmg.arrays[k]), k
    for k in ck_names:      # the ecCKD spectrum sees its mixing ratios through mass mixing ratio and back
        assert rel_err(f_mix.arrays[k], f_ecckd.arrays[k]) < 1.0e-12, k
    assert (cm.n_g_sw, cm.n_g_lw) == ((32, 140) if ecckd_spectrum == "sw" else (112, 32))


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["Tripleclouds", "McICA", "Homogeneous", "Cloudless", "SPARTACUS"])
@pytest.mark.parametrize("ecckd_spectrum", ["sw", "lw"])
def test_hip_matches_oracle_with_mixed_gas_models(solver, ecckd_spectrum, oracle_lib):
    if not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    kw = dict(do_3d_effects=True) if solver == "SPARTACUS" else {}
    c1, c2 = mixed_config(solver, ecckd_spectrum, **kw), mixed_config(solver, ecckd_spectrum, **kw)
    f_hip, _, rad = run_case(c1, "hip")
    rad.close()
    f_ora, _, _ = run_case(c2, oracle_lib.make_rrtmg_backend(c2))
    worst = compare_flux(f_hip, f_ora, 1.0)
    # (tolerances as in test_hip_rrtmg.py: 1e-8 on the profiles, the bar itself on the almost purely Rayleigh g-points)
    bad = {k: v for k, v in worst.items() if v > (1.0e-6 if k.endswith(("_g", "_band", "_canopy")) else 1.0e-8)}
    assert not bad, bad


@pytest.mark.gpu
def test_hip_mixed_shortwave_equals_hip_ecckd_shortwave():
    """The same pin as on the oracle, on the device: the concentration scaling only changes the last bits."""
    f_mix, _, rad = run_case(mixed_config("Tripleclouds", "sw"), "hip")
    rad.close()
    f_ck, _, rad = run_case(make_config("Tripleclouds", **ECCKD_LIKE), "hip")
    rad.close()
    for k in SW_PROFILES:
        assert rel_err(f_mix.arrays[k], f_ck.arrays[k]) < 1.0e-12, k
