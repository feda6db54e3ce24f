"""Every run of the reference's own test suite (test/ifs/Makefile: `make test` = test_orig + test_ecckd, plus the optional
targets), with the namelists as they are -- do_save_spectral_flux = true in both, which McICA ignores
(radiation_config.F90:1331-1334) -- and the change_namelist edits of each target: the HIP path through the C-ABI against the
oracle on the reference's 32-column slice.  (Where the reference holds a golden output of the run,
tests/test_reference_goldens.py compares with that as well.)"""
import pytest

from ecrad_amd.config import IGasModelECCKD, IGasModelIFSRRTMG
from helpers import compare_flux, make_config, make_config_rrtmg, run_case

pytestmark = pytest.mark.gpu

NML = dict(do_save_spectral_flux=True, do_lw_aerosol_scattering=False)      # what both namelists say
MIXED = dict(NML, do_save_spectral_flux=False, use_general_cloud_optics=True)     # configCY49R1_mixed.nam

# target -> (namelist family, solver, edits); Makefile line numbers of test/ifs/Makefile
TARGETS = {
    "test_default": ("rrtmg", "McICA", {}),                                                         # :34
    "test_noaer": ("rrtmg", "McICA", dict(use_aerosols=False)),                                       # :50
    "test_expexp": ("rrtmg", "McICA", dict(i_overlap_scheme=2)),                                      # :56
    "test_tripleclouds": ("rrtmg", "Tripleclouds", {}),                                               # :62
    "test_lwscat": ("rrtmg", "McICA", dict(do_lw_cloud_scattering=True)),                             # :68
    "test_spartacus": ("rrtmg", "SPARTACUS", dict(do_3d_effects=True, do_sw_delta_scaling_with_gases=False)),      # :74
    "test_spartacus_maxentr": ("rrtmg", "SPARTACUS", dict(do_3d_effects=True, i_3d_sw_entrapment=4, do_sw_delta_scaling_with_gases=False)),   # :82
    "test_cloudless": ("rrtmg", "Cloudless", dict(use_aerosols=False)),                               # :91
    "test_vec": ("rrtmg", "McICA", dict(use_vectorizable_generator=True)),                            # :98
    "test_ifsdriver": ("rrtmg", "Tripleclouds", {}),                                                  # :37 (solver settings; the IFS-style caller: tests/test_ifs_scheme.py)
    "test_ecckd_mcica": ("ecckd", "McICA", {}),                                                       # :106
    "test_ecckd_tc": ("ecckd", "Tripleclouds", {}),                                                   # :111
    "test_ecckd_noaer": ("ecckd", "Tripleclouds", dict(use_aerosols=False)),                          # :124, :165
    "test_ecckd_spartacus": ("ecckd", "SPARTACUS", dict(do_3d_effects=True)),                         # :158
    # the namelists of earlier IFS cycles that test/ifs holds next to CY49R1 (Makefile:10-14): aerosol optics from the
    # band-wise file (use_general_aerosol_optics = false), no spectral surface fluxes; CY47R1 also Exp-Exp overlap and
    # another aerosol type map
    "configCY47R3": ("rrtmg", "McICA", dict(use_general_aerosol_optics=False, do_surface_sw_spectral_flux=False)),
    "configCY47R1": ("rrtmg", "McICA", dict(use_general_aerosol_optics=False, do_surface_sw_spectral_flux=False, i_overlap_scheme=2,
                                            i_aerosol_type_map=[-1, -2, -3, 1, 2, 3, -4, 10, 11, 11, -5, 14])),
    # test_mixed_gas (:114-122): configCY49R1_mixed.nam and its three edits
    "test_mixed_gas_ecckd_ecckd": ("ecckd", "Tripleclouds", dict(do_save_spectral_flux=False)),
    "test_mixed_gas_sw_ecckd_lw_rrtmg": ("rrtmg", "Tripleclouds", dict(MIXED, i_gas_model_sw=IGasModelECCKD, do_cloud_aerosol_per_sw_g_point=True)),
    "test_mixed_gas_sw_rrtmg_lw_ecckd": ("rrtmg", "Tripleclouds", dict(MIXED, i_gas_model_lw=IGasModelECCKD, do_cloud_aerosol_per_lw_g_point=True)),
    "test_mixed_gas_rrtmg_rrtmg": ("rrtmg", "Tripleclouds", dict(MIXED)),
}


def _config(name):
    family, solver, edits = TARGETS[name]
    kw = dict(NML, **edits)
    return make_config_rrtmg(solver, **kw) if family == "rrtmg" else make_config(solver, **kw)


@pytest.mark.parametrize("target", sorted(TARGETS))
def test_reference_test_target_runs_and_matches_the_oracle(target, oracle_lib):
    c1, c2 = _config(target), _config(target)
    rrtmg = IGasModelIFSRRTMG in (c1.i_gas_model_sw, c1.i_gas_model_lw)
    if rrtmg and not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    f_hip, _, rad = run_case(c1, "hip")
    rad.close()
    f_ora, _, _ = run_case(c2, oracle_lib.make_rrtmg_backend(c2) if rrtmg else oracle_lib.backend)
    worst = compare_flux(f_hip, f_ora, 1.0)
    # 1e-8 on the broadband profiles; the bar itself on the per-g-point / per-band values (RRTMG has almost purely Rayleigh
    # g-points whose two-stream coefficients amplify the last bits of the optical depths, see tests/test_hip_rrtmg.py)
    bad = {k: v for k, v in worst.items() if v > (1.0e-6 if k.endswith(("_g", "_band", "_canopy")) else 1.0e-8)}
    assert not bad, bad
