"""use_spectral_solar_cycle (radiation_config.F90:173-174, :1200-1218; ckd_model_type%read_spectral_solar_cycle,
radiation_ecckd.F90:295-451; calc_incoming_sw :935-965): the solar-cycle amplitude of the spectral solar irradiance on the
g-points of the shortwave ecCKD model, applied with single_level%spectral_solar_cycle_multiplier."""
import numpy as np
import pytest

from helpers import compare_flux, load_meridian, make_config, rel_err, run_case


def _run(backend, multiplier, **kw):
    config = make_config("Tripleclouds", use_spectral_solar_cycle=True, **kw)
    inputs = load_meridian(config)
    inputs[2].spectral_solar_cycle_multiplier = multiplier
    f, _, rad = run_case(config, backend, inputs=inputs)
    return config, f, rad


def test_amplitude_table_properties():
    from ecrad_amd.interface import setup_radiation
    config = make_config("Tripleclouds", use_spectral_solar_cycle=True)
    setup_radiation(config)
    m = config.gas_optics_sw
    a, n = m.norm_amplitude_solar_irradiance, m.norm_solar_irradiance
    assert abs(a.sum()) < 1e-15 and abs(n.sum() - 1.0) < 1e-12          # the total solar irradiance is the host's
    assert np.all(np.abs(a) < 0.01 * n)                                    # a fraction of a per cent per g-point ...
    iuv = np.argmax(a / n)
    assert m.spectral_def.i_band_number[iuv] == m.spectral_def.i_band_number.max()      # ... largest in the ultraviolet band
    # the updated solar spectrum changes the mean irradiances by a few per cent at most and keeps them normalised
    c2 = make_config("Tripleclouds", use_spectral_solar_cycle=True, use_updated_solar_spectrum=True)
    setup_radiation(c2)
    n2 = c2.gas_optics_sw.norm_solar_irradiance
    assert abs(n2.sum() - 1.0) < 1e-12 and 1e-4 < np.abs(n2 / n - 1.0).max() < 0.1


def test_oracle_solar_cycle_moves_energy_between_g_points_only(oracle_lib):
    _, f0, _ = _run(oracle_lib.backend, 0.0)
    _, f1, _ = _run(oracle_lib.backend, 1.0)
    _, fm, _ = _run(oracle_lib.backend, -1.0)
    fref, _, _ = run_case(make_config("Tripleclouds"), oracle_lib.backend)
    assert np.array_equal(f0.arrays["sw_dn"], fref.arrays["sw_dn"])            # multiplier 0 = no solar cycle
    day = f0.arrays["sw_dn"][0] > 0
    assert rel_err(f1.arrays["sw_dn"][0], f0.arrays["sw_dn"][0]) < 1e-13        # same incoming flux at the top ...
    d = f1.arrays["sw_dn"][-1][day] / f0.arrays["sw_dn"][-1][day] - 1.0
    assert np.all(d < 0.0) and np.abs(d).max() < 5e-3                           # ... more of it in the absorbed ultraviolet at solar maximum
    assert rel_err(0.5 * (f1.arrays["sw_up"] + fm.arrays["sw_up"]), f0.arrays["sw_up"]) < 1e-12      # linear in the multiplier


@pytest.mark.gpu
@pytest.mark.parametrize("multiplier", [1.0, -0.5])
def test_hip_matches_oracle_with_the_solar_cycle(multiplier, oracle_lib):
    _, f_hip, rad = _run("hip", multiplier)
    rad.close()
    _, f_ora, _ = _run(oracle_lib.backend, multiplier)
    compare_flux(f_hip, f_ora, 1.0e-8)
    _, f_0, _ = _run(oracle_lib.backend, 0.0)
    assert rel_err(f_hip.arrays["sw_dn"][-1], f_0.arrays["sw_dn"][-1]) > 1e-5      # (the multiplier reached the device)
