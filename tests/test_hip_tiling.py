"""Column tiling inside ecrad_hip_radiation (include/ecrad_hip.h: ecrad_hip_set_work_bytes) and the argument checks of
ecrad_hip_setup / ecrad_hip_radiation that keep a NULL table from reaching the device."""
import ctypes as C

import numpy as np
import pytest

from ecrad_amd import abi
from ecrad_amd.interface import EcradHipError, Radiation, build_config_struct, load_library, setup_radiation
from ecrad_amd.types import Flux
from helpers import load_meridian, make_config, make_config_rrtmg, run_case

pytestmark = pytest.mark.gpu


def _replicate(inputs, times):
    from test_hip_parity import _replicate as rep
    return rep(inputs, times)


def _run(config, inputs, work_bytes=None, device=False):
    rad = Radiation(config, backend="hip")
    if work_bytes:
        assert rad.lib.ecrad_hip_set_work_bytes(rad.handle, work_bytes) == 0
    ncol, nlev, sl, th, gas, cloud, aer = inputs
    rad.set_gas_units(gas)
    th.calc_saturation_wrt_liquid()
    flux = Flux.allocate(config, ncol, nlev)
    if device:
        import torch
        from ecrad_amd.device import DeviceCase
        case = DeviceCase(config, ncol, nlev, sl, th, gas, cloud, aer, flux)
        st = rad.lib.ecrad_hip_radiation(rad.handle, ncol, nlev, 1, ncol, C.byref(case.inputs), C.byref(case.flux))
        assert st == 0, rad.lib.ecrad_hip_last_error(rad.handle).decode()
        rad.lib.ecrad_hip_synchronize(rad.handle)
        torch.cuda.synchronize()
        case.flux_to_host(flux)
        frac = case.tensors["cloud_fraction"].cpu().numpy() if "cloud_fraction" in case.tensors else None
    else:
        rad.radiation(ncol, nlev, 1, ncol, sl, th, gas, cloud, aer, flux)
        frac = None if cloud is None else cloud.fraction.copy()
    info = abi.CallInfo()
    assert rad.lib.ecrad_hip_last_call_info(rad.handle, C.byref(info)) == 0
    ms = C.c_double()
    assert rad.lib.ecrad_hip_last_kernel_ms(rad.handle, C.byref(ms)) == 0 and ms.value > 0.0
    rad.close()
    return flux, frac, info


@pytest.mark.parametrize("device", [False, True], ids=["host_memory", "device_memory"])
@pytest.mark.parametrize("case", ["tripleclouds_ecckd", "mcica_rrtmg", "mcica_ecckd"])
def test_tiled_call_is_bitwise_the_untiled_call(case, device):
    """9 600 columns (300 copies of the meridian slice) in one tile, and with a work budget so small that the call
    runs as three tiles of 4 096 columns (the minimum): identical bits in every output and in the cropped
    cloud fraction."""
    mk = {"tripleclouds_ecckd": lambda: make_config("Tripleclouds"),
          "mcica_rrtmg": lambda: make_config_rrtmg("McICA", do_lw_aerosol_scattering=False),
          "mcica_ecckd": lambda: make_config("McICA")}[case]
    c1, c2 = mk(), mk()
    # (a host-memory call of 8 192 columns or more is pipelined as tiles by default, tests/test_hip_pool.py: the one-tile call
    #  that the tiled one is compared with has that switched off; the tiled call then runs its three tiles through the pipeline)
    import os
    os.environ["ECRAD_HIP_NO_PIPELINE"] = "1"
    try:
        f1, frac1, info1 = _run(c1, _replicate(load_meridian(c1), 300), device=device)
    finally:
        del os.environ["ECRAD_HIP_NO_PIPELINE"]
    f2, frac2, info2 = _run(c2, _replicate(load_meridian(c2), 300), work_bytes=1 << 20, device=device)
    assert info1.n_tiles == 1 and info1.tile_columns == 9600
    assert info2.n_tiles == 3 and info2.tile_columns == 4096
    assert info2.work_bytes < info1.work_bytes
    for name, a in f1.arrays.items():
        assert np.array_equal(a, f2.arrays[name]), name
    assert np.array_equal(frac1, frac2)
    if case == "mcica_rrtmg":
        # (140 g-points as chunks of 64 + 64 + 16 lanes, 112 as 64 + 32 + 16: chunk_plan in setup.hip; lanes = the widest)
        assert (info1.launches_lw, info1.lanes_lw, info1.launches_sw, info1.lanes_sw) == (3, 64, 3, 64)


def _setup_status(config, edit):
    setup_radiation(config)
    cc, keep = build_config_struct(config)
    edit(cc)
    lib = load_library()
    h = C.c_void_p()
    assert lib.ecrad_hip_create(C.byref(h), -1) == 0
    st = lib.ecrad_hip_setup(h, C.byref(cc))
    msg = lib.ecrad_hip_last_error(h).decode()
    lib.ecrad_hip_destroy(h)
    return st, msg


@pytest.mark.parametrize("what", ["i_emiss_from_band_lw", "i_albedo_from_band_sw", "aerosol_iclass", "cloud_ssa",
                                  "norm_solar_irradiance", "rayleigh_molar_scat", "rh_lower"])
def test_setup_rejects_missing_tables_instead_of_faulting(what):
    null_i = C.POINTER(C.c_int32)()
    null_d = C.POINTER(C.c_double)()

    def edit(cc):
        if what == "i_emiss_from_band_lw":
            cc.do_nearest_spectral_lw_emiss = 1
            cc.i_emiss_from_band_lw = null_i
        elif what == "i_albedo_from_band_sw":
            cc.do_nearest_spectral_sw_albedo = 1
            cc.i_albedo_from_band_sw = null_i
        elif what == "aerosol_iclass":
            cc.aerosol_optics.iclass = null_i
        elif what == "cloud_ssa":
            cc.cloud_optics_sw[0].ssa = null_d
        elif what == "norm_solar_irradiance":
            cc.gas_optics_sw.norm_solar_irradiance = null_d
        elif what == "rayleigh_molar_scat":
            cc.gas_optics_sw.rayleigh_molar_scat = null_d
        elif what == "rh_lower":
            cc.aerosol_optics.rh_lower = null_d
    st, msg = _setup_status(make_config("Tripleclouds"), edit)
    assert st == -1 and msg, (st, msg)       # ECRAD_EINVAL


def test_solar_cycle_without_amplitude_table_is_an_error():
    """radiation_ecckd.F90:955-961: a non-zero spectral_solar_cycle_multiplier with a gas-optics file that carries no
    solar-cycle information aborts in the reference; here it is ECRAD_EINVAL (the shipped 32-term SW model has none)."""
    config = make_config("Cloudless")
    rad = Radiation(config, backend="hip")
    assert config.gas_optics_sw.norm_amplitude_solar_irradiance is None
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    rad.set_gas_units(gas)
    th.calc_saturation_wrt_liquid()
    sl.spectral_solar_cycle_multiplier = 0.5
    flux = Flux.allocate(config, ncol, nlev)
    with pytest.raises(EcradHipError, match="solar cycle"):
        rad.radiation(ncol, nlev, 1, ncol, sl, th, gas, cloud, aer, flux)
    rad.close()


@pytest.mark.parametrize("case", ["tripleclouds_ecckd", "mcica_rrtmg", "spartacus_ecckd", "homogeneous_ecckd"])
def test_column_order_does_not_change_the_results(case, monkeypatch):
    """The cloudy solvers take the columns of every window of 16 / 256 in the order of their cloud structure
    (column_order_kernel, kernel_prep.hip) so that the columns sharing a wave or a block have similar work; no sum runs
    over columns, so every output must have the same bits as with the columns taken as they come."""
    mk = {"tripleclouds_ecckd": lambda: make_config("Tripleclouds"),
          "mcica_rrtmg": lambda: make_config_rrtmg("McICA", do_lw_aerosol_scattering=False),
          "spartacus_ecckd": lambda: make_config("SPARTACUS", do_3d_effects=True, do_lw_derivatives=True),
          "homogeneous_ecckd": lambda: make_config("Homogeneous")}[case]
    c1, c2 = mk(), mk()
    f1, frac1, _ = _run(c1, _replicate(load_meridian(c1), 40))
    monkeypatch.setenv("ECRAD_NO_COLUMN_ORDER", "1")
    f2, frac2, _ = _run(c2, _replicate(load_meridian(c2), 40))
    for name, a in f1.arrays.items():
        assert np.array_equal(a, f2.arrays[name]), name
    assert np.array_equal(frac1, frac2)
