"""Synthetic "IFS-shaped" column generator for benchmarks.

Column i is built from base profile ``b = i mod 32`` of the reference's meridian test slice
(tests/golden/ecrad_meridian.nc) with seeded perturbations:

* ``pressure_hl`` scaled by ``ps_i/ps_b``, ``ps_i ~ U(0.95,1.05) ps_b``;
* ``temperature_hl += N(0, 2 K)`` (one draw per column), ``skin_temperature = T_hl(surface) + N(0,1)``;
* ``q *= LogN(0, 0.2)``; all other gases, albedos, emissivities, aerosols from ``b``;
* ``cos_sza ~ U(0.05, 1)`` so that no column short-circuits the shortwave;
* clouds (when wanted): fraction of ``b`` times ``U(0.5,1.5)`` clipped to [0,1]; ``iseed = i+1``;
  clear-sky configurations set the fraction to zero everywhere.

Everything is float64 in the C-ABI layout (column index fastest).
"""
from __future__ import annotations

import os

import numpy as np

from .config import Config
from .driver import DriverConfig, read_input
from .types import Aerosol, Cloud, Gas, SingleLevel, Thermodynamics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MERIDIAN = os.path.join(ROOT, "tests", "golden", "ecrad_meridian.nc")
NAMELIST = os.path.join(ROOT, "tests", "golden", "configCY49R1_ecckd.nam")
SEED = 20260929


def make_columns(config: Config, ncol: int, clear_sky: bool, seed: int = SEED, first_column: int = 0,
                 base_file: str = MERIDIAN):
    """Return (ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol) with gas units already
    set for the gas model and h2o_sat_liq computed (i.e. ready for Radiation.radiation)."""
    from .config import ISolverSpartacus
    spartacus = ISolverSpartacus in (config.i_solver_sw, config.i_solver_lw)
    # (SPARTACUS: the cloud sizes come from the test namelist's cloud_separation_scale_* as in test/ifs)
    dc = DriverConfig.read(NAMELIST) if spartacus else DriverConfig()
    nb, nlev, sl0, th0, gas0, cloud0, aer0 = read_input(base_file, config, dc)
    # set_gas_units: volume mixing ratios for ecCKD, mass mixing ratios for RRTMG (radiation_ifs_rrtm.F90:203-213)
    gas0.set_units(0 if getattr(config, "rrtmg", None) is not None else 1)
    rng = np.random.default_rng([seed, first_column])
    # column i takes base profile (first_column + i) mod nb: the base arrays rolled to the first profile and tiled
    # (a fancy-index gather along the fastest axis is ~10x slower for the 100 000-column batches)
    reps = -(-ncol // nb)

    group = int(os.environ.get("ECRAD_SYNTH_SAME_PROFILE", "1"))     # (diagnostic) runs of this many columns share a base profile

    def take(a):
        r = np.roll(a, -(first_column % nb), axis=-1)
        if group > 1:
            r = np.repeat(r, group, axis=-1)
            out = np.empty(a.shape[:-1] + (-(-ncol // (nb * group)), nb * group), dtype=a.dtype)
            out[...] = r[..., None, :]
            return out.reshape(a.shape[:-1] + (-1,))[..., :ncol]
        out = np.empty(a.shape[:-1] + (reps, nb), dtype=a.dtype)
        out[...] = r[..., None, :]
        return out.reshape(a.shape[:-1] + (reps * nb,))[..., :ncol]
    ps_scale = rng.uniform(0.95, 1.05, ncol)
    dT = rng.normal(0.0, 2.0, ncol)
    qfac = np.exp(rng.normal(0.0, 0.2, ncol))
    cos_sza = rng.uniform(0.05, 1.0, ncol)
    dTskin = rng.normal(0.0, 1.0, ncol)
    cf_fac = rng.uniform(0.5, 1.5, ncol)

    pressure_hl = np.ascontiguousarray(take(th0.pressure_hl) * ps_scale[None, :])
    temperature_hl = np.ascontiguousarray(take(th0.temperature_hl) + dT[None, :])
    th = Thermodynamics(pressure_hl, temperature_hl)
    th.calc_saturation_wrt_liquid()
    sl = SingleLevel(
        cos_sza=np.ascontiguousarray(cos_sza),
        skin_temperature=np.ascontiguousarray(temperature_hl[nlev] + dTskin),
        sw_albedo=np.ascontiguousarray(take(sl0.sw_albedo)),
        lw_emissivity=np.ascontiguousarray(take(sl0.lw_emissivity)),
        sw_albedo_direct=None if sl0.sw_albedo_direct is None else np.ascontiguousarray(take(sl0.sw_albedo_direct)),
        solar_irradiance=sl0.solar_irradiance,
        iseed=(first_column + 1 + np.arange(ncol)).astype(np.int32))
    gas = Gas(mixing_ratio=np.ascontiguousarray(take(gas0.mixing_ratio)))
    gas.iunits = list(gas0.iunits)
    gas.scale_factor = list(gas0.scale_factor)
    gas.is_present = list(gas0.is_present)
    gas.mixing_ratio[0] *= qfac[None, :]          # H2O
    cloud = None
    if config.do_clouds:
        frac = np.zeros((nlev, ncol)) if clear_sky else np.clip(take(cloud0.fraction) * cf_fac[None, :], 0.0, 1.0)
        cloud = Cloud(fraction=np.ascontiguousarray(frac),
                      mixing_ratio=np.ascontiguousarray(take(cloud0.mixing_ratio)),
                      effective_radius=np.ascontiguousarray(take(cloud0.effective_radius)),
                      fractional_std=np.ascontiguousarray(take(cloud0.fractional_std)),
                      overlap_param=np.ascontiguousarray(take(cloud0.overlap_param)))
        if spartacus:       # the cloud scales of the base profile (radiation_cloud.F90:602-690)
            cloud.inv_cloud_effective_size = np.ascontiguousarray(take(cloud0.inv_cloud_effective_size))
            if cloud0.inv_inhom_effective_size is not None:
                cloud.inv_inhom_effective_size = np.ascontiguousarray(take(cloud0.inv_inhom_effective_size))
    aerosol = None
    if config.use_aerosols and aer0 is not None:
        aerosol = Aerosol(mixing_ratio=np.ascontiguousarray(take(aer0.mixing_ratio)),
                          istartlev=aer0.istartlev, iendlev=aer0.iendlev)
    return ncol, nlev, sl, th, gas, cloud, aerosol


# The BASELINE.json configurations that the implemented scope covers, as namelist-style edits to
# test/ifs/configCY49R1_ecckd.nam.
BENCH_CONFIGS = {
    # configs[1]: 100k clear-sky columns, ecCKD-32 SW+LW, homogeneous solver, double precision
    "clear_homogeneous_ecckd32": dict(sw_solver="Homogeneous", use_aerosols=False, clear_sky=True),
    # north-star target configuration: ecCKD-32 Tripleclouds with clouds and aerosols
    "tripleclouds_ecckd32": dict(sw_solver="Tripleclouds", use_aerosols=True, clear_sky=False),
    "mcica_ecckd32": dict(sw_solver="McICA", use_aerosols=True, clear_sky=False),
    # BASELINE configs[3] shape on one GPU: Tripleclouds with the 64-term SW model (LW-64 is a missing blob upstream)
    "tripleclouds_ecckd64": dict(sw_solver="Tripleclouds", use_aerosols=True, clear_sky=False,
                                 gas_optics_sw_override_file_name="ecckd-1.2_sw_climate_window-64b_ckd-definition.nc"),
    # the 96-term SW model: three launches of 32 g-points
    "tripleclouds_ecckd96": dict(sw_solver="Tripleclouds", use_aerosols=True, clear_sky=False,
                                 gas_optics_sw_override_file_name="ecckd-1.4_sw_climate_vfine-96b_ckd-definition.nc"),
    # attribution variants (not bench lines): the same without aerosols / without clouds
    "tripleclouds_noaer": dict(sw_solver="Tripleclouds", use_aerosols=False, clear_sky=False),
    "tripleclouds_clear_aer": dict(sw_solver="Tripleclouds", use_aerosols=True, clear_sky=True),
    "tripleclouds_clear_noaer": dict(sw_solver="Tripleclouds", use_aerosols=False, clear_sky=True),
    "mcica_noaer": dict(sw_solver="McICA", use_aerosols=False, clear_sky=False),
    "mcica_vectorizable": dict(sw_solver="McICA", use_aerosols=True, clear_sky=False, use_vectorizable_generator=True),
    "homogeneous_clear_aer": dict(sw_solver="Homogeneous", use_aerosols=True, clear_sky=True),
    "cloudless_clear_noaer": dict(sw_solver="Cloudless", use_aerosols=False, clear_sky=True),
    # BASELINE configs[2]: RRTMG 140/112 g-points, McICA with clouds (the reference's default configuration
    # test/ifs/configCY49R1.nam: SOCRATES/Fu band cloud optics, 12 aerosol types, no LW aerosol scattering)
    "mcica_rrtmg": dict(sw_solver="McICA", use_aerosols=True, clear_sky=False, rrtmg=True, do_lw_aerosol_scattering=False),
    "mcica_rrtmg_noaer": dict(sw_solver="McICA", use_aerosols=False, clear_sky=False, rrtmg=True, do_lw_aerosol_scattering=False),
    # BASELINE configs[4]: ecCKD-32, SPARTACUS (3 regions, 3-D effects, explicit entrapment), single precision
    # (do_3d_effects is spelt out: the test namelist the configurations start from switches it off)
    "spartacus_ecckd32_sp": dict(sw_solver="SPARTACUS", use_aerosols=True, clear_sky=False, i_precision=1, do_3d_effects=True),
    "spartacus_ecckd32_dp": dict(sw_solver="SPARTACUS", use_aerosols=True, clear_sky=False, do_3d_effects=True),
    "tripleclouds_rrtmg": dict(sw_solver="Tripleclouds", use_aerosols=True, clear_sky=False, rrtmg=True, do_lw_aerosol_scattering=False),
}
