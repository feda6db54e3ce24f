"""Named configurations of the path: the reference's two test namelists and namelist-style edits to them.

`make_config` starts from tests/golden/configCY49R1_ecckd.nam (byte-identical to the reference's
test/ifs/configCY49R1_ecckd.nam: a configuration fixture, data); `make_config_rrtmg` expresses
test/ifs/configCY49R1.nam as its differences from that file (the `diff` of the two namelists).  Tests, bench.py
and smoke() all build their configurations here, the way test/common/change_namelist.sh does for the
reference's own test targets (test/ifs/Makefile:34-118)."""
from __future__ import annotations

import os

import numpy as np

from .config import Config, SOLVER_NAMES
from .driver import DriverConfig, read_input
from .types import Flux

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA_DIR = os.path.join(ROOT, "data")
NAMELIST = os.path.join(ROOT, "tests", "golden", "configCY49R1_ecckd.nam")
MERIDIAN = os.path.join(ROOT, "tests", "golden", "ecrad_meridian.nc")


def make_config(sw_solver="Tripleclouds", lw_solver=None, **overrides) -> Config:
    """configCY49R1_ecckd.nam (the reference's ecCKD test configuration) + change_namelist-style edits."""
    c = Config.read(NAMELIST)
    c.directory_name = DATA_DIR
    c.i_solver_sw = SOLVER_NAMES.index(sw_solver)
    c.i_solver_lw = SOLVER_NAMES.index(lw_solver or sw_solver)
    # Spectral flux *profiles* (nspec x ncol x nlev+1 outputs) are off unless a case asks for them; the
    # reference's McICA configuration forces them off too (radiation_config.F90:1331-1334).
    c.do_save_spectral_flux = False
    for k, v in overrides.items():
        if not hasattr(c, k):
            raise AttributeError(f"config_type has no component {k}")
        setattr(c, k, v)
    return c


def make_config_rrtmg(sw_solver="McICA", lw_solver=None, **overrides) -> Config:
    """The reference's RRTMG test configuration test/ifs/configCY49R1.nam: gas_model_name "RRTMG-IFS",
    SOCRATES/Fu-IFS band cloud optics, cloud and aerosol optics per band, nearest-interval longwave emissivity,
    spectral surface fluxes, unweighted surface mapping."""
    from .config import IGasModelIFSRRTMG, IIceModelFu, ILiquidModelSOCRATES
    kw = dict(i_gas_model_sw=IGasModelIFSRRTMG, i_gas_model_lw=IGasModelIFSRRTMG, use_general_cloud_optics=False,
              i_liq_model=ILiquidModelSOCRATES, i_ice_model=IIceModelFu, do_cloud_aerosol_per_sw_g_point=False,
              do_cloud_aerosol_per_lw_g_point=False, do_nearest_spectral_lw_emiss=True, do_surface_sw_spectral_flux=True,
              do_weighted_surface_mapping=False)
    kw.update(overrides)
    return make_config(sw_solver, lw_solver, **kw)


def load_meridian(config):
    """The reference's 32-column pole-to-pole slice (test/ifs/ecrad_meridian.nc) read as its driver reads it."""
    dc = DriverConfig.read(NAMELIST)
    return read_input(MERIDIAN, config, dc)


# The reference's golden output files (test/ifs/*_out_REFERENCE.nc, committed under tests/golden/) and the
# configuration that produced each (test/ifs/Makefile:34-118).  Both namelists set do_save_spectral_flux = true
# and do_lw_aerosol_scattering = false; McICA forces the spectral profiles off (radiation_config.F90:1331-1334),
# so only the Tripleclouds and Cloudless files hold the (column, half_level, band) profiles.
GOLDEN_CASES = {
    "ecckd_mcica": ("ecckd", dict(sw_solver="McICA")),
    "default": ("rrtmg", dict(sw_solver="McICA", do_lw_aerosol_scattering=False)),
    "noaer": ("rrtmg", dict(sw_solver="McICA", use_aerosols=False, do_lw_aerosol_scattering=False)),
    "expexp": ("rrtmg", dict(sw_solver="McICA", i_overlap_scheme=2, do_lw_aerosol_scattering=False)),
    "tripleclouds": ("rrtmg", dict(sw_solver="Tripleclouds", do_save_spectral_flux=True, do_lw_aerosol_scattering=False)),
    "cloudless": ("rrtmg", dict(sw_solver="Cloudless", use_aerosols=False, do_save_spectral_flux=True,
                                do_lw_aerosol_scattering=False)),
}


def make_golden_config(name: str) -> Config:
    family, kw = GOLDEN_CASES[name]
    kw = dict(kw)
    sw = kw.pop("sw_solver")
    return make_config_rrtmg(sw, **kw) if family == "rrtmg" else make_config(sw, **kw)


def run_case(config, backend, columns=None, inputs=None):
    """Run radiation() over the meridian slice (or given inputs). Returns (flux, thermodynamics, rad)."""
    from .interface import Radiation
    rad = Radiation(config, backend=backend)
    ncol, nlev, sl, th, gas, cloud, aer = inputs if inputs is not None else load_meridian(config)
    rad.set_gas_units(gas)
    th.calc_saturation_wrt_liquid()
    flux = Flux.allocate(config, ncol, nlev)
    i1, i2 = columns if columns else (1, ncol)
    rad.radiation(ncol, nlev, i1, i2, sl, th, gas, cloud, aer, flux)
    return flux, th, rad


def rel_err(a, b, floor_frac=1e-3):
    """max |a-b| / max(|b|, floor_frac*max|b|): relative error with a floor so that near-zero
    entries (e.g. night-time SW) are judged against the field's scale."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b), floor_frac * np.abs(b).max() + 1e-300)
    return float(np.max(np.abs(a - b) / scale))


def compare_flux(f1: Flux, f2: Flux, tol: float, cols=None):
    """Assert every allocated member of two Flux objects agrees to relative tolerance tol."""
    worst = {}
    for name, a in f1.arrays.items():
        b = f2.arrays[name]
        if cols is not None:
            i1, i2 = cols
            if a.ndim == 1:
                a, b = a[i1 - 1:i2], b[i1 - 1:i2]
            elif a.shape[-1] == f1.ncol:
                a, b = a[..., i1 - 1:i2], b[..., i1 - 1:i2]
            else:
                a, b = a[i1 - 1:i2], b[i1 - 1:i2]
        worst[name] = rel_err(a, b)
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, f"fields beyond tolerance {tol}: {bad}"
    return worst
