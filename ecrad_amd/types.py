"""Host mirrors of the reference's input/output derived types.

Array convention: every array is a C-contiguous float64 numpy array whose bytes equal the
reference's Fortran array, i.e. numpy shape is the Fortran shape reversed.  Fortran
``pressure_hl(ncol,nlev+1)`` (radiation_thermodynamics.F90:29-49) is numpy ``(nlev+1, ncol)``:
the column index is fastest in memory, exactly as the C-ABI (include/ecrad_hip.h) expects.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import abi
from .tables import (AIR_MOLAR_MASS, GAS_MOLAR_MASS, NMaxGases, IH2O)

IMassMixingRatio, IVolumeMixingRatio = 0, 1   # radiation_gas.F90:28-31


@dataclass
class Thermodynamics:
    """thermodynamics_type (radiation_thermodynamics.F90:29-49)."""
    pressure_hl: np.ndarray            # (nlev+1, ncol) Pa
    temperature_hl: np.ndarray         # (nlev+1, ncol) K
    h2o_sat_liq: Optional[np.ndarray] = None   # (nlev, ncol)

    def calc_saturation_wrt_liquid(self) -> None:
        """radiation_thermodynamics.F90:118-158."""
        p = 0.5 * (self.pressure_hl[:-1] + self.pressure_hl[1:])
        t = 0.5 * (self.temperature_hl[:-1] + self.temperature_hl[1:])
        e_sat = 6.11e2 * np.exp(17.269 * (t - 273.16) / (t - 35.86))
        self.h2o_sat_liq = np.ascontiguousarray(np.minimum(1.0, 0.622 * e_sat / p))


@dataclass
class SingleLevel:
    """single_level_type (radiation_single_level.F90:29-102)."""
    cos_sza: np.ndarray                # (ncol)
    skin_temperature: np.ndarray       # (ncol)
    sw_albedo: np.ndarray              # (nalbedobands, ncol)
    lw_emissivity: np.ndarray          # (nemissbands, ncol)
    sw_albedo_direct: Optional[np.ndarray] = None
    solar_irradiance: float = 1366.0
    spectral_solar_cycle_multiplier: float = 0.0
    spectral_solar_scaling: object = None      # (n_bands_sw) with config%use_spectral_solar_scaling (radiation_single_level.F90:76)
    iseed: Optional[np.ndarray] = None  # (ncol) int32

    def init_seed_simple(self, ncol: int) -> None:
        """radiation_single_level.F90:196-211: iseed(j) = j."""
        self.iseed = np.arange(1, ncol + 1, dtype=np.int32)


@dataclass
class Gas:
    """gas_type (radiation_gas.F90:36-80): mixing_ratio(ncol,nlev,NMaxGases) == numpy (12,nlev,ncol)."""
    mixing_ratio: np.ndarray
    iunits: list = field(default_factory=lambda: [IMassMixingRatio] * (NMaxGases + 1))
    scale_factor: list = field(default_factory=lambda: [1.0] * (NMaxGases + 1))
    is_present: list = field(default_factory=lambda: [False] * (NMaxGases + 1))

    @classmethod
    def allocate(cls, ncol: int, nlev: int) -> "Gas":
        return cls(mixing_ratio=np.zeros((NMaxGases, nlev, ncol)))

    def put(self, igas: int, iunits: int, mixing_ratio, scale_factor: float = 1.0) -> None:
        """put_gas / put_well_mixed_gas (radiation_gas.F90:197-364)."""
        self.mixing_ratio[igas - 1, :, :] = mixing_ratio
        self.iunits[igas] = iunits
        self.scale_factor[igas] = scale_factor
        self.is_present[igas] = True

    def scale(self, igas: int, scale_factor: float) -> None:
        if scale_factor != 1.0:
            self.scale_factor[igas] *= scale_factor

    def set_units(self, iunits: int, scale_factor: float = 1.0) -> None:
        """set_units_gas (radiation_gas.F90:412-470) for every present gas."""
        for igas in range(1, NMaxGases + 1):
            if not self.is_present[igas]:
                continue
            sf = 1.0 / scale_factor
            if iunits == IMassMixingRatio and self.iunits[igas] == IVolumeMixingRatio:
                sf = sf * GAS_MOLAR_MASS[igas] / AIR_MOLAR_MASS
            elif iunits == IVolumeMixingRatio and self.iunits[igas] == IMassMixingRatio:
                sf = sf * AIR_MOLAR_MASS / GAS_MOLAR_MASS[igas]
            sf = sf * self.scale_factor[igas]
            if sf != 1.0:
                self.mixing_ratio[igas - 1] *= sf
            self.iunits[igas] = iunits
            self.scale_factor[igas] = scale_factor

    def get(self, igas: int, iunits: int) -> np.ndarray:
        """get_gas (radiation_gas.F90:555-623)."""
        if not self.is_present[igas]:
            return np.zeros_like(self.mixing_ratio[0])
        sf = 1.0
        if iunits == IMassMixingRatio and self.iunits[igas] == IVolumeMixingRatio:
            sf = sf * GAS_MOLAR_MASS[igas] / AIR_MOLAR_MASS
        elif iunits == IVolumeMixingRatio and self.iunits[igas] == IMassMixingRatio:
            sf = sf * AIR_MOLAR_MASS / GAS_MOLAR_MASS[igas]
        sf = sf * self.scale_factor[igas]
        return self.mixing_ratio[igas - 1] * sf if sf != 1.0 else self.mixing_ratio[igas - 1].copy()


@dataclass
class Cloud:
    """cloud_type (radiation_cloud.F90:33-96)."""
    fraction: np.ndarray               # (nlev, ncol)  -- modified in place by radiation() (crop)
    mixing_ratio: np.ndarray           # (ntype, nlev, ncol)
    effective_radius: np.ndarray       # (ntype, nlev, ncol)
    fractional_std: np.ndarray         # (nlev, ncol)
    overlap_param: np.ndarray          # (nlev-1, ncol)
    inv_cloud_effective_size: Optional[np.ndarray] = None    # (nlev, ncol) m-1: SPARTACUS 3-D effects (radiation_cloud.F90:75-87)
    inv_inhom_effective_size: Optional[np.ndarray] = None    # (nlev, ncol) m-1

    @property
    def ntype(self) -> int:
        return self.mixing_ratio.shape[0]


@dataclass
class Aerosol:
    """aerosol_type (radiation_aerosol.F90:28-57): mixing_ratio(ncol, istartlev:iendlev, ntype)."""
    mixing_ratio: np.ndarray           # (ntype, nlev_aer, ncol)
    istartlev: int = 1
    iendlev: int = 0


@dataclass
class Flux:
    """flux_type (radiation_flux.F90:38-118); members that are None are "not allocated"."""
    ncol: int
    nlev: int
    arrays: dict = field(default_factory=dict)

    def __getattr__(self, name):
        arrays = self.__dict__.get("arrays", {})
        if name in arrays:
            return arrays[name]
        if name in abi.FLUX_FIELDS:
            return None
        raise AttributeError(name)

    @classmethod
    def allocate(cls, config, ncol: int, nlev: int) -> "Flux":
        """allocate_flux_type (radiation_flux.F90:133-326)."""
        f = cls(ncol=ncol, nlev=nlev)
        a = f.arrays
        prof = lambda: np.zeros((nlev + 1, ncol))
        if config.do_lw:
            a["lw_up"], a["lw_dn"] = prof(), prof()
            if config.do_clear:
                a["lw_up_clear"], a["lw_dn_clear"] = prof(), prof()
            if config.do_lw_derivatives:
                a["lw_derivatives"] = prof()
            if config.do_toa_spectral_flux:
                a["lw_up_toa_band"] = np.zeros((ncol, config.n_bands_lw))
                if config.do_clear:
                    a["lw_up_toa_clear_band"] = np.zeros((ncol, config.n_bands_lw))
            a["lw_dn_surf_g"] = np.zeros((ncol, config.n_g_lw))
            a["lw_up_toa_g"] = np.zeros((ncol, config.n_g_lw))
            if config.do_clear:
                a["lw_dn_surf_clear_g"] = np.zeros((ncol, config.n_g_lw))
                a["lw_up_toa_clear_g"] = np.zeros((ncol, config.n_g_lw))
            if config.do_canopy_fluxes_lw:
                a["lw_dn_surf_canopy"] = np.zeros((ncol, config.n_canopy_bands_lw))
            if config.do_save_spectral_flux:        # (nspec, ncol, nlev+1), radiation_flux.F90:156-170
                spec = lambda: np.zeros((nlev + 1, ncol, config.n_spec_lw))
                a["lw_up_band"], a["lw_dn_band"] = spec(), spec()
                if config.do_clear:
                    a["lw_up_clear_band"], a["lw_dn_clear_band"] = spec(), spec()
        if config.do_sw:
            a["sw_up"], a["sw_dn"] = prof(), prof()
            if config.do_sw_direct:
                a["sw_dn_direct"] = prof()
            if config.do_clear:
                a["sw_up_clear"], a["sw_dn_clear"] = prof(), prof()
                if config.do_sw_direct:
                    a["sw_dn_direct_clear"] = prof()
            if config.do_surface_sw_spectral_flux:
                a["sw_dn_surf_band"] = np.zeros((ncol, config.n_bands_sw))
                a["sw_dn_direct_surf_band"] = np.zeros((ncol, config.n_bands_sw))
                if config.do_clear:
                    a["sw_dn_surf_clear_band"] = np.zeros((ncol, config.n_bands_sw))
                    a["sw_dn_direct_surf_clear_band"] = np.zeros((ncol, config.n_bands_sw))
            if config.do_toa_spectral_flux:
                a["sw_dn_toa_band"] = np.zeros((ncol, config.n_bands_sw))
                a["sw_up_toa_band"] = np.zeros((ncol, config.n_bands_sw))
                if config.do_clear:
                    a["sw_up_toa_clear_band"] = np.zeros((ncol, config.n_bands_sw))
            for n in ("sw_dn_diffuse_surf_g", "sw_dn_direct_surf_g", "sw_dn_toa_g", "sw_up_toa_g"):
                a[n] = np.zeros((ncol, config.n_g_sw))
            if config.do_clear:
                for n in ("sw_dn_diffuse_surf_clear_g", "sw_dn_direct_surf_clear_g", "sw_up_toa_clear_g"):
                    a[n] = np.zeros((ncol, config.n_g_sw))
            if config.do_canopy_fluxes_sw:
                a["sw_dn_diffuse_surf_canopy"] = np.zeros((ncol, config.n_canopy_bands_sw))
                a["sw_dn_direct_surf_canopy"] = np.zeros((ncol, config.n_canopy_bands_sw))
            if config.do_save_spectral_flux:        # radiation_flux.F90:219-242
                spec = lambda: np.zeros((nlev + 1, ncol, config.n_spec_sw))
                a["sw_up_band"], a["sw_dn_band"] = spec(), spec()
                if config.do_sw_direct:
                    a["sw_dn_direct_band"] = spec()
                if config.do_clear:
                    a["sw_up_clear_band"], a["sw_dn_clear_band"] = spec(), spec()
                    if config.do_sw_direct:
                        a["sw_dn_direct_clear_band"] = spec()
        a["cloud_cover_lw"] = np.full(ncol, -1.0)
        a["cloud_cover_sw"] = np.full(ncol, -1.0)
        return f
