"""Offline-driver counterpart: netCDF input -> types, types -> netCDF output.

Mirrors driver/ecrad_driver_read_input.F90:21-620 (variable names, unit handling, overrides) and
the variable names of radiation/radiation_save.F90:35-460 (save_fluxes).  The program flow of
driver/ecrad_driver.F90:28-420 is in :func:`main`:  ``python -m ecrad_amd.driver cfg.nam in.nc out.nc``.
"""
from __future__ import annotations

import sys
import time
from dataclasses import dataclass

import numpy as np

from .config import Config
from .namelist import read_namelist
from .ncfile import NcFile, write_nc
from .hdf5file import write_nc4
from .tables import GAS_LOWER_CASE_NAMES, NMaxGases, IH2O, IO3
from .types import (Aerosol, Cloud, Flux, Gas, SingleLevel, Thermodynamics,
                    IMassMixingRatio, IVolumeMixingRatio)


@dataclass
class DriverConfig:
    """driver_config_type (driver/ecrad_driver_config.F90:27-140), defaults :230-300."""
    do_parallel: bool = True
    nblocksize: int = 8
    istartcol: int = 0
    iendcol: int = 0
    nrepeat: int = 1
    iverbose: int = 2
    do_write_double_precision: bool = False
    do_save_net_fluxes: bool = False
    do_save_inputs: bool = False             # the inputs as radiation() gets them, to "inputs.nc" (ecrad_driver.F90:283-289)
    do_save_aerosol_optics: bool = False     # the mapped aerosol tables, to "aerosol_optics.nc" (ecrad_driver.F90:224-226)
    do_save_cloud_optics: bool = False       # the mapped general cloud-optics tables, "hydrometeor_optics_{sw,lw}_<type>.nc" (:228-230)
    do_write_hdf5: bool = False              # netCDF-4/HDF5 output instead of classic netCDF (ecrad_driver_config.F90:119)
    # shortwave diagnostics in user-specified wavelength intervals (m), written to a second file
    # (driver/ecrad_driver_config.F90:72-82: the first negative bound ends the list)
    sw_diag_wavelength_bound: list = None
    sw_diag_file_name: str = "sw_diagnostics.nc"
    experiment_name: str = ""
    fractional_std_override: float = -1.0
    overlap_decorr_length_override: float = -1.0
    overlap_decorr_length_scaling: float = -1.0
    sw_albedo_override: float = -1.0
    lw_emissivity_override: float = -1.0
    q_liq_scaling: float = -1.0
    q_ice_scaling: float = -1.0
    cloud_fraction_scaling: float = -1.0
    skin_temperature_override: float = -1.0
    solar_irradiance_override: float = -1.0
    cos_sza_override: float = -1.0
    solar_cycle_multiplier_override: float = -2.0e6      # ecrad_driver_config.F90:262
    low_inv_effective_size_override: float = -1.0
    middle_inv_effective_size_override: float = -1.0
    high_inv_effective_size_override: float = -1.0
    cloud_separation_scale_surface: float = -1.0
    cloud_separation_scale_toa: float = -1.0
    cloud_separation_scale_power: float = 1.0
    cloud_inhom_separation_factor: float = 1.0
    effective_size_scaling: float = -1.0
    do_ignore_inhom_effective_size: bool = False
    do_correct_unphysical_inputs: bool = False           # clip inputs outside their physical range (ecrad_driver.F90:313-323)
    vmr_suffix_str: str = "_vmr"
    gas_scaling: dict = None

    @classmethod
    def read(cls, file_name: str) -> "DriverConfig":
        nml = read_namelist(file_name).get("radiation_driver", {})
        d = cls()
        d.gas_scaling = {}
        # the namelist names of the overrides differ from the names of the members they set (ecrad_driver_config.F90:216-234,
        # :328-383); the member names are accepted as well
        for k, v in nml.items():
            k = NAMELIST_TO_MEMBER.get(k, k)
            if k.endswith("_scaling") and k[:-8] in GAS_LOWER_CASE_NAMES:
                d.gas_scaling[k[:-8]] = float(v)
            elif k == "inv_effective_size":
                pass
            elif hasattr(d, k) and k != "gas_scaling" and v is not None:
                cur = getattr(d, k)
                setattr(d, k, float(v) if isinstance(cur, float) else v)
        # one inverse effective size for all heights, which the three height ranges override (:337-363)
        ies = nml.get("inv_effective_size")
        ranges = ("high_inv_effective_size_override", "middle_inv_effective_size_override", "low_inv_effective_size_override")
        some = any(getattr(d, r) >= 0.0 for r in ranges)
        if ies is not None and float(ies) >= 0.0:
            for r, given in zip(ranges, ("high_inv_effective_size", "middle_inv_effective_size", "low_inv_effective_size")):
                if not (given in nml or r in nml) or getattr(d, r) < 0.0:
                    setattr(d, r, float(ies))
        if some and any(getattr(d, r) < 0.0 for r in ranges):
            raise ValueError("Driver configuration error: inverse effective cloud size not specified for high, middle and low clouds")
        return d


NAMELIST_TO_MEMBER = {
    "fractional_std": "fractional_std_override", "overlap_decorr_length": "overlap_decorr_length_override",
    "sw_albedo": "sw_albedo_override", "lw_emissivity": "lw_emissivity_override", "q_liquid_scaling": "q_liq_scaling",
    "skin_temperature": "skin_temperature_override", "cos_solar_zenith_angle": "cos_sza_override",
    "high_inv_effective_size": "high_inv_effective_size_override", "middle_inv_effective_size": "middle_inv_effective_size_override",
    "low_inv_effective_size": "low_inv_effective_size_override",
}


def out_of_physical_bounds(istartcol: int, iendcol: int, do_fix: bool, single_level, thermodynamics, gas, cloud, aerosol,
                           out=print) -> bool:
    """The driver's check of its inputs (ecrad_driver.F90:313-323): every array of the input types against the physical
    range the reference gives it (the `out_of_physical_bounds` of radiation_gas.F90:679, radiation_single_level.F90:401-410,
    radiation_thermodynamics.F90:319-324, radiation_cloud.F90:768-781, radiation_aerosol.F90:179-192; messages of
    radiation_check.F90:124-134), columns istartcol..iendcol; with ``do_fix`` the values are clipped in place.  Pressure is
    never clipped (a layer could end up with no pressure difference)."""
    c = slice(istartcol - 1, iendcol)
    checks = [(gas, "mixing_ratio", "gas%mixing_ratio", 0.0, 1.0, do_fix),
              (single_level, "cos_sza", "cos_sza", -1.0, 1.0, do_fix),
              (single_level, "skin_temperature", "skin_temperature", 173.0, 373.0, do_fix),
              (single_level, "sw_albedo", "sw_albedo", 0.0, 1.0, do_fix),
              (single_level, "sw_albedo_direct", "sw_albedo", 0.0, 1.0, do_fix),
              (single_level, "lw_emissivity", "lw_emissivity", 0.0, 1.0, do_fix),
              (thermodynamics, "pressure_hl", "pressure_hl", 0.0, 110000.0, False),
              (thermodynamics, "temperature_hl", "temperature_hl", 100.0, 400.0, do_fix),
              (thermodynamics, "h2o_sat_liq", "h2o_sat_liq", 0.0, 1.0, do_fix),
              (cloud, "mixing_ratio", "cloud%mixing_ratio", 0.0, 1.0, do_fix),
              (cloud, "effective_radius", "cloud%effective_radius", 0.0, 0.1, do_fix),
              (cloud, "fraction", "cloud%fraction", 0.0, 1.0, do_fix),
              (cloud, "fractional_std", "fractional_std", 0.0, 10.0, do_fix),
              (cloud, "inv_cloud_effective_size", "inv_cloud_effective_size", 0.0, 1.0, do_fix),
              (cloud, "inv_inhom_effective_size", "inv_inhom_effective_size", 0.0, 1.0, do_fix),
              (cloud, "overlap_param", "overlap_param", -0.5, 1.0, do_fix),
              (aerosol, "mixing_ratio", "aerosol%mixing_ratio", 0.0, 1.0, do_fix)]
    is_bad = False
    for obj, member, name, lo, hi, fix in checks:
        var = getattr(obj, member, None) if obj is not None else None
        if var is None or not isinstance(var, np.ndarray) or var.size == 0:
            continue
        part = var[..., c]                      # (the column is the fastest index of every input array)
        vmin, vmax = float(part.min()), float(part.max())
        if vmin < lo or vmax > hi:
            is_bad = True
            out(f"*** Warning: {name} range{vmin:12.4g} to{vmax:12.4g} is out of physical range{lo:12.4g}to{hi:12.4g}"
                + (": corrected" if fix else ""))
            if fix:
                np.clip(part, lo, hi, out=part)
    return is_bad


def _colfast(a: np.ndarray) -> np.ndarray:
    """netCDF (column, x) -> numpy (x, column): what file%transpose_matrices(.true.) achieves
    (driver/ecrad_driver.F90:255, utilities/easy_netcdf.F90:1040-1150)."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).T)


def read_input(path: str, config: Config, driver_config: DriverConfig):
    """read_input (driver/ecrad_driver_read_input.F90:21-620).

    Returns (ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol)."""
    dc = driver_config
    with NcFile(path) as f:
        pressure_hl = _colfast(f.get("pressure_hl"))
        temperature_hl = _colfast(f.get("temperature_hl"))
        nlev = pressure_hl.shape[0] - 1
        ncol = pressure_hl.shape[1]
        thermodynamics = Thermodynamics(pressure_hl, temperature_hl)

        if dc.solar_irradiance_override > 0.0:
            solar_irradiance = dc.solar_irradiance_override
        elif f.exists("solar_irradiance"):
            solar_irradiance = f.get_scalar("solar_irradiance")
        else:
            solar_irradiance = 1366.0
        if dc.cos_sza_override >= 0.0:
            cos_sza = np.full(ncol, dc.cos_sza_override)
        elif f.exists("cos_solar_zenith_angle"):
            cos_sza = np.ascontiguousarray(f.get("cos_solar_zenith_angle"), dtype=np.float64)
        elif not config.do_sw:
            cos_sza = np.zeros(ncol)
        else:
            raise RuntimeError("cos_solar_zenith_angle not provided")

        cloud = None
        iseed = None
        if config.do_clouds:
            fraction = _colfast(f.get("cloud_fraction"))
            fractional_std = _colfast(f.get("fractional_std")) if f.exists("fractional_std") else None
            if f.exists("q_hydrometeor"):
                # (column, type, level) -> numpy (type, level, column)
                mixing_ratio = np.ascontiguousarray(np.transpose(f.get("q_hydrometeor"), (1, 2, 0)))
                effective_radius = np.ascontiguousarray(np.transpose(f.get("re_hydrometeor"), (1, 2, 0)))
            else:
                mixing_ratio = np.stack([_colfast(f.get("q_liquid")), _colfast(f.get("q_ice"))])
                effective_radius = np.stack([_colfast(f.get("re_liquid")), _colfast(f.get("re_ice"))])
            iseed = np.arange(1, ncol + 1, dtype=np.int32)          # init_seed_simple
            if f.exists("iseed"):
                iseed = np.ascontiguousarray(f.get("iseed")).astype(np.int32)
            overlap_param = _colfast(f.get("overlap_param")) if f.exists("overlap_param") else None
            if dc.q_liq_scaling >= 0.0 and dc.q_liq_scaling != 1.0:
                mixing_ratio[0] *= dc.q_liq_scaling
            if dc.q_ice_scaling >= 0.0 and dc.q_ice_scaling != 1.0:
                mixing_ratio[1] *= dc.q_ice_scaling
            if dc.cloud_fraction_scaling >= 0.0 and dc.cloud_fraction_scaling != 1.0:
                fraction *= dc.cloud_fraction_scaling
            if dc.overlap_decorr_length_override > 0.0 or overlap_param is None:
                # cloud%set_overlap_param with the namelist's decorrelation length, or the driver's default of 2 km when the
                # file has no overlap parameter (ecrad_driver_read_input.F90:68, :233-246)
                from .ifs import set_overlap_param
                overlap_param = np.ascontiguousarray(set_overlap_param(
                    thermodynamics, dc.overlap_decorr_length_override if dc.overlap_decorr_length_override > 0.0 else 2000.0))
            elif dc.overlap_decorr_length_scaling > 0.0:
                pos = overlap_param > 0.0
                overlap_param[pos] = overlap_param[pos] ** (1.0 / dc.overlap_decorr_length_scaling)
            elif dc.overlap_decorr_length_scaling == 0.0:
                overlap_param[:] = 0.0
            if dc.fractional_std_override >= 0.0:
                fractional_std = np.full((nlev, ncol), dc.fractional_std_override)
            elif fractional_std is None:
                fractional_std = np.zeros((nlev, ncol))
            cloud = Cloud(fraction, np.ascontiguousarray(mixing_ratio),
                          np.ascontiguousarray(effective_radius), fractional_std, overlap_param)
            from .config import ISolverSpartacus
            if ISolverSpartacus in (config.i_solver_sw, config.i_solver_lw):
                # driver/ecrad_driver_read_input.F90:290-470: the four ways to specify the cloud scale
                scalable = False
                sizes = (dc.low_inv_effective_size_override, dc.middle_inv_effective_size_override, dc.high_inv_effective_size_override)
                if max(sizes) >= 0.0:
                    # (1) cloud%create_inv_cloud_effective_size_eta (radiation_cloud.F90:524-594) with the driver's bounds
                    # eta = 0.8 and 0.45 between low / mid-level / high clouds (ecrad_driver_read_input.F90:305-331)
                    if min(sizes) < 0.0:
                        raise RuntimeError("if one of [low|middle|high]_inv_effective_size_override is provided then all must be")
                    isurf = 0 if pressure_hl[0, 0] > pressure_hl[1, 0] else nlev
                    eta = (pressure_hl[:-1] + pressure_hl[1:]) * (0.5 / pressure_hl[isurf][None, :])
                    cloud.inv_cloud_effective_size = np.ascontiguousarray(
                        np.where(eta > 0.8, sizes[0], np.where(eta > 0.45, sizes[1], sizes[2])))
                elif dc.cloud_separation_scale_surface > 0.0 and dc.cloud_separation_scale_toa > 0.0:
                    # (2) cloud%param_cloud_effective_separation_eta (radiation_cloud.F90:602-690): what the IFS test
                    # namelists use (cloud_separation_scale_*)
                    coeff_e = 1.0 - np.exp(-1.0)
                    coeff_b = (dc.cloud_separation_scale_toa - dc.cloud_separation_scale_surface) / coeff_e
                    coeff_a = dc.cloud_separation_scale_toa - coeff_b
                    isurf = 0 if pressure_hl[0, 0] > pressure_hl[1, 0] else nlev
                    eta = (pressure_hl[:-1] + pressure_hl[1:]) * (0.5 / pressure_hl[isurf][None, :])
                    eff_separation = coeff_a + coeff_b * np.exp(-eta ** dc.cloud_separation_scale_power)
                    cloud.inv_cloud_effective_size = np.ascontiguousarray(
                        1.0 / (eff_separation * np.sqrt(np.maximum(1.0e-5, fraction * (1.0 - fraction)))))
                    cloud.inv_inhom_effective_size = np.ascontiguousarray(
                        1.0 / (eff_separation * dc.cloud_inhom_separation_factor
                               * np.sqrt(np.maximum(1.0e-5, 0.5 * fraction * (1.0 - 0.5 * fraction)))))
                elif f.exists("inv_cloud_effective_size"):      # (3)
                    scalable = True
                    cloud.inv_cloud_effective_size = _colfast(f.get("inv_cloud_effective_size"))
                    if f.exists("inv_inhom_effective_size") and not dc.do_ignore_inhom_effective_size:
                        cloud.inv_inhom_effective_size = _colfast(f.get("inv_inhom_effective_size"))
                elif f.exists("inv_cloud_effective_separation"):      # (4) :379-434
                    scalable = True
                    thr = config.cloud_fraction_threshold
                    sep = _colfast(f.get("inv_cloud_effective_separation"))
                    partly = (fraction > thr) & (fraction < 1.0 - thr)
                    cloud.inv_cloud_effective_size = np.ascontiguousarray(
                        np.where(partly, sep / np.sqrt(np.where(partly, fraction * (1.0 - fraction), 1.0)), 0.0))
                    if f.exists("inv_inhom_effective_separation"):
                        isep = _colfast(f.get("inv_inhom_effective_separation"))
                    else:      # the separation of the clouds, divided by the user's factor (inverse sizes)
                        isep = (1.0 / dc.cloud_inhom_separation_factor) * sep
                    cloudy = fraction > thr
                    cloud.inv_inhom_effective_size = np.ascontiguousarray(
                        np.where(cloudy, isep / np.sqrt(np.where(cloudy, 0.5 * fraction * (1.0 - 0.5 * fraction), 1.0)), 0.0))
                else:
                    raise RuntimeError("SPARTACUS solver specified but cloud size not, either in namelist or input file")
                if scalable and dc.effective_size_scaling > 0.0:           # :443-461
                    cloud.inv_cloud_effective_size /= dc.effective_size_scaling
                    if cloud.inv_inhom_effective_size is not None:
                        cloud.inv_inhom_effective_size /= dc.effective_size_scaling

        if f.exists("skin_temperature"):
            skin_temperature = np.ascontiguousarray(f.get("skin_temperature"), dtype=np.float64)
        else:
            skin_temperature = temperature_hl[nlev].copy()
        if dc.sw_albedo_override >= 0.0:
            sw_albedo = np.full((1, ncol), dc.sw_albedo_override)
            sw_albedo_direct = None
        else:
            a = f.get("sw_albedo")
            sw_albedo = a.reshape(1, ncol).copy() if a.ndim == 1 else _colfast(a)
            sw_albedo_direct = None
            if f.exists("sw_albedo_direct"):
                a = f.get("sw_albedo_direct")
                sw_albedo_direct = a.reshape(1, ncol).copy() if a.ndim == 1 else _colfast(a)
        if dc.lw_emissivity_override >= 0.0:
            lw_emissivity = np.full((1, ncol), dc.lw_emissivity_override)
        else:
            a = f.get("lw_emissivity")
            lw_emissivity = a.reshape(1, ncol).copy() if a.ndim == 1 else _colfast(a)
        if dc.skin_temperature_override >= 0.0:
            skin_temperature[:] = dc.skin_temperature_override
        single_level = SingleLevel(cos_sza=cos_sza, skin_temperature=skin_temperature,
                                   sw_albedo=sw_albedo, lw_emissivity=lw_emissivity,
                                   sw_albedo_direct=sw_albedo_direct,
                                   solar_irradiance=solar_irradiance, iseed=iseed)
        # position in the solar cycle, +1 = maximum, -1 = minimum (ecrad_driver_read_input.F90:115-127)
        if dc.solar_cycle_multiplier_override > -1.0e6:
            single_level.spectral_solar_cycle_multiplier = float(dc.solar_cycle_multiplier_override)
        elif f.exists("spectral_solar_cycle_multiplier"):
            single_level.spectral_solar_cycle_multiplier = float(f.get_scalar("spectral_solar_cycle_multiplier"))

        aerosol = None
        if config.use_aerosols:
            # (column, type, level) -> Fortran (col, lev, type) == numpy (type, level, column)
            mr = np.ascontiguousarray(np.transpose(f.get("aerosol_mmr"), (1, 2, 0)))
            aerosol = Aerosol(mixing_ratio=mr, istartlev=1, iendlev=mr.shape[1])

        gas = Gas.allocate(ncol, nlev)
        for jgas in range(1, NMaxGases + 1):
            if jgas == IH2O:
                if f.exists("q"):
                    gas.put(IH2O, IMassMixingRatio, _colfast(f.get("q")))
                elif f.exists("h2o_mmr"):
                    gas.put(IH2O, IMassMixingRatio, _colfast(f.get("h2o_mmr")))
                else:
                    gas.put(IH2O, IVolumeMixingRatio, _colfast(f.get("h2o" + dc.vmr_suffix_str)))
            elif jgas == IO3:
                if f.exists("o3_mmr"):
                    gas.put(IO3, IMassMixingRatio, _colfast(f.get("o3_mmr")))
                else:
                    gas.put(IO3, IVolumeMixingRatio, _colfast(f.get("o3" + dc.vmr_suffix_str)))
            else:
                name = GAS_LOWER_CASE_NAMES[jgas - 1] + dc.vmr_suffix_str
                rank = f.rank(name)
                if rank == 0:
                    gas.put(jgas, IVolumeMixingRatio, f.get_scalar(name))
                elif rank == 2:
                    gas.put(jgas, IVolumeMixingRatio, _colfast(f.get(name)))
                elif rank > 0:
                    raise RuntimeError(f"{name} does not have 0 or 2 dimensions")
        for name, s in (dc.gas_scaling or {}).items():
            gas.scale(GAS_LOWER_CASE_NAMES.index(name) + 1, s)
    return ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol


# (netCDF name, flux member, second dimension)
_SAVE_SPEC = [
    ("flux_up_lw", "lw_up", "half_level"), ("flux_dn_lw", "lw_dn", "half_level"),
    ("flux_up_lw_clear", "lw_up_clear", "half_level"), ("flux_dn_lw_clear", "lw_dn_clear", "half_level"),
    ("lw_derivative", "lw_derivatives", "half_level"),
    ("spectral_flux_up_lw_toa", "lw_up_toa_band", "band_lw"),
    ("spectral_flux_up_lw_toa_clear", "lw_up_toa_clear_band", "band_lw"),
    ("canopy_flux_dn_lw_surf", "lw_dn_surf_canopy", "canopy_band_lw"),
    ("flux_up_sw", "sw_up", "half_level"), ("flux_dn_sw", "sw_dn", "half_level"),
    ("flux_dn_direct_sw", "sw_dn_direct", "half_level"),
    ("flux_up_sw_clear", "sw_up_clear", "half_level"), ("flux_dn_sw_clear", "sw_dn_clear", "half_level"),
    ("flux_dn_direct_sw_clear", "sw_dn_direct_clear", "half_level"),
    ("spectral_flux_dn_sw_surf", "sw_dn_surf_band", "band_sw"),
    ("spectral_flux_dn_direct_sw_surf", "sw_dn_direct_surf_band", "band_sw"),
    ("spectral_flux_dn_sw_surf_clear", "sw_dn_surf_clear_band", "band_sw"),
    ("spectral_flux_dn_direct_sw_surf_clear", "sw_dn_direct_surf_clear_band", "band_sw"),
    ("spectral_flux_dn_sw_toa", "sw_dn_toa_band", "band_sw"),
    ("spectral_flux_up_sw_toa", "sw_up_toa_band", "band_sw"),
    ("spectral_flux_up_sw_toa_clear", "sw_up_toa_clear_band", "band_sw"),
    ("canopy_flux_dn_diffuse_sw_surf", "sw_dn_diffuse_surf_canopy", "canopy_band_sw"),
    ("canopy_flux_dn_direct_sw_surf", "sw_dn_direct_surf_canopy", "canopy_band_sw"),
]


# spectral flux profiles (do_save_spectral_flux): dims (column, half_level, band), radiation_save.F90:185-203,
# :257-286, :359-366, :405-420
_SAVE_SPECTRAL = [
    ("spectral_flux_up_lw", "lw_up_band", "band_lw"), ("spectral_flux_dn_lw", "lw_dn_band", "band_lw"),
    ("spectral_flux_up_lw_clear", "lw_up_clear_band", "band_lw"), ("spectral_flux_dn_lw_clear", "lw_dn_clear_band", "band_lw"),
    ("spectral_flux_up_sw", "sw_up_band", "band_sw"), ("spectral_flux_dn_sw", "sw_dn_band", "band_sw"),
    ("spectral_flux_dn_direct_sw", "sw_dn_direct_band", "band_sw"),
    ("spectral_flux_up_sw_clear", "sw_up_clear_band", "band_sw"), ("spectral_flux_dn_sw_clear", "sw_dn_clear_band", "band_sw"),
    ("spectral_flux_dn_direct_sw_clear", "sw_dn_direct_clear_band", "band_sw"),
]


def flux_to_output_dict(config: Config, thermodynamics: Thermodynamics, flux: Flux) -> dict:
    """Arrays keyed by the reference's output variable names, each in netCDF order
    (column first), as save_fluxes writes them (radiation_save.F90:153-460)."""
    out = {"pressure_hl": thermodynamics.pressure_hl.T.copy()}
    for ncname, member, dim in _SAVE_SPEC:
        arr = getattr(flux, member)
        if arr is None:
            continue
        if dim == "half_level":
            out[ncname] = arr.T.copy()          # numpy (nlev+1, ncol) -> (column, half_level)
        else:
            out[ncname] = arr.copy()            # numpy (ncol, nband) already (column, band)
    if config.do_save_spectral_flux:
        for ncname, member, _ in _SAVE_SPECTRAL:
            arr = getattr(flux, member)
            if arr is not None:
                out[ncname] = np.ascontiguousarray(arr.transpose(1, 0, 2))    # (nlev+1, ncol, nspec) -> (column, half_level, band)
    for n in ("cloud_cover_lw", "cloud_cover_sw"):
        if getattr(flux, n) is not None and ((n.endswith("lw") and config.do_lw) or
                                             (n.endswith("sw") and config.do_sw)):
            if config.do_clouds:
                out[n] = getattr(flux, n).copy()
    return out


def _units(name: str) -> dict:
    """units attribute as radiation_save.F90:153-337 gives it"""
    if name == "pressure_hl":
        return {"units": "Pa"}
    if name.startswith("cloud_cover") or name == "lw_derivative":
        return {"units": "1"}
    return {"units": "W m-2"}


def save_fluxes(path: str, config: Config, thermodynamics: Thermodynamics, flux: Flux,
                is_double_precision: bool = False, is_hdf5_file: bool = False, experiment_name: str = "") -> None:
    """save_fluxes (radiation_save.F90:35-460); ``is_hdf5_file`` selects netCDF-4/HDF5 (ecrad_amd/hdf5file.py) instead
    of classic netCDF, as easy_netcdf's create(..., is_hdf5_file) does (utilities/easy_netcdf.F90:212-245)."""
    out = flux_to_output_dict(config, thermodynamics, flux)
    ncol = thermodynamics.pressure_hl.shape[1]
    dims = {"column": ncol, "half_level": thermodynamics.pressure_hl.shape[0]}
    variables = {}
    dim2 = {n: d for n, _, d in _SAVE_SPEC}
    dim3 = {n: d for n, _, d in _SAVE_SPECTRAL}
    for name, arr in out.items():
        if arr.ndim == 1:
            variables[name] = (("column",), arr, _units(name))
        elif arr.ndim == 3:
            dims.setdefault(dim3[name], arr.shape[2])
            variables[name] = (("column", "half_level", dim3[name]), arr, _units(name))
        else:
            d = "half_level" if name == "pressure_hl" else dim2[name]
            dims.setdefault(d, arr.shape[1])
            variables[name] = (("column", d), arr, _units(name))
    attrs = {"title": "Radiative flux profiles from the ecrad_amd MI355X radiation path", "source": "ecrad_amd"}
    if experiment_name.strip():
        attrs["experiment"] = experiment_name
    (write_nc4 if is_hdf5_file else write_nc)(path, dims, variables, attrs=attrs, double=is_double_precision)


def save_inputs(path: str, config: Config, single_level, thermodynamics, gas, cloud, aerosol, lat=None, lon=None) -> None:
    """save_inputs (radiation_save.F90:1026-1320): the input variables of radiation() in a file the offline driver can read
    back -- water vapour and ozone as mass mixing ratios ("q", "o3_mmr"), the other gases present as volume mixing ratios."""
    from .tables import GAS_LOWER_CASE_NAMES
    from .types import IMassMixingRatio, IVolumeMixingRatio
    T = lambda a: np.ascontiguousarray(np.asarray(a).T)            # numpy (x, column) -> netCDF (column, x)
    nhl, ncol = thermodynamics.pressure_hl.shape
    nlev = nhl - 1
    dims = {"column": ncol, "level": nlev, "half_level": nlev + 1}
    v = {}
    do_aerosol = config.use_aerosols and aerosol is not None
    dims["sw_albedo_band"] = single_level.sw_albedo.shape[0]
    dims["lw_emissivity_band"] = single_level.lw_emissivity.shape[0]
    v["solar_irradiance"] = ((), np.float64(single_level.solar_irradiance), {"units": "W m-2"})
    if lat is not None:
        v["lat"] = (("column",), np.asarray(lat, dtype=np.float64), {"units": "degrees_north"})
    if lon is not None:
        v["lon"] = (("column",), np.asarray(lon, dtype=np.float64), {"units": "degrees_east"})
    v["skin_temperature"] = (("column",), single_level.skin_temperature, {"units": "K"})
    if config.do_sw:
        v["cos_solar_zenith_angle"] = (("column",), single_level.cos_sza, {"units": "1"})
    v["sw_albedo"] = (("column", "sw_albedo_band"), T(single_level.sw_albedo), {"units": "1"})
    if single_level.sw_albedo_direct is not None:
        v["sw_albedo_direct"] = (("column", "sw_albedo_band"), T(single_level.sw_albedo_direct), {"units": "1"})
    v["lw_emissivity"] = (("column", "lw_emissivity_band"), T(single_level.lw_emissivity), {"units": "1"})
    if config.do_clouds and single_level.iseed is not None:
        v["iseed"] = (("column",), np.asarray(single_level.iseed, dtype=np.float64), {"units": "1"})      # (is_double in the reference)
    v["pressure_hl"] = (("column", "half_level"), T(thermodynamics.pressure_hl), {"units": "Pa"})
    v["temperature_hl"] = (("column", "half_level"), T(thermodynamics.temperature_hl), {"units": "K"})
    v["q"] = (("column", "level"), T(gas.get(IH2O, IMassMixingRatio)), {"units": "1"})
    v["o3_mmr"] = (("column", "level"), T(gas.get(IO3, IMassMixingRatio)), {"units": "1"})
    for jgas in range(1, NMaxGases + 1):
        if gas.is_present[jgas] and jgas not in (IH2O, IO3):
            v[GAS_LOWER_CASE_NAMES[jgas - 1] + "_vmr"] = (("column", "level"), T(gas.get(jgas, IVolumeMixingRatio)), {"units": "1"})
    if config.do_clouds and cloud is not None:
        v["cloud_fraction"] = (("column", "level"), T(cloud.fraction), {"units": "1"})
        v["q_liquid"] = (("column", "level"), T(cloud.mixing_ratio[0]), {"units": "1"})
        v["q_ice"] = (("column", "level"), T(cloud.mixing_ratio[1]), {"units": "1"})
        v["re_liquid"] = (("column", "level"), T(cloud.effective_radius[0]), {"units": "m"})
        v["re_ice"] = (("column", "level"), T(cloud.effective_radius[1]), {"units": "m"})
        if cloud.overlap_param is not None:
            dims["level_interface"] = nlev - 1
            v["overlap_param"] = (("column", "level_interface"), T(cloud.overlap_param), {"units": "1"})
        if cloud.fractional_std is not None:
            v["fractional_std"] = (("column", "level"), T(cloud.fractional_std), {"units": "1"})
        if getattr(cloud, "inv_cloud_effective_size", None) is not None:
            v["inv_cloud_effective_size"] = (("column", "level"), T(cloud.inv_cloud_effective_size), {"units": "m-1"})
        if getattr(cloud, "inv_inhom_effective_size", None) is not None:
            v["inv_inhom_effective_size"] = (("column", "level"), T(cloud.inv_inhom_effective_size), {"units": "m-1"})
    if do_aerosol:
        dims["aerosol_type"] = aerosol.mixing_ratio.shape[0]
        # numpy (type, level, column) -> netCDF (column, aerosol_type, level)
        v["aerosol_mmr"] = (("column", "aerosol_type", "level"), np.ascontiguousarray(np.transpose(aerosol.mixing_ratio, (2, 0, 1))),
                            {"units": "kg kg-1"})
    write_nc(path, dims, v, attrs={"title": "Input profiles to the ecrad_amd MI355X radiation path", "source": "ecrad_amd"}, double=True)


def save_radiative_properties(path: str, config: Config, nlev: int, istartcol: int, iendcol: int, single_level, thermodynamics,
                              cloud, props: dict) -> None:
    """save_radiative_properties (radiation_save.F90:716-1021): the intermediate arrays of radiation() for columns
    istartcol..iendcol -- variable and dimension names, the conditions on each variable and lw_emissivity = 1 - lw_albedo
    as in the reference.  ``props``: Radiation.optics(), arrays (column, level[+1], g-point or band)."""
    c0, c1 = istartcol - 1, iendcol
    cols = lambda a: np.ascontiguousarray(np.asarray(a)[..., c0:c1].T)        # (nlev, ncol) -> (column, level)
    dims = {"column": c1 - c0, "level": nlev, "half_level": nlev + 1}
    v = {}
    v["pressure_hl"] = (("column", "half_level"), cols(thermodynamics.pressure_hl), {"units": "Pa"})
    if thermodynamics.h2o_sat_liq is not None and config.use_aerosols:
        v["q_sat_liquid"] = (("column", "level"), cols(thermodynamics.h2o_sat_liq), {"units": "kg kg-1"})
    if config.do_sw:
        v["cos_solar_zenith_angle"] = (("column",), np.asarray(single_level.cos_sza)[c0:c1], {"units": "1"})
    if config.do_clouds:
        dims["level_interface"] = nlev - 1
        v["cloud_fraction"] = (("column", "level"), cols(cloud.fraction), {"units": "1"})
        v["overlap_param"] = (("column", "level_interface"), cols(cloud.overlap_param), {"units": "1"})
    if config.do_lw:
        dims["gpoint_lw"] = config.n_g_lw
        v["planck_hl"] = (("column", "half_level", "gpoint_lw"), props["planck_hl"], {"units": "W m-2"})
        v["lw_emission"] = (("column", "gpoint_lw"), props["lw_emission"], {"units": "W m-2"})
        v["lw_emissivity"] = (("column", "gpoint_lw"), 1.0 - props["lw_albedo"], {"units": "1"})
        v["od_lw"] = (("column", "level", "gpoint_lw"), props["od_lw"], {"units": "1"})
        if config.do_lw_aerosol_scattering:
            v["ssa_lw"] = (("column", "level", "gpoint_lw"), props["ssa_lw"], {"units": "1"})
            v["asymmetry_lw"] = (("column", "level", "gpoint_lw"), props["g_lw"], {"units": "1"})
        if config.do_clouds:
            dims["band_lw"] = config.n_bands_lw
            v["od_lw_cloud"] = (("column", "level", "band_lw"), props["od_lw_cloud"], {"units": "1"})
            if config.do_lw_cloud_scattering:
                v["ssa_lw_cloud"] = (("column", "level", "band_lw"), props["ssa_lw_cloud"], {"units": "1"})
                v["asymmetry_lw_cloud"] = (("column", "level", "band_lw"), props["g_lw_cloud"], {"units": "1"})
    if config.do_sw:
        dims["gpoint_sw"] = config.n_g_sw
        v["incoming_sw"] = (("column", "gpoint_sw"), props["incoming_sw"], {"units": "W m-2"})
        v["sw_albedo"] = (("column", "gpoint_sw"), props["sw_albedo_diffuse"], {"units": "1"})
        v["sw_albedo_direct"] = (("column", "gpoint_sw"), props["sw_albedo_direct"], {"units": "1"})
        for name, key in (("od_sw", "od_sw"), ("ssa_sw", "ssa_sw"), ("asymmetry_sw", "g_sw")):
            v[name] = (("column", "level", "gpoint_sw"), props[key], {"units": "1"})
        if config.do_clouds:
            dims["band_sw"] = config.n_bands_sw
            for name, key in (("od_sw_cloud", "od_sw_cloud"), ("ssa_sw_cloud", "ssa_sw_cloud"), ("asymmetry_sw_cloud", "g_sw_cloud")):
                v[name] = (("column", "level", "band_sw"), props[key], {"units": "1"})
    if config.do_clouds:
        if cloud.fractional_std is not None:
            v["fractional_std"] = (("column", "level"), cols(cloud.fractional_std), {"units": "1"})
        if getattr(cloud, "inv_cloud_effective_size", None) is not None:
            v["inv_cloud_effective_size"] = (("column", "level"), cols(cloud.inv_cloud_effective_size), {"units": "m-1"})
        if getattr(cloud, "inv_inhom_effective_size", None) is not None:
            v["inv_inhom_effective_size"] = (("column", "level"), cols(cloud.inv_inhom_effective_size), {"units": "m-1"})
    write_nc(path, dims, v, attrs={"title": "Radiative property profiles from the ecrad_amd MI355X radiation path", "source": "ecrad_amd"},
             double=True)


def save_net_fluxes(path: str, config: Config, thermodynamics: Thermodynamics, flux: Flux,
                    is_double_precision: bool = False, experiment_name: str = "", is_hdf5_file: bool = False) -> None:
    """save_net_fluxes (radiation_save.F90:464-715): net (down minus up) flux profiles and the surface / TOA downwelling
    fluxes instead of the separate up and down profiles; same variable names and dimensions as the reference."""
    ncol = thermodynamics.pressure_hl.shape[1]
    nhl = thermodynamics.pressure_hl.shape[0]
    dims = {"column": ncol, "half_level": nhl}
    v = {"pressure_hl": (("column", "half_level"), thermodynamics.pressure_hl.T.copy())}
    prof = lambda a: (("column", "half_level"), np.ascontiguousarray(a.T))
    col = lambda a: (("column",), np.ascontiguousarray(a))
    if config.do_lw:
        v["flux_net_lw"] = prof(flux.lw_dn - flux.lw_up)
        v["flux_dn_lw_surf"] = col(flux.lw_dn[nhl - 1])
        if config.do_clear:
            v["flux_net_lw_clear"] = prof(flux.lw_dn_clear - flux.lw_up_clear)
            v["flux_dn_lw_clear_surf"] = col(flux.lw_dn_clear[nhl - 1])
        if config.do_lw_derivatives:
            v["lw_derivative"] = prof(flux.lw_derivatives)
        if config.do_canopy_fluxes_lw:
            dims["canopy_band_lw"] = flux.lw_dn_surf_canopy.shape[1]
            v["canopy_flux_dn_lw_surf"] = (("column", "canopy_band_lw"), flux.lw_dn_surf_canopy.copy())
    if config.do_sw:
        v["flux_net_sw"] = prof(flux.sw_dn - flux.sw_up)
        v["flux_dn_sw_surf"] = col(flux.sw_dn[nhl - 1])
        v["flux_dn_sw_toa"] = col(flux.sw_dn[0])
        if config.do_sw_direct:
            v["flux_dn_direct_sw_surf"] = col(flux.sw_dn_direct[nhl - 1])
        if config.do_clear:
            v["flux_net_sw_clear"] = prof(flux.sw_dn_clear - flux.sw_up_clear)
            v["flux_dn_sw_clear_surf"] = col(flux.sw_dn_clear[nhl - 1])
            if config.do_sw_direct:
                v["flux_dn_direct_sw_clear_surf"] = col(flux.sw_dn_direct_clear[nhl - 1])
        if config.do_canopy_fluxes_sw:
            dims["canopy_band_sw"] = flux.sw_dn_diffuse_surf_canopy.shape[1]
            v["canopy_flux_dn_diffuse_sw_surf"] = (("column", "canopy_band_sw"), flux.sw_dn_diffuse_surf_canopy.copy())
            v["canopy_flux_dn_direct_sw_surf"] = (("column", "canopy_band_sw"), flux.sw_dn_direct_surf_canopy.copy())
    attrs = {"title": "Radiative flux profiles from the ecrad_amd MI355X radiation path", "source": "ecrad_amd"}
    if experiment_name.strip():
        attrs["experiment"] = experiment_name
    (write_nc4 if is_hdf5_file else write_nc)(path, dims, v, attrs=attrs, double=is_double_precision)


def get_sw_mapping(config: Config, wavelength_bound) -> np.ndarray:
    """config%get_sw_mapping (radiation_config.F90:1766-1815): matrix (ninterval, nband) that turns the shortwave
    fluxes of the bands (or g-points) into fluxes of user-specified wavelength intervals; the two extra rows of
    calc_mapping_from_bands for wavelengths below the first and above the last bound are dropped."""
    if config.n_bands_sw <= 0:
        raise RuntimeError("get_sw_mapping called before number of shortwave bands set")
    wb = np.asarray(wavelength_bound, dtype=np.float64)
    ninterval = wb.size - 1
    m = config.gas_optics_sw.spectral_def.calc_mapping_from_bands(
        wb, np.arange(1, ninterval + 3), use_bands=not config.do_cloud_aerosol_per_sw_g_point)    # (nband, ninterval+2)
    return np.ascontiguousarray(m[:, 1:ninterval + 1].T)


def save_sw_diagnostics(path: str, config: Config, wavelength_bound, mapping: np.ndarray, flux: Flux,
                        is_double_precision: bool = False, experiment_name: str = "", is_hdf5_file: bool = False) -> None:
    """save_sw_diagnostics (radiation_save.F90:1314-1470): surface (and, with do_save_spectral_flux, TOA) shortwave
    fluxes in user-specified wavelength intervals."""
    wb = np.asarray(wavelength_bound, dtype=np.float64)
    nwav = wb.size - 1
    ncol = flux.sw_dn_surf_band.shape[0]
    dims = {"column": ncol, "wavelength": nwav}
    m = np.asarray(mapping)                                    # (nwav, nband)
    to_wav = lambda band_flux: (("column", "wavelength"), np.ascontiguousarray(band_flux @ m.T))
    v = {"wavelength1": (("wavelength",), wb[:nwav].copy()), "wavelength2": (("wavelength",), wb[1:nwav + 1].copy()),
         "flux_dn_sw_surf": to_wav(flux.sw_dn_surf_band), "flux_dn_direct_sw_surf": to_wav(flux.sw_dn_direct_surf_band)}
    if config.do_clear:
        v["flux_dn_sw_surf_clear"] = to_wav(flux.sw_dn_surf_clear_band)
        v["flux_dn_direct_sw_surf_clear"] = to_wav(flux.sw_dn_direct_surf_clear_band)
    if flux.sw_up_band is not None:                            # numpy (nlev+1, ncol, nspec)
        v["flux_up_sw_surf"] = to_wav(flux.sw_up_band[-1])
        v["flux_up_sw_toa"] = to_wav(flux.sw_up_band[0])
        v["flux_dn_sw_toa"] = to_wav(flux.sw_dn_band[0])
        if flux.sw_up_clear_band is not None:
            v["flux_up_sw_toa_clear"] = to_wav(flux.sw_up_clear_band[0])
            v["flux_up_sw_surf_clear"] = to_wav(flux.sw_up_clear_band[-1])
    attrs = {"title": "Shortwave spectral diagnostics from the ecrad_amd MI355X radiation path", "source": "ecrad_amd"}
    if experiment_name.strip():
        attrs["experiment"] = experiment_name
    (write_nc4 if is_hdf5_file else write_nc)(path, dims, v, attrs=attrs, double=is_double_precision)


def main(argv=None) -> int:
    from .interface import Radiation
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 3:
        print("Usage: python -m ecrad_amd.driver config.nam input_file.nc output_file.nc")
        return 1
    config = Config.read(argv[0])
    dc = DriverConfig.read(argv[0])
    rad = Radiation(config)                      # setup_radiation
    # shortwave diagnostics in user-specified intervals (driver/ecrad_driver.F90:211-222)
    bounds = [float(b) for b in (dc.sw_diag_wavelength_bound or [])]
    n_sw_diag = 0
    for j, b in enumerate(bounds + [-1.0]):
        if b < 0.0:
            n_sw_diag = max(0, j - 1)
            break
    sw_diag_mapping = None
    if n_sw_diag > 0:
        if not config.do_surface_sw_spectral_flux:
            raise SystemExit("Error: shortwave spectral diagnostics require do_surface_sw_spectral_flux=true")
        sw_diag_mapping = get_sw_mapping(config, bounds[:n_sw_diag + 1])
    ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol = read_input(argv[1], config, dc)
    iend = dc.iendcol if 1 <= dc.iendcol <= ncol else ncol
    istart = max(dc.istartcol, 1)
    if dc.do_save_aerosol_optics and config.use_aerosols and config.aerosol_optics is not None:      # driver/ecrad_driver.F90:224-226
        config.aerosol_optics.save("aerosol_optics.nc")
    if dc.do_save_cloud_optics and config.do_clouds and config.use_general_cloud_optics:          # driver/ecrad_driver.F90:228-230
        for tag, tables in (("sw", config.cloud_optics_sw if config.do_sw else []), ("lw", config.cloud_optics_lw if config.do_lw else [])):
            for co in tables or []:
                co.save(f"hydrometeor_optics_{tag}_{co.type_name}.nc")
    if dc.do_save_inputs:                        # driver/ecrad_driver.F90:283-289 (before set_gas_units, lat = lon = 0)
        save_inputs("inputs.nc", config, single_level, thermodynamics, gas, cloud, aerosol, lat=np.zeros(ncol), lon=np.zeros(ncol))
    rad.set_gas_units(gas)
    thermodynamics.calc_saturation_wrt_liquid()
    out_of_physical_bounds(istart, iend, dc.do_correct_unphysical_inputs, single_level, thermodynamics, gas, cloud, aerosol)
    flux = Flux.allocate(config, ncol, nlev)
    t0 = time.perf_counter()
    for _ in range(max(dc.nrepeat, 1)):
        rad.radiation(ncol, nlev, istart, iend, single_level, thermodynamics, gas, cloud, aerosol, flux)
    print(f"Time elapsed in radiative transfer: {time.perf_counter() - t0:12.5g} seconds")
    if not dc.do_save_net_fluxes:                # driver/ecrad_driver.F90:398-417
        save_fluxes(argv[2], config, thermodynamics, flux, is_double_precision=dc.do_write_double_precision,
                    is_hdf5_file=dc.do_write_hdf5, experiment_name=dc.experiment_name)
    else:
        save_net_fluxes(argv[2], config, thermodynamics, flux, is_double_precision=dc.do_write_double_precision,
                        experiment_name=dc.experiment_name, is_hdf5_file=dc.do_write_hdf5)
    if n_sw_diag > 0:
        save_sw_diagnostics(dc.sw_diag_file_name, config, bounds[:n_sw_diag + 1], sw_diag_mapping, flux,
                            is_double_precision=dc.do_write_double_precision, experiment_name=dc.experiment_name,
                            is_hdf5_file=dc.do_write_hdf5)
    return 0


if __name__ == "__main__":
    sys.exit(main())
