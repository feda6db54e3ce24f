"""Host mirror of ``config_type`` (radiation/radiation_config.F90:163-649).

Only members that influence the hot path are kept; names, defaults and namelist spellings are the
reference's (radiation_config.F90:163-512 defaults, :730-764 namelist, :1020-1061 enum decoding,
:1106-1362 consolidate) so that test/ifs/config*.nam files can be read unchanged.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .namelist import read_namelist

# Enumerations (radiation_config.F90:53-126, radiation_cloud_cover.F90:32-38)
SOLVER_NAMES = ["Cloudless", "Homogeneous", "McICA", "SPARTACUS", "Tripleclouds"]
ISolverCloudless, ISolverHomogeneous, ISolverMcICA, ISolverSpartacus, ISolverTripleclouds = range(5)
GAS_MODEL_NAMES = ["Monochromatic", "RRTMG-IFS", "ECCKD"]
IGasModelMonochromatic, IGasModelIFSRRTMG, IGasModelECCKD = range(3)
OVERLAP_NAMES = ["Max-Ran", "Exp-Ran", "Exp-Exp"]
IOverlapMaximumRandom, IOverlapExponentialRandom, IOverlapExponential = range(3)
LIQUID_MODEL_NAMES = ["Monochromatic", "SOCRATES", "Slingo", "Jahangir", "Nielsen"]     # radiation_config.F90:109-116
ILiquidModelMonochromatic, ILiquidModelSOCRATES, ILiquidModelSlingo, ILiquidModelJahangir, ILiquidModelNielsen = range(5)
ICE_MODEL_NAMES = ["Monochromatic", "Fu-IFS", "Baran-EXPERIMENTAL", "Baran2016", "Baran2017", "Yi"]   # :124-133
IIceModelMonochromatic, IIceModelFu, IIceModelBaran, IIceModelBaran2016, IIceModelBaran2017, IIceModelYi = range(6)
ENTRAPMENT_NAMES = ["Zero", "Edge-only", "Explicit", "Non-fractal", "Maximum"]        # :72-86
ENCROACHMENT_NAMES = ["Zero", "Minimum", "Fractal", "Computed", "Maximum"]             # :90-94 (deprecated spelling)
(IEntrapmentZero, IEntrapmentEdgeOnly, IEntrapmentExplicit, IEntrapmentExplicitNonFractal,
 IEntrapmentMaximum) = range(5)
IPrecisionDouble, IPrecisionSingle = range(2)
PDF_SHAPE_NAMES = ["Lognormal", "Gamma"]
IPdfShapeLognormal, IPdfShapeGamma = range(2)

NMaxAerosolTypes = 256
NMaxCloudTypes = 12
NMaxAlbedoIntervals = 256


class ConfigError(RuntimeError):
    """Raised where the reference calls radiation_abort('Radiation configuration error')."""


def _enum(value: str, names, var_name: str) -> int:
    """get_enum_code (radiation_config.F90:2103-2135): exact match, fatal otherwise."""
    for i, n in enumerate(names):
        if value.strip() == n:
            return i
    raise ConfigError(f'{var_name} must be one of: ' + ", ".join(f'"{n}"' for n in names)
                      + f' (got "{value}")')


@dataclass
class Config:
    # --- user switches, defaults as radiation_config.F90:163-512 --------------------------------
    directory_name: str = "."
    use_general_cloud_optics: bool = True
    use_general_aerosol_optics: bool = True
    cloud_fraction_threshold: float = 1.0e-6
    cloud_mixing_ratio_threshold: float = 1.0e-9
    i_overlap_scheme: int = IOverlapExponentialRandom
    use_beta_overlap: bool = False
    use_vectorizable_generator: bool = False
    i_cloud_pdf_shape: int = IPdfShapeGamma
    cloud_inhom_decorr_scaling: float = 0.5
    max_cloud_od: float = 16.0
    # SPARTACUS (radiation_config.F90:226-260, 268, 341-411)
    nregions: int = 3
    do_3d_effects: bool = True
    i_3d_sw_entrapment: int = IEntrapmentExplicit
    do_3d_lw_multilayer_effects: bool = False
    do_lw_side_emissivity: bool = True
    max_3d_transfer_rate: float = 10.0
    max_gas_od_3d: float = 8.0
    min_cloud_effective_size: float = 100.0
    overhang_factor: float = 0.0
    clear_to_thick_fraction: float = 0.0
    overhead_sun_factor: float = 0.0
    use_expm_everywhere: bool = False
    # Working precision of the SPARTACUS solver kernels (not a namelist entry of the reference, where it is the
    # compile-time choice PARKIND1_SINGLE): IPrecisionDouble or IPrecisionSingle
    i_precision: int = IPrecisionDouble
    do_lw_cloud_scattering: bool = True
    do_lw_aerosol_scattering: bool = True
    i_solver_sw: int = ISolverMcICA
    i_solver_lw: int = ISolverMcICA
    do_sw_delta_scaling_with_gases: bool = False
    i_gas_model_sw: int = IGasModelIFSRRTMG
    i_gas_model_lw: int = IGasModelIFSRRTMG
    i_liq_model: int = ILiquidModelSOCRATES            # radiation_config.F90:307
    i_ice_model: int = IIceModelBaran                  # :308
    do_fu_lw_ice_optics_bug: bool = False
    min_gas_od_lw: float = 1.0e-15                     # :244-245
    min_gas_od_sw: float = 0.0
    liq_optics_override_file_name: str = ""
    ice_optics_override_file_name: str = ""
    liq_optics_file_name: str = ""
    ice_optics_file_name: str = ""
    rrtmg: object = None                               # RrtmgTables (ecrad_amd/rrtmg.py)
    do_nearest_spectral_sw_albedo: bool = False
    do_nearest_spectral_lw_emiss: bool = False
    sw_albedo_wavelength_bound: List[float] = field(default_factory=list)
    lw_emiss_wavelength_bound: List[float] = field(default_factory=list)
    i_sw_albedo_index: List[int] = field(default_factory=list)
    i_lw_emiss_index: List[int] = field(default_factory=list)
    do_lw: bool = True
    do_sw: bool = True
    do_clear: bool = True
    do_sw_direct: bool = True
    cloud_type_name: List[str] = field(default_factory=list)
    use_thick_cloud_spectral_averaging: List[bool] = field(default_factory=lambda: [True] * NMaxCloudTypes)
    use_aerosols: bool = False
    n_aerosol_types: int = 0
    i_aerosol_type_map: List[int] = field(default_factory=list)
    do_save_spectral_flux: bool = False
    do_save_gpoint_flux: bool = False
    do_save_radiative_properties: bool = False      # radiation() dumps the stage-interface arrays (radiation_interface.F90:403-419)
    n_spec_sw: int = 0
    n_spec_lw: int = 0
    i_spec_from_reordered_g_sw: object = None
    i_spec_from_reordered_g_lw: object = None
    do_surface_sw_spectral_flux: bool = True
    do_toa_spectral_flux: bool = False
    do_lw_derivatives: bool = False
    do_canopy_fluxes_sw: bool = False
    do_canopy_fluxes_lw: bool = False
    use_canopy_full_spectrum_sw: bool = False
    use_canopy_full_spectrum_lw: bool = False
    aerosol_optics_override_file_name: str = ""
    gas_optics_sw_override_file_name: str = ""
    gas_optics_lw_override_file_name: str = ""
    cloud_pdf_override_file_name: str = ""
    do_cloud_aerosol_per_sw_g_point: bool = True
    do_cloud_aerosol_per_lw_g_point: bool = True
    do_weighted_surface_mapping: bool = True
    use_spectral_solar_cycle: bool = False
    use_spectral_solar_scaling: bool = False     # single_level%spectral_solar_scaling scales the RRTMG shortwave bands (radiation_config.F90:169)
    use_updated_solar_spectrum: bool = False
    ssi_override_file_name: str = ""
    ssi_file_name: str = ""
    iverbose: int = 1
    iverbosesetup: int = 2

    # --- derived by consolidate()/setup_radiation() ---------------------------------------------
    is_consolidated: bool = False
    do_clouds: bool = True
    is_homogeneous: bool = False
    gas_optics_sw_file_name: str = ""
    gas_optics_lw_file_name: str = ""
    aerosol_optics_file_name: str = ""
    cloud_pdf_file_name: str = ""
    n_g_sw: int = 0
    n_g_lw: int = 0
    n_bands_sw: int = 0
    n_bands_lw: int = 0
    n_g_lw_if_scattering: int = 0
    n_bands_lw_if_scattering: int = 0
    n_canopy_bands_sw: int = 1
    n_canopy_bands_lw: int = 1
    n_cloud_types: int = 0
    # tables (filled by interface.setup_radiation)
    gas_optics_sw: object = None
    gas_optics_lw: object = None
    cloud_optics_sw: list = field(default_factory=list)
    cloud_optics_lw: list = field(default_factory=list)
    aerosol_optics: object = None
    pdf_sampler: object = None
    i_band_from_reordered_g_sw: Optional[np.ndarray] = None
    i_band_from_reordered_g_lw: Optional[np.ndarray] = None
    sw_albedo_weights: Optional[np.ndarray] = None    # numpy (n_bands_sw, nalb) == Fortran (nalb, n_bands_sw)
    lw_emiss_weights: Optional[np.ndarray] = None
    i_albedo_from_band_sw: Optional[np.ndarray] = None
    i_emiss_from_band_lw: Optional[np.ndarray] = None

    # ---------------------------------------------------------------------------------------------
    @classmethod
    def read(cls, file_name: str) -> "Config":
        """config%read (radiation_config.F90:664-1104): the ``&radiation`` group."""
        return cls().read_into(file_name)

    def read_into(self, file_name: str) -> "Config":
        """config%read on an object that already holds settings (the reference's read starts from the current
        values, radiation_config.F90:787-880: ifs/radiation_setup.F90:504-506 relies on that to let a namelist
        override what the host model chose); entries absent from the namelist keep their values."""
        nml = read_namelist(file_name).get("radiation", {})
        c = self
        simple = [
            "do_sw", "do_lw", "do_sw_direct", "do_clear", "do_save_spectral_flux", "do_save_gpoint_flux", "do_save_radiative_properties",
            "do_surface_sw_spectral_flux", "do_lw_derivatives", "do_toa_spectral_flux",
            "do_lw_aerosol_scattering", "do_lw_cloud_scattering", "directory_name",
            "aerosol_optics_override_file_name", "cloud_pdf_override_file_name",
            "gas_optics_sw_override_file_name", "gas_optics_lw_override_file_name",
            "use_canopy_full_spectrum_sw", "use_canopy_full_spectrum_lw", "do_canopy_fluxes_sw",
            "do_canopy_fluxes_lw", "use_general_cloud_optics", "use_general_aerosol_optics",
            "do_sw_delta_scaling_with_gases", "use_beta_overlap", "use_vectorizable_generator",
            "iverbose", "iverbosesetup", "cloud_inhom_decorr_scaling", "cloud_fraction_threshold",
            "max_cloud_od", "cloud_mixing_ratio_threshold", "n_aerosol_types", "use_aerosols",
            "do_nearest_spectral_sw_albedo", "do_nearest_spectral_lw_emiss",
            "do_cloud_aerosol_per_lw_g_point", "do_cloud_aerosol_per_sw_g_point",
            "do_weighted_surface_mapping", "use_spectral_solar_cycle", "use_spectral_solar_scaling", "use_updated_solar_spectrum", "ssi_override_file_name", "do_fu_lw_ice_optics_bug",
            "min_gas_od_lw", "min_gas_od_sw", "liq_optics_override_file_name", "ice_optics_override_file_name",
            "do_3d_effects", "do_3d_lw_multilayer_effects", "do_lw_side_emissivity", "max_3d_transfer_rate",
            "max_gas_od_3d", "min_cloud_effective_size", "overhang_factor", "clear_to_thick_fraction",
            "overhead_sun_factor", "use_expm_everywhere",
        ]
        for k in simple:
            if k in nml and nml[k] is not None:
                cur = getattr(c, k)
                v = nml[k]
                if isinstance(cur, float):
                    v = float(v)
                setattr(c, k, v)

        def as_list(v):
            if v is None:
                return []
            return [x for x in (v if isinstance(v, list) else [v]) if x is not None]

        for k in ("sw_albedo_wavelength_bound", "lw_emiss_wavelength_bound"):
            if k in nml:
                setattr(c, k, [float(x) for x in as_list(nml[k])])
        for k in ("i_sw_albedo_index", "i_lw_emiss_index", "i_aerosol_type_map"):
            if k in nml:
                setattr(c, k, [int(x) for x in as_list(nml[k])])
        if "cloud_type_name" in nml:
            c.cloud_type_name = [str(x) for x in as_list(nml["cloud_type_name"])]
        if "use_thick_cloud_spectral_averaging" in nml:
            v = as_list(nml["use_thick_cloud_spectral_averaging"])
            c.use_thick_cloud_spectral_averaging[:len(v)] = [bool(x) for x in v]

        # Enumerations by name (radiation_config.F90:1020-1061)
        if nml.get("sw_solver_name"):
            c.i_solver_sw = _enum(nml["sw_solver_name"], SOLVER_NAMES, "sw_solver_name")
        if nml.get("lw_solver_name"):
            c.i_solver_lw = _enum(nml["lw_solver_name"], SOLVER_NAMES, "lw_solver_name")
        if nml.get("gas_model_name"):
            g = _enum(nml["gas_model_name"], GAS_MODEL_NAMES, "gas_model_name")
            c.i_gas_model_sw = c.i_gas_model_lw = g
        if nml.get("liquid_model_name"):
            c.i_liq_model = _enum(nml["liquid_model_name"], LIQUID_MODEL_NAMES, "liquid_model_name")
        if nml.get("ice_model_name"):
            c.i_ice_model = _enum(nml["ice_model_name"], ICE_MODEL_NAMES, "ice_model_name")
        if nml.get("sw_gas_model_name"):
            c.i_gas_model_sw = _enum(nml["sw_gas_model_name"], GAS_MODEL_NAMES, "sw_gas_model_name")
        if nml.get("lw_gas_model_name"):
            c.i_gas_model_lw = _enum(nml["lw_gas_model_name"], GAS_MODEL_NAMES, "lw_gas_model_name")
        if nml.get("overlap_scheme_name"):
            c.i_overlap_scheme = _enum(nml["overlap_scheme_name"], OVERLAP_NAMES, "overlap_scheme_name")
        if nml.get("n_regions") is not None:
            c.nregions = int(nml["n_regions"])
        # radiation_config.F90:1046-1054 (sw_encroachment_name is the deprecated spelling; encroachment_scaling of overhang_factor)
        if nml.get("sw_encroachment_name"):
            c.i_3d_sw_entrapment = _enum(nml["sw_encroachment_name"], ENCROACHMENT_NAMES, "sw_encroachment_name")
        elif nml.get("sw_entrapment_name"):
            c.i_3d_sw_entrapment = _enum(nml["sw_entrapment_name"], ENTRAPMENT_NAMES, "sw_entrapment_name")
        if nml.get("encroachment_scaling") is not None and float(nml["encroachment_scaling"]) >= 0.0:
            c.overhang_factor = float(nml["encroachment_scaling"])
        c.min_cloud_effective_size = max(1.0e-6, c.min_cloud_effective_size)          # :970
        if nml.get("cloud_pdf_shape_name"):
            c.i_cloud_pdf_shape = _enum(nml["cloud_pdf_shape_name"], PDF_SHAPE_NAMES, "cloud_pdf_shape_name")
        if c.do_save_gpoint_flux:
            c.do_save_spectral_flux = True
        if c.iverbose < 0:
            c.iverbose = 0
        if c.iverbosesetup < 0:
            c.iverbosesetup = 0
        return c

    # ---------------------------------------------------------------------------------------------
    def _data_path(self, override: str, default: str) -> str:
        if override:
            return override if override.startswith("/") else os.path.join(self.directory_name, override)
        return os.path.join(self.directory_name, default)

    def consolidate(self) -> None:
        """consolidate_config (radiation_config.F90:1106-1362)."""
        if self.do_canopy_fluxes_sw and not self.do_surface_sw_spectral_flux:
            self.do_surface_sw_spectral_flux = True
        self.do_clouds = bool((self.do_sw and self.i_solver_sw != ISolverCloudless)
                              or (self.do_lw and self.i_solver_lw != ISolverCloudless))
        uses_regions = (self.i_solver_sw in (ISolverSpartacus, ISolverTripleclouds)
                        or self.i_solver_lw in (ISolverSpartacus, ISolverTripleclouds))
        if uses_regions and self.i_overlap_scheme != IOverlapExponentialRandom:
            raise ConfigError("SPARTACUS/Tripleclouds solvers can only do Exponential-Random overlap")
        if self.i_gas_model_sw == IGasModelECCKD:
            self.gas_optics_sw_file_name = self._data_path(
                self.gas_optics_sw_override_file_name, "ecckd-1.4_sw_climate_rgb-32b_ckd-definition.nc")
        if self.i_gas_model_lw == IGasModelECCKD:
            self.gas_optics_lw_file_name = self._data_path(
                self.gas_optics_lw_override_file_name, "ecckd-1.0_lw_climate_fsck-32b_ckd-definition.nc")
        # radiation_config.F90:1243-1290: optics files of the band-fit cloud schemes
        if not self.use_general_cloud_optics:
            # (the reference's cloud_optics has no branch for the Jahangir and Nielsen liquid models either,
            #  radiation_cloud_optics.F90:325-347, and ships no coefficient files for them)
            liq = {ILiquidModelSOCRATES: "socrates_droplet_scattering_rrtm.nc", ILiquidModelSlingo: "slingo_droplet_scattering_rrtm.nc"}
            ice = {IIceModelFu: "fu_ice_scattering_rrtm.nc", IIceModelBaran: "baran_ice_scattering_rrtm.nc",
                   IIceModelBaran2016: "baran2016_ice_scattering_rrtm.nc", IIceModelBaran2017: "baran2017_ice_scattering_rrtm.nc",
                   IIceModelYi: "yi_ice_scattering_rrtm.nc"}
            if self.i_liq_model not in liq:
                raise ConfigError(f"band cloud optics: liquid_model_name='{LIQUID_MODEL_NAMES[self.i_liq_model]}' is not implemented")
            if self.i_ice_model not in ice:
                raise ConfigError(f"band cloud optics: ice_model_name='{ICE_MODEL_NAMES[self.i_ice_model]}' is not implemented")
            self.liq_optics_file_name = self._data_path(self.liq_optics_override_file_name, liq[self.i_liq_model])
            self.ice_optics_file_name = self._data_path(self.ice_optics_override_file_name, ice[self.i_ice_model])
        if self.use_spectral_solar_cycle:                 # radiation_config.F90:1200-1218
            if IGasModelECCKD != self.i_gas_model_sw:
                raise ConfigError("solar cycle only available with ecCKD gas optics model")
            self.ssi_file_name = self._data_path(self.ssi_override_file_name, "ssi_nrl2.nc")
        self.aerosol_optics_file_name = self._data_path(
            self.aerosol_optics_override_file_name,
            "aerosol_ifs_49R1_20230119.nc" if self.use_general_aerosol_optics
            else "aerosol_ifs_rrtm_46R1_with_NI_AM.nc")
        self.cloud_pdf_file_name = self._data_path(
            self.cloud_pdf_override_file_name,
            "mcica_lognormal.nc" if self.i_cloud_pdf_shape == IPdfShapeLognormal else "mcica_gamma.nc")
        if self.n_aerosol_types < 0 or self.n_aerosol_types > NMaxAerosolTypes:
            raise ConfigError("number of aerosol types out of range")
        spartacus = ((self.do_sw and self.i_solver_sw == ISolverSpartacus) or (self.do_lw and self.i_solver_lw == ISolverSpartacus))
        if spartacus and self.nregions not in (2, 3):
            raise ConfigError("SPARTACUS: n_regions must be 2 or 3")      # radiation_config.F90:268
        if self.do_sw and self.i_solver_sw == ISolverSpartacus and self.do_sw_delta_scaling_with_gases:
            raise ConfigError("SW delta-Eddington scaling with gases not possible with SPARTACUS solver")     # :1336-1340
        if self.i_solver_sw == ISolverMcICA:
            self.do_save_spectral_flux = False
        if self.do_lw and self.do_sw and ((self.i_solver_sw == ISolverHomogeneous)
                                          != (self.i_solver_lw == ISolverHomogeneous)):
            raise ConfigError("if one solver is Homogeneous then the other must be")
        self.is_homogeneous = bool((self.do_sw and self.i_solver_sw == ISolverHomogeneous)
                                   or (self.do_lw and self.i_solver_lw == ISolverHomogeneous))
        self.is_consolidated = True
