"""The IFS-style caller of the path: ``setup_radiation_scheme`` / ``radiation_scheme`` and NPROMA blocking.

Host-side counterpart of the reference's second caller of ``radiation()``:
  ifs/radiation_setup.F90:58-562    SETUP_RADIATION_SCHEME (host-model switches -> config_type, UV/PAR weights)
  ifs/radiation_scheme.F90:1-675    RADIATION_SCHEME (IFS arrays -> ecRad types -> radiation() -> net fluxes and
                                    the surface diagnostics the rest of the model needs)
  ifs/liquid_effective_radius.F90, ifs/ice_effective_radius.F90, ifs/cloud_overlap_decorr_len.F90
  radiation/radiation_cloud.F90:195-385 (set_overlap_param), :470-500 (create_fractional_std), :602-690
  driver/ifs_blocking.F90           the NPROMA-blocked field array ZRGP(NPROMA, NFIELDS, NGPBLKS)
  driver/ecrad_ifs_driver.F90, driver/ecrad_ifs_driver_blocked.F90   the offline drivers around them

Array convention as everywhere in this package (ecrad_amd/types.py): the numpy shape is the Fortran shape reversed, so
``PQ(KLON,KLEV)`` is ``(klev, klon)`` with the column fastest in memory.

What is different on an MI355X.  NPROMA blocks exist to fit a CPU cache: the reference loops over blocks of 8-80 columns
and calls ``radiation_scheme`` once per block.  The GPU wants the opposite -- one call covering every column the rank
owns (10^5-10^6) -- so ``radiation_scheme`` here takes any ``kidia:kfdia`` range (the whole array by default) and
``radiation_scheme_blocked`` views the blocked array ZRGP as ONE batch of ``nproma * ngpblks`` columns (a reshape of the
field axis, no per-block loop), runs the path once and scatters the outputs back into the blocks.  The block-by-block
loop of the reference is available too (``per_block=True``) and gives identical numbers: columns are independent.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .config import (Config, ConfigError, IGasModelECCKD, IGasModelIFSRRTMG, IIceModelBaran, IIceModelFu, IIceModelYi,
                     ILiquidModelSlingo, ILiquidModelSOCRATES, IOverlapExponential, IOverlapExponentialRandom,
                     IOverlapMaximumRandom, ISolverCloudless, ISolverMcICA, ISolverSpartacus, ISolverTripleclouds)
from .types import (Aerosol, Cloud, Flux, Gas, IMassMixingRatio, IVolumeMixingRatio, SingleLevel, Thermodynamics)
from .tables import (IH2O, ICO2, IO3, IN2O, ICH4, IO2, ICFC11, ICFC12, IHCFC22, ICCl4, INO2)

# ifsaux/yomcst_ecrad.F90
RPI = 3.14159265358979323846
RSIGMA = 5.67037321e-8
RD = 287.058
RTT = 273.16
# ifs/yoerdu.F90:29,35; ifs/yoecld.F90:22-23
REPLOG = 1.0e-12
REPSCW = 1.0e-12
RDECORR_CF = 2.0
RDECORR_CW = 1.0
# radiation/radiation_constants.F90:26,30
ACCEL_DUE_TO_GRAVITY = 9.80665
GAS_CONSTANT_DRY_AIR = 287.058

ITYPE_TROP_BG_AER = 8       # ifs/radiation_setup.F90:45-46
ITYPE_STRAT_BG_AER = 12


@dataclass
class TERAD:
    """TERAD (ifs/yoerad.F90:20-70): the host model's radiation switches, with the reference's defaults."""
    NSW: int = 6
    NLWEMISS: int = 2
    NICEOPT: int = 3
    NLIQOPT: int = 4
    NRADIP: int = 3
    NRADLP: int = 2
    NLWOUT: int = 1
    NDECOLAT: int = 2
    NMINICE: int = 1
    NAERMACC: int = 1
    NMCVAR: int = 12
    NLWSCATTERING: int = 1
    NSWSOLVER: int = 3
    NLWSOLVER: int = 3
    NSOLARSPECTRUM: int = 0
    NDUMPBADINPUTS: int = 0
    NDUMPINPUTS: int = 0
    NCLOUDOVERLAP: int = 3
    RCLOUD_FRAC_STD: float = 1.0
    RCLOUD_SEPARATION_SCALE_TOA: float = 14000.0
    RCLOUD_SEPARATION_SCALE_SURF: float = 2500.0
    LFU_LW_ICE_OPTICS_BUG: bool = False
    LDIAGFORCING: bool = False
    LAPPROXLWUPDATE: bool = True
    LAPPROXSWUPDATE: bool = False
    LCCNL: bool = True
    LCCNO: bool = True
    RCCNLND: float = 900.0
    RCCNSEA: float = 50.0
    RRE2DE: float = 0.64952
    RMINICE: float = 60.0


@dataclass
class TRADIATION:
    """TRADIATION (ifs/radiation_setup.F90:49-66)."""
    yrerad: TERAD = field(default_factory=TERAD)
    rad_config: Config = field(default_factory=Config)
    iband_uv: np.ndarray = None
    weight_uv: np.ndarray = None
    iband_par: np.ndarray = None
    weight_par: np.ndarray = None
    trop_bg_aer_mass_ext: float = 0.0
    strat_bg_aer_mass_ext: float = 0.0
    radiation: object = None          # the configured operator (ecrad_amd.interface.Radiation)

    @property
    def nweight_uv(self) -> int:
        return 0 if self.iband_uv is None else int(self.iband_uv.size)

    @property
    def nweight_par(self) -> int:
        return 0 if self.iband_par is None else int(self.iband_par.size)


def get_sw_weights(config: Config, wavelength1: float, wavelength2: float):
    """config%get_sw_weights (radiation_config.F90:1625-1719): the shortwave bands (or g-points) that overlap the
    wavelength range and the fraction of each that lies in it.  Returns (iband [1-based], weight)."""
    if config.n_bands_sw <= 0:
        raise ConfigError("get_sw_weights called before number of shortwave bands set")
    m = config.gas_optics_sw.spectral_def.calc_mapping_from_bands(
        [wavelength1, wavelength2], [1, 2, 3], use_bands=not config.do_cloud_aerosol_per_sw_g_point, use_fluxes=True)
    w = m[:, 1]
    idx = np.nonzero(w > 0.0)[0]
    if idx.size == 0:
        raise ConfigError(f"wavelength range {wavelength1:g} to {wavelength2:g} m is outside shortwave band")
    return (idx + 1).astype(np.int32), np.ascontiguousarray(w[idx])


def setup_radiation_scheme(yradiation: TRADIATION, file_name: Optional[str] = None, directory_name: Optional[str] = None,
                           backend="hip", device_id: int = -1) -> TRADIATION:
    """SETUP_RADIATION_SCHEME (ifs/radiation_setup.F90:70-562): translate the host model's switches (YRERAD) into the
    configuration of the path, let a namelist override them, set the path up (tables -> device) and work out the UV and
    PAR weights.  ``backend`` is 'hip' (the product) or the checker the tests pass in."""
    from .interface import Radiation
    y, c = yradiation.yrerad, yradiation.rad_config
    c.iverbosesetup = 1
    c.iverbose = 1
    if directory_name is not None:
        c.directory_name = directory_name
    c.do_lw_derivatives = y.LAPPROXLWUPDATE                 # :155-158
    c.do_canopy_fluxes_sw = y.LAPPROXSWUPDATE
    if y.NLWOUT > 1:
        c.do_canopy_fluxes_lw = True
    c.do_surface_sw_spectral_flux = True                    # :166
    if y.NLIQOPT == 2:                                      # :217-226
        c.i_liq_model = ILiquidModelSlingo
    elif y.NLIQOPT == 4:
        c.i_liq_model = ILiquidModelSOCRATES
    else:
        raise ConfigError(f"Unavailable liquid optics model in modular radiation scheme: NLIQOPT={y.NLIQOPT}")
    if y.NICEOPT in (3, 6):                                 # :228-250
        c.i_ice_model = IIceModelFu
        if y.NICEOPT == 3 and y.LFU_LW_ICE_OPTICS_BUG:
            c.do_fu_lw_ice_optics_bug = True
    elif y.NICEOPT == 4:
        c.i_ice_model = IIceModelBaran
        if IGasModelECCKD in (c.i_gas_model_sw, c.i_gas_model_lw):
            raise ConfigError("Baran ice optics unavailable with generalized cloud optics")
    elif y.NICEOPT == 5:
        c.i_ice_model = IIceModelYi
    else:
        raise ConfigError(f"Unavailable ice optics model in modular radiation scheme: NICEOPT={y.NICEOPT}")
    c.do_sw_delta_scaling_with_gases = False                # :257
    c.use_aerosols = True                                   # :260-333
    if y.NAERMACC == 1:
        c.n_aerosol_types = 12
        c.i_aerosol_type_map = [-1, -2, -3, 7, 8, 9, -4, 10, 11, 11, -5, 14]
        c.aerosol_optics_override_file_name = ("aerosol_ifs_49R1_20230119.nc" if c.use_general_aerosol_optics
                                               else "aerosol_ifs_rrtm_46R1_with_NI_AM.nc")
    else:
        c.n_aerosol_types = 6
        c.i_aerosol_type_map = [1, 2, 3, 4, 5, 6]
        c.aerosol_optics_override_file_name = "aerosol_ifs_rrtm_tegen.nc"
    c.do_3d_effects = False                                 # :336-390
    lw = {0: ISolverMcICA, 1: ISolverSpartacus, 2: ISolverSpartacus, 3: ISolverTripleclouds, 4: ISolverCloudless}
    if y.NLWSOLVER not in lw:
        raise ConfigError(f"Unknown value for NLWSOLVER: {y.NLWSOLVER}")
    if y.NSWSOLVER not in lw:
        raise ConfigError(f"Unknown value for NSWSOLVER: {y.NSWSOLVER}")
    c.i_solver_lw = lw[y.NLWSOLVER]
    if y.NLWSOLVER == 2:
        c.do_3d_effects = True
    c.i_solver_sw = lw[y.NSWSOLVER]
    if y.NSWSOLVER == 1:
        c.do_3d_effects = False
        if y.NLWSOLVER == 2:
            raise ConfigError("cannot represent 3D effects in LW but not SW")
    elif y.NSWSOLVER == 2:
        c.do_3d_effects = True
        if y.NLWSOLVER == 1:
            raise ConfigError("cannot represent 3D effects in SW but not LW")
    c.min_cloud_effective_size = 500.0                      # :393
    c.do_lw_cloud_scattering = y.NLWSCATTERING in (1, 2)    # :402-417
    c.do_lw_aerosol_scattering = (y.NLWSCATTERING == 2 and y.NAERMACC > 0)
    ov = {1: IOverlapMaximumRandom, 2: IOverlapExponential, 3: IOverlapExponentialRandom}   # :419-450
    if y.NCLOUDOVERLAP not in ov:
        raise ConfigError(f"Unknown value for NCLOUDOVERLAP: {y.NCLOUDOVERLAP}")
    c.i_overlap_scheme = ov[y.NCLOUDOVERLAP]
    regions = (ISolverTripleclouds, ISolverSpartacus)
    if c.i_overlap_scheme != IOverlapExponentialRandom and (c.i_solver_sw in regions or c.i_solver_lw in regions):
        if c.i_solver_sw == c.i_solver_lw:
            c.i_overlap_scheme = IOverlapExponentialRandom
        else:
            raise ConfigError("Tripleclouds and SPARTACUS solvers can only simulate exponential-random overlap")
        c.cloud_fraction_threshold = 2.5e-5
    # Surface albedo in six intervals, longwave emissivity in two (window / rest): :452-500
    y.NLWEMISS = 2
    y.NLWOUT = 1
    y.NSW = 6
    c.sw_albedo_wavelength_bound = [0.25e-6, 0.44e-6, 0.69e-6, 1.19e-6, 2.38e-6]
    c.i_sw_albedo_index = [1, 2, 3, 4, 5, 6]
    c.do_nearest_spectral_sw_albedo = False
    c.lw_emiss_wavelength_bound = [8.0e-6, 13.0e-6]        # YSPECTPLANCK%INIT(2, [8e-6, 13e-6], [1,2,1])
    c.i_lw_emiss_index = [1, 2, 1]
    c.do_nearest_spectral_lw_emiss = True
    if file_name is not None:                               # :504-506: the namelist has the last word
        c.read_into(file_name)
        if directory_name is not None:
            c.directory_name = directory_name
    if y.NSOLARSPECTRUM > 0 and c.i_gas_model_sw == IGasModelIFSRRTMG:             # :523-533 (decided before the handle is configured)
        c.use_spectral_solar_scaling = True      # (RRTMG always has the 14 bands the scaling needs)
    yradiation.radiation = Radiation(c, backend=backend, device_id=device_id)      # SETUP_RADIATION (:514)
    if c.do_sw:
        yradiation.iband_uv, yradiation.weight_uv = get_sw_weights(c, 0.2e-6, 0.4415e-6)     # :538-545
        yradiation.iband_par, yradiation.weight_par = get_sw_weights(c, 0.4e-6, 0.7e-6)
    yradiation.trop_bg_aer_mass_ext = 0.0                   # :548-549
    yradiation.strat_bg_aer_mass_ext = 0.0
    return yradiation


# ---------------------------------------------------------------------------------------------------------------------
def liquid_effective_radius(yderad: TERAD, ppressure, ptemperature, pcloud_frac, pq_liq, pq_rain, pland_frac,
                            pccn_land, pccn_sea) -> np.ndarray:
    """LIQUID_EFFECTIVE_RADIUS (ifs/liquid_effective_radius.F90:1-205): effective radius in microns, (klev, klon)."""
    klev, klon = ppressure.shape
    if yderad.NRADLP == 0:
        return 10.0 + (100000.0 - ppressure) * 3.5
    if yderad.NRADLP == 1:
        return np.broadcast_to(np.where(pland_frac < 0.5, 13.0, 10.0)[None, :], (klev, klon)).copy()
    if yderad.NRADLP != 2:
        raise ConfigError(f"LIQUID EFFECTIVE RADIUS OPTION IRADLP={yderad.NRADLP} NOT AVAILABLE")
    pp_min, pp_max = 4.0, 30.0
    sea = pland_frac < 0.5
    ccn_sea = pccn_sea if yderad.LCCNO else np.full(klon, yderad.RCCNSEA)
    ccn_land = pccn_land if yderad.LCCNL else np.full(klon, yderad.RCCNLND)
    disp = np.where(sea, 0.77, 0.69)
    ntot = np.where(sea, -1.15e-03 * ccn_sea * ccn_sea + 0.963 * ccn_sea + 5.30,
                    -2.10e-04 * ccn_land * ccn_land + 0.568 * ccn_land - 27.9)
    ratio = (0.222 / disp) ** 0.333
    cloudy = (pcloud_frac >= 0.001) & ((pq_liq + pq_rain) > 0.0)
    frac = np.where(cloudy, pcloud_frac, 1.0)
    rho = 1000.0 * ppressure / (RD * ptemperature)
    lwc = rho * pq_liq / frac
    rwc = rho * pq_rain / frac
    with np.errstate(divide="ignore", invalid="ignore"):
        rain_ratio = np.where(lwc > REPSCW, rwc / np.where(lwc > REPSCW, lwc, 1.0), 0.0)
        wood = np.where(lwc > REPSCW, (1.0 + rain_ratio) ** 0.666 / (1.0 + 0.2 * ratio[None, :] * rain_ratio), 1.0)
        re_cubed = (3.0 * (lwc + rwc)) / (4.0 * RPI * ntot[None, :] * disp[None, :])
        big = re_cubed > REPLOG
        re = wood * 100.0 * np.exp(0.333 * np.log(np.where(big, re_cubed, 1.0)))
    re = np.maximum(pp_min, np.minimum(re, pp_max))
    re = np.where(big, re, pp_min)
    return np.where(cloudy, re, pp_min)


def ice_effective_radius(yderad: TERAD, ppressure, ptemperature, pcloud_frac, pq_ice, pq_snow, pgemu) -> np.ndarray:
    """ICE_EFFECTIVE_RADIUS (ifs/ice_effective_radius.F90:1-160): effective radius in microns, (klev, klon)."""
    klev, klon = ppressure.shape
    if yderad.NRADIP == 0:
        return np.full((klev, klon), 40.0)
    if yderad.NRADIP in (1, 2):
        tc = np.minimum(ptemperature - RTT, -0.1)
        re = 326.3 + tc * (12.42 + tc * (0.197 + tc * 0.0012))
        return np.clip(re, 40.0, 130.0) if yderad.NRADIP == 1 else np.clip(re, 30.0, 60.0)
    if yderad.NRADIP != 3:
        raise ConfigError(f"ICE EFFECTIVE RADIUS OPTION NRADLP={yderad.NRADIP} NOT AVAILABLE")
    default_re = 80.0 * yderad.RRE2DE
    if yderad.NMINICE == 0:
        min_diameter = np.full(klon, yderad.RMINICE)
    else:
        min_diameter = 20.0 + (yderad.RMINICE - 20.0) * np.cos(np.arcsin(pgemu))
    cloudy = (pcloud_frac > 0.001) & ((pq_ice + pq_snow) > 0.0)
    frac = np.where(cloudy, pcloud_frac, 1.0)
    rho = 1000.0 * ppressure / (RD * ptemperature)
    iwc = np.where(cloudy, rho * (pq_ice + pq_snow) / frac, 1.0)
    tc = ptemperature - RTT
    aiwc = 45.8966 * iwc ** 0.2214
    biwc = 0.7957 * iwc ** 0.2535
    diameter = (1.2351 + 0.0105 * tc) * (aiwc + biwc * (ptemperature - 83.15))
    diameter = np.minimum(np.maximum(diameter, min_diameter[None, :]), 155.0)
    return np.where(cloudy, diameter * yderad.RRE2DE, default_re)


def cloud_overlap_decorr_len(pgemu, kdecolat: int):
    """CLOUD_OVERLAP_DECORR_LEN (ifs/cloud_overlap_decorr_len.F90:1-120).  Returns (decorrelation length of cloud
    edges in km per column, ratio of the condensate length to it)."""
    if kdecolat == 0:
        return np.full(pgemu.shape, RDECORR_CF), RDECORR_CW / RDECORR_CF
    if kdecolat == 1:
        abs_lat_deg = np.abs(np.arcsin(pgemu) * (180.0 / RPI))
        return 2.899 - 0.02759 * abs_lat_deg, 0.5
    cos_lat = np.cos(np.arcsin(pgemu))
    return 0.75 + 2.149 * cos_lat * cos_lat, 0.5


def set_overlap_param(thermodynamics: Thermodynamics, decorrelation_length) -> np.ndarray:
    """cloud%set_overlap_param (radiation_cloud.F90:195-385, the fixed and the per-column decorrelation length in
    metres): overlap parameter (nlev-1, ncol) of adjacent layers from their separation in a scale-height atmosphere."""
    p, t = thermodynamics.pressure_hl, thermodynamics.temperature_hl
    nlev = p.shape[0] - 1
    d = np.broadcast_to(np.asarray(decorrelation_length, dtype=np.float64), p.shape[1:])
    r_over_g = GAS_CONSTANT_DRY_AIR / ACCEL_DUE_TO_GRAVITY
    out = np.empty((nlev - 1, p.shape[1]))
    with np.errstate(divide="ignore"):
        if p[1, 0] > p[0, 0]:            # top of atmosphere first (:247-265 / :336-354)
            # (the top layer's upper half level may be at zero pressure: the reference uses the layer below only)
            out[0] = np.exp(-(r_over_g / d) * t[1] * np.log(p[2] / p[1]))
            for j in range(1, nlev - 1):
                out[j] = np.exp(-(0.5 * r_over_g / d) * t[j + 1] * np.log(p[j + 2] / p[j]))
        else:                            # surface first (:267-288 / :356-375)
            for j in range(0, nlev - 2):
                out[j] = np.exp(-(0.5 * r_over_g / d) * t[j + 1] * np.log(p[j] / p[j + 2]))
            out[nlev - 2] = np.exp(-(r_over_g / d) * t[nlev - 1] * np.log(p[nlev - 2] / p[nlev - 1]))
    return out


def param_cloud_effective_separation_eta(fraction, pressure_hl, separation_surf, separation_toa, power,
                                         inhom_separation_factor=1.0):
    """cloud%param_cloud_effective_separation_eta (radiation_cloud.F90:602-690).  Returns (inv_cloud_effective_size,
    inv_inhom_effective_size), each (nlev, ncol)."""
    nlev = fraction.shape[0]
    coeff_e = 1.0 - math.exp(-1.0)
    coeff_b = (separation_toa - separation_surf) / coeff_e
    coeff_a = separation_toa - coeff_b
    isurf = 0 if pressure_hl[0, 0] > pressure_hl[1, 0] else nlev
    eta = (pressure_hl[:-1] + pressure_hl[1:]) * (0.5 / pressure_hl[isurf][None, :])
    eff_separation = coeff_a + coeff_b * np.exp(-eta ** power)
    inv_cloud = 1.0 / (eff_separation * np.sqrt(np.maximum(1.0e-5, fraction * (1.0 - fraction))))
    inv_inhom = 1.0 / (eff_separation * inhom_separation_factor
                       * np.sqrt(np.maximum(1.0e-5, 0.5 * fraction * (1.0 - 0.5 * fraction))))
    return np.ascontiguousarray(inv_cloud), np.ascontiguousarray(inv_inhom)


IFS_OUTPUTS_PROFILE = ("PFLUX_SW", "PFLUX_LW", "PFLUX_SW_CLEAR", "PFLUX_LW_CLEAR", "PLWDERIVATIVE")
IFS_OUTPUTS_SURFACE = ("PFLUX_SW_DN", "PFLUX_LW_DN", "PFLUX_SW_DN_CLEAR", "PFLUX_LW_DN_CLEAR", "PFLUX_DIR",
                       "PFLUX_DIR_CLEAR", "PFLUX_DIR_INTO_SUN", "PFLUX_UV", "PFLUX_PAR", "PFLUX_PAR_CLEAR",
                       "PFLUX_SW_DN_TOA", "PEMIS_OUT")
IFS_OUTPUTS_BAND = ("PSWDIFFUSEBAND", "PSWDIRECTBAND")


def allocate_ifs_outputs(yradiation: TRADIATION, klon: int, klev: int) -> dict:
    """The INTENT(OUT) arrays of RADIATION_SCHEME (ifs/radiation_scheme.F90:172-213)."""
    out = {n: np.zeros((klev + 1, klon)) for n in IFS_OUTPUTS_PROFILE}
    out.update({n: np.zeros(klon) for n in IFS_OUTPUTS_SURFACE})
    out.update({n: np.zeros((yradiation.yrerad.NSW, klon)) for n in IFS_OUTPUTS_BAND})
    return out


def radiation_scheme(yradiation: TRADIATION, kidia: int, kfdia: int, klon: int, klev: int, kaerosol: int, *,
                     PSOLAR_IRRADIANCE, PMU0, PTEMPERATURE_SKIN, PALBEDO_DIF, PALBEDO_DIR, PSPECTRALEMISS,
                     PCCN_LAND, PCCN_SEA, PGELAM, PGEMU, PLAND_SEA_MASK, PPRESSURE, PTEMPERATURE, PPRESSURE_H,
                     PTEMPERATURE_H, PQ, PCO2, PCH4, PN2O, PNO2, PCFC11, PCFC12, PHCFC22, PCCL4, PO3, PCLOUD_FRAC,
                     PQ_LIQUID, PQ_ICE, PQ_RAIN, PQ_SNOW, PAEROSOL_OLD, PAEROSOL, out: Optional[dict] = None,
                     PRE_LIQ=None, PRE_ICE=None, ISEED=None, PCLOUD_OVERLAP=None) -> dict:
    """RADIATION_SCHEME (ifs/radiation_scheme.F90:1-675) with the reference's argument names.  Columns outside
    kidia..kfdia (1-based, inclusive) of ``out`` are not touched.  The optional PRE_LIQ / PRE_ICE / ISEED /
    PCLOUD_OVERLAP are the reference's BITIDENTITY_TESTING arguments (effective radii in metres)."""
    y, config, rad = yradiation.yrerad, yradiation.rad_config, yradiation.radiation
    if rad is None:
        raise RuntimeError("radiation_scheme called before setup_radiation_scheme")
    if out is None:
        out = allocate_ifs_outputs(yradiation, klon, klev)
    cols = slice(kidia - 1, kfdia)
    f64 = lambda a: np.asarray(a, dtype=np.float64)

    # thermodynamics (:329-345): the half-level temperature at the surface is extrapolated from the lowest full level
    temperature_hl = np.array(f64(PTEMPERATURE_H), copy=True)
    temperature_hl[klev, cols] = (f64(PTEMPERATURE)[klev - 1, cols]
                                  + 0.5 * (temperature_hl[klev, cols] - temperature_hl[klev - 1, cols]))
    thermodynamics = Thermodynamics(np.ascontiguousarray(f64(PPRESSURE_H)), np.ascontiguousarray(temperature_hl))
    thermodynamics.calc_saturation_wrt_liquid()

    # single-level fields (:347-387)
    iseed = np.arange(1, klon + 1, dtype=np.int32)                    # init_seed_simple
    if ISEED is not None:
        iseed[cols] = np.asarray(ISEED)[cols]
    single_level = SingleLevel(cos_sza=np.ascontiguousarray(f64(PMU0)),
                               skin_temperature=np.ascontiguousarray(f64(PTEMPERATURE_SKIN)),
                               sw_albedo=np.ascontiguousarray(f64(PALBEDO_DIF)),
                               lw_emissivity=np.ascontiguousarray(f64(PSPECTRALEMISS)),
                               sw_albedo_direct=np.ascontiguousarray(f64(PALBEDO_DIR)),
                               solar_irradiance=float(PSOLAR_IRRADIANCE), iseed=iseed)
    if config.use_spectral_solar_scaling:
        # ifs/radiation_scheme.F90:369-397: RRTMG uses the old Kurucz solar spectrum; per-band factors towards the Whole
        # Heliosphere Interval 2008 reference spectrum (NSOLARSPECTRUM = 1) or the Coddington et al. (2016) climate data record
        single_level.spectral_solar_scaling = np.array(
            [1.0000, 1.0000, 1.0000, 1.0478, 1.0404, 1.0317, 1.0231, 1.0054, 0.98413, 0.99863, 0.99907, 0.90589, 0.92213, 1.0000]
            if yradiation.yrerad.NSOLARSPECTRUM == 1 else
            [0.99892, 0.99625, 1.00822, 1.01587, 1.01898, 1.01044, 1.08441, 0.99398, 1.00553, 0.99533, 1.01509, 0.92331, 0.92681, 0.99749])

    # clouds (:389-455)
    cloud = None
    if config.do_clouds:
        q_liq = f64(PQ_LIQUID)
        q_ice = f64(PQ_ICE) + f64(PQ_SNOW)
        if PRE_LIQ is not None:
            re_liq = f64(PRE_LIQ)
        else:
            re_liq = liquid_effective_radius(y, f64(PPRESSURE), f64(PTEMPERATURE), f64(PCLOUD_FRAC), q_liq, f64(PQ_RAIN),
                                             f64(PLAND_SEA_MASK), f64(PCCN_LAND), f64(PCCN_SEA)) * 1.0e-6
        if PRE_ICE is not None:
            re_ice = f64(PRE_ICE)
        else:
            re_ice = ice_effective_radius(y, f64(PPRESSURE), f64(PTEMPERATURE), f64(PCLOUD_FRAC), f64(PQ_ICE), f64(PQ_SNOW),
                                          f64(PGEMU)) * 1.0e-6
        if PCLOUD_OVERLAP is not None:
            overlap_param = np.ascontiguousarray(f64(PCLOUD_OVERLAP))
        else:
            decorr_len_km, _ratio = cloud_overlap_decorr_len(f64(PGEMU), y.NDECOLAT)
            overlap_param = set_overlap_param(thermodynamics, decorr_len_km * 1000.0)
        fraction = np.array(f64(PCLOUD_FRAC), copy=True)              # radiation() crops it in place
        cloud = Cloud(fraction, np.ascontiguousarray(np.stack([q_liq, q_ice])),
                      np.ascontiguousarray(np.stack([re_liq, re_ice])),
                      np.full((klev, klon), y.RCLOUD_FRAC_STD),        # create_fractional_std
                      overlap_param)
        if ISolverSpartacus in (config.i_solver_lw, config.i_solver_sw):
            cloud.inv_cloud_effective_size, cloud.inv_inhom_effective_size = param_cloud_effective_separation_eta(
                fraction, thermodynamics.pressure_hl, y.RCLOUD_SEPARATION_SCALE_SURF, y.RCLOUD_SEPARATION_SCALE_TOA,
                3.5, 0.75)

    # aerosols (:457-515): prognostic/climatological mass mixing ratios, or the Tegen optical depths turned into them
    layer_mass = (thermodynamics.pressure_hl[1:] - thermodynamics.pressure_hl[:-1]) * (1.0 / ACCEL_DUE_TO_GRAVITY)
    aerosol = None
    if config.use_aerosols:
        if y.NAERMACC == 1:
            mr = np.maximum(f64(PAEROSOL), 0.0)
            old = f64(PAEROSOL_OLD)                                    # (klev, 6, klon)
            if yradiation.trop_bg_aer_mass_ext > 0.0:
                mr[ITYPE_TROP_BG_AER - 1] += old[:, 0, :] / (layer_mass * yradiation.trop_bg_aer_mass_ext)
            if yradiation.strat_bg_aer_mass_ext > 0.0:
                mr[ITYPE_STRAT_BG_AER - 1] += old[:, 5, :] / (layer_mass * yradiation.strat_bg_aer_mass_ext)
        else:
            mr = np.transpose(f64(PAEROSOL_OLD), (1, 0, 2)) / layer_mass[None, :, :]
        aerosol = Aerosol(mixing_ratio=np.ascontiguousarray(mr), istartlev=1, iendlev=klev)

    # gases (:517-535): mass mixing ratios, O2 well mixed
    gas = Gas.allocate(klon, klev)
    for igas, a in ((IH2O, PQ), (ICO2, PCO2), (ICH4, PCH4), (IN2O, PN2O), (ICFC11, PCFC11), (ICFC12, PCFC12),
                    (IHCFC22, PHCFC22), (ICCl4, PCCL4), (IO3, PO3)):
        gas.put(igas, IMassMixingRatio, f64(a))
    gas.put(IO2, IVolumeMixingRatio, 0.20944)
    rad.set_gas_units(gas)

    flux = Flux.allocate(config, klon, klev)
    rad.radiation(klon, klev, kidia, kfdia, single_level, thermodynamics, gas, cloud, aerosol, flux)

    # net fluxes and the surface diagnostics (:585-655)
    z = lambda name: getattr(flux, name)
    if config.do_sw:
        out["PFLUX_SW"][:, cols] = (z("sw_dn") - z("sw_up"))[:, cols]
        out["PFLUX_SW_CLEAR"][:, cols] = (z("sw_dn_clear") - z("sw_up_clear"))[:, cols]
        out["PFLUX_SW_DN"][cols] = z("sw_dn")[klev, cols]
        out["PFLUX_SW_DN_CLEAR"][cols] = z("sw_dn_clear")[klev, cols]
        out["PFLUX_DIR"][cols] = z("sw_dn_direct")[klev, cols]
        out["PFLUX_DIR_CLEAR"][cols] = z("sw_dn_direct_clear")[klev, cols]
        mu0 = f64(PMU0)[cols]
        sunny = mu0 > np.finfo(np.float64).eps
        out["PFLUX_DIR_INTO_SUN"][cols] = np.where(sunny, out["PFLUX_DIR"][cols] / np.where(sunny, mu0, 1.0), 0.0)
        out["PFLUX_SW_DN_TOA"][cols] = z("sw_dn")[0, cols]
        band = z("sw_dn_surf_band")                                    # (ncol, nband)
        band_clear = z("sw_dn_surf_clear_band")
        out["PFLUX_UV"][cols] = band[cols][:, yradiation.iband_uv - 1] @ yradiation.weight_uv
        out["PFLUX_PAR"][cols] = band[cols][:, yradiation.iband_par - 1] @ yradiation.weight_par
        out["PFLUX_PAR_CLEAR"][cols] = band_clear[cols][:, yradiation.iband_par - 1] @ yradiation.weight_par
    if config.do_lw:
        out["PFLUX_LW"][:, cols] = (z("lw_dn") - z("lw_up"))[:, cols]
        out["PFLUX_LW_CLEAR"][:, cols] = (z("lw_dn_clear") - z("lw_up_clear"))[:, cols]
        out["PFLUX_LW_DN"][cols] = z("lw_dn")[klev, cols]
        out["PFLUX_LW_DN_CLEAR"][cols] = z("lw_dn_clear")[klev, cols]
        # the emissivity a black-body-emitting surface would need to give the same net longwave flux (:636-645)
        black_body_net_lw = out["PFLUX_LW_DN"][cols] - RSIGMA * f64(PTEMPERATURE_SKIN)[cols] ** 4
        ok = np.abs(black_body_net_lw) > 1.0e-5
        ratio = out["PFLUX_LW"][klev, cols] / np.where(ok, black_body_net_lw, 1.0)
        out["PEMIS_OUT"][cols] = np.where(ok, np.maximum(0.8, np.minimum(0.99, ratio)), f64(PSPECTRALEMISS)[0, cols])
        if y.LAPPROXLWUPDATE:
            out["PLWDERIVATIVE"][:, cols] = z("lw_derivatives")[:, cols]
    if y.LAPPROXSWUPDATE and config.do_sw:
        out["PSWDIFFUSEBAND"][:, cols] = z("sw_dn_diffuse_surf_canopy")[cols].T
        out["PSWDIRECTBAND"][:, cols] = z("sw_dn_direct_surf_canopy")[cols].T
    out["_flux"] = flux                # (for the offline driver and the tests; not an IFS output)
    out["_thermodynamics"] = thermodynamics
    return out


# ---------------------------------------------------------------------------------------------------------------------
# NPROMA blocking (driver/ifs_blocking.F90)
@dataclass
class IfsConfig:
    """ifs_config_type (driver/ifs_blocking.F90:22-36): first field (0-based) of every variable in the blocked array
    ZRGP(NPROMA, IFLDSTOT, NGPBLKS) == numpy (ngpblks, ifldstot, nproma); -1 = the variable is not held."""
    off: dict = field(default_factory=dict)
    ifldstot: int = 0
    nlev: int = 0

    def __getattr__(self, name):
        off = self.__dict__.get("off", {})
        if name in off:
            return off[name]
        raise AttributeError(name)


def ifs_setup_indices(yradiation: TRADIATION, nlev: int, lldebug: bool = False, bitidentity: bool = False) -> IfsConfig:
    """ifs_setup_indices (driver/ifs_blocking.F90:55-282): the order of the fields in the blocked array, the way the
    IFS lays out its radiation grid-point array: inputs, outputs, then the fields only used with diagnostics."""
    y, c = yradiation.yrerad, yradiation.rad_config
    ic = IfsConfig(nlev=nlev)
    nxt = [0]

    def indrad(name, kflds, lduse):
        if lduse:
            ic.off[name] = nxt[0]
            nxt[0] += kflds
        else:
            ic.off[name] = -1

    llactaero = 0 < c.n_aerosol_types <= 21 and y.NAERMACC == 0
    indrad("igi", 1, lldebug)
    for name, n in (("imu0", 1), ("iamu0", 1), ("iemiss", y.NLWEMISS), ("its", 1), ("islm", 1), ("iccnl", 1),
                    ("iccno", 1), ("ibas", 1), ("itop", 1), ("igelam", 1), ("igemu", 1), ("iclon", 1), ("islon", 1),
                    ("iald", y.NSW), ("ialp", y.NSW), ("iti", nlev), ("ipr", nlev), ("iqs", nlev), ("iwv", nlev),
                    ("iclc", nlev), ("ilwa", nlev), ("iiwa", nlev), ("iswa", nlev), ("irwa", nlev), ("irra", nlev),
                    ("idp", nlev)):
        indrad(name, n, True)
    indrad("ioz", nlev, False)
    indrad("iecpo3", nlev, False)
    indrad("ihpr", nlev + 1, True)
    indrad("iaprs", nlev + 1, True)
    indrad("ihti", nlev + 1, True)
    indrad("iaero", c.n_aerosol_types * nlev, llactaero and y.NAERMACC == 0)
    if y.NAERMACC == 1:
        indrad("iaero", c.n_aerosol_types * nlev, y.LDIAGFORCING)
    for name, n, use in (("ifrsod", 1, True), ("ifrted", y.NLWOUT, True), ("ifrsodc", 1, True), ("ifrtedc", 1, True),
                         ("iemit", 1, True), ("isudu", 1, True), ("iuvdf", 1, True), ("iparf", 1, True),
                         ("iparcf", 1, True), ("itincf", 1, True), ("ifdir", 1, True), ("ifdif", 1, True),
                         ("icdir", 1, True), ("ilwderivative", nlev + 1, y.LAPPROXLWUPDATE),
                         ("iswdirectband", y.NSW, y.LAPPROXSWUPDATE), ("iswdiffuseband", y.NSW, y.LAPPROXSWUPDATE),
                         ("ifrso", nlev + 1, True), ("iswfc", nlev + 1, True), ("ifrth", nlev + 1, True),
                         ("ilwfc", nlev + 1, True)):
        indrad(name, n, use)
    diag = (("iaer", 6 * nlev), ("ioz", nlev), ("iico2", nlev), ("iich4", nlev), ("iin2o", nlev), ("ino2", nlev),
            ("ic11", nlev), ("ic12", nlev), ("ic22", nlev), ("icl4", nlev))
    for name, n in diag:
        indrad(name, n, y.LDIAGFORCING)
    indrad("igix", 1, lldebug)
    if not y.LDIAGFORCING:
        if c.n_aerosol_types == 0 or y.NAERMACC == 1:
            indrad("iaero", c.n_aerosol_types * nlev, True)
        for name, n in diag:
            indrad(name, n, True)
    if bitidentity:
        indrad("ire_liq", nlev, True)
        indrad("ire_ice", nlev, True)
        indrad("ioverlap", nlev - 1, True)
    ic.ifldstot = nxt[0]
    return ic


def ifs_copy_inputs_to_blocked(nproma: int, ifs_config: IfsConfig, yradiation: TRADIATION, ncol: int, nlev: int,
                               single_level, thermodynamics, gas, cloud, aerosol, sin_latitude, longitude_rad, land_frac,
                               pressure_fl, temperature_fl, bitidentity: bool = False):
    """ifs_copy_inputs_to_blocked (driver/ifs_blocking.F90:285-470).  Returns (zrgp, iseed): zrgp is numpy
    (ngpblks, ifldstot, nproma); iseed (ngpblks, nproma) int32 or None."""
    y, c, ic = yradiation.yrerad, yradiation.rad_config, ifs_config
    ngpblks = (ncol - 1) // nproma + 1
    zrgp = np.zeros((ngpblks, ic.ifldstot, nproma))
    iseed = np.zeros((ngpblks, nproma), dtype=np.int32) if bitidentity else None
    npad = ngpblks * nproma

    def put(first, a):
        """a: (nfield, ncol) or (ncol,) -> fields first.. of every block"""
        a = np.atleast_2d(np.asarray(a, dtype=np.float64))
        buf = np.zeros((a.shape[0], npad))
        buf[:, :ncol] = a
        zrgp[:, first:first + a.shape[0], :] = buf.reshape(a.shape[0], ngpblks, nproma).transpose(1, 0, 2)

    put(ic.iamu0, single_level.cos_sza)
    put(ic.iemiss, single_level.lw_emissivity[:y.NLWEMISS])
    put(ic.its, single_level.skin_temperature)
    put(ic.islm, land_frac)
    put(ic.iccnl, np.full(ncol, y.RCCNLND))
    put(ic.iccno, np.full(ncol, y.RCCNSEA))
    put(ic.igelam, longitude_rad)
    put(ic.igemu, sin_latitude)
    put(ic.iald, single_level.sw_albedo[:y.NSW])
    put(ic.ialp, (single_level.sw_albedo_direct if single_level.sw_albedo_direct is not None else single_level.sw_albedo)[:y.NSW])
    put(ic.iti, temperature_fl)
    put(ic.ipr, pressure_fl)
    put(ic.iwv, gas.get(IH2O, IMassMixingRatio))
    if c.do_clouds:
        put(ic.iclc, cloud.fraction)
        put(ic.ilwa, cloud.mixing_ratio[0])
        put(ic.iiwa, cloud.mixing_ratio[1])
    if y.NAERMACC == 1 and aerosol is not None:
        put(ic.iaero, aerosol.mixing_ratio.reshape(-1, ncol))          # (type, level) -> type-major fields
    put(ic.iaprs, thermodynamics.pressure_hl)
    put(ic.ihti, thermodynamics.temperature_hl)
    for name, igas in (("iico2", ICO2), ("iich4", ICH4), ("iin2o", IN2O), ("ic11", ICFC11), ("ic12", ICFC12),
                       ("ic22", IHCFC22), ("icl4", ICCl4), ("ioz", IO3)):
        put(getattr(ic, name), gas.get(igas, IMassMixingRatio))
    if bitidentity:
        if c.do_clouds:
            put(ic.ire_liq, cloud.effective_radius[0])
            put(ic.ire_ice, cloud.effective_radius[1])
            put(ic.ioverlap, cloud.overlap_param)
            seeds = np.zeros(npad, dtype=np.int32)
            seeds[:ncol] = single_level.iseed
            iseed[:] = seeds.reshape(ngpblks, nproma)
    return zrgp, iseed


_SCHEME_INPUTS = (  # (argument of RADIATION_SCHEME, field of the blocked array, number of fields as a function of (y, nlev, naer))
    ("PMU0", "iamu0", lambda y, n, a: 0), ("PTEMPERATURE_SKIN", "its", lambda y, n, a: 0),
    ("PALBEDO_DIF", "iald", lambda y, n, a: y.NSW), ("PALBEDO_DIR", "ialp", lambda y, n, a: y.NSW),
    ("PSPECTRALEMISS", "iemiss", lambda y, n, a: y.NLWEMISS), ("PCCN_LAND", "iccnl", lambda y, n, a: 0),
    ("PCCN_SEA", "iccno", lambda y, n, a: 0), ("PGELAM", "igelam", lambda y, n, a: 0), ("PGEMU", "igemu", lambda y, n, a: 0),
    ("PLAND_SEA_MASK", "islm", lambda y, n, a: 0), ("PPRESSURE", "ipr", lambda y, n, a: n),
    ("PTEMPERATURE", "iti", lambda y, n, a: n), ("PPRESSURE_H", "iaprs", lambda y, n, a: n + 1),
    ("PTEMPERATURE_H", "ihti", lambda y, n, a: n + 1), ("PQ", "iwv", lambda y, n, a: n), ("PCO2", "iico2", lambda y, n, a: n),
    ("PCH4", "iich4", lambda y, n, a: n), ("PN2O", "iin2o", lambda y, n, a: n), ("PNO2", "ino2", lambda y, n, a: n),
    ("PCFC11", "ic11", lambda y, n, a: n), ("PCFC12", "ic12", lambda y, n, a: n), ("PHCFC22", "ic22", lambda y, n, a: n),
    ("PCCL4", "icl4", lambda y, n, a: n), ("PO3", "ioz", lambda y, n, a: n), ("PCLOUD_FRAC", "iclc", lambda y, n, a: n),
    ("PQ_LIQUID", "ilwa", lambda y, n, a: n), ("PQ_ICE", "iiwa", lambda y, n, a: n), ("PQ_RAIN", "irwa", lambda y, n, a: n),
    ("PQ_SNOW", "iswa", lambda y, n, a: n))
_SCHEME_OUTPUTS = (  # (output of RADIATION_SCHEME, field) in the order of the call at driver/ecrad_ifs_driver_blocked.F90:196-234
    ("PFLUX_SW", "ifrso"), ("PFLUX_LW", "ifrth"), ("PFLUX_SW_CLEAR", "iswfc"), ("PFLUX_LW_CLEAR", "ilwfc"),
    ("PFLUX_SW_DN", "ifrsod"), ("PFLUX_LW_DN", "ifrted"), ("PFLUX_SW_DN_CLEAR", "ifrsodc"), ("PFLUX_LW_DN_CLEAR", "ifrtedc"),
    ("PFLUX_DIR", "ifdir"), ("PFLUX_DIR_CLEAR", "icdir"), ("PFLUX_DIR_INTO_SUN", "isudu"), ("PFLUX_UV", "iuvdf"),
    ("PFLUX_PAR", "iparf"), ("PFLUX_PAR_CLEAR", "iparcf"), ("PFLUX_SW_DN_TOA", "itincf"), ("PEMIS_OUT", "iemit"),
    ("PLWDERIVATIVE", "ilwderivative"), ("PSWDIFFUSEBAND", "iswdiffuseband"), ("PSWDIRECTBAND", "iswdirectband"))


def radiation_scheme_blocked(yradiation: TRADIATION, ifs_config: IfsConfig, zrgp: np.ndarray, ncol: int, nlev: int,
                             solar_irradiance: float, iseed: Optional[np.ndarray] = None, per_block: bool = False,
                             bitidentity: bool = False) -> None:
    """The column loop of driver/ecrad_ifs_driver_blocked.F90:180-236 over the blocked array: inputs are taken from
    and outputs written into ``zrgp``.  By default every block goes through the path in ONE call (the blocks' columns
    gathered into one batch: what the GPU wants); ``per_block`` reproduces the reference's call per block."""
    y, c, ic = yradiation.yrerad, yradiation.rad_config, ifs_config
    ngpblks, _, nproma = zrgp.shape
    naer = c.n_aerosol_types

    def run(view, ncols_here, seeds):
        """view: (nfields, ncols) gather of the blocked array for the columns of one call"""
        def get(first, n):
            if first < 0:
                return np.zeros((max(n, 1), ncols_here))[0] if n == 0 else np.zeros((n, ncols_here))
            return view[first] if n == 0 else view[first:first + n]
        kw = {arg: get(getattr(ic, fld), nf(y, nlev, naer)) for arg, fld, nf in _SCHEME_INPUTS}
        # PAEROSOL_OLD(KLON,6,KLEV) and PAEROSOL(KLON,KLEV,KAEROSOL) as the reference's call sees the field runs
        kw["PAEROSOL_OLD"] = get(ic.iaer, 6 * nlev).reshape(nlev, 6, ncols_here)
        kw["PAEROSOL"] = get(ic.iaero, naer * nlev).reshape(naer, nlev, ncols_here) if naer > 0 else np.zeros((0, nlev, ncols_here))
        if bitidentity:
            kw["PRE_LIQ"] = get(ic.ire_liq, nlev)
            kw["PRE_ICE"] = get(ic.ire_ice, nlev)
            kw["PCLOUD_OVERLAP"] = get(ic.ioverlap, nlev - 1)
            kw["ISEED"] = seeds
        return radiation_scheme(yradiation, 1, ncols_here, ncols_here, nlev, naer, PSOLAR_IRRADIANCE=solar_irradiance, **kw)

    def scatter(out, dst_fields, ncols_here):
        """outputs of one call -> rows of a (nfields, ncols) buffer"""
        for name, fld in _SCHEME_OUTPUTS:
            first = getattr(ic, fld)
            if first < 0:
                continue
            a = np.atleast_2d(out[name])
            dst_fields[first:first + a.shape[0], :ncols_here] = a

    if per_block:
        for ib in range(ngpblks):
            il = min(nproma, ncol - ib * nproma)
            view = zrgp[ib][:, :il]
            out = run(view, il, None if iseed is None else iseed[ib, :il])
            scatter(out, zrgp[ib], il)
        return
    # one batch: (ngpblks, nfields, nproma) -> (nfields, ngpblks * nproma), valid columns only
    batch = np.ascontiguousarray(zrgp.transpose(1, 0, 2)).reshape(ic.ifldstot, ngpblks * nproma)
    out = run(batch[:, :ncol], ncol, None if iseed is None else iseed.reshape(-1)[:ncol])
    scatter(out, batch, ncol)
    zrgp[:] = batch.reshape(ic.ifldstot, ngpblks, nproma).transpose(1, 0, 2)


def ifs_copy_fluxes_from_blocked(ifs_config: IfsConfig, yradiation: TRADIATION, ncol: int, nlev: int, zrgp: np.ndarray,
                                 flux: Flux) -> dict:
    """ifs_copy_fluxes_from_blocked (driver/ifs_blocking.F90:473-545): net fluxes land in flux%{sw,lw}_up[_clear] and
    the surface / TOA values in the last / first half level of the downward profiles, exactly as the reference's
    driver parks them before save_net_fluxes.  Returns the single-level diagnostics."""
    y, ic = yradiation.yrerad, ifs_config
    ngpblks, _, nproma = zrgp.shape
    batch = zrgp.transpose(1, 0, 2).reshape(ic.ifldstot, ngpblks * nproma)[:, :ncol]
    prof = lambda first: batch[first:first + nlev + 1]
    a = flux.arrays
    a["sw_up"][:] = prof(ic.ifrso)
    a["lw_up"][:] = prof(ic.ifrth)
    a["sw_up_clear"][:] = prof(ic.iswfc)
    a["lw_up_clear"][:] = prof(ic.ilwfc)
    if "lw_derivatives" in a:
        a["lw_derivatives"][:] = prof(ic.ilwderivative) if y.LAPPROXLWUPDATE else 0.0
    a["sw_dn"][nlev] = batch[ic.ifrsod]
    a["lw_dn"][nlev] = batch[ic.ifrted]
    a["sw_dn_clear"][nlev] = batch[ic.ifrsodc]
    a["lw_dn_clear"][nlev] = batch[ic.ifrtedc]
    a["sw_dn_direct"][nlev] = batch[ic.ifdir]
    a["sw_dn_direct_clear"][nlev] = batch[ic.icdir]
    a["sw_dn"][0] = batch[ic.itincf]
    nsw = y.NSW
    return {"flux_sw_direct_normal": batch[ic.isudu].copy(), "flux_uv": batch[ic.iuvdf].copy(),
            "flux_par": batch[ic.iparf].copy(), "flux_par_clear": batch[ic.iparcf].copy(),
            "emissivity_out": batch[ic.iemit].copy(),
            "flux_diffuse_band": batch[ic.iswdiffuseband:ic.iswdiffuseband + nsw].copy() if y.LAPPROXSWUPDATE else np.zeros((nsw, ncol)),
            "flux_direct_band": batch[ic.iswdirectband:ic.iswdirectband + nsw].copy() if y.LAPPROXSWUPDATE else np.zeros((nsw, ncol))}


def net_to_up(config: Config, flux: Flux, nlev: int) -> None:
    """driver/ecrad_ifs_driver.F90:392-418: the drivers hold NET fluxes in flux%*_up and only the TOA / surface values of
    the downward profiles; turn that into an upward profile such that save_net_fluxes (down minus up) writes the net
    flux everywhere."""
    a = flux.arrays
    for sfx in ("sw", "lw"):
        if not getattr(config, "do_" + sfx):
            continue
        for clear in ("", "_clear") if config.do_clear else ("",):
            up, dn = a[sfx + "_up" + clear], a[sfx + "_dn" + clear]
            up *= -1.0
            up[0] += dn[0]
            up[nlev] += dn[nlev]


def run_ifs_driver(namelist: str, input_file: str, output_file: Optional[str] = None, blocked: bool = False,
                   bitidentity: bool = False, per_block: bool = False, backend="hip", directory_name: Optional[str] = None,
                   yrerad: Optional[TERAD] = None):
    """The offline IFS-style drivers (driver/ecrad_ifs_driver.F90, driver/ecrad_ifs_driver_blocked.F90): read the
    input file as the ordinary driver does, hand the fields to ``radiation_scheme`` as IFS arrays (optionally through
    the NPROMA-blocked array), and write the net fluxes.  Returns (config, thermodynamics, flux, diagnostics)."""
    from .driver import DriverConfig, read_input, save_net_fluxes
    dc = DriverConfig.read(namelist)
    yr = TRADIATION() if yrerad is None else TRADIATION(yrerad=yrerad)          # (host switches the namelist does not hold)
    yr.rad_config.read_into(namelist)
    yr.yrerad.NAERMACC = 1 if yr.rad_config.use_aerosols else 0            # ecrad_ifs_driver.F90:153-157
    setup_radiation_scheme(yr, file_name=namelist, directory_name=directory_name, backend=backend)
    c, y = yr.rad_config, yr.yrerad
    ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol = read_input(input_file, c, dc)
    from .ncfile import NcFile
    with NcFile(input_file) as f:
        sin_latitude = np.sin(np.asarray(f.get("lat"), dtype=np.float64) * RPI / 180.0) if f.exists("lat") else np.zeros(ncol)
        longitude_rad = np.asarray(f.get("lon"), dtype=np.float64) * RPI / 180.0 if f.exists("lon") else np.zeros(ncol)
    gas.set_units(IMassMixingRatio)
    land_frac = np.zeros(ncol)
    pressure_fl = 0.5 * (thermodynamics.pressure_hl[:-1] + thermodynamics.pressure_hl[1:])
    temperature_fl = 0.5 * (thermodynamics.temperature_hl[:-1] + thermodynamics.temperature_hl[1:])
    zeros = np.zeros((nlev, ncol))
    flux = Flux.allocate(c, ncol, nlev)
    if blocked:
        ic = ifs_setup_indices(yr, nlev, bitidentity=bitidentity)
        zrgp, iseed = ifs_copy_inputs_to_blocked(dc.nblocksize, ic, yr, ncol, nlev, single_level, thermodynamics, gas, cloud,
                                                 aerosol, sin_latitude, longitude_rad, land_frac, pressure_fl,
                                                 temperature_fl, bitidentity=bitidentity)
        radiation_scheme_blocked(yr, ic, zrgp, ncol, nlev, single_level.solar_irradiance, iseed=iseed, per_block=per_block,
                                 bitidentity=bitidentity)
        diag = ifs_copy_fluxes_from_blocked(ic, yr, ncol, nlev, zrgp, flux)
    else:
        g = lambda igas: gas.mixing_ratio[igas - 1]
        kw = {}
        if bitidentity and c.do_clouds:
            kw = dict(PRE_LIQ=cloud.effective_radius[0], PRE_ICE=cloud.effective_radius[1],
                      PCLOUD_OVERLAP=cloud.overlap_param, ISEED=single_level.iseed)
        out = radiation_scheme(
            yr, 1, ncol, ncol, nlev, c.n_aerosol_types, PSOLAR_IRRADIANCE=single_level.solar_irradiance,
            PMU0=single_level.cos_sza, PTEMPERATURE_SKIN=single_level.skin_temperature, PALBEDO_DIF=single_level.sw_albedo,
            PALBEDO_DIR=single_level.sw_albedo_direct if single_level.sw_albedo_direct is not None else single_level.sw_albedo,
            PSPECTRALEMISS=single_level.lw_emissivity, PCCN_LAND=np.full(ncol, y.RCCNLND), PCCN_SEA=np.full(ncol, y.RCCNSEA),
            PGELAM=longitude_rad, PGEMU=sin_latitude, PLAND_SEA_MASK=land_frac, PPRESSURE=pressure_fl,
            PTEMPERATURE=temperature_fl, PPRESSURE_H=thermodynamics.pressure_hl, PTEMPERATURE_H=thermodynamics.temperature_hl,
            PQ=g(IH2O), PCO2=g(ICO2), PCH4=g(ICH4), PN2O=g(IN2O), PNO2=g(INO2), PCFC11=g(ICFC11), PCFC12=g(ICFC12),
            PHCFC22=g(IHCFC22), PCCL4=g(ICCl4), PO3=g(IO3),
            PCLOUD_FRAC=cloud.fraction if c.do_clouds else zeros, PQ_LIQUID=cloud.mixing_ratio[0] if c.do_clouds else zeros,
            PQ_ICE=cloud.mixing_ratio[1] if c.do_clouds else zeros, PQ_RAIN=zeros, PQ_SNOW=zeros,
            PAEROSOL_OLD=np.zeros((nlev, 6, ncol)),
            PAEROSOL=aerosol.mixing_ratio if aerosol is not None else np.zeros((0, nlev, ncol)), **kw)
        a = flux.arrays                                                    # what the reference's call writes where
        a["sw_up"][:], a["lw_up"][:] = out["PFLUX_SW"], out["PFLUX_LW"]
        a["sw_up_clear"][:], a["lw_up_clear"][:] = out["PFLUX_SW_CLEAR"], out["PFLUX_LW_CLEAR"]
        a["sw_dn"][nlev], a["lw_dn"][nlev] = out["PFLUX_SW_DN"], out["PFLUX_LW_DN"]
        a["sw_dn_clear"][nlev], a["lw_dn_clear"][nlev] = out["PFLUX_SW_DN_CLEAR"], out["PFLUX_LW_DN_CLEAR"]
        a["sw_dn_direct"][nlev], a["sw_dn_direct_clear"][nlev] = out["PFLUX_DIR"], out["PFLUX_DIR_CLEAR"]
        a["sw_dn"][0] = out["PFLUX_SW_DN_TOA"]
        if "lw_derivatives" in a:
            a["lw_derivatives"][:] = out["PLWDERIVATIVE"]
        diag = {"flux_sw_direct_normal": out["PFLUX_DIR_INTO_SUN"], "flux_uv": out["PFLUX_UV"], "flux_par": out["PFLUX_PAR"],
                "flux_par_clear": out["PFLUX_PAR_CLEAR"], "emissivity_out": out["PEMIS_OUT"],
                "flux_diffuse_band": out["PSWDIFFUSEBAND"], "flux_direct_band": out["PSWDIRECTBAND"]}
    net_to_up(c, flux, nlev)
    if output_file:
        import copy
        cs = copy.copy(c)                                                  # ecrad_ifs_driver.F90:426-428
        cs.do_surface_sw_spectral_flux = False
        cs.do_canopy_fluxes_sw = False
        cs.do_canopy_fluxes_lw = False
        save_net_fluxes(output_file, cs, thermodynamics, flux, is_double_precision=dc.do_write_double_precision,
                        experiment_name=dc.experiment_name, is_hdf5_file=dc.do_write_hdf5)
    return c, thermodynamics, flux, diag


def main(argv=None) -> int:
    import sys
    argv = sys.argv[1:] if argv is None else list(argv)
    blocked = "--blocked" in argv
    bitid = "--bitidentity" in argv
    argv = [a for a in argv if not a.startswith("--")]
    if len(argv) < 3:
        print("Usage: python -m ecrad_amd.ifs [--blocked] [--bitidentity] config.nam input_file.nc output_file.nc")
        return 1
    run_ifs_driver(argv[0], argv[1], argv[2], blocked=blocked, bitidentity=bitid)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
