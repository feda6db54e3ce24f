"""ctypes mirror of include/ecrad_hip.h (struct layouts must match; tests check sizeof via
``ecrad_hip_abi_sizeof``).  This is the Python stand-in for the ISO_C_BINDING layer a Fortran host
uses (ecrad_amd/fortran/ecrad_hip_binding.F90 is the Fortran twin of this file)."""
from __future__ import annotations

import ctypes as C

import numpy as np

ABI_VERSION = 8
NMAXGASES = 12
NMAXCLOUDTYPES = 12

MEM_HOST, MEM_DEVICE = 0, 1
COMM_ID_BYTES = 128
OK, EINVAL, ENODEVICE, EUNSUPPORTED, EHIP, ENOMEM, ENOTSETUP = 0, -1, -2, -3, -4, -5, -6      # include/ecrad_hip.h:42-48

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class CkdGas(C.Structure):
    _fields_ = [
        ("i_gas_code", C.c_int32), ("i_conc_dependence", C.c_int32),
        ("n_mole_frac", C.c_int32), ("reserved_", C.c_int32),
        ("reference_mole_frac", C.c_double), ("log_mole_frac1", C.c_double),
        ("d_log_mole_frac", C.c_double),
        ("molar_abs", c_double_p),
    ]


class CkdModel(C.Structure):
    _fields_ = [
        ("is_sw", C.c_int32), ("ng", C.c_int32), ("npress", C.c_int32), ("ntemp", C.c_int32),
        ("ngas", C.c_int32), ("nplanck", C.c_int32),
        ("log_pressure1", C.c_double), ("d_log_pressure", C.c_double), ("d_temperature", C.c_double),
        ("temperature1_planck", C.c_double), ("d_temperature_planck", C.c_double),
        ("temperature1", c_double_p), ("planck_function", c_double_p),
        ("norm_solar_irradiance", c_double_p), ("norm_amplitude_solar_irradiance", c_double_p),
        ("rayleigh_molar_scat", c_double_p),
        ("single_gas", CkdGas * NMAXGASES),
    ]


class CloudOptics(C.Structure):
    _fields_ = [
        ("n_bands", C.c_int32), ("n_effective_radius", C.c_int32),
        ("effective_radius_0", C.c_double), ("d_effective_radius", C.c_double),
        ("mass_ext", c_double_p), ("ssa", c_double_p), ("asymmetry", c_double_p),
    ]


class AerosolOptics(C.Structure):
    _fields_ = [
        ("n_bands_sw", C.c_int32), ("n_bands_lw", C.c_int32),
        ("n_type_phobic", C.c_int32), ("n_type_philic", C.c_int32), ("nrh", C.c_int32),
        ("use_hydrophilic", C.c_int32), ("ntype", C.c_int32), ("reserved_", C.c_int32),
        ("iclass", c_int32_p), ("itype", c_int32_p), ("rh_lower", c_double_p),
        ("mass_ext_sw_phobic", c_double_p), ("ssa_sw_phobic", c_double_p), ("g_sw_phobic", c_double_p),
        ("mass_ext_lw_phobic", c_double_p), ("ssa_lw_phobic", c_double_p), ("g_lw_phobic", c_double_p),
        ("mass_ext_sw_philic", c_double_p), ("ssa_sw_philic", c_double_p), ("g_sw_philic", c_double_p),
        ("mass_ext_lw_philic", c_double_p), ("ssa_lw_philic", c_double_p), ("g_lw_philic", c_double_p),
    ]


class PdfSampler(C.Structure):
    _fields_ = [
        ("ncdf", C.c_int32), ("nfsd", C.c_int32),
        ("fsd1", C.c_double), ("inv_fsd_interval", C.c_double),
        ("val", c_double_p),
    ]


class RrtmgBand(C.Structure):
    _fields_ = [
        ("ng", C.c_int32), ("ld", C.c_int32), ("nspa", C.c_int32), ("nspb", C.c_int32),
        ("layreffr", C.c_int32), ("n_forref", C.c_int32),
        ("strrat", C.c_double), ("rayl", C.c_double), ("factor", C.c_double),
        ("absa", c_double_p), ("absb", c_double_p), ("selfref", c_double_p), ("forref", c_double_p),
        ("fracrefa", c_double_p), ("fracrefb", c_double_p),
        ("minor", c_double_p * 6), ("xsec", c_double_p * 2), ("rayl_g", c_double_p * 2),
    ]


class Rrtmg(C.Structure):
    _fields_ = [
        ("chi_mls", c_double_p), ("preflog_lw", c_double_p), ("tref_lw", c_double_p),
        ("preflog_sw", c_double_p), ("tref_sw", c_double_p), ("totplnk", c_double_p), ("delwave", c_double_p),
        ("lw", RrtmgBand * 16), ("sw", RrtmgBand * 14),
        ("i_g_from_reordered_g_lw", c_int32_p), ("i_g_from_reordered_g_sw", c_int32_p),
    ]


_CONFIG_INTS = [
    "abi_version",
    "do_sw", "do_lw", "do_clear", "do_sw_direct", "do_lw_derivatives",
    "do_clouds", "use_aerosols",
    "i_solver_sw", "i_solver_lw",
    "i_gas_model_sw", "i_gas_model_lw",
    "do_lw_cloud_scattering", "do_lw_aerosol_scattering",
    "do_sw_delta_scaling_with_gases",
    "use_general_cloud_optics", "is_homogeneous",
    "i_overlap_scheme", "use_beta_overlap", "use_vectorizable_generator", "i_cloud_pdf_shape",
    "do_cloud_aerosol_per_sw_g_point", "do_cloud_aerosol_per_lw_g_point",
    "do_surface_sw_spectral_flux", "do_toa_spectral_flux",
    "do_canopy_fluxes_sw", "do_canopy_fluxes_lw",
    "use_canopy_full_spectrum_sw", "use_canopy_full_spectrum_lw",
    "do_nearest_spectral_sw_albedo", "do_nearest_spectral_lw_emiss",
    "do_save_spectral_flux",
    "n_g_sw", "n_g_lw", "n_bands_sw", "n_bands_lw",
    "n_g_lw_if_scattering", "n_bands_lw_if_scattering",
    "n_canopy_bands_sw", "n_canopy_bands_lw",
    "n_albedo_intervals_sw", "n_emiss_intervals_lw",
    "n_cloud_types", "reserved_", "n_spec_sw", "n_spec_lw",
]


class Config(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in _CONFIG_INTS]
        + [("cloud_fraction_threshold", C.c_double), ("cloud_mixing_ratio_threshold", C.c_double),
           ("cloud_inhom_decorr_scaling", C.c_double), ("max_cloud_od", C.c_double),
           ("i_band_from_reordered_g_sw", c_int32_p), ("i_band_from_reordered_g_lw", c_int32_p),
           ("sw_albedo_weights", c_double_p), ("lw_emiss_weights", c_double_p),
           ("i_albedo_from_band_sw", c_int32_p), ("i_emiss_from_band_lw", c_int32_p),
           ("i_spec_from_reordered_g_sw", c_int32_p), ("i_spec_from_reordered_g_lw", c_int32_p),
           ("gas_optics_sw", CkdModel), ("gas_optics_lw", CkdModel),
           ("cloud_optics_sw", CloudOptics * NMAXCLOUDTYPES),
           ("cloud_optics_lw", CloudOptics * NMAXCLOUDTYPES),
           ("aerosol_optics", AerosolOptics),
           ("pdf_sampler", PdfSampler),
           ("rrtmg", C.POINTER(Rrtmg)), ("min_gas_od_lw", C.c_double), ("min_gas_od_sw", C.c_double),
           ("i_liq_model", C.c_int32), ("i_ice_model", C.c_int32),
           ("do_fu_lw_ice_optics_bug", C.c_int32), ("reserved2_", C.c_int32),
           ("nregions", C.c_int32), ("i_3d_sw_entrapment", C.c_int32), ("do_3d_effects", C.c_int32),
           ("do_3d_lw_multilayer_effects", C.c_int32), ("do_lw_side_emissivity", C.c_int32),
           ("use_expm_everywhere", C.c_int32), ("i_precision", C.c_int32), ("reserved3_", C.c_int32),
           ("max_3d_transfer_rate", C.c_double), ("max_gas_od_3d", C.c_double),
           ("min_cloud_effective_size", C.c_double), ("overhang_factor", C.c_double),
           ("clear_to_thick_fraction", C.c_double), ("overhead_sun_factor", C.c_double)]
    )


class Inputs(C.Structure):
    _fields_ = [
        ("memory", C.c_int32), ("n_sw_albedo", C.c_int32), ("n_lw_emissivity", C.c_int32),
        ("n_cloud_types", C.c_int32), ("n_aerosol_types", C.c_int32),
        ("aerosol_istartlev", C.c_int32), ("aerosol_iendlev", C.c_int32), ("reserved_", C.c_int32),
        ("solar_irradiance", C.c_double), ("spectral_solar_cycle_multiplier", C.c_double),
        ("pressure_hl", c_double_p), ("temperature_hl", c_double_p), ("h2o_sat_liq", c_double_p),
        ("cos_sza", c_double_p), ("skin_temperature", c_double_p),
        ("sw_albedo", c_double_p), ("sw_albedo_direct", c_double_p), ("lw_emissivity", c_double_p),
        ("iseed", c_int32_p),
        ("gas_mixing_ratio", c_double_p),
        ("cloud_fraction", c_double_p), ("cloud_mixing_ratio", c_double_p),
        ("cloud_effective_radius", c_double_p), ("cloud_fractional_std", c_double_p),
        ("cloud_overlap_param", c_double_p),
        ("aerosol_mixing_ratio", c_double_p),
        ("cloud_inv_cloud_effective_size", c_double_p), ("cloud_inv_inhom_effective_size", c_double_p),
        ("spectral_solar_scaling", c_double_p),      # host memory always
    ]


FLUX_PROFILE_FIELDS = ["lw_up", "lw_dn", "sw_up", "sw_dn", "sw_dn_direct",
                       "lw_up_clear", "lw_dn_clear", "sw_up_clear", "sw_dn_clear", "sw_dn_direct_clear",
                       "lw_derivatives"]
FLUX_G_FIELDS = ["lw_dn_surf_g", "lw_dn_surf_clear_g", "sw_dn_diffuse_surf_g", "sw_dn_direct_surf_g",
                 "sw_dn_diffuse_surf_clear_g", "sw_dn_direct_surf_clear_g",
                 "lw_up_toa_g", "lw_up_toa_clear_g", "sw_dn_toa_g", "sw_up_toa_g", "sw_up_toa_clear_g"]
FLUX_BAND_FIELDS = ["sw_dn_surf_band", "sw_dn_direct_surf_band", "sw_dn_surf_clear_band",
                    "sw_dn_direct_surf_clear_band", "lw_up_toa_band", "lw_up_toa_clear_band",
                    "sw_dn_toa_band", "sw_up_toa_band", "sw_up_toa_clear_band"]
FLUX_CANOPY_FIELDS = ["lw_dn_surf_canopy", "sw_dn_diffuse_surf_canopy", "sw_dn_direct_surf_canopy"]
FLUX_COL_FIELDS = ["cloud_cover_lw", "cloud_cover_sw"]
# (nspec, ncol, nlev+1) spectral flux profiles (do_save_spectral_flux)
FLUX_SPEC_LW_FIELDS = ["lw_up_band", "lw_dn_band", "lw_up_clear_band", "lw_dn_clear_band"]
FLUX_SPEC_SW_FIELDS = ["sw_up_band", "sw_dn_band", "sw_dn_direct_band",
                       "sw_up_clear_band", "sw_dn_clear_band", "sw_dn_direct_clear_band"]
FLUX_SPEC_FIELDS = FLUX_SPEC_LW_FIELDS + FLUX_SPEC_SW_FIELDS
FLUX_FIELDS = (FLUX_PROFILE_FIELDS + FLUX_G_FIELDS + FLUX_BAND_FIELDS + FLUX_CANOPY_FIELDS + FLUX_COL_FIELDS
               + FLUX_SPEC_FIELDS)


class Flux(C.Structure):
    _fields_ = [("memory", C.c_int32), ("reserved_", C.c_int32)] + [(n, c_double_p) for n in FLUX_FIELDS]


OPTICS_FIELDS = ["od_lw", "ssa_lw", "g_lw", "od_sw", "ssa_sw", "g_sw", "planck_hl",
                 "lw_emission", "lw_albedo", "sw_albedo_direct", "sw_albedo_diffuse", "incoming_sw",
                 "od_lw_cloud", "ssa_lw_cloud", "g_lw_cloud", "od_sw_cloud", "ssa_sw_cloud", "g_sw_cloud"]


class Optics(C.Structure):
    _fields_ = [("memory", C.c_int32), ("reserved_", C.c_int32)] + [(n, c_double_p) for n in OPTICS_FIELDS]


class CallInfo(C.Structure):
    """ecrad_call_info_t"""
    _fields_ = [("n_tiles", C.c_int32), ("tile_columns", C.c_int32), ("launches_lw", C.c_int32),
                ("launches_sw", C.c_int32), ("lanes_lw", C.c_int32), ("lanes_sw", C.c_int32),
                ("work_bytes", C.c_size_t), ("staged_in_bytes", C.c_size_t), ("staged_out_bytes", C.c_size_t)]


MAX_POOL_DEVICES = 16


class PoolInfo(C.Structure):
    """ecrad_pool_info_t"""
    _fields_ = [("n_devices", C.c_int32), ("n_contexts", C.c_int32), ("in_flight", C.c_int32), ("max_in_flight", C.c_int32),
                ("calls_total", C.c_int64), ("batches_total", C.c_int64), ("device_ids", C.c_int32 * MAX_POOL_DEVICES),
                ("calls_on_device", C.c_int64 * MAX_POOL_DEVICES)]


STRUCT_BY_INDEX = [Config, Inputs, Flux, Optics, CkdModel, CkdGas, CloudOptics, AerosolOptics, PdfSampler,
                   Rrtmg, RrtmgBand]


# ---------------------------------------------------------------------------------------------------
def dptr(a):
    """Pointer to a float64 C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], \
        "expected C-contiguous float64 array"
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.int32 and a.flags["C_CONTIGUOUS"], \
        "expected C-contiguous int32 array"
    return a.ctypes.data_as(c_int32_p)


def raw_dptr(addr: int):
    """Device address (e.g. torch.Tensor.data_ptr()) as a double*."""
    return C.cast(C.c_void_p(addr), c_double_p) if addr else None


def raw_iptr(addr: int):
    return C.cast(C.c_void_p(addr), c_int32_p) if addr else None


def declare_prototypes(lib) -> None:
    """Set argtypes/restype for every entry point include/ecrad_hip.h declares."""
    H = C.c_void_p
    lib.ecrad_hip_create.argtypes = [C.POINTER(H), C.c_int]
    lib.ecrad_hip_create.restype = C.c_int
    lib.ecrad_hip_setup.argtypes = [H, C.POINTER(Config)]
    lib.ecrad_hip_setup.restype = C.c_int
    lib.ecrad_hip_set_stream.argtypes = [H, C.c_void_p]
    lib.ecrad_hip_set_stream.restype = C.c_int
    lib.ecrad_hip_radiation.argtypes = [H, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(Inputs), C.POINTER(Flux)]
    lib.ecrad_hip_radiation.restype = C.c_int
    lib.ecrad_hip_radiation_f32.argtypes = lib.ecrad_hip_radiation.argtypes      # (same structs, float arrays behind the pointers)
    lib.ecrad_hip_radiation_f32.restype = C.c_int
    lib.ecrad_hip_comm_id.argtypes = [H, C.c_void_p]
    lib.ecrad_hip_comm_id.restype = C.c_int
    lib.ecrad_hip_comm_init.argtypes = [H, C.c_void_p, C.c_int, C.c_int]
    lib.ecrad_hip_comm_init.restype = C.c_int
    lib.ecrad_hip_gather_profiles.argtypes = [H, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    lib.ecrad_hip_gather_profiles.restype = C.c_int
    lib.ecrad_hip_comm_destroy.argtypes = [H]
    lib.ecrad_hip_comm_destroy.restype = C.c_int
    lib.ecrad_hip_host_alloc.argtypes = [H, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.ecrad_hip_host_alloc.restype = C.c_int
    lib.ecrad_hip_host_free.argtypes = [H, C.c_void_p]
    lib.ecrad_hip_host_free.restype = C.c_int
    lib.ecrad_hip_host_register.argtypes = [H, C.c_void_p, C.c_size_t]
    lib.ecrad_hip_host_register.restype = C.c_int
    lib.ecrad_hip_host_unregister.argtypes = [H, C.c_void_p]
    lib.ecrad_hip_host_unregister.restype = C.c_int
    lib.ecrad_hip_optics.argtypes = [H, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(Inputs), C.POINTER(Optics)]
    lib.ecrad_hip_optics.restype = C.c_int
    lib.ecrad_hip_synchronize.argtypes = [H]
    lib.ecrad_hip_synchronize.restype = C.c_int
    lib.ecrad_hip_last_kernel_ms.argtypes = [H, C.POINTER(C.c_double)]
    lib.ecrad_hip_last_kernel_ms.restype = C.c_int
    lib.ecrad_hip_last_stage_ms.argtypes = [H, C.c_int, C.POINTER(C.c_double)]
    lib.ecrad_hip_last_stage_ms.restype = C.c_int
    lib.ecrad_hip_hbm_triad.argtypes = [H, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    lib.ecrad_hip_hbm_triad.restype = C.c_int
    lib.ecrad_hip_hbm_rates.argtypes = [H, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ecrad_hip_hbm_rates.restype = C.c_int
    lib.ecrad_hip_scratch_bytes.argtypes = [H, C.POINTER(C.c_size_t)]
    lib.ecrad_hip_scratch_bytes.restype = C.c_int
    lib.ecrad_hip_last_error.argtypes = [H]
    lib.ecrad_hip_last_error.restype = C.c_char_p
    lib.ecrad_hip_destroy.argtypes = [H]
    lib.ecrad_hip_destroy.restype = C.c_int
    lib.ecrad_hip_abi_sizeof.argtypes = [C.c_int]
    lib.ecrad_hip_abi_sizeof.restype = C.c_size_t
    lib.ecrad_hip_abi_version.argtypes = []
    lib.ecrad_hip_abi_version.restype = C.c_int
    lib.ecrad_hip_set_work_bytes.argtypes = [H, C.c_size_t]
    lib.ecrad_hip_set_work_bytes.restype = C.c_int
    lib.ecrad_hip_last_call_info.argtypes = [H, C.POINTER(CallInfo)]
    lib.ecrad_hip_last_call_info.restype = C.c_int
    lib.ecrad_hip_pcie_bandwidth.argtypes = [H, C.c_size_t, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ecrad_hip_pcie_bandwidth.restype = C.c_int
    lib.ecrad_hip_set_concurrency.argtypes = [H, C.c_int, C.c_int]
    lib.ecrad_hip_set_concurrency.restype = C.c_int
    lib.ecrad_hip_pool_info.argtypes = [H, C.POINTER(PoolInfo)]
    lib.ecrad_hip_pool_info.restype = C.c_int
    lib.ecrad_hip_pool_reset.argtypes = [H]
    lib.ecrad_hip_pool_reset.restype = C.c_int


EXPORTED_SYMBOLS = [
    "ecrad_hip_create", "ecrad_hip_setup", "ecrad_hip_set_stream", "ecrad_hip_radiation",
    "ecrad_hip_optics", "ecrad_hip_synchronize", "ecrad_hip_last_kernel_ms", "ecrad_hip_last_stage_ms",
    "ecrad_hip_scratch_bytes", "ecrad_hip_last_error", "ecrad_hip_destroy",
    "ecrad_hip_abi_sizeof", "ecrad_hip_abi_version", "ecrad_hip_set_work_bytes", "ecrad_hip_last_call_info",
    "ecrad_hip_hbm_triad", "ecrad_hip_hbm_rates", "ecrad_hip_set_concurrency", "ecrad_hip_pool_info", "ecrad_hip_pool_reset",
    "ecrad_hip_pcie_bandwidth", "ecrad_hip_radiation_f32", "ecrad_hip_comm_id", "ecrad_hip_comm_init", "ecrad_hip_gather_profiles", "ecrad_hip_comm_destroy", "ecrad_hip_host_alloc", "ecrad_hip_host_free", "ecrad_hip_host_register", "ecrad_hip_host_unregister",
]
