"""Tagged binary "case file" used to hand a fully set-up configuration + one set of columns to the
Fortran host driver (ecrad_amd/fortran/ecrad_hip_driver.F90), and to read its fluxes back.

Record = name (48 bytes, space padded) | dtype int32 (0 = int32, 1 = float64) | rank int32 |
dims 4 x int64 (Fortran order, unused = 1) | data (Fortran order == our numpy C-order bytes).
The Fortran side keeps table preparation in principle (setup_radiation); until its netCDF/spectral
mapping layer is written this file carries the mapped tables produced by ecrad_amd.setup_radiation.
"""
from __future__ import annotations

import struct

import numpy as np

CONFIG_INT_FIELDS = [
    "do_sw", "do_lw", "do_clear", "do_sw_direct", "do_lw_derivatives", "do_clouds", "use_aerosols",
    "i_solver_sw", "i_solver_lw", "i_gas_model_sw", "i_gas_model_lw", "do_lw_cloud_scattering",
    "do_lw_aerosol_scattering", "do_sw_delta_scaling_with_gases", "is_homogeneous", "i_overlap_scheme",
    "i_cloud_pdf_shape", "use_beta_overlap", "use_vectorizable_generator", "do_cloud_aerosol_per_sw_g_point",
    "do_cloud_aerosol_per_lw_g_point", "do_surface_sw_spectral_flux", "do_toa_spectral_flux",
    "do_canopy_fluxes_sw", "do_canopy_fluxes_lw", "use_canopy_full_spectrum_sw", "use_canopy_full_spectrum_lw",
    "do_nearest_spectral_sw_albedo", "do_nearest_spectral_lw_emiss", "n_g_sw", "n_g_lw", "n_bands_sw",
    "n_bands_lw", "n_canopy_bands_sw", "n_canopy_bands_lw", "n_cloud_types",
    "use_general_cloud_optics", "i_liq_model", "i_ice_model", "do_fu_lw_ice_optics_bug", "n_g_lw_if_scattering",
    "n_bands_lw_if_scattering", "nregions", "i_3d_sw_entrapment", "do_3d_effects", "do_3d_lw_multilayer_effects",
    "do_lw_side_emissivity", "use_expm_everywhere",
]
CONFIG_REAL_FIELDS = ["cloud_fraction_threshold", "cloud_mixing_ratio_threshold", "cloud_inhom_decorr_scaling", "max_cloud_od",
                      "min_gas_od_lw", "min_gas_od_sw", "max_3d_transfer_rate", "max_gas_od_3d", "min_cloud_effective_size",
                      "overhang_factor", "clear_to_thick_fraction", "overhead_sun_factor"]


def _rec(f, name, arr):
    arr = np.asarray(arr)
    if arr.dtype.kind in "iub":
        arr = arr.astype(np.int32)
        code = 0
    else:
        arr = arr.astype(np.float64)
        code = 1
    arr = np.ascontiguousarray(arr)
    dims = list(arr.shape[::-1]) if arr.ndim else [1]
    rank = len(dims)
    dims = dims + [1] * (4 - rank)
    f.write(name.encode().ljust(48))
    f.write(struct.pack("<ii4q", code, rank, *dims))
    f.write(arr.tobytes())


def write_case(path, config, ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol, istartcol=1, iendcol=None):
    iendcol = iendcol or ncol
    with open(path, "wb") as f:
        _rec(f, "config.ints", [int(getattr(config, k, 0) or 0) for k in CONFIG_INT_FIELDS])
        _rec(f, "config.reals", [float(getattr(config, k, 0.0) or 0.0) for k in CONFIG_REAL_FIELDS])
        if config.do_sw:
            _rec(f, "config.i_band_from_reordered_g_sw", config.i_band_from_reordered_g_sw)
            if getattr(config, "sw_albedo_weights", None) is not None:
                _rec(f, "config.sw_albedo_weights", config.sw_albedo_weights)
            if getattr(config, "i_albedo_from_band_sw", None) is not None:
                _rec(f, "config.i_albedo_from_band_sw", config.i_albedo_from_band_sw)
        if config.do_lw:
            _rec(f, "config.i_band_from_reordered_g_lw", config.i_band_from_reordered_g_lw)
            if getattr(config, "lw_emiss_weights", None) is not None:
                _rec(f, "config.lw_emiss_weights", config.lw_emiss_weights)
            if getattr(config, "i_emiss_from_band_lw", None) is not None:
                _rec(f, "config.i_emiss_from_band_lw", config.i_emiss_from_band_lw)
        for tag, m in (("gas_sw", config.gas_optics_sw), ("gas_lw", config.gas_optics_lw)):
            if m is None or getattr(config, "rrtmg", None) is not None:
                continue
            _rec(f, f"{tag}.ints", [int(m.is_sw), m.ng, m.npress, m.ntemp, m.ngas, m.nplanck])
            _rec(f, f"{tag}.reals", [m.log_pressure1, m.d_log_pressure, m.d_temperature,
                                     m.temperature1_planck, m.d_temperature_planck])
            _rec(f, f"{tag}.temperature1", m.temperature1)
            if m.is_sw:
                _rec(f, f"{tag}.norm_solar_irradiance", m.norm_solar_irradiance)
                _rec(f, f"{tag}.rayleigh_molar_scat", m.rayleigh_molar_scat)
                if getattr(m, "norm_amplitude_solar_irradiance", None) is not None:
                    _rec(f, f"{tag}.norm_amplitude_solar_irradiance", m.norm_amplitude_solar_irradiance)
            else:
                _rec(f, f"{tag}.planck_function", m.planck_function)
            for j, g in enumerate(m.single_gas, start=1):
                _rec(f, f"{tag}.gas{j:02d}.ints", [g.i_gas_code, g.i_conc_dependence, g.n_mole_frac])
                _rec(f, f"{tag}.gas{j:02d}.reals", [g.reference_mole_frac, g.log_mole_frac1, g.d_log_mole_frac])
                # radiation_ecckd_gas.F90:54,59: molar_abs(ng,np,nt), or molar_abs_conc(ng,np,nt,nconc) for a look-up table
                _rec(f, f"{tag}.gas{j:02d}." + ("molar_abs_conc" if np.ndim(g.molar_abs) == 4 else "molar_abs"), g.molar_abs)
        for tag, lst in (("cloud_sw", config.cloud_optics_sw), ("cloud_lw", config.cloud_optics_lw)):
            for j, co in enumerate(lst or [], start=1):
                _rec(f, f"{tag}.{j:02d}.ints", [co.n_bands, co.n_effective_radius])
                _rec(f, f"{tag}.{j:02d}.reals", [co.effective_radius_0, co.d_effective_radius])
                _rec(f, f"{tag}.{j:02d}.mass_ext", co.mass_ext)
                if co.ssa is not None:
                    _rec(f, f"{tag}.{j:02d}.ssa", co.ssa)
                if co.asymmetry is not None:
                    _rec(f, f"{tag}.{j:02d}.asymmetry", co.asymmetry)
        ao = config.aerosol_optics
        if config.use_aerosols and ao is not None:
            _rec(f, "aerosol.ints", [ao.n_bands_sw, ao.n_bands_lw, ao.n_type_phobic, ao.n_type_philic, ao.nrh,
                                     int(ao.use_hydrophilic), ao.ntype])
            _rec(f, "aerosol.iclass", ao.iclass)
            _rec(f, "aerosol.itype", ao.itype)
            _rec(f, "aerosol.rh_lower", ao.rh_lower)
            for t in ("sw", "lw"):
                for kind in ("phobic", "philic"):
                    for qn in ("mass_ext", "ssa", "g"):
                        n = f"{qn}_{t}_{kind}"
                        if getattr(ao, n, None) is not None:
                            _rec(f, "aerosol." + n, getattr(ao, n))
        ps = config.pdf_sampler
        if ps is not None:
            _rec(f, "pdf.ints", [ps.ncdf, ps.nfsd])
            _rec(f, "pdf.reals", [ps.fsd1, ps.inv_fsd_interval])
            _rec(f, "pdf.val", ps.val)
        nct = cloud.ntype if cloud is not None else 0
        nat = aerosol.mixing_ratio.shape[0] if aerosol is not None else 0
        _rec(f, "inputs.ints", [ncol, nlev, istartcol, iendcol, nct, nat,
                                aerosol.istartlev if aerosol is not None else 1,
                                aerosol.iendlev if aerosol is not None else 0])
        _rec(f, "inputs.reals", [single_level.solar_irradiance, single_level.spectral_solar_cycle_multiplier])
        _rec(f, "inputs.pressure_hl", thermodynamics.pressure_hl)
        _rec(f, "inputs.temperature_hl", thermodynamics.temperature_hl)
        if thermodynamics.h2o_sat_liq is not None:
            _rec(f, "inputs.h2o_sat_liq", thermodynamics.h2o_sat_liq)
        _rec(f, "inputs.cos_sza", single_level.cos_sza)
        _rec(f, "inputs.skin_temperature", single_level.skin_temperature)
        _rec(f, "inputs.sw_albedo", single_level.sw_albedo)
        if single_level.sw_albedo_direct is not None:
            _rec(f, "inputs.sw_albedo_direct", single_level.sw_albedo_direct)
        _rec(f, "inputs.lw_emissivity", single_level.lw_emissivity)
        if single_level.iseed is not None:
            _rec(f, "inputs.iseed", single_level.iseed)
        _rec(f, "inputs.gas_mixing_ratio", gas.mixing_ratio)
        if cloud is not None:
            _rec(f, "inputs.cloud_fraction", cloud.fraction)
            _rec(f, "inputs.cloud_mixing_ratio", cloud.mixing_ratio)
            _rec(f, "inputs.cloud_effective_radius", cloud.effective_radius)
            _rec(f, "inputs.cloud_fractional_std", cloud.fractional_std)
            _rec(f, "inputs.cloud_overlap_param", cloud.overlap_param)
            if getattr(cloud, "inv_cloud_effective_size", None) is not None:
                _rec(f, "inputs.cloud_inv_cloud_effective_size", cloud.inv_cloud_effective_size)
            if getattr(cloud, "inv_inhom_effective_size", None) is not None:
                _rec(f, "inputs.cloud_inv_inhom_effective_size", cloud.inv_inhom_effective_size)
        if aerosol is not None:
            _rec(f, "inputs.aerosol_mixing_ratio", aerosol.mixing_ratio)
        _rec(f, "end", [0])


def read_records(path) -> dict:
    out = {}
    with open(path, "rb") as f:
        while True:
            head = f.read(48)
            if len(head) < 48:
                break
            name = head.decode().strip()
            code, rank, *dims = struct.unpack("<ii4q", f.read(40))
            n = int(np.prod(dims[:rank]))
            dt = np.int32 if code == 0 else np.float64
            data = np.frombuffer(f.read(n * np.dtype(dt).itemsize), dtype=dt)
            out[name] = data.reshape(dims[:rank][::-1]).copy()
    return out
