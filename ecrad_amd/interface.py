"""Host mirror of radiation/radiation_interface.F90: ``setup_radiation``, ``set_gas_units`` and
``radiation`` with the reference's argument meaning, implemented over the C-ABI of
include/ecrad_hip.h (libecrad_hip.so: hand-written HIP kernels for gfx950).

There is deliberately no CPU path here: if the HIP library cannot be loaded or no GPU is present,
construction of :class:`Radiation` with ``backend="hip"`` raises.  (tests/ and bench.py's
``cpu_baseline`` leg drive the *same marshalling* into oracle/ through ``backend=<callable>``.)
"""
from __future__ import annotations

import ctypes as C
from typing import Optional
import os

import numpy as np

from . import abi
from .config import (Config, ConfigError, IGasModelECCKD, IGasModelIFSRRTMG, IGasModelMonochromatic,
                     ISolverMcICA, ISolverSpartacus, NMaxAlbedoIntervals)
from .spectral import SOLAR_REFERENCE_TEMPERATURE, TERRESTRIAL_REFERENCE_TEMPERATURE
from .tables import AerosolOptics, CkdModel, GeneralCloudOptics, PdfSampler
from .types import Flux, IVolumeMixingRatio

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ECRAD_HIP_LIB", os.path.join(_HERE, "csrc", "libecrad_hip.so"))


class EcradHipError(RuntimeError):
    pass


class HostArrays:
    """Page-locked host memory for a Python host (include/ecrad_hip.h: ecrad_hip_host_alloc / ecrad_hip_host_free): numpy arrays whose
    storage the library allocated page-locked, so that the tiles of a pipelined host-memory call move by the copy engines directly.
    ``empty`` / ``copy_of`` hand out arrays; ``close`` gives the memory back (no array handed out may be used after it).  A host model
    allocates its long-lived arrays here once; nothing is registered or guessed about the allocator (round 5 registered numpy arrays
    where glibc happened to put them: profiles/NOTES_r06.md section 1)."""

    def __init__(self, rad: "Radiation"):
        self.lib, self.handle = rad.lib, rad.handle
        self._blocks = []
        self.nbytes = 0

    def empty(self, shape, dtype=np.float64):
        dtype = np.dtype(dtype)
        count = int(np.prod(shape, dtype=np.int64))
        nbytes = max(count * dtype.itemsize, 1)
        p = C.c_void_p()
        st = self.lib.ecrad_hip_host_alloc(self.handle, C.c_size_t(nbytes), C.byref(p))
        if st != 0 or not p.value:
            raise EcradHipError(f"ecrad_hip_host_alloc({nbytes}) failed with status {st}")
        self._blocks.append(p.value)
        self.nbytes += nbytes
        buf = (C.c_char * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)

    def copy_of(self, a):
        b = self.empty(a.shape, a.dtype)
        b[...] = a
        return b

    def close(self):
        for p in self._blocks:
            self.lib.ecrad_hip_host_free(self.handle, C.c_void_p(p))
        self._blocks = []
        self.nbytes = 0


def page_aligned_empty(shape, dtype=np.float64):
    """A numpy array that is a private mapping of whole pages (``mmap``), i.e. a range ecrad_hip_host_register accepts: returns
    (array, address, bytes) with `bytes` the length of the mapping, a multiple of the page size.  The mapping lives as long as the array."""
    import mmap
    dtype = np.dtype(dtype)
    count = int(np.prod(shape, dtype=np.int64))
    nbytes = -(-max(count * dtype.itemsize, 1) // mmap.PAGESIZE) * mmap.PAGESIZE
    m = mmap.mmap(-1, nbytes)
    a = np.frombuffer(m, dtype=dtype, count=count).reshape(shape)
    return a, a.ctypes.data, nbytes


def relocate_call_arrays(copy_of, objects, flux=None, min_bytes=1 << 16):
    """Move the numpy arrays of a call -- the ndarray members of `objects` (single_level, thermodynamics, gas, cloud, aerosol; None
    entries skipped) and the arrays of `flux` -- of `min_bytes` or more into memory that `copy_of(array)` provides (HostArrays.copy_of:
    page-locked; a wrapper of page_aligned_empty: whole pages for ecrad_hip_host_register).  In place: the objects then hold the new
    arrays.  Returns the list of new arrays."""
    moved = []
    for obj in objects:
        if obj is None:
            continue
        for k, v in list(vars(obj).items()):
            if isinstance(v, np.ndarray) and v.nbytes >= min_bytes:
                setattr(obj, k, copy_of(np.ascontiguousarray(v)))
                moved.append(getattr(obj, k))
    if flux is not None:
        for k, v in list(flux.arrays.items()):
            if v.nbytes >= min_bytes:
                flux.arrays[k] = copy_of(v)
                moved.append(flux.arrays[k])
    return moved


def load_library(path: str = LIB_PATH):
    """dlopen libecrad_hip.so and declare its prototypes; raises if it has not been built."""
    if not os.path.exists(path):
        raise EcradHipError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
    lib = C.CDLL(path)
    abi.declare_prototypes(lib)
    if lib.ecrad_hip_abi_version() != abi.ABI_VERSION:
        raise EcradHipError("libecrad_hip.so ABI version mismatch")
    for i, s in enumerate(abi.STRUCT_BY_INDEX):
        if lib.ecrad_hip_abi_sizeof(i) != C.sizeof(s):
            raise EcradHipError(f"struct layout mismatch for {s.__name__}: "
                                f"C {lib.ecrad_hip_abi_sizeof(i)} vs ctypes {C.sizeof(s)}")
    return lib


# ----------------------------------------------------------------------------------------------------
def setup_radiation(config: Config) -> None:
    """setup_radiation (radiation_interface.F90:37-158): read and map every look-up table."""
    config.consolidate()
    for m in (config.i_gas_model_sw, config.i_gas_model_lw):
        if m == IGasModelMonochromatic:
            raise ConfigError("the monochromatic gas model is not implemented in this build")
    if (config.do_sw and config.i_gas_model_sw == IGasModelIFSRRTMG) or (config.do_lw and config.i_gas_model_lw == IGasModelIFSRRTMG):
        return _setup_radiation_rrtmg(config)     # (RRTMG in at least one spectrum; the other may be ecCKD)
    # setup_gas_optics (radiation_ecckd_interface.F90:27-150)
    if config.do_sw:
        config.gas_optics_sw = CkdModel(config.gas_optics_sw_file_name)
        if config.use_spectral_solar_cycle:       # radiation_ecckd_interface.F90:78-82
            config.gas_optics_sw.read_spectral_solar_cycle(config.ssi_file_name, config.use_updated_solar_spectrum)
        config.n_g_sw = config.gas_optics_sw.ng
        if config.do_cloud_aerosol_per_sw_g_point:
            config.n_bands_sw = config.n_g_sw
            config.i_band_from_reordered_g_sw = np.arange(1, config.n_g_sw + 1, dtype=np.int32)
        else:
            config.n_bands_sw = config.gas_optics_sw.spectral_def.nband
            config.i_band_from_reordered_g_sw = config.gas_optics_sw.spectral_def.i_band_number.astype(np.int32)
    if config.do_lw:
        config.gas_optics_lw = CkdModel(config.gas_optics_lw_file_name)
        config.n_g_lw = config.gas_optics_lw.ng
        if config.do_cloud_aerosol_per_lw_g_point:
            config.n_bands_lw = config.n_g_lw
            config.i_band_from_reordered_g_lw = np.arange(1, config.n_g_lw + 1, dtype=np.int32)
        else:
            config.n_bands_lw = config.gas_optics_lw.spectral_def.nband
            config.i_band_from_reordered_g_lw = config.gas_optics_lw.spectral_def.i_band_number.astype(np.int32)
    # Spectral intervals of the flux profiles saved with do_save_spectral_flux: g-points or bands
    # (radiation_config.F90:1568-1590)
    # (radiation_ecckd_interface.F90:124-143)
    for sfx in ("sw", "lw"):
        if not getattr(config, "do_" + sfx) or not config.do_save_spectral_flux:
            setattr(config, "n_spec_" + sfx, 0)
            setattr(config, "i_spec_from_reordered_g_" + sfx, None)
            continue
        ng = getattr(config, "n_g_" + sfx)
        if config.do_save_gpoint_flux:
            setattr(config, "n_spec_" + sfx, ng)
            setattr(config, "i_spec_from_reordered_g_" + sfx, np.arange(1, ng + 1, dtype=np.int32))
        else:
            setattr(config, "n_spec_" + sfx, getattr(config, "n_bands_" + sfx))
            setattr(config, "i_spec_from_reordered_g_" + sfx,
                    np.asarray(getattr(config, "i_band_from_reordered_g_" + sfx), dtype=np.int32))

    if config.do_lw_aerosol_scattering and not config.do_lw_cloud_scattering:
        raise ConfigError("longwave aerosol scattering requires longwave cloud scattering")
    config.n_g_lw_if_scattering = config.n_g_lw if config.do_lw_aerosol_scattering else 0
    config.n_bands_lw_if_scattering = config.n_bands_lw if config.do_lw_cloud_scattering else 0
    if config.do_lw_cloud_scattering and config.i_solver_lw == ISolverMcICA:
        config.n_g_lw_if_scattering = config.n_g_lw

    # consolidate_sw_albedo_intervals / consolidate_lw_emiss_intervals (radiation_config.F90:1947-2100)
    def intervals(index, bound, full_spectrum, ng):
        ninterval = 0
        for j, v in enumerate(index[:NMaxAlbedoIntervals]):
            if v > 0:
                ninterval = j + 1
            else:
                break
        if ninterval < 1:
            idx, ncanopy = [1], (ng if full_spectrum else 1)
            ninterval = 1
        else:
            idx = list(index[:ninterval])
            ncanopy = ng if full_spectrum else max(idx)
        return idx, list(bound[:ninterval - 1]), ncanopy

    if config.do_sw:
        idx, bnd, config.n_canopy_bands_sw = intervals(config.i_sw_albedo_index, config.sw_albedo_wavelength_bound,
                                                       config.use_canopy_full_spectrum_sw, config.n_g_sw)
        config.sw_albedo_weights = np.ascontiguousarray(
            config.gas_optics_sw.spectral_def.calc_mapping_from_bands(
                bnd, idx, use_bands=not config.do_cloud_aerosol_per_sw_g_point))
        if config.do_nearest_spectral_sw_albedo:
            config.i_albedo_from_band_sw = (np.argmax(config.sw_albedo_weights, axis=1) + 1).astype(np.int32)
    if config.do_lw:
        idx, bnd, config.n_canopy_bands_lw = intervals(config.i_lw_emiss_index, config.lw_emiss_wavelength_bound,
                                                       config.use_canopy_full_spectrum_lw, config.n_g_lw)
        config.lw_emiss_weights = np.ascontiguousarray(
            config.gas_optics_lw.spectral_def.calc_mapping_from_bands(
                bnd, idx, use_bands=not config.do_cloud_aerosol_per_lw_g_point))
        if config.do_nearest_spectral_lw_emiss:
            config.i_emiss_from_band_lw = (np.argmax(config.lw_emiss_weights, axis=1) + 1).astype(np.int32)

    # setup_general_cloud_optics (radiation_general_cloud_optics.F90:32-128)
    if config.do_clouds:
        if not config.use_general_cloud_optics:
            raise ConfigError("ecCKD gas optics requires use_general_cloud_optics=true")
        names = [n for n in config.cloud_type_name if n]
        if not names:
            names = ["mie_droplet", "baum-general-habit-mixture_ice"]
        config.n_cloud_types = len(names)
        config.cloud_optics_sw, config.cloud_optics_lw = [], []
        for j, name in enumerate(names):
            if name.startswith("/"):
                fn = name
            elif name.endswith(".nc"):
                fn = os.path.join(config.directory_name, name)
            else:
                fn = os.path.join(config.directory_name, name + "_scattering.nc")
            thick = config.use_thick_cloud_spectral_averaging[j]
            if config.do_sw:
                config.cloud_optics_sw.append(GeneralCloudOptics(
                    fn, config.gas_optics_sw.spectral_def, not config.do_cloud_aerosol_per_sw_g_point,
                    thick, SOLAR_REFERENCE_TEMPERATURE, name))
            if config.do_lw:
                config.cloud_optics_lw.append(GeneralCloudOptics(
                    fn, config.gas_optics_lw.spectral_def, not config.do_cloud_aerosol_per_lw_g_point,
                    thick, TERRESTRIAL_REFERENCE_TEMPERATURE, name))

    # setup_aerosol_optics (radiation_aerosol_optics.F90:36-90)
    if config.use_aerosols:
        if config.n_aerosol_types > 0:
            if not config.use_general_aerosol_optics:
                raise ConfigError("band-wise aerosol files (use_general_aerosol_optics=false) hold the RRTMG bands: not for ecCKD")
            config.aerosol_optics = AerosolOptics(
                config.aerosol_optics_file_name,
                config.gas_optics_sw.spectral_def if config.do_sw else None,
                config.gas_optics_lw.spectral_def if config.do_lw else None,
                config.do_cloud_aerosol_per_sw_g_point, config.do_cloud_aerosol_per_lw_g_point,
                config.do_sw, config.do_lw)
            config.aerosol_optics.set_types(config.i_aerosol_type_map[:config.n_aerosol_types])
        else:
            config.use_aerosols = False   # "Aerosols are deactivated if this is zero"

    if ISolverMcICA in (config.i_solver_sw, config.i_solver_lw):
        config.pdf_sampler = PdfSampler(config.cloud_pdf_file_name)


class _BandsOnlyGasOptics:
    """What setup_radiation needs from a gas-optics model when it is RRTMG: the spectral definition of its bands
    (radiation_ifs_rrtm.F90:111-121, :155-164)."""

    def __init__(self, spectral_def):
        self.spectral_def = spectral_def


def _setup_radiation_rrtmg(config: Config) -> None:
    """setup_radiation with "RRTMG-IFS" as the gas model of at least one spectrum: setup_gas_optics of
    radiation_ifs_rrtm.F90:58-213 for the spectra that use it (and of radiation_ecckd_interface.F90:27-150 for one that
    uses ecCKD: sw_gas_model_name / lw_gas_model_name may differ, the reference's test_mixed_gas configurations), then
    the same surface-interval, cloud, aerosol and McICA set-up as above, with bands instead of g-points where RRTMG is."""
    from .rrtmg import (LW_WAVENUMBER1, LW_WAVENUMBER2, SW_WAVENUMBER1, SW_WAVENUMBER2, RrtmgTables)
    from .spectral import SpectralDefinition
    from .tables import BandFitCloudOptics
    config.rrtmg = RrtmgTables()
    rrtmg_sw = config.do_sw and config.i_gas_model_sw == IGasModelIFSRRTMG
    rrtmg_lw = config.do_lw and config.i_gas_model_lw == IGasModelIFSRRTMG
    sd_sw = sd_lw = None
    # SPARTACUS gets RRTMG's g-points in approximately increasing order of optical depth (radiation_ifs_rrtm.F90:122-130, :167-174)
    reorder_sw, reorder_lw = rrtmg_sw and config.i_solver_sw == ISolverSpartacus, rrtmg_lw and config.i_solver_lw == ISolverSpartacus
    config.rrtmg.set_reordering(reorder_lw, reorder_sw)
    if config.do_sw and rrtmg_sw:
        config.do_cloud_aerosol_per_sw_g_point = False
        sd_sw = SpectralDefinition.bands_only(SOLAR_REFERENCE_TEMPERATURE, SW_WAVENUMBER1, SW_WAVENUMBER2)
        config.n_g_sw, config.n_bands_sw = 112, 14
        config.i_band_from_reordered_g_sw = config.rrtmg.i_band_from_g_sw.copy()
        if reorder_sw:
            config.i_band_from_reordered_g_sw = config.rrtmg.i_band_from_g_sw[config.rrtmg.i_g_from_reordered_g_sw - 1].copy()
        config.gas_optics_sw = _BandsOnlyGasOptics(sd_sw)
    elif config.do_sw:
        config.gas_optics_sw = CkdModel(config.gas_optics_sw_file_name)
        sd_sw = config.gas_optics_sw.spectral_def
        config.n_g_sw = config.gas_optics_sw.ng
        if config.do_cloud_aerosol_per_sw_g_point:
            config.n_bands_sw = config.n_g_sw
            config.i_band_from_reordered_g_sw = np.arange(1, config.n_g_sw + 1, dtype=np.int32)
        else:
            config.n_bands_sw = sd_sw.nband
            config.i_band_from_reordered_g_sw = sd_sw.i_band_number.astype(np.int32)
    if config.do_lw and rrtmg_lw:
        config.do_cloud_aerosol_per_lw_g_point = False
        sd_lw = SpectralDefinition.bands_only(TERRESTRIAL_REFERENCE_TEMPERATURE, LW_WAVENUMBER1, LW_WAVENUMBER2)
        config.n_g_lw, config.n_bands_lw = 140, 16
        config.i_band_from_reordered_g_lw = config.rrtmg.i_band_from_g_lw.copy()
        if reorder_lw:
            config.i_band_from_reordered_g_lw = config.rrtmg.i_band_from_g_lw[config.rrtmg.i_g_from_reordered_g_lw - 1].copy()
        config.gas_optics_lw = _BandsOnlyGasOptics(sd_lw)
    elif config.do_lw:
        config.gas_optics_lw = CkdModel(config.gas_optics_lw_file_name)
        sd_lw = config.gas_optics_lw.spectral_def
        config.n_g_lw = config.gas_optics_lw.ng
        if config.do_cloud_aerosol_per_lw_g_point:
            config.n_bands_lw = config.n_g_lw
            config.i_band_from_reordered_g_lw = np.arange(1, config.n_g_lw + 1, dtype=np.int32)
        else:
            config.n_bands_lw = sd_lw.nband
            config.i_band_from_reordered_g_lw = sd_lw.i_band_number.astype(np.int32)
    mixed = (config.do_sw and not rrtmg_sw) or (config.do_lw and not rrtmg_lw)
    use_bands_sw, use_bands_lw = not config.do_cloud_aerosol_per_sw_g_point, not config.do_cloud_aerosol_per_lw_g_point
    if mixed and config.do_clouds and not config.use_general_cloud_optics:
        raise ConfigError("the per-band cloud-optics fits exist for the RRTMG bands only: an ecCKD spectrum needs use_general_cloud_optics=true")
    for sfx in ("sw", "lw"):
        is_rrtmg = rrtmg_sw if sfx == "sw" else rrtmg_lw
        want = (config.do_save_spectral_flux or config.do_toa_spectral_flux) if is_rrtmg else config.do_save_spectral_flux
        if not getattr(config, "do_" + sfx) or not want:
            setattr(config, "n_spec_" + sfx, 0)
            setattr(config, "i_spec_from_reordered_g_" + sfx, None)
        elif config.do_save_gpoint_flux:
            ng = getattr(config, "n_g_" + sfx)
            setattr(config, "n_spec_" + sfx, ng)
            # radiation_ifs_rrtm.F90:139-141: i_spec_from_reordered_g => i_g_from_reordered_g -- with SPARTACUS's
            # reordering the per-g-point flux profiles are stored by the NATIVE RRTMG g-point index
            perm = getattr(config.rrtmg, "i_g_from_reordered_g_" + sfx, None) if is_rrtmg else None
            setattr(config, "i_spec_from_reordered_g_" + sfx,
                    np.arange(1, ng + 1, dtype=np.int32) if perm is None else np.asarray(perm, dtype=np.int32).copy())
        else:
            setattr(config, "n_spec_" + sfx, getattr(config, "n_bands_" + sfx))
            setattr(config, "i_spec_from_reordered_g_" + sfx,
                    np.asarray(getattr(config, "i_band_from_reordered_g_" + sfx), dtype=np.int32))
    if config.do_lw_aerosol_scattering and not config.do_lw_cloud_scattering:
        raise ConfigError("longwave aerosol scattering requires longwave cloud scattering")
    config.n_g_lw_if_scattering = config.n_g_lw if config.do_lw_aerosol_scattering else 0
    config.n_bands_lw_if_scattering = config.n_bands_lw if config.do_lw_cloud_scattering else 0
    if config.do_lw_cloud_scattering and config.i_solver_lw == ISolverMcICA:
        config.n_g_lw_if_scattering = config.n_g_lw

    def intervals(index, bound, full_spectrum, ng):       # radiation_config.F90:1947-2100, as in setup_radiation
        ninterval = 0
        for j, v in enumerate(index[:NMaxAlbedoIntervals]):
            if v > 0:
                ninterval = j + 1
            else:
                break
        if ninterval < 1:
            return [1], [], (ng if full_spectrum else 1)
        idx = list(index[:ninterval])
        return idx, list(bound[:ninterval - 1]), (ng if full_spectrum else max(idx))

    if config.do_sw:
        idx, bnd, config.n_canopy_bands_sw = intervals(config.i_sw_albedo_index, config.sw_albedo_wavelength_bound,
                                                       config.use_canopy_full_spectrum_sw, config.n_g_sw)
        config.sw_albedo_weights = np.ascontiguousarray(sd_sw.calc_mapping_from_bands(bnd, idx, use_bands=use_bands_sw))
        if config.do_nearest_spectral_sw_albedo:
            config.i_albedo_from_band_sw = (np.argmax(config.sw_albedo_weights, axis=1) + 1).astype(np.int32)
    if config.do_lw:
        idx, bnd, config.n_canopy_bands_lw = intervals(config.i_lw_emiss_index, config.lw_emiss_wavelength_bound,
                                                       config.use_canopy_full_spectrum_lw, config.n_g_lw)
        config.lw_emiss_weights = np.ascontiguousarray(sd_lw.calc_mapping_from_bands(bnd, idx, use_bands=use_bands_lw))
        if config.do_nearest_spectral_lw_emiss:
            config.i_emiss_from_band_lw = (np.argmax(config.lw_emiss_weights, axis=1) + 1).astype(np.int32)

    if config.do_clouds:
        config.cloud_optics_sw, config.cloud_optics_lw = [], []
        if config.use_general_cloud_optics:
            names = [n for n in config.cloud_type_name if n] or ["mie_droplet", "baum-general-habit-mixture_ice"]
            config.n_cloud_types = len(names)
            for j, name in enumerate(names):
                fn = name if name.startswith("/") else os.path.join(
                    config.directory_name, name if name.endswith(".nc") else name + "_scattering.nc")
                thick = config.use_thick_cloud_spectral_averaging[j]
                if config.do_sw:
                    config.cloud_optics_sw.append(GeneralCloudOptics(fn, sd_sw, use_bands_sw, thick, SOLAR_REFERENCE_TEMPERATURE, name))
                if config.do_lw:
                    config.cloud_optics_lw.append(GeneralCloudOptics(fn, sd_lw, use_bands_lw, thick, TERRESTRIAL_REFERENCE_TEMPERATURE, name))
        else:
            # setup_cloud_optics (radiation_cloud_optics.F90:38-213): liquid then ice
            config.n_cloud_types = 2
            if config.do_sw:
                config.cloud_optics_sw = [BandFitCloudOptics(config.liq_optics_file_name, "coeff_sw", 14),
                                          BandFitCloudOptics(config.ice_optics_file_name, "coeff_sw", 14)]
            if config.do_lw:
                config.cloud_optics_lw = [BandFitCloudOptics(config.liq_optics_file_name, "coeff_lw", 16),
                                          BandFitCloudOptics(config.ice_optics_file_name, "coeff_lw", 16)]
            # number of coefficients each scheme expects (radiation_cloud_optics.F90:84-213)
            from .config import (IIceModelBaran, IIceModelBaran2016, IIceModelBaran2017, IIceModelFu, IIceModelYi,
                                 ILiquidModelSlingo, ILiquidModelSOCRATES)
            want_liq = {ILiquidModelSOCRATES: (16, 16), ILiquidModelSlingo: (6, 13)}[config.i_liq_model]
            want_ice = {IIceModelFu: (10, 11), IIceModelBaran: (9, 9), IIceModelBaran2016: (5, 5), IIceModelBaran2017: (9, 9),
                        IIceModelYi: (69, 69)}[config.i_ice_model]
            for lst, k in ((config.cloud_optics_sw, 0), (config.cloud_optics_lw, 1)):
                if lst and (lst[0].n_effective_radius != want_liq[k] or lst[1].n_effective_radius != want_ice[k]):
                    raise ConfigError("number of cloud optical coefficients does not match number expected")
            if config.i_ice_model == IIceModelBaran2017:
                from .tables import GeneralCoefficients
                gen = GeneralCoefficients(config.ice_optics_file_name)
                if gen.n_effective_radius != 5:
                    raise ConfigError("number of general ice-particle optical coefficients does not match number expected (5)")
                for lst in (config.cloud_optics_sw, config.cloud_optics_lw):
                    if lst:
                        lst.append(gen)
    if config.use_aerosols:
        if config.n_aerosol_types > 0:
            if not config.use_general_aerosol_optics and mixed:
                raise ConfigError("band-wise aerosol files (use_general_aerosol_optics=false) hold the RRTMG bands: not for an ecCKD spectrum")
            config.aerosol_optics = AerosolOptics(config.aerosol_optics_file_name, sd_sw if config.do_sw else None,
                                                  sd_lw if config.do_lw else None, config.do_cloud_aerosol_per_sw_g_point,
                                                  config.do_cloud_aerosol_per_lw_g_point, config.do_sw, config.do_lw)
            # radiation_aerosol_optics.F90:67-77
            if (config.do_lw and config.n_bands_lw != config.aerosol_optics.n_bands_lw) or \
               (config.do_sw and config.n_bands_sw != config.aerosol_optics.n_bands_sw):
                raise ConfigError("number of bands does not match aerosol optics look-up table")
            config.aerosol_optics.set_types(config.i_aerosol_type_map[:config.n_aerosol_types])
        else:
            config.use_aerosols = False
    if ISolverMcICA in (config.i_solver_sw, config.i_solver_lw):
        config.pdf_sampler = PdfSampler(config.cloud_pdf_file_name)


# ----------------------------------------------------------------------------------------------------
def build_config_struct(config: Config):
    """Flatten ``config`` into an ecrad_config_t.  Returns (struct, keepalive list)."""
    keep = []

    def d(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        return abi.dptr(a)

    def i(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.int32)
        keep.append(a)
        return abi.iptr(a)

    c = abi.Config()
    c.abi_version = abi.ABI_VERSION
    for name in ("do_sw", "do_lw", "do_clear", "do_sw_direct", "do_lw_derivatives", "do_clouds",
                 "use_aerosols", "i_solver_sw", "i_solver_lw", "i_gas_model_sw", "i_gas_model_lw",
                 "do_lw_cloud_scattering", "do_lw_aerosol_scattering", "do_sw_delta_scaling_with_gases",
                 "use_general_cloud_optics", "is_homogeneous", "i_overlap_scheme", "use_beta_overlap",
                 "use_vectorizable_generator", "i_cloud_pdf_shape", "do_cloud_aerosol_per_sw_g_point",
                 "do_cloud_aerosol_per_lw_g_point", "do_surface_sw_spectral_flux", "do_toa_spectral_flux",
                 "do_canopy_fluxes_sw", "do_canopy_fluxes_lw", "use_canopy_full_spectrum_sw",
                 "use_canopy_full_spectrum_lw", "do_nearest_spectral_sw_albedo",
                 "do_nearest_spectral_lw_emiss", "do_save_spectral_flux", "n_g_sw", "n_g_lw", "n_bands_sw",
                 "n_bands_lw", "n_g_lw_if_scattering", "n_bands_lw_if_scattering", "n_canopy_bands_sw",
                 "n_canopy_bands_lw", "n_cloud_types", "n_spec_sw", "n_spec_lw"):
        setattr(c, name, int(getattr(config, name)))
    c.cloud_fraction_threshold = config.cloud_fraction_threshold
    c.cloud_mixing_ratio_threshold = config.cloud_mixing_ratio_threshold
    c.cloud_inhom_decorr_scaling = config.cloud_inhom_decorr_scaling
    c.max_cloud_od = config.max_cloud_od
    c.nregions = config.nregions
    c.i_3d_sw_entrapment = config.i_3d_sw_entrapment
    c.do_3d_effects = int(config.do_3d_effects)
    c.do_3d_lw_multilayer_effects = int(config.do_3d_lw_multilayer_effects)
    c.do_lw_side_emissivity = int(config.do_lw_side_emissivity)
    c.use_expm_everywhere = int(config.use_expm_everywhere)
    c.i_precision = int(config.i_precision)
    c.max_3d_transfer_rate = config.max_3d_transfer_rate
    c.max_gas_od_3d = config.max_gas_od_3d
    c.min_cloud_effective_size = config.min_cloud_effective_size
    c.overhang_factor = config.overhang_factor
    c.clear_to_thick_fraction = config.clear_to_thick_fraction
    c.overhead_sun_factor = config.overhead_sun_factor
    c.i_band_from_reordered_g_sw = i(config.i_band_from_reordered_g_sw)
    c.i_band_from_reordered_g_lw = i(config.i_band_from_reordered_g_lw)
    if config.sw_albedo_weights is not None:
        c.n_albedo_intervals_sw = config.sw_albedo_weights.shape[1]
        c.sw_albedo_weights = d(config.sw_albedo_weights)
    if config.lw_emiss_weights is not None:
        c.n_emiss_intervals_lw = config.lw_emiss_weights.shape[1]
        c.lw_emiss_weights = d(config.lw_emiss_weights)
    c.i_albedo_from_band_sw = i(config.i_albedo_from_band_sw)
    c.i_emiss_from_band_lw = i(config.i_emiss_from_band_lw)
    c.i_spec_from_reordered_g_sw = i(config.i_spec_from_reordered_g_sw)
    c.i_spec_from_reordered_g_lw = i(config.i_spec_from_reordered_g_lw)

    def fill_ckd(dst, m):
        if m is None:
            return
        dst.is_sw = int(m.is_sw)
        dst.ng, dst.npress, dst.ntemp, dst.ngas, dst.nplanck = m.ng, m.npress, m.ntemp, m.ngas, m.nplanck
        dst.log_pressure1, dst.d_log_pressure, dst.d_temperature = m.log_pressure1, m.d_log_pressure, m.d_temperature
        dst.temperature1_planck, dst.d_temperature_planck = m.temperature1_planck, m.d_temperature_planck
        dst.temperature1 = d(m.temperature1)
        dst.planck_function = d(m.planck_function)
        dst.norm_solar_irradiance = d(m.norm_solar_irradiance)
        dst.norm_amplitude_solar_irradiance = d(m.norm_amplitude_solar_irradiance)
        dst.rayleigh_molar_scat = d(m.rayleigh_molar_scat)
        for j, g in enumerate(m.single_gas):
            sg = dst.single_gas[j]
            sg.i_gas_code, sg.i_conc_dependence, sg.n_mole_frac = g.i_gas_code, g.i_conc_dependence, g.n_mole_frac
            sg.reference_mole_frac, sg.log_mole_frac1, sg.d_log_mole_frac = \
                g.reference_mole_frac, g.log_mole_frac1, g.d_log_mole_frac
            sg.molar_abs = d(g.molar_abs)

    if config.rrtmg is not None:
        keep.append(config.rrtmg)
        c.rrtmg = C.pointer(config.rrtmg.struct)
    if isinstance(config.gas_optics_sw, CkdModel):
        fill_ckd(c.gas_optics_sw, config.gas_optics_sw)
    if isinstance(config.gas_optics_lw, CkdModel):
        fill_ckd(c.gas_optics_lw, config.gas_optics_lw)
    c.min_gas_od_lw, c.min_gas_od_sw = config.min_gas_od_lw, config.min_gas_od_sw
    c.i_liq_model, c.i_ice_model = config.i_liq_model, config.i_ice_model
    c.do_fu_lw_ice_optics_bug = int(config.do_fu_lw_ice_optics_bug)
    for arr, src in ((c.cloud_optics_sw, config.cloud_optics_sw), (c.cloud_optics_lw, config.cloud_optics_lw)):
        for j, co in enumerate(src or []):
            arr[j].n_bands, arr[j].n_effective_radius = co.n_bands, co.n_effective_radius
            arr[j].effective_radius_0, arr[j].d_effective_radius = co.effective_radius_0, co.d_effective_radius
            arr[j].mass_ext, arr[j].ssa, arr[j].asymmetry = d(co.mass_ext), d(co.ssa), d(co.asymmetry)
    ao = config.aerosol_optics
    if ao is not None and config.use_aerosols:
        a = c.aerosol_optics
        a.n_bands_sw, a.n_bands_lw = ao.n_bands_sw, ao.n_bands_lw
        a.n_type_phobic, a.n_type_philic, a.nrh = ao.n_type_phobic, ao.n_type_philic, ao.nrh
        a.use_hydrophilic, a.ntype = int(ao.use_hydrophilic), ao.ntype
        a.iclass, a.itype = i(ao.iclass), i(ao.itype)
        a.rh_lower = d(getattr(ao, "rh_lower", None))
        for tag in ("sw", "lw"):
            for kind in ("phobic", "philic"):
                for q in ("mass_ext", "ssa", "g"):
                    n = f"{q}_{tag}_{kind}"
                    setattr(a, n, d(getattr(ao, n, None)))
    ps = config.pdf_sampler
    if ps is not None:
        c.pdf_sampler.ncdf, c.pdf_sampler.nfsd = ps.ncdf, ps.nfsd
        c.pdf_sampler.fsd1, c.pdf_sampler.inv_fsd_interval = ps.fsd1, ps.inv_fsd_interval
        c.pdf_sampler.val = d(ps.val)
    return c, keep


def build_inputs_struct(config: Config, ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol):
    """Flatten the five input types into an ecrad_inputs_t over host numpy arrays."""
    keep = []

    def d(a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float64)
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError(f"array has shape {a.shape}, expected {shape}")
        keep.append(a)
        return abi.dptr(a)

    s = abi.Inputs()
    s.memory = abi.MEM_HOST
    s.solar_irradiance = float(single_level.solar_irradiance)
    s.spectral_solar_cycle_multiplier = float(single_level.spectral_solar_cycle_multiplier)
    s.pressure_hl = d(thermodynamics.pressure_hl, (nlev + 1, ncol))
    s.temperature_hl = d(thermodynamics.temperature_hl, (nlev + 1, ncol))
    s.h2o_sat_liq = d(thermodynamics.h2o_sat_liq)
    s.cos_sza = d(single_level.cos_sza, (ncol,))
    s.skin_temperature = d(single_level.skin_temperature, (ncol,))
    s.n_sw_albedo = single_level.sw_albedo.shape[0]
    s.sw_albedo = d(single_level.sw_albedo)
    s.sw_albedo_direct = d(single_level.sw_albedo_direct)
    s.n_lw_emissivity = single_level.lw_emissivity.shape[0]
    s.lw_emissivity = d(single_level.lw_emissivity)
    if single_level.iseed is not None:
        iseed = np.ascontiguousarray(single_level.iseed, dtype=np.int32)
        keep.append(iseed)
        s.iseed = abi.iptr(iseed)
    if config.use_spectral_solar_scaling and config.do_sw:       # radiation_ifs_rrtm.F90:545-551
        if single_level.spectral_solar_scaling is None:
            raise ValueError("use_spectral_solar_scaling needs single_level%spectral_solar_scaling")
        s.spectral_solar_scaling = d(single_level.spectral_solar_scaling, (config.n_bands_sw,))
    s.gas_mixing_ratio = d(gas.mixing_ratio, (12, nlev, ncol))
    if cloud is not None and config.do_clouds:
        s.n_cloud_types = cloud.ntype
        # INOUT: must alias the caller's array so the crop side effect is visible
        if not (cloud.fraction.dtype == np.float64 and cloud.fraction.flags["C_CONTIGUOUS"]):
            raise ValueError("cloud.fraction must be C-contiguous float64 (it is modified in place)")
        s.cloud_fraction = abi.dptr(cloud.fraction)
        s.cloud_mixing_ratio = d(cloud.mixing_ratio, (cloud.ntype, nlev, ncol))
        s.cloud_effective_radius = d(cloud.effective_radius, (cloud.ntype, nlev, ncol))
        s.cloud_fractional_std = d(cloud.fractional_std, (nlev, ncol))
        s.cloud_overlap_param = d(cloud.overlap_param, (nlev - 1, ncol))
        if cloud.inv_cloud_effective_size is not None:
            s.cloud_inv_cloud_effective_size = d(cloud.inv_cloud_effective_size, (nlev, ncol))
        if cloud.inv_inhom_effective_size is not None:
            s.cloud_inv_inhom_effective_size = d(cloud.inv_inhom_effective_size, (nlev, ncol))
    if aerosol is not None and config.use_aerosols:
        s.n_aerosol_types = aerosol.mixing_ratio.shape[0]
        s.aerosol_istartlev, s.aerosol_iendlev = aerosol.istartlev, aerosol.iendlev
        s.aerosol_mixing_ratio = d(aerosol.mixing_ratio)
    return s, keep


def build_flux_struct(flux: Flux):
    f = abi.Flux()
    f.memory = abi.MEM_HOST
    for name in abi.FLUX_FIELDS:
        arr = getattr(flux, name)
        if arr is not None:
            setattr(f, name, abi.dptr(arr))
    return f


# ----------------------------------------------------------------------------------------------------
class Radiation:
    """Owns a configured handle: ``Radiation(config)`` == ``call setup_radiation(config)``;
    ``.radiation(...)`` == ``call radiation(ncol,nlev,istartcol,iendcol,config,...)``."""

    def __init__(self, config: Config, backend="hip", device_id: int = -1, lib_path: Optional[str] = None,
                 concurrency: Optional[tuple] = None):
        """concurrency = (n_devices, contexts_per_device): the pool of contexts that concurrent host-memory calls are spread
        over (include/ecrad_hip.h: ecrad_hip_set_concurrency; n_devices 0 = every visible device); None keeps the
        library's default, one device and eight contexts."""
        if not config.is_consolidated or config.gas_optics_lw is None and config.gas_optics_sw is None:
            setup_radiation(config)
        self.config = config
        self.cconfig, self._keep = build_config_struct(config)
        self.backend = backend
        self.lib = None
        self.handle = None
        if backend == "hip":
            self.lib = load_library(lib_path or LIB_PATH)      # (lib_path: a tuning / test variant of the library)
            h = C.c_void_p()
            st = self.lib.ecrad_hip_create(C.byref(h), device_id)
            if st != 0:
                raise EcradHipError(f"ecrad_hip_create failed with status {st} (no usable gfx950 device?)")
            self.handle = h
            if concurrency is not None:
                self._check(self.lib.ecrad_hip_set_concurrency(self.handle, int(concurrency[0]), int(concurrency[1])), "ecrad_hip_set_concurrency")
            self._check(self.lib.ecrad_hip_setup(self.handle, C.byref(self.cconfig)), "ecrad_hip_setup")
        elif not callable(backend):
            raise ValueError("backend must be 'hip' or a callable (tests/bench cpu_baseline only)")

    def _check(self, status: int, what: str) -> None:
        if status != 0:
            msg = self.lib.ecrad_hip_last_error(self.handle)
            raise EcradHipError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")

    def set_gas_units(self, gas) -> None:
        """set_gas_units (radiation_interface.F90:164-190 -> radiation_ecckd_interface.F90:153-171)."""
        if self.config.rrtmg is not None:      # radiation_ifs_rrtm.F90:203-213
            from .types import IMassMixingRatio
            gas.set_units(IMassMixingRatio)
        else:
            gas.set_units(IVolumeMixingRatio)

    def optics(self, ncol, nlev, istartcol, iendcol, single_level, thermodynamics, gas, cloud, aerosol) -> dict:
        """The arrays radiation() passes between its stages (radiation_interface.F90:260-301) for columns
        istartcol..iendcol, as numpy arrays (ncol_local, nlev[+1], ng): ecrad_hip_optics."""
        cin, keep = build_inputs_struct(self.config, ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol)
        c, nloc = self.config, iendcol - istartcol + 1
        shapes = {"od_lw": (nloc, nlev, c.n_g_lw), "ssa_lw": (nloc, nlev, c.n_g_lw), "g_lw": (nloc, nlev, c.n_g_lw),
                  "od_sw": (nloc, nlev, c.n_g_sw), "ssa_sw": (nloc, nlev, c.n_g_sw), "g_sw": (nloc, nlev, c.n_g_sw),
                  "planck_hl": (nloc, nlev + 1, c.n_g_lw), "lw_emission": (nloc, c.n_g_lw), "lw_albedo": (nloc, c.n_g_lw),
                  "sw_albedo_direct": (nloc, c.n_g_sw), "sw_albedo_diffuse": (nloc, c.n_g_sw), "incoming_sw": (nloc, c.n_g_sw),
                  "od_lw_cloud": (nloc, nlev, c.n_bands_lw), "ssa_lw_cloud": (nloc, nlev, c.n_bands_lw),
                  "g_lw_cloud": (nloc, nlev, c.n_bands_lw), "od_sw_cloud": (nloc, nlev, c.n_bands_sw),
                  "ssa_sw_cloud": (nloc, nlev, c.n_bands_sw), "g_sw_cloud": (nloc, nlev, c.n_bands_sw)}
        want = {k: v for k, v in shapes.items()
                if (c.do_lw if "lw" in k or k == "planck_hl" else c.do_sw) and (c.do_clouds or not k.endswith("_cloud"))}
        if self.backend == "hip":
            out = abi.Optics()
            arrs = {k: np.zeros(v) for k, v in want.items()}
            for k, a in arrs.items():
                setattr(out, k, abi.dptr(a))
            self._check(self.lib.ecrad_hip_optics(self.handle, ncol, nlev, istartcol, iendcol, C.byref(cin), C.byref(out)),
                        "ecrad_hip_optics")
        else:
            fn = getattr(self.backend, "optics", None)
            if fn is None:
                raise RuntimeError("this backend has no optics()")
            arrs = {k: a for k, a in fn(self.config, self.cconfig, ncol, nlev, istartcol, iendcol, cin).items() if k in want}
        del keep
        return arrs

    def radiation(self, ncol, nlev, istartcol, iendcol, single_level, thermodynamics, gas,
                  cloud, aerosol, flux) -> None:
        if self.config.do_save_radiative_properties:       # radiation_interface.F90:403-419
            from .driver import save_radiative_properties
            name = "radiative_properties.nc" if (istartcol == 1 and iendcol == ncol) else \
                f"radiative_properties_{istartcol:04d}-{iendcol:04d}.nc"
            props = self.optics(ncol, nlev, istartcol, iendcol, single_level, thermodynamics, gas, cloud, aerosol)
            save_radiative_properties(name, self.config, nlev, istartcol, iendcol, single_level, thermodynamics, cloud, props)
        cin, keep = build_inputs_struct(self.config, ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol)
        cflux = build_flux_struct(flux)
        if self.backend == "hip":
            self._check(self.lib.ecrad_hip_radiation(self.handle, ncol, nlev, istartcol, iendcol,
                                                     C.byref(cin), C.byref(cflux)), "ecrad_hip_radiation")
        else:
            st = self.backend(self.cconfig, ncol, nlev, istartcol, iendcol, cin, cflux)
            if st != 0:
                raise RuntimeError(f"backend returned status {st}")
        del keep

    def pool_info(self, reset: bool = False) -> dict:
        """ecrad_hip_pool_info: how the calls so far were spread over the pool's devices and contexts."""
        info = abi.PoolInfo()
        self._check(self.lib.ecrad_hip_pool_info(self.handle, C.byref(info)), "ecrad_hip_pool_info")
        out = {"n_devices": info.n_devices, "n_contexts": info.n_contexts, "in_flight": info.in_flight,
               "max_in_flight": info.max_in_flight, "calls_total": info.calls_total, "batches_total": info.batches_total,
               "calls_on_device": {int(info.device_ids[i]): int(info.calls_on_device[i]) for i in range(info.n_devices)}}
        if reset:
            self._check(self.lib.ecrad_hip_pool_reset(self.handle), "ecrad_hip_pool_reset")
        return out

    def last_kernel_ms(self) -> float:
        ms = C.c_double()
        self._check(self.lib.ecrad_hip_last_kernel_ms(self.handle, C.byref(ms)), "ecrad_hip_last_kernel_ms")
        return ms.value

    def close(self) -> None:
        if self.handle is not None and self.lib is not None:
            self.lib.ecrad_hip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
