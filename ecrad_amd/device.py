"""Device-resident inputs/outputs for the C-ABI's ECRAD_MEM_DEVICE mode.

PyTorch is used here only as plumbing: it owns HBM allocations and the HIP stream; the kernels are
launched by libecrad_hip.so through the C-ABI with raw device pointers (no torch types cross the
boundary)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi


class DeviceCase:
    """Holds torch tensors for one set of columns and the ecrad_inputs_t / ecrad_flux_t over them."""

    def __init__(self, config, ncol, nlev, single_level, thermodynamics, gas, cloud, aerosol, flux, device="cuda:0"):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        self.ncol, self.nlev = ncol, nlev
        self.tensors = {}

        def up(name, a, dtype=None):
            if a is None:
                return 0
            t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            self.tensors[name] = t
            return t.data_ptr()

        s = abi.Inputs()
        s.memory = abi.MEM_DEVICE
        s.solar_irradiance = float(single_level.solar_irradiance)
        s.spectral_solar_cycle_multiplier = float(single_level.spectral_solar_cycle_multiplier)
        s.pressure_hl = abi.raw_dptr(up("pressure_hl", thermodynamics.pressure_hl))
        s.temperature_hl = abi.raw_dptr(up("temperature_hl", thermodynamics.temperature_hl))
        s.h2o_sat_liq = abi.raw_dptr(up("h2o_sat_liq", thermodynamics.h2o_sat_liq))
        s.cos_sza = abi.raw_dptr(up("cos_sza", single_level.cos_sza))
        s.skin_temperature = abi.raw_dptr(up("skin_temperature", single_level.skin_temperature))
        s.n_sw_albedo = single_level.sw_albedo.shape[0]
        s.sw_albedo = abi.raw_dptr(up("sw_albedo", single_level.sw_albedo))
        s.sw_albedo_direct = abi.raw_dptr(up("sw_albedo_direct", single_level.sw_albedo_direct))
        s.n_lw_emissivity = single_level.lw_emissivity.shape[0]
        s.lw_emissivity = abi.raw_dptr(up("lw_emissivity", single_level.lw_emissivity))
        if single_level.iseed is not None:
            s.iseed = abi.raw_iptr(up("iseed", np.ascontiguousarray(single_level.iseed, dtype=np.int32)))
        if config.use_spectral_solar_scaling and config.do_sw and single_level.spectral_solar_scaling is not None:
            self._solar_scaling = np.ascontiguousarray(single_level.spectral_solar_scaling, dtype=np.float64)      # host memory (include/ecrad_hip.h)
            s.spectral_solar_scaling = abi.dptr(self._solar_scaling)
        s.gas_mixing_ratio = abi.raw_dptr(up("gas_mixing_ratio", gas.mixing_ratio))
        if cloud is not None and config.do_clouds:
            s.n_cloud_types = cloud.ntype
            s.cloud_fraction = abi.raw_dptr(up("cloud_fraction", cloud.fraction))
            s.cloud_mixing_ratio = abi.raw_dptr(up("cloud_mixing_ratio", cloud.mixing_ratio))
            s.cloud_effective_radius = abi.raw_dptr(up("cloud_effective_radius", cloud.effective_radius))
            s.cloud_fractional_std = abi.raw_dptr(up("cloud_fractional_std", cloud.fractional_std))
            s.cloud_overlap_param = abi.raw_dptr(up("cloud_overlap_param", cloud.overlap_param))
            if cloud.inv_cloud_effective_size is not None:
                s.cloud_inv_cloud_effective_size = abi.raw_dptr(up("cloud_inv_cloud_effective_size", cloud.inv_cloud_effective_size))
            if cloud.inv_inhom_effective_size is not None:
                s.cloud_inv_inhom_effective_size = abi.raw_dptr(up("cloud_inv_inhom_effective_size", cloud.inv_inhom_effective_size))
        if aerosol is not None and config.use_aerosols:
            s.n_aerosol_types = aerosol.mixing_ratio.shape[0]
            s.aerosol_istartlev, s.aerosol_iendlev = aerosol.istartlev, aerosol.iendlev
            s.aerosol_mixing_ratio = abi.raw_dptr(up("aerosol_mixing_ratio", aerosol.mixing_ratio))
        self.inputs = s
        f = abi.Flux()
        f.memory = abi.MEM_DEVICE
        self.flux_tensors = {}
        for name in abi.FLUX_FIELDS:
            arr = getattr(flux, name)
            if arr is not None:
                t = torch.from_numpy(arr).to(self.device)
                self.flux_tensors[name] = t
                setattr(f, name, abi.raw_dptr(t.data_ptr()))
        self.flux = f

    def flux_to_host(self, flux) -> None:
        for name, t in self.flux_tensors.items():
            getattr(flux, name)[...] = t.cpu().numpy()

    def input_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors.values())

    def output_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.flux_tensors.values())

    @classmethod
    def concatenate(cls, config, cases):
        """One DeviceCase over the columns of several (same configuration, same level count), in order.  Used to
        build batches larger than what is comfortable to generate on the host in one piece."""
        import torch
        first = cases[0]
        self = cls.__new__(cls)
        self.torch, self.device, self.nlev = torch, first.device, first.nlev
        self.ncol = sum(c.ncol for c in cases)

        def col_axis(t, n):
            if t.ndim == 1 or t.shape[-1] == n:
                return t.ndim - 1
            return 0 if t.shape[0] == n else 1

        def cat(name, group):
            parts = [getattr(c, group)[name] for c in cases]
            out = torch.cat(parts, dim=col_axis(parts[0], first.ncol)).contiguous()
            for c in cases:
                del getattr(c, group)[name]
            return out

        self.tensors = {n: cat(n, "tensors") for n in list(first.tensors)}
        self.flux_tensors = {n: cat(n, "flux_tensors") for n in list(first.flux_tensors)}
        s = abi.Inputs()
        for fname, _ in abi.Inputs._fields_:
            setattr(s, fname, getattr(first.inputs, fname))
        for n, t in self.tensors.items():
            setattr(s, n, abi.raw_iptr(t.data_ptr()) if n == "iseed" else abi.raw_dptr(t.data_ptr()))
        self.inputs = s
        f = abi.Flux()
        f.memory = abi.MEM_DEVICE
        for n, t in self.flux_tensors.items():
            setattr(f, n, abi.raw_dptr(t.data_ptr()))
        self.flux = f
        return self
