"""ecrad_amd -- MI355X-native implementation of ecRad's ``radiation()`` hot path.

Host side (this package): the reference's operator interface (``Config``, ``setup_radiation``,
``Radiation.radiation``) and offline-driver I/O, marshalling into the C-ABI of
``include/ecrad_hip.h``.  Device side (``ecrad_amd/csrc``): hand-written HIP kernels for gfx950.
"""
from .config import Config, ConfigError  # noqa: F401
from .interface import Radiation, setup_radiation, EcradHipError  # noqa: F401
from .types import (Aerosol, Cloud, Flux, Gas, SingleLevel, Thermodynamics)  # noqa: F401

__all__ = ["Config", "ConfigError", "Radiation", "setup_radiation", "EcradHipError",
           "Aerosol", "Cloud", "Flux", "Gas", "SingleLevel", "Thermodynamics"]
