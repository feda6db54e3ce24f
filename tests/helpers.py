"""Shared helpers for the test-suite: build configs from the reference's namelist spellings, load
the meridian test case, run a backend, compare against goldens."""
import copy
import os

import numpy as np

from ecrad_amd.config import Config
from ecrad_amd.driver import DriverConfig, flux_to_output_dict, read_input
from ecrad_amd.interface import Radiation
from ecrad_amd.ncfile import NcFile
from ecrad_amd.types import Flux

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA_DIR = os.path.join(ROOT, "data")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
from ecrad_amd.cases import (GOLDEN_CASES, MERIDIAN, NAMELIST, load_meridian, make_config,  # noqa: F401,E402
                             make_config_rrtmg, make_golden_config, run_case, rel_err, compare_flux)
