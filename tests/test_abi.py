"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/ecrad_hip.h declares, and its struct layouts agree with the ctypes/Fortran mirrors.
No compute call is made (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

from ecrad_amd import abi
from ecrad_amd.interface import LIB_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return C.CDLL(LIB_PATH)


def header_functions():
    text = open(os.path.join(ROOT, "include", "ecrad_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ecrad_hip_\w+)\s*\(", text)))


def test_header_and_python_symbol_lists_agree():
    assert header_functions() == sorted(abi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in header_functions():
        assert hasattr(lib, name), name


def test_struct_layouts_match(lib):
    abi.declare_prototypes(lib)
    assert lib.ecrad_hip_abi_version() == abi.ABI_VERSION
    for i, s in enumerate(abi.STRUCT_BY_INDEX):
        assert lib.ecrad_hip_abi_sizeof(i) == C.sizeof(s), s.__name__


def test_create_without_gpu_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    abi.declare_prototypes(lib)
    h = C.c_void_p()
    st = lib.ecrad_hip_create(C.byref(h), -1)
    assert st == -2          # ECRAD_ENODEVICE: no CPU fallback
    from ecrad_amd import Radiation, EcradHipError
    from helpers import make_config
    with pytest.raises(EcradHipError):
        Radiation(make_config("Cloudless"), backend="hip")
