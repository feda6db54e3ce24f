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


def test_call_arrays_move_into_whole_pages():
    """ecrad_amd.interface.page_aligned_empty / relocate_call_arrays (what a Python host uses before it page-locks arrays with
    ecrad_hip_host_register, which takes whole pages only): every large array of a call becomes a private mapping that begins on a page
    boundary and spans whole pages, with the same values; small arrays stay where they are."""
    import mmap
    import numpy as np
    from ecrad_amd.interface import page_aligned_empty, relocate_call_arrays
    from ecrad_amd.synthetic import make_columns
    from ecrad_amd.types import Flux
    from helpers import make_config
    config = make_config("Tripleclouds")
    config.n_g_lw = config.n_g_sw = 32      # (what Flux.allocate needs of a set-up configuration)
    config.n_bands_lw = config.n_bands_sw = 13
    n, nlev, sl, th, gas, cloud, aer = make_columns(config, 200, False)
    flux = Flux.allocate(config, n, nlev)
    before = {k: v.copy() for k, v in vars(cloud).items() if isinstance(v, np.ndarray)}
    ranges = []

    def own_pages(a):
        b, addr, nbytes = page_aligned_empty(a.shape, a.dtype)
        b[...] = a
        ranges.append((addr, nbytes, a.nbytes))
        return b

    moved = relocate_call_arrays(own_pages, (sl, th, gas, cloud, aer, None), flux, min_bytes=1 << 12)
    assert len(moved) >= 20
    for addr, nbytes, used in ranges:
        assert addr % mmap.PAGESIZE == 0 and nbytes % mmap.PAGESIZE == 0 and used <= nbytes < used + mmap.PAGESIZE
    for k, v in before.items():
        assert np.array_equal(getattr(cloud, k), v)
    assert sl.cos_sza.nbytes < 1 << 12 and sl.cos_sza.ctypes.data not in [r[0] for r in ranges]      # a small array: left alone
    assert all(a.flags.c_contiguous and a.flags.writeable for a in moved)
