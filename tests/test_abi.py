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


def test_large_arrays_in_pages_of_their_own():
    """ecrad_amd.interface.private_pages_for_large_arrays / owns_its_pages (what a Python host uses before it page-locks arrays with
    ecrad_hip_host_register): arrays well above the threshold are mappings of their own -- 16 bytes into a page, above the heap --, a small
    array never is, and an array the allocator served from the heap's top is recognised as such.  Run in a child: the setting stays for
    the life of a process."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from ecrad_amd.interface import owns_its_pages, private_pages_for_large_arrays\n"
            "assert private_pages_for_large_arrays(1 << 16)\n"
            "big = [np.zeros(n) for n in (1 << 16, 1 << 18, 1 << 20) for _ in range(4)]\n"
            "small = [np.zeros(n) for n in (16, 128, 1024)]\n"
            "edge = [np.zeros(1 << 13) for _ in range(4)]\n"
            "ok = all(owns_its_pages(a) for a in big) and not any(owns_its_pages(a) for a in small)\n"
            "ok = ok and all(owns_its_pages(a) == ((a.ctypes.data & 0xfff) == 0x10 and a.ctypes.data > 0x700000000000) for a in edge)\n"
            "print(ok)\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "True", (p.stdout, p.stderr[-400:])
