"""ctypes access to oracle/libecrad_oracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It reuses ecrad_amd's marshalling (ecrad_amd.abi structs) so that the oracle sees byte-identical
inputs to the HIP library; nothing in ecrad_amd imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libecrad_oracle.so")
REF_LEAF_PATH = os.path.join(_HERE, "_ref", "libecrad_refleaf.so")


def build(ref: bool = False) -> None:
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    if ref and os.path.isdir("/root/reference"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        from ecrad_amd import abi
        L = C.CDLL(LIB_PATH)
        L.ecrad_oracle_radiation.argtypes = [C.POINTER(abi.Config), C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(abi.Inputs), C.POINTER(abi.Flux)]
        L.ecrad_oracle_radiation.restype = C.c_int
        L.ecrad_oracle_radiation_blocked.argtypes = [C.POINTER(abi.Config), C.c_int, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_int, C.POINTER(abi.Inputs), C.POINTER(abi.Flux)]
        L.ecrad_oracle_radiation_blocked.restype = C.c_int
        L.ecrad_oracle_optics.argtypes = [C.POINTER(abi.Config), C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(abi.Inputs), C.POINTER(abi.Optics)]
        L.ecrad_oracle_optics.restype = C.c_int
        L.ecrad_oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def backend(cconfig, ncol, nlev, istartcol, iendcol, cin, cflux) -> int:
    """Drop-in ``backend=`` callable for ecrad_amd.interface.Radiation (tests only)."""
    return lib().ecrad_oracle_radiation(C.byref(cconfig), ncol, nlev, istartcol, iendcol,
                                        C.byref(cin), C.byref(cflux))


def make_blocked_backend(nblocksize: int, nthreads: int = 0):
    def _b(cconfig, ncol, nlev, istartcol, iendcol, cin, cflux) -> int:
        return lib().ecrad_oracle_radiation_blocked(C.byref(cconfig), ncol, nlev, istartcol, iendcol,
                                                    nblocksize, nthreads, C.byref(cin), C.byref(cflux))
    return _b


def optics(config, cconfig, ncol, nlev, istartcol, iendcol, cin) -> dict:
    """Run the pre-solver stages; returns numpy arrays shaped (ncol_local, nlev[+1], ng)."""
    from ecrad_amd import abi
    nloc = iendcol - istartcol + 1
    shapes = optics_shapes(config, nlev, nloc)
    out = abi.Optics()
    arrs = {}
    for k, shp in shapes.items():
        arrs[k] = np.zeros(shp)
        setattr(out, k, abi.dptr(arrs[k]))
    st = lib().ecrad_oracle_optics(C.byref(cconfig), ncol, nlev, istartcol, iendcol, C.byref(cin), C.byref(out))
    if st != 0:
        raise RuntimeError(f"ecrad_oracle_optics status {st}")
    return arrs


def optics_shapes(config, nlev, nloc) -> dict:
    return {
        "od_lw": (nloc, nlev, config.n_g_lw), "ssa_lw": (nloc, nlev, config.n_g_lw), "g_lw": (nloc, nlev, config.n_g_lw),
        "od_sw": (nloc, nlev, config.n_g_sw), "ssa_sw": (nloc, nlev, config.n_g_sw), "g_sw": (nloc, nlev, config.n_g_sw),
        "planck_hl": (nloc, nlev + 1, config.n_g_lw), "lw_emission": (nloc, config.n_g_lw),
        "lw_albedo": (nloc, config.n_g_lw), "sw_albedo_direct": (nloc, config.n_g_sw),
        "sw_albedo_diffuse": (nloc, config.n_g_sw), "incoming_sw": (nloc, config.n_g_sw),
        "od_lw_cloud": (nloc, nlev, config.n_bands_lw), "ssa_lw_cloud": (nloc, nlev, config.n_bands_lw),
        "g_lw_cloud": (nloc, nlev, config.n_bands_lw), "od_sw_cloud": (nloc, nlev, config.n_bands_sw),
        "ssa_sw_cloud": (nloc, nlev, config.n_bands_sw), "g_sw_cloud": (nloc, nlev, config.n_bands_sw),
    }
