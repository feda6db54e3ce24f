#!/usr/bin/env python
"""oracle/make_rrtm_golden.py -- TEST INFRASTRUCTURE: golden vectors for RRTMG gas optics.

Runs the reference's own RRTMG routines (oracle/_ref/libecrad_refrrtm.so, built by
oracle/build_ref_rrtm.sh from /root/reference/ifsrrtm, unmodified) on columns of the reference's test
case test/ifs/ecrad_meridian.nc and stores inputs and outputs in tests/golden/rrtmg_gas_optics.npz:

  inputs   pressure_hl, temperature_hl (nlev+1, ncol); mass mixing ratios q, co2, ch4, n2o, no2, cfc11,
           cfc12, hcfc22, ccl4, o3 (nlev, ncol); cos_sza (ncol)           -- level 0 = top of atmosphere
  outputs  od_lw (ncol, nlev, 140), pfrac (nlev, 140, ncol), od_sw, ssa_sw (112, nlev, ncol),
           incsol (112, ncol)                                              -- levels counted from the SURFACE,
           exactly as RRTM_GAS_OPTICAL_DEPTH / SRTM_GAS_OPTICAL_DEPTH return them (numpy order = reversed
           Fortran order)

The calling sequence is the one of radiation/radiation_ifs_rrtm.F90:406-542 (see oracle/ref_rrtm_wrappers.F90).
Only this container has /root/reference; the .npz travels with the repo.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ecrad_amd.tables import ICCl4, ICFC11, ICFC12, ICH4, ICO2, IH2O, IHCFC22, IN2O, INO2, IO3   # noqa: E402
from ecrad_amd.types import IMassMixingRatio                                                     # noqa: E402
from helpers import load_meridian, make_config                                                   # noqa: E402

REF_DATA = os.environ.get("ECRAD_REF_DATA", "/root/reference/data")
COLUMNS = [0, 5, 11, 16, 21, 26, 30, 31]     # pole to pole incl. night-time columns


def main():
    lib = C.CDLL(os.path.join(HERE, "_ref", "libecrad_refrrtm.so"))
    d = REF_DATA.encode()
    lib.ref_rrtm_setup(d, C.c_int(len(d)))
    ng_lw, ng_sw = C.c_int(), C.c_int()
    lib.ref_rrtm_sizes(C.byref(ng_lw), C.byref(ng_sw))
    ng_lw, ng_sw = ng_lw.value, ng_sw.value

    config = make_config("McICA")
    ncol_all, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    gas.set_units(IMassMixingRatio)          # radiation_ifs_rrtm.F90:208 (set_gas_units) / :399 (assert_units)
    cols = np.array(COLUMNS)
    ncol = len(cols)
    f = lambda a: np.asfortranarray(np.ascontiguousarray(a).T)     # (nlev, ncol) C order -> Fortran (ncol, nlev)
    inputs = dict(pressure_hl=th.pressure_hl[:, cols].copy(), temperature_hl=th.temperature_hl[:, cols].copy(),
                  cos_sza=sl.cos_sza[cols].copy())
    names = dict(q=IH2O, co2=ICO2, ch4=ICH4, n2o=IN2O, no2=INO2, cfc11=ICFC11, cfc12=ICFC12, hcfc22=IHCFC22,
                 ccl4=ICCl4, o3=IO3)
    for n, ig in names.items():
        inputs[n] = gas.mixing_ratio[ig - 1][:, cols].copy()
    od_lw = np.zeros((ng_lw, nlev, ncol), order="F")
    pfrac = np.zeros((ncol, ng_lw, nlev), order="F")
    od_sw = np.zeros((ncol, nlev, ng_sw), order="F")
    ssa_sw = np.zeros((ncol, nlev, ng_sw), order="F")
    incsol = np.zeros((ncol, ng_sw), order="F")
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    args = [f(inputs["pressure_hl"]), f(inputs["temperature_hl"])] + [f(inputs[n]) for n in names] \
        + [np.ascontiguousarray(inputs["cos_sza"])]
    lib.ref_rrtm_gas_optics(C.c_int(ncol), C.c_int(nlev), *[p(a) for a in args],
                            p(od_lw), p(pfrac), p(od_sw), p(ssa_sw), p(incsol))
    out = os.path.join(ROOT, "tests", "golden", "rrtmg_gas_optics.npz")
    np.savez_compressed(out, columns=cols, **inputs,
                        od_lw=np.ascontiguousarray(od_lw.T), pfrac=np.ascontiguousarray(pfrac.T),
                        od_sw=np.ascontiguousarray(od_sw.T), ssa_sw=np.ascontiguousarray(ssa_sw.T),
                        incsol=np.ascontiguousarray(incsol.T))
    print("wrote", out, os.path.getsize(out), "bytes")
    print("od_lw", od_lw.min(), od_lw.max(), "pfrac sum over g (should be ~16 bands)", pfrac[0, :, 0].sum(),
          "od_sw max", od_sw.max(), "incsol sum", incsol.sum(axis=1))


if __name__ == "__main__":
    main()
