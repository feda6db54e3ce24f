"""Row a6: the product's RRTMG device code (ecrad_amd/csrc/rrtmg_device.h: band descriptors, evaluators, table
packing), compiled for the CPU by g++ (tests/_src/rrtmg_hostcheck.cpp), against the outputs of the
reference's own ifsrrtm routines for 8 meridian columns (tests/golden/rrtmg_gas_optics.npz).  The GPU tests
(test_hip_parity.py) check the same quantities through the C-ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "rrtmg_gas_optics.npz")
GAS_CODE = dict(q=1, co2=2, o3=3, n2o=4, ch4=6, cfc11=8, cfc12=9, hcfc22=10, ccl4=11, no2=12)


def build_hostcheck():
    src = os.path.join(ROOT, "tests", "_src", "rrtmg_hostcheck.cpp")
    out = os.path.join(ROOT, "tests", "_build", "librrtmg_hostcheck.so")
    hdr = os.path.join(ROOT, "ecrad_amd", "csrc", "rrtmg_device.h")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", out], check=True)
    return C.CDLL(out)


def run_hostcheck(g, solar_irradiance=1361.0, skin_temperature=None):
    from ecrad_amd.rrtmg import RrtmgTables
    lib = build_hostcheck()
    tables = RrtmgTables()
    nlev, ncol = g["q"].shape
    F = lambda a: np.ascontiguousarray(a)          # (nlev, ncol) C order == Fortran (ncol, nlev)
    gas = np.zeros((12, nlev, ncol))
    for n, code in GAS_CODE.items():
        gas[code - 1] = g[n]
    skin = g["temperature_hl"][-1].copy() if skin_temperature is None else skin_temperature
    out = dict(od_lw=np.zeros((ncol, nlev, 140)), pfrac=np.zeros((ncol, nlev, 140)), planck_hl=np.zeros((ncol, nlev + 1, 140)),
               lw_emission=np.zeros((ncol, 140)), od_sw=np.zeros((ncol, nlev, 112)), ssa_sw=np.zeros((ncol, nlev, 112)),
               incoming_sw=np.zeros((ncol, 112)), incsol_raw=np.zeros((ncol, 112)))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.rrtmg_hostcheck.restype = C.c_int
    st = lib.rrtmg_hostcheck(C.byref(tables.struct), C.c_int(ncol), C.c_int(nlev), p(F(g["pressure_hl"])), p(F(g["temperature_hl"])),
                             p(gas), p(F(g["cos_sza"])), p(skin), C.c_double(solar_irradiance),
                             *[p(out[k]) for k in ("od_lw", "pfrac", "planck_hl", "lw_emission", "od_sw", "ssa_sw", "incoming_sw", "incsol_raw")])
    assert st == 0
    return out


def rel(a, b, floor=0.0):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor + 1e-300))


@pytest.fixture(scope="module")
def pair():
    g = np.load(FIXTURE)
    return g, run_hostcheck(g)


def test_longwave_optical_depth_and_planck_fractions(pair):
    g, o = pair
    ref_od = np.maximum(g["od_lw"][:, ::-1, :], 1e-15)          # (ncol, nlev from top, 140)
    ref_pf = np.transpose(g["pfrac"], (2, 0, 1))[:, ::-1, :]     # (ncol, nlev from top, 140)
    # per band, to name the culprit
    edges = np.cumsum([0, 10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2])
    for b in range(16):
        s = slice(edges[b], edges[b + 1])
        assert rel(o["od_lw"][:, :, s], ref_od[:, :, s], 1e-13) < 1e-10, f"od_lw band {b + 1}"
        assert np.max(np.abs(o["pfrac"][:, :, s] - ref_pf[:, :, s])) < 1e-13, f"pfrac band {b + 1}"


def test_shortwave_optical_depth_ssa_and_solar_source(pair):
    g, o = pair
    ref_od = np.transpose(g["od_sw"], (2, 1, 0))[:, ::-1, :]
    ref_ssa = np.transpose(g["ssa_sw"], (2, 1, 0))[:, ::-1, :]
    ref_inc = g["incsol"].T
    edges = np.cumsum([0, 6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12])
    day = g["cos_sza"] > 0
    assert day.any() and (~day).any()
    for b in range(14):
        s = slice(edges[b], edges[b + 1])
        assert rel(o["od_sw"][day][:, :, s], ref_od[day][:, :, s]) < 1e-10, f"od_sw band {b + 16}"
        assert np.max(np.abs(o["ssa_sw"][day][:, :, s] - ref_ssa[day][:, :, s])) < 1e-12, f"ssa_sw band {b + 16}"
        assert rel(o["incsol_raw"][day][:, s], ref_inc[day][:, s]) < 1e-13, f"incsol band {b + 16}"
    assert np.all(o["od_sw"][~day] == 0) and np.all(o["incoming_sw"][~day] == 0)
    assert np.allclose(o["incoming_sw"][day].sum(axis=1), 1361.0, rtol=1e-13)


def test_planck_function(pair):
    """planck_function_atmos/_surf (radiation_ifs_rrtm.F90:618-852) restated in numpy from the dumped
    TOTPLNK/DELWAVE tables, times the reference's Planck fractions."""
    g, o = pair
    t = np.load(os.path.join(ROOT, "data", "rrtmg_tables.npz"))
    totplnk, delwave = t["yoerrtwn.totplnk"], t["yoerrtwn.delwave"]
    band = np.repeat(np.arange(16), [10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2])

    def planck(T):
        T = np.asarray(T)
        ind = np.where(T >= 339.0, 180, np.where(T >= 160.0, (T - 159.0).astype(int), 1))
        frac = np.where(T >= 339.0, T - 339.0, np.where(T >= 160.0, T - np.trunc(T), 0.0))
        fac = 2.0 * np.arcsin(1.0) * 1.0e4 * delwave[band]
        return fac * (totplnk[ind[..., None] - 1, band] + frac[..., None] * (totplnk[ind[..., None], band] - totplnk[ind[..., None] - 1, band]))

    pf = np.transpose(g["pfrac"], (2, 0, 1))[:, ::-1, :]                # (ncol, layer from top, 140)
    thl = g["temperature_hl"].T                                        # (ncol, nlev+1)
    ref = planck(thl) * np.concatenate([pf[:, :1, :], pf], axis=1)
    assert rel(o["planck_hl"], ref, 1e-30) < 1e-12
    assert rel(o["lw_emission"], planck(thl[:, -1]) * pf[:, -1, :], 1e-30) < 1e-12
