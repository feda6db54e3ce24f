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
        subprocess.run(["make", "-C", _HERE, "refifs"], check=True, capture_output=True)


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


_lib_fma = None


def make_fma_variant_backend(nblocksize: int = 32, nthreads: int = 0):
    """The oracle compiled with floating-point contraction (oracle/Makefile: fma): NOT the parity reference --
    the difference to the plain build measures the sensitivity of the formulas to last-bit differences."""
    global _lib_fma
    if _lib_fma is None:
        subprocess.run(["make", "-C", _HERE, "fma"], check=True, capture_output=True)
        from ecrad_amd import abi
        L = C.CDLL(os.path.join(_HERE, "libecrad_oracle_fma.so"))
        L.ecrad_oracle_radiation_blocked.argtypes = [C.POINTER(abi.Config), C.c_int, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_int, C.POINTER(abi.Inputs), C.POINTER(abi.Flux)]
        L.ecrad_oracle_radiation_blocked.restype = C.c_int
        _lib_fma = L

    def _b(cconfig, ncol, nlev, istartcol, iendcol, cin, cflux) -> int:
        return _lib_fma.ecrad_oracle_radiation_blocked(C.byref(cconfig), ncol, nlev, istartcol, iendcol,
                                                       nblocksize, nthreads, C.byref(cin), C.byref(cflux))
    _b.lib = _lib_fma        # (the library whose RRTMG stage hand-over make_rrtmg_backend must use)
    return _b


_lib_variants = {}


def make_variant_backend(target: str, nblocksize: int = 32, nthreads: int = 0):
    """Another build of the same restatement (oracle/Makefile targets): "sp" = the SPARTACUS solvers and their matrix
    algebra in single precision (the reference's PARKIND1_SINGLE build); "sp_fma" = that with floating-point contraction
    allowed, the measure of how sensitive the single-precision formulas are to the last bit."""
    if target not in _lib_variants:
        subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)
        from ecrad_amd import abi
        L = C.CDLL(os.path.join(_HERE, f"libecrad_oracle_{target}.so"))
        L.ecrad_oracle_radiation_blocked.argtypes = [C.POINTER(abi.Config), C.c_int, C.c_int, C.c_int, C.c_int,
                                                     C.c_int, C.c_int, C.POINTER(abi.Inputs), C.POINTER(abi.Flux)]
        L.ecrad_oracle_radiation_blocked.restype = C.c_int
        _lib_variants[target] = L
    L = _lib_variants[target]

    def _b(cconfig, ncol, nlev, istartcol, iendcol, cin, cflux) -> int:
        return L.ecrad_oracle_radiation_blocked(C.byref(cconfig), ncol, nlev, istartcol, iendcol,
                                                nblocksize, nthreads, C.byref(cin), C.byref(cflux))
    _b.lib = L
    return _b


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


backend.optics = optics        # (Radiation.optics with the oracle as the backend: tests of save_radiative_properties)


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


# ---------------------------------------------------------------------------------------------------
# CPU baseline in the reference's own code: the clear-sky solver stage (two-stream + adding, SW + LW) by the reference's
# leaf routines in the reference's calling order, OpenMP over column blocks (oracle/ref_leaf_wrappers.F90:
# ref_clear_sky_solvers, compiled with the reference's modules into oracle/_ref/libecrad_refleaf.so).
_ref_leaf = None


def have_ref_leaf() -> bool:
    return os.path.exists(REF_LEAF_PATH)


def ref_clear_sky_solvers(stage: dict, cos_sza, nblocksize: int = 32):
    """stage: the arrays of optics() for a clear-sky, aerosol-free configuration.  Returns dict of (nlev+1, ncol) fluxes."""
    global _ref_leaf
    if _ref_leaf is None:
        _ref_leaf = C.CDLL(REF_LEAF_PATH)
    ncol, nlev, ng_sw = stage["od_sw"].shape
    ng_lw = stage["od_lw"].shape[2]
    out = {k: np.zeros((nlev + 1, ncol)) for k in ("sw_up", "sw_dn", "sw_dn_direct", "lw_up", "lw_dn")}
    p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))
    keep = [np.ascontiguousarray(stage[k], dtype=np.float64) for k in
            ("od_sw", "ssa_sw", "g_sw", "incoming_sw", "sw_albedo_diffuse", "sw_albedo_direct", "od_lw", "planck_hl", "lw_emission", "lw_albedo")]
    mu0 = np.ascontiguousarray(cos_sza, dtype=np.float64)
    _ref_leaf.ref_clear_sky_solvers(C.c_int(ncol), C.c_int(nlev), C.c_int(ng_sw), C.c_int(ng_lw), C.c_int(nblocksize), p(mu0),
                                    *[p(a) for a in keep], *[p(out[k]) for k in ("sw_up", "sw_dn", "sw_dn_direct", "lw_up", "lw_dn")])
    return out


# ---------------------------------------------------------------------------------------------------
# The reference's IFS-side parametrisations (ifs/liquid_effective_radius.F90, ifs/ice_effective_radius.F90,
# ifs/cloud_overlap_decorr_len.F90) compiled into oracle/_ref/libecrad_refifs.so (oracle/Makefile: refifs) with the
# bind(C) shims of oracle/ref_ifs_wrappers.F90.  Arrays are numpy (klev, klon) == Fortran (KLON, KLEV).
REF_IFS_PATH = os.path.join(_HERE, "_ref", "libecrad_refifs.so")
_ref_ifs = None


def have_ref_ifs() -> bool:
    return os.path.exists(REF_IFS_PATH)


def _ref_ifs_lib():
    global _ref_ifs
    if _ref_ifs is None:
        _ref_ifs = C.CDLL(REF_IFS_PATH)
    return _ref_ifs


def _dp(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ref_liquid_effective_radius(y, ppressure, ptemperature, pcloud_frac, pq_liq, pq_rain, pland_frac, pccn_land, pccn_sea):
    klev, klon = ppressure.shape
    out = np.zeros((klev, klon))
    a = [_dp(x) for x in (ppressure, ptemperature, pcloud_frac, pq_liq, pq_rain, pland_frac, pccn_land, pccn_sea)]
    P = C.POINTER(C.c_double)
    _ref_ifs_lib().ref_liquid_effective_radius(C.c_int(y.NRADLP), C.c_int(int(y.LCCNL)), C.c_int(int(y.LCCNO)),
                                               C.c_double(y.RCCNLND), C.c_double(y.RCCNSEA), C.c_int(klon), C.c_int(klev),
                                               *[x.ctypes.data_as(P) for x in a], out.ctypes.data_as(P))
    return out


def ref_ice_effective_radius(y, ppressure, ptemperature, pcloud_frac, pq_ice, pq_snow, pgemu):
    klev, klon = ppressure.shape
    out = np.zeros((klev, klon))
    a = [_dp(x) for x in (ppressure, ptemperature, pcloud_frac, pq_ice, pq_snow, pgemu)]
    P = C.POINTER(C.c_double)
    _ref_ifs_lib().ref_ice_effective_radius(C.c_int(y.NRADIP), C.c_int(y.NMINICE), C.c_double(y.RRE2DE), C.c_double(y.RMINICE),
                                            C.c_int(klon), C.c_int(klev), *[x.ctypes.data_as(P) for x in a], out.ctypes.data_as(P))
    return out


def ref_cloud_overlap_decorr_len(pgemu, kdecolat):
    g = _dp(pgemu)
    out = np.zeros(g.size)
    ratio = C.c_double(0.0)
    P = C.POINTER(C.c_double)
    _ref_ifs_lib().ref_cloud_overlap_decorr_len(C.c_int(g.size), g.ctypes.data_as(P), C.c_int(kdecolat), out.ctypes.data_as(P),
                                                C.byref(ratio))
    return out, ratio.value


# ---------------------------------------------------------------------------------------------------
# RRTMG: the oracle's gas optics for this model are the reference's OWN ifsrrtm routines
# (oracle/_ref/libecrad_refrrtm.so, built by oracle/build_ref_rrtm.sh from /root/reference, unmodified; the
# library travels to the GPU box with the repo).  What radiation_ifs_rrtm.F90 does around them is restated here.
REF_RRTM_PATH = os.path.join(_HERE, "_ref", "libecrad_refrrtm.so")
_DATA = os.path.join(_HERE, "..", "data")
_ref_rrtm = None
_NG_LW = [10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2]
_NG_SW = [6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12]


def have_ref_rrtm() -> bool:
    return os.path.exists(REF_RRTM_PATH) and os.path.exists(os.path.join(_DATA, "RADRRTM"))


def ref_rrtm():
    global _ref_rrtm
    if _ref_rrtm is None:
        L = C.CDLL(REF_RRTM_PATH)
        d = os.path.abspath(_DATA).encode()          # RADRRTM / RADSRTM (data files of the reference)
        L.ref_rrtm_setup(d, C.c_int(len(d)))
        _ref_rrtm = L
    return _ref_rrtm


def _np_from(ptr, shape, dtype=np.float64):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double if dtype == np.float64 else C.c_int32)), shape=(n,)).reshape(shape)


def rrtmg_gas_stage(config, ncol, nlev, cin, nthreads=1):
    """gas_optics + planck_function_atmos/_surf of radiation/radiation_ifs_rrtm.F90:216-852 for all ncol columns:
    RRTM_PREPARE_GASES ... SRTM_GAS_OPTICAL_DEPTH by the reference library, then (numpy) the reversal of the level
    order (:509, :593), max(min_gas_od, .) (:506-512, :590-594), the Planck function from TOTPLNK/DELWAVE (:618-852)
    and the normalisation of the incoming solar flux (:552-560).  Returns dict of (ncol, nlev[+1], ng) arrays;
    lw_emission is the surface Planck term before the (1 - albedo) factor."""
    L = ref_rrtm()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    phl = np.ascontiguousarray(_np_from(cin.pressure_hl, (nlev + 1, ncol)))
    thl = np.ascontiguousarray(_np_from(cin.temperature_hl, (nlev + 1, ncol)))
    if not np.all(phl[1] > phl[0]):
        raise NotImplementedError("the RRTMG oracle glue expects levels ordered from the top")
    gas = _np_from(cin.gas_mixing_ratio, (12, nlev, ncol))
    mu0 = np.ascontiguousarray(_np_from(cin.cos_sza, (ncol,))) if cin.cos_sza else np.zeros(ncol)
    order = [1, 2, 6, 4, 12, 8, 9, 10, 11, 3]      # q co2 ch4 n2o no2 cfc11 cfc12 hcfc22 ccl4 o3 (gas codes)
    gl = [np.ascontiguousarray(gas[k - 1]) for k in order]
    t = np.load(os.path.join(_DATA, "rrtmg_tables.npz"))
    totplnk, delwave = t["yoerrtwn.totplnk"], t["yoerrtwn.delwave"]
    band = np.repeat(np.arange(16), _NG_LW)
    fac = 2.0 * np.arcsin(1.0) * 1.0e4 * delwave[band]

    def planck(T):
        T = np.asarray(T)
        ind = np.where(T >= 339.0, 180, np.where(T >= 160.0, (T - 159.0).astype(int), 1))
        frac = np.where(T >= 339.0, T - 339.0, np.where(T >= 160.0, T - np.trunc(T), 0.0))
        lo = totplnk[ind[..., None] - 1, band]
        return fac * (lo + frac[..., None] * (totplnk[ind[..., None], band] - lo))

    skin = _np_from(cin.skin_temperature, (ncol,)) if cin.skin_temperature else thl[-1]
    out = {"od_lw": np.empty((ncol, nlev, 140)), "planck_hl": np.empty((ncol, nlev + 1, 140)), "lw_emission": np.empty((ncol, 140)),
           "od_sw": np.empty((ncol, nlev, 112)), "ssa_sw": np.empty((ncol, nlev, 112)), "incoming_sw": np.empty((ncol, 112))}
    # A few columns at a time: the reference routines keep (ncol, 140, nlev) automatic arrays on the stack.
    # nthreads > 1: blocks of columns on a thread pool, like the OpenMP loop over blocks of the reference's driver
    # (driver/ecrad_driver.F90:348); the routines only read their module tables, ctypes and numpy release the GIL.
    nblock = 4 if nthreads <= 1 else 8

    def block(c0):
        c1 = min(ncol, c0 + nblock)
        n = c1 - c0
        cut = lambda a: np.ascontiguousarray(a[..., c0:c1])
        o_lw = np.zeros((140, nlev, n), order="F"); o_pf = np.zeros((n, 140, nlev), order="F")
        o_sw = np.zeros((n, nlev, 112), order="F"); o_ssa = np.zeros((n, nlev, 112), order="F"); o_inc = np.zeros((n, 112), order="F")
        args = [cut(phl), cut(thl)] + [cut(a) for a in gl] + [cut(mu0)]
        L.ref_rrtm_gas_optics(C.c_int(n), C.c_int(nlev), *[p(a) for a in args], p(o_lw), p(o_pf), p(o_sw), p(o_ssa), p(o_inc))
        # what radiation_ifs_rrtm.F90 does around the routines, on the block
        out["od_lw"][c0:c1] = np.maximum(np.transpose(o_lw, (2, 1, 0))[:, ::-1, :], config.min_gas_od_lw)
        pf = np.transpose(o_pf, (0, 2, 1))[:, ::-1, :]                               # (n, layer from top, 140)
        out["planck_hl"][c0:c1] = planck(thl[:, c0:c1].T) * np.concatenate([pf[:, :1, :], pf], axis=1)
        out["lw_emission"][c0:c1] = planck(skin[c0:c1]) * pf[:, -1, :]
        out["od_sw"][c0:c1] = np.maximum(o_sw[:, ::-1, :], config.min_gas_od_sw)
        out["ssa_sw"][c0:c1] = o_ssa[:, ::-1, :]
        if cin.spectral_solar_scaling:        # radiation_ifs_rrtm.F90:545-551: per band, before the normalisation
            # (native g-point jg takes the factor of band i_band_from_reordered_g_sw(jg): with SPARTACUS's reordering that
            #  is the band of the g-point AT POSITION jg of the reordered spectrum, as the reference has it)
            ib = getattr(config, "i_band_from_reordered_g_sw", None)
            ib = np.repeat(np.arange(14), _NG_SW) if ib is None or len(ib) != 112 else np.asarray(ib) - 1
            o_inc = o_inc * _np_from(cin.spectral_solar_scaling, (14,))[ib][None, :]
        tot = o_inc.sum(axis=1)
        scale = np.where(mu0[c0:c1] > 0.0, cin.solar_irradiance / np.where(tot > 0, tot, 1.0), 1.0)
        out["incoming_sw"][c0:c1] = o_inc * scale[:, None]

    if nthreads > 1 and ncol > nblock:
        import threading
        from concurrent.futures import ThreadPoolExecutor
        old_size = threading.stack_size(256 * 1024 * 1024)
        try:
            with ThreadPoolExecutor(max_workers=nthreads) as ex:
                list(ex.map(block, range(0, ncol, nblock)))
        finally:
            threading.stack_size(old_size)
    else:
        for c0 in range(0, ncol, nblock):
            block(c0)
    # the g-points of a spectrum SPARTACUS works on, in the reference's order (radiation_ifs_rrtm.F90:480-503, :571-586)
    r = getattr(config, "rrtmg", None)
    for names, perm in ((("od_lw", "planck_hl", "lw_emission"), getattr(r, "i_g_from_reordered_g_lw", None)),
                        (("od_sw", "ssa_sw", "incoming_sw"), getattr(r, "i_g_from_reordered_g_sw", None))):
        if perm is not None:
            for n in names:
                out[n] = np.ascontiguousarray(out[n][..., np.asarray(perm) - 1])
    return out


def make_rrtmg_backend(config, inner=None, nthreads=1):
    """``backend=`` callable for configurations with gas_model_name = "RRTMG-IFS": computes the gas-optics stage with
    the reference's routines, hands it to the C oracle, then runs ``inner`` (default: the plain oracle backend)."""
    from ecrad_amd import abi
    inner = inner or backend

    def _b(cconfig, ncol, nlev, istartcol, iendcol, cin, cflux) -> int:
        stage = rrtmg_gas_stage(config, ncol, nlev, cin, nthreads=nthreads)
        st = abi.Optics()
        for k, a in stage.items():
            setattr(st, k, abi.dptr(a))
        L = getattr(inner, "lib", None) or lib()
        L.ecrad_oracle_set_gas_stage(C.byref(st))
        try:
            return inner(cconfig, ncol, nlev, istartcol, iendcol, cin, cflux)
        finally:
            L.ecrad_oracle_set_gas_stage(None)
    return _b
