"""RRTMG tables on the host side: the contents of the reference's ifsrrtm/yoerrta*, yoesrta*, yoerrtrf,
yoerrtwn, yoesrtwn modules after RRTM_INIT_140GP / SRTM_INIT (radiation_ifs_rrtm.F90:89-99), read from
data/rrtmg_tables.npz (written by oracle/make_rrtm_tables.py from the reference's own initialisation), and
their hand-over to the library as an ``ecrad_rrtmg_t`` (include/ecrad_hip.h).  A Fortran host would pass
c_loc() of the module arrays instead."""
from __future__ import annotations

import os

import numpy as np

from . import abi

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data", "rrtmg_tables.npz")

# band-specific arrays, in the order include/ecrad_hip.h documents
_LW_MINOR = {1: ["ka_mn2", None, None, "kb_mn2"], 3: ["ka_mn2o", None, None, "kb_mn2o"], 5: ["ka_mo3"], 6: ["ka_mco2"],
             7: ["ka_mco2", None, None, "kb_mco2"], 8: ["ka_mco2", "ka_mo3", "ka_mn2o", "kb_mco2", "kb_mn2o"],
             9: ["ka_mn2o", None, None, "kb_mn2o"], 11: ["ka_mo2", None, None, "kb_mo2"],
             13: ["ka_mco2", None, None, "kb_mo3"], 15: ["ka_mn2"]}
_LW_XSEC = {5: ["ccl4"], 6: ["cfc11adj", "cfc12"], 8: ["cfc12", "cfc22adj"]}
_SW_XSEC = {20: ["absch4c"], 24: ["abso3ac", "abso3bc"], 25: ["abso3ac", "abso3bc"], 29: ["absh2oc", "absco2c"]}
_SW_RAYL = {23: ["raylc"], 24: ["raylac", "raylbc"], 25: ["raylc"], 26: ["raylc"], 27: ["raylc"]}
# wavenumber bounds of the bands (radiation_ifs_rrtm.F90:117-121, :159-164), cm-1
SW_WAVENUMBER1 = [2600.0, 3250.0, 4000.0, 4650.0, 5150.0, 6150.0, 7700.0, 8050.0, 12850.0, 16000.0, 22650.0, 29000.0, 38000.0, 820.0]
SW_WAVENUMBER2 = [3250.0, 4000.0, 4650.0, 5150.0, 6150.0, 7700.0, 8050.0, 12850.0, 16000.0, 22650.0, 29000.0, 38000.0, 50000.0, 2600.0]
LW_WAVENUMBER1 = [10.0, 350.0, 500.0, 630.0, 700.0, 820.0, 980.0, 1080.0, 1180.0, 1390.0, 1480.0, 1800.0, 2080.0, 2250.0, 2380.0, 2600.0]
LW_WAVENUMBER2 = [350.0, 500.0, 630.0, 700.0, 820.0, 980.0, 1080.0, 1180.0, 1390.0, 1480.0, 1800.0, 2080.0, 2250.0, 2380.0, 2600.0, 3250.0]


# RRTM_GPOINT_REORDERING_LW / _SW (radiation_ifs_rrtm.F90:51-72; data): RRTMG's g-points in approximately increasing order of gas
# optical depth, the order in which the reference hands them to SPARTACUS (config%i_g_from_reordered_g_*, 1-based)
GPOINT_REORDERING_LW = np.array([
    89, 90, 139, 77, 137, 69, 131, 97, 91, 70, 78, 71, 53, 72, 123, 54, 79, 98, 92, 55,
    80, 132, 124, 81, 73, 56, 99, 82, 57, 23, 125, 100, 24, 74, 93, 58, 25, 83, 126, 75,
    26, 11, 101, 133, 59, 27, 76, 140, 12, 84, 102, 94, 28, 127, 85, 13, 39, 60, 86, 103,
    87, 109, 14, 29, 115, 40, 95, 15, 61, 88, 41, 110, 104, 1, 116, 42, 30, 134, 128, 138,
    96, 62, 16, 43, 117, 63, 111, 44, 2, 64, 31, 65, 105, 17, 45, 66, 118, 32, 3, 33,
    67, 18, 129, 135, 46, 112, 34, 106, 68, 35, 4, 119, 36, 47, 107, 19, 37, 38, 113, 48,
    130, 5, 120, 49, 108, 20, 50, 51, 114, 21, 121, 52, 136, 122, 6, 22, 7, 8, 9, 10,
], dtype=np.int32)
GPOINT_REORDERING_SW = np.array([
    35, 45, 19, 27, 36, 57, 20, 46, 58, 21, 28, 67, 55, 68, 37, 1, 69, 22, 29, 59,
    78, 101, 79, 77, 70, 76, 47, 75, 30, 81, 60, 102, 80, 82, 23, 2, 83, 84, 85, 86,
    103, 61, 31, 87, 56, 38, 71, 48, 88, 3, 62, 89, 24, 7, 49, 32, 104, 72, 90, 63,
    39, 4, 8, 50, 91, 64, 40, 33, 25, 51, 95, 96, 73, 65, 9, 41, 97, 92, 105, 52,
    5, 98, 10, 42, 99, 100, 66, 11, 74, 34, 53, 26, 6, 106, 12, 43, 13, 54, 93, 44,
    107, 94, 14, 108, 15, 16, 109, 17, 18, 110, 111, 112,
], dtype=np.int32)


class RrtmgTables:
    """Holds the arrays (Fortran order, float64) and the ctypes struct that points into them."""

    def __init__(self, path: str = DATA):
        z = np.load(path)
        self.arrays = {}
        self.t = {k: z[k] for k in z.files}
        self.struct = abi.Rrtmg()
        s = self.struct
        s.chi_mls = self._p("yoerrtrf.chi_mls")
        s.preflog_lw = self._p("yoerrtrf.preflog")
        s.tref_lw = self._p("yoerrtrf.tref")
        s.preflog_sw = self._p("yoesrtwn.preflog")
        s.tref_sw = self._p("yoesrtwn.tref")
        s.totplnk = self._p("yoerrtwn.totplnk")
        s.delwave = self._p("yoerrtwn.delwave")
        self.ng_lw = [int(v) for v in self.t["yoerrtftr.ngc"]]
        self.ng_sw = [int(v) for v in self.t["yoesrtwn.ngc"]]
        for ib in range(16):
            self._band(s.lw[ib], f"yoerrta{ib + 1}", ib + 1, self.ng_lw[ib], True)
        for ib in range(14):
            self._band(s.sw[ib], f"yoesrta{ib + 16}", ib + 16, self.ng_sw[ib], False)
        # g-point -> band (1-based), radiation_ifs_rrtm.F90:128, :171
        self.i_band_from_g_lw = np.repeat(np.arange(1, 17), self.ng_lw).astype(np.int32)
        self.i_band_from_g_sw = np.repeat(np.arange(1, 15), self.ng_sw).astype(np.int32)
        self.i_g_from_reordered_g_lw = self.i_g_from_reordered_g_sw = None

    def set_reordering(self, lw: bool, sw: bool) -> None:
        """radiation_ifs_rrtm.F90:122-130, :167-174: the spectrum a SPARTACUS solver works on has its g-points reordered."""
        self.i_g_from_reordered_g_lw = GPOINT_REORDERING_LW.copy() if lw else None
        self.i_g_from_reordered_g_sw = GPOINT_REORDERING_SW.copy() if sw else None
        self.struct.i_g_from_reordered_g_lw = abi.iptr(self.i_g_from_reordered_g_lw) if lw else None
        self.struct.i_g_from_reordered_g_sw = abi.iptr(self.i_g_from_reordered_g_sw) if sw else None

    def _p(self, name, optional=False):
        if name is None or name not in self.t:
            if optional or name is None:
                return None
            raise KeyError(name)
        if name not in self.arrays:
            self.arrays[name] = np.ascontiguousarray(np.asarray(self.t[name], dtype=np.float64).ravel(order="F"))
        return abi.dptr(self.arrays[name])

    def _scalar(self, mod, *names):
        for n in names:
            if f"{mod}.{n}" in self.t:
                return float(self.t[f"{mod}.{n}"])
        return 0.0

    def _band(self, b, mod, band, ng, lw):
        t = self.t
        b.ng = ng
        absa = t.get(f"{mod}.absa")
        b.ld = int(absa.shape[1]) if absa is not None else (ng if lw else 16)
        wn = "yoerrtwn" if lw else "yoesrtwn"
        k = band - 1 if lw else band - 16
        b.nspa = int(t[f"{wn}.nspa"][k])
        b.nspb = int(t[f"{wn}.nspb"][k])
        b.layreffr = int(self._scalar(mod, "layreffr"))
        forname = "forref" if lw else "forrefc"
        b.n_forref = int(t[f"{mod}.{forname}"].shape[0]) if f"{mod}.{forname}" in t else 0
        b.strrat = self._scalar(mod, "strrat", "strrat1")
        b.rayl = self._scalar(mod, "rayl") if not (f"{mod}.rayl" in t and t[f"{mod}.rayl"].shape != ()) else 0.0
        b.factor = self._scalar(mod, "givfac", "scalekur")
        b.absa = self._p(f"{mod}.absa", True)
        b.absb = self._p(f"{mod}.absb", True)
        b.selfref = self._p(f"{mod}.{'selfref' if lw else 'selfrefc'}", True)
        b.forref = self._p(f"{mod}.{forname}", True)
        b.fracrefa = self._p(f"{mod}.{'fracrefa' if lw else 'sfluxrefc'}", True)
        b.fracrefb = self._p(f"{mod}.fracrefb", True) if lw else None
        if lw:
            for i, n in enumerate(_LW_MINOR.get(band, [])):
                b.minor[i] = self._p(f"{mod}.{n}") if n else None
            for i, n in enumerate(_LW_XSEC.get(band, [])):
                b.xsec[i] = self._p(f"{mod}.{n}")
        else:
            for i, n in enumerate(_SW_XSEC.get(band, [])):
                b.xsec[i] = self._p(f"{mod}.{n}")
            for i, n in enumerate(_SW_RAYL.get(band, [])):
                b.rayl_g[i] = self._p(f"{mod}.{n}")
