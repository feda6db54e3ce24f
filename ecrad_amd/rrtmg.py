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
