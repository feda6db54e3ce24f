#!/usr/bin/env python
"""tools/sp_nonfinite.py [ncol] -- BASELINE configs[4] (ecCKD-32, SPARTACUS with 3-D effects, SINGLE precision): which columns come back
with non-finite fluxes from the HIP path and from the oracle's float build, in which g-points, from which level on (the columns in question
are run once more with do_save_spectral_flux, which gives per-g-point profiles), and what the cloud profile looks like there.
The reference's single-precision SPARTACUS is unstable and says so (radiation_config.F90:1144-1148)."""
import copy, json, os, sys
import torch  # noqa: F401  (first: tests/conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ecrad_amd.interface import Radiation
from ecrad_amd.synthetic import make_columns
from ecrad_amd.types import Flux

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
config, clear_sky, desc = bench.build_config("spartacus_ecckd32_sp")
inputs = make_columns(config, ncol, clear_sky)
n, nlev, sl, th, gas, cloud, aer = inputs


def run_hip(cfg, inp):
    rad = Radiation(cfg, backend="hip")
    m, nl, *objs = inp
    objs = [copy.deepcopy(o) for o in objs]
    fl = Flux.allocate(cfg, m, nl)
    rad.radiation(m, nl, 1, m, *objs, fl)
    rad.close()
    return fl


def bad_columns(fl, m):
    bad = np.zeros(m, bool)
    for name, a in fl.arrays.items():
        nf = ~np.isfinite(a)
        if a.shape[-1] == m:
            bad |= nf.reshape(-1, m).any(axis=0)
        elif a.shape[0] == m:
            bad |= nf.reshape(m, -1).any(axis=1)
    return np.flatnonzero(bad)


hip = run_hip(config, inputs)
bad_hip = bad_columns(hip, n)
osp = bench.with_stdout_on_stderr(bench.oracle_flux_of, config, copy.deepcopy(inputs))
bad_ora = bad_columns(osp, n)
print(f"{n} columns: non-finite in {len(bad_hip)} HIP columns {bad_hip.tolist()[:40]}, in {len(bad_ora)} columns of the oracle's float build {bad_ora.tolist()[:40]}; "
      f"in both: {sorted(set(bad_hip) & set(bad_ora))}")
report = {"ncol": n, "hip": bad_hip.tolist(), "oracle_float": bad_ora.tolist(), "columns": {}}
cols = sorted(set(bad_hip.tolist()) | set(bad_ora.tolist()))
if cols:
    for c in cols:
        one = bench.columns_of(inputs, c, 1)

        def spec_config():      # (a fresh configuration per use: Radiation sets it up)
            cc = bench.build_config("spartacus_ecckd32_sp")[0]
            cc.do_save_spectral_flux = True
            cc.do_save_gpoint_flux = True
            return cc
        fh = run_hip(spec_config(), one)
        fo = bench.with_stdout_on_stderr(bench.oracle_flux_of, spec_config(), copy.deepcopy(one))
        rec = {}
        for label, fl in (("hip", fh), ("oracle_float", fo)):
            r = {}
            for name in ("sw_dn_band", "sw_up_band", "lw_dn_band", "lw_up_band"):
                a = fl.arrays.get(name)      # (nlev+1, 1, nspec)
                if a is None:
                    continue
                nf = ~np.isfinite(a[:, 0, :])
                gs = np.flatnonzero(nf.any(axis=0))
                if len(gs):
                    r[name] = {"g_points": gs.tolist(), "first_half_level": {int(g): int(np.flatnonzero(nf[:, g])[0]) for g in gs}}
            rec[label] = r
        frac = one[5].fraction[:, 0]
        cl = np.flatnonzero(frac > 0)
        rec["cloudy_layers"] = cl.tolist()
        rec["cloud_fraction"] = [float(f"{frac[k]:.3g}") for k in cl]
        rec["cos_sza"] = float(one[2].cos_sza[0])
        report["columns"][int(c)] = rec
        print(c, json.dumps(rec))
out = os.path.join(ROOT, "gpurun_out", "sp_nonfinite.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(report, open(out, "w"), indent=1)
