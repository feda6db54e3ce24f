"""(GPU box) ncol = 80 arrays of which only columns 1..32 are valid, garbage beyond (what the reference's NPROMA-blocked IFS driver
hands over): does anything beyond iendcol leak into the results?  One input array at a time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import copy
import numpy as np
from helpers import make_config_rrtmg, make_config, load_meridian
from ecrad_amd.interface import Radiation
from ecrad_amd.types import Flux

def run(config, inputs, ncol_total, garbage_for=None, seed=0):
    ncol, nlev, sl, th, gas, cloud, aer = inputs
    rng = np.random.default_rng(seed)
    objs = [copy.deepcopy(o) for o in (sl, th, gas, cloud, aer)]
    for obj in objs:
        if obj is None: continue
        for k, v in list(vars(obj).items()):
            if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == ncol:
                pad_shape = v.shape[:-1] + (ncol_total - ncol,)
                if garbage_for is None or garbage_for == k or garbage_for == "all":
                    if np.issubdtype(v.dtype, np.floating):
                        pad = rng.uniform(-1e3, 1e3, size=pad_shape) if garbage_for is not None else np.zeros(pad_shape)
                    else:
                        pad = rng.integers(0, 1000, size=pad_shape).astype(v.dtype) if garbage_for is not None else np.zeros(pad_shape, dtype=v.dtype)
                else:
                    pad = np.repeat(v[..., -1:], ncol_total - ncol, axis=-1)      # benign: copies of the last valid column
                setattr(obj, k, np.ascontiguousarray(np.concatenate([v, pad.astype(v.dtype)], axis=-1)))
    rad = Radiation(config, backend="hip")
    flux = Flux.allocate(config, ncol_total, nlev)
    rad.radiation(ncol_total, nlev, 1, ncol, *objs, flux)
    rad.close()
    return {k: (a[..., :ncol] if a.shape[-1] == ncol_total else a[:ncol]).copy() for k, a in flux.arrays.items()}

for label, cfgf in (("tripleclouds_rrtmg", lambda: make_config_rrtmg("Tripleclouds")), ("tripleclouds_ecckd", lambda: make_config("Tripleclouds"))):
    config = cfgf()
    inputs = load_meridian(config)
    rad0 = Radiation(config, backend="hip"); rad0.set_gas_units(inputs[4]); rad0.close()
    inputs[3].calc_saturation_wrt_liquid()
    base = run(cfgf(), inputs, 80, garbage_for=None)
    names = ["all"] + sorted({k for o in inputs[2:] if o is not None for k, v in vars(o).items() if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == inputs[0]})
    for g in names:
        out = run(cfgf(), inputs, 80, garbage_for=g, seed=1)
        bad = {}
        for k in base:
            d = np.abs(out[k] - base[k])
            with np.errstate(invalid="ignore"):
                m = np.nanmax(d / (np.abs(base[k]).max() + 1e-300)) if d.size else 0.0
            if not (m < 1e-12): bad[k] = float(m)
        print(label, "garbage in", g, "->", ("LEAK " + str(dict(list(bad.items())[:4]))) if bad else "clean")
