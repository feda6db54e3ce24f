"""Diagnostic: where does lw_derivatives of the HIP path differ from the oracle (Tripleclouds, meridian slice)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from helpers import make_config, run_case

c = make_config("Tripleclouds", use_aerosols=False)
f_hip, _, _ = run_case(c, "hip")
from oracle import pyoracle
pyoracle.build(ref=False)
f_or, _, _ = run_case(make_config("Tripleclouds", use_aerosols=False), pyoracle.backend)
a, b = f_hip.arrays["lw_derivatives"], f_or.arrays["lw_derivatives"]
d = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
bad = np.where(d.max(axis=0) > 1e-8)[0]
print("bad columns", bad)
cf = None
inp = __import__("helpers").load_meridian(c)
frac = inp[5].fraction
for col in bad[:6]:
    cl = np.where(frac[:, col] > 0)[0]
    print("   cloudy layers", cl.min() if len(cl) else None, cl.max() if len(cl) else None, "n", len(cl))
    lev = np.where(d[:, col] > 1e-8)[0]
    print("col", col, "first bad half-level (from top)", lev.min(), "last", lev.max(), "max", d[:, col].max())
    print("   hip", a[lev.max() - 2:lev.max() + 3, col], "\n   ora", b[lev.max() - 2:lev.max() + 3, col])
