"""Stage arrays of the HIP path vs the oracle for given (column, g) of the synthetic clear-sky workload."""
import sys, ctypes as C, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import build_config
from ecrad_amd import abi
from ecrad_amd.interface import Radiation, build_inputs_struct
from ecrad_amd.synthetic import make_columns
from oracle import pyoracle
config, clear, _ = build_config("clear_homogeneous_ecckd32")
rad = Radiation(config, backend="hip")
inputs = make_columns(config, 2048, clear)
n, nlev, sl, th, gas, cloud, aer = inputs
cin, keep = build_inputs_struct(config, n, nlev, sl, th, gas, cloud, aer)
out = abi.Optics()
got = {k: np.zeros(v) for k, v in pyoracle.optics_shapes(config, nlev, n).items()}
for k, a in got.items():
    setattr(out, k, abi.dptr(a))
assert rad.lib.ecrad_hip_optics(rad.handle, n, nlev, 1, n, C.byref(cin), C.byref(out)) == 0
want = pyoracle.optics(config, rad.cconfig, n, nlev, 1, n, cin)
for c, g in ((196, 2), (86, 3)):
    for k in ("od_sw", "ssa_sw", "incoming_sw", "sw_albedo_diffuse", "sw_albedo_direct"):
        a, b = got[k][c, ..., g], want[k][c, ..., g]
        rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        print(c, g, k, "max rel", rel.max(), "at level", int(np.argmax(rel)) if rel.ndim else "-", "value", np.ravel(b)[int(np.argmax(rel))] if rel.ndim else b)
    print("  od range", want["od_sw"][c, :, g].min(), want["od_sw"][c, :, g].max(), "mu0", sl.cos_sza[c])
