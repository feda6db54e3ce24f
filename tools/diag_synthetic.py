"""Which field / column / level is behind the largest HIP-vs-oracle difference on the synthetic bench columns?
usage: python tools/diag_synthetic.py WORKLOAD [ncol] [ncheck]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401

from bench import build_config, first_columns, oracle_backend
from ecrad_amd.device import DeviceCase
from ecrad_amd.interface import Radiation
from ecrad_amd.synthetic import make_columns
from ecrad_amd.types import Flux

workload = sys.argv[1]
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
ncheck = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
config, clear_sky, _ = build_config(workload)
rad = Radiation(config, backend="hip")
inputs = make_columns(config, ncol, clear_sky)
n, nlev, sl, th, gas, cloud, aer = inputs
flux = Flux.allocate(config, n, nlev)
case = DeviceCase(config, n, nlev, sl, th, gas, cloud, aer, flux)
assert rad.lib.ecrad_hip_radiation(rad.handle, n, nlev, 1, n, C.byref(case.inputs), C.byref(case.flux)) == 0
rad.lib.ecrad_hip_synchronize(rad.handle)
case.flux_to_host(flux)
sample = first_columns(inputs, ncheck)
config2, _, _ = build_config(workload)
orad = Radiation(config2, backend=oracle_backend(config2)[0])
m = sample[0]
oflux = Flux.allocate(config2, m, nlev)
orad.radiation(m, nlev, 1, m, *sample[2:], oflux)
rows = []
for name, ref in oflux.arrays.items():
    got = flux.arrays[name]
    got = got[..., :m] if got.shape[-1] == n else got[:m]
    scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max() + 1e-300)
    err = np.abs(got - ref) / scale
    idx = np.unravel_index(np.argmax(err), err.shape)
    rows.append((float(err[idx]), name, idx, float(got[idx]), float(ref[idx])))
for r in sorted(rows, reverse=True)[:12]:
    print("%.3e  %-28s index %s  hip %.17g  oracle %.17g" % r)
