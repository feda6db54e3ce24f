#!/usr/bin/env python
"""tools/sp_column.py COLUMN [LIB ...] -- one column of the synthetic single-precision SPARTACUS workload through several builds of the library
(ECRAD_HIP_LIB must be set per process: this script re-executes itself per LIB) and with cos_sza nudged: does it come back finite?"""
import copy, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("SP_COLUMN_CHILD") != "1":
    col = sys.argv[1]
    for lib in sys.argv[2:] or [os.path.join(ROOT, "ecrad_amd", "csrc", "libecrad_hip.so")]:
        env = dict(os.environ, SP_COLUMN_CHILD="1", ECRAD_HIP_LIB=os.path.abspath(lib))
        p = subprocess.run([sys.executable, os.path.abspath(__file__), col], env=env, capture_output=True, text=True)
        print("==", lib); print(p.stdout[-1500:]); print(p.stderr[-500:] if p.returncode else "")
    sys.exit(0)
import torch  # noqa: F401
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ecrad_amd.interface import Radiation
from ecrad_amd.synthetic import make_columns
from ecrad_amd.types import Flux
c = int(sys.argv[1])
config, clear_sky, desc = bench.build_config("spartacus_ecckd32_sp")
inputs = make_columns(config, 100000, clear_sky)
one = bench.columns_of(inputs, c, 1)
rad = Radiation(config, backend="hip")
for label, scale in (("as is", 1.0), ("cos_sza x (1 + 1e-6)", 1.0 + 1e-6), ("cos_sza x (1 - 1e-6)", 1.0 - 1e-6), ("cos_sza x 1.01", 1.01)):
    m, nl, sl, th, gas, cloud, aer = copy.deepcopy(one)
    sl.cos_sza = sl.cos_sza * scale
    fl = Flux.allocate(config, m, nl)
    rad.radiation(m, nl, 1, m, sl, th, gas, cloud, aer, fl)
    bad = {k: int((~np.isfinite(v)).sum()) for k, v in fl.arrays.items() if not np.isfinite(v).all()}
    gb = np.flatnonzero(~np.isfinite(fl.arrays["sw_up_toa_g"][0])).tolist()
    print(f"column {c}, {label}: non-finite fields {sorted(bad)}; g-points {gb}; sw_up at TOA {fl.arrays['sw_up'][0, 0]:.4f}", flush=True)
    if "sptrace" in os.environ.get("ECRAD_HIP_LIB", ""):      # (a build with -DECRAD_SP_TRACE: 1000 * layer + stage of the first bad quantity per g-point)
        print("   trace codes per g-point:", [round(float(v), 4) for v in fl.arrays["sw_up_toa_g"][0]], flush=True)
        break
# the two spectra apart: is the longwave involved at all?
