#!/usr/bin/env python
"""End-to-end rate of ecrad_hip_radiation in HOST-memory mode (inputs staged H2D, outputs D2H over PCIe
inside the call) for the bench.py workload -- the number DESIGN.md section 7 quotes next to the
HBM-resident `value`.  Usage: python tools/host_mode_rate.py [workload] [ncol]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import make_config
from ecrad_amd.interface import Radiation
from ecrad_amd.synthetic import BENCH_CONFIGS, make_columns
from ecrad_amd.types import Flux

workload = sys.argv[1] if len(sys.argv) > 1 else "clear_homogeneous_ecckd32"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
spec = dict(BENCH_CONFIGS[workload]); clear = spec.pop("clear_sky"); solver = spec.pop("sw_solver")
config = make_config(solver, **spec)
rad = Radiation(config, backend="hip")
n, nlev, sl, th, gas, cloud, aer = make_columns(config, ncol, clear)
flux = Flux.allocate(config, n, nlev)
frac0 = None if cloud is None else cloud.fraction.copy()
for rep in range(4):
    if frac0 is not None:
        cloud.fraction[...] = frac0
    t0 = time.perf_counter()
    rad.radiation(n, nlev, 1, n, sl, th, gas, cloud, aer, flux)
    dt = time.perf_counter() - t0
    print(f"{workload}: host-memory call {rep}: {dt*1e3:.1f} ms -> {n/dt:.0f} columns/s")
