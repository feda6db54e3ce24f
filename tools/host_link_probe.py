#!/usr/bin/env python
"""tools/host_link_probe.py [torch|notorch] [workload] [ncol] -- what the host link gives a host-memory call, with and without PyTorch's
bundled copy of the HIP runtime in the process (tests/conftest.py: whichever copy is loaded first serves everything; a Fortran host has
/opt/rocm's only).  Prints ecrad_hip_pcie_bandwidth (one way, both ways at once) and the rate of the pipelined call on pageable arrays and on
page-locked arrays (ecrad_hip_host_alloc)."""
import ctypes as C, os, sys, time
which = sys.argv[1] if len(sys.argv) > 1 else "notorch"
if which == "torch":
    import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from helpers import make_config
from ecrad_amd.interface import HostArrays, Radiation, build_flux_struct, build_inputs_struct, relocate_call_arrays
from ecrad_amd.synthetic import BENCH_CONFIGS, make_columns
from ecrad_amd.types import Flux

workload = sys.argv[2] if len(sys.argv) > 2 else "clear_homogeneous_ecckd32"
ncol = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
spec = dict(BENCH_CONFIGS[workload]); clear = spec.pop("clear_sky"); solver = spec.pop("sw_solver")
config = make_config(solver, **spec)
rad = Radiation(config, backend="hip")
maps = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l]
print(which, "HIP runtime mapped:", sorted(set(maps)))
a, b, c = C.c_double(), C.c_double(), C.c_double()
assert rad.lib.ecrad_hip_pcie_bandwidth(rad.handle, C.c_size_t(1 << 30), 3, C.byref(a), C.byref(b), C.byref(c)) == 0
print(f"{which}: link host->device {a.value:.1f} GB/s, device->host {b.value:.1f} GB/s, both at once {c.value:.1f} GB/s")
n, nlev, sl, th, gas, cloud, aer = make_columns(config, ncol, clear)
flux = Flux.allocate(config, n, nlev)
frac0 = None if cloud is None else cloud.fraction.copy()


def timed(call, label):
    best = 1e9
    for rep in range(4):
        if frac0 is not None:
            cloud.fraction[...] = frac0
        t0 = time.perf_counter()
        call()
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    print(f"{which}: {workload} {label}: {best*1e3:.1f} ms -> {n/best:.0f} columns/s", flush=True)


timed(lambda: rad.radiation(n, nlev, 1, n, sl, th, gas, cloud, aer, flux), "pageable arrays")
arena = HostArrays(rad)
relocate_call_arrays(arena.copy_of, (sl, th, gas, cloud, aer), flux)
cin, keep = build_inputs_struct(config, n, nlev, sl, th, gas, cloud, aer)
cflux = build_flux_struct(flux)


def call():
    assert rad.lib.ecrad_hip_radiation(rad.handle, n, nlev, 1, n, C.byref(cin), C.byref(cflux)) == 0


timed(call, "page-locked arrays")
