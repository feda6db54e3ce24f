"""tools/concurrent_probe.py WORKLOAD [NCOL] -- (tuning) does the device have room for two calls of a workload side by side?
Two handles, each with its own stream and half of the columns in device memory, called from two threads; against one handle with all of them."""
import ctypes as C
import sys
import threading
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bench import build_config  # noqa: E402
from ecrad_amd.device import DeviceCase  # noqa: E402
from ecrad_amd.interface import Radiation  # noqa: E402
from ecrad_amd.synthetic import make_columns  # noqa: E402
from ecrad_amd.types import Flux  # noqa: E402


def setup(name, ncol, first):
    config, clear_sky, _ = build_config(name)
    rad = Radiation(config, backend="hip")
    st = torch.cuda.Stream()
    rad.lib.ecrad_hip_set_stream(rad.handle, C.c_void_p(st.cuda_stream))
    n, nlev, sl, th, gas, cloud, aer = make_columns(config, ncol, clear_sky, first_column=first)
    flux = Flux.allocate(config, n, nlev)
    case = DeviceCase(config, n, nlev, sl, th, gas, cloud, aer, flux)
    frac0 = case.tensors["cloud_fraction"].clone() if "cloud_fraction" in case.tensors else None

    def call():
        if frac0 is not None:
            with torch.cuda.stream(st):
                case.tensors["cloud_fraction"].copy_(frac0)
        assert rad.lib.ecrad_hip_radiation(rad.handle, n, nlev, 1, n, C.byref(case.inputs), C.byref(case.flux)) == 0
    return rad, st, call, case


def timed(calls, streams, reps=6):
    best = 1e9
    for r in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=c) for c in calls]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for s in streams:
            s.synchronize()
        dt = time.perf_counter() - t0
        if r >= 2:
            best = min(best, dt)
    return best


name = sys.argv[1]
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
r0, s0, c0, k0 = setup(name, ncol, 0)
t1 = timed([c0], [s0])
print(f"{name}: one call of {ncol} columns: {1e3 * t1:.2f} ms", flush=True)
r0.close()
del k0
ra, sa, ca, ka = setup(name, ncol // 2, 0)
rb, sb, cb, kb = setup(name, ncol // 2, ncol // 2)
ta = timed([ca], [sa])
t2 = timed([ca, cb], [sa, sb])
print(f"{name}: one call of {ncol // 2}: {1e3 * ta:.2f} ms; two calls of {ncol // 2} side by side: {1e3 * t2:.2f} ms", flush=True)
