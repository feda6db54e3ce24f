"""tools/small_call_latency.py -- time of one ecrad_hip_radiation call on small batches (device-memory mode): what an
NPROMA-blocked host (ifs/radiation_scheme.F90 called block by block) would see per block."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ecrad_amd.cases import make_config  # noqa: E402
from ecrad_amd.device import DeviceCase  # noqa: E402
from ecrad_amd.interface import Radiation  # noqa: E402
from ecrad_amd.synthetic import make_columns  # noqa: E402
from ecrad_amd.types import Flux  # noqa: E402


def main():
    for solver in ("Tripleclouds", "McICA"):
        for ncol in (32, 256, 1024, 4096, 16384):
            config = make_config(solver)
            rad = Radiation(config, backend="hip")
            rad.lib.ecrad_hip_set_stream(rad.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            n, nlev, sl, th, gas, cloud, aer = make_columns(config, ncol, False)
            flux = Flux.allocate(config, n, nlev)
            case = DeviceCase(config, n, nlev, sl, th, gas, cloud, aer, flux)
            call = lambda: rad.lib.ecrad_hip_radiation(rad.handle, n, nlev, 1, n, C.byref(case.inputs), C.byref(case.flux))
            for _ in range(5):
                assert call() == 0
            torch.cuda.synchronize()
            reps = 50
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            t_enqueue = (time.perf_counter() - t0) / reps
            torch.cuda.synchronize()
            t_total = (time.perf_counter() - t0) / reps
            print(f"{solver:13s} ncol {ncol:6d}: {t_total*1e3:8.3f} ms per call ({t_enqueue*1e3:7.3f} ms to enqueue)  {ncol/t_total:12.0f} columns/s")
            rad.close()


if __name__ == "__main__":
    main()
