"""tools/small_call_latency.py -- what an NPROMA-blocked host sees: ecrad_hip_radiation called block by block.

  python tools/small_call_latency.py                     device-memory mode, one caller: time per call for 32 ... 16384 columns
  python tools/small_call_latency.py --host [--ncol 80] [--threads 1,4,16] [--contexts 16] [--solver Tripleclouds]
                                                         HOST-memory mode (the mode every Fortran host uses): T threads each
                                                         calling radiation() on its own blocks of `ncol` columns of shared
                                                         arrays, as the reference's driver does with `!$OMP PARALLEL DO`
                                                         (driver/ecrad_driver.F90:348-370); columns/s per thread count, and how
                                                         many calls the library's pool of contexts had in flight at once"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ecrad_amd.cases import make_config  # noqa: E402
from ecrad_amd.interface import Radiation, build_flux_struct, build_inputs_struct  # noqa: E402
from ecrad_amd.synthetic import make_columns  # noqa: E402
from ecrad_amd.types import Flux  # noqa: E402


def device_mode():
    import torch
    from ecrad_amd.device import DeviceCase
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


def host_mode(solver, nblock, thread_counts, contexts, nblocks_per_thread=48, clear=False, devices=1):
    """Returns {threads: columns/s}.  The ctypes structs are built once (what the Fortran wrapper's c_loc() calls cost is
    not what is measured here); a call = ecrad_hip_radiation(handle, ncol, nlev, i0, i1, inputs, flux), GIL released."""
    config = make_config(solver)
    out = {}
    tmax = max(thread_counts)
    ncol = nblock * nblocks_per_thread * tmax
    rad = Radiation(config, backend="hip", concurrency=(devices, contexts))      # (setup_radiation: the structs below need the configured sizes)
    n, nlev, sl, th, gas, cloud, aer = make_columns(config, ncol, clear)
    frac0 = None if cloud is None else cloud.fraction.copy()
    flux = Flux.allocate(config, n, nlev)
    cin, keep = build_inputs_struct(config, n, nlev, sl, th, gas, cloud, aer)
    cflux = build_flux_struct(flux)
    for nthreads in thread_counts:
        if frac0 is not None:
            cloud.fraction[...] = frac0
        blocks = [(i + 1, i + nblock) for i in range(0, nblock * nblocks_per_thread * nthreads, nblock)]
        # warm-up: every context allocates its work arrays on its first call
        warm = threading.Barrier(min(nthreads, contexts * max(devices, 1)))

        def warm_up(k):
            warm.wait()
            rad.lib.ecrad_hip_radiation(rad.handle, n, nlev, blocks[k][0], blocks[k][1], C.byref(cin), C.byref(cflux))
        ws = [threading.Thread(target=warm_up, args=(k,)) for k in range(warm.parties)]
        for w in ws:
            w.start()
        for w in ws:
            w.join()
        for rnd in range(2):      # (the first round is a warm-up under the same load: the batches' buffers reach their size)
            rad.pool_info(reset=True)
            lock = threading.Lock()
            todo = list(blocks)
            errors = []
            start = threading.Barrier(nthreads + 1)

            def worker():
                start.wait()
                while True:
                    with lock:
                        if not todo:
                            return
                        i0, i1 = todo.pop()
                    st = rad.lib.ecrad_hip_radiation(rad.handle, n, nlev, i0, i1, C.byref(cin), C.byref(cflux))
                    if st != 0:
                        errors.append(rad.lib.ecrad_hip_last_error(rad.handle))
                        return
            threads = [threading.Thread(target=worker) for _ in range(nthreads)]
            for t in threads:
                t.start()
            start.wait()
            t0 = time.perf_counter()
            for t in threads:
                t.join()
            dt = time.perf_counter() - t0
            assert not errors, errors
        info = rad.pool_info()
        rate = len(blocks) * nblock / dt
        out[nthreads] = rate
        print(f"{solver:13s} host memory, blocks of {nblock} columns, {nthreads:3d} threads, {info['n_contexts']} contexts on {info['n_devices']} device(s): "
              f"{dt / len(blocks) * nthreads * 1e3:7.3f} ms per call, {rate:10.0f} columns/s, max calls in flight {info['max_in_flight']}, "
              f"{info['batches_total']} batches, calls per device {info['calls_on_device']}")
    rad.close()
    del keep
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--host", action="store_true")
    ap.add_argument("--ncol", type=int, default=80, help="columns per call (the reference's test namelist: nblocksize = 80)")
    ap.add_argument("--threads", default="1,4,16")
    ap.add_argument("--contexts", type=int, default=16)
    ap.add_argument("--devices", type=int, default=1, help="devices of the pool (0 = all visible)")
    ap.add_argument("--solver", default="Tripleclouds")
    args = ap.parse_args()
    if not args.host:
        return device_mode()
    host_mode(args.solver, args.ncol, [int(t) for t in args.threads.split(",")], args.contexts, devices=args.devices)


if __name__ == "__main__":
    main()
