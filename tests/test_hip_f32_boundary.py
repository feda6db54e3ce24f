"""ecrad_hip_radiation_f32 (include/ecrad_hip.h): the boundary of a SINGLE-PRECISION host -- the reference built with
-DPARKIND1_SINGLE (ifsaux/parkind1.F90: jprb = real32) passes real32 arrays to radiation() (radiation_interface.F90:200-251).
Same structs, float data behind the pointers; the library widens the columns of the call, and only those, on its side.

What must hold: the float call equals, BIT FOR BIT, the double call on the same (float-representable) inputs rounded to
float; columns outside istartcol..iendcol are neither read for their values nor written; several threads may be in the call
at once (no critical section, no shared copy pool)."""
import ctypes as C
import threading

import numpy as np
import pytest

from ecrad_amd import abi
from ecrad_amd.interface import Radiation, build_flux_struct, build_inputs_struct
from ecrad_amd.synthetic import make_columns
from ecrad_amd.types import Flux
from helpers import make_config

pytestmark = pytest.mark.gpu


def _representable(objs):
    """Every float64 array of the input objects rounded to a float32-representable value (in place)."""
    for o in objs:
        if o is None:
            continue
        for k, v in vars(o).items():
            if isinstance(v, np.ndarray) and v.dtype == np.float64:
                v[...] = v.astype(np.float32).astype(np.float64)


def _float_twin(struct, arrays):
    """A copy of an ecrad_inputs_t / ecrad_flux_t whose real arrays are float32 copies of `arrays` (matched by address).
    Returns (struct, {address of the double array: its float32 copy})."""
    twin = type(struct)()
    C.memmove(C.byref(twin), C.byref(struct), C.sizeof(struct))
    by_addr = {a.ctypes.data: a for a in arrays}
    copies = {}
    for name, ctype in struct._fields_:
        if ctype is not abi.c_double_p:
            continue
        p = getattr(struct, name)
        if not p:
            continue
        addr = C.addressof(p.contents)
        a32 = copies.get(addr)
        if a32 is None:
            a32 = copies[addr] = np.ascontiguousarray(by_addr[addr].astype(np.float32))
        setattr(twin, name, C.cast(a32.ctypes.data, abi.c_double_p))
    return twin, copies


@pytest.mark.parametrize("solver", ["Tripleclouds", "McICA", "Homogeneous"])
def test_float_call_is_the_double_call_rounded_to_float(solver):
    ncol, i0, i1 = 640, 101, 420
    config = make_config(solver)
    n, nlev, sl, th, gas, cloud, aer = make_columns(config, ncol, False)
    _representable((sl, th, gas, cloud, aer))
    rad = Radiation(config, backend="hip")
    frac0 = cloud.fraction.copy()
    # the double call on the range
    f64 = Flux.allocate(config, n, nlev)
    for a in f64.arrays.values():
        a[...] = -77.0
    rad.radiation(n, nlev, i0, i1, sl, th, gas, cloud, aer, f64)
    frac64 = cloud.fraction.copy()
    cloud.fraction[...] = frac0
    # the float call: float32 twins of every array
    cin, keep = build_inputs_struct(config, n, nlev, sl, th, gas, cloud, aer)
    cin32, in32 = _float_twin(cin, keep + [cloud.fraction])
    f32 = Flux.allocate(config, n, nlev)
    for a in f32.arrays.values():
        a[...] = -77.0
    cfl = build_flux_struct(f32)
    cfl32, out32 = _float_twin(cfl, list(f32.arrays.values()))
    # values outside the range that would poison any result that read them
    col_last = {a.ctypes.data: (a.shape[-1] == n) for a in keep + [cloud.fraction]}
    before = {}
    for addr, a32 in in32.items():
        if col_last.get(addr) and a32.ndim >= 1:
            a32[..., : i0 - 1] = np.nan
            a32[..., i1:] = np.nan
        before[addr] = a32.copy()
    st = rad.lib.ecrad_hip_radiation_f32(rad.handle, n, nlev, i0, i1, C.byref(cin32), C.byref(cfl32))
    assert st == 0, rad.lib.ecrad_hip_last_error(rad.handle).decode()
    sl_cols = slice(i0 - 1, i1)
    for name, ref in f64.arrays.items():
        got = out32[f32.arrays[name].ctypes.data]
        assert got.dtype == np.float32
        col_is_last = ref.shape[-1] == n
        r_in = ref[..., sl_cols] if col_is_last else ref[sl_cols]
        g_in = got[..., sl_cols] if col_is_last else got[sl_cols]
        assert np.array_equal(g_in, r_in.astype(np.float32), equal_nan=True), name
        # outside the range: what the caller had put there
        g_out = np.delete(got, np.s_[i0 - 1:i1], axis=-1 if col_is_last else 0)
        assert np.all(g_out == np.float32(-77.0)), name
    # inputs: untouched everywhere, except the cropped cloud fraction inside the range
    frac32 = in32[cloud.fraction.ctypes.data]
    for addr, a32 in in32.items():
        if a32 is frac32:
            continue
        assert np.array_equal(a32, before[addr], equal_nan=True)
    assert np.array_equal(frac32[:, sl_cols], frac64[:, sl_cols].astype(np.float32))
    assert np.all(np.isnan(frac32[:, : i0 - 1])) and np.all(np.isnan(frac32[:, i1:]))
    rad.close()


def test_concurrent_float_blocks():
    """16 threads, blocks of 80 columns of shared float arrays (the reference driver's OpenMP loop, nblocksize = 80, in a
    PARKIND1_SINGLE build): the same bits as one float call over all columns, and the calls did run side by side."""
    ncol, nblock, nthreads = 80 * 64, 80, 16
    config = make_config("Tripleclouds")
    n, nlev, sl, th, gas, cloud, aer = make_columns(config, ncol, False)
    _representable((sl, th, gas, cloud, aer))
    rad = Radiation(config, backend="hip", concurrency=(1, 16))
    frac0 = cloud.fraction.copy()
    cin, keep = build_inputs_struct(config, n, nlev, sl, th, gas, cloud, aer)

    def run(blocks, nt):
        cloud.fraction[...] = frac0
        cin32, in32 = _float_twin(cin, keep + [cloud.fraction])
        flux = Flux.allocate(config, n, nlev)
        cfl32, out32 = _float_twin(build_flux_struct(flux), list(flux.arrays.values()))
        todo, lock, errors = list(blocks), threading.Lock(), []

        def worker():
            while True:
                with lock:
                    if not todo:
                        return
                    a, b = todo.pop()
                if rad.lib.ecrad_hip_radiation_f32(rad.handle, n, nlev, a, b, C.byref(cin32), C.byref(cfl32)) != 0:
                    errors.append(rad.lib.ecrad_hip_last_error(rad.handle))
                    return
        threads = [threading.Thread(target=worker) for _ in range(nt)]
        rad.pool_info(reset=True)
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        return {name: out32[a.ctypes.data] for name, a in flux.arrays.items()}, in32[cloud.fraction.ctypes.data], rad.pool_info()
    whole, frac_w, _ = run([(1, n)], 1)
    blocks, frac_b, pool = run([(i + 1, i + nblock) for i in range(0, n, nblock)], nthreads)
    assert pool["calls_total"] == n // nblock and pool["max_in_flight"] >= 8, pool
    for name, ref in whole.items():
        assert np.array_equal(ref, blocks[name], equal_nan=True), name
    assert np.array_equal(frac_w, frac_b)
    rad.close()


def test_float_call_rejects_device_memory():
    config = make_config("Homogeneous", use_aerosols=False)
    n, nlev, sl, th, gas, cloud, aer = make_columns(config, 64, True)
    rad = Radiation(config, backend="hip")
    cin, keep = build_inputs_struct(config, n, nlev, sl, th, gas, cloud, aer)
    flux = Flux.allocate(config, n, nlev)
    cfl = build_flux_struct(flux)
    cin.memory = abi.MEM_DEVICE
    assert rad.lib.ecrad_hip_radiation_f32(rad.handle, n, nlev, 1, n, C.byref(cin), C.byref(cfl)) == -3      # ECRAD_EUNSUPPORTED
    assert b"host arrays" in rad.lib.ecrad_hip_last_error(rad.handle)
    rad.close()
