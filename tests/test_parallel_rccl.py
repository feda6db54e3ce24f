"""The one collective of the path on the hardware that exists: `gather_profiles` / `assemble` with the `nccl` backend
(= RCCL on ROCm) at world size 1, on device tensors filled by the HIP path through the C-ABI.  (World sizes 2+ are
covered on CPU with gloo in test_parallel_gloo.py and on GPUs by the driver's scaling run of bench.py, whose
`value_with_gather` times this same code.)"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "{port}"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from ecrad_amd.cases import load_meridian, make_config
    from ecrad_amd.device import DeviceCase
    from ecrad_amd.interface import Radiation
    from ecrad_amd.parallel import assemble, gather_profiles, pack_profiles, shard_range
    from ecrad_amd.types import Flux
    config = make_config("Tripleclouds")
    rad = Radiation(config, backend="hip")
    ncol, nlev, sl, th, gas, cloud, aer = load_meridian(config)
    rad.set_gas_units(gas); th.calc_saturation_wrt_liquid()
    flux = Flux.allocate(config, ncol, nlev)
    case = DeviceCase(config, ncol, nlev, sl, th, gas, cloud, aer, flux)
    rad.lib.ecrad_hip_set_stream(rad.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rad.lib.ecrad_hip_radiation(rad.handle, ncol, nlev, 1, ncol, C.byref(case.inputs), C.byref(case.flux)) == 0
    names = ["lw_up", "lw_dn", "sw_up", "sw_dn", "sw_dn_direct", "lw_up_clear", "sw_dn_clear", "lw_derivatives"]
    i0, i1 = shard_range(ncol, 0, 1)
    packed = pack_profiles(case.flux_tensors, names)          # enqueued behind the kernels on the same stream
    assert packed.is_cuda
    bufs, work = gather_profiles(packed, [ncol], dst=0, async_op=True)
    work.wait()
    out = assemble(bufs, [ncol])
    torch.cuda.synchronize()
    want = torch.stack([case.flux_tensors[n] for n in names])
    assert out.is_cuda and out.shape == want.shape and torch.equal(out, want)
    assert float(out.abs().max()) > 100.0
    dist.barrier(); dist.destroy_process_group(); rad.close()
    print("RCCL gather OK", tuple(out.shape))
""")


def test_rccl_gather_of_device_profiles_world_size_one():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT, port=port)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "RCCL gather OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


# ---- the library's own gather (include/ecrad_hip.h: ecrad_hip_comm_id / _comm_init / ecrad_hip_gather_profiles), no torch in the process ----
LIB_SCRIPT = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    from ecrad_amd import abi
    from ecrad_amd.cases import make_config
    from ecrad_amd.interface import Radiation
    from ecrad_amd.parallel import library_gather_profiles
    rad = Radiation(make_config("Cloudless"), backend="hip")
    assert "torch" not in sys.modules
    lib, h = rad.lib, rad.handle
    # before there is a communicator: a status and a text, not a crash
    one = np.zeros((3, 4))
    ptr = (C.c_void_p * 1)(one.ctypes.data)
    cnt = (C.c_int * 1)(4)
    assert lib.ecrad_hip_gather_profiles(h, 1, ptr, ptr, 3, 4, cnt, 0, abi.MEM_HOST) == abi.EINVAL
    assert b"no communicator" in lib.ecrad_hip_last_error(h)
    ident = (C.c_ubyte * abi.COMM_ID_BYTES)()
    assert lib.ecrad_hip_comm_id(h, ident) == 0, lib.ecrad_hip_last_error(h)
    assert any(ident)
    assert lib.ecrad_hip_comm_init(h, ident, 1, 1) == abi.EINVAL          # rank out of range
    assert lib.ecrad_hip_comm_init(h, ident, 0, 1) == 0, lib.ecrad_hip_last_error(h)
    assert lib.ecrad_hip_comm_init(h, ident, 0, 1) == abi.EINVAL          # already in a communicator
    rng = np.random.default_rng(3)
    nrows, ncol = 138, 1000
    fields = [rng.standard_normal((nrows, ncol)) for _ in range(5)]
    # host arrays: every field goes through the device and RCCL (the root sends to itself) and comes back in place
    got = library_gather_profiles(rad, fields, [ncol], root=0)
    assert all(np.array_equal(a, b) for a, b in zip(fields, got))
    # device arrays (hipMalloc / hipMemcpy of the runtime the library has loaded, through ctypes)
    hip = C.CDLL("libamdhip64.so.7")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def dev(nbytes):
        p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), nbytes) == 0; return p.value
    dl = [dev(f.nbytes) for f in fields]; dg = [dev(f.nbytes) for f in fields]
    for p, f in zip(dl, fields):
        assert hip.hipMemcpy(p, f.ctypes.data, f.nbytes, 1) == 0
    lp = (C.c_void_p * 5)(*dl); gp = (C.c_void_p * 5)(*dg)
    cnt = (C.c_int * 1)(ncol)
    assert lib.ecrad_hip_gather_profiles(h, 5, lp, gp, nrows, ncol, cnt, 0, abi.MEM_DEVICE) == 0, lib.ecrad_hip_last_error(h)
    for p, f in zip(dg, fields):
        back = np.empty_like(f)
        assert hip.hipMemcpy(back.ctypes.data, p, f.nbytes, 2) == 0
        assert np.array_equal(back, f)
    # wrong counts are refused
    bad = (C.c_int * 1)(ncol + 1)
    assert lib.ecrad_hip_gather_profiles(h, 5, lp, gp, nrows, ncol, bad, 0, abi.MEM_DEVICE) == abi.EINVAL
    assert lib.ecrad_hip_comm_destroy(h) == 0 and lib.ecrad_hip_comm_destroy(h) == 0
    for p in dl + dg:
        hip.hipFree(p)
    rad.close()
    print("library gather OK")
""")


def test_library_gather_over_rccl_world_size_one():
    """ecrad_hip_gather_profiles at world size 1 in a process WITHOUT torch (what a Fortran host is): librccl loaded by the library,
    the root's share sent to itself over RCCL, host and device memory.  World sizes 2+ need as many GPUs (RCCL refuses two ranks on one
    device); the assembly arithmetic of several ranks is covered on CPU in test_parallel_gloo.py."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", LIB_SCRIPT.format(root=ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "library gather OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
