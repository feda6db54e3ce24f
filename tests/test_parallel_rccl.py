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
