"""Column sharding across the GPUs of one node and reassembly of flux profiles.

Columns are independent end to end (no horizontal coupling in any implemented solver), so the path
shards embarrassingly: rank r owns the contiguous range [r*ncol/N, (r+1)*ncol/N) -- the same
istartcol/iendcol convention as the reference driver's blocks (driver/ecrad_driver.F90:351-354) --
and look-up tables are replicated.  The ONLY collective is the gather of flux profiles to rank 0
(RCCL over xGMI when the tensors live on GPUs; gloo in the CPU tests).  No reduction is ever needed.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np


def shard_range(ncol: int, rank: int, world_size: int):
    """1-based inclusive (istartcol, iendcol) of the columns owned by `rank`; balanced to +-1."""
    base, rem = divmod(ncol, world_size)
    start = rank * base + min(rank, rem)
    n = base + (1 if rank < rem else 0)
    return start + 1, start + n


def pack_profiles(flux_arrays: Dict[str, "np.ndarray"], names: List[str]):
    """Stack (nlev+1, nloc) profile arrays into one (nfield, nlev+1, nloc) buffer (torch or numpy)."""
    import torch
    first = flux_arrays[names[0]]
    if isinstance(first, np.ndarray):
        return torch.from_numpy(np.stack([flux_arrays[n] for n in names]))
    return torch.stack([flux_arrays[n] for n in names])


def gather_profiles(local, counts: List[int], dst: int = 0, async_op: bool = False, group=None):
    """Gather per-rank (nfield, nlev+1, nloc_r) buffers on `dst`.

    Column counts may differ by one between ranks, so every rank pads to max(counts) columns; rank
    `dst` gets the list of per-rank buffers (padding still attached, see `assemble`) and, when
    `async_op`, the work handle to wait on.  Non-destination ranks get (None, work)."""
    import torch
    import torch.distributed as dist
    nmax = max(counts)
    if local.shape[-1] != nmax:
        pad = torch.zeros(*local.shape[:-1], nmax - local.shape[-1], dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=-1)
    local = local.contiguous()
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    work = dist.gather(local, bufs, dst=dst, group=group, async_op=async_op)
    return bufs, work


def assemble(bufs, counts: List[int]):
    """Concatenate the gathered buffers along the column axis, dropping the padding."""
    import torch
    return torch.cat([b[..., :n] for b, n in zip(bufs, counts)], dim=-1)


def gather_offsets(counts: List[int]):
    """Column offset of every rank's share in the gathered arrays (rank r's columns follow those of ranks 0 .. r-1)."""
    off, out = 0, []
    for n in counts:
        out.append(off)
        off += int(n)
    return out, off


def library_comm_init(rad, rank: int, world: int, bcast_id):
    """Join the library's own RCCL communicator (include/ecrad_hip.h: ecrad_hip_comm_id / ecrad_hip_comm_init).  `bcast_id(id_or_None)`
    is the host's way of handing rank 0's 128-byte id to every rank -- MPI_Bcast in an MPI host, torch.distributed.broadcast_object_list
    in bench.py, the identity at world size 1: it gets the id (bytes) on rank 0 and None elsewhere and returns the id on every rank."""
    import ctypes as C
    from . import abi
    ident = (C.c_ubyte * abi.COMM_ID_BYTES)()
    mine = None
    if rank == 0:
        if rad.lib.ecrad_hip_comm_id(rad.handle, ident) != 0:
            raise RuntimeError(rad.lib.ecrad_hip_last_error(rad.handle).decode())
        mine = bytes(ident)
    got = bcast_id(mine)
    ident = (C.c_ubyte * abi.COMM_ID_BYTES).from_buffer_copy(got)
    if rad.lib.ecrad_hip_comm_init(rad.handle, ident, rank, world) != 0:
        raise RuntimeError(rad.lib.ecrad_hip_last_error(rad.handle).decode())


def library_gather_profiles(rad, fields, counts: List[int], root: int = 0, rank: int = 0, device_pointers=None):
    """ecrad_hip_gather_profiles on host numpy arrays `fields` (each (n_rows, counts[rank]), C-contiguous float64): returns on `root` the
    list of (n_rows, sum(counts)) arrays, elsewhere None.  With `device_pointers` = (local pointers, global pointers or None, n_rows) the
    arrays are device memory and nothing is returned (the root's global arrays are the caller's).  The handle must be in a communicator
    (library_comm_init)."""
    import ctypes as C
    from . import abi
    lib, h = rad.lib, rad.handle
    world = len(counts)
    cnt = (C.c_int * world)(*[int(c) for c in counts])
    if device_pointers is not None:
        lp_list, gp_list, n_rows = device_pointers
        n = len(lp_list)
        lp = (C.c_void_p * n)(*lp_list)
        gp = (C.c_void_p * n)(*(gp_list if gp_list is not None else [None] * n))
        st = lib.ecrad_hip_gather_profiles(h, n, lp, gp, int(n_rows), int(counts[rank]), cnt, root, abi.MEM_DEVICE)
        if st != 0:
            raise RuntimeError(lib.ecrad_hip_last_error(h).decode())
        return None
    fields = [np.ascontiguousarray(f, dtype=np.float64) for f in fields]
    n_rows = fields[0].shape[0]
    _, total = gather_offsets(counts)
    out = [np.empty((n_rows, total)) for _ in fields] if rank == root else None
    n = len(fields)
    lp = (C.c_void_p * n)(*[f.ctypes.data for f in fields])
    gp = (C.c_void_p * n)(*([o.ctypes.data for o in out] if out is not None else [None] * n))
    if lib.ecrad_hip_gather_profiles(h, n, lp, gp, n_rows, int(counts[rank]), cnt, root, abi.MEM_HOST) != 0:
        raise RuntimeError(lib.ecrad_hip_last_error(h).decode())
    return out


def launch_ranks(argv: List[str], nproc: int, env: Optional[dict] = None, timeout: Optional[float] = None) -> int:
    """One process per GPU of this node: run `argv` `nproc` times with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set (what `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would set), stdout/stderr
    inherited.  Rank r owns GPU r and -- with `shard_range` -- the r-th contiguous column range, the reference driver's
    partitioning of its blocks (driver/ecrad_driver.F90:348-354).  Returns the largest exit status; when one rank fails or
    the timeout expires the others are terminated (exactly the PIDs started here), so a missing GPU is an error, not a hang."""
    import os
    import socket
    import subprocess
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(nproc):
        procs.append(subprocess.Popen(argv, env=dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")))
    t0, rc = time.monotonic(), 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                rc = max(rc, abs(code))
        failed = rc != 0 or (timeout is not None and time.monotonic() - t0 > timeout)
        if failed and live:
            for p in live:
                p.terminate()
            for p in live:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            return rc or 124
        time.sleep(0.05)
    return rc
