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
