"""N>1 path on CPU: two gloo ranks shard the columns (no data-path collective), each runs its range,
flux profiles are gathered on rank 0 and must equal the single-process result bit for bit.
The per-rank compute uses the oracle (tests may; the product never does)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ecrad_amd.parallel import assemble, gather_profiles, pack_profiles, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["lw_up", "lw_dn", "sw_up", "sw_dn", "sw_dn_direct", "lw_up_clear", "sw_dn_clear", "lw_derivatives"]


def test_shard_range_partitions_exactly():
    for ncol in (1, 7, 32, 100000, 10_000_001):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(ncol, r, world) for r in range(world)]
            assert ranges[0][0] == 1 and ranges[-1][1] == ncol
            for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
                assert b0 == a1 + 1
            sizes = [b - a + 1 for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, ncol_total, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import make_config, run_case
    from oracle import pyoracle
    i0, i1 = shard_range(ncol_total, rank, world)
    flux, _, _ = run_case(make_config("Tripleclouds"), pyoracle.backend, columns=(i0, i1))
    local = {n: np.ascontiguousarray(flux.arrays[n][:, i0 - 1:i1]) for n in NAMES}
    counts = [shard_range(ncol_total, r, world)[1] - shard_range(ncol_total, r, world)[0] + 1 for r in range(world)]
    bufs, work = gather_profiles(pack_profiles(local, NAMES), counts, dst=0, async_op=True)
    work.wait()
    if rank == 0:
        np.save(os.path.join(tmpdir, "gathered.npy"), assemble(bufs, counts).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ncol_total", [32, 31])
def test_two_rank_gloo_sharded_run_equals_single_process(tmp_path, oracle_lib, ncol_total):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, ncol_total, str(tmp_path)), nprocs=2, join=True)
    gathered = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    from helpers import make_config, run_case
    flux, _, _ = run_case(make_config("Tripleclouds"), oracle_lib.backend, columns=(1, ncol_total))
    want = np.stack([flux.arrays[n][:, :ncol_total] for n in NAMES])
    assert gathered.shape == want.shape
    assert np.array_equal(gathered, want)
