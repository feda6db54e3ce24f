"""bench.py --gpus N started plainly starts its own N ranks (ecrad_amd/parallel.py: launch_ranks): the children get RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_*, a collective over them sees N ranks, a failing rank takes the launch down instead of
hanging it, and a node with fewer GPUs than asked for is a clear error."""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

from ecrad_amd.parallel import launch_ranks, shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(t)
    out = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out["ranks_seen_by_all_reduce"] = int(t.item())
    json.dump(out, open(os.path.join(sys.argv[1], f"rank{rank}.json"), "w"))
    dist.barrier(); dist.destroy_process_group()
""")


def test_launch_ranks_sets_the_rendezvous_environment(tmp_path):
    rc = launch_ranks([sys.executable, "-c", CHILD, str(tmp_path)], 2, timeout=300)
    assert rc == 0
    seen = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r, d in enumerate(seen):
        assert d["RANK"] == str(r) and d["LOCAL_RANK"] == str(r) and d["WORLD_SIZE"] == "2"
        assert d["MASTER_ADDR"] == "127.0.0.1" and d["MASTER_PORT"] == seen[0]["MASTER_PORT"]
        assert d["ranks_seen_by_all_reduce"] == 2
    # the ranks' column ranges are the reference driver's contiguous blocks (driver/ecrad_driver.F90:348-354)
    assert [shard_range(200000, r, 2) for r in range(2)] == [(1, 100000), (100001, 200000)]


def test_a_failing_rank_ends_the_launch_instead_of_hanging_it():
    child = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(600)\n"
    t0 = time.monotonic()
    rc = launch_ranks([sys.executable, "-c", child], 2, timeout=120)
    assert rc == 3 and time.monotonic() - t0 < 60


def test_timeout_ends_the_launch():
    t0 = time.monotonic()
    rc = launch_ranks([sys.executable, "-c", "import time; time.sleep(600)"], 2, timeout=2)
    assert rc == 124 and time.monotonic() - t0 < 60


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="a GPU is visible")
def test_bench_asked_for_two_gpus_without_any_fails_clearly():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "no GPU visible" in p.stderr


@pytest.mark.gpu
def test_bench_asked_for_more_gpus_than_the_node_has_fails_clearly():
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and f"--gpus {n} asked for" in p.stderr
    # ... and the same from a rank that a launcher started without a GPU of its own
    env = dict(os.environ, RANK="1", LOCAL_RANK=str(n - 1), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 2 and "has no GPU" in p.stderr


@pytest.mark.gpu
def test_bench_two_ranks_end_to_end_on_a_shared_gpu():
    """The N>1 code of bench.py (own launcher, barrier + max-over-ranks timing, the gather on rank 0) run for real with two
    ranks that share this box's GPU over gloo (ECRAD_BENCH_TEST_SHARED_GPU: RCCL refuses two ranks on one device)."""
    env = dict(os.environ, ECRAD_BENCH_TEST_SHARED_GPU="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--ncol", "4096",
                        "--headline-only", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak" and "test_shared_gpu" in d
    assert d["value"] > 0 and d["value_with_gather"] > 0 and d["value_with_gather"] <= d["value"] * 1.05
    assert d["ms_per_step_ranks"]["min"] <= d["ms_per_step_ranks"]["max"] == d["ms_per_step"]
    assert d["gathered"]["ranks"] == 2 and d["gathered"]["shape_per_rank"][-1] == 4096


@pytest.mark.gpu
def test_bench_pool_mode_over_eight_device_slots_on_this_gpu():
    """`bench.py --gpus 8 --threads-per-process 16` -- ONE process whose 16 host threads spread blocks of host arrays over the
    pool's eight devices -- run for real with ECRAD_HIP_FAKE_DEVICES=8 (eight device slots on this box's GPU): the compact line
    parses, says what it is, every slot served blocks and the blocks of the eight shards are identical."""
    env = dict(os.environ, ECRAD_HIP_FAKE_DEVICES="8")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--threads-per-process", "16", "--steps", "2", "--warmup", "1",
                        "--ncol", "4096", "--block-columns", "1024", "--workload", "tripleclouds_ecckd32"], capture_output=True, text=True, timeout=1200, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and "test_fake_devices" in d and d["blocks_identical_across_devices"] is True and d["value"] > 0
    assert d["pool"]["n_devices"] == 8 and len(d["pool"]["calls_on_device"]) == 8 and all(v > 0 for v in d["pool"]["calls_on_device"].values()), d["pool"]


@pytest.mark.gpu
def test_bench_library_gather_leg_on_a_one_rank_communicator():
    """The leg of bench.py that puts the profiles together with the LIBRARY's own RCCL gather (bench.py: library_gather_leg; what runs at
    every N > 1), forced at world size 1: the communicator forms, the timed steps run, rank 0's share arrives bit for bit."""
    env = dict(os.environ, ECRAD_BENCH_FORCE_LIBRARY_GATHER="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--ncol", "4096", "--headline-only",
                        "--no-cpu-baseline", "--no-host-mode"], capture_output=True, text=True, timeout=1200, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    lg = d["library_gather"]
    assert "error" not in lg and lg["value"] > 0 and lg["own_share_intact"] is True, lg


@pytest.mark.gpu
def test_bench_line_survives_a_library_gather_that_cannot_form():
    """Two ranks on ONE GPU with the library-gather leg forced on: RCCL refuses a communicator with two ranks on one device, the leg fails on
    both ranks -- and the run still prints its one line, with the failure in `library_gather`, every other field as usual, exit status 0
    (bench.py: LIBRARY_GATHER_FAILED -- the ranks leave without the final barrier).  What a communicator that cannot form on a real
    multi-GPU node would do to the record."""
    env = dict(os.environ, ECRAD_BENCH_TEST_SHARED_GPU="1", ECRAD_BENCH_TEST_LIBRARY_GATHER_ANYWAY="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--ncol", "4096",
                        "--headline-only", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["value_with_gather"] > 0
    assert "error" in d["library_gather"], d["library_gather"]
